#!/bin/bash
# usage (GPU box, repo root): tools/gpu_round_profile.sh <tag>  -- bench line + kernel stats + PMC traffic for profiles/
# (bench.py runs warmup + steps + ONE extra untimed step for the stage times: the per-step divisors below count it)
set -o pipefail
tag=${1:-v5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/round_$tag; mkdir -p $o
timeout -k 10 400 python3 bench.py --steps 5 --warmup 2 > $o/bench.json 2> $o/bench.err || { tail -5 $o/bench.err; exit 1; }
cut -c1-400 $o/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/kt -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $o/bench_under_rocprof.json 2> $o/kt.err || { tail -5 $o/kt.err; exit 1; }
cp $(ls $o/kt/*/*kernel_stats.csv) $o/kernel_stats.csv
python3 tools/prof_summary.py $(ls $o/kt/*/*kernel_trace.csv) 4 45 > $o/kernel_summary.txt
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/pmc_f -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-verify --no-extra > $o/pmc_f.log 2>&1 || { tail -5 $o/pmc_f.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/pmc_w -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-verify --no-extra > $o/pmc_w.log 2>&1 || { tail -5 $o/pmc_w.log; exit 1; }
cp $(ls $o/pmc_f/*/*counter_collection.csv) $o/fetch_size_counter_collection.csv
cp $(ls $o/pmc_w/*/*counter_collection.csv) $o/write_size_counter_collection.csv
python3 tools/pmc_summary.py $o/fetch_size_counter_collection.csv $o/write_size_counter_collection.csv 3 100000000 > $o/pmc_traffic.json
grep rs_scatter_ $o/pmc_traffic.json
rm -rf $o/kt $o/pmc_f $o/pmc_w
head -12 $o/kernel_summary.txt
