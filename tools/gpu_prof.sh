#!/bin/bash
# usage (GPU box, repo root): tools/gpu_prof.sh <tag> "<ENV=val ...>" [extra bench args] -- kernel-trace of a 2-step bench
set -o pipefail
tag=$1; e=${2:-X=0}; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/prof_${tag}
env $e timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra "$@" > gpurun_out/prof_${tag}.log 2>&1 || { tail -20 gpurun_out/prof_${tag}.log; exit 1; }
f=$(ls gpurun_out/prof_${tag}/*/*kernel_trace.csv)
{ echo "== $tag ($e)"; python3 tools/prof_summary.py $f 4 ${TOPN:-16}; python3 tools/prof_timeline.py $f; } > gpurun_out/prof_${tag}.txt
head -20 gpurun_out/prof_${tag}.txt
cp $f gpurun_out/prof_${tag}_trace.csv
rm -rf gpurun_out/prof_${tag}
