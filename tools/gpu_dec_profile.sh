#!/bin/bash
# usage (GPU box, repo root): tools/gpu_dec_profile.sh <tag> [MB]  -- kernel stats of Bzip2.decompressFile (tools/dec_time.py: 6 calls)
set -o pipefail
tag=${1:-d1}; mb=${2:-100}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/dec_$tag; mkdir -p $o
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/kt -- python3 tools/dec_time.py $mb > $o/dec_time.json 2> $o/kt.err || { tail -5 $o/kt.err; exit 1; }
python3 tools/prof_summary.py $(ls $o/kt/*/*kernel_trace.csv) 6 40 > $o/kernel_summary.txt
cp $(ls $o/kt/*/*kernel_stats.csv) $o/kernel_stats.csv
rm -rf $o/kt
tail -1 $o/dec_time.json; head -32 $o/kernel_summary.txt
