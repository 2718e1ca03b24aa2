#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_quick.sh <tag> [pytest -k expr]
# runs the BWT/golden stage tests, then a profiled 2-step bench, and prints the per-kernel summary
set -o pipefail
tag=${1:-q}; kexpr=${2:-bwt or golden}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_stages.py -x -q -m gpu -k "$kexpr" > gpurun_out/${tag}_test.log 2>&1
rc=$?; tail -3 gpurun_out/${tag}_test.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/prof_${tag}.log 2>&1 || { tail -20 gpurun_out/prof_${tag}.log; exit 1; }
f=$(ls gpurun_out/prof_${tag}/*/*kernel_trace.csv)
python3 tools/prof_summary.py $f 3 ${3:-12}
grep -o '"bit_exact[^}]*' gpurun_out/prof_${tag}.log
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for k in ('bwt_tile_sort','bwt_apply','bwt_gather_keys'):
    ts=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if k in r['Kernel_Name']]
    n=len(ts)//3
    print(k,[round(x) for x in ts[-n:]])
PY
