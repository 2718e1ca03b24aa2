#!/bin/bash
# usage: tools/gpu_dbg.sh <kernel-substr> "<ENV=val>"... : first-launch durations of a kernel under env variants (results may be wrong)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
k=$1; shift
i=0
for e in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/dbg_$i; env $e timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/dbg_$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify > gpurun_out/dbg_$i.log 2>&1
  python3 - "$(ls gpurun_out/dbg_$i/*/*kernel_trace.csv)" "$k" "$e" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
ts=[round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in rows if sys.argv[2] in r['Kernel_Name']]
print(sys.argv[3], ts[:6])
PY
done
