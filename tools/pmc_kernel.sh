#!/bin/bash
# usage (GPU box): tools/pmc_kernel.sh <kernel-substr> "<counters>" -- <python script args...> : prints the counters of the kernel's launches
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
k=$1; c=$2; shift 3
rm -rf gpurun_out/pmck
timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmck -- python3 "$@" > gpurun_out/pmck.log 2>&1 || { tail -5 gpurun_out/pmck.log; exit 1; }
python3 - "$(ls gpurun_out/pmck/*/*counter_collection.csv)" "$k" <<'PY'
import csv,sys
acc={}
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        a=acc.setdefault(r["Counter_Name"],[0,0.0]); a[0]+=1; a[1]+=float(r["Counter_Value"])
for k,v in sorted(acc.items()): print("%-28s launches %d  total %.4g  per launch %.4g" % (k,v[0],v[1],v[1]/v[0]))
PY
