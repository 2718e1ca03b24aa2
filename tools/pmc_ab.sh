#!/bin/bash
# usage (GPU box): tools/pmc_ab.sh "<kernel substrings, | separated>" "<ENV=val ...>" ... -- FETCH_SIZE and WRITE_SIZE per kernel (summed over one bench step)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
ks=$1; shift
for e in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmcab
    env $e timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmcab -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-extra > gpurun_out/pmcab.log 2>&1 || { tail -5 gpurun_out/pmcab.log; exit 1; }
    python3 - "$(ls gpurun_out/pmcab/*/*counter_collection.csv)" "$ks" "$e" "$c" <<'PY'
import csv,sys
acc={}
keys=sys.argv[2].split('|')
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"].replace('cjs::','').replace('void ','')
    for k in keys:
        if k in n:
            nm=n.split('(')[0][:34]
            a=acc.setdefault(nm,[0,0.0]); a[0]+=1; a[1]+=float(r["Counter_Value"])
for k,v in sorted(acc.items()): print("%-40s %-11s %-34s launches %3d  total %8.3f GB (raw counter; FETCH is half of wide reads)" % (sys.argv[3][:40], sys.argv[4], k, v[0], v[1]*1024/1e9))
PY
  done
done
rm -rf gpurun_out/pmcab
