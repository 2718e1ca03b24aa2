#!/bin/bash
# usage (GPU box): tools/gpu_dec_pmc.sh "<counters of pass 1>" ["<counters of pass 2>" ...] -- SQ counters of bz_chain (tools/dec_time.py, 100 MB)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; : > gpurun_out/dec_pmc.log
for c in "$@"; do
  o=gpurun_out/dec_pmc_tmp; rm -rf $o; mkdir -p $o
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $o/p -- python3 tools/dec_time.py ${AB_MB:-100} > $o/t.json 2> $o/err || { tail -5 $o/err; exit 1; }
  python3 - "$o" <<'PY' | tee -a gpurun_out/dec_pmc.log
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/p/*/*counter_collection.csv')[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0][-28:]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k in acc:
    if 'bz_chain' in k or 'bz_group' in k or 'ib_walk' in k:
        print(k, {c: round(v / 6) for c, v in acc[k].items()})
PY
done
rm -rf gpurun_out/dec_pmc_tmp
