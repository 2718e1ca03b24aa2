#!/bin/bash
# usage (GPU box): tools/gpu_sq_counters.sh -- SQ counters per kernel over one compress step of bench.py (two passes of 8 counters)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; : > gpurun_out/sq_counters.log
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; do
  o=gpurun_out/sq_tmp; rm -rf $o; mkdir -p $o
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $o/p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-extra > $o/log 2>&1 || { tail -5 $o/log; exit 1; }
  python3 - "$o" <<'PY' | tee -a gpurun_out/sq_counters.log
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/p/*/*counter_collection.csv')[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('cjs::', '')[:34]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
keys = sorted(acc, key=lambda k: -acc[k].get('SQ_WAVE_CYCLES', acc[k].get('SQ_INSTS_VALU', 0)))[:16]
for k in keys:
    print(k.ljust(34), ' '.join('%s=%.3g' % (c.replace('SQ_', ''), v) for c, v in sorted(acc[k].items())))
PY
done
rm -rf gpurun_out/sq_tmp
