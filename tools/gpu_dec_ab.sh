#!/bin/bash
# usage (GPU box): tools/gpu_dec_ab.sh "<ENV=val ...>" ... -- bz_chain kernel time and decompress wall time per arm
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; : > gpurun_out/dec_ab.log
for e in "$@"; do
  o=gpurun_out/dec_ab_tmp; rm -rf $o; mkdir -p $o
  env $e timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/kt -- python3 tools/dec_time.py ${AB_MB:-100} > $o/t.json 2> $o/err || { tail -5 $o/err; exit 1; }
  echo "== $e $(grep -o '"decompress_ms_median": [0-9.]*\|"round_trip": [a-z]*' $o/t.json | tr '\n' ' ') $(python3 -c "
import csv,sys,glob
for r in csv.DictReader(open(glob.glob('$o/kt/*/*kernel_stats.csv')[0])):
    if 'bz_chain' in r['Name']: print('bz_chain avg us', round(float(r['AverageNs'])/1e3,1))
")" | tee -a gpurun_out/dec_ab.log
done
rm -rf gpurun_out/dec_ab_tmp
