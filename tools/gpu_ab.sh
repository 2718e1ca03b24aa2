#!/bin/bash
# usage: tools/gpu_ab.sh "<ENV=val ...>" ["<ENV=val ...>" ...] -- A/B of bench stage times under different env settings
# (each arm is checked bit-exact against the reference golden by bench.py's verify, outside the timed region)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; : > gpurun_out/ab.log
for e in "$@"; do
  echo "== $e" | tee -a gpurun_out/ab.log
  env $(echo $e | sed "s#=ab/#=$GRAFT_REPO_ROOT/ab/#") timeout -k 10 200 python bench.py --steps ${AB_STEPS:-3} --warmup 1 --no-cpu-baseline --no-extra 2>gpurun_out/ab_last.err | grep -o '"ms_per_step": [0-9.]*\|"stage_ms_per_step": {[^}]*}\|"bit_exact_vs_reference_js": [a-z]*\|"bwt_rounds": [0-9]*' | tee -a gpurun_out/ab.log || { tail -5 gpurun_out/ab_last.err; exit 1; }
done
