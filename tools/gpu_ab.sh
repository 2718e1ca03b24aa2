#!/bin/bash
# usage: tools/gpu_ab.sh "<ENV=val ...>" ["<ENV=val ...>" ...] -- A/B of bench step times under different env settings on ONE box
# (each arm is checked bit-exact against the reference golden by bench.py's verify, outside the timed region); AB_ARGS adds bench flags
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; : > gpurun_out/ab.log
for e in "$@"; do
  echo "== $e" | tee -a gpurun_out/ab.log
  env $e timeout -k 10 200 python bench.py --steps ${AB_STEPS:-10} --warmup 2 --no-cpu-baseline --no-extra $AB_ARGS 2>gpurun_out/ab_last.err | grep -o '"ms_per_step": [0-9.]*\|"median_step_ms": [0-9.]*\|"stage_ms_per_step": {[^}]*}\|"bit_exact_vs_reference_js": [a-z]*\|"bwt_rounds": [0-9]*' | tr '\n' ' ' | tee -a gpurun_out/ab.log || { tail -5 gpurun_out/ab_last.err; exit 1; }
  echo | tee -a gpurun_out/ab.log
done
