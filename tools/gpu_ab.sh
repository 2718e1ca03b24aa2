#!/bin/bash
# usage: tools/gpu_ab.sh "<ENV=val ...>" ["<ENV=val ...>" ...] -- A/B of bench stage times under different env settings
cd "$GRAFT_REPO_ROOT" || exit 1
for e in "$@"; do
  echo "== $e"
  env $(echo $e | sed "s#=ab/#=$GRAFT_REPO_ROOT/ab/#") timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"stage_ms_per_step": {[^}]*}' || exit 1
done
