#!/usr/bin/env python3
"""BWTC.compressFile level 9 on the 2^30-byte golden input (BASELINE configs[3] on one GPU): wall clock + golden check."""
import importlib, os, sys, time, hashlib, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import recipes
pkg = importlib.import_module("compressjs-flattened_amd")
g = json.load(open(os.path.join(ROOT, "tests/golden/golden_big_bwtc_9_1g.json")))["cases"][0]
data = recipes.build(g["recipe"])
for i in range(2):
    t0 = time.perf_counter(); c = pkg.BWTC.compressFile(data, None, 9); dt = time.perf_counter() - t0
    ok = c.size == g["out_len"] and hashlib.sha256(c.tobytes()).hexdigest() == g["out_sha256"]
    print("BWTC -9 %d bytes: %.1f ms  %.1f MB/s golden %s" % (data.size, dt * 1e3, data.size / dt / 1e6, ok), flush=True)
