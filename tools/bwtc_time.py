#!/usr/bin/env python3
"""Times BWTC.compressFile level 9 on the 100 MB bench input (host-buffer C ABI)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import recipes
pkg = importlib.import_module("compressjs-flattened_amd")
data = recipes.textgen(100000000, 1)
for i in range(3):
    t0 = time.perf_counter(); c = pkg.BWTC.compressFile(data, None, 9); dt = time.perf_counter() - t0
    print("BWTC.compressFile: %.1f ms  %.1f MB/s (%d bytes out)" % (dt * 1e3, data.size / dt / 1e6, c.size), flush=True)
