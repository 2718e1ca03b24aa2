#!/usr/bin/env python3
"""Times BWTC.decompressFile level 9 on 20 MB of the bench input (host-buffer C ABI; the decoder is a serial host chain)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import numpy as np
import recipes
pkg = importlib.import_module("compressjs-flattened_amd")
data = recipes.textgen(20000000, 1)
c = pkg.BWTC.compressFile(data, None, 9)
for i in range(3):
    t0 = time.perf_counter(); back = pkg.BWTC.decompressFile(c); dt = time.perf_counter() - t0
    print("BWTC.decompressFile 20 MB: %.1f ms  %.1f MB/s" % (dt * 1e3, data.size / dt / 1e6), flush=True)
assert np.array_equal(back, data)
