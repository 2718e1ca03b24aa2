#!/usr/bin/env python3
"""Times Bzip2.decompressFile of the 100 MB level-9 bench stream through the host-buffer C ABI (CJS_DEBUG prints phases)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import recipes
pkg = importlib.import_module("compressjs-flattened_amd")
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 100
lvl = int(sys.argv[2]) if len(sys.argv) > 2 else 9
data = recipes.textgen(mb * 1000000, 1)
dbg = os.environ.pop("CJS_DEBUG", None)
c = pkg.Bzip2.compressFile(data, None, lvl)
if dbg: os.environ["CJS_DEBUG"] = dbg
for i in range(3):
    t0 = time.perf_counter(); back = pkg.Bzip2.decompressFile(c); dt = time.perf_counter() - t0
    print("decompress %d MB level %d: %.1f ms  %.1f MB/s" % (mb, lvl, dt * 1e3, data.size / dt / 1e6), flush=True)
assert np.array_equal(back, data)
