#!/usr/bin/env python3
"""Decompress timing (host-buffer C ABI, wall clock, median of 5 after a warm-up): python tools/dec_time.py <MB> [level]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import importlib
import recipes, support
sys.path.insert(0, ROOT)
pkg = importlib.import_module("compressjs-flattened_amd")      # zero-copy adoption of the result (the test wrapper copies it)
hip = support.HipLib()
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 100
level = int(sys.argv[2]) if len(sys.argv) > 2 else 9
n = (1 << 30) if mb == 1024 else mb * 1000000
d = recipes.textgen(n, 1)
rc, comp = hip.bzip2_compress(d, level)
assert rc == 0
hip.L.cjs_trim()
ts = []
for i in range(6):
    t0 = time.perf_counter()
    back = pkg.Bzip2.decompressFile(comp)
    ts.append(time.perf_counter() - t0)
ok = bool(np.array_equal(back, d))
med = float(np.median(ts[1:]))
print('{"input_bytes": %d, "level": %d, "compressed": %d, "decompress_ms_median": %.2f, "MBps": %.1f, "round_trip": %s, "all_ms": %s}'
      % (n, level, comp.size, med * 1e3, n / med / 1e6, str(ok).lower(), [round(t * 1e3, 1) for t in ts]))
