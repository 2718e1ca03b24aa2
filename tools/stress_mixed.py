#!/usr/bin/env python3
"""One-off stress: random mixtures (see tests/test_gpu_streams.py::_mixed_input) through the HIP library vs the oracle.
usage: stress_mixed.py <first seed> <count>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import numpy as np
import support
from test_gpu_streams import _mixed_input
hip, orc = support.HipLib(), support.Oracle()
s0, cnt = int(sys.argv[1]), int(sys.argv[2])
bad = 0
t0 = time.time()
for seed in range(s0, s0 + cnt):
    data = _mixed_input(seed)
    level = 1 + seed % 9
    rc, want = orc.bzip2_compress(data, level)
    rc2, got = hip.bzip2_compress(data, level)
    ok = rc == 0 and rc2 == 0 and np.array_equal(got, want)
    rc3, back = hip.bzip2_decompress(got) if rc2 == 0 else (-1, None)
    ok = ok and rc3 == 0 and np.array_equal(back, data)
    rc, want = orc.bwtc_compress(data, level)
    rc2, got = hip.bwtc_compress(data, level)
    ok2 = rc == 0 and rc2 == 0 and np.array_equal(got, want)
    if not (ok and ok2):
        bad += 1
        print("MISMATCH seed %d level %d n %d (bzip2 %s, bwtc %s)" % (seed, level, data.size, ok, ok2), flush=True)
    if (seed - s0) % 10 == 9:
        print("  %d done, %d bad, %.0f s" % (seed - s0 + 1, bad, time.time() - t0), flush=True)
print("stress: %d cases, %d mismatches" % (cnt, bad))
sys.exit(1 if bad else 0)
