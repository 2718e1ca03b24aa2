#!/usr/bin/env python3
"""Debug helper: CJS_BWTC_CHECK=1 python tools/bwtc_check.py <seed> <level> -- one stress case through BWTC.compressFile
(the library then compares the chunk-parallel model kernel with the one-symbol-at-a-time kernel step by step)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import numpy as np
from test_gpu_streams import _mixed_input
pkg = importlib.import_module("compressjs-flattened_amd")
seed, level = int(sys.argv[1]), int(sys.argv[2])
data = _mixed_input(seed)
print("n =", data.size, flush=True)
pkg.BWTC.compressFile(data, None, level)
