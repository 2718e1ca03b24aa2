import sys, os
sys.path.insert(0, 'tests')
import torch, numpy as np, support, recipes
hip = support.HipLib()
n = int(sys.argv[1])
d = recipes.textgen(n, 1)
rc, comp = hip.bzip2_compress(d, 9)
print("compress", rc, comp.size, flush=True)
hip.L.cjs_trim()
rc, back = hip.bzip2_decompress(comp)
print("decompress", rc, None if back is None else (back.size, bool(np.array_equal(back, d))), hip.last_error_detail(), flush=True)
