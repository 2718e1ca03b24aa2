#!/usr/bin/env python3
"""bzip2 -9 round trip of an input above 4 GiB through the host-buffer C ABI (compress: block ranges in sequence,
decompress: batches): python tools/big_roundtrip.py [bytes]"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import recipes, support
n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 32) + 123456789
hip = support.HipLib()
t0 = time.time(); d = recipes.textgen(n, 1); t1 = time.time()
h_in = hashlib.sha256(d.tobytes()).hexdigest()
t2 = time.time(); rc, comp = hip.bzip2_compress(d, 9); t3 = time.time()
print("gen %.1f s, compress rc %d: %d -> %d bytes in %.2f s" % (t1 - t0, rc, n, comp.size if rc == 0 else -1, t3 - t2), flush=True)
assert rc == 0
hip.L.cjs_trim()
t4 = time.time(); rc, back = hip.bzip2_decompress(comp); t5 = time.time()
print("decompress rc %d: %d bytes in %.2f s" % (rc, back.size if rc == 0 else -1, t5 - t4), flush=True)
assert rc == 0 and back.size == n, (rc, back.size if back is not None else None)
if hashlib.sha256(back.tobytes()).hexdigest() != h_in:
    diff = np.flatnonzero(back != d)
    print("MISMATCH: %d bytes differ, first at %d, last at %d" % (diff.size, int(diff[0]), int(diff[-1])), flush=True)
    # which GiB-aligned regions are affected
    print("regions (256 MiB units):", sorted(set((diff >> 28).tolist()))[:40], flush=True)
    rc2, back2 = support.Oracle().bzip2_decompress(comp[: 400000000]) if False else (0, None)
    sys.exit(1)
# the first GiB of the stream is the reference golden's input: its blocks are the golden's blocks
print('{"big_roundtrip_bytes": %d, "compressed": %d, "compress_s": %.2f, "decompress_s": %.2f, "round_trip": true}' % (n, comp.size, t3 - t2, t5 - t4))
