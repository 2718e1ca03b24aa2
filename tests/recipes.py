"""Input recipes — Python mirror of build() in tests/golden/make_golden.js."""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "data")
_textgen = None


def textgen(n, seed):
    global _textgen
    if _textgen is None:
        path = os.path.join(ROOT, "tools", "libcjs_textgen.so")
        if not os.path.exists(path):
            raise RuntimeError("tools/libcjs_textgen.so missing: run `make textgen`")
        _textgen = ctypes.CDLL(path)
        _textgen.cjs_textgen.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]
        _textgen.cjs_textgen.restype = ctypes.c_int
    out = np.empty(max(n, 1), dtype=np.uint8)
    _textgen.cjs_textgen(out.ctypes.data, n, seed)
    return out[:n]


def xorshift_bytes(n, seed, mask=255, add=0):
    out = np.empty(n, dtype=np.uint8)
    s = seed & 0xFFFFFFFF
    # vectorising xorshift is awkward; n is small (<= 300k) so a tight loop over python ints is fine
    buf = bytearray(n)
    for i in range(n):
        s ^= (s << 13) & 0xFFFFFFFF
        s ^= s >> 17
        s ^= (s << 5) & 0xFFFFFFFF
        buf[i] = (((s >> 24) & mask) + add) & 255
    out[:] = np.frombuffer(bytes(buf), dtype=np.uint8)
    return out


def build(recipe):
    kind = recipe["kind"]
    if kind == "file":
        return np.fromfile(os.path.join(DATA, recipe["name"]), dtype=np.uint8)
    if kind == "textgen":
        return textgen(recipe["n"], recipe["seed"])
    if kind == "repeat":
        unit = np.frombuffer(bytes.fromhex(recipe["unit_hex"]), dtype=np.uint8)
        n = recipe["n"]
        if n == 0:
            return np.empty(0, dtype=np.uint8)
        reps = -(-n // len(unit))
        return np.tile(unit, reps)[:n].copy()
    if kind == "xorshift":
        return xorshift_bytes(recipe["n"], recipe["seed"], recipe.get("mask", 255), recipe.get("add", 0))
    if kind == "range256":
        return (np.arange(recipe["n"], dtype=np.int64) & 255).astype(np.uint8)
    if kind == "concat":
        parts = [build(p) for p in recipe["parts"]]
        return np.concatenate(parts) if parts else np.empty(0, dtype=np.uint8)
    raise ValueError(recipe)
