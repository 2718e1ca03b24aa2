"""compressjs-flattened_amd — MI355X-native block-sorting core behind the compressjs Bzip2/BWTC API.

The product is the HIP library `libcjs_hip.so` (C ABI: include/cjs_hip.h) plus the JavaScript fronts
under js/ (Node + N-API).  This Python module is plumbing for tests and bench.py: a ctypes binding of
the same C ABI and `Bzip2` / `BWTC` objects that mirror the reference's method names
(`compressFile`, `decompressFile`; J/Bzip2_joined_.js:2198-2253, J/BWTC_joined_.js:1696-1698,1827).
There is no CPU fallback: if the library or a GPU is missing every call raises.
"""
import ctypes
import os
import weakref

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CJS_HIP_LIB") or os.path.join(PKG_DIR, "libcjs_hip.so")   # same override as the N-API addon
u8p = ctypes.POINTER(ctypes.c_uint8)


class CjsError(Exception):
    def __init__(self, code, msg):
        super().__init__("%s (code %d)" % (msg, code))
        self.errorCode = code


class Stats(ctypes.Structure):
    _fields_ = [("ms_total", ctypes.c_double), ("ms_rle1", ctypes.c_double), ("ms_bwt", ctypes.c_double),
                ("ms_mtf", ctypes.c_double), ("ms_huff", ctypes.c_double), ("ms_pack", ctypes.c_double),
                ("ms_bwt_dominant", ctypes.c_double), ("bwt_dominant_launches", ctypes.c_uint64),
                ("bwt_dominant_bytes", ctypes.c_uint64), ("blocks", ctypes.c_uint64), ("bytes_in", ctypes.c_uint64),
                ("bytes_out", ctypes.c_uint64), ("bwt_rounds", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class ShardMeta(ctypes.Structure):
    """cjs_shard_meta: what the ranks of a multi-GPU job exchange between the block phase and the pack phase (32 bytes)."""
    _fields_ = [("bits", ctypes.c_uint64), ("total_blocks", ctypes.c_uint64), ("first_block", ctypes.c_uint64),
                ("blocks", ctypes.c_uint32), ("crc_fold", ctypes.c_uint32)]


_lib = None


def load_library():
    """dlopen the HIP C-ABI library; raises if it has not been built (`python __graft_entry__.py`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libcjs_hip.so not built: run `make hip` / `python __graft_entry__.py` (no CPU fallback exists)")
    try:
        # torch wheels bundle their own libamdhip64: load it first so that both share ONE HIP runtime in this
        # process (two copies of the runtime cannot both own the device)
        import torch  # noqa: F401
    except Exception:
        pass
    L = ctypes.CDLL(LIB_PATH)
    S, I, V = ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p
    PS, PP = ctypes.POINTER(S), ctypes.POINTER(u8p)
    L.cjs_bzip2_compress.argtypes = [u8p, S, I, PP, PS, V]
    L.cjs_bwtc_compress.argtypes = [u8p, S, I, PP, PS, V]
    L.cjs_bzip2_decompress.argtypes = [u8p, S, I, PP, PS, V]
    L.cjs_bwtc_decompress.argtypes = [u8p, S, PP, PS, V]
    L.cjs_free.argtypes = [V]
    L.cjs_free.restype = None
    L.cjs_trim.argtypes = []
    L.cjs_trim.restype = None
    L.cjs_strerror.argtypes = [I]
    L.cjs_strerror.restype = ctypes.c_char_p
    L.cjs_version.restype = ctypes.c_char_p
    L.cjs_last_error_detail.restype = ctypes.c_char_p
    L.cjs_device_count.restype = I
    L.cjs_ctx_create.argtypes = [ctypes.POINTER(V), I, S, I]
    L.cjs_ctx_create_sharded.argtypes = [ctypes.POINTER(V), I, S, ctypes.c_long, I]
    L.cjs_ctx_destroy.argtypes = [V]
    L.cjs_ctx_destroy.restype = None
    L.cjs_ctx_set_stage_times.argtypes = [V, I]
    L.cjs_ctx_set_stage_times.restype = None
    L.cjs_bzip2_compress_device.argtypes = [V, V, S, I, V, S, PS, ctypes.POINTER(Stats)]
    L.cjs_bzip2_compress_device_range.argtypes = [V, V, S, I, ctypes.c_long, ctypes.c_long, V, S, ctypes.POINTER(ctypes.c_uint64),
                                                  V, ctypes.c_long, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(Stats)]
    L.cjs_bzip2_shard_share_bytes.argtypes = [S, I]
    L.cjs_bzip2_shard_share_bytes.restype = S
    L.cjs_bzip2_shard_tiles.argtypes = [V, V, S, I, I, V]
    L.cjs_bzip2_shard_blocks.argtypes = [V, V, S, I, I, I, V, ctypes.POINTER(ShardMeta), ctypes.POINTER(Stats)]
    L.cjs_bzip2_shard_pack.argtypes = [V, I, I, I, ctypes.POINTER(ShardMeta), V, S, PS, PS, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        L = load_library()
        detail = L.cjs_last_error_detail().decode()          # the reference's optDetail (J/Bzip2_joined_.js:1385-1391)
        raise CjsError(rc, L.cjs_strerror(rc).decode() + (": " + detail if detail else ""))


def _coerce_input(data):
    # Util.coerceInputStream (J/Bzip2_joined_.js:178-220): anything indexable with a length
    if isinstance(data, (bytes, bytearray, memoryview)):
        return np.frombuffer(bytes(data), dtype=np.uint8)
    return np.ascontiguousarray(data, dtype=np.uint8)


def _adopt(out, n):
    """the library's malloc'd result as a uint8 ndarray without a copy; cjs_free runs when the array is collected
    (what the N-API addon does with an external ArrayBuffer and a finalizer)"""
    L = load_library()
    if not n:
        L.cjs_free(out)
        return np.empty(0, np.uint8)
    addr = ctypes.cast(out, ctypes.c_void_p).value
    buf = (ctypes.c_uint8 * n).from_address(addr)
    weakref.finalize(buf, L.cjs_free, ctypes.c_void_p(addr))
    return np.frombuffer(buf, dtype=np.uint8)


def _stream_call(fn, data, *mid):
    data = _coerce_input(data)
    keep = data if data.size else np.zeros(1, dtype=np.uint8)
    out, out_n = u8p(), ctypes.c_size_t(0)
    _check(fn(keep.ctypes.data_as(u8p), data.size, *mid, ctypes.byref(out), ctypes.byref(out_n), None))
    return _adopt(out, out_n.value)


class Bzip2:
    """Same surface as the reference's `Bzip2` object for the hot path (returns a uint8 ndarray)."""

    @staticmethod
    def compressFile(input, output=None, props=None):
        # Q17: typeof props === 'number' -> the level, anything else -> 9 (5.0 is the number 5 in JavaScript)
        level = props if isinstance(props, (int, float)) and not isinstance(props, bool) else 9
        if isinstance(level, float) and level == int(level):
            level = int(level)
        if level < 1 or level > 9 or not isinstance(level, int):      # (a fractional level is meaningless; the reference does not guard it)
            raise CjsError(-20, "Invalid block size multiplier")
        res = _stream_call(load_library().cjs_bzip2_compress, input, level)
        return _deliver(res, output)

    @staticmethod
    def decompressFile(input, output=None, multistream=False):
        res = _stream_call(load_library().cjs_bzip2_decompress, input, 1 if multistream else 0)
        return _deliver(res, output)


class BWTC:
    MAGIC = "bwtc"

    @staticmethod
    def compressFile(input, output=None, props=None):
        level = int(props) if isinstance(props, (int, float)) and not isinstance(props, bool) and 1 <= props <= 9 and props == int(props) else 9   # W2
        return _deliver(_stream_call(load_library().cjs_bwtc_compress, input, level), output)

    @staticmethod
    def decompressFile(input, output=None):
        L = load_library()
        data = _coerce_input(input)
        keep = data if data.size else np.zeros(1, dtype=np.uint8)
        out, out_n = u8p(), ctypes.c_size_t(0)
        _check(L.cjs_bwtc_decompress(keep.ctypes.data_as(u8p), data.size, ctypes.byref(out), ctypes.byref(out_n), None))
        return _deliver(_adopt(out, out_n.value), output)


def _deliver(res, output):
    # Util.coerceOutputStream (J/Bzip2_joined_.js:254-272)
    if output is None:
        return res
    if isinstance(output, int):
        if output != res.size:
            raise TypeError("outputsize does not match decoded input")
        return res
    if hasattr(output, "writeByte"):
        for b in res.tolist():
            output.writeByte(b)
        return output
    if len(output) != res.size:
        raise TypeError("outputsize does not match decoded input")
    output[:] = res
    return output


class DeviceContext:
    """Per-GPU workspace + stream for the device-resident pipeline (what bench.py times)."""

    def __init__(self, device, max_input, level, max_range_blocks=0):
        self.L = load_library()
        self.h = ctypes.c_void_p()
        self.level = level
        _check(self.L.cjs_ctx_create_sharded(ctypes.byref(self.h), device, max_input, max_range_blocks, level))

    def set_stage_times(self, on):
        """stats of later calls: per-stage times (a stream synchronisation per stage) or events only"""
        self.L.cjs_ctx_set_stage_times(self.h, 1 if on else 0)

    def close(self):
        if self.h:
            self.L.cjs_ctx_destroy(self.h)
            self.h = ctypes.c_void_p()

    def compress(self, d_in_ptr, n, d_out_ptr, out_cap, stats=None):
        out_n = ctypes.c_size_t(0)
        _check(self.L.cjs_bzip2_compress_device(self.h, d_in_ptr, n, self.level, d_out_ptr, out_cap, ctypes.byref(out_n),
                                                ctypes.byref(stats) if stats is not None else None))
        return out_n.value

    def compress_range(self, d_in_ptr, n, first, count, d_out_ptr, out_cap, stats=None, crc_cap=1 << 16):
        bits = ctypes.c_uint64(0)
        total = ctypes.c_long(0)
        crcs = np.zeros(crc_cap, dtype=np.uint32)
        _check(self.L.cjs_bzip2_compress_device_range(self.h, d_in_ptr, n, self.level, first, count, d_out_ptr, out_cap,
                                                      ctypes.byref(bits), crcs.ctypes.data, crc_cap, ctypes.byref(total),
                                                      ctypes.byref(stats) if stats is not None else None))
        return bits.value, total.value, crcs[: total.value]


    # ---- the three phases of a one-process-per-GPU job (include/cjs_hip.h); the caller owns the exchange between them
    def share_bytes(self, n, world):
        return self.L.cjs_bzip2_shard_share_bytes(n, world)

    def shard_tiles(self, d_in_ptr, n, rank, world, d_share_ptr):
        _check(self.L.cjs_bzip2_shard_tiles(self.h, d_in_ptr, n, rank, world, d_share_ptr))

    def shard_blocks(self, d_in_ptr, n, rank, world, d_shares_ptr, stats=None):
        meta = ShardMeta()
        _check(self.L.cjs_bzip2_shard_blocks(self.h, d_in_ptr, n, self.level, rank, world, d_shares_ptr, ctypes.byref(meta),
                                             ctypes.byref(stats) if stats is not None else None))
        return meta

    def shard_pack(self, rank, metas, d_out_ptr, out_cap):
        """-> (frag_off, frag_len, stream_off, stream_len): bytes [frag_off, +frag_len) of d_out are stream bytes [stream_off, +frag_len)"""
        arr = (ShardMeta * len(metas))(*metas)
        fo, fl = ctypes.c_size_t(0), ctypes.c_size_t(0)
        so, sl = ctypes.c_uint64(0), ctypes.c_uint64(0)
        _check(self.L.cjs_bzip2_shard_pack(self.h, self.level, rank, len(metas), arr, d_out_ptr, out_cap, ctypes.byref(fo), ctypes.byref(fl),
                                           ctypes.byref(so), ctypes.byref(sl)))
        return fo.value, fl.value, so.value, sl.value


def trim():
    """Give the workspace that cjs_bzip2_compress keeps per device between calls back to the driver."""
    load_library().cjs_trim()
