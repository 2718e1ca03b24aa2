"""-m gpu: hand-built bzip2 blocks that no compressor emits, decoded by the HIP path and by the oracle.

The decoder's symbol stage is rebuilt from the reference's loop (J/Bzip2_joined_.js:1597-1670) as parallel stages; these streams
pin the corners of that loop: zero-rank runs of 19 / 20 / 31 / 32 / 33 / 64+ digits (the reference keeps the digit weight in an
int32 that it shifts left: the 32nd digit makes the weight 0 and with it the run is forgotten), runs that cross a 50-symbol
group, a 4096-symbol bookkeeping tile, the end of block in front of the last selector group, more selectors than symbols.
The block CRC of a crafted stream is taken from the HIP library's own error text ("Bad block CRC (got X ...") and written
into the stream: the ORACLE then accepts the stream only if its independent decode gives the same bytes.
"""
import re

import numpy as np
import pytest

import support

pytestmark = pytest.mark.gpu
RUNA, RUNB = 0, 1


class Bits:
    def __init__(self):
        self.bits = []

    def put(self, value, n):
        self.bits.extend((value >> (n - 1 - i)) & 1 for i in range(n))

    def bytes(self):
        b = self.bits + [0] * (-len(self.bits) % 8)
        return np.packbits(np.array(b, dtype=np.uint8))


def make_stream(symbols, used=(97, 98), crc=0, orig=0, level=9, extra_selectors=0):
    """one block: `symbols` (RUNA / RUNB / rank symbols 2.. ; the end-of-block symbol is appended) under two equal tables of
    fixed-length codes; selectors all 0"""
    nsym_alpha = len(used) + 2
    eob = len(used) + 1
    syms = list(symbols) + [eob]
    L = max(1, (nsym_alpha - 1).bit_length())
    w = Bits()
    for ch in b"BZh%d" % level:
        w.put(ch, 8)
    w.put(0x314159265359, 48)
    w.put(crc, 32)
    w.put(0, 1)
    w.put(orig, 24)
    coarse = 0
    fine = [0] * 16
    for u in used:
        coarse |= 0x8000 >> (u >> 4)
        fine[u >> 4] |= 0x8000 >> (u & 15)
    w.put(coarse, 16)
    for r in range(16):
        if coarse & (0x8000 >> r):
            w.put(fine[r], 16)
    w.put(2, 3)
    nsel = (len(syms) + 49) // 50 + extra_selectors
    w.put(nsel, 15)
    for _ in range(nsel):
        w.put(0, 1)
    for _ in range(2):
        w.put(L, 5)
        for _ in range(nsym_alpha):
            w.put(0, 1)
    for s in syms:
        w.put(s, L)                  # canonical codes of equal length are the symbol numbers
    w.put(0x177245385090, 48)
    w.put(crc, 32)                   # one block: the stream CRC is the block's
    return w.bytes()


def decode_both(hip, oracle, symbols, **kw):
    s0 = make_stream(symbols, crc=0, **kw)
    rc, _ = hip.bzip2_decompress(s0)
    if rc == 0:
        return 0, 0, hip.bzip2_decompress(s0)[1], oracle.bzip2_decompress(s0)[1]
    m = re.search(r"Bad block CRC \(got ([0-9a-f]+) expected", hip.last_error_detail())
    if not m:                                    # not a CRC complaint: the block itself is bad for the HIP path -- it must be for the oracle
        return rc, oracle.bzip2_decompress(s0)[0], None, None
    s1 = make_stream(symbols, crc=int(m.group(1), 16), **kw)
    rc_h, out_h = hip.bzip2_decompress(s1)
    rc_o, out_o = oracle.bzip2_decompress(s1)
    return rc_h, rc_o, out_h, out_o


CASES = {
    "plain": [2, 2, RUNA, 2, RUNB, RUNB, 2],
    "run19": [RUNA] * 19 + [2],                          # 2^19 - 1 bytes: fits a level-9 block
    "run20": [RUNA] * 20 + [2],                          # 2^20 - 1 bytes: does not (:1647)
    "run31": [RUNA] * 31 + [2],
    "run32-forgotten": [2] + [RUNA] * 32 + [2],          # the 32nd digit zeroes the weight: no flush, no bytes
    "run32b-forgotten": [2] + [RUNA] * 31 + [RUNB, 2],
    "run33": [2] + [RUNA] * 33 + [2],                    # 32 forgotten, the 33rd starts over: one byte
    "run64+3": [2] + [RUNB] * 64 + [RUNB, RUNA, RUNB] + [2, 2],
    "run-at-end": [2, 2] + [RUNA, RUNB, RUNA],           # flushed by the end-of-block symbol (:1643)
    "run32-at-end": [2, 2] + [RUNA] * 32,
    "only-run": [RUNB] * 5,
    "across-groups": [2] * 47 + [RUNA] * 7 + [2] * 60 + [RUNB] * 3 + [2],
    "across-tile": [2] * 4090 + [RUNA] * 12 + [2] * 5000,
    "forgotten-across-tile": [2] * 4080 + [RUNA] * 40 + [2] * 100,
    "exactly-50": [2] * 49,                              # the end-of-block symbol is the 50th of the group
    "exactly-51": [2] * 50,                              # ... the first of the next
    "three-bytes": ([2, 3, RUNA, 3, 2, RUNB, 3] * 40, dict(used=(65, 66, 67))),
    "spare-selectors": ([2, RUNA, 2] * 30, dict(extra_selectors=3)),
    "orig-out-of-range": ([2, 2, 2], dict(orig=3)),      # :1677
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_crafted_block(hip, oracle, name):
    case = CASES[name]
    symbols, kw = case if isinstance(case, tuple) else (case, {})
    rc_h, rc_o, out_h, out_o = decode_both(hip, oracle, symbols, **kw)
    assert rc_h == rc_o, (name, rc_h, rc_o, hip.last_error_detail())
    if rc_h == 0:
        assert out_h.size == out_o.size and np.array_equal(out_h, out_o), name
    if name in ("run20", "run31", "orig-out-of-range"):
        assert rc_h == -5
    if name in ("run19", "run32-forgotten", "run33", "run64+3", "across-tile", "forgotten-across-tile", "exactly-50", "exactly-51", "spare-selectors"):
        assert rc_h == 0, (name, hip.last_error_detail())
    if name == "run19":
        assert out_h.size > 500000


def test_missing_end_of_block_is_a_data_error(hip, oracle):
    # the selectors run out before an end-of-block symbol shows up (:1602)
    full = make_stream([2] * 120)
    # same header, but announce one group less than the symbols need: rebuild with a hand-made count
    w_ok = make_stream([2] * 99)                 # 100 symbols with the end of block: 2 groups
    rc_h, _ = hip.bzip2_decompress(w_ok)
    rc_o, _ = oracle.bzip2_decompress(w_ok)
    assert rc_h == rc_o                          # (CRC 0: both complain about the CRC or both accept)
    cut = make_stream([2] * 120, extra_selectors=-1)
    assert hip.bzip2_decompress(cut)[0] == oracle.bzip2_decompress(cut)[0] == -5
    assert hip.bzip2_decompress(full)[0] == oracle.bzip2_decompress(full)[0]
