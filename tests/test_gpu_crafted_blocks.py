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


def make_stream(symbols, used=(97, 98), crc=0, orig=0, level=9, extra_selectors=0, lengths=None):
    """one block: `symbols` (RUNA / RUNB / rank symbols 2.. ; the end-of-block symbol is appended) under two equal tables;
    `lengths`: code length per symbol (canonical codes by (length, symbol)), default: all codes of one length; selectors all 0"""
    nsym_alpha = len(used) + 2
    eob = len(used) + 1
    syms = list(symbols) + [eob]
    if lengths is None:
        lengths = [max(1, (nsym_alpha - 1).bit_length())] * nsym_alpha
    assert len(lengths) == nsym_alpha
    codes, code, prev = {}, 0, min(lengths)
    for ln in range(min(lengths), max(lengths) + 1):            # first[L + 1] = (first[L] + cnt[L]) << 1
        code <<= (ln - prev)
        prev = ln
        for sy in range(nsym_alpha):
            if lengths[sy] == ln:
                codes[sy] = code
                code += 1
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
        cur = lengths[0]
        w.put(cur, 5)
        for ln in lengths:                                       # '10' = one longer, '11' = one shorter, '0' = this symbol's length
            while cur < ln:
                w.put(2, 2)
                cur += 1
            while cur > ln:
                w.put(3, 2)
                cur -= 1
            w.put(0, 1)
    for s in syms:
        w.put(codes[s], lengths[s])
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


@pytest.mark.parametrize("seed", range(24))
def test_groups_of_every_length_in_bits(hip, oracle, seed):
    # The decoder finds the end of a group of 50 codes by pointer doubling over a window of bit positions that is shorter than the
    # longest possible group, working a group out again when its chain leaves the window, and speculates on the group behind it.
    # Mixes of a 1-bit and a 20-bit code give groups of 50 .. 1000 bits whose code boundaries fall on every window edge.
    rng = np.random.RandomState(1000 + seed)
    p_long = (0.02, 0.2, 0.45, 0.5, 0.55, 0.8, 0.98, 1.0)[seed % 8]
    n = (60, 260, 700, 1500)[seed % 4]
    symbols = [3 if x < p_long else 2 for x in rng.random_sample(n)]
    if seed % 3 == 0:                                            # a few short zero-rank runs in between (20-bit codes too)
        for at in rng.randint(0, n, 5):
            symbols[at] = RUNA
    rc_h, rc_o, out_h, out_o = decode_both(hip, oracle, symbols, used=(65, 66, 67), lengths=[20, 20, 1, 20, 20])
    assert rc_h == rc_o == 0, (seed, rc_h, rc_o, hip.last_error_detail())
    assert out_h.size == out_o.size and np.array_equal(out_h, out_o), seed


def test_group_ends_on_a_window_edge(hip, oracle):
    # code boundaries exactly on bit 512 of a group (the first attempt's window) in the middle of the group and as its very end,
    # and the same for the group behind it (whose window starts 50 bits into the step: bit 1074)
    L, S = 3, 2                                                  # 20-bit and 1-bit rank symbols
    mid = [L] * 25 + [S] * 12 + [L] * 5 + [S] * 8                # 512 bits after 37 codes, 50 codes in all
    end = [L] * 24 + [S] * 18 + [L] * 0 + [S] * 8                # 498 + ... (made exact below)
    end = [L] * 24 + [S] * 26                                    # 480 + 26 = 506: a little below the edge
    exact = [L] * 25 + [S] * 12 + [S] * 13                       # 512 + 13 = 525 bits
    at_edge = [L] * 24 + [S] * 24 + [L] * 1 + [S] * 1            # 480 + 24 = 504, + 20 = 524 ...
    edge_end = [S] * 28 + [L] * 22                               # 28 + 440 = 468
    full = [L] * 50                                              # 1000 bits: the whole span
    for name, groups in (("mid", [mid, mid, mid]), ("short-long", [end, full, exact, full, mid]), ("full", [full, full, full]),
                         ("mixed", [at_edge, edge_end, mid, exact, end, full, mid])):
        symbols = [x for g in groups for x in g]
        rc_h, rc_o, out_h, out_o = decode_both(hip, oracle, symbols, used=(65, 66, 67), lengths=[20, 20, 1, 20, 20])
        assert rc_h == rc_o == 0, (name, rc_h, rc_o, hip.last_error_detail())
        assert np.array_equal(out_h, out_o), name


def group_of_bits(bits, rng=None):
    """50 rank symbols whose codes (20, 10 and 1 bits under lengths [20, 20, 1, 10, 20]) take exactly `bits` bits, or None"""
    for a in range(50, -1, -1):                                  # 19 a + 9 c = bits - 50
        rest = bits - 50 - 19 * a
        if rest >= 0 and rest % 9 == 0 and a + rest // 9 <= 50:
            c = rest // 9
            g = [4] * a + [3] * c + [2] * (50 - a - c)           # symbol 4: 20 bits, 3: 10 bits, 2: 1 bit
            if rng is not None:
                rng.shuffle(g)
            return g
    return None


@pytest.mark.parametrize("seed", range(12))
def test_later_groups_miss_their_windows_in_every_way(hip, oracle, seed):
    # Four groups go in one step of the decoder's chain: the first from its known start, the others over 448 positions placed by the
    # shortest codes of the tables in front and by the length of the LAST group.  Group lengths that jump between 50 and 1000 bits
    # put later groups in front of their positions, behind them and across their end, and the first group beyond its first attempt.
    rng = np.random.RandomState(7000 + seed)
    groups = []
    while len(groups) < 240:
        mode = rng.randint(0, 4)
        bits = int(rng.choice([50, 59, 200, 213, 440, 447, 448, 449, 450, 600, 896, 1000])) if mode == 0 else int(rng.randint(50, 1001)) if mode == 1 \
            else int(rng.randint(150, 300)) if mode == 2 else int(rng.randint(400, 500))
        g = group_of_bits(bits, rng)
        if g is not None:
            groups.extend([g] * int(rng.randint(1, 5)))          # runs of equal lengths let the placement settle, then it jumps
    symbols = [x for g in groups for x in g][:-1]                # (the end-of-block symbol is the last group's 50th)
    rc_h, rc_o, out_h, out_o = decode_both(hip, oracle, symbols, used=(65, 66, 67, 68), lengths=[20, 20, 1, 10, 20, 20])
    assert rc_h == rc_o == 0, (seed, rc_h, rc_o, hip.last_error_detail())
    assert out_h.size == out_o.size and np.array_equal(out_h, out_o), seed


def test_tables_whose_shortest_code_is_long(hip, oracle):
    # 50 x the shortest code is where the next group's positions start at the earliest: with no code below 11 bits that is beyond
    # the 448 positions of the group in front, and the third group's start is capped
    rng = np.random.RandomState(5)
    for lengths in ([11, 12, 11, 20, 12], [16, 20, 15, 17, 20], [9, 9, 9, 9, 9]):
        symbols = [int(x) for x in rng.randint(2, 4, 700)]
        rc_h, rc_o, out_h, out_o = decode_both(hip, oracle, symbols, used=(65, 66, 67), lengths=lengths)
        assert rc_h == rc_o == 0, (lengths, rc_h, rc_o, hip.last_error_detail())
        assert np.array_equal(out_h, out_o), lengths


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
