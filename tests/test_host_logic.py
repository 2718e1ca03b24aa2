"""Host logic of the product that needs no GPU: the serial entropy stage of BWTC.decompressFile
(cjs_stage_bwtc_entropy_decode: range decoder + adaptive model + RLE2 + MTF inverse, J/BWTC_joined_.js:1827-1913).

The streams come from the oracle's BWTC encoder (pinned by the reference-cut goldens in test_oracle.py); the decoded BWT
columns / primary indices are compared with the oracle's sentinel BWT of every block.  Negative cases: truncated and
bit-flipped streams (DefSum model at level 1, Fenwick model at level 9) and a forged stream of thousands of empty blocks
(the allocation-amplification case): the stage must return an error code or finish, never crash or over-allocate.
"""
import resource

import numpy as np
import pytest

import recipes
import support


@pytest.fixture(scope="module")
def hip():
    return support.HipLib()


def _cases():
    rng = np.random.default_rng(5)
    text = recipes.textgen(250000, 3)
    yield "empty", np.empty(0, np.uint8)
    yield "one", np.frombuffer(b"a", np.uint8)
    yield "banana", np.frombuffer(b"banana", np.uint8)
    yield "zeros", np.zeros(120000, np.uint8)
    yield "abab", np.frombuffer(b"ab" * 5000, np.uint8)
    yield "all256", np.tile(np.arange(256, dtype=np.uint8), 40)
    yield "random", rng.integers(0, 256, 70000, dtype=np.uint8)
    yield "text", text
    yield "text_exact_block", text[:200000]
    yield "skewed", rng.choice(np.array([0, 0, 0, 0, 1, 2, 3, 255], np.uint8), 150000)


@pytest.mark.parametrize("level", [1, 2, 5, 6, 9])
def test_entropy_decode_matches_oracle_bwt(hip, oracle, level):
    for name, data in _cases():
        rc, stream = oracle.bwtc_compress(data, level)
        assert rc == 0
        nb, lv, blocks = hip.stage_bwtc_entropy_decode(stream)
        assert nb >= 0, (name, nb)
        assert lv == level
        bs = level * 100000
        want = [data[i:i + bs] for i in range(0, data.size, bs)]
        assert nb == len(want), name
        for k, blk in enumerate(want):
            U, pidx = oracle.bwt_sentinel(blk)
            assert blocks[k][1] == pidx, (name, k)
            assert np.array_equal(blocks[k][0], U), (name, k)


def test_bad_magic_and_garbage(hip):
    assert hip.stage_bwtc_entropy_decode(b"bwtx\x81\x09")[0] == -21
    assert hip.stage_bwtc_entropy_decode(b"")[0] == -21
    assert hip.stage_bwtc_entropy_decode(b"bwtc")[0] == -5           # no size varint
    rng = np.random.default_rng(9)
    for n in (5, 6, 7, 16, 100, 5000):
        junk = np.concatenate([np.frombuffer(b"bwtc\x81", np.uint8), rng.integers(0, 256, n, dtype=np.uint8)])
        nb, _, _ = hip.stage_bwtc_entropy_decode(junk)
        assert nb < 0 or nb >= 0          # any answer is fine; the call must come back


@pytest.mark.parametrize("level", [1, 9])
def test_truncated_and_flipped_streams(hip, oracle, level):
    data = recipes.textgen(150000, 11)
    rc, stream = oracle.bwtc_compress(data, level)
    assert rc == 0
    U0, p0 = oracle.bwt_sentinel(data[: level * 100000])
    for cut in (stream.size - 1, stream.size - 4, stream.size // 2, 12, 6):
        nb, _, blocks = hip.stage_bwtc_entropy_decode(stream[:cut])
        if nb > 0 and cut < stream.size - 8:      # a shortened stream must not decode to the full answer
            assert not (len(blocks) == -(-data.size // (level * 100000)) and np.array_equal(blocks[0][0], U0) and blocks[0][1] == p0 and
                        sum(b[0].size for b in blocks) == data.size and cut < stream.size // 2 + 1)
    rng = np.random.default_rng(level)
    differs = 0
    for _ in range(24):
        bad = stream.copy()
        pos = int(rng.integers(6, stream.size - 6))
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        nb, _, blocks = hip.stage_bwtc_entropy_decode(bad)
        if nb < 0 or sum(b[0].size for b in blocks) != data.size or not np.array_equal(blocks[0][0], U0):
            differs += 1
    assert differs >= 20          # a flipped bit derails the adaptive model almost always


class _Enc:
    """the format's range encoder (RangeCoder encode side, J/BWTC_joined_.js:40-153), enough to forge streams"""

    def __init__(self, first_byte):
        self.out = bytearray()
        self.low, self.range, self.buffer, self.help, self.bytecount = 0, 0x80000000, first_byte, 0, 1

    def _norm(self):
        while self.range <= 0x00800000:
            if self.low < (0xFF << 23):
                self.out.append(self.buffer)
                self.out.extend(b"\xff" * self.help)
                self.help = 0
                self.buffer = (self.low >> 23) & 0xFF
            elif self.low & 0x80000000:
                self.out.append((self.buffer + 1) & 0xFF)
                self.out.extend(b"\x00" * self.help)
                self.help = 0
                self.buffer = (self.low >> 23) & 0xFF
            else:
                self.help += 1
            self.range = (self.range << 8) & 0xFFFFFFFF
            self.low = (self.low << 8) & 0x7FFFFFFF
            self.bytecount += 1

    def freq(self, sy, lt, tot):
        self._norm()
        r = self.range // tot
        tmp = r * lt
        self.low += tmp
        self.range = r * sy if lt + sy < tot else self.range - tmp

    def shift(self, sy, lt, sh):
        self._norm()
        r = self.range >> sh
        tmp = r * lt
        self.low += tmp
        self.range = self.range - tmp if (lt + sy) >> sh else r * sy

    def bits(self, k, v):
        for i in range(k - 1, -1, -1):
            self.shift(1, (v >> i) & 1, 1)

    def finish(self):
        self._norm()
        self.bytecount += 5
        tmp = self.low >> 23
        if (self.low & 0x7FFFFF) >= ((self.bytecount & 0xFFFFFF) >> 1):
            tmp += 1
        if tmp > 0xFF:
            self.out.append((self.buffer + 1) & 0xFF)
            self.out.extend(b"\x00" * self.help)
        else:
            self.out.append(self.buffer)
            self.out.extend(b"\xff" * self.help)
        self.out.append(tmp & 0xFF)
        self.out.extend(bytes([(self.bytecount >> 16) & 0xFF, (self.bytecount >> 8) & 0xFF, self.bytecount & 0xFF]))
        return bytes(self.out)


def _forged_empty_blocks(nblocks, level=9):
    e = _Enc(0x80)                      # varint(0): size unknown
    e.shift(1, level, 8)
    lgbits = 5                          # fls(1 + fls(900000 - 1) - 1) for every level
    for _ in range(nblocks):
        e.freq(1, 1, 3)                 # "short block"
        e.bits(lgbits, 0)               # length 0
        e.bits(lgbits, 0)               # pidx 0
        e.freq(1, 0, 3)                 # use-tree root: empty
    e.freq(1, 2, 3)
    return b"bwtc" + e.finish()


def test_forged_empty_blocks_do_not_amplify_memory(hip, oracle):
    stream = _forged_empty_blocks(60000)
    assert len(stream) < 120000
    # the oracle agrees that this is a well-formed stream of nothing
    rc, back = oracle.bwtc_decompress(stream)
    assert rc == 0 and back.size == 0
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    nb, lv, blocks = hip.stage_bwtc_entropy_decode(stream, cap=16)
    after = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    assert nb == 0 and lv == 9          # empty blocks contribute nothing
    assert after - before < 200 * 1024  # KiB: no 900000-byte row per 13 input bits (that would be ~50 GB)
