"""Pins the oracle (oracle/cjs_oracle.c) against
  * the reference's own known answers / fixtures: cyclic-BWT KATs
    (NPM/test/bwtest.js:39-79), allocator KATs (NPM/test/huffman.js:15-76), decoder goldens
    sample0-4.bz2 -> .ref (bzip2-basic.js), .bzt block tables (bzip2-table.js), block-offset
    dumps (bzip2-block.js);
  * outputs of the reference JS run under Node (tests/golden/*.json, cut by make_golden.js).
CPU only."""
import os

import numpy as np
import pytest

import recipes
import support

DATA = recipes.DATA
SMALL = support.load_golden("golden_small.json")
KAT = support.load_golden("kat.json")


def _id(c):
    return "%s-%s-%d" % (c["name"], c["algo"], c["level"])


@pytest.mark.parametrize("case", SMALL["cases"], ids=_id)
def test_compress_matches_reference(oracle, case):
    data = recipes.build(case["recipe"])
    assert data.size == case["in_len"] and support.sha256(data) == case["in_sha256"]
    fn = oracle.bzip2_compress if case["algo"] == "Bzip2" else oracle.bwtc_compress
    rc, out = fn(data, case["level"])
    assert rc == 0
    assert out.size == case["out_len"]
    assert support.sha256(out) == case["out_sha256"]
    if "out_hex" in case:
        assert out.tobytes().hex() == case["out_hex"]
    # and the oracle's decoder inverts it
    back = oracle.bzip2_decompress(out)[1] if case["algo"] == "Bzip2" else oracle.bwtc_decompress(out)[1]
    assert back is not None and back.size == data.size and np.array_equal(back, data)


# tiny full streams recorded in SURVEY.md §8(c) (Bzip2 -9 / BWTC -9)
SURVEY_STREAMS = [
    (b"", "425a683917724538509000000000", "627774638109ab000007"),
    (b"a", "425a683931415926535919939b6b00000001002000200021184682ee48a70a120332736d60", "627774638209581eea7e1900000b"),
    (b"aaaa", "425a6839314159265359881233a600000241004000200020002127a820538bb9229c2848440919d300", "6277746385095d65eea49d7300000c"),
    (b"abab", "425a68393141592653598738e0f60000008100300020002127a8204b8bb9229c2848439c707b00", "6277746385095d6132a1fe447900000d"),
    (b"banana", "425a6839314159265359efb6ec01000001810030012000218f506610bc5dc914e14243bedbb004", "6277746387095ebb43fb18c9132e3c00000f"),
]


@pytest.mark.parametrize("text,bz,bw", SURVEY_STREAMS)
def test_survey_tiny_streams(oracle, text, bz, bw):
    assert oracle.bzip2_compress(text, 9)[1].tobytes().hex() == bz
    assert oracle.bwtc_compress(text, 9)[1].tobytes().hex() == bw


def test_bwt_cyclic_kats(oracle):
    for k in KAT["bwt_cyclic"]:
        data = bytes.fromhex(k["input_hex"]) if "input_hex" in k else recipes.build(k["recipe"])
        U, pidx = oracle.bwt_cyclic(data)
        assert pidx == k["pidx"]
        if "out_hex" in k:
            assert U.tobytes().hex() == k["out_hex"]
        else:
            assert support.sha256(U) == k["out_sha256"]


def test_bwt_cyclic_reference_test_vectors(oracle):
    # NPM/test/bwtest.js:39-79
    mary = b"Mary had a little lamb, its fleece was white as snow" * 8 + b"Nary had a little lamb, its fleece was white as snow"
    vec = [
        (b"bcababa", b"cbbaaab", 5),
        (b"ABCDEFGHIJKLMNOPQRSTUVWXYZ", b"ZABCDEFGHIJKLMNOPQRSTUVWXY", 0),
        (b"ZYXWVUTSRQPONMLKJIHGFEDCBA", b"BCDEFGHIJKLMNOPQRSTUVWXYZA", 25),
        (b"SIX.MIXED.PIXIES.SIFT.SIXTY.PIXIE.DUST.BOXES", b"TEXYDST.E.IXIXIXXSSMPPS.B..E.S.EUSFXDIIOIIIT", 29),
    ]
    for t, u, p in vec:
        U, pidx = oracle.bwt_cyclic(t)
        assert U.tobytes() == u and pidx == p
    U, pidx = oracle.bwt_cyclic(mary)
    assert pidx == 99
    assert U.tobytes().startswith(b"dddddddddeeeeeeeeesssssssssyyyyyyyyy,,,,,,,,,")
    assert U.tobytes().endswith(b"ooooooooo                  rrrrrrrrr")


def test_bwt_sentinel_kats(oracle):
    for k in KAT["bwt_sentinel"]:
        U, pidx = oracle.bwt_sentinel(bytes.fromhex(k["input_hex"]))
        assert U.tobytes().hex() == k["out_hex"] and pidx == k["pidx"]
    U, pidx = oracle.bwt_sentinel(b"banana")
    assert U.tobytes() == b"annbaa" and pidx == 4


def test_suffix_array_property(oracle):
    # NPM/test/suftest.js sufcheck: permutation + sorted order
    for name in ("sample1.ref", "sample3.ref"):
        T = np.fromfile(os.path.join(DATA, name), dtype=np.uint8)[:40000]
        SA = oracle.suffix_array(T)
        assert np.array_equal(np.sort(SA), np.arange(T.size))
        tb = T.tobytes()
        for i in range(0, T.size - 1, 97):
            assert tb[SA[i]:] < tb[SA[i + 1]:]


def test_huffman_allocator_kats(oracle):
    for k in KAT["huffman_alloc"]:
        assert oracle.huff_alloc(k["freq_sorted"], k["limit"]) == k["lengths"]


def test_huffman_allocator_reference_test_vectors(oracle):
    # NPM/test/huffman.js:15-76
    fib = [0, 1]
    while len(fib) < 37:
        fib.append(fib[-1] + fib[-2])
    assert oracle.huff_alloc([1], 32) == [1]
    assert oracle.huff_alloc([1, 1], 32) == [1, 1]
    assert oracle.huff_alloc([1] * 5, 32) == [3, 3, 2, 2, 2]
    assert oracle.huff_alloc([0, 0, 1, 1, 1, 1], 3) == [3, 3, 3, 3, 2, 2]
    assert oracle.huff_alloc(fib[:36], 20) == [20] * 16 + [19, 19, 18, 17, 16, 16, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1]
    assert oracle.huff_alloc(fib[:22], 20) == [20, 20, 19, 19, 19, 17, 16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1]
    assert oracle.huff_alloc(fib[:21], 20) == [20, 20, 19, 18, 17, 16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1]
    assert oracle.huff_alloc(fib[:36], 6) == [6] * 30 + [5, 5, 5, 4, 3, 2]
    # SURVEY.md Q10b
    assert oracle.huff_alloc([0, 0, 0, 0, 0], 20) == [3, 3, 2, 2, 2]
    assert oracle.huff_alloc([0, 0, 0], 20) == [2, 2, 1]
    assert oracle.huff_alloc([0, 0, 0, 1, 1000], 20) == [4, 4, 3, 2, 1]


def test_crc_kats(oracle):
    for k in KAT["crc"]:
        assert oracle.crc32(k["ascii"].encode()) == k["crc"]


@pytest.mark.parametrize("name", ["sample0", "sample1", "sample2", "sample3", "sample4"])
def test_decoder_goldens(oracle, name):
    # NPM/test/bzip2-basic.js: streams made by real bzip2
    comp = np.fromfile(os.path.join(DATA, name + ".bz2"), dtype=np.uint8)
    ref = np.fromfile(os.path.join(DATA, name + ".ref"), dtype=np.uint8)
    rc, out = oracle.bzip2_decompress(comp)
    assert rc == 0 and np.array_equal(out, ref)
    # NPM/test/bzip2-table.js
    rc, table = oracle.bzip2_table(comp)
    text = "".join("%d\t%d\n" % (p, s) for p, s in table)
    assert rc == 0 and text == open(os.path.join(DATA, name + ".bzt")).read()


@pytest.mark.parametrize("name,bitpos", [("sample2", 544888), ("sample4", 32), ("sample4", 1596228), ("sample4", 2342106)])
def test_decoder_block_goldens(oracle, name, bitpos):
    # NPM/test/bzip2-block.js
    comp = np.fromfile(os.path.join(DATA, name + ".bz2"), dtype=np.uint8)
    ref = np.fromfile(os.path.join(DATA, "%s.%d" % (name, bitpos)), dtype=np.uint8)
    rc, out = oracle.bzip2_decompress_block(comp, bitpos)
    assert rc == 0 and np.array_equal(out, ref)


def test_decoder_errors(oracle):
    # garbage -> -2, bit flip -> -5, bad level, bad magic
    assert oracle.bzip2_decompress(b"garbage data here")[0] == -2
    assert oracle.bzip2_decompress(b"BZh0")[0] == -2
    comp = np.fromfile(os.path.join(DATA, "sample1.bz2"), dtype=np.uint8).copy()
    comp[2000] ^= 0x10
    assert oracle.bzip2_decompress(comp)[0] == -5
    assert oracle.bzip2_compress(b"abc", 0)[0] == -20
    assert oracle.bzip2_compress(b"abc", 10)[0] == -20
    assert oracle.bwtc_decompress(b"nope")[0] == -21
    # BWTC: invalid level silently becomes 9 (W2)
    assert np.array_equal(oracle.bwtc_compress(b"banana", 0)[1], oracle.bwtc_compress(b"banana", 9)[1])


def test_multistream(oracle):
    a = oracle.bzip2_compress(b"first stream ", 1)[1]
    b = oracle.bzip2_compress(b"second stream", 9)[1]
    both = np.concatenate([a, b])
    assert oracle.bzip2_decompress(both, 0)[1].tobytes() == b"first stream "
    assert oracle.bzip2_decompress(both, 1)[1].tobytes() == b"first stream second stream"


@pytest.mark.slow
def test_big_golden_10m(oracle):
    import glob
    for f in sorted(glob.glob(os.path.join(support.GOLDEN, "golden_big_*_10m.json"))):
        for case in support.load_golden(os.path.basename(f))["cases"]:
            data = recipes.build(case["recipe"])
            assert support.sha256(data) == case["in_sha256"]
            fn = oracle.bzip2_compress if case["algo"] == "Bzip2" else oracle.bwtc_compress
            rc, out = fn(data, case["level"])
            assert rc == 0 and out.size == case["out_len"] and support.sha256(out) == case["out_sha256"]
