"""Stage-level parity of the HIP kernels against the oracle (GPU box only).  Every call goes
through the C ABI of libcjs_hip.so (cjs_stage_*)."""
import numpy as np
import pytest

import recipes

pytestmark = pytest.mark.gpu


def _check_bwt(hip, oracle, data, block_len, cyclic):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    rc, U, pidx = hip.stage_bwt(data, block_len, cyclic)
    assert rc == 0
    nb = -(-data.size // block_len)
    for k in range(nb):
        blk = data[k * block_len:(k + 1) * block_len]
        eu, ep = (oracle.bwt_cyclic if cyclic else oracle.bwt_sentinel)(blk)
        got = U[k * block_len:k * block_len + blk.size]
        assert pidx[k] == ep, "pidx block %d: got %d want %d (n=%d cyclic=%s)" % (k, pidx[k], ep, blk.size, cyclic)
        if not np.array_equal(got, eu):
            bad = np.nonzero(got != eu)[0]
            raise AssertionError("BWT bytes differ in block %d at %d positions, first %d (n=%d cyclic=%s)" % (k, bad.size, bad[0], blk.size, cyclic))


BWT_INPUTS = {
    "banana": (np.frombuffer(b"banana", dtype=np.uint8), 6),
    "kat_bcababa": (np.frombuffer(b"bcababa", dtype=np.uint8), 7),
    "single_byte": (np.frombuffer(b"x", dtype=np.uint8), 1),
    "two_blocks_tiny": (np.frombuffer(b"abracadabra", dtype=np.uint8), 7),
    "aaaa": (np.frombuffer(b"aaaa", dtype=np.uint8), 4),
    "abab": (np.frombuffer(b"abababababab", dtype=np.uint8), 12),
    "mary": (np.frombuffer(b"Mary had a little lamb, its fleece was white as snow" * 8 + b"Nary had a little lamb, its fleece was white as snow", dtype=np.uint8), 1000),
    "zeros_5000": (np.zeros(5000, dtype=np.uint8), 5000),
    "zeros_3blocks": (np.zeros(10000, dtype=np.uint8), 4096),
    "ab_10000": (recipes.build({"kind": "repeat", "unit_hex": "6162", "n": 10000}), 10000),
    "abc_period3": (recipes.build({"kind": "repeat", "unit_hex": "616263", "n": 30000}), 30000),
    "range256": (recipes.build({"kind": "range256", "n": 70000}), 70000),
    "random_multi": (recipes.build({"kind": "xorshift", "n": 250000, "seed": 12345}), 99981),
    "random4sym": (recipes.build({"kind": "xorshift", "n": 200000, "seed": 777, "mask": 3, "add": 97}), 99981),
    "random2sym": (recipes.build({"kind": "xorshift", "n": 150000, "seed": 4242, "mask": 1, "add": 48}), 150000),
    "textgen_multi": (recipes.build({"kind": "textgen", "n": 1000000, "seed": 1}), 99981),
    "textgen_900k": (recipes.build({"kind": "textgen", "n": 1000000, "seed": 3}), 899981),
    # a long periodic stretch inside text: eight groups of 25,000 suffixes that stay unresolved for ~15 doubling rounds
    # (compact -> radix -> scatter-back path beside the in-LDS tile sort), text groups around them
    "text_plus_periodic": (recipes.build({"kind": "concat", "parts": [
        {"kind": "textgen", "n": 600000, "seed": 5}, {"kind": "repeat", "unit_hex": "6162636465666768", "n": 200000},
        {"kind": "textgen", "n": 99981, "seed": 6}]}), 899981),
    # medium groups: a 2000-byte page repeated 150 times with one changed byte per copy (groups of ~150, then splits)
    "page_repeats": (np.concatenate([np.concatenate([recipes.build({"kind": "textgen", "n": 2000, "seed": 9}),
                                                     np.array([65 + (i % 26)], dtype=np.uint8)]) for i in range(150)]), 300150),
    "sample3": (recipes.build({"kind": "file", "name": "sample3.ref"}), 120244),
    "sample1": (recipes.build({"kind": "file", "name": "sample1.ref"}), 98696),
}


@pytest.mark.parametrize("name", sorted(BWT_INPUTS))
@pytest.mark.parametrize("cyclic", [True, False], ids=["cyclic", "sentinel"])
def test_bwt_stage(hip, oracle, name, cyclic):
    data, block_len = BWT_INPUTS[name]
    _check_bwt(hip, oracle, data, block_len, cyclic)


RLE_INPUTS = {
    "empty": ({"kind": "repeat", "unit_hex": "00", "n": 0}, [1, 9]),
    "one": ({"kind": "repeat", "unit_hex": "41", "n": 1}, [1]),
    "run4_at_eof": ({"kind": "concat", "parts": [{"kind": "repeat", "unit_hex": "78797a", "n": 3}, {"kind": "repeat", "unit_hex": "71", "n": 4}]}, [9]),
    "zeros_300000": ({"kind": "repeat", "unit_hex": "00", "n": 300000}, [1, 9]),
    "zeros_12M": ({"kind": "repeat", "unit_hex": "00", "n": 12000000}, [1]),
    "ab_250001": ({"kind": "repeat", "unit_hex": "6162", "n": 250001}, [1]),
    "long_runs_mixed": ({"kind": "concat", "parts": [
        {"kind": "repeat", "unit_hex": "61", "n": 255}, {"kind": "repeat", "unit_hex": "62", "n": 256},
        {"kind": "repeat", "unit_hex": "63", "n": 259}, {"kind": "repeat", "unit_hex": "64", "n": 4},
        {"kind": "repeat", "unit_hex": "65", "n": 5}, {"kind": "repeat", "unit_hex": "66", "n": 1000},
        {"kind": "repeat", "unit_hex": "67", "n": 3}, {"kind": "repeat", "unit_hex": "61", "n": 260},
        {"kind": "repeat", "unit_hex": "6162", "n": 9}, {"kind": "repeat", "unit_hex": "00", "n": 511}]}, [1, 9]),
    "q2_run_at_block_end": ({"kind": "concat", "parts": [
        {"kind": "xorshift", "n": 99977, "seed": 99, "mask": 63, "add": 32}, {"kind": "repeat", "unit_hex": "00", "n": 4},
        {"kind": "repeat", "unit_hex": "41", "n": 50}, {"kind": "xorshift", "n": 1000, "seed": 5, "mask": 63, "add": 32}]}, [1]),
    "q2_count_fills_block": ({"kind": "concat", "parts": [
        {"kind": "xorshift", "n": 99976, "seed": 98, "mask": 63, "add": 32}, {"kind": "repeat", "unit_hex": "00", "n": 40},
        {"kind": "xorshift", "n": 1000, "seed": 6, "mask": 63, "add": 32}]}, [1]),
    "q2_run_crosses_block": ({"kind": "concat", "parts": [
        {"kind": "xorshift", "n": 99979, "seed": 97, "mask": 63, "add": 32}, {"kind": "repeat", "unit_hex": "7a", "n": 700},
        {"kind": "xorshift", "n": 500, "seed": 7, "mask": 63, "add": 32}]}, [1]),
    "exact_block_99981": ({"kind": "repeat", "unit_hex": "6162636465666768696a", "n": 99981}, [1]),
    "exact_2blocks": ({"kind": "repeat", "unit_hex": "6162636465666768696a6b", "n": 199962}, [1]),
    "random2sym": ({"kind": "xorshift", "n": 400000, "seed": 4242, "mask": 1, "add": 48}, [1]),
    "random4sym": ({"kind": "xorshift", "n": 500000, "seed": 777, "mask": 3, "add": 97}, [1, 2]),
    "textgen_3M": ({"kind": "textgen", "n": 3000000, "seed": 2}, [1, 9]),
    "sample4": ({"kind": "file", "name": "sample4.ref"}, [1, 3]),
    "sample2": ({"kind": "file", "name": "sample2.ref"}, [1]),
}


@pytest.mark.parametrize("name", sorted(RLE_INPUTS))
def test_rle1_stage(hip, oracle, name):
    recipe, levels = RLE_INPUTS[name]
    data = recipes.build(recipe)
    for level in levels:
        want = oracle.rle1_blocks(data, level)
        rc, got = hip.stage_rle1(data, level)
        assert rc == 0
        assert len(got) == len(want), "level %d: %d blocks, want %d" % (level, len(got), len(want))
        for k, ((gb, gcrc, gs), (wb, wcrc, ws, we)) in enumerate(zip(got, want)):
            assert gs == ws, "level %d block %d start %d want %d" % (level, k, gs, ws)
            assert gb.size == wb.size, "level %d block %d len %d want %d" % (level, k, gb.size, wb.size)
            if not np.array_equal(gb, wb):
                bad = np.nonzero(gb != wb)[0]
                raise AssertionError("level %d block %d: %d bytes differ, first at %d" % (level, k, bad.size, bad[0]))
            assert gcrc == wcrc, "level %d block %d crc %08x want %08x" % (level, k, gcrc, wcrc)


MTF_INPUTS = ["banana", "aaaa", "mary", "zeros_5000", "ab_10000", "range256", "random_multi", "random4sym", "textgen_multi", "textgen_900k", "sample3", "sample1"]


@pytest.mark.parametrize("name", MTF_INPUTS)
def test_mtf_stage(hip, oracle, name):
    data, block_len = BWT_INPUTS[name]
    data = np.ascontiguousarray(data, dtype=np.uint8)
    nb = -(-data.size // block_len)
    U = np.concatenate([oracle.bwt_cyclic(data[k * block_len:(k + 1) * block_len])[0] for k in range(nb)])
    rc, A, npos, freq, asz = hip.stage_mtf(U, data, block_len)
    assert rc == 0
    for k in range(nb):
        blk = data[k * block_len:(k + 1) * block_len]
        wa, wf, wasz = oracle.mtf_rle2(U[k * block_len:k * block_len + blk.size], blk)
        assert asz[k] == wasz
        assert npos[k] == wa.size, "block %d: npos %d want %d" % (k, npos[k], wa.size)
        got = A[k, :wa.size]
        if not np.array_equal(got, wa):
            bad = np.nonzero(got != wa)[0]
            raise AssertionError("block %d: %d symbols differ, first at %d (got %d want %d)" % (k, bad.size, bad[0], got[bad[0]], wa[bad[0]]))
        assert np.array_equal(freq[k, :wasz + 2], wf)


@pytest.mark.parametrize("name", MTF_INPUTS)
def test_huff_stage(hip, oracle, name):
    data, block_len = BWT_INPUTS[name]
    data = np.ascontiguousarray(data, dtype=np.uint8)
    blk = data[:block_len]
    U, _ = oracle.bwt_cyclic(blk)
    A, freq, asz = oracle.mtf_rle2(U, blk)
    wng, wsel, wlens = oracle.huff_groups(A, asz)
    rc, ng, sel, lens = hip.stage_huff(A, asz)
    assert rc == 0
    assert ng == wng
    assert np.array_equal(lens, wlens), "code lengths differ"
    assert np.array_equal(sel, wsel), "selectors differ at %s" % np.nonzero(sel != wsel)[0][:5]


STREAMS = ["tiny_empty", "tiny_a", "tiny_banana", "sample0", "sample1", "sample3", "zeros_300000", "ab_10000", "range256_70000",
           "random_250000", "random4sym_200000", "q2_run_at_block_end", "q2_count_fills_block", "q2_run_crosses_block",
           "exact_block_99981", "exact_2blocks_199962", "long_runs_mixed", "run4_at_eof", "textgen_1000000_s1", "textgen_65536_s1"]


def _golden_cases():
    import support
    g = support.load_golden("golden_small.json")
    return [c for c in g["cases"] if c["algo"] == "Bzip2"]


@pytest.mark.parametrize("case", _golden_cases(), ids=lambda c: "%s-%d" % (c["name"], c["level"]))
def test_bzip2_compress_golden(hip, case):
    import support
    data = recipes.build(case["recipe"])
    rc, out = hip.bzip2_compress(data, case["level"])
    assert rc == 0, hip.L.cjs_strerror(rc)
    assert out.size == case["out_len"], "length %d want %d" % (out.size, case["out_len"])
    assert support.sha256(out) == case["out_sha256"]
