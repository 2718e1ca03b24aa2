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
    "sample3": (recipes.build({"kind": "file", "name": "sample3.ref"}), 120244),
    "sample1": (recipes.build({"kind": "file", "name": "sample1.ref"}), 98696),
}


@pytest.mark.parametrize("name", sorted(BWT_INPUTS))
@pytest.mark.parametrize("cyclic", [True, False], ids=["cyclic", "sentinel"])
def test_bwt_stage(hip, oracle, name, cyclic):
    data, block_len = BWT_INPUTS[name]
    _check_bwt(hip, oracle, data, block_len, cyclic)
