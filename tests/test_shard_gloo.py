"""The N>1 path without GPUs: gloo ranks each produce the bare bit string of their contiguous block range (here with
the oracle standing in for the per-rank HIP pipeline), all-gather one (bit length, block count, CRC fold) per rank, place
their blocks at the FINAL bit offset (shard.fragment: the Python mirror of pipeline.hip's shard_layout / shard_pack_core) and
rank 0's concatenation of the word-aligned fragments must equal the single-rank stream bit for bit."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, level, n, seed, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import recipes
    import support
    shard = importlib.import_module("compressjs-flattened_amd.shard")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = recipes.textgen(n, seed)            # every rank holds the stream (as in bench.py)
    orc = support.Oracle()
    _, _, _, total, _ = orc.bzip2_compress_range(data, level, 0, 0)
    first, count = shard.plan_ranges(total, world)[rank]
    rc, bits, nbits, _, crcs = orc.bzip2_compress_range(data, level, first, count)
    assert rc == 0
    import torch
    mine = torch.tensor([int(nbits), count, shard.fold_stream_crc(crcs[first:first + count].tolist())], dtype=torch.int64)
    parts = [torch.empty(3, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(parts, mine)                                  # the only exchange in front of the packing
    metas = [tuple(int(x) for x in p.tolist()) for p in parts]
    off, frag = shard.fragment(level, metas, rank, bits)
    gathered = [None] * world
    dist.all_gather_object(gathered, (off, frag.tobytes()))       # verification only
    if rank == 0:
        _, _, stream_len = shard.layout(metas)
        stream = shard.concat([(o, np.frombuffer(b, dtype=np.uint8)) for o, b in gathered], stream_len)
        rc, want = orc.bzip2_compress(data, level)
        q.put(bool(rc == 0 and np.array_equal(stream, want)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("level,n,world", [(1, 1000000, 2), (2, 450000, 2), (1, 250000, 4), (1, 0, 2)])
def test_rank_fragments_tile_the_single_stream(level, n, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, level, n, 11, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_plan_ranges_and_crc_fold():
    shard = importlib.import_module("compressjs-flattened_amd.shard")
    assert shard.plan_ranges(10, 4) == [(0, 3), (3, 3), (6, 3), (9, 1)]
    assert shard.plan_ranges(2, 4) == [(0, 1), (1, 1), (2, 0), (2, 0)]
    assert shard.plan_ranges(0, 2) == [(0, 0), (0, 0)]
    assert shard.fold_stream_crc([]) == 0
    assert shard.fold_stream_crc([0x80000001]) == 0x80000001
    assert shard.fold_stream_crc([0x80000001, 1]) == ((0x00000003) ^ 1)
    # chaining the ranks' folds = folding all block CRCs in order
    crcs = [0xDEADBEEF, 0x12345678, 0x80000001, 7, 0xFFFFFFFF, 0x0F0F0F0F, 1]
    for cut in ([3, 4], [1, 1, 5], [7, 0], [2, 2, 2, 1, 0, 0]):
        metas, k = [], 0
        for c in cut:
            metas.append((100, c, shard.fold_stream_crc(crcs[k:k + c])))
            k += c
        assert shard.chain_folds(metas) == shard.fold_stream_crc(crcs)
    # 33 blocks on one rank: the rotation count wraps
    many = list(range(1, 40))
    assert shard.chain_folds([(1, 33, shard.fold_stream_crc(many[:33])), (1, 6, shard.fold_stream_crc(many[33:]))]) == shard.fold_stream_crc(many)


def test_layout_tiles_the_stream():
    shard = importlib.import_module("compressjs-flattened_amd.shard")
    for metas in ([(1000, 2, 0), (777, 2, 0), (64, 1, 0)], [(96, 1, 0), (0, 0, 0)], [(0, 0, 0), (0, 0, 0)], [(2016, 3, 0), (2048, 3, 0), (31, 1, 0), (0, 0, 0)]):
        lay, writer, slen = shard.layout(metas)
        pos = 0
        for (_, off, n) in lay:
            if n:
                assert off == pos and off % 4 == 0
                pos += n
        assert pos == slen == (32 + sum(m[0] for m in metas) + 80 + 7) // 8
