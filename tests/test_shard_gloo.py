"""The N>1 path without GPUs: two gloo ranks each produce the bare bit string of their contiguous block
range (here with the oracle standing in for the per-rank HIP pipeline), exchange only (bit length, block
CRCs) + the strings, and rank 0's assembled stream must equal the single-rank stream bit for bit."""
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
    gathered = [None] * world
    dist.all_gather_object(gathered, (bits.tobytes(), int(nbits), crcs[first:first + count].tolist()))
    if rank == 0:
        parts = [(np.frombuffer(b, dtype=np.uint8), nb) for b, nb, _ in gathered]
        all_crcs = [c for _, _, cs in gathered for c in cs]
        stream = shard.assemble(level, parts, all_crcs)
        rc, want = orc.bzip2_compress(data, level)
        q.put(bool(rc == 0 and np.array_equal(stream, want)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("level,n", [(1, 1000000), (2, 450000)])
def test_two_rank_assembly_equals_single_stream(level, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, level, n, 11, q)) for r in range(2)]
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
