#!/usr/bin/env python
"""bench.py — headline metric of BASELINE.json: bzip2 -9 compress MB/s on MI355X, bit-exact vs reference.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload): BASELINE.json configs[2] — Bzip2.compressFile level 9 (899,981-byte blocks)
on 100,000,000 bytes of the synthetic enwik8-shaped stream (tools/textgen.c, seed 1) PER GPU.
A "step" = one pass of the whole hot path (RLE1+CRC+block boundaries -> cyclic BWT -> MTF/RLE2 ->
Huffman tables -> bit packing) over that input, input resident in HBM, output left in HBM.
N>1 (weak scaling): the stream is N x 100 MB (the N=1 stream is its prefix); every rank holds the
stream, runs the cheap boundary pass on all of it and the per-block pipeline on its own contiguous range
of blocks (cjs_bzip2_compress_device_range).  The path has no cross-block collective; torch.distributed
(RCCL) is used only for the barriers / max-over-ranks of the timing contract.

One JSON line on stdout (rank 0).  `roofline` = dominant kernel (LSD radix scatter of the suffix sort)
priced at SURVEY.md §8(d)'s algorithmic bytes (1 B read + out/in B written per input byte) x the
suffixes one launch processes, over its live hipEvent-measured duration.  `cpu_baseline` = the oracle
(plain C restatement of the reference algorithm, 1 thread) on the same 100 MB input, timed in this run;
it is also the bit-exactness check at full size.
"""
import argparse
import ctypes
import hashlib
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PER_GPU_BYTES = 100_000_000
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--level", type=int, default=9)
    ap.add_argument("--mb", type=int, default=100, help="input MB (10^6 bytes) per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for the barriers (nccl = RCCL)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus != world and world > 1:
        log("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world))
    n_gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    if os.environ.get("BENCH_FORCE_DEVICE0"):      # rehearsal of the N>1 path on a one-GPU box (use with --backend gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)

    pkg = importlib.import_module("compressjs-flattened_amd")
    import recipes
    import support

    per = args.mb * 1_000_000
    stream_bytes = per * n_gpus
    t0 = time.time()
    data = recipes.textgen(stream_bytes, 1)
    log("[rank %d] generated %d bytes in %.1f s" % (rank, stream_bytes, time.time() - t0))
    d_in = torch.from_numpy(data).to(dev)
    cap = args.level * 100000 - 19
    nb_upper = stream_bytes // (cap * 4 // 5) + 2
    per_rank_blocks = -(-nb_upper // n_gpus) + 1 if n_gpus > 1 else 0
    ctx = pkg.DeviceContext(local_rank, stream_bytes, args.level, per_rank_blocks)
    out_cap = (per + per // 4 + (1 << 20)) & ~3
    d_out = torch.zeros(out_cap, dtype=torch.uint8, device=dev)

    if n_gpus > 1:
        _, total_blocks, _ = ctx.compress_range(d_in.data_ptr(), stream_bytes, 0, 0, d_out.data_ptr(), out_cap)
        share = -(-total_blocks // n_gpus)
        first = min(rank * share, total_blocks)
        count = min(share, total_blocks - first)
    else:
        total_blocks, first, count = None, 0, -1

    def step(stats=None):
        if n_gpus == 1:
            return ctx.compress(d_in.data_ptr(), stream_bytes, d_out.data_ptr(), out_cap, stats)
        bits, _, crcs = ctx.compress_range(d_in.data_ptr(), stream_bytes, first, count, d_out.data_ptr(), out_cap, stats)
        return bits, crcs

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    agg = {"dom_ms": 0.0, "dom_launches": 0, "dom_elems": 0, "stage": {}}
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = None
    for _ in range(args.steps):
        st = pkg.Stats()
        res = step(st)
        agg["dom_ms"] += st.ms_bwt_dominant * st.bwt_dominant_launches
        agg["dom_launches"] += st.bwt_dominant_launches
        agg["dom_elems"] += st.bwt_dominant_bytes
        for k in ("ms_total", "ms_rle1", "ms_bwt", "ms_mtf", "ms_huff", "ms_pack"):
            agg["stage"][k] = agg["stage"].get(k, 0.0) + getattr(st, k)
        last_stats = st
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---------------- verification (outside the timed region)
    verify = {}
    if n_gpus == 1:
        out_n = res
        out = d_out[:out_n].cpu().numpy()
        verify["out_len"] = int(out_n)
        verify["out_sha256"] = hashlib.sha256(out.tobytes()).hexdigest()
        gname = "golden_big_bzip2_%d_%dm.json" % (args.level, args.mb)
        gpath = os.path.join(ROOT, "tests", "golden", gname)
        if os.path.exists(gpath):
            g = json.load(open(gpath))["cases"][0]
            verify["golden"] = gname
            verify["bit_exact_vs_reference_js"] = bool(g["out_len"] == out_n and g["out_sha256"] == verify["out_sha256"])
            if not args.no_verify:
                assert verify["bit_exact_vs_reference_js"], "output differs from the reference JS golden (%s)" % gname
        out_bytes_total = int(out_n)
    else:
        bits, crcs = res
        nbytes = (bits + 7) // 8
        mine = d_out[: nbytes + 16].cpu().numpy().copy()
        ok = True
        if not args.no_verify:
            # make the rank's blocks a standalone .bz2 (header + blocks + trailer) and round-trip it with the oracle
            scrc = 0
            for c in crcs[first:first + count].tolist():
                scrc = (((scrc << 1) | (scrc >> 31)) & 0xFFFFFFFF) ^ c
            trailer = (0x177245385090 << 32) | scrc
            stream = np.zeros(4 + nbytes + 16, dtype=np.uint8)
            stream[:4] = np.frombuffer(b"BZh%d" % args.level, dtype=np.uint8)
            stream[4:4 + nbytes] = mine[:nbytes]
            bitpos = 32 + bits
            for i in range(80):
                if (trailer >> (79 - i)) & 1:
                    stream[(bitpos + i) >> 3] |= 0x80 >> ((bitpos + i) & 7)
            total_len = (bitpos + 80 + 7) // 8
            rc, back = support.Oracle().bzip2_decompress(stream[:total_len])
            ok = rc == 0
            lens = [None] * world
            dist.all_gather_object(lens, int(back.size) if ok else -1)
            if ok and all(l >= 0 for l in lens):
                off = sum(lens[:rank])
                ok = bool(np.array_equal(back, data[off:off + back.size])) and (rank != world - 1 or off + back.size == stream_bytes)
        oks = [None] * world
        dist.all_gather_object(oks, ok)
        bl = [None] * world
        dist.all_gather_object(bl, int(bits))
        verify["round_trip_all_ranks"] = bool(all(oks))
        verify["blocks"] = int(total_blocks)
        assert all(oks), "round trip failed on some rank: %s" % oks
        out_bytes_total = (sum(bl) + 32 + 80 + 7) // 8

    cpu_baseline = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        orc = support.Oracle()
        t1 = time.perf_counter()
        rc, want = orc.bzip2_compress(data, args.level)
        dt = time.perf_counter() - t1
        same = rc == 0 and want.size == verify["out_len"] and hashlib.sha256(want.tobytes()).hexdigest() == verify["out_sha256"]
        verify["bit_exact_vs_oracle_full_size"] = bool(same)
        assert same, "HIP output differs from the oracle at full size"
        cpu_baseline = {"value": round(stream_bytes / dt / 1e6, 3), "unit": "MB/s", "cores": 1, "kind": "port",
                        "sample": "the full workload: %d bytes, oracle/cjs_oracle.c (SA-IS restatement), 1 thread, %.1f s" % (stream_bytes, dt)}

    if rank == 0:
        total_in = stream_bytes * args.steps
        value = total_in / elapsed / 1e6
        ratio = (stream_bytes + out_bytes_total) / stream_bytes          # 1 B read + out/in B written
        dom_s = agg["dom_ms"] / 1e3
        achieved = ratio * agg["dom_elems"] / dom_s / 1e9 if dom_s > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("rs_scatter_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "bzip2 -9 compress MB/s at 1/2/4/8 MI355X; bit-exact output size vs ref",
            "value": round(value, 3), "unit": "MB/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "Bzip2.compressFile level %d (%d-byte blocks) on %d x %d bytes enwik8-shaped synthetic text (tools/textgen.c seed 1), device-resident"
                                   % (args.level, cap, n_gpus, per),
                       "per_gpu_bytes": per, "level": args.level, "parallelism": "blocks sharded by contiguous range over %d GPU(s), no collective" % n_gpus},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "kernel": "rs_scatter (LSD radix scatter of the suffix sort)",
                         "launches_per_step": agg["dom_launches"] // max(args.steps, 1),
                         "avg_launch_ms": round(agg["dom_ms"] / max(agg["dom_launches"], 1), 4),
                         "algorithmic_bytes_per_input_byte": round(ratio, 4),
                         "pipeline_achieved_GBs": round(ratio * total_in / elapsed / 1e9, 3)},
            "cpu_baseline": cpu_baseline,
            "stage_ms_per_step": {k: round(v / args.steps, 3) for k, v in agg["stage"].items()},
            "bwt_rounds": int(last_stats.bwt_rounds),
            "verify": verify,
        }
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
