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
--strong: ONE stream of 2^30 bytes (the north star's 1 GiB input; --strong-mb for rehearsals) split over the
N ranks the same way; rank 0 assembles the ranks' bit strings and checks the stream against the reference
JS golden (golden_big_bzip2_9_1g.json).

At N=1 the same run also times the other BASELINE.json configs as extra keys of the JSON line (each
median of 5 after one warm-up, each checked against its golden / by round trip): `e2e_host_buffer`
(configs[2] through the host-buffer C ABI: H2D + kernels + D2H), `bzip2_1_compress` (configs[1],
device-resident), `bzip2_9_decompress` (configs[4], host-buffer ABI), `bwtc_9_compress` (configs[3]
at 100 MB, host-buffer ABI, GPU / serial-coder split).

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


def _median_time(fn, reps=5, warm=1):
    out = None
    for _ in range(warm):
        out = fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), out


def extra_configs(pkg, support, data, verify, args, torch, dev):
    """BASELINE.json configs[1], [3] (at 100 MB), [4] and the end-to-end protocol of configs[2]; wall clock, median of 5."""
    n = int(data.size)
    mb = n // 1_000_000
    ex = {}

    def golden(name):
        p = os.path.join(ROOT, "tests", "golden", name)
        return json.load(open(p))["cases"][0] if os.path.exists(p) else None

    def check(out, g):
        return bool(g is not None and out.size == g["out_len"] and hashlib.sha256(out.tobytes()).hexdigest() == g["out_sha256"])

    # configs[2] end to end: host Uint8Array in, host Uint8Array out (H2D + kernels + D2H; workspace kept between calls)
    dt, c9 = _median_time(lambda: pkg.Bzip2.compressFile(data, None, 9))
    ex["e2e_host_buffer"] = {"workload": "Bzip2.compressFile level 9, %d bytes, host buffer in / host buffer out" % n, "MBps": round(n / dt / 1e6, 1),
                             "ms": round(dt * 1e3, 2), "verify": bool(c9.size == verify.get("out_len") and hashlib.sha256(c9.tobytes()).hexdigest() == verify.get("out_sha256"))}
    # configs[4]: decompress the reference-identical -9 stream
    dt, back = _median_time(lambda: pkg.Bzip2.decompressFile(c9))
    ex["bzip2_9_decompress"] = {"workload": "Bzip2.decompressFile of the level-9 stream of the %d-byte input, host buffers" % n, "MBps": round(n / dt / 1e6, 1),
                                "ms": round(dt * 1e3, 2), "verify": bool(back.size == n and np.array_equal(back, data))}
    del back
    pkg.trim()
    # configs[1]: level 1 (1001 blocks of 99,981 bytes), device-resident like the headline
    d_in = torch.from_numpy(data).to(dev)
    out_cap = (n + n // 4 + (1 << 20)) & ~3
    d_out = torch.zeros(out_cap, dtype=torch.uint8, device=dev)
    ctx1 = pkg.DeviceContext(dev.index or 0, n, 1)
    dt, out_n = _median_time(lambda: ctx1.compress(d_in.data_ptr(), n, d_out.data_ptr(), out_cap))
    o1 = d_out[:out_n].cpu().numpy()
    ctx1.close()
    del d_in, d_out
    torch.cuda.empty_cache()
    ex["bzip2_1_compress"] = {"workload": "Bzip2.compressFile level 1 (99,981-byte blocks), %d bytes, device-resident" % n, "MBps": round(n / dt / 1e6, 1),
                              "ms": round(dt * 1e3, 2), "verify": check(o1, golden("golden_big_bzip2_1_%dm.json" % mb))}
    # configs[3] at this size: BWTC level 9; the range coder is one serial host chain over the GPU-produced step lists
    L = pkg.load_library()

    class Opts(ctypes.Structure):
        _fields_ = [("struct_size", ctypes.c_uint32), ("device", ctypes.c_int32), ("n_devices", ctypes.c_uint32),
                    ("flags", ctypes.c_uint32), ("stats", ctypes.POINTER(pkg.Stats))]
    st = pkg.Stats()
    opts = Opts(ctypes.sizeof(Opts), -1, 0, 0, ctypes.pointer(st))
    u8p = ctypes.POINTER(ctypes.c_uint8)

    def bwtc():
        out, out_n = u8p(), ctypes.c_size_t(0)
        rc = L.cjs_bwtc_compress(data.ctypes.data_as(u8p), n, 9, ctypes.byref(out), ctypes.byref(out_n), ctypes.byref(opts))
        assert rc == 0, "cjs_bwtc_compress failed: %d" % rc
        res = np.ctypeslib.as_array(out, shape=(out_n.value,)).copy()
        L.cjs_free(out)
        return res
    dt, w9 = _median_time(bwtc, reps=5)
    ex["bwtc_9_compress"] = {"workload": "BWTC.compressFile level 9 (900,000-byte blocks), %d bytes, host buffers" % n, "MBps": round(n / dt / 1e6, 1),
                             "ms": round(dt * 1e3, 2), "gpu_ms": round(st.ms_bwt, 2), "first_step_list_ms": round(st.ms_mtf, 2),
                             "serial_coder_ms": round(st.ms_pack, 2), "coder_waited_for_gpu_ms": round(st.ms_rle1, 2),
                             "verify": check(w9, golden("golden_big_bwtc_9_%dm.json" % mb))}
    pkg.trim()
    for k, v in ex.items():
        assert v["verify"], "extra config %s failed its check: %s" % (k, v)
    return ex


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
    ap.add_argument("--strong", action="store_true", help="strong scaling: one 2^30-byte stream split over the ranks")
    ap.add_argument("--strong-mb", type=int, default=0, help="strong scaling on this many 10^6 bytes instead of 2^30")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra configs timed at N=1")
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

    strong = args.strong or args.strong_mb > 0
    if strong:
        stream_bytes = args.strong_mb * 1_000_000 if args.strong_mb else 1 << 30
        per = stream_bytes // n_gpus
    else:
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
    agg = {"dom_ms": 0.0, "dom_launches": 0, "dom_elems": 0, "stage": {}, "step_ms": []}
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = None
    for _ in range(args.steps):
        # inside the timed loop the library only records events (whole call + every launch of the dominant kernel): the
        # per-stage times need a stream synchronisation per stage and come from one extra, untimed step below
        st = pkg.Stats()
        st.flags = pkg.Stats.NO_STAGE_TIMES
        res = step(st)
        agg["dom_ms"] += st.ms_bwt_dominant * st.bwt_dominant_launches
        agg["dom_launches"] += st.bwt_dominant_launches
        agg["dom_elems"] += st.bwt_dominant_bytes
        agg["step_ms"].append(st.ms_total)
        last_stats = st
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    st = pkg.Stats()
    step(st)                                             # untimed: stage breakdown (synchronises between the stages)
    for k in ("ms_total", "ms_rle1", "ms_bwt", "ms_mtf", "ms_huff", "ms_pack"):
        agg["stage"][k] = getattr(st, k) * args.steps
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
        gname = "golden_big_bzip2_%d_1g.json" % args.level if stream_bytes == 1 << 30 else "golden_big_bzip2_%d_%dm.json" % (args.level, stream_bytes // 1_000_000)
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
        gname = "golden_big_bzip2_%d_1g.json" % args.level
        if not args.no_verify and strong and stream_bytes == 1 << 30 and os.path.exists(os.path.join(ROOT, "tests", "golden", gname)):
            # the ranks' bit strings assembled on rank 0 (host funnel shift, shard.py) must be the reference's 1 GiB stream
            shard = importlib.import_module("compressjs-flattened_amd.shard")
            parts = [None] * world
            dist.all_gather_object(parts, (mine[:nbytes].tobytes(), int(bits), crcs[first:first + count].tolist()))
            if rank == 0:
                stream = shard.assemble(args.level, [(np.frombuffer(b, dtype=np.uint8), nb) for b, nb, _ in parts], [c for _, _, cs in parts for c in cs])
                g = json.load(open(os.path.join(ROOT, "tests", "golden", gname)))["cases"][0]
                verify["golden"] = gname
                verify["bit_exact_vs_reference_js"] = bool(stream.size == g["out_len"] and hashlib.sha256(stream.tobytes()).hexdigest() == g["out_sha256"])
                assert verify["bit_exact_vs_reference_js"], "assembled stream differs from the reference JS golden"

    cpu_baseline = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        orc = support.Oracle()
        sample = data if stream_bytes <= 200_000_000 else data[:100_000_000]        # bounded: ~10-20 s of CPU work
        t1 = time.perf_counter()
        rc, want = orc.bzip2_compress(sample, args.level)
        dt = time.perf_counter() - t1
        if sample.size == stream_bytes:
            same = rc == 0 and want.size == verify["out_len"] and hashlib.sha256(want.tobytes()).hexdigest() == verify["out_sha256"]
            verify["bit_exact_vs_oracle_full_size"] = bool(same)
            assert same, "HIP output differs from the oracle at full size"
        cpu_baseline = {"value": round(sample.size / dt / 1e6, 3), "unit": "MB/s", "cores": 1, "kind": "port",
                        "sample": "%s: %d bytes, oracle/cjs_oracle.c (SA-IS restatement of the reference algorithm), 1 thread, %.1f s"
                                  % ("the full workload" if sample.size == stream_bytes else "the first 100 MB of the workload", sample.size, dt)}

    # ---------------- the other BASELINE.json configs, timed in the same run (N=1 only; never part of `value`)
    extra = {}
    if rank == 0 and n_gpus == 1 and not args.no_extra and not strong and args.level == 9:
        ctx.close()
        ctx = None
        del d_out
        torch.cuda.empty_cache()
        extra = extra_configs(pkg, support, data, verify, args, torch, dev)

    if rank == 0:
        total_in = stream_bytes * args.steps
        value = total_in / elapsed / 1e6
        ratio = (stream_bytes + out_bytes_total) / stream_bytes          # 1 B read + out/in B written
        dom_s = agg["dom_ms"] / 1e3
        achieved = ratio * agg["dom_elems"] / dom_s / 1e9 if dom_s > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")     # PMC pass of the same command (tools/gpu_round_profile.sh), committed per round
        if not os.path.exists(tpath):
            tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("rs_scatter_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "bzip2 -9 compress MB/s at 1/2/4/8 MI355X; bit-exact output size vs ref",
            "value": round(value, 3), "unit": "MB/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "median_step_ms": round(float(np.median(agg["step_ms"])), 3),
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
            "stage_ms_per_step": {k: round(v / args.steps, 3) for k, v in agg["stage"].items()},      # one extra untimed step with per-stage synchronisation
            "bwt_rounds": int(last_stats.bwt_rounds),
            "verify": verify,
        }
        line.update(extra)
        print(json.dumps(line), flush=True)
    if ctx is not None:
        ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
