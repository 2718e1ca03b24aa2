#!/usr/bin/env python
"""bench.py — headline metric of BASELINE.json: bzip2 -9 compress MB/s on MI355X, bit-exact vs reference.

  python bench.py --gpus N --steps K --warmup W
  N > 1 without a launcher: this process starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD
  (before anything here touches the GPU), relays its JSON line and exits with its code.  Under a launcher (RANK / WORLD_SIZE
  set) it is one rank of the job.

Workload (config.workload): BASELINE.json configs[2] — Bzip2.compressFile level 9 (899,981-byte blocks)
on 100,000,000 bytes of the synthetic enwik8-shaped stream (tools/textgen.c, seed 1) PER GPU.
A "step" = one pass of the whole hot path (RLE1+CRC+block boundaries -> cyclic BWT -> MTF/RLE2 ->
Huffman tables -> bit packing) over that input, input resident in HBM, output left in HBM.

N > 1 (weak scaling: the stream is N x 100 MB, the N=1 stream is its prefix; --strong: ONE stream of 2^30 bytes, the north
star's 1 GiB).  Every rank holds the stream.  A step is the three phases of include/cjs_hip.h, with the exchanges INSIDE the
timed region:
  1. boundary tables of the rank's 1/N of the input tiles          -> all-gather of the shares (72 B per 4 KiB tile; RCCL)
  2. boundary walk (replicated, serial by the format) + the rank's contiguous range of blocks through the Huffman tables
                                                                    -> all-gather of one 32-byte meta per rank
  3. the rank's blocks packed at their FINAL bit offset: the ranks' fragments are disjoint runs of whole words of the one
     .bz2 stream (rank 0 writes the header, the last rank with blocks the trailer + combined CRC), left in the ranks' HBM.
The path has no cross-block data collective; what is exchanged is boundary metadata.  After the clock stops rank 0 concatenates
the fragments and checks the stream (reference golden at 100 MB / 2^30 bytes, else decompressed by the oracle and compared).

At N=1 the same run also times the other BASELINE.json configs as extra keys of the JSON line (each
median of 5 after one warm-up, each checked against its golden / by round trip): `e2e_host_buffer`
(configs[2] through the host-buffer C ABI: H2D + kernels + D2H), `e2e_js_front` (the same call from Node through
js/index.js on a Uint8Array — the real boundary), `bzip2_1_compress` (configs[1], device-resident), `bzip2_9_decompress`
(configs[4], host-buffer ABI), `bwtc_9_compress` (configs[3] at 100 MB, host-buffer ABI, GPU / serial-coder split).
At N>1 rank 0 times `bzip2_9_decompress` and `bwtc_9_compress` on the whole stream with cjs_opts.n_devices = N (the other
ranks wait at a host-side barrier): configs[4] and [3] on N GPUs, "replicable stages x N, serial tail x 1".

One JSON line on stdout (rank 0).  `roofline` = dominant kernel (LSD radix scatter of the suffix sort)
priced at SURVEY.md §8(d)'s algorithmic bytes (1 B read + out/in B written per input byte) x the
suffixes one FULL-SIZE launch processes, over the live hipEvent-measured duration of those launches.  `cpu_baseline` = the oracle
(plain C restatement of the reference algorithm, 1 thread) on the same 100 MB input, timed in this run;
it is also the bit-exactness check at full size.  (The reference JS itself cannot travel to the GPU box: its figure,
0.41 MB/s under Node 12 in the build container, is in BASELINE.md.)
"""
import argparse
import ctypes
import hashlib
import importlib
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
PROFILE_DIR = "r03_final"          # profiles/<dir>/pmc_traffic.json: PMC pass of this command (tools/gpu_round_profile.sh)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--level", type=int, default=9)
    ap.add_argument("--mb", type=int, default=100, help="input MB (10^6 bytes) per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for the exchanges and barriers (nccl = RCCL)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: one 2^30-byte stream split over the ranks")
    ap.add_argument("--strong-mb", type=int, default=0, help="strong scaling on this many 10^6 bytes instead of 2^30")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra configs")
    return ap.parse_args()


def spawn_ranks(args):
    """--gpus N > 1 and no launcher: start the N ranks as a child job.  Nothing in this process has touched the GPU (torch is
    not even imported), so the child owns the devices; this process only relays stdout/stderr and the exit code."""
    port = 29000 + os.getpid() % 3000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    log("[bench] --gpus %d without a launcher: starting %s" % (args.gpus, " ".join(cmd[1:9])))
    return subprocess.call(cmd, env=env)


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


def _golden(name):
    p = os.path.join(ROOT, "tests", "golden", name)
    return json.load(open(p))["cases"][0] if os.path.exists(p) else None


def _matches(out, g):
    return bool(g is not None and out.size == g["out_len"] and hashlib.sha256(out.tobytes()).hexdigest() == g["out_sha256"])


class _Opts(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("device", ctypes.c_int32), ("n_devices", ctypes.c_uint32),
                ("flags", ctypes.c_uint32), ("stats", ctypes.c_void_p)]


def host_abi_calls(pkg, n_devices):
    """cjs_bzip2_compress / _decompress / cjs_bwtc_compress through the C ABI with cjs_opts.n_devices set"""
    L = pkg.load_library()
    u8p = ctypes.POINTER(ctypes.c_uint8)
    st = pkg.Stats()
    opts = _Opts(ctypes.sizeof(_Opts), -1, n_devices, 0, ctypes.cast(ctypes.pointer(st), ctypes.c_void_p))

    def call(fn, data, *mid):
        out, out_n = u8p(), ctypes.c_size_t(0)
        rc = fn(data.ctypes.data_as(u8p), data.size, *mid, ctypes.byref(out), ctypes.byref(out_n), ctypes.byref(opts))
        assert rc == 0, "%s failed: %d" % (fn.__name__, rc)
        return pkg._adopt(out, out_n.value)          # zero copy: cjs_free runs when the array is dropped (what the N-API addon does)
    return (lambda d, lvl: call(L.cjs_bzip2_compress, d, lvl), lambda d: call(L.cjs_bzip2_decompress, d, 0),
            lambda d, lvl: call(L.cjs_bwtc_compress, d, lvl), st)


def js_front_leg(data, level, verify):
    """configs[2] end to end from a JS Uint8Array: node runs js/index.js Bzip2.compressFile in a process of its own"""
    node = shutil.which("node")
    if node is None:
        return {"skipped": "node is not installed on this box"}
    addon = os.path.join(ROOT, "compressjs-flattened_amd", "js", "cjs_napi.node")
    if not os.path.exists(addon):
        return {"skipped": "cjs_napi.node was not built (node_api.h missing at build time)"}
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "cjs_bench_input_%d.bin" % os.getpid())
    data.tofile(path)
    try:
        out = subprocess.run([node, "--expose-gc", os.path.join(ROOT, "compressjs-flattened_amd", "js", "bench_front.js"), path, str(level), "5"],
                             capture_output=True, text=True, timeout=600)
    finally:
        os.unlink(path)
    if out.returncode != 0:
        return {"skipped": "node front failed: " + out.stderr.strip()[-300:]}
    r = json.loads(out.stdout.strip().splitlines()[-1])
    ok = bool(r["out_len"] == verify.get("out_len") and r["out_sha256"] == verify.get("out_sha256"))
    return {"workload": "js/index.js Bzip2.compressFile(Uint8Array of %d bytes, level %d) under Node %s: N-API shim -> C ABI (H2D + kernels + D2H), median of %d"
                        % (data.size, level, r["node"], r["reps"]),
            "MBps": round(data.size / (r["median_ms"] / 1e3) / 1e6, 1), "ms": round(r["median_ms"], 2),
            "ms_results_collected_between_calls": round(r["median_ms_results_collected"], 2) if r.get("median_ms_results_collected") else None,
            "note": "ms: calls back to back in one synchronous stretch of JS (V8 finalises dropped results late: every call pins a fresh result buffer); "
                    "the second figure has the dropped result collected between the calls (its pinned buffer is reused)",
            "verify": ok}


def extra_configs_one_gpu(pkg, data, verify, torch, dev):
    """BASELINE.json configs[1], [3] (at 100 MB), [4] and the end-to-end protocol of configs[2]; wall clock, median of 5."""
    n = int(data.size)
    mb = n // 1_000_000
    ex = {}
    # configs[2] end to end: host Uint8Array in, host Uint8Array out (H2D + kernels + D2H; workspace kept between calls)
    dt, c9 = _median_time(lambda: pkg.Bzip2.compressFile(data, None, 9))
    ex["e2e_host_buffer"] = {"workload": "Bzip2.compressFile level 9, %d bytes, host buffer in / host buffer out (Python ctypes over the C ABI)" % n, "MBps": round(n / dt / 1e6, 1),
                             "ms": round(dt * 1e3, 2), "verify": bool(c9.size == verify.get("out_len") and hashlib.sha256(c9.tobytes()).hexdigest() == verify.get("out_sha256"))}
    # configs[4]: decompress the reference-identical -9 stream
    dt, back = _median_time(lambda: pkg.Bzip2.decompressFile(c9))
    ex["bzip2_9_decompress"] = {"workload": "Bzip2.decompressFile of the level-9 stream of the %d-byte input, host buffers" % n, "MBps": round(n / dt / 1e6, 1),
                                "ms": round(dt * 1e3, 2), "verify": bool(back.size == n and np.array_equal(back, data))}
    del back
    pkg.trim()
    ex["e2e_js_front"] = js_front_leg(data, 9, verify)
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
                              "ms": round(dt * 1e3, 2), "verify": _matches(o1, _golden("golden_big_bzip2_1_%dm.json" % mb))}
    # configs[3] at this size: BWTC level 9; the range coder is one serial host chain over the GPU-produced step lists
    _, _, bwtc, st = host_abi_calls(pkg, 0)
    dt, w9 = _median_time(lambda: bwtc(data, 9), reps=5)
    ex["bwtc_9_compress"] = {"workload": "BWTC.compressFile level 9 (900,000-byte blocks), %d bytes, host buffers" % n, "MBps": round(n / dt / 1e6, 1),
                             "ms": round(dt * 1e3, 2), "gpu_ms": round(st.ms_bwt, 2), "first_step_list_ms": round(st.ms_mtf, 2),
                             "serial_coder_ms": round(st.ms_pack, 2), "coder_waited_for_gpu_ms": round(st.ms_rle1, 2),
                             "verify": _matches(w9, _golden("golden_big_bwtc_9_%dm.json" % mb))}
    pkg.trim()
    for k, v in ex.items():
        assert v.get("verify", True), "extra config %s failed its check: %s" % (k, v)
    return ex


def extra_configs_multi_gpu(pkg, data, stream, n_gpus, force0):
    """configs[4] and [3] on N GPUs, driven by rank 0 through the host-buffer C ABI with cjs_opts.n_devices = N (one worker thread
    per device; with BENCH_FORCE_DEVICE0 rehearsals the shards share GPU 0).  `stream` = the assembled .bz2 of `data`."""
    n = int(data.size)
    ex = {}
    _, bunzip, bwtc, st = host_abi_calls(pkg, n_gpus)
    reps = 3 if n > 400_000_000 else 5
    dt, back = _median_time(lambda: bunzip(stream), reps=reps)
    ex["bzip2_9_decompress"] = {"workload": "Bzip2.decompressFile of the level-9 stream of the %d-byte input, host buffers, byte-range shares over %d GPU(s)%s"
                                            % (n, n_gpus, " (rehearsal: all shares on GPU 0)" if force0 else ""),
                                "n_devices": n_gpus, "MBps": round(n / dt / 1e6, 1), "ms": round(dt * 1e3, 2), "verify": bool(back.size == n and np.array_equal(back, data))}
    del back
    pkg.trim()
    dt, w9 = _median_time(lambda: bwtc(data, 9), reps=reps)
    g = _golden("golden_big_bwtc_9_1g.json") if n == 1 << 30 else _golden("golden_big_bwtc_9_%dm.json" % (n // 1_000_000)) if n % 1_000_000 == 0 else None
    import support
    if g is not None:
        ok = _matches(w9, g)
    else:                                   # no reference golden at this size: the oracle's decoder must give the input back
        rc, rt = support.Oracle().bwtc_decompress(w9)
        ok = bool(rc == 0 and rt.size == n and np.array_equal(rt, data))
    ex["bwtc_9_compress"] = {"workload": "BWTC.compressFile level 9 (900,000-byte blocks), %d bytes, host buffers, block ranges over %d GPU(s)%s; "
                                         "replicable stages (sentinel BWT, MTF/RLE2, model) x %d, serial tail (ONE host range coder, serial by the format) x 1"
                                         % (n, n_gpus, " (rehearsal: all slots on GPU 0)" if force0 else "", n_gpus),
                             "n_devices": n_gpus, "MBps": round(n / dt / 1e6, 1), "ms": round(dt * 1e3, 2), "gpu_ms_longest_batch": round(st.ms_bwt, 2),
                             "first_step_list_ms": round(st.ms_mtf, 2), "serial_coder_ms": round(st.ms_pack, 2), "coder_waited_for_gpu_ms": round(st.ms_rle1, 2),
                             "checked_against": "reference golden" if g is not None else "oracle round trip", "verify": ok}
    pkg.trim()
    for k, v in ex.items():
        assert v["verify"], "extra config %s failed its check: %s" % (k, v)
    return ex


def main():
    args = parse_args()
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        sys.exit(spawn_ranks(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    rank = int(os.environ.get("RANK", "0")) if launched else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if launched else 0
    assert world == args.gpus, "--gpus %d but the launcher started %d ranks: pass the same number to both" % (args.gpus, world)
    n_gpus = world
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    force0 = bool(os.environ.get("BENCH_FORCE_DEVICE0"))      # rehearsal of the N>1 path on a one-GPU box (use with --backend gloo)
    if force0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_nccl = args.backend == "nccl"
    host_pg = None
    if world > 1:
        if use_nccl:
            dist.init_process_group(backend="nccl", device_id=dev)
            host_pg = dist.new_group(backend="gloo")          # host-side barriers: ranks that wait must not spin on their GPU
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

    # exchange buffers of the N>1 step (device for RCCL, host for gloo)
    if n_gpus > 1:
        share_bytes = ctx.share_bytes(stream_bytes, n_gpus)
        d_share = torch.zeros(share_bytes, dtype=torch.uint8, device=dev)
        d_shares = torch.zeros(share_bytes * n_gpus, dtype=torch.uint8, device=dev)
        meta_dev = torch.zeros(5, dtype=torch.int64, device=dev if use_nccl else "cpu")
        metas_dev = torch.zeros(5 * n_gpus, dtype=torch.int64, device=dev if use_nccl else "cpu")

    def gather_shares():
        if use_nccl:
            dist.all_gather_into_tensor(d_shares, d_share)
            torch.cuda.current_stream().synchronize()
        else:
            parts = [torch.empty(share_bytes, dtype=torch.uint8) for _ in range(n_gpus)]
            dist.all_gather(parts, d_share.cpu())
            d_shares.copy_(torch.cat(parts))
            torch.cuda.current_stream().synchronize()

    def gather_metas(m):
        mine = torch.tensor([m.bits, m.total_blocks, m.first_block, m.blocks, m.crc_fold], dtype=torch.int64)
        if use_nccl:
            meta_dev.copy_(mine)
            dist.all_gather_into_tensor(metas_dev, meta_dev)
            allm = metas_dev.cpu().tolist()
        else:
            parts = [torch.empty(5, dtype=torch.int64) for _ in range(n_gpus)]
            dist.all_gather(parts, mine)
            allm = torch.cat(parts).tolist()
        out = []
        for r in range(n_gpus):
            b, t, f, nblk, fold = allm[5 * r: 5 * r + 5]
            out.append(pkg.ShardMeta(b, t, f, nblk, fold))
        return out

    def step(stats=None):
        if n_gpus == 1:
            return ctx.compress(d_in.data_ptr(), stream_bytes, d_out.data_ptr(), out_cap, stats)
        ctx.shard_tiles(d_in.data_ptr(), stream_bytes, rank, n_gpus, d_share.data_ptr())
        gather_shares()
        meta = ctx.shard_blocks(d_in.data_ptr(), stream_bytes, rank, n_gpus, d_shares.data_ptr(), stats)
        metas = gather_metas(meta)
        return ctx.shard_pack(rank, metas, d_out.data_ptr(), out_cap), metas

    def barrier():
        if world > 1:
            dist.barrier()

    def host_barrier():
        if world > 1:
            dist.barrier(group=host_pg) if host_pg is not None else dist.barrier()

    for _ in range(args.warmup):
        step()
    agg = {"dom_ms": 0.0, "dom_launches": 0, "dom_elems": 0, "stage": {}, "step_ms": []}
    barrier()
    torch.cuda.synchronize()
    ctx.set_stage_times(False)
    t0 = time.perf_counter()
    res = None
    for _ in range(args.steps):
        # inside the timed loop the library only records events (whole call + every full-size launch of the dominant kernel): the
        # per-stage times need a stream synchronisation per stage and come from one extra, untimed step below
        st = pkg.Stats()
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
    ctx.set_stage_times(True)
    step(st)                                             # untimed: stage breakdown (synchronises between the stages)
    for k in ("ms_total", "ms_rle1", "ms_bwt", "ms_mtf", "ms_huff", "ms_pack"):
        agg["stage"][k] = getattr(st, k) * args.steps
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if use_nccl else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---------------- verification (outside the timed region)
    verify = {}
    stream = None
    if n_gpus == 1:
        out_n = res
        stream = d_out[:out_n].cpu().numpy()
    else:
        (fo, fl, so, sl), metas = res
        frag = d_out[fo:fo + fl].cpu().numpy()
        parts = [None] * world
        dist.all_gather_object(parts, (int(so), int(sl), frag.tobytes()), group=host_pg)
        if rank == 0:
            pos, chunks = 0, []
            for r, (o, l, b) in enumerate(parts):
                assert l == parts[0][1], "ranks disagree on the stream length"
                if len(b):
                    assert o == pos, "rank %d's fragment starts at stream byte %d, expected %d" % (r, o, pos)
                    chunks.append(np.frombuffer(b, dtype=np.uint8))
                    pos += len(b)
            assert pos == parts[0][1], "fragments cover %d of %d stream bytes" % (pos, parts[0][1])
            stream = np.concatenate(chunks)
            verify["fragments"] = [len(b) for _, _, b in parts]
            verify["blocks"] = int(metas[0].total_blocks)
    if rank == 0:
        verify["out_len"] = int(stream.size)
        verify["out_sha256"] = hashlib.sha256(stream.tobytes()).hexdigest()
        gname = "golden_big_bzip2_%d_1g.json" % args.level if stream_bytes == 1 << 30 else "golden_big_bzip2_%d_%dm.json" % (args.level, stream_bytes // 1_000_000)
        g = _golden(gname) if stream_bytes == 1 << 30 or stream_bytes % 1_000_000 == 0 else None
        if g is not None:
            verify["golden"] = gname
            verify["bit_exact_vs_reference_js"] = bool(g["out_len"] == stream.size and g["out_sha256"] == verify["out_sha256"])
            if not args.no_verify:
                assert verify["bit_exact_vs_reference_js"], "output differs from the reference JS golden (%s)" % gname
        elif not args.no_verify:
            # no reference golden at this size: the oracle's decoder (CPU restatement of Bunzip) must give the input back
            t1 = time.perf_counter()
            rc, back = support.Oracle().bzip2_decompress(stream)
            verify["oracle_round_trip"] = bool(rc == 0 and back.size == stream_bytes and np.array_equal(back, data))
            verify["oracle_round_trip_s"] = round(time.perf_counter() - t1, 1)
            assert verify["oracle_round_trip"], "the assembled stream does not decode to the input (oracle rc %d)" % rc
    out_bytes_total = int(stream.size) if rank == 0 else 0

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
                        "sample": "%s: %d bytes, oracle/cjs_oracle.c (SA-IS restatement of the reference algorithm), 1 thread, %.1f s; the reference JS cannot travel to "
                                  "this box (its container-only figure: 0.41 MB/s under Node 12, BASELINE.md)"
                                  % ("the full workload" if sample.size == stream_bytes else "the first 100 MB of the workload", sample.size, dt)}

    # ---------------- the other BASELINE.json configs, timed in the same run (never part of `value`)
    extra = {}
    if not args.no_extra and args.level == 9 and not args.no_verify:
        ctx.close()
        ctx = None
        del d_out, d_in
        if n_gpus > 1:
            del d_share, d_shares
        torch.cuda.empty_cache()
        host_barrier()
        if rank == 0:
            if n_gpus == 1 and not strong:
                extra = extra_configs_one_gpu(pkg, data, verify, torch, dev)
            elif n_gpus > 1:
                extra = extra_configs_multi_gpu(pkg, data, stream, n_gpus, force0)
            elif strong:                     # --strong --gpus 1: the north star's 1 GiB stream, decompress made driver-measurable
                _, bunzip, _, _ = host_abi_calls(pkg, 0)
                dt, back = _median_time(lambda: bunzip(stream), reps=3)
                extra["bzip2_9_decompress"] = {"workload": "Bzip2.decompressFile of the level-9 stream of the %d-byte input, host buffers, 1 GPU" % stream_bytes,
                                               "MBps": round(stream_bytes / dt / 1e6, 1), "ms": round(dt * 1e3, 2), "verify": bool(back.size == stream_bytes and np.array_equal(back, data))}
                assert extra["bzip2_9_decompress"]["verify"]
                del back
                pkg.trim()
        host_barrier()

    if rank == 0:
        total_in = stream_bytes * args.steps
        value = total_in / elapsed / 1e6
        ratio = (stream_bytes + out_bytes_total) / stream_bytes          # 1 B read + out/in B written
        dom_s = agg["dom_ms"] / 1e3
        achieved = ratio * agg["dom_elems"] / dom_s / 1e9 if dom_s > 0 else 0.0
        traffic = {}
        tpath = os.path.join(ROOT, "profiles", PROFILE_DIR, "pmc_traffic.json")
        if not os.path.exists(tpath):
            tpath = os.path.join(ROOT, "profiles", "r02_v12", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath))
            except Exception:
                traffic = {}
        pipe_traffic = traffic.get("pipeline_hbm_bytes_per_step")
        line = {
            "metric": "bzip2 -9 compress MB/s at 1/2/4/8 MI355X; bit-exact output size vs ref",
            "value": round(value, 3), "unit": "MB/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "median_step_ms": round(float(np.median(agg["step_ms"])), 3),
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "Bzip2.compressFile level %d (%d-byte blocks) on %d x %d bytes enwik8-shaped synthetic text (tools/textgen.c seed 1), device-resident"
                                   % (args.level, cap, n_gpus, per),
                       "per_gpu_bytes": per, "level": args.level,
                       "parallelism": ("blocks sharded by contiguous range over %d GPUs; boundary tables sharded by input tile + all-gather, one 32-byte meta per rank all-gathered, "
                                       "fragments packed at their final bit offset; no data-path collective" % n_gpus) if n_gpus > 1 else "1 GPU"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": traffic.get("rs_scatter_full_hbm_bytes_per_launch", traffic.get("rs_scatter_hbm_bytes_per_launch")),
                         "kernel": "rs_scatter (LSD radix scatter of the suffix sort), full-size launches (one pass over all suffixes of the step) only",
                         "launches_per_step": agg["dom_launches"] // max(args.steps, 1),
                         "avg_launch_ms": round(agg["dom_ms"] / max(agg["dom_launches"], 1), 4),
                         "algorithmic_bytes_per_input_byte": round(ratio, 4),
                         "pipeline_achieved_GBs": round(ratio * total_in / elapsed / 1e9, 3),
                         "pipeline_traffic_bytes": pipe_traffic,
                         "traffic_ratio": round(pipe_traffic / (ratio * traffic.get("input_bytes", 100_000_000)), 1) if pipe_traffic else None,
                         "traffic_source": os.path.relpath(tpath, ROOT) if traffic else None},
            "cpu_baseline": cpu_baseline,
            "stage_ms_per_step": {k: round(v / args.steps, 3) for k, v in agg["stage"].items()},      # one extra untimed step with per-stage synchronisation (rank 0)
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
