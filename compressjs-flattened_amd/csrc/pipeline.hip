// pipeline.hip — per-GPU context (workspace + stream) and the bzip2 compress pipeline:
//   RLE1/CRC/boundaries -> batched cyclic BWT -> MTF/RLE2 -> Huffman tables -> bit packing.
// Mirrors the block loop of Bzip2.compressFile (J/Bzip2_joined_.js:2199-2249) for all blocks at once.
#include "cjs_internal.h"
#include "rle1.h"
#include "mtf.h"
#include "huff.h"
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <new>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

using namespace cjs;

namespace cjs { int select_device(const cjs_opts* opts); }

struct cjs_ctx {
  int device = 0, level = 0;
  uint32_t cap = 0;
  size_t max_input = 0, max_blocks = 0, range_blocks = 0;
  hipStream_t stream = nullptr;
  hipStream_t side = nullptr;            // block CRCs run here, beside the suffix sort
  hipStream_t tail = nullptr;            // MTF / Huffman tables of a finished piece run here, beside the suffix sort of the next piece
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_piece[8] = {}, ev_tail = nullptr;
  Arena arena;
  Rle1Work rle;
  BwtWork bwt;
  MtfWork mtf;
  HuffWork huff;
  uint8_t* d_blocks = nullptr;
  uint8_t* d_U = nullptr;
  uint32_t* d_pidx = nullptr;
  uint64_t* h_scalars = nullptr;   // pinned
  EventTimer timer;
  // phase state of a multi-GPU job (cjs_bzip2_shard_tiles -> _blocks -> _pack)
  uint32_t sh_nb = 0, sh_first = 0, sh_cnt = 0, sh_state = 0;
  bool stage_times = true;         // cjs_ctx_set_stage_times
};

extern "C" int cjs_ctx_create(cjs_ctx** out, int device, size_t max_input, int level) {
  return cjs_ctx_create_sharded(out, device, max_input, 0, level);
}

extern "C" int cjs_ctx_create_sharded(cjs_ctx** out, int device, size_t max_input, long max_range_blocks, int level) {
  CJS_GUARD_BEGIN
  if (!out) return CJS_E_INVALID_ARG;
  *out = nullptr;
  if (level < 1 || level > 9) return CJS_E_BAD_LEVEL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CJS_E_NO_DEVICE;
  if (device < 0) { if (hipGetDevice(&device) != hipSuccess) return CJS_E_NO_DEVICE; }
  if (device >= ndev) return CJS_E_INVALID_ARG;
  CJS_HIP_TRY(hipSetDevice(device));
  cjs_ctx* c = new (std::nothrow) cjs_ctx();
  if (!c) return CJS_E_OUT_OF_MEMORY;
  c->device = device; c->level = level; c->cap = (uint32_t)level * 100000u - 19u;
  if (max_input == 0) max_input = 1;
  c->max_input = max_input;
  c->max_blocks = Rle1Work::max_blocks_for(max_input, c->cap);
  c->range_blocks = (max_range_blocks > 0 && (size_t)max_range_blocks < c->max_blocks) ? (size_t)max_range_blocks : c->max_blocks;
  const size_t rb = c->range_blocks;
  const size_t elems = rb * c->cap;
  size_t bytes = Rle1Work::bytes_needed(max_input, c->cap, rb) + BwtWork::bytes_needed(elems) +
                 MtfWork::bytes_needed(rb, c->cap) + HuffWork::bytes_needed(rb, c->cap) +
                 2 * (elems + 512) + 4 * rb + 65536;
  int rc = c->arena.init(bytes);
  if (!rc) rc = c->rle.carve(c->arena, max_input, c->cap, rb);
  if (!rc) rc = c->bwt.carve(c->arena, elems);
  if (!rc) rc = c->mtf.carve(c->arena, rb, c->cap);
  if (!rc) rc = c->huff.carve(c->arena, rb, c->cap);
  if (!rc) {
    c->d_blocks = c->arena.take<uint8_t>(elems);
    c->d_U = c->arena.take<uint8_t>(elems);
    c->d_pidx = c->arena.take<uint32_t>(rb);
    if (!c->d_pidx) rc = CJS_E_OUT_OF_MEMORY;
  }
  if (!rc && hipStreamCreate(&c->stream) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipStreamCreateWithFlags(&c->tail, hipStreamNonBlocking) != hipSuccess) rc = CJS_E_HIP;
  for (int i = 0; i < 8 && !rc; i++) if (hipEventCreateWithFlags(&c->ev_piece[i], hipEventDisableTiming) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipHostMalloc((void**)&c->h_scalars, 256) != hipSuccess) rc = CJS_E_HIP;
  if (!rc) rc = c->timer.init(c->stream);
  if (rc) { cjs_ctx_destroy(c); return rc; }
  *out = c;
  return 0;
  CJS_GUARD_END(CJS_E_OUT_OF_MEMORY, CJS_E_HIP)
}

extern "C" void cjs_ctx_set_stage_times(cjs_ctx* c, int on) { if (c) c->stage_times = on != 0; }

extern "C" void cjs_ctx_destroy(cjs_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  c->timer.destroy();
  for (int i = 0; i < 8; i++) if (c->ev_piece[i]) (void)hipEventDestroy(c->ev_piece[i]);
  if (c->ev_tail) (void)hipEventDestroy(c->ev_tail);
  if (c->tail) (void)hipStreamDestroy(c->tail);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->side) (void)hipStreamDestroy(c->side);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->h_scalars) (void)hipHostFree(c->h_scalars);
  c->bwt.release_host();
  c->rle.release();
  c->arena.destroy();
  delete c;
}

// Per-device state kept between host-buffer calls (cjs_bzip2_compress); guarded by its mutex for the whole call.
// Slot 0 of a device serves the one-GPU call and the first shard of a multi-GPU call on that device; further slots serve the
// other shards that land on the same device (more shards than GPUs); the last slot is the boundary pass of a multi-GPU call.
constexpr int MAX_CACHED_DEVICES = 64, CACHE_SLOTS = 5, BOUNDARY_SLOT = CACHE_SLOTS - 1;
struct HostCache {
  std::mutex mu;
  cjs_ctx* ctx = nullptr;
  uint8_t *d_in = nullptr, *d_out = nullptr;
  size_t in_cap = 0, out_cap = 0;
  void release() {
    if (ctx) cjs_ctx_destroy(ctx);
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    ctx = nullptr; d_in = d_out = nullptr; in_cap = out_cap = 0;
  }
  // context for n input bytes at `level` on the CURRENT device (range_blocks as in cjs_ctx_create_sharded), staging buffers of at
  // least in_bytes / out_bytes (0 = not needed).  Grows, never shrinks; the caller holds mu.
  int ensure(size_t n, int level, long range_blocks, size_t in_bytes, size_t out_bytes) {
    if (!ctx || ctx->level != level || ctx->max_input < n || (range_blocks ? ctx->range_blocks != (size_t)range_blocks : ctx->range_blocks != ctx->max_blocks)) {
      if (ctx) { cjs_ctx_destroy(ctx); ctx = nullptr; }
      const int rc = cjs_ctx_create_sharded(&ctx, -1, n, range_blocks, level);
      if (rc) { ctx = nullptr; return rc; }
    }
    if (in_bytes && (in_cap < in_bytes || !d_in)) {
      if (d_in) (void)hipFree(d_in);
      d_in = nullptr; in_cap = 0;
      if (hipMalloc((void**)&d_in, in_bytes) != hipSuccess) return CJS_E_OUT_OF_MEMORY;
      in_cap = in_bytes;
    }
    if (out_bytes && (out_cap < out_bytes || !d_out)) {
      if (d_out) (void)hipFree(d_out);
      d_out = nullptr; out_cap = 0;
      if (hipMalloc((void**)&d_out, out_bytes) != hipSuccess) return CJS_E_OUT_OF_MEMORY;
      out_cap = out_bytes;
    }
    return 0;
  }
};
static HostCache g_host_cache[MAX_CACHED_DEVICES][CACHE_SLOTS];

// Shared body: stage 0..tables for the whole stream, then pack blocks [first, first+count).
static int compress_core_impl(cjs_ctx* c, const uint8_t* d_in, size_t n, int level, long first, long count, bool framed,
                              uint8_t* d_out, size_t out_cap, uint64_t* out_bits, uint32_t* block_crcs, long crc_cap,
                              long* total_blocks, cjs_stats* st);
static int compress_core(cjs_ctx* c, const uint8_t* d_in, size_t n, int level, long first, long count, bool framed,
                         uint8_t* d_out, size_t out_cap, uint64_t* out_bits, uint32_t* block_crcs, long crc_cap,
                         long* total_blocks, cjs_stats* st) {
  CJS_GUARD_BEGIN
  const int rc = compress_core_impl(c, d_in, n, level, first, count, framed, d_out, out_cap, out_bits, block_crcs, crc_cap, total_blocks, st);
  // an early return may leave the block-CRC kernels of the side stream in flight against a context the caller reuses
  if (rc && c) { if (c->side) (void)hipStreamSynchronize(c->side); if (c->stream) (void)hipStreamSynchronize(c->stream); }
  return rc;
  CJS_GUARD_END(CJS_E_OUT_OF_MEMORY, CJS_E_HIP)
}
// whole-call timing events; released on every return path
struct EvPair {
  hipEvent_t a = nullptr, b = nullptr;
  ~EvPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};
// blocks [f, f + cnt) of the stream whose boundaries the context's tables hold (nb blocks in all): RLE1 bytes + CRCs, suffix
// sort, MTF / RLE2, Huffman tables.  Everything but the bit packing; nothing here waits for the stream.
static int blocks_through_tables(cjs_ctx* c, const uint8_t* d_in, size_t n, uint32_t nb, uint32_t last_len, uint32_t f, uint32_t cnt, cjs_stats* st, bool stage_times) {
  hipStream_t s = c->stream;
  if (cnt > c->range_blocks) return CJS_E_INVALID_ARG;
  uint32_t n_last = c->cap;
  if (cnt && f + cnt == nb) n_last = last_len;        // (came to the host with the block count)
  // all per-block buffers below are indexed relative to `f`; only rle.block_len / block_crc are absolute
  CJS_TRY(rle1_finish(s, c->rle, d_in, n, f, cnt, c->d_blocks, c->side, c->ev_fork, c->ev_join));
  if (stage_times) { CJS_HIP_TRY(hipStreamSynchronize(s)); st->ms_rle1 = c->timer.stop(); }
  // The suffix sort keeps the memory system busy and the SIMDs idle; MTF / RLE2 and the Huffman tables are latency-bound chains
  // of small kernels that leave the memory system idle.  So the blocks go in `pieces` runs: while the sort of piece i + 1 runs on
  // the work stream, MTF and the tables of piece i run beside it on the tail stream (per-stage times: one piece, one stream).
  // Every piece pays the sort's ~90 launches again (each followed by ~6 us in which its write-back drains, and the tail rounds are
  // launch-bound whatever the piece holds): 100 MB in 2 / 3 / 4 pieces 12.5 / 13.2 / 14.2 ms against 11.7 in one; 2^30 bytes
  // (1,194 blocks) in 1 / 2 / 4 / 8 pieces 117.4 / 111.4 / 109.4 / 111.6 ms.  So: four pieces from 400 MB of blocks on.
  const uint32_t pieces = (stage_times || !c->tail || (uint64_t)cnt * c->cap < 400000000ull) ? 1u : 4u;
  if (cnt && pieces == 1) {
    if (stage_times) c->timer.start();
    CJS_TRY(bwt_run(s, c->bwt, c->d_blocks, cnt, c->cap, n_last, true, c->d_U, c->d_pidx, st, stage_times));
    if (stage_times) { st->ms_bwt = c->timer.stop(); c->timer.start(); }
    CJS_TRY(mtf_run(s, c->mtf, c->d_U, cnt, c->rle.block_len + f));
    if (stage_times) { CJS_HIP_TRY(hipStreamSynchronize(s)); st->ms_mtf = c->timer.stop(); c->timer.start(); }
    CJS_TRY(huff_tables_run(s, c->huff, cnt, c->mtf.b.A, c->mtf.b.a_stride, c->mtf.b.npos, c->mtf.b.asz, c->mtf.b.freq, c->mtf.b.alist));
    if (stage_times) { CJS_HIP_TRY(hipStreamSynchronize(s)); st->ms_huff = c->timer.stop(); }
  } else if (cnt) {
    const uint32_t per = (cnt + pieces - 1) / pieces;
    LaunchTimes keep;                                      // the dominant-kernel events of all pieces are resolved together
    for (uint32_t i = 0, k0 = 0; k0 < cnt; i++, k0 += per) {
      const uint32_t kc = std::min(per, cnt - k0);
      const bool has_last = k0 + kc == cnt;
      CJS_TRY(bwt_run(s, c->bwt, c->d_blocks + (size_t)k0 * c->cap, kc, c->cap, has_last ? n_last : c->cap, true, c->d_U + (size_t)k0 * c->cap, c->d_pidx + k0, st, false));
      if (st) { c->bwt.lt.move_into(keep); }
      CJS_HIP_TRY(hipEventRecord(c->ev_piece[i], s));
      CJS_HIP_TRY(hipStreamWaitEvent(c->tail, c->ev_piece[i], 0));
      MtfWork mv = c->mtf.view(k0, kc);
      HuffWork hv = c->huff.view(k0, kc);
      CJS_TRY(mtf_run(c->tail, mv, c->d_U + (size_t)k0 * c->cap, kc, c->rle.block_len + f + k0));
      CJS_TRY(huff_tables_run(c->tail, hv, kc, mv.b.A, mv.b.a_stride, mv.b.npos, mv.b.asz, mv.b.freq, mv.b.alist));
    }
    if (st) keep.move_into(c->bwt.lt);
    CJS_HIP_TRY(hipEventRecord(c->ev_tail, c->tail));
    CJS_HIP_TRY(hipStreamWaitEvent(s, c->ev_tail, 0));
  }
  return 0;
}
static int compress_core_impl(cjs_ctx* c, const uint8_t* d_in, size_t n, int level, long first, long count, bool framed,
                              uint8_t* d_out, size_t out_cap, uint64_t* out_bits, uint32_t* block_crcs, long crc_cap,
                              long* total_blocks, cjs_stats* st) {
  if (!c || level != c->level) return CJS_E_INVALID_ARG;
  if (n > c->max_input) return CJS_E_INVALID_ARG;
  if (((uintptr_t)d_out & 3) != 0) return CJS_E_INVALID_ARG;
  CJS_HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  c->sh_state = 0;
  EvPair evp;
  hipEvent_t &ev0 = evp.a, &ev1 = evp.b;
  const bool stage_times = st && c->stage_times;
  if (st) { memset(st, 0, sizeof *st); CJS_HIP_TRY(hipEventCreate(&ev0)); CJS_HIP_TRY(hipEventCreate(&ev1)); (void)hipEventRecord(ev0, s); }
  uint32_t nb = 0, last_len = 0;
  if (stage_times) c->timer.start();
  CJS_TRY(rle1_run(s, c->rle, d_in, n, &nb, &last_len));
  if (total_blocks) *total_blocks = (long)nb;
  if (first < 0 || first > (long)nb) return CJS_E_INVALID_ARG;
  if (count < 0 || first + count > (long)nb) count = (long)nb - first;
  const uint32_t f = (uint32_t)first, cnt = (uint32_t)count;
  CJS_TRY(blocks_through_tables(c, d_in, n, nb, last_len, f, cnt, st, stage_times));
  if (stage_times) c->timer.start();
  if (cnt && n && c->side) CJS_HIP_TRY(hipStreamWaitEvent(s, c->ev_join, 0));      // block CRCs (side stream) before the headers are packed
  const uint64_t start_bit = framed ? 32 : 0;
  // (the output size check is made on the device, by huff_offsets: no host round trip in front of the packing)
  CJS_TRY(huff_pack_run(s, c->huff, nb, f, cnt, start_bit, level, framed ? 1 : 0, framed ? 1 : 0, c->mtf.b.A, c->mtf.b.a_stride,
                        c->mtf.b.npos, c->mtf.b.asz, c->mtf.b.alist, c->rle.block_crc, c->d_pidx, (uint32_t*)d_out, out_cap & ~(size_t)3));
  CJS_HIP_TRY(hipMemcpyAsync(c->h_scalars, c->huff.scalars, 24, hipMemcpyDeviceToHost, s));
  if (block_crcs && nb) {
    if ((long)nb > crc_cap) return CJS_E_OUTPUT_TOO_SMALL;
    CJS_HIP_TRY(hipMemcpyAsync(block_crcs, c->rle.block_crc, 4 * (size_t)nb, hipMemcpyDeviceToHost, s));
  }
  CJS_HIP_TRY(hipStreamSynchronize(s));
  if (c->h_scalars[2]) return CJS_E_OUTPUT_TOO_SMALL;
  *out_bits = c->h_scalars[0];
  if (st) {
    if (stage_times) st->ms_pack = c->timer.stop();
    if (!stage_times && cnt) c->bwt.lt.resolve(st);      // the stream has drained
    (void)hipEventRecord(ev1, s); (void)hipEventSynchronize(ev1);
    float ms = 0; (void)hipEventElapsedTime(&ms, ev0, ev1);
    st->ms_total = ms;
    st->blocks = cnt; st->bytes_in = n; st->bytes_out = (*out_bits + 7) / 8;
  }
  return 0;
}

// ------------------------------------------------------------------ one process (or worker thread) per GPU: the phases of a job
// Every rank holds the stream.  Phase 1 makes the boundary tables of the rank's share of the input tiles; the caller exchanges the
// shares (an all-gather, 72 B per 4 KiB tile: the library itself never calls a collective).  Phase 2 walks the block boundaries
// (replicated: O(#blocks) and serial by nature, Q1-Q3) and takes the rank's contiguous range of blocks through the Huffman
// tables; the caller exchanges one cjs_shard_meta per rank.  Phase 3 packs the rank's blocks at their FINAL bit offset: the
// ranks' fragments are disjoint runs of whole 32-bit words of the one .bz2 stream.
extern "C" size_t cjs_bzip2_shard_share_bytes(size_t n, int world) {
  if (world < 1) return 0;
  return Rle1Work::share_bytes(Rle1Work::tiles_per_rank(n, (uint32_t)world));
}
extern "C" int cjs_bzip2_shard_tiles(cjs_ctx* c, const uint8_t* d_in, size_t n, int rank, int world, void* d_share) {
  CJS_GUARD_BEGIN
  if (!c || !d_share || world < 1 || rank < 0 || rank >= world || n > c->max_input) return CJS_E_INVALID_ARG;
  CJS_HIP_TRY(hipSetDevice(c->device));
  c->sh_state = 0;
  const uint32_t tpr = Rle1Work::tiles_per_rank(n, (uint32_t)world);
  const uint64_t t0 = (uint64_t)rank * tpr, Tn = Rle1Work::tiles_for(n);
  if (t0 < Tn) CJS_TRY(rle1_tiles(c->stream, c->rle, d_in, n, (uint32_t)t0, (uint32_t)std::min<uint64_t>(Tn, t0 + tpr), (uint8_t*)d_share, tpr));
  CJS_HIP_TRY(hipStreamSynchronize(c->stream));        // the caller's collective runs on a stream of its own
  c->sh_state = 1;
  return 0;
  CJS_GUARD_END(CJS_E_OUT_OF_MEMORY, CJS_E_HIP)
}
// bit length of blocks [0, cnt) of the range and their CRCs folded from 0 -> out[0], out[1]
__global__ void shard_meta_kernel(const uint32_t* __restrict__ bitlen, const uint32_t* __restrict__ block_crc, uint32_t first, uint32_t cnt, uint64_t* __restrict__ out) {
  if (threadIdx.x || blockIdx.x) return;
  uint64_t bits = 0; uint32_t c = 0;
  for (uint32_t k = 0; k < cnt; k++) { bits += bitlen[k]; c = ((c << 1) | (c >> 31)) ^ block_crc[first + k]; }
  out[0] = bits; out[1] = c;
}
static void shard_range(uint32_t total, int rank, int world, uint32_t& first, uint32_t& count) {
  const uint32_t share = total ? (total + (uint32_t)world - 1u) / (uint32_t)world : 0u;
  first = std::min<uint64_t>((uint64_t)rank * share, total);
  count = std::min<uint32_t>(share, total - first);
}
static int shard_blocks_impl(cjs_ctx* c, const uint8_t* d_in, size_t n, int level, int rank, int world, const void* d_shares, cjs_shard_meta* meta, cjs_stats* st) {
  if (!c || !meta || level != c->level || world < 1 || rank < 0 || rank >= world || n > c->max_input || (world > 1 && !d_shares)) return CJS_E_INVALID_ARG;
  if (c->sh_state != 1 && d_shares) return CJS_E_INVALID_ARG;          // phase order
  CJS_HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  c->sh_state = 0;
  EvPair evp;
  const bool stage_times = st && c->stage_times;
  if (st) { memset(st, 0, sizeof *st); CJS_HIP_TRY(hipEventCreate(&evp.a)); CJS_HIP_TRY(hipEventCreate(&evp.b)); (void)hipEventRecord(evp.a, s); }
  if (stage_times) c->timer.start();
  uint32_t nb = 0, last_len = 0;
  if (d_shares) CJS_TRY(rle1_tables_from_shares(s, c->rle, n, (const uint8_t*)d_shares, Rle1Work::tiles_per_rank(n, (uint32_t)world)));
  else if (n) CJS_TRY(rle1_tiles(s, c->rle, d_in, n, 0u, Rle1Work::tiles_for(n), nullptr, 0u));
  CJS_TRY(rle1_walk_run(s, c->rle, d_in, n, &nb, &last_len));
  uint32_t f = 0, cnt = 0;
  shard_range(nb, rank, world, f, cnt);
  CJS_TRY(blocks_through_tables(c, d_in, n, nb, last_len, f, cnt, st, stage_times));
  if (cnt && n && c->side) CJS_HIP_TRY(hipStreamWaitEvent(s, c->ev_join, 0));
  hipLaunchKernelGGL(shard_meta_kernel, dim3(1), dim3(1), 0, s, c->huff.b.bitlen, c->rle.block_crc, f, cnt, c->huff.scalars + 4);
  CJS_HIP_TRY(hipMemcpyAsync(c->h_scalars + 4, c->huff.scalars + 4, 16, hipMemcpyDeviceToHost, s));
  CJS_HIP_TRY(hipStreamSynchronize(s));
  meta->bits = c->h_scalars[4]; meta->crc_fold = (uint32_t)c->h_scalars[5];
  meta->total_blocks = nb; meta->first_block = f; meta->blocks = cnt;
  c->sh_nb = nb; c->sh_first = f; c->sh_cnt = cnt; c->sh_state = 2;
  if (st) {
    if (!stage_times && cnt) c->bwt.lt.resolve(st);
    (void)hipEventRecord(evp.b, s); (void)hipEventSynchronize(evp.b);
    float ms = 0; (void)hipEventElapsedTime(&ms, evp.a, evp.b);
    st->ms_total = ms; st->blocks = cnt; st->bytes_in = n; st->bytes_out = (meta->bits + 7) / 8;
  }
  return 0;
}
extern "C" int cjs_bzip2_shard_blocks(cjs_ctx* c, const uint8_t* d_in, size_t n, int level, int rank, int world, const void* d_shares,
                                      cjs_shard_meta* meta, cjs_stats* stats) {
  CJS_GUARD_BEGIN
  const int rc = shard_blocks_impl(c, d_in, n, level, rank, world, d_shares, meta, stats);
  if (rc && c) { if (c->side) (void)hipStreamSynchronize(c->side); if (c->stream) (void)hipStreamSynchronize(c->stream); }
  return rc;
  CJS_GUARD_END(CJS_E_OUT_OF_MEMORY, CJS_E_HIP)
}
// Packs the context's blocks (phase 2 left them behind) at the stream's bit `start` (the rank's first block; rank 0: 32).
// header / trailer: this rank opens / ends the stream; a rank that does not end it is followed by another rank's blocks.
static int shard_pack_core(cjs_ctx* c, int level, bool header, bool trailer, uint64_t start, uint64_t bits, uint32_t scrc, uint8_t* d_out, size_t out_cap,
                           size_t* frag_off, size_t* frag_len, uint64_t* stream_off) {
  hipStream_t s = c->stream;
  const uint64_t local_start = header ? 32 : (start & 31), end_local = local_start + bits;
  const PackShard ps{scrc, trailer ? 0 : 1};
  CJS_TRY(huff_pack_run(s, c->huff, c->sh_nb, c->sh_first, c->sh_cnt, local_start, level, header ? 1 : 0, trailer ? 1 : 0, c->mtf.b.A, c->mtf.b.a_stride,
                        c->mtf.b.npos, c->mtf.b.asz, c->mtf.b.alist, c->rle.block_crc, c->d_pidx, (uint32_t*)d_out, out_cap & ~(size_t)3, &ps));
  CJS_HIP_TRY(hipMemcpyAsync(c->h_scalars, c->huff.scalars, 24, hipMemcpyDeviceToHost, s));
  CJS_HIP_TRY(hipStreamSynchronize(s));
  if (c->h_scalars[2]) return CJS_E_OUTPUT_TOO_SMALL;
  // the fragment: whole words from the first word that starts inside this rank's bits (rank 0: the stream start) to the word its
  // last bit lands in (completed with the next rank's leading bits), or to the end of the stream
  const uint64_t word0 = header ? 0 : (start >> 5);                             // stream word at d_out[0]
  const uint64_t skip = (!header && (start & 31)) ? 4 : 0;
  const uint64_t end_bytes = trailer ? (end_local + 80 + 7) / 8 : ((end_local + 31) / 32) * 4;
  *frag_off = (size_t)skip; *frag_len = (size_t)(end_bytes > skip ? end_bytes - skip : 0); *stream_off = word0 * 4 + skip;
  return 0;
}
// global bit offset of rank `rank`'s first block, stream CRC (the ranks' folds chained: c -> rol(c, blocks) ^ fold), the rank
// that ends the stream (the last one with blocks) and the stream's bit length without the trailer
static void shard_layout(const cjs_shard_meta* metas, int world, int rank, uint64_t& start, uint64_t& total, uint32_t& scrc, int& writer) {
  start = 32; total = 32; scrc = 0; writer = 0;
  for (int r = 0; r < world; r++) {
    if (r < rank) start += metas[r].bits;
    total += metas[r].bits;
    const uint32_t rot = metas[r].blocks & 31u;
    scrc = (rot ? ((scrc << rot) | (scrc >> (32 - rot))) : scrc) ^ metas[r].crc_fold;
    if (metas[r].blocks) writer = r;
  }
}
extern "C" int cjs_bzip2_shard_pack(cjs_ctx* c, int level, int rank, int world, const cjs_shard_meta* metas, uint8_t* d_out, size_t out_cap,
                                    size_t* frag_off, size_t* frag_len, uint64_t* stream_off, uint64_t* stream_len) {
  CJS_GUARD_BEGIN
  if (!c || !metas || !frag_off || !frag_len || !stream_off || level != c->level || world < 1 || rank < 0 || rank >= world) return CJS_E_INVALID_ARG;
  if (c->sh_state != 2 || ((uintptr_t)d_out & 3) != 0) return CJS_E_INVALID_ARG;
  const cjs_shard_meta& me = metas[rank];
  if (me.blocks != c->sh_cnt || me.first_block != c->sh_first || me.total_blocks != c->sh_nb) return CJS_E_INVALID_ARG;
  for (int r = 0; r < world; r++) if (metas[r].total_blocks != me.total_blocks) return CJS_E_INVALID_ARG;    // the ranks disagree on the boundaries
  CJS_HIP_TRY(hipSetDevice(c->device));
  uint64_t start, total; uint32_t scrc; int writer;
  shard_layout(metas, world, rank, start, total, scrc, writer);
  if (stream_len) *stream_len = (total + 80 + 7) / 8;
  c->sh_state = 0;
  if (rank && !me.blocks) { *frag_off = 0; *frag_len = 0; *stream_off = (total + 80 + 7) / 8; return 0; }      // nothing of the stream lands here
  return shard_pack_core(c, level, rank == 0, rank == writer, start, me.bits, scrc, d_out, out_cap, frag_off, frag_len, stream_off);
  CJS_GUARD_END(CJS_E_OUT_OF_MEMORY, CJS_E_HIP)
}

extern "C" int cjs_bzip2_compress_device(cjs_ctx* c, const uint8_t* d_in, size_t n, int level, uint8_t* d_out, size_t out_cap,
                                         size_t* out_n, cjs_stats* stats) {
  uint64_t bits = 0;
  CJS_TRY(compress_core(c, d_in, n, level, 0, -1, true, d_out, out_cap, &bits, nullptr, 0, nullptr, stats));
  *out_n = (size_t)((bits + 7) / 8);
  return 0;
}

extern "C" int cjs_bzip2_compress_device_range(cjs_ctx* c, const uint8_t* d_in, size_t n, int level, long first_block, long count,
                                               uint8_t* d_out, size_t out_cap, uint64_t* out_bits, uint32_t* block_crcs, long crc_cap,
                                               long* total_blocks, cjs_stats* stats) {
  return compress_core(c, d_in, n, level, first_block, count, false, d_out, out_cap, out_bits, block_crcs, crc_cap, total_blocks, stats);
}

// One shard of a multi-GPU job: its own device, context and stream.  A shard is a run of consecutive blocks; because the
// RLE1 state is fresh at every block start, the input bytes [start(first), start(first + count)) form a
// stream of their own whose blocks are exactly those blocks, so a shard uploads and processes ONLY its byte range.
// All shards at once (one per GPU): the shards meet once -- every shard publishes (bit length, CRC fold) of its blocks -- and
// then pack at their FINAL bit offset; their fragments are disjoint runs of whole words of the stream and go from the device
// straight to their place in the result buffer (no merge pass).  In waves (more ranges than may run at a time: very large
// inputs): a shard packs from bit 0 and keeps its bytes, the host shifts them into place at the end.
struct MultiSync {                       // the one meeting of the shards of a call
  std::mutex mu;
  std::condition_variable cv;
  uint32_t published = 0, nshards = 0;
  int rc = 0;                            // first failure of any shard: everyone stops
  std::vector<cjs_shard_meta> metas;
  uint8_t* out = nullptr;                // result buffer, allocated by the coordinating thread once the length is known
  bool out_ready = false;
  void publish(uint32_t i, const cjs_shard_meta& m, int shard_rc) {
    std::lock_guard<std::mutex> lock(mu);
    metas[i] = m;
    if (shard_rc && !rc) rc = shard_rc;
    published++;
    cv.notify_all();
  }
};
struct Shard {
  int device = 0, slot = 0, rc = 0;
  uint32_t index = 0;
  long first = 0, count = 0;
  uint64_t byte_lo = 0, byte_hi = 0;
  const uint8_t* d_resident = nullptr;   // the range is already in this device's memory (the boundary pass put it there)
  uint64_t bits = 0;
  std::vector<uint8_t> bytes;            // wave mode: the shard's bit string from bit 0
  uint32_t crc_fold = 0;
};
static void run_shard_body(Shard* sh, const uint8_t* in, int level, MultiSync* sync, bool& published) {
  if (hipSetDevice(sh->device) != hipSuccess) { sh->rc = CJS_E_HIP; return; }
  const size_t n = (size_t)(sh->byte_hi - sh->byte_lo);
  const size_t per = (size_t)sh->count * ((size_t)level * 100000);
  const size_t out_cap = (per + per / 4 + 65536 + 3) & ~(size_t)3;
  static const bool no_cache = getenv("CJS_NO_CTX_CACHE") != nullptr;
  HostCache local;                                                       // shards beyond the cached slots of a device: a context of their own
  HostCache& hc = sh->slot < BOUNDARY_SLOT ? g_host_cache[sh->device][sh->slot] : local;
  std::lock_guard<std::mutex> lock(hc.mu);
  struct Cleanup { HostCache& h; bool drop; ~Cleanup() { if (drop) h.release(); } } cleanup{hc, &hc == &local || no_cache};
  sh->rc = hc.ensure(n, level, 0, sh->d_resident ? 0 : (n ? n : 4), out_cap);
  if (sh->rc) { cleanup.drop = true; return; }
  cjs_ctx* c = hc.ctx;
  const uint8_t* d_in = sh->d_resident ? sh->d_resident : hc.d_in;
  if (!sh->d_resident && n && hipMemcpyAsync(hc.d_in, in + sh->byte_lo, n, hipMemcpyHostToDevice, c->stream) != hipSuccess) { sh->rc = CJS_E_HIP; cleanup.drop = true; return; }
  static const bool dbg = getenv("CJS_DEBUG") != nullptr;
  if (dbg) fprintf(stderr, "[cjs] shard %u on device %d (slot %d): blocks [%ld, %ld), bytes [%llu, %llu): H2D %zu B%s\n", sh->index, sh->device, sh->slot, sh->first, sh->first + sh->count,
                   (unsigned long long)sh->byte_lo, (unsigned long long)sh->byte_hi, sh->d_resident ? (size_t)0 : n, sh->d_resident ? " (resident from the boundary pass)" : "");
  if (!sync) {                                                           // wave mode: bare bit string from bit 0
    long total = 0;
    std::vector<uint32_t> crcs((size_t)sh->count + 1, 0u);
    sh->rc = cjs_bzip2_compress_device_range(c, d_in, n, level, 0, -1, hc.d_out, out_cap, &sh->bits, crcs.data(), (long)crcs.size(), &total, nullptr);
    if (!sh->rc && total != sh->count) sh->rc = CJS_E_HIP;          // cannot happen: the range was cut at block starts
    if (!sh->rc) {
      for (long k = 0; k < sh->count; k++) sh->crc_fold = ((sh->crc_fold << 1) | (sh->crc_fold >> 31)) ^ crcs[(size_t)k];
      sh->bytes.resize((size_t)((sh->bits + 7) / 8) + 16);
      if (hipMemcpy(sh->bytes.data(), hc.d_out, sh->bytes.size(), hipMemcpyDeviceToHost) != hipSuccess) sh->rc = CJS_E_HIP;
    }
    if (sh->rc) cleanup.drop = true;
    return;
  }
  cjs_shard_meta meta{};
  sh->rc = shard_blocks_impl(c, d_in, n, level, 0, 1, nullptr, &meta, nullptr);      // the byte range is a stream of its own
  if (!sh->rc && (long)meta.total_blocks != sh->count) sh->rc = CJS_E_HIP;           // cannot happen: the range was cut at block starts
  sync->publish(sh->index, meta, sh->rc);
  published = true;
  if (sh->rc) { (void)hipStreamSynchronize(c->stream); if (c->side) (void)hipStreamSynchronize(c->side); cleanup.drop = true; return; }
  {
    std::unique_lock<std::mutex> lk(sync->mu);
    sync->cv.wait(lk, [&] { return sync->out_ready || sync->rc; });
    if (sync->rc) { c->sh_state = 0; return; }
  }
  uint64_t start, total; uint32_t scrc; int writer;
  shard_layout(sync->metas.data(), (int)sync->nshards, (int)sh->index, start, total, scrc, writer);
  if (!meta.blocks && sh->index) { c->sh_state = 0; return; }
  size_t fo = 0, fl = 0; uint64_t so = 0;
  sh->rc = shard_pack_core(c, level, sh->index == 0, (int)sh->index == writer, start, meta.bits, scrc, hc.d_out, out_cap, &fo, &fl, &so);
  c->sh_state = 0;
  if (!sh->rc && fl && hipMemcpy(sync->out + so, hc.d_out + fo, fl, hipMemcpyDeviceToHost) != hipSuccess) sh->rc = CJS_E_HIP;
  if (sh->rc) cleanup.drop = true;
}
static void run_shard(Shard* sh, const uint8_t* in, int level, MultiSync* sync) {
  bool published = false;
  if (sh->count == 0) { if (sync) sync->publish(sh->index, cjs_shard_meta{}, 0); return; }      // no blocks: nothing of the stream comes from here
  try { run_shard_body(sh, in, level, sync, published); }              // nothing may leave a worker thread (std::terminate)
  catch (const std::bad_alloc&) { sh->rc = CJS_E_OUT_OF_MEMORY; }
  catch (...) { sh->rc = CJS_E_HIP; }
  if (sync && !published) sync->publish(sh->index, cjs_shard_meta{}, sh->rc ? sh->rc : CJS_E_HIP);
  if (sync && sh->rc) { std::lock_guard<std::mutex> lock(sync->mu); if (!sync->rc) sync->rc = sh->rc; sync->cv.notify_all(); }
}
struct JoinAll {                         // worker threads are joined on every path out of the scope that started them
  std::vector<std::thread> th;
  ~JoinAll() { for (auto& t : th) if (t.joinable()) t.join(); }
};
// dst bits [pos, pos + nbits) |= the first nbits bits of src (MSB first); src is readable 9 bytes past its last bit.  Bytes
// that lie wholly inside the range are STORED (8 at a time, one 64-bit funnel shift), the partial bytes at the two ends OR-ed
// (the neighbours' bits live there): ranges of different shards may be merged by different threads when `edges` tells them
// apart -- 0: interior only (parallel part), 1: the two ends only (serial part).
static void funnel_merge(uint8_t* dst, uint64_t pos, const uint8_t* src, uint64_t nbits, int edges) {
  auto src_bits = [&](uint64_t off, unsigned k) -> uint32_t {             // k <= 8 bits of src from bit `off`
    uint32_t v = 0;
    for (unsigned i = 0; i < k; i++) { const uint64_t b = off + i; v = (v << 1) | ((src[b >> 3] >> (7 - (b & 7))) & 1u); }
    return v;
  };
  const uint64_t end = pos + nbits;
  const uint64_t j0 = (pos + 7) >> 3, j1 = end >> 3;                      // whole bytes of dst inside the range: [j0, j1)
  if (edges) {
    if (j0 > j1) { const unsigned k = (unsigned)nbits; dst[pos >> 3] |= (uint8_t)(src_bits(0, k) << (8 - (pos & 7) - k)); return; }   // inside one byte
    if (pos & 7) { const unsigned k = 8 - (unsigned)(pos & 7); dst[pos >> 3] |= (uint8_t)src_bits(0, k); }
    if (end & 7) { const unsigned k = (unsigned)(end & 7); dst[end >> 3] |= (uint8_t)(src_bits(nbits - k, k) << (8 - k)); }
    return;
  }
  if (j0 >= j1) return;
  const uint64_t o = 8 * j0 - pos;                                        // src bit of dst byte j0 (< 8)
  const unsigned r = (unsigned)(o & 7);
  const uint8_t* q = src + (o >> 3);
  uint64_t j = j0;
  for (; j + 8 <= j1; j += 8, q += 8) {
    uint64_t hi; memcpy(&hi, q, 8); hi = __builtin_bswap64(hi);
    const uint64_t v = r ? (hi << r) | ((uint64_t)q[8] >> (8 - r)) : hi;
    const uint64_t be = __builtin_bswap64(v);
    memcpy(dst + j, &be, 8);
  }
  for (; j < j1; j++, q++) dst[j] = r ? (uint8_t)((q[0] << r) | (q[1] >> (8 - r))) : q[0];
}

// Multi-GPU host path: ONE boundary pass over the stream (device 0: the input start of every block),
// then blocks are dealt in contiguous ranges to per-GPU worker threads, each of which gets only its byte range; the only
// cross-shard data are (bit length, CRC fold).  Contexts and staging buffers are kept per device between calls.
// max_parallel = shards in flight at a time (0 = all): one at a time bounds the workspace when the ranges are only there to
// cut a very large input into pieces (each piece's workspace is ~70 B per byte of its blocks)
static int compress_multi(const uint8_t* in, size_t n, int level, uint32_t nshards, uint8_t** out, size_t* out_n, uint32_t max_parallel = 0) {
  int ndev = 0, dev0 = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || hipGetDevice(&dev0) != hipSuccess) return CJS_E_NO_DEVICE;
  if (ndev > MAX_CACHED_DEVICES) ndev = MAX_CACHED_DEVICES;
  struct RestoreDevice { int d; ~RestoreDevice() { (void)hipSetDevice(d); } } restore{dev0};
  static const bool no_cache = getenv("CJS_NO_CTX_CACHE") != nullptr;
  // ---- boundary pass: block starts of the whole stream (device 0; its copy of the input serves the shards that run there)
  HostCache& bc = g_host_cache[0][BOUNDARY_SLOT];
  std::lock_guard<std::mutex> block(bc.mu);
  struct DropBoundary { HostCache& h; bool drop; ~DropBoundary() { if (drop) { (void)hipSetDevice(0); h.release(); } } } dropb{bc, no_cache};
  std::vector<uint64_t> starts;
  {
    CJS_HIP_TRY(hipSetDevice(0));
    int rc = bc.ensure(n, level, 1, n ? n : 4, 0);
    uint32_t nbk = 0;
    if (!rc && hipMemcpyAsync(bc.d_in, in, n, hipMemcpyHostToDevice, bc.ctx->stream) != hipSuccess) rc = CJS_E_HIP;
    if (!rc) rc = rle1_run(bc.ctx->stream, bc.ctx->rle, bc.d_in, n, &nbk);
    std::vector<RleBlock> hb(nbk);
    if (!rc && nbk && hipMemcpy(hb.data(), bc.ctx->rle.blocks, sizeof(RleBlock) * nbk, hipMemcpyDeviceToHost) != hipSuccess) rc = CJS_E_HIP;
    if (rc) { dropb.drop = true; return rc; }
    starts.resize((size_t)nbk + 1);
    for (uint32_t k = 0; k < nbk; k++) starts[k] = hb[k].s;
    starts[nbk] = n;
  }
  const long total = (long)starts.size() - 1;
  const long share = total ? (total + nshards - 1) / nshards : 0;
  const bool waves = max_parallel && max_parallel < nshards;
  std::vector<Shard> sh(nshards);
  for (uint32_t i = 0; i < nshards; i++) {
    sh[i].index = i;
    sh[i].device = (int)(i % (uint32_t)ndev);
    sh[i].slot = waves ? 0 : (int)(i / (uint32_t)ndev);          // shards that share a device at the same time need contexts of their own
    sh[i].first = std::min<long>((long)i * share, total);
    sh[i].count = std::min<long>(share, total - sh[i].first);
    sh[i].byte_lo = starts[(size_t)sh[i].first]; sh[i].byte_hi = starts[(size_t)(sh[i].first + sh[i].count)];
    if (sh[i].device == 0) sh[i].d_resident = bc.d_in + sh[i].byte_lo;
  }
  if (!waves) {
    MultiSync sync;
    sync.nshards = nshards; sync.metas.assign(nshards, cjs_shard_meta{});
    uint8_t* result = nullptr; size_t len = 0;
    {
      JoinAll workers;
      for (uint32_t i = 0; i < nshards; i++) workers.th.emplace_back(run_shard, &sh[i], in, level, &sync);
      std::unique_lock<std::mutex> lk(sync.mu);
      sync.cv.wait(lk, [&] { return sync.published == nshards; });
      if (!sync.rc) {
        uint64_t start, tbits; uint32_t scrc; int writer;
        shard_layout(sync.metas.data(), (int)nshards, 0, start, tbits, scrc, writer);
        len = (size_t)((tbits + 80 + 7) / 8);
        result = (uint8_t*)HostPool::take(len);
        if (!result) sync.rc = CJS_E_OUT_OF_MEMORY;
        sync.out = result; sync.out_ready = true;
      }
      sync.cv.notify_all();
    }                                                           // (joined)
    int rc = sync.rc;
    for (auto& x : sh) if (x.rc && !rc) rc = x.rc;
    if (rc) { HostPool::give(result); return rc; }
    *out = result; *out_n = len;
    return 0;
  }
  for (uint32_t i0 = 0; i0 < nshards; i0 += max_parallel) {
    JoinAll workers;
    for (uint32_t i = i0; i < nshards && i < i0 + max_parallel; i++) workers.th.emplace_back(run_shard, &sh[i], in, level, (MultiSync*)nullptr);
  }
  uint64_t total_bits = 32 + 80;
  for (auto& x : sh) { if (x.rc) return x.rc; total_bits += x.bits; }
  const size_t len = (size_t)((total_bits + 7) / 8);
  uint8_t* o = (uint8_t*)calloc(len + 16, 1);
  if (!o) return CJS_E_OUT_OF_MEMORY;
  o[0] = 'B'; o[1] = 'Z'; o[2] = 'h'; o[3] = (uint8_t)('0' + level);
  std::vector<uint64_t> at(nshards);
  uint64_t pos = 32; uint32_t scrc = 0;
  for (uint32_t i = 0; i < nshards; i++) {
    at[i] = pos; pos += sh[i].bits;
    const uint32_t rot = (uint32_t)sh[i].count & 31u;
    scrc = (rot ? ((scrc << rot) | (scrc >> (32 - rot))) : scrc) ^ sh[i].crc_fold;
  }
  {
    // interiors by a few threads (disjoint whole bytes), then the shared end bytes one shard after the other
    JoinAll mergers;
    const uint32_t nt = std::min<uint32_t>(nshards, 8u);
    for (uint32_t t = 0; t < nt; t++) mergers.th.emplace_back([&, t] { for (uint32_t i = t; i < nshards; i += nt) if (sh[i].bits) funnel_merge(o, at[i], sh[i].bytes.data(), sh[i].bits, 0); });
  }
  for (uint32_t i = 0; i < nshards; i++) if (sh[i].bits) funnel_merge(o, at[i], sh[i].bytes.data(), sh[i].bits, 1);
  const uint64_t trailer[2] = {0x177245385090ull, scrc}; const int tb[2] = {48, 32};
  for (int q = 0; q < 2; q++) for (int i = tb[q] - 1; i >= 0; i--, pos++) if ((trailer[q] >> i) & 1) o[pos >> 3] |= (uint8_t)(0x80 >> (pos & 7));
  *out = o; *out_n = len;
  return 0;
}

extern "C" int cjs_bzip2_compress(const uint8_t* in, size_t n, int level, uint8_t** out, size_t* out_n, const cjs_opts* opts) {
  if (!out || !out_n) return CJS_E_INVALID_ARG;
  *out = nullptr; *out_n = 0;
  clear_detail();
  if (level < 1 || level > 9) return CJS_E_BAD_LEVEL;                 // J/Bzip2_joined_.js:2208
  CJS_GUARD_BEGIN
  CJS_TRY(select_device(opts));
  uint32_t nshards = (opts && opts->struct_size >= sizeof(cjs_opts)) ? opts->n_devices : 0;
  if (const char* e = getenv("CJS_DEVICES")) nshards = (uint32_t)atoi(e);   // lets JS / Python callers shard without an opts struct
  {
    // several GPUs and / or a very large input: contiguous block ranges.  With more ranges than devices (inputs above
    // CJS_CHUNK_BYTES, default 2 GiB, are cut so that a range's workspace stays bounded) the ranges run in waves of one per device.
    static const size_t chunk = getenv("CJS_CHUNK_BYTES") ? (size_t)strtoull(getenv("CJS_CHUNK_BYTES"), nullptr, 10) : ((size_t)2 << 30);
    if (nshards > 64) nshards = 64;
    const size_t pieces = (chunk && n > chunk) ? (n + chunk / 2 - 1) / (chunk / 2 ? chunk / 2 : 1) : 0;
    if (n > 0 && (nshards > 1 || pieces > 1)) {
      const uint32_t par = nshards > 1 ? nshards : 1;
      const uint32_t ranges = (uint32_t)std::max<size_t>(par, std::min<size_t>(pieces, 4096));
      return compress_multi(in, n, level, ranges, out, out_n, ranges > par ? par : 0);
    }
  }
  // The workspace (~70 B per input byte), the staging buffers and the streams are kept per device between calls
  // (creating and freeing them costs more than compressing 100 MB); cjs_trim() or CJS_NO_CTX_CACHE=1 gives them back.
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_CACHED_DEVICES) return CJS_E_HIP;
  static const bool no_cache = getenv("CJS_NO_CTX_CACHE") != nullptr;
  HostCache& hc = g_host_cache[dev][0];
  std::lock_guard<std::mutex> lock(hc.mu);
  const size_t out_cap = (n + n / 4 + 4096 + 3) & ~(size_t)3;
  static const bool dbg = getenv("CJS_DEBUG") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto t0 = now();
  int rc = hc.ensure(n, level, 0, n ? n : 4, out_cap);
  if (rc) { hc.release(); return rc; }
  cjs_ctx* c = hc.ctx;
  const auto t1 = now();
  if (!rc && n && hipMemcpyAsync(hc.d_in, in, n, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = CJS_E_HIP;
  if (dbg && !rc) (void)hipStreamSynchronize(c->stream);
  const auto t2 = now();
  size_t len = 0;
  cjs_stats* st = (opts && opts->struct_size >= sizeof(cjs_opts)) ? opts->stats : nullptr;
  c->stage_times = !(opts && opts->struct_size >= sizeof(cjs_opts) && (opts->flags & CJS_FLAG_NO_STAGE_TIMES));
  if (!rc) rc = cjs_bzip2_compress_device(c, hc.d_in, n, level, hc.d_out, out_cap, &len, st);
  const auto t3 = now();
  uint8_t* host = nullptr;
  if (!rc) { host = (uint8_t*)HostPool::take(len ? len : 1); if (!host) rc = CJS_E_OUT_OF_MEMORY; }
  if (!rc && hipMemcpy(host, hc.d_out, len, hipMemcpyDeviceToHost) != hipSuccess) rc = CJS_E_HIP;
  const auto t4 = now();
  if (dbg) fprintf(stderr, "[cjs] host compress: workspace %.2f ms, H2D %.2f ms, pipeline %.2f ms, malloc + D2H %.2f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4));
  if (rc || no_cache) hc.release();           // after an error the cached state is not trusted
  if (rc) { HostPool::give(host); return rc; }
  *out = host; *out_n = len;
  return 0;
  CJS_GUARD_END(CJS_E_OUT_OF_MEMORY, CJS_E_HIP)
}

extern "C" void cjs_trim(void) {
  DevPool::trim();
  HostPool::trim();
  int cur = 0;
  const bool have = hipGetDevice(&cur) == hipSuccess;
  for (int d = 0; d < MAX_CACHED_DEVICES; d++)
    for (int k = 0; k < CACHE_SLOTS; k++) {
      HostCache& hc = g_host_cache[d][k];
      std::lock_guard<std::mutex> lock(hc.mu);
      if (!hc.ctx && !hc.d_in && !hc.d_out) continue;
      if (hipSetDevice(d) == hipSuccess) hc.release();
    }
  if (have) (void)hipSetDevice(cur);
}

// ------------------------------------------------------------------ stage-level entry points (tests)
extern "C" int cjs_stage_mtf(const uint8_t* U, const uint8_t* blocks, size_t n, int block_len, uint16_t* A, uint32_t* npos,
                             uint32_t* freq, uint32_t* alphabet, const cjs_opts* opts) {
  CJS_GUARD_BEGIN
  (void)blocks;   // the used-symbol set of a block equals that of its BWT (a permutation of it)
  CJS_TRY(select_device(opts));
  if (n == 0) return 0;
  if (block_len <= 0) return CJS_E_INVALID_ARG;
  const uint32_t stride = (uint32_t)block_len, nb = (uint32_t)((n + stride - 1) / stride);
  Arena arena;
  CJS_TRY(arena.init(MtfWork::bytes_needed(nb, stride) + (size_t)nb * stride + 4 * (size_t)nb + 65536));
  MtfWork w;
  int rc = w.carve(arena, nb, stride);
  uint8_t* d_U = arena.take<uint8_t>((size_t)nb * stride);
  uint32_t* d_len = arena.take<uint32_t>(nb);
  if (!rc && (!d_U || !d_len)) rc = CJS_E_OUT_OF_MEMORY;
  std::vector<uint32_t> lens(nb, stride);
  lens[nb - 1] = (uint32_t)(n - (size_t)(nb - 1) * stride);
  hipStream_t s = nullptr;
  if (!rc && hipStreamCreate(&s) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipMemcpy(d_U, U, n, hipMemcpyHostToDevice) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipMemcpy(d_len, lens.data(), 4 * (size_t)nb, hipMemcpyHostToDevice) != hipSuccess) rc = CJS_E_HIP;
  if (!rc) rc = mtf_run(s, w, d_U, nb, d_len);
  if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = CJS_E_HIP;
  if (!rc) {
    std::vector<uint32_t> hnpos(nb);
    if (hipMemcpy(hnpos.data(), w.b.npos, 4 * (size_t)nb, hipMemcpyDeviceToHost) != hipSuccess) rc = CJS_E_HIP;
    if (!rc && hipMemcpy(freq, w.b.freq, 4 * 258 * (size_t)nb, hipMemcpyDeviceToHost) != hipSuccess) rc = CJS_E_HIP;
    if (!rc && hipMemcpy(alphabet, w.b.asz, 4 * (size_t)nb, hipMemcpyDeviceToHost) != hipSuccess) rc = CJS_E_HIP;
    for (uint32_t k = 0; k < nb && !rc; k++) {
      npos[k] = hnpos[k];
      if (hipMemcpy(A + (size_t)k * (stride + 1), w.b.A + (size_t)k * w.b.a_stride, 2 * (size_t)hnpos[k], hipMemcpyDeviceToHost) != hipSuccess) rc = CJS_E_HIP;
    }
  }
  if (s) (void)hipStreamDestroy(s);
  arena.destroy();
  return rc;
  CJS_GUARD_END(CJS_E_OUT_OF_MEMORY, CJS_E_HIP)
}

extern "C" int cjs_stage_huff(const uint16_t* A, uint32_t npos, uint32_t alphabet, uint8_t* selectors, uint8_t* lengths,
                              uint32_t* ngroups, const cjs_opts* opts) {
  CJS_GUARD_BEGIN
  CJS_TRY(select_device(opts));
  if (npos == 0 || alphabet == 0 || alphabet > 256) return CJS_E_INVALID_ARG;
  const uint32_t stride = npos;    // any stride >= npos-1 works for the selector buffers
  Arena arena;
  CJS_TRY(arena.init(HuffWork::bytes_needed(1, stride) + 2 * (size_t)npos + 4096 * 4 + 65536));
  HuffWork w;
  int rc = w.carve(arena, 1, stride);
  uint16_t* d_A = arena.take<uint16_t>(npos);
  uint32_t* d_misc = arena.take<uint32_t>(2 + 258);
  uint8_t* d_alist = arena.take<uint8_t>(256);
  if (!rc && (!d_A || !d_misc || !d_alist)) rc = CJS_E_OUT_OF_MEMORY;
  std::vector<uint32_t> misc(2 + 258, 0);
  misc[0] = npos; misc[1] = alphabet;
  for (uint32_t i = 0; i < npos; i++) { if (A[i] > alphabet + 1) { arena.destroy(); return CJS_E_INVALID_ARG; } misc[2 + A[i]]++; }
  uint8_t al[256]; for (int i = 0; i < 256; i++) al[i] = (uint8_t)i;
  hipStream_t s = nullptr;
  if (!rc && hipStreamCreate(&s) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipMemcpy(d_A, A, 2 * (size_t)npos, hipMemcpyHostToDevice) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipMemcpy(d_misc, misc.data(), 4 * misc.size(), hipMemcpyHostToDevice) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipMemcpy(d_alist, al, 256, hipMemcpyHostToDevice) != hipSuccess) rc = CJS_E_HIP;
  if (!rc) rc = huff_tables_run(s, w, 1, d_A, npos, d_misc, d_misc + 1, d_misc + 2, d_alist);
  if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = CJS_E_HIP;
  const uint32_t nsel = (npos + 49) / 50;
  if (!rc && hipMemcpy(selectors, w.b.sel, nsel, hipMemcpyDeviceToHost) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipMemcpy(lengths, w.b.lens, 6 * 258, hipMemcpyDeviceToHost) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipMemcpy(ngroups, w.b.ngroups, 4, hipMemcpyDeviceToHost) != hipSuccess) rc = CJS_E_HIP;
  if (s) (void)hipStreamDestroy(s);
  arena.destroy();
  return rc;
  CJS_GUARD_END(CJS_E_OUT_OF_MEMORY, CJS_E_HIP)
}
