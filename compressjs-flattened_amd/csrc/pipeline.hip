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
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
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
  if (!rc && !getenv("CJS_NO_SIDE_STREAM") && hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipHostMalloc((void**)&c->h_scalars, 256) != hipSuccess) rc = CJS_E_HIP;
  if (!rc) rc = c->timer.init(c->stream);
  if (rc) { cjs_ctx_destroy(c); return rc; }
  *out = c;
  return 0;
  CJS_GUARD_END(CJS_E_OUT_OF_MEMORY, CJS_E_HIP)
}

extern "C" void cjs_ctx_destroy(cjs_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  c->timer.destroy();
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
constexpr int MAX_CACHED_DEVICES = 64;
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
};
static HostCache g_host_cache[MAX_CACHED_DEVICES];

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
static int compress_core_impl(cjs_ctx* c, const uint8_t* d_in, size_t n, int level, long first, long count, bool framed,
                              uint8_t* d_out, size_t out_cap, uint64_t* out_bits, uint32_t* block_crcs, long crc_cap,
                              long* total_blocks, cjs_stats* st) {
  if (!c || level != c->level) return CJS_E_INVALID_ARG;
  if (n > c->max_input) return CJS_E_INVALID_ARG;
  if (((uintptr_t)d_out & 3) != 0) return CJS_E_INVALID_ARG;
  CJS_HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  struct EvPair {                                        // whole-call timing events; released on every return path
    hipEvent_t a = nullptr, b = nullptr;
    ~EvPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  } evp;
  hipEvent_t &ev0 = evp.a, &ev1 = evp.b;
  const bool stage_times = st && !(st->flags & CJS_STATS_NO_STAGE_TIMES);
  if (st) { memset(st, 0, sizeof *st); CJS_HIP_TRY(hipEventCreate(&ev0)); CJS_HIP_TRY(hipEventCreate(&ev1)); (void)hipEventRecord(ev0, s); }
  uint32_t nb = 0, last_len = 0;
  if (stage_times) c->timer.start();
  CJS_TRY(rle1_run(s, c->rle, d_in, n, &nb, &last_len));
  if (total_blocks) *total_blocks = (long)nb;
  if (first < 0 || first > (long)nb) return CJS_E_INVALID_ARG;
  if (count < 0 || first + count > (long)nb) count = (long)nb - first;
  const uint32_t f = (uint32_t)first, cnt = (uint32_t)count;
  if (cnt > c->range_blocks) return CJS_E_INVALID_ARG;
  uint32_t n_last = c->cap;
  if (cnt && f + cnt == nb) n_last = last_len;        // (came to the host with the block count)
  // all per-block buffers below are indexed relative to `first`; only rle.block_len / block_crc are absolute
  CJS_TRY(rle1_finish(s, c->rle, d_in, n, f, cnt, c->d_blocks, c->side, c->ev_fork, c->ev_join));
  if (stage_times) { CJS_HIP_TRY(hipStreamSynchronize(s)); st->ms_rle1 = c->timer.stop(); }
  if (cnt) {
    if (stage_times) c->timer.start();
    CJS_TRY(bwt_run(s, c->bwt, c->d_blocks, cnt, c->cap, n_last, true, c->d_U, c->d_pidx, st, stage_times));
    if (stage_times) { st->ms_bwt = c->timer.stop(); c->timer.start(); }
    CJS_TRY(mtf_run(s, c->mtf, c->d_U, cnt, c->rle.block_len + f));
    if (stage_times) { CJS_HIP_TRY(hipStreamSynchronize(s)); st->ms_mtf = c->timer.stop(); c->timer.start(); }
    CJS_TRY(huff_tables_run(s, c->huff, cnt, c->mtf.b.A, c->mtf.b.a_stride, c->mtf.b.npos, c->mtf.b.asz, c->mtf.b.freq, c->mtf.b.alist));
    if (stage_times) { CJS_HIP_TRY(hipStreamSynchronize(s)); st->ms_huff = c->timer.stop(); }
  }
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
// stream of their own whose blocks are exactly those blocks, so a shard uploads and processes ONLY its byte range and
// brings its bare bit string back to the host.
struct Shard {
  int device = 0, rc = 0;
  long first = 0, count = 0;
  uint64_t byte_lo = 0, byte_hi = 0;
  const uint8_t* d_resident = nullptr;   // the range is already in this device's memory (the boundary pass put it there)
  uint64_t bits = 0;
  std::vector<uint8_t> bytes;
  std::vector<uint32_t> crcs;            // CRCs of the shard's blocks
};
static void run_shard(Shard* sh, const uint8_t* in, int level) {
  if (sh->count == 0) return;
  cjs_ctx* c = nullptr;
  if (hipSetDevice(sh->device) != hipSuccess) { sh->rc = CJS_E_HIP; return; }
  const size_t n = (size_t)(sh->byte_hi - sh->byte_lo);
  sh->rc = cjs_ctx_create(&c, sh->device, n, level);
  if (sh->rc) return;
  const size_t per = (size_t)sh->count * ((size_t)level * 100000);
  const size_t out_cap = (per + per / 4 + 65536 + 3) & ~(size_t)3;
  uint8_t *d_in = nullptr, *d_out = nullptr;
  if (!sh->d_resident && hipMalloc((void**)&d_in, n ? n : 4) != hipSuccess) sh->rc = CJS_E_OUT_OF_MEMORY;
  if (!sh->rc && hipMalloc((void**)&d_out, out_cap) != hipSuccess) sh->rc = CJS_E_OUT_OF_MEMORY;
  if (!sh->rc && d_in && n && hipMemcpyAsync(d_in, in + sh->byte_lo, n, hipMemcpyHostToDevice, c->stream) != hipSuccess) sh->rc = CJS_E_HIP;
  static const bool dbg = getenv("CJS_DEBUG") != nullptr;
  if (dbg) fprintf(stderr, "[cjs] shard on device %d: blocks [%ld, %ld), bytes [%llu, %llu): H2D %zu B%s\n", sh->device, sh->first, sh->first + sh->count,
                   (unsigned long long)sh->byte_lo, (unsigned long long)sh->byte_hi, sh->d_resident ? (size_t)0 : n, sh->d_resident ? " (resident from the boundary pass)" : "");
  sh->crcs.assign((size_t)sh->count + 1, 0u);
  long total = 0;
  if (!sh->rc) sh->rc = cjs_bzip2_compress_device_range(c, sh->d_resident ? sh->d_resident : d_in, n, level, 0, -1, d_out, out_cap, &sh->bits, sh->crcs.data(),
                                                        (long)sh->crcs.size(), &total, nullptr);
  if (!sh->rc && total != sh->count) sh->rc = CJS_E_HIP;          // cannot happen: the range was cut at block starts
  if (!sh->rc) {
    sh->bytes.resize((size_t)((sh->bits + 7) / 8) + 8);
    if (hipMemcpy(sh->bytes.data(), d_out, sh->bytes.size(), hipMemcpyDeviceToHost) != hipSuccess) sh->rc = CJS_E_HIP;
  }
  if (d_in) (void)hipFree(d_in);
  if (d_out) (void)hipFree(d_out);
  cjs_ctx_destroy(c);
}

// Multi-GPU host path: ONE boundary pass over the stream (device 0: the input start of every block),
// then blocks are dealt in contiguous ranges to per-GPU worker threads, each of which gets only its byte range; the only
// cross-shard data are (bit length, block CRCs).  The host funnel-shifts the bit strings into one stream.
// max_parallel = shards in flight at a time (0 = all): one at a time bounds the workspace when the ranges are only there to
// cut a very large input into pieces (each piece's workspace is ~70 B per byte of its blocks)
static int compress_multi(const uint8_t* in, size_t n, int level, uint32_t nshards, uint8_t** out, size_t* out_n, uint32_t max_parallel = 0) {
  int ndev = 0, dev0 = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || hipGetDevice(&dev0) != hipSuccess) return CJS_E_NO_DEVICE;
  // ---- boundary pass: block starts of the whole stream
  std::vector<uint64_t> starts;
  uint8_t* d_all = nullptr;
  {
    cjs_ctx* c = nullptr;
    CJS_TRY(cjs_ctx_create_sharded(&c, 0, n, 1, level));
    int rc = 0; uint32_t nbk = 0;
    if (hipMalloc((void**)&d_all, n ? n : 4) != hipSuccess) rc = CJS_E_OUT_OF_MEMORY;
    if (!rc && hipMemcpyAsync(d_all, in, n, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = CJS_E_HIP;
    if (!rc) rc = rle1_run(c->stream, c->rle, d_all, n, &nbk);
    std::vector<RleBlock> hb(nbk);
    if (!rc && nbk && hipMemcpy(hb.data(), c->rle.blocks, sizeof(RleBlock) * nbk, hipMemcpyDeviceToHost) != hipSuccess) rc = CJS_E_HIP;
    cjs_ctx_destroy(c);
    if (rc) { if (d_all) (void)hipFree(d_all); (void)hipSetDevice(dev0); return rc; }
    starts.resize((size_t)nbk + 1);
    for (uint32_t k = 0; k < nbk; k++) starts[k] = hb[k].s;
    starts[nbk] = n;
  }
  const long total = (long)starts.size() - 1;
  const long share = total ? (total + nshards - 1) / nshards : 0;
  std::vector<Shard> sh(nshards);
  std::vector<std::thread> th;
  for (uint32_t i = 0; i < nshards; i++) {
    sh[i].device = (int)(i % (uint32_t)ndev);
    sh[i].first = std::min<long>((long)i * share, total);
    sh[i].count = std::min<long>(share, total - sh[i].first);
    sh[i].byte_lo = starts[(size_t)sh[i].first]; sh[i].byte_hi = starts[(size_t)(sh[i].first + sh[i].count)];
    if (sh[i].device == 0) sh[i].d_resident = d_all + sh[i].byte_lo;
  }
  const uint32_t par = max_parallel ? max_parallel : nshards;
  for (uint32_t i0 = 0; i0 < nshards; i0 += par) {
    th.clear();
    for (uint32_t i = i0; i < nshards && i < i0 + par; i++) th.emplace_back(run_shard, &sh[i], in, level);
    for (auto& t : th) t.join();
  }
  (void)hipSetDevice(0);
  (void)hipFree(d_all);
  (void)hipSetDevice(dev0);
  uint64_t total_bits = 32 + 80;
  for (auto& x : sh) { if (x.rc) return x.rc; total_bits += x.bits; }
  const size_t len = (size_t)((total_bits + 7) / 8);
  uint8_t* o = (uint8_t*)calloc(len + 16, 1);
  if (!o) return CJS_E_OUT_OF_MEMORY;
  o[0] = 'B'; o[1] = 'Z'; o[2] = 'h'; o[3] = (uint8_t)('0' + level);
  uint64_t pos = 32; uint32_t scrc = 0;
  for (auto& x : sh) {
    const size_t nbytes = (size_t)((x.bits + 7) / 8);
    const unsigned s = (unsigned)(pos & 7); size_t ob = (size_t)(pos >> 3);
    for (size_t i = 0; i < nbytes; i++) {
      uint8_t b = x.bytes[i];
      if (i == nbytes - 1 && (x.bits & 7)) b &= (uint8_t)(0xFF << (8 - (x.bits & 7)));
      o[ob + i] |= (uint8_t)(b >> s);
      if (s) o[ob + i + 1] |= (uint8_t)(b << (8 - s));
    }
    pos += x.bits;
    for (long k = 0; k < x.count; k++) scrc = ((scrc << 1) | (scrc >> 31)) ^ x.crcs[(size_t)k];
  }
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
  HostCache& hc = g_host_cache[dev];
  std::lock_guard<std::mutex> lock(hc.mu);
  const size_t out_cap = (n + n / 4 + 4096 + 3) & ~(size_t)3;
  int rc = 0;
  static const bool dbg = getenv("CJS_DEBUG") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto t0 = now();
  if (!hc.ctx || hc.ctx->level != level || hc.ctx->max_input < n) {
    if (hc.ctx) { cjs_ctx_destroy(hc.ctx); hc.ctx = nullptr; }
    rc = cjs_ctx_create(&hc.ctx, -1, n, level);
    if (rc) { hc.ctx = nullptr; return rc; }
  }
  if (hc.in_cap < n || !hc.d_in) {
    if (hc.d_in) (void)hipFree(hc.d_in);
    hc.d_in = nullptr; hc.in_cap = 0;
    if (hipMalloc((void**)&hc.d_in, n ? n : 4) != hipSuccess) rc = CJS_E_OUT_OF_MEMORY; else hc.in_cap = n ? n : 4;
  }
  if (!rc && (hc.out_cap < out_cap || !hc.d_out)) {
    if (hc.d_out) (void)hipFree(hc.d_out);
    hc.d_out = nullptr; hc.out_cap = 0;
    if (hipMalloc((void**)&hc.d_out, out_cap) != hipSuccess) rc = CJS_E_OUT_OF_MEMORY; else hc.out_cap = out_cap;
  }
  cjs_ctx* c = hc.ctx;
  const auto t1 = now();
  if (!rc && n && hipMemcpyAsync(hc.d_in, in, n, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = CJS_E_HIP;
  if (dbg && !rc) (void)hipStreamSynchronize(c->stream);
  const auto t2 = now();
  size_t len = 0;
  cjs_stats* st = (opts && opts->struct_size >= sizeof(cjs_opts)) ? opts->stats : nullptr;
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
  for (int d = 0; d < MAX_CACHED_DEVICES; d++) {
    HostCache& hc = g_host_cache[d];
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
