// api.hip — C ABI of libcjs_hip.so (include/cjs_hip.h): contexts, host-buffer entry points,
// stage-level entry points.  No CPU fallback: without a HIP device every call fails loudly.
#include "cjs_internal.h"
#include "rle1.h"
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>

using namespace cjs;

// detail text of the last failing call on this thread: the reference's `optDetail` (J/Bzip2_joined_.js:1385-1391)
namespace cjs {
static thread_local char g_detail[192] = {0};
void clear_detail() { g_detail[0] = 0; }
void set_detail(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt);
  vsnprintf(g_detail, sizeof g_detail, fmt, ap);
  va_end(ap);
}
}  // namespace cjs

extern "C" {

const char* cjs_version(void) { return "cjs_hip 0.1 (gfx950)"; }

int cjs_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void cjs_free(void* p) { cjs::HostPool::give(p); }

const char* cjs_last_error_detail(void) { return g_detail; }

const char* cjs_strerror(int code) {
  switch (code) {
    case CJS_OK: return "ok";
    case CJS_E_NOT_BZIP_DATA: return "Not bzip data";
    case CJS_E_DATA_ERROR: return "Data error";
    case CJS_E_OUT_OF_MEMORY: return "Out of memory";
    case CJS_E_OBSOLETE_INPUT: return "Obsolete (pre 0.9.5) bzip format not supported.";
    case CJS_E_BAD_LEVEL: return "Invalid block size multiplier";
    case CJS_E_BAD_MAGIC: return "Bad magic";
    case CJS_E_NO_DEVICE: return "no HIP device available (this library has no CPU fallback)";
    case CJS_E_HIP: return "HIP runtime error";
    case CJS_E_INVALID_ARG: return "invalid argument";
    case CJS_E_OUTPUT_TOO_SMALL: return "output buffer too small";
    case CJS_E_UNSUPPORTED: return "not supported";
    default: return "unknown error";
  }
}

}  // extern "C"

namespace cjs {

// ---- DevPool (see cjs_internal.h)
namespace {
struct PoolBuf { void* p; size_t bytes; int device; bool busy; };
std::mutex g_pool_mu;
std::vector<PoolBuf> g_pool;
}  // namespace
// Fit rule (the same as HostPool's): requests below 1 MiB are plain allocations of their own size class and never claim a big
// idle buffer; a cached buffer serves a request only if it is at most twice as large.  give() keeps at most
// CJS_DEVICE_POOL_MB (default 65536) of idle buffers per process: beyond that the largest idle ones are freed.
namespace {
size_t dev_idle_limit() {
  static const size_t mb = getenv("CJS_DEVICE_POOL_MB") ? (size_t)strtoull(getenv("CJS_DEVICE_POOL_MB"), nullptr, 10) : 65536;
  return mb << 20;
}
}  // namespace
void* DevPool::take(size_t bytes) {
  if (!bytes) bytes = 4;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    PoolBuf* best = nullptr;
    for (auto& b : g_pool)
      if (!b.busy && b.device == dev && b.bytes >= bytes && (b.bytes / 2 <= bytes || b.bytes <= ((size_t)1 << 20)) && (!best || b.bytes < best->bytes)) best = &b;
    if (best) { best->busy = true; return best->p; }
  }
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) {
    trim();                                              // cached-but-idle buffers may be what is in the way
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  }
  std::lock_guard<std::mutex> lock(g_pool_mu);
  g_pool.push_back(PoolBuf{p, bytes, dev, true});
  return p;
}
void DevPool::give(void* p) {
  if (!p) return;
  static const bool no_cache = getenv("CJS_NO_CTX_CACHE") != nullptr;
  std::lock_guard<std::mutex> lock(g_pool_mu);
  for (size_t i = 0; i < g_pool.size(); i++) if (g_pool[i].p == p) {
    if (no_cache) { (void)hipFree(p); g_pool.erase(g_pool.begin() + (long)i); return; }
    g_pool[i].busy = false;
    size_t idle = 0;
    for (auto& b : g_pool) if (!b.busy) idle += b.bytes;
    while (idle > dev_idle_limit()) {                    // over the limit: the largest idle buffers go first
      size_t big = g_pool.size();
      for (size_t k = 0; k < g_pool.size(); k++) if (!g_pool[k].busy && (big == g_pool.size() || g_pool[k].bytes > g_pool[big].bytes)) big = k;
      if (big == g_pool.size()) break;
      idle -= g_pool[big].bytes;
      (void)hipFree(g_pool[big].p);                      // (hipFree finds the owning device from the pointer)
      g_pool.erase(g_pool.begin() + (long)big);
    }
    return;
  }
  (void)hipFree(p);                                      // not ours: plain buffer
}
void DevPool::trim() {
  int cur = 0;
  const bool have = hipGetDevice(&cur) == hipSuccess;
  std::lock_guard<std::mutex> lock(g_pool_mu);
  for (size_t i = 0; i < g_pool.size();) {
    if (g_pool[i].busy) { i++; continue; }
    if (hipSetDevice(g_pool[i].device) == hipSuccess) (void)hipFree(g_pool[i].p);
    g_pool.erase(g_pool.begin() + (long)i);
  }
  if (have) (void)hipSetDevice(cur);
}

// ---- HostPool (see cjs_internal.h)
namespace {
struct HostBuf { void* p; size_t bytes; bool busy; };
std::mutex g_host_mu;
std::vector<HostBuf> g_host;
size_t host_idle_limit() {
  static const size_t mb = getenv("CJS_PINNED_RESULT_MB") ? (size_t)strtoull(getenv("CJS_PINNED_RESULT_MB"), nullptr, 10) : 2048;
  return mb << 20;
}
}  // namespace
void* HostPool::take(size_t bytes) {
  if (!bytes) bytes = 1;
  const size_t limit = host_idle_limit();
  if (bytes < ((size_t)1 << 20) || !limit) return malloc(bytes);
  {
    std::lock_guard<std::mutex> lock(g_host_mu);
    HostBuf* best = nullptr;
    for (auto& b : g_host) if (!b.busy && b.bytes >= bytes && b.bytes / 2 <= bytes && (!best || b.bytes < best->bytes)) best = &b;
    if (best) { best->busy = true; return best->p; }
  }
  void* p = nullptr;
  const size_t cap = bytes + bytes / 16;                 // a later result of about the same size fits too
  if (hipHostMalloc(&p, cap, hipHostMallocPortable) != hipSuccess || !p)      // (portable: every GPU of a multi-device call copies into it) { (void)hipGetLastError(); return malloc(bytes); }
  std::lock_guard<std::mutex> lock(g_host_mu);
  g_host.push_back(HostBuf{p, cap, true});
  return p;
}
void HostPool::give(void* p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lock(g_host_mu);
    for (size_t i = 0; i < g_host.size(); i++) if (g_host[i].p == p) {
      g_host[i].busy = false;
      size_t idle = 0;
      for (auto& b : g_host) if (!b.busy) idle += b.bytes;
      for (size_t j = 0; j < g_host.size() && idle > host_idle_limit();) {      // over the limit: the largest idle buffers go first
        size_t big = g_host.size();
        for (size_t k = 0; k < g_host.size(); k++) if (!g_host[k].busy && (big == g_host.size() || g_host[k].bytes > g_host[big].bytes)) big = k;
        if (big == g_host.size()) break;
        idle -= g_host[big].bytes;
        (void)hipHostFree(g_host[big].p);
        g_host.erase(g_host.begin() + (long)big);
      }
      return;
    }
  }
  free(p);                                               // not pinned: plain malloc
}
void HostPool::trim() {
  std::lock_guard<std::mutex> lock(g_host_mu);
  for (size_t i = 0; i < g_host.size();) {
    if (g_host[i].busy) { i++; continue; }
    (void)hipHostFree(g_host[i].p);
    g_host.erase(g_host.begin() + (long)i);
  }
}

int select_device(const cjs_opts* opts) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return CJS_E_NO_DEVICE;
  if (opts && opts->struct_size >= sizeof(cjs_opts) && opts->device >= 0) {
    if (opts->device >= n) return CJS_E_INVALID_ARG;
    CJS_HIP_TRY(hipSetDevice(opts->device));
  }
  return 0;
}

}  // namespace cjs

extern "C" int cjs_stage_bwt(const uint8_t* in, size_t n, int block_len, int cyclic, uint8_t* out, int32_t* pidx, const cjs_opts* opts) {
  CJS_GUARD_BEGIN
  CJS_TRY(select_device(opts));
  if (n == 0) return 0;
  if (block_len <= 0) return CJS_E_INVALID_ARG;
  const uint32_t stride = (uint32_t)block_len;
  const uint32_t nb = (uint32_t)((n + stride - 1) / stride);
  const uint32_t n_last = (uint32_t)(n - (size_t)(nb - 1) * stride);
  Arena arena;
  CJS_TRY(arena.init(BwtWork::bytes_needed(n) + 2 * ((n + 511) & ~(size_t)255) + 4 * (size_t)nb + 8192));
  BwtWork w;
  int rc = w.carve(arena, n);
  uint8_t* d_T = arena.take<uint8_t>(n);
  uint8_t* d_U = arena.take<uint8_t>(n);
  uint32_t* d_p = arena.take<uint32_t>(nb);
  hipStream_t s = nullptr;
  if (!rc && (!d_T || !d_U || !d_p)) rc = CJS_E_OUT_OF_MEMORY;
  if (!rc && hipStreamCreate(&s) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipMemcpyAsync(d_T, in, n, hipMemcpyHostToDevice, s) != hipSuccess) rc = CJS_E_HIP;
  if (!rc) rc = bwt_run(s, w, d_T, nb, stride, n_last, cyclic != 0, d_U, d_p, opts && opts->struct_size >= sizeof(cjs_opts) ? opts->stats : nullptr);
  if (!rc && hipMemcpyAsync(out, d_U, n, hipMemcpyDeviceToHost, s) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipMemcpyAsync(pidx, d_p, 4 * (size_t)nb, hipMemcpyDeviceToHost, s) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = CJS_E_HIP;
  if (s) (void)hipStreamDestroy(s);
  w.release_host();
  arena.destroy();
  return rc;
  CJS_GUARD_END(CJS_E_OUT_OF_MEMORY, CJS_E_HIP)
}

extern "C" int cjs_stage_rle1(const uint8_t* in, size_t n, int level, uint8_t* blocks, size_t blocks_cap,
                              uint32_t* block_len, uint32_t* block_crc, uint64_t* block_start, long cap_blocks, long* nblocks,
                              const cjs_opts* opts) {
  CJS_GUARD_BEGIN
  CJS_TRY(select_device(opts));
  if (level < 1 || level > 9) return CJS_E_BAD_LEVEL;
  const uint32_t cap = (uint32_t)level * 100000u - 19u;
  *nblocks = 0;
  if (n == 0) return 0;
  Arena arena;
  const size_t maxb = Rle1Work::max_blocks_for(n, cap);
  CJS_TRY(arena.init(Rle1Work::bytes_needed(n, cap) + n + maxb * cap + 65536));
  Rle1Work w;
  int rc = w.carve(arena, n, cap);
  uint8_t* d_in = arena.take<uint8_t>(n);
  uint8_t* d_blocks = arena.take<uint8_t>(maxb * cap);
  hipStream_t s = nullptr;
  if (!rc && (!d_in || !d_blocks)) rc = CJS_E_OUT_OF_MEMORY;
  if (!rc && hipStreamCreate(&s) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipMemcpyAsync(d_in, in, n, hipMemcpyHostToDevice, s) != hipSuccess) rc = CJS_E_HIP;
  uint32_t nb = 0;
  if (!rc) rc = rle1_run(s, w, d_in, n, &nb);
  if (!rc) rc = rle1_finish(s, w, d_in, n, 0, nb, d_blocks);
  if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && ((long)nb > cap_blocks || (size_t)nb * cap > blocks_cap)) rc = CJS_E_OUTPUT_TOO_SMALL;
  if (!rc && nb) {
    std::vector<RleBlock> hb(nb);
    if (hipMemcpy(hb.data(), w.blocks, sizeof(RleBlock) * nb, hipMemcpyDeviceToHost) != hipSuccess) rc = CJS_E_HIP;
    if (!rc && hipMemcpy(block_len, w.block_len, 4 * (size_t)nb, hipMemcpyDeviceToHost) != hipSuccess) rc = CJS_E_HIP;
    if (!rc && hipMemcpy(block_crc, w.block_crc, 4 * (size_t)nb, hipMemcpyDeviceToHost) != hipSuccess) rc = CJS_E_HIP;
    if (!rc && hipMemcpy(blocks, d_blocks, (size_t)nb * cap, hipMemcpyDeviceToHost) != hipSuccess) rc = CJS_E_HIP;
    if (!rc) for (uint32_t k = 0; k < nb; k++) block_start[k] = hb[k].s;
  }
  *nblocks = (long)nb;
  if (s) (void)hipStreamDestroy(s);
  w.release();
  arena.destroy();
  return rc;
  CJS_GUARD_END(CJS_E_OUT_OF_MEMORY, CJS_E_HIP)
}

