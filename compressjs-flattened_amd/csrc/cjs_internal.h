// cjs_internal.h — shared host-side declarations of the HIP library (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <new>
#include "cjs_hip.h"

#define CJS_HIP_TRY(expr)                                                                   \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      fprintf(stderr, "[cjs_hip] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return _e == hipErrorOutOfMemory ? CJS_E_OUT_OF_MEMORY : CJS_E_HIP;                    \
    }                                                                                       \
  } while (0)
#define CJS_TRY(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

namespace cjs {

// thread-local detail text behind cjs_last_error_detail() (api.hip)
void clear_detail();
void set_detail(const char* fmt, ...);

// Body of an extern "C" entry point: C++ exceptions (std::bad_alloc from a container fed by untrusted sizes) never
// cross the C ABI, they become return codes.
#define CJS_GUARD_BEGIN try {
#define CJS_GUARD_END(oom_value, other_value)                          \
  } catch (const std::bad_alloc&) { return (oom_value); }              \
  catch (...) { return (other_value); }

struct DevPool;
// simple device arena: one hipMalloc (or one buffer of the per-device pool), bump allocation, 256-byte aligned
struct Arena {
  uint8_t* base = nullptr;
  size_t cap = 0, used = 0;
  bool pooled = false;
  int init(size_t bytes) {
    CJS_HIP_TRY(hipMalloc((void**)&base, bytes));
    cap = bytes; used = 0; pooled = false;
    return 0;
  }
  inline int init_pooled(size_t bytes);      // from DevPool: kept between calls (cjs_trim / CJS_NO_CTX_CACHE give it back)
  inline void destroy();
  template <typename T> T* take(size_t n) {
    size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
    if (used + bytes > cap) return nullptr;
    T* p = (T*)(base + used);
    used += bytes;
    return p;
  }
};

struct EventTimer {   // accumulates device time of bracketed regions on one stream
  hipStream_t s;
  hipEvent_t a, b;
  bool ok = false;
  int init(hipStream_t st) { s = st; CJS_HIP_TRY(hipEventCreate(&a)); CJS_HIP_TRY(hipEventCreate(&b)); ok = true; return 0; }
  void destroy() { if (ok) { (void)hipEventDestroy(a); (void)hipEventDestroy(b); ok = false; } }
  void start() { (void)hipEventRecord(a, s); }
  double stop() { (void)hipEventRecord(b, s); (void)hipEventSynchronize(b); float ms = 0; (void)hipEventElapsedTime(&ms, a, b); return ms; }
};

// Per-device cache of device buffers for the host-buffer entry points (hipMalloc / hipFree of the multi-GB scratch of
// one call cost milliseconds): take() hands out the smallest cached free buffer that is large enough or allocates one,
// give() returns it to the cache, trim() frees everything cached (cjs_trim; CJS_NO_CTX_CACHE=1 makes give() free at once).
struct DevPool {
  static void* take(size_t bytes);
  static void give(void* p);
  static void trim();
};
// Result buffers handed to the caller by the host-buffer entry points ("malloc'd by the library; release with cjs_free").
// Large results come from a cache of PINNED host buffers: the device-to-host copy of the result is then one DMA at link speed
// into pages that exist already, instead of a staged copy into fresh pageable memory that faults page by page (100 MB:
// ~20 ms).  cjs_free() returns such a buffer to the cache (at most CJS_PINNED_RESULT_MB of idle buffers are kept, default
// 2048; 0 = results are plain malloc), cjs_trim() frees the idle ones.  Small results are plain malloc.
struct HostPool {
  static void* take(size_t bytes);      // never pinned below 1 MiB; falls back to malloc when pinning fails
  static void give(void* p);            // any pointer take() returned (or malloc'd memory)
  static void trim();
};
inline int Arena::init_pooled(size_t bytes) {
  base = (uint8_t*)DevPool::take(bytes);
  if (!base) return CJS_E_OUT_OF_MEMORY;
  cap = bytes; used = 0; pooled = true;
  return 0;
}
inline void Arena::destroy() {
  if (base) { if (pooled) DevPool::give(base); else (void)hipFree(base); }
  base = nullptr; cap = used = 0;
}

constexpr uint32_t RS_TILE = 4096;   // radix-sort tile (256 threads x 16 keys)

// Workspace of the suffix sorter for up to `cap` suffixes (all blocks of a batch together).
struct LaunchTimes {   // event pairs around the dominant kernel; resolved after the stream has drained
  static constexpr int MAXP = 512;
  hipEvent_t ev[2 * MAXP];
  uint64_t elems[MAXP];
  int n = 0, made = 0;
  bool enabled = false, open = false;
  uint64_t min_elems = 0;          // only launches over at least this many elements are timed (the full-size passes of a sort)
  void begin(hipStream_t s, uint64_t e) {
    open = false;
    if (!enabled || n >= MAXP || e < min_elems) return;
    open = true;
    while (made < 2 * (n + 1)) { if (hipEventCreate(&ev[made]) != hipSuccess) { enabled = false; open = false; return; } made++; }
    elems[n] = e;
    (void)hipEventRecord(ev[2 * n], s);
  }
  void end(hipStream_t s) { if (!enabled || !open || n >= MAXP) return; (void)hipEventRecord(ev[2 * n + 1], s); n++; open = false; }
  void resolve(cjs_stats* st) {
    double ms = 0; uint64_t e = 0;
    for (int i = 0; i < n; i++) { float t = 0; if (hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]) == hipSuccess) { ms += t; e += elems[i]; } }
    if (st && n) { st->ms_bwt_dominant = ms / n; st->bwt_dominant_launches = (uint64_t)n; st->bwt_dominant_bytes = e; }
    reset();
  }
  void reset() { for (int i = 0; i < made; i++) (void)hipEventDestroy(ev[i]); n = made = 0; }
  ~LaunchTimes() { reset(); }
  // hands the recorded launches to `to` (appended; this object is left empty, still enabled)
  void move_into(LaunchTimes& to) {
    for (int i = 0; i < n && to.n < MAXP; i++) {
      if (to.made > 2 * to.n) { for (int j = 2 * to.n; j < to.made; j++) (void)hipEventDestroy(to.ev[j]); to.made = 2 * to.n; }
      to.ev[2 * to.n] = ev[2 * i]; to.ev[2 * to.n + 1] = ev[2 * i + 1]; to.elems[to.n] = elems[i]; to.n++; to.made = 2 * to.n;
    }
    for (int i = 2 * n; i < made; i++) (void)hipEventDestroy(ev[i]);
    n = made = 0;
  }
};

struct BwtWork {
  size_t cap = 0;
  uint64_t* key[2] = {nullptr, nullptr};
  uint32_t* val[2] = {nullptr, nullptr};
  uint32_t* pos[2] = {nullptr, nullptr};
  uint32_t* gord = nullptr;
  uint32_t* R = nullptr;
  uint32_t* SA = nullptr;
  uint8_t* dflag = nullptr;      // per slot of a round >= 2: 1 = its group is too large for the tile sorters
  uint32_t* hist = nullptr;      // hist_words(tiles): 256 per tile (tile-major) + the chunk sums of the long-segment scan
  uint32_t* bintot = nullptr;    // 256 per segment
  uint32_t* tile_cnt = nullptr;  // 3 * tiles (+ scanned copies)
  uint32_t* counters = nullptr;  // 16: [0] survivors [1] groups [8] tile ticket [9] look-back error
  uint32_t* ghist = nullptr;     // [8][256] digit histograms + [8][256] their exclusive scans (onesweep passes)
  uint32_t* h_counters = nullptr;  // pinned host mirror
  hipEvent_t ev_scan = nullptr;    // recorded behind the tile scan of a round (the host waits for the counters, not for the round)
  void release_host() { if (h_counters) (void)hipHostFree(h_counters); h_counters = nullptr; if (ev_scan) (void)hipEventDestroy(ev_scan); ev_scan = nullptr; lt.reset(); }
  uint32_t hist_tiles = 0, bintot_segs = 0;   // capacity of hist (tiles) and bintot (segments)
  LaunchTimes lt;                  // dominant-kernel events of the last bwt_run that was given a stats struct
  bool no_large_groups = false;    // per bwt_run: no unresolved group exceeds the tile sorter's limit any more
  // segmented sorts round every block up to whole tiles: room for one extra tile per 64 Ki elements
  static size_t hist_tiles_for(size_t cap) { return (cap + RS_TILE - 1) / RS_TILE + cap / 65536 + 258; }
  static size_t segs_for(size_t cap) { return cap / 65536 + 2; }
  static size_t hist_words(size_t tiles) { return 256 * (tiles + tiles / 64 + 2); }
  static size_t bytes_needed(size_t cap);
  int carve(Arena& a, size_t cap);
};

// Suffix-sorts nb blocks (block k = d_T[k*stride .. +n_k), n_k = stride except the last = n_last)
// and writes the BWT bytes to d_U (same layout) and the primary indices to d_pidx[nb].
int bwt_run(hipStream_t s, BwtWork& w, const uint8_t* d_T, uint32_t nb, uint32_t stride, uint32_t n_last,
            bool cyclic, uint8_t* d_U, uint32_t* d_pidx, cjs_stats* stats, bool resolve_stats = true);

}  // namespace cjs
