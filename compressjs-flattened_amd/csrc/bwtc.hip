// bwtc.hip — BWTC.compressFile (J/BWTC_joined_.js:1698-1825) on the MI355X.
//
// Per block (level*100000 input bytes, no RLE1): sentinel BWT (bwt.hip, shared with bzip2) -> symbol map +
// MTF + RLE2 (mtf.hip, shared: the RLE2 stream is bzip2's minus the EOB symbol, J/BWTC_joined_.js:1767-1819)
// -> adaptive model evaluation on the GPU, one wave per block: the model (FenwickModel :1496-1661 for
// levels 6-9, DefSumModel :1327-1439 for levels 1-5) is rebuilt per block (:1791) and does not depend on
// the coder state, so each block yields its list of coder steps (sy_f, lt_f, tot_f | shift).
// The range coder itself (RangeCoder :40-153) carries ONE (low, range) state across all blocks (:1699):
// range' = floor(range / tot) * sy is a serial integer chain over the whole file, so that tail runs on
// one host thread, in block order, over the GPU-produced steps.
#include "cjs_internal.h"
#include "prims.hpp"
#include "mtf.h"
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <sched.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace cjs {
int select_device(const cjs_opts* opts);
int ibwt_sentinel_run(hipStream_t s, const uint8_t* d_T, uint32_t max_len, uint32_t nb, const uint32_t* lens, const uint32_t* pidx, uint8_t* d_out);
}
using namespace cjs;

namespace cjs {

constexpr uint32_t F_MAX = 0xFF00u, F_INC = 0x0100u;
constexpr uint64_t STEP_SHIFT_FLAG = 1ull << 63;   // step is encodeShift(sy, lt, shift) instead of encodeFreq

// ---------------------------------------------------------------- FenwickModel evaluated 64 symbols at a time
// Between two rescales the model only counts: every coded symbol adds F_INC to its own frequency, and a symbol whose
// frequency is zero is announced by the escape symbol first.  So for a chunk of 64 symbols with NO rescale inside,
// everything a coder step needs is a prefix count over the chunk:
//   frequency of s at position j      = leaf(s) at the chunk start + F_INC * #{j' < j : s_j' = s}
//   cumulative frequency below s      = cum(s)  at the chunk start + F_INC * #{j' < j : s_j' < s}
//   total                             = root    at the chunk start + F_INC * (steps so far)
//   escape events                     = first occurrences of symbols whose frequency was zero at the chunk start
// The 64x64 relations are two 64-bit masks per lane (same symbol / smaller symbol), built with 64 readlanes; the counts
// are popcounts.  The position whose steps make the total reach F_MAX (rescale), and the rare "last unseen symbol"
// escape, are done one at a time by fm_serial on the same flat leaf array; the chunk continues behind them.
// The model is kept as flat leaves (packed hi = main count, lo = escape-context count, as in the reference's tree
// leaves) plus an exclusive prefix array rebuilt per chunk: no tree.
__device__ __forceinline__ void fm_build_cum(const uint32_t* leaf, uint32_t* cum, int ns) {
  const int lane = lane_id(), i0 = 5 * lane;
  uint32_t pre[5], sum = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) { pre[k] = sum; sum += i0 + k < ns ? leaf[i0 + k] : 0u; }
  const uint32_t ex = wave_incl_sum(sum) - sum;
#pragma unroll
  for (int k = 0; k < 5; k++) if (i0 + k <= ns) cum[i0 + k] = ex + pre[k];        // cum[ns] = root
  __builtin_amdgcn_wave_barrier();
}
__device__ void fm_rescale(uint32_t* leaf, int ns, int e) {    // _rescale (:1623-1654) on the flat leaves; e = slot of the escape symbol
  const int lane = lane_id();
  bool esc_here = false;
  for (int i = lane; i < ns; i += 64) {
    if (i == e) continue;
    uint32_t prob = leaf[i];
    if (prob & 0xFFFFu) { esc_here = true; continue; }
    prob = (prob & 0xFFFEFFFEu) >> 1;
    if (prob == 0) { prob = 1u; esc_here = true; }
    leaf[i] = prob;
  }
  const bool no_escape = __ballot(esc_here) == 0ull;
  if (lane == 0) {
    uint32_t prob = leaf[e];
    prob = (prob & 0xFFFEFFFEu) >> 1;
    if (no_escape) prob = 0; else if (prob == 0) prob = 1u << 16;
    leaf[e] = prob;
  }
  __builtin_amdgcn_wave_barrier();
}
// encode(symbol) for ONE symbol (:1530-1571), any state: escape, last escape, rescales
__device__ void fm_serial(uint32_t* leaf, uint32_t* cum, int ns, int e, int symbol, uint64_t* out, uint32_t& n) {
  const int lane = lane_id();
  fm_build_cum(leaf, cum, ns);
  uint32_t root = __builtin_amdgcn_readfirstlane(cum[ns]);
  const uint32_t leafv = __builtin_amdgcn_readfirstlane(leaf[symbol]);
  const bool esc = (leafv >> 16) == 0;
  if (esc) {
    const uint32_t sy_raw = __builtin_amdgcn_readfirstlane(leaf[e]), lt = __builtin_amdgcn_readfirstlane(cum[e]);
    uint32_t update = F_INC << 16;
    if ((root & 0xFFFFu) == 1u) update = 0u - sy_raw;          // last escape: zero it out
    if (lane == 0) { out[n] = (uint64_t)(sy_raw >> 16) | ((uint64_t)(lt >> 16) << 16) | ((uint64_t)(root >> 16) << 32); leaf[e] = sy_raw + update; }
    n++;
    __builtin_amdgcn_wave_barrier();
    root += update;
    if ((root >> 16) >= F_MAX) fm_rescale(leaf, ns, e);
    fm_build_cum(leaf, cum, ns);
    root = __builtin_amdgcn_readfirstlane(cum[ns]);
    const uint32_t sy2 = __builtin_amdgcn_readfirstlane(leaf[symbol]), lt2 = __builtin_amdgcn_readfirstlane(cum[symbol]);
    if (lane == 0) { out[n] = (uint64_t)(sy2 & 0xFFFFu) | ((uint64_t)(lt2 & 0xFFFFu) << 16) | ((uint64_t)(root & 0xFFFFu) << 32); leaf[symbol] = sy2 + ((F_INC << 16) - 1u); }
    n++;
    __builtin_amdgcn_wave_barrier();
    root += (F_INC << 16) - 1u;
    if ((root >> 16) >= F_MAX) fm_rescale(leaf, ns, e);
  } else {
    const uint32_t lt = __builtin_amdgcn_readfirstlane(cum[symbol]);
    if (lane == 0) { out[n] = (uint64_t)(leafv >> 16) | ((uint64_t)(lt >> 16) << 16) | ((uint64_t)(root >> 16) << 32); leaf[symbol] = leafv + (F_INC << 16); }
    n++;
    __builtin_amdgcn_wave_barrier();
    root += F_INC << 16;
    if ((root >> 16) >= F_MAX) fm_rescale(leaf, ns, e);
  }
}

// The chunks of a block are a chain only through the model state (leaf counts, steps so far); the 64x64 relation masks of a
// chunk -- most of its instructions -- depend on nothing but its symbols.  FP_WAVES waves share a block: wave w takes the
// chunks w, w + FP_WAVES, ...; it builds its masks while the waves in front of it run their state parts, then waits for the
// baton (an LDS ticket), runs the state part of its chunk on the shared leaf array and passes the baton on.  A lone wave
// issues about one instruction per 8 cycles, so the block's critical path shrinks to the state parts alone.
constexpr int FP_WAVES = 8;
__global__ __launch_bounds__(64 * FP_WAVES) void bwtc_fenwick_par(MtfBufs mb, uint64_t* __restrict__ steps, size_t step_stride, uint32_t* __restrict__ nsteps) {
  __shared__ uint32_t leaf[328], cum[328];
  __shared__ uint32_t turn_s, n_s;           // next chunk whose state part may run; coder steps emitted so far
  const uint32_t blk = blockIdx.x;
  const uint32_t wv = threadIdx.x >> 6;
  const uint32_t asz = mb.asz[blk], nsym = mb.npos[blk] - 1;      // drop bzip2's EOB
  const uint16_t* A = mb.A + (size_t)blk * mb.a_stride;
  uint64_t* out = steps + (size_t)blk * step_stride;
  const int lane = lane_id();
  const int ns = (int)asz + 2;
  // The reference keeps the counts in an implicit binary tree with the leaves at [ns, 2ns): "cumulative frequency below s"
  // is the sum over the leaves LEFT of s in that tree, and when ns is not a power of two the bottom-level leaves
  // (symbols >= r0) come first.  All slots below are in that order: slot(s) = (s - r0) mod ns.
  int dpt = 0; while ((2 << dpt) <= 2 * ns - 1) dpt++;                  // depth of the deepest leaf: floor(log2(2ns - 1))
  const int r0 = (1 << dpt) - ns, e = ns - 1 - r0;                      // slot of the escape symbol (symbol ns-1)
  for (int i = (int)threadIdx.x; i < 328; i += 64 * FP_WAVES) { leaf[i] = i == e ? (F_INC << 16) : i < ns ? 1u : 0u; cum[i] = 0; }
  if (threadIdx.x == 0) { turn_s = 0; n_s = 0; }
  __syncthreads();
  const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;       // lanes < lane
  const uint32_t nchunks = (nsym + 63u) / 64u;
  for (uint32_t ch = wv; ch < nchunks; ch += FP_WAVES) {
    const uint32_t base = ch * 64u;
    const uint32_t cnt = __builtin_amdgcn_readfirstlane(nsym - base < 64 ? nsym - base : 64);
    uint32_t mine = 0xFFFFu;
    if ((uint32_t)lane < cnt) { const int sy = (int)A[base + lane]; mine = (uint32_t)(sy >= r0 ? sy - r0 : sy + ns - r0); }
    const uint64_t LTE = __ballot(mine < (uint32_t)e);                  // positions whose slot lies left of the escape symbol
    // relations inside the chunk: lanes with the same / a smaller symbol (all 64 pairs, any window is a mask away)
    uint64_t eqm = 0, ltm = 0;
    for (uint32_t jp = 0; jp < cnt; jp++) {
      const uint32_t sj = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)jp);
      eqm |= sj == mine ? 1ull << jp : 0ull;
      ltm |= sj < mine ? 1ull << jp : 0ull;
    }
    // ---- the baton: everything below reads and writes the shared model state
    while ((uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&turn_s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) != ch) __builtin_amdgcn_s_sleep(1);
    uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&n_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    uint32_t start = 0;
    while (start < cnt) {
      fm_build_cum(leaf, cum, ns);
      const uint32_t root = __builtin_amdgcn_readfirstlane(cum[ns]);
      const uint32_t esc_leaf = __builtin_amdgcn_readfirstlane(leaf[e]);
      const uint64_t win = (~0ull << start) & (cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull));    // positions [start, cnt)
      const bool active = (win >> lane) & 1ull;
      const uint64_t before = below & win;                                                   // positions [start, lane)
      const uint32_t leafv = active ? leaf[mine] : 0x10000u;
      const uint32_t cumv = active ? cum[mine] : 0u;
      const uint32_t eqb = (uint32_t)__builtin_popcountll(eqm & before);
      const bool escj = active && (leafv >> 16) == 0 && eqb == 0;                            // first occurrence of a zero-count symbol
      const uint64_t ESCM = __ballot(escj);
      const uint32_t Eb = (uint32_t)__builtin_popcountll(ESCM & before);
      const uint32_t rel = (uint32_t)lane - start;
      const uint32_t through = rel + 1u + Eb + (escj ? 1u : 0u);                             // steps up to and including this position
      const uint32_t root_hi = root >> 16, root_lo = root & 0xFFFFu;
      const uint32_t kstar = root_hi >= F_MAX ? 1u : (F_MAX - root_hi + F_INC - 1u) / F_INC;  // the step after which the total reaches F_MAX
      const bool cut = active && (through >= kstar || (escj && root_lo - Eb == 1u));
      const uint64_t cutm = __ballot(cut);
      const uint32_t c = cutm ? (uint32_t)__builtin_ctzll(cutm) : cnt;                       // [start, c) in parallel, c one at a time
      const uint64_t parw = win & (c == 64 ? ~0ull : ((1ull << c) - 1ull));
      if ((parw >> lane) & 1ull) {
        const uint32_t tot = root_hi + F_INC * (rel + Eb);
        const uint32_t o = n + rel + Eb;
        if (escj) {
          const uint32_t esc_hi = (esc_leaf >> 16) + F_INC * Eb;
          const uint32_t cum_e = (uint32_t)__builtin_amdgcn_readfirstlane(cum[e]);      // (the builtin returns int: shift the unsigned copy)
          const uint32_t lt_e = (cum_e >> 16) + F_INC * (uint32_t)__builtin_popcountll(LTE & before);
          out[o] = (uint64_t)esc_hi | ((uint64_t)lt_e << 16) | ((uint64_t)tot << 32);
          const uint32_t lt_lo = (cumv & 0xFFFFu) - (uint32_t)__builtin_popcountll(ESCM & ltm & before);
          out[o + 1] = 1ull | ((uint64_t)lt_lo << 16) | ((uint64_t)(root_lo - Eb) << 32);
        } else {
          const uint32_t sy = (leafv >> 16) + F_INC * eqb;
          // counts left of this slot: earlier symbols in lower slots, and the escape symbol's own increments if it lies left
          const uint32_t lt = (cumv >> 16) + F_INC * ((uint32_t)__builtin_popcountll(ltm & before) + (mine > (uint32_t)e ? Eb : 0u));
          out[o] = (uint64_t)sy | ((uint64_t)lt << 16) | ((uint64_t)tot << 32);
        }
        // the last occurrence of a symbol inside [start, c) writes its new leaf (count so far, escape count 0)
        const uint64_t later = eqm & parw & ~below & ~(1ull << lane);
        if (!later) leaf[mine] = ((leafv >> 16) + F_INC * (eqb + 1u)) << 16;
      }
      const uint32_t nesc = (uint32_t)__builtin_popcountll(ESCM & parw);
      if (lane == 0 && nesc) leaf[e] = esc_leaf + ((F_INC * nesc) << 16);
      n += (c - start) + nesc;
      __builtin_amdgcn_wave_barrier();
      if (c < cnt) {
        fm_serial(leaf, cum, ns, e, __builtin_amdgcn_readlane((int)mine, (int)c), out, n);
        start = c + 1;
      } else start = cnt;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      __hip_atomic_store(&n_s, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_store(&turn_s, ch + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);      // (orders the leaf / n_s stores in front of it)
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) nsteps[blk] = n_s;
}

// ---------------------------------------------------------------- DefSumModel (levels 1-5), one lane per block
__global__ __launch_bounds__(64) void bwtc_defsum(MtfBufs mb, uint64_t* __restrict__ steps, size_t step_stride, uint32_t* __restrict__ nsteps) {
  __shared__ uint16_t prob[304], esc[304], upd[304];
  const uint32_t blk = blockIdx.x;
  if (threadIdx.x != 0) return;                         // tiny state, O(1) per symbol: serial on one lane
  const uint32_t asz = mb.asz[blk], nsym = mb.npos[blk] - 1;
  const uint16_t* A = mb.A + (size_t)blk * mb.a_stride;
  uint64_t* out = steps + (size_t)blk * step_stride;
  const int ns = (int)asz + 1;                          // numSyms = size; ESCAPE = ns
  for (int i = 0; i < 304; i++) { prob[i] = 0; esc[i] = 0; upd[i] = 0; }
  prob[ns + 1] = 256;
  for (int i = 0; i <= ns; i++) esc[i] = (uint16_t)i;
  int update_count = 0, update_thresh = 128;
  uint32_t n = 0;
  auto do_update = [&](int symbol) {                    // _update (:1359-1421), encoder side
    if (symbol == ns) {
      if (upd[symbol] >= 40) return;
      if (update_count >= update_thresh - 1) return;
    }
    upd[symbol]++; update_count++;
    if (update_count < update_thresh) return;
    int cum = 0, cum_esc = 0, odd = 0, i;
    esc[0] = 0; prob[0] = 0;
    for (i = 0; i < ns + 1; i++) {
      const int np = ((prob[i + 1] - prob[i]) >> 1) + upd[i];
      prob[i] = (uint16_t)cum; esc[i] = (uint16_t)cum_esc;
      if (np) { cum += np; odd += np & 1; } else cum_esc++;
    }
    prob[i] = (uint16_t)cum;
    update_thresh = 256 - (cum - odd) / 2;
    for (i = 0; i < ns + 1; i++) upd[i] = 0;
    upd[ns] = 1; update_count = 1;
  };
  for (uint32_t k = 0; k < nsym; k++) {
    int symbol = A[k];
    uint32_t lt = prob[symbol], sy = (uint32_t)prob[symbol + 1] - lt;
    if (sy) { out[n++] = STEP_SHIFT_FLAG | sy | ((uint64_t)lt << 16) | (8ull << 32); do_update(symbol); continue; }
    // escape (:1430-1438)
    { const uint32_t elt = prob[ns], esy = (uint32_t)prob[ns + 1] - elt;
      out[n++] = STEP_SHIFT_FLAG | esy | ((uint64_t)elt << 16) | (8ull << 32); do_update(ns); }
    lt = esc[symbol]; sy = (uint32_t)esc[symbol + 1] - lt;
    out[n++] = (uint64_t)sy | ((uint64_t)lt << 16) | ((uint64_t)esc[ns] << 32);
    do_update(symbol);
  }
  nsteps[blk] = n;
}

}  // namespace cjs

// ---------------------------------------------------------------- host: framing + serial range coder
namespace {

// RangeCoder encode side (J/BWTC_joined_.js:40-153).  The interval arithmetic is one serial chain over the whole file, but it
// is TWO chains: `range` never depends on `low` (range' = f(range, step); low only receives r * lt and is shifted when range is).
// In split mode (default) the calling thread runs the range chain alone and leaves one record per step -- the addend r * lt and
// the number of byte shifts in front of it -- in a ring of buffers; a second host thread replays them on `low` (carry
// propagation, pending 0xFF bytes, output).  Same bytes as the one-thread form (CJS_BWTC_SPLIT_CODER=0), which runs both
// chains in one loop at ~5 ns per step.
struct HostCoder {
  std::vector<uint8_t>& out;                            // bytes [0, len) are output; the vector is kept larger (reserve())
  static constexpr uint32_t RING = 8, RCAP = 1u << 15;
  std::vector<uint64_t> ring[RING];
  uint32_t ring_n[RING];
  int ring_cpu[RING];
  std::thread low_thread;
  // the two threads of the split mode write their own state at every step: each side on cache lines of its own (with range
  // and low on one line the pair ran 8x slower than one thread)
  alignas(128) uint8_t* data = nullptr;                 // ---- low side
  size_t len = 0;
  uint32_t low = 0, help = 0, bytecount = 0;
  int buffer = 0;
  alignas(128) uint32_t range = 0x80000000u;            // ---- range side
  bool split = false;
  uint64_t* cur = nullptr;
  uint32_t cur_n = 0;
  alignas(128) std::atomic<uint64_t> produced{0};       // written by the range side
  alignas(128) std::atomic<uint64_t> consumed{0};       // written by the low side
  alignas(128) std::atomic<bool> closing{false};
  std::atomic<bool> failed{false};                      // the low thread ran out of memory (it keeps draining the ring)
  uint64_t dbg_full_spins = 0;                          // range side: polls while the ring was full (CJS_DEBUG)
  alignas(128) uint64_t dbg_empty_spins = 0;            // low side: polls while the ring was empty
  int dbg_low_cpu[2] = {-1, -1};
  int near_cpu = -1;                                    // low side: the cpu the range side was last seen on
  cpu_set_t allowed;                                    // the caller's affinity when the split started
  bool have_allowed = false;
  explicit HostCoder(std::vector<uint8_t>& o) : out(o) { len = o.size(); reserve(4096); }
  ~HostCoder() { stop_thread(); }
  void reserve(size_t extra) {                          // room for `extra` more bytes (a coder step emits at most 3)
    if (len + extra > out.size()) out.resize(std::max(out.size() * 2, len + extra));
    data = out.data();
  }
  void reserve_steps(size_t nsteps) { if (!split) reserve(3 * nsteps + help + 4096); }      // (split mode: the low thread reserves per buffer)
  inline void emit(uint8_t b) { data[len++] = b; }
  void start(int c, uint32_t initlen) { low = 0; range = 0x80000000u; buffer = c; help = 0; bytecount = initlen; }
  // one byte shift of the low chain: what leaves (or stays pending) is decided by low alone
  inline void shift_low() {
    if (low < (0xFFu << 23)) { emit((uint8_t)buffer); for (; help; help--) emit(0xFF); buffer = (low >> 23) & 0xFF; }
    else if (low & 0x80000000u) { emit((uint8_t)(buffer + 1)); for (; help; help--) emit(0x00); buffer = (low >> 23) & 0xFF; }
    else help++;
    low = (low << 8) & 0x7FFFFFFFu; bytecount++;
  }
  inline void normalize() {
    while (range <= 0x00800000u) { shift_low(); range <<= 8; }
  }
  // normalize() for the step loop: whether a byte leaves is a coin flip per step, so the usual case (at most one
  // byte, no pending carry bytes, no carry) is written without a branch: unconditional store, conditional advance
  inline void normalize_step() {
    const uint32_t need = range <= 0x00800000u;
    if (__builtin_expect((range <= 0x00008000u) | (need & (uint32_t)((low >= (0xFFu << 23)) | (help != 0))), 0)) { normalize(); return; }
    data[len] = (uint8_t)buffer;
    len += need;
    buffer = need ? (int)((low >> 23) & 0xFF) : buffer;
    low = need ? (low << 8) & 0x7FFFFFFFu : low;
    range = need ? range << 8 : range;
    bytecount += need;
  }
  // ---- split mode, range side (calling thread)
  void start_split() {
    for (auto& r : ring) r.assign(RCAP, 0ull);
    cur = ring[0].data(); cur_n = 0; split = true;
    have_allowed = sched_getaffinity(0, sizeof allowed, &allowed) == 0;
    near_cpu = sched_getcpu();
    low_thread = std::thread([this] { low_loop(); });
    keep_off_my_core(low_thread);
  }
  // Where the second thread runs decides what the split is worth (2-socket EPYC host of the MI355X box, 100 MB): on another core
  // of the caller's L3 domain 169-200 ms of coding, elsewhere on the caller's socket 200-225 ms, on the OTHER socket 310-360 ms
  // (every ring line the range side writes was last read over there: slower than one thread, 308 ms), on the caller's own core
  // (its second hardware thread) ~320 ms.  So the low thread's affinity is set to the CPUs that share the caller's L3 minus the
  // caller's core; if that cannot be read, to the caller's package minus its core; if that cannot be read either, nothing is
  // changed.  The caller's own affinity is never touched.  CJS_BWTC_NO_AFFINITY=1 switches this off.
  static bool read_cpu_list(const char* fmt, int cpu, cpu_set_t* set) {
    char path[128];
    snprintf(path, sizeof path, fmt, cpu);
    FILE* f = fopen(path, "r");
    if (!f) return false;
    char line[512] = {0};
    const bool ok = fgets(line, sizeof line, f) != nullptr;
    fclose(f);
    if (!ok) return false;
    CPU_ZERO(set);
    int n = 0;
    for (char* p = line; *p;) {                          // "a,b-c,d" lists
      char* e = nullptr;
      const long a = strtol(p, &e, 10);
      if (e == p) break;
      long b2 = a;
      if (*e == '-') { p = e + 1; b2 = strtol(p, &e, 10); }
      for (long c = a; c <= b2 && c < CPU_SETSIZE; c++) if (c >= 0) { CPU_SET((int)c, set); n++; }
      if (*e != ',') break;
      p = e + 1;
    }
    return n > 0;
  }
  void keep_off_my_core(std::thread& t) { place_near(t.native_handle(), sched_getcpu()); }
  void place_near(pthread_t th, int cpu) {        // th: the low thread; cpu: where the range side runs
    static const bool off = getenv("CJS_BWTC_NO_AFFINITY") != nullptr;
    if (off || cpu < 0) return;
    cpu_set_t mine, near;
    if (!read_cpu_list("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", cpu, &mine)) return;
    if (!have_allowed) return;
    const char* domains[2] = {"/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", "/sys/devices/system/cpu/cpu%d/topology/package_cpus_list"};
    for (const char* d : domains) {
      if (!read_cpu_list(d, cpu, &near)) continue;
      cpu_set_t want;
      CPU_ZERO(&want);
      int left = 0;
      for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &near) && CPU_ISSET(c, &allowed) && !CPU_ISSET(c, &mine)) { CPU_SET(c, &want); left++; }
      if (left > 0) { (void)pthread_setaffinity_np(th, sizeof want, &want); return; }
    }
  }
  inline uint32_t shifts_needed() {                     // (zero or one shift is a coin flip per step: no branch for it)
    if (__builtin_expect(range <= 0x00008000u, 0)) { uint32_t k = 0; while (range <= 0x00800000u) { range <<= 8; k++; } return k; }
    const uint32_t k = range <= 0x00800000u;
    range <<= (k << 3);
    return k;
  }
  inline void push(uint32_t add, uint32_t k) {
    cur[cur_n++] = (uint64_t)add | ((uint64_t)k << 32);
    if (cur_n == RCAP) hand_over();
  }
  void hand_over() {                                    // the full (or last) buffer goes to the low thread
    const uint64_t p = produced.load(std::memory_order_relaxed);
    ring_n[p % RING] = cur_n;
    ring_cpu[p % RING] = sched_getcpu();                 // (the low thread follows this thread when the scheduler moves it)
    produced.store(p + 1, std::memory_order_release);
    while (p + 1 - consumed.load(std::memory_order_acquire) >= RING) { __builtin_ia32_pause(); dbg_full_spins++; }
    cur = ring[(p + 1) % RING].data(); cur_n = 0;
  }
  // ---- split mode, low side (its own thread): replays (shifts, addend) on low
  void low_loop() {
    dbg_low_cpu[0] = sched_getcpu();
    for (;;) {
      dbg_low_cpu[1] = sched_getcpu();
      const uint64_t c = consumed.load(std::memory_order_relaxed);
      uint32_t spins = 0;
      while (produced.load(std::memory_order_acquire) == c) {
        if (closing.load(std::memory_order_acquire) && produced.load(std::memory_order_acquire) == c) return;
        __builtin_ia32_pause();                          // (a polling loop without it starves the other hardware thread of the core)
        dbg_empty_spins++;
        if (++spins > 4096) { std::this_thread::yield(); spins = 0; }
      }
      const uint64_t* rec = ring[c % RING].data();
      uint32_t n = ring_n[c % RING];
      if (ring_cpu[c % RING] != near_cpu) { near_cpu = ring_cpu[c % RING]; if (c) place_near(pthread_self(), near_cpu); }      // the range side moved
      if (!failed.load(std::memory_order_relaxed)) {
        try { reserve(4 * (size_t)n + help + 64); } catch (...) { failed.store(true, std::memory_order_relaxed); }
      }
      if (failed.load(std::memory_order_relaxed)) n = 0;
      for (uint32_t i = 0; i < n; i++) {
        const uint64_t e = rec[i];
        uint32_t k = (uint32_t)(e >> 32);
        // the usual case without a branch, as in normalize_step(): at most one shift, no pending bytes, no carry
        const uint32_t need = k;
        if (__builtin_expect((k > 1u) | ((k != 0u) & (uint32_t)((low >= (0xFFu << 23)) | (help != 0))), 0)) { for (; k; k--) shift_low(); }
        else {
          data[len] = (uint8_t)buffer;
          len += need;
          buffer = need ? (int)((low >> 23) & 0xFF) : buffer;
          low = need ? (low << 8) & 0x7FFFFFFFu : low;
          bytecount += need;
        }
        low += (uint32_t)e;
      }
      consumed.store(c + 1, std::memory_order_release);
    }
  }
  void stop_thread() {
    if (!low_thread.joinable()) return;
    if (cur_n) hand_over();
    closing.store(true, std::memory_order_release);
    low_thread.join();
    split = false;
  }
  inline void freq(uint32_t sy, uint32_t lt, uint32_t tot) {
    if (split) { const uint32_t k = shifts_needed(); const uint32_t r = range / tot, tmp = r * lt; range = (lt + sy < tot) ? r * sy : range - tmp; push(tmp, k); return; }
    normalize();
    const uint32_t r = range / tot, tmp = r * lt;
    low += tmp;
    if (lt + sy < tot) range = r * sy; else range -= tmp;
  }
  // same arithmetic with the division replaced by a multiply with a 64-bit reciprocal: the quotient sits on the serial
  // (low, range) chain, the reciprocal (looked up by tot, which comes from the step list) does not
  inline void freq_rcp(uint32_t sy, uint32_t lt, uint32_t tot, const uint64_t* rcp) {
    // rcp[tot] = floor(2^64 / tot) + 1: the high half of range * rcp is floor(range / tot) exactly (range < 2^32, tot < 2^17:
    // range * (rcp*tot - 2^64) < 2^49 < 2^64), so no correction step sits on the chain
    if (split) {
      const uint32_t k = shifts_needed();
      const uint32_t r = (uint32_t)(((unsigned __int128)range * rcp[tot]) >> 64), tmp = r * lt;
      range = (lt + sy < tot) ? r * sy : range - tmp;
      push(tmp, k);
      return;
    }
    normalize_step();
    const uint32_t r = (uint32_t)(((unsigned __int128)range * rcp[tot]) >> 64);
    const uint32_t tmp = r * lt;
    low += tmp;
    range = (lt + sy < tot) ? r * sy : range - tmp;
  }
  inline void shift(uint32_t sy, uint32_t lt, int sh) {
    if (split) { const uint32_t k = shifts_needed(); const uint32_t r = range >> sh, tmp = r * lt; range = ((lt + sy) >> sh) ? range - tmp : r * sy; push(tmp, k); return; }
    normalize();
    const uint32_t r = range >> sh, tmp = r * lt;
    low += tmp;
    if ((lt + sy) >> sh) range -= tmp; else range = r * sy;
  }
  void finish() {
    if (split) { const uint32_t k = shifts_needed(); push(0u, k); stop_thread(); }      // the closing normalisation as a record; then the low side is ours again
    reserve(help + 16);
    normalize();
    bytecount += 5;
    uint32_t tmp = low >> 23;
    if ((low & 0x7FFFFFu) >= ((bytecount & 0xFFFFFFu) >> 1)) tmp++;
    if (tmp > 0xFF) { emit((uint8_t)(buffer + 1)); for (; help; help--) emit(0x00); }
    else { emit((uint8_t)buffer); for (; help; help--) emit(0xFF); }
    emit((uint8_t)(tmp & 0xFF));
    emit((uint8_t)((bytecount >> 16) & 0xFF)); emit((uint8_t)((bytecount >> 8) & 0xFF)); emit((uint8_t)(bytecount & 0xFF));
    out.resize(len);
  }
};
int fls32(uint32_t v) { int r = 0; while (v) { r++; v >>= 1; } return r; }
void nomodel(HostCoder& c, int bits, uint32_t sym) { for (int i = bits - 1; i >= 0; i--) c.shift(1, (sym >> i) & 1, 1); }   // :1281-1287
void logdist(HostCoder& c, int block_size, uint32_t d) {                                                                   // :1241-1253
  const int lgbits = fls32((uint32_t)(1 + fls32((uint32_t)block_size - 1)) - 1);
  if (d < 2) { nomodel(c, lgbits, d); return; }
  const int lg = fls32(d);
  nomodel(c, lgbits, (uint32_t)lg);
  nomodel(c, lg - 1, d & ((1u << (lg - 1)) - 1));
}

}  // namespace

// One batch of consecutive blocks: BWT -> MTF/RLE2 -> model evaluation on one GPU (worker thread), leaving the per-block
// step lists in HBM until the coder has consumed them.  Blocks are dealt to the device slots in contiguous ranges
// (cjs_opts.n_devices / CJS_DEVICES; SURVEY §8e "BWTC compress": replicable stages x N, serial tail x 1); the first
// batch of the job is small so that the coder starts early, and at most two batches per slot are alive.
namespace {

constexpr uint32_t BWTC_FIRST_BATCH = 8, BWTC_MAX_BATCH = 512;

struct BwtcBatch {
  int slot = 0, device = 0, seq = 0;          // seq: order of the batch inside its slot
  uint32_t first = 0, count = 0;
  int rc = 0;
  bool done = false;
  Arena arena; BwtWork bw;
  hipStream_t s = nullptr, cs = nullptr;      // work stream, copy stream (step lists -> pinned host buffers)
  hipEvent_t cev[2] = {nullptr, nullptr};
  uint64_t* d_steps = nullptr; size_t step_stride = 0;
  std::vector<uint32_t> pidx, asz, nsteps; std::vector<uint8_t> alist;
  double ms = 0;
  void release() {
    if (hipSetDevice(device) != hipSuccess) return;
    if (s) (void)hipStreamSynchronize(s);
    if (cs) (void)hipStreamSynchronize(cs);
    for (int q = 0; q < 2; q++) if (cev[q]) { (void)hipEventDestroy(cev[q]); cev[q] = nullptr; }
    if (cs) { (void)hipStreamDestroy(cs); cs = nullptr; }
    if (s) { (void)hipStreamDestroy(s); s = nullptr; }
    bw.release_host();
    arena.destroy();
  }
};

struct BwtcJob {
  const uint8_t* in = nullptr; size_t n = 0;
  uint32_t bs = 0, nb = 0, n_last = 0; bool fast = false;
  std::vector<BwtcBatch> batches;
  std::mutex mu; std::condition_variable cv;
  std::vector<int> live, next_seq;            // per slot: batches alive, next batch allowed to start
  bool abort = false;
};

void bwtc_batch_body(BwtcJob* J, BwtcBatch* B);
// nothing may leave a worker thread (std::terminate): an exception becomes the batch's return code and stops the job
void bwtc_batch_worker(BwtcJob* J, BwtcBatch* B) {
  int rc = 0;
  try { bwtc_batch_body(J, B); return; }
  catch (const std::bad_alloc&) { rc = CJS_E_OUT_OF_MEMORY; }
  catch (...) { rc = CJS_E_HIP; }
  std::lock_guard<std::mutex> lk(J->mu);
  B->rc = rc; B->done = true; J->abort = true;
  J->cv.notify_all();
}
void bwtc_batch_body(BwtcJob* J, BwtcBatch* B) {
  {
    std::unique_lock<std::mutex> lk(J->mu);
    J->cv.wait(lk, [&] { return J->abort || (J->next_seq[B->slot] == B->seq && J->live[B->slot] < 2); });
    if (J->abort) { B->rc = CJS_E_HIP; B->done = true; J->next_seq[B->slot]++; J->cv.notify_all(); return; }
    J->live[B->slot]++; J->next_seq[B->slot]++;
    J->cv.notify_all();
  }
  const auto T0 = std::chrono::steady_clock::now();
  int rc = 0;
  const uint32_t bs = J->bs, cnt = B->count;
  const bool last = B->first + cnt == J->nb;
  const uint32_t n_last = last ? J->n_last : bs;
  const size_t elems = (size_t)cnt * bs, nbytes = (size_t)(cnt - 1) * bs + n_last;
  const size_t a_stride = MtfWork::a_stride_for(bs);
  B->step_stride = 2 * a_stride;
  MtfWork mw;
  uint8_t *d_T = nullptr, *d_U = nullptr; uint32_t *d_pidx = nullptr, *d_len = nullptr, *d_nsteps = nullptr;
  if (hipSetDevice(B->device) != hipSuccess) rc = CJS_E_HIP;
  if (!rc) rc = B->arena.init_pooled(BwtWork::bytes_needed(elems) + MtfWork::bytes_needed(cnt, bs) + 2 * (elems + 512) + 8 * (size_t)cnt * B->step_stride + 16 * (size_t)cnt + 65536);
  if (!rc) rc = B->bw.carve(B->arena, elems);
  if (!rc) rc = mw.carve(B->arena, cnt, bs);
  if (!rc) {
    d_T = B->arena.take<uint8_t>(elems); d_U = B->arena.take<uint8_t>(elems);
    d_pidx = B->arena.take<uint32_t>(cnt); d_len = B->arena.take<uint32_t>(cnt); d_nsteps = B->arena.take<uint32_t>(cnt);
    B->d_steps = B->arena.take<uint64_t>((size_t)cnt * B->step_stride);
    if (!B->d_steps) rc = CJS_E_OUT_OF_MEMORY;
  }
  if (!rc && B->slot == 0 && B->seq == 0 && J->batches.size() > 1) {   // the batch the coder is waiting for goes ahead of the others' kernels
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || hipStreamCreateWithPriority(&B->s, hipStreamDefault, greatest) != hipSuccess) B->s = nullptr;
  }
  if (!rc && ((!B->s && hipStreamCreate(&B->s) != hipSuccess) || hipStreamCreate(&B->cs) != hipSuccess ||
              hipEventCreateWithFlags(&B->cev[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&B->cev[1], hipEventDisableTiming) != hipSuccess)) rc = CJS_E_HIP;
  std::vector<uint32_t> lens(cnt, bs); lens[cnt - 1] = n_last;
  hipStream_t s = B->s;
  if (!rc && hipMemcpyAsync(d_T, J->in + (size_t)B->first * bs, nbytes, hipMemcpyHostToDevice, s) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipMemcpyAsync(d_len, lens.data(), 4 * (size_t)cnt, hipMemcpyHostToDevice, s) != hipSuccess) rc = CJS_E_HIP;
  if (!rc) rc = bwt_run(s, B->bw, d_T, cnt, bs, n_last, false, d_U, d_pidx, nullptr);
  if (!rc) rc = mtf_run(s, mw, d_U, cnt, d_len);
  if (!rc) {
    if (J->fast) hipLaunchKernelGGL(bwtc_defsum, dim3(cnt), dim3(64), 0, s, mw.b, B->d_steps, B->step_stride, d_nsteps);
    else hipLaunchKernelGGL(bwtc_fenwick_par, dim3(cnt), dim3(64 * FP_WAVES), 0, s, mw.b, B->d_steps, B->step_stride, d_nsteps);
    if (hipGetLastError() != hipSuccess) rc = CJS_E_HIP;
  }
  B->pidx.resize(cnt); B->asz.resize(cnt); B->nsteps.resize(cnt); B->alist.resize((size_t)cnt * 256);
  if (!rc && hipMemcpyAsync(B->pidx.data(), d_pidx, 4 * (size_t)cnt, hipMemcpyDeviceToHost, s) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipMemcpyAsync(B->asz.data(), mw.b.asz, 4 * (size_t)cnt, hipMemcpyDeviceToHost, s) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipMemcpyAsync(B->nsteps.data(), d_nsteps, 4 * (size_t)cnt, hipMemcpyDeviceToHost, s) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipMemcpyAsync(B->alist.data(), mw.b.alist, (size_t)cnt * 256, hipMemcpyDeviceToHost, s) != hipSuccess) rc = CJS_E_HIP;
  if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = CJS_E_HIP;
  B->ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count();
  std::lock_guard<std::mutex> lk(J->mu);
  B->rc = rc; B->done = true;
  if (rc) J->abort = true;
  J->cv.notify_all();
}

}  // namespace

extern "C" int cjs_bwtc_compress(const uint8_t* in, size_t n, int level, uint8_t** out, size_t* out_n, const cjs_opts* opts) {
  if (!out || !out_n) return CJS_E_INVALID_ARG;
  *out = nullptr; *out_n = 0;
  clear_detail();
  CJS_GUARD_BEGIN
  CJS_TRY(select_device(opts));
  if (level < 1 || level > 9) level = 9;                             // J/BWTC_joined_.js:1702-1705
  int ndev = 0, dev0 = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || hipGetDevice(&dev0) != hipSuccess) return CJS_E_NO_DEVICE;
  BwtcJob J;
  J.in = in; J.n = n; J.fast = level <= 5; J.bs = (uint32_t)level * 100000u;
  J.nb = (uint32_t)((n + J.bs - 1) / J.bs);
  J.n_last = J.nb ? (uint32_t)(n - (size_t)(J.nb - 1) * J.bs) : 0;
  const uint32_t bs = J.bs, nb = J.nb;
  std::vector<uint8_t> o;
  o.reserve(n / 3 + 64);
  o.push_back('b'); o.push_back('w'); o.push_back('t'); o.push_back('c');
  uint8_t vb[12]; int nv = 0;                                        // writeUnsignedNumber(size+1) :605-620
  const bool size_unknown = opts && opts->struct_size >= sizeof(cjs_opts) && (opts->flags & CJS_FLAG_SIZE_UNKNOWN);
  { uint64_t v = size_unknown ? 0 : (uint64_t)n + 1; do { vb[nv++] = (uint8_t)(v & 0x7F); v >>= 7; } while (v); vb[0] |= 0x80; }   // W1: a stream without .size gives varint(0)
  for (int i = nv - 1; i >= 1; i--) o.push_back(vb[i]);
  HostCoder coder(o);
  coder.start(vb[0], 1);                                             // :1700 (the last varint byte is the coder's first byte)
  static const bool env_split_coder = getenv("CJS_BWTC_SPLIT_CODER") == nullptr || atoi(getenv("CJS_BWTC_SPLIT_CODER")) != 0;
  const int dbg_cpu0 = sched_getcpu();
  if (env_split_coder && nb) coder.start_split();                    // range chain here, low chain on a second host thread
  coder.shift(1, (uint32_t)level, 8);                                // encodeByte(level) :1706
  int rc = 0;
  const auto T0 = std::chrono::steady_clock::now();
  auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
  double ms_coder = 0, ms_stall = 0, ms_first = 0;
  if (nb) {
    // ---- plan: contiguous block ranges per device slot, cut into batches
    uint32_t nslots = (opts && opts->struct_size >= sizeof(cjs_opts)) ? opts->n_devices : 0;
    if (const char* e = getenv("CJS_DEVICES")) nslots = (uint32_t)atoi(e);
    if (nslots < 1) nslots = 1;
    if (nslots > 64) nslots = 64;
    if (nslots > nb) nslots = nb;
    static const uint32_t first_batch = getenv("CJS_BWTC_FIRST_BATCH") ? (uint32_t)atoi(getenv("CJS_BWTC_FIRST_BATCH")) : BWTC_FIRST_BATCH;
    const uint32_t share = (nb + nslots - 1) / nslots;
    J.live.assign(nslots, 0); J.next_seq.assign(nslots, 0);
    for (uint32_t sl = 0; sl < nslots; sl++) {
      uint32_t k = std::min<uint32_t>(sl * share, nb);
      const uint32_t end = std::min<uint32_t>(k + share, nb);
      int seq = 0;
      while (k < end) {
        uint32_t c = std::min<uint32_t>(end - k, BWTC_MAX_BATCH);
        if (sl == 0 && seq == 0 && first_batch && end - k > 2 * first_batch) c = first_batch;      // the coder starts on this one (doubling batches behind it: no gain, 232-243 ms either way)
        J.batches.emplace_back();
        BwtcBatch& B = J.batches.back();
        B.slot = (int)sl; B.device = nslots == 1 ? dev0 : (int)(sl % (uint32_t)ndev); B.seq = seq++; B.first = k; B.count = c;
        k += c;
      }
    }
    // on every path out of this scope (an exception of the coder side included: out.resize in reserve_steps) the workers are
    // stopped and joined, the batches give their device memory back and the pinned step buffers are freed
    struct Workers {
      BwtcJob& J; int dev0; std::vector<std::thread> th; uint64_t* h_buf[2] = {nullptr, nullptr};
      ~Workers() {
        bool running = false;
        for (auto& t : th) running |= t.joinable();
        if (running) { { std::lock_guard<std::mutex> lk(J.mu); J.abort = true; } J.cv.notify_all(); for (auto& t : th) if (t.joinable()) t.join(); }
        for (auto& B : J.batches) B.release();
        (void)hipSetDevice(dev0);
        for (int q = 0; q < 2; q++) if (h_buf[q]) (void)hipHostFree(h_buf[q]);
      }
    } wk{J, dev0};
    std::vector<std::thread>& workers = wk.th;
    uint64_t** h_buf = wk.h_buf;
    for (auto& B : J.batches) workers.emplace_back(bwtc_batch_worker, &J, &B);
    // reciprocals floor(2^64 / tot) + 1 for every total a step can carry (17 bits); tot < 2 keeps the division
    static std::vector<uint64_t> rcp;
    static std::once_flag rcp_once;
    std::call_once(rcp_once, [] { rcp.assign(1u << 17, 0ull); for (uint32_t t = 2; t < (1u << 17); t++) rcp[t] = (uint64_t)(((unsigned __int128)1 << 64) / t) + 1; });
    // the steps of block k+1 travel (pinned buffer, copy stream of its batch) while block k goes through the coder
    const size_t step_stride = 2 * MtfWork::a_stride_for(bs);
    if (hipHostMalloc((void**)&h_buf[0], 8 * step_stride, hipHostMallocPortable) != hipSuccess || hipHostMalloc((void**)&h_buf[1], 8 * step_stride, hipHostMallocPortable) != hipSuccess) rc = CJS_E_HIP;
    size_t bi_of_next = 0;                                           // batch that holds the next block to fetch
    hipEvent_t pending[2] = {nullptr, nullptr}; int pending_dev[2] = {0, 0};
    auto wait_batch = [&](size_t bi) -> int {                        // blocks until the batch's GPU work is done
      const auto Tw = std::chrono::steady_clock::now();
      std::unique_lock<std::mutex> lk(J.mu);
      J.cv.wait(lk, [&] { return J.batches[bi].done; });
      ms_stall += since(Tw);
      return J.batches[bi].rc;
    };
    auto fetch = [&](uint32_t k) -> int {
      while (J.batches[bi_of_next].first + J.batches[bi_of_next].count <= k) bi_of_next++;
      BwtcBatch& B = J.batches[bi_of_next];
      CJS_TRY(wait_batch(bi_of_next));
      const uint32_t r = k - B.first;
      if (hipSetDevice(B.device) != hipSuccess) return (int)CJS_E_HIP;
      if (B.nsteps[r] && hipMemcpyAsync(h_buf[k & 1], B.d_steps + (size_t)r * B.step_stride, 8 * (size_t)B.nsteps[r], hipMemcpyDeviceToHost, B.cs) != hipSuccess) return (int)CJS_E_HIP;
      if (hipEventRecord(B.cev[k & 1], B.cs) != hipSuccess) return (int)CJS_E_HIP;
      pending[k & 1] = B.cev[k & 1]; pending_dev[k & 1] = B.device;
      return 0;
    };
    auto retire = [&](size_t bi) {                                   // the coder is through with the batch: give its memory back
      BwtcBatch& B = J.batches[bi];
      B.release();
      std::lock_guard<std::mutex> lk(J.mu);
      J.live[B.slot]--;
      J.cv.notify_all();
    };
    if (!rc) rc = fetch(0);
    ms_first = since(T0);
    size_t bi_cur = 0;
    for (uint32_t k = 0; k < nb && !rc; k++) {
      while (J.batches[bi_cur].first + J.batches[bi_cur].count <= k) { retire(bi_cur); bi_cur++; }
      const BwtcBatch& B = J.batches[bi_cur];
      const uint32_t r = k - B.first;
      if (k + 1 < nb) rc = fetch(k + 1);
      if (rc) break;
      const auto Tc = std::chrono::steady_clock::now();
      coder.reserve_steps(B.nsteps[r]);                              // everything this block can emit
      const uint32_t length = k + 1 == nb ? J.n_last : bs;
      if (length == bs) coder.freq(1, 0, 3);                         // "full size block" :1734
      else { coder.freq(1, 1, 3); logdist(coder, (int)bs, length); } // "short block" :1737-1738
      logdist(coder, (int)bs, B.pidx[r]);                            // :1742
      uint16_t tree[512]; memset(tree, 0, sizeof tree);              // use-tree :1744-1765
      for (uint32_t i = 0; i < B.asz[r]; i++) tree[256 + B.alist[(size_t)r * 256 + i]] = 1;
      for (int i = 255; i > 0; i--) tree[i] = (uint16_t)(tree[2 * i] + tree[2 * i + 1]);
      tree[0] = 1;
      for (int i = 1; i < 512; i++) {
        const int parent = i >> 1, full = 1 << (9 - fls32((uint32_t)i));
        if (tree[parent] == 0 || tree[parent] == full * 2) continue;
        if (i >= 256) coder.shift(1, tree[i] ? 1 : 0, 1);
        else coder.freq(1, tree[i] == 0 ? 0u : tree[i] == full ? 2u : 1u, 3);
      }
      if (hipSetDevice(pending_dev[k & 1]) != hipSuccess || hipEventSynchronize(pending[k & 1]) != hipSuccess) { rc = CJS_E_HIP; break; }
      const uint64_t* h_steps = h_buf[k & 1];
      const uint64_t* rc_tab = rcp.data();
      const uint32_t ns = B.nsteps[r];
      for (uint32_t i = 0; i < ns; i++) {                             // the serial tail
        const uint64_t st = h_steps[i];
        const uint32_t sy = (uint32_t)(st & 0xFFFF), lt = (uint32_t)((st >> 16) & 0xFFFF), tot = (uint32_t)((st >> 32) & 0x1FFFF);
        if (st & STEP_SHIFT_FLAG) coder.shift(sy, lt, (int)tot);
        else if (tot >= 2) coder.freq_rcp(sy, lt, tot, rc_tab);
        else coder.freq(sy, lt, tot);
      }
      ms_coder += since(Tc);
    }
    if (rc) { std::lock_guard<std::mutex> lk(J.mu); J.abort = true; J.cv.notify_all(); }
    for (auto& t : workers) t.join();
    double ms_gpu_max = 0;
    for (auto& B : J.batches) { if (!rc && B.rc) rc = B.rc; ms_gpu_max = std::max(ms_gpu_max, B.ms); }
    if (getenv("CJS_DEBUG")) fprintf(stderr, "[cjs bwtc] %zu batch(es) on %u slot(s): first step list after %.1f ms, longest batch (workspace + H2D + BWT + MTF + model) %.1f ms, "
                                             "range coder over the step lists (host, serial) %.1f ms, coder waited for the GPU %.1f ms, total %.1f ms\n",
                                     J.batches.size(), (unsigned)J.live.size(), ms_first, ms_gpu_max, ms_coder, ms_stall, since(T0));
    cjs_stats* st = (opts && opts->struct_size >= sizeof(cjs_opts)) ? opts->stats : nullptr;
    if (st && !rc) {                                                 // BWTC meaning of the fields: see include/cjs_hip.h
      memset(st, 0, sizeof *st);
      st->ms_total = since(T0); st->ms_bwt = ms_gpu_max; st->ms_pack = ms_coder; st->ms_rle1 = ms_stall; st->ms_mtf = ms_first;
      st->blocks = nb; st->bytes_in = n;
    }
  }
  if (rc) return rc;
  coder.reserve_steps(64);
  coder.freq(1, 2, 3);                                               // "no more blocks" :1823
  const int dbg_cpu1 = sched_getcpu();
  coder.finish();
  if (coder.failed.load()) return CJS_E_OUT_OF_MEMORY;
  if (getenv("CJS_DEBUG")) fprintf(stderr, "[cjs bwtc] split coder: range side on cpu %d -> %d, low side on cpu %d -> %d, polls with the ring full %llu, empty %llu\n", dbg_cpu0, dbg_cpu1,
                                   coder.dbg_low_cpu[0], coder.dbg_low_cpu[1], (unsigned long long)coder.dbg_full_spins, (unsigned long long)coder.dbg_empty_spins);
  uint8_t* host = (uint8_t*)malloc(o.size() ? o.size() : 1);
  if (!host) return CJS_E_OUT_OF_MEMORY;
  memcpy(host, o.data(), o.size());
  *out = host; *out_n = o.size();
  if (opts && opts->struct_size >= sizeof(cjs_opts) && opts->stats) opts->stats->bytes_out = o.size();
  return 0;
  CJS_GUARD_END(CJS_E_OUT_OF_MEMORY, CJS_E_HIP)
}


// ---------------------------------------------------------------- BWTC.decompressFile (J/BWTC_joined_.js:1827-1920)
// Range decoding and the adaptive model are one serial chain over the whole file (every decoded symbol feeds the
// model that decodes the next one), so that part runs on one host thread; the inverse BWT of all blocks
// (BWT.unbwtransform, n dependent gathers per block in the reference) runs on the GPU (decode.hip).
//
// The host side keeps the stream format's arithmetic (it IS the format) but not the reference's data structures: the
// adaptive models are flat per-symbol counters laid out in CODING ORDER with 16-entry bucket sums, searched linearly
// (2 short scans per symbol), instead of the reference's implicit binary tree walked and updated level by level.
namespace {

// Byte source + the decoder half of the carry-less range coder (interval state as RangeCoder's decode side, :159-238).
class RangeDecoder {
 public:
  RangeDecoder(const uint8_t* p, size_t n, size_t at) : in_(p), n_(n), pos_(at) {}
  void begin() { last_ = next_byte(); low_ = (uint32_t)last_ >> 1; range_ = 1u << 7; }           // decodeStart(skipInitialRead)
  // cumulative-frequency target under total `tot` / under 2^sh; the matching commit() must follow
  uint32_t target(uint32_t tot) { refill(); unit_ = range_ / tot; const uint32_t t = unit_ ? low_ / unit_ : 0; return t >= tot ? tot - 1 : t; }
  uint32_t target_pow2(int sh) { refill(); unit_ = range_ >> sh; const uint32_t t = unit_ ? low_ / unit_ : 0; return (t >> sh) ? (1u << sh) - 1 : t; }
  void commit(uint32_t sy, uint32_t lt, uint32_t tot) { const uint32_t base = unit_ * lt; low_ -= base; if (lt + sy < tot) range_ = unit_ * sy; else range_ -= base; }
  uint32_t bit() { const uint32_t t = target_pow2(1); commit(1, t, 2); return t; }
  uint32_t bits(int k) { uint32_t r = 0; while (k-- > 0) r = (r << 1) | bit(); return r; }         // NoModel (:1274-1287)
  uint32_t log_distance(int block_size) {                                                           // LogDistanceModel.decode (:1254-1261)
    const int lgbits = fls32((uint32_t)(1 + fls32((uint32_t)block_size - 1)) - 1);
    const uint32_t lg = bits(lgbits);
    return lg < 2 ? lg : (1u << (lg - 1)) + bits((int)lg - 1);
  }
  bool exhausted() const { return past_end_ > 8; }       // a well-formed stream never reads more than a few bytes past its end
  size_t consumed() const { return pos_; }
 private:
  int32_t next_byte() { if (pos_ < n_) return in_[pos_++]; past_end_++; return -1; }
  void refill() {
    while (range_ <= 0x00800000u) {
      low_ = (low_ << 8) | (((uint32_t)last_ << 7) & 0xFF);
      last_ = next_byte();
      low_ |= (uint32_t)last_ >> 1;
      range_ <<= 8;
    }
  }
  const uint8_t* in_; size_t n_, pos_;
  uint32_t low_ = 0, range_ = 0, unit_ = 0, past_end_ = 0; int32_t last_ = 0;
};

// FenwickModel, decode side (:1572-1661), as flat counters.  The reference keeps (count << 16 | unseen marker) in an
// implicit binary tree over num_syms leaves (num_syms not a power of two): the cumulative order of the symbols is the
// left-to-right order of the leaves = the deepest level first.  Here: slot o of the coding order holds the count and the
// unseen marker of symbol sym_[o]; 16-slot bucket sums make the cumulative search two short linear scans.
class RankModelDecoder {
 public:
  RankModelDecoder(RangeDecoder& d, int size) : d_(d), n_(size + 1) {
    int deep = 1; while (deep * 2 <= 2 * n_ - 1) deep *= 2;          // first heap position of the deepest level
    int o = 0;
    for (int leaf = deep; leaf < 2 * n_; leaf++) sym_[o++] = (uint16_t)(leaf - n_);
    for (int leaf = n_; leaf < deep; leaf++) sym_[o++] = (uint16_t)(leaf - n_);
    for (o = 0; o < n_; o++) { const bool esc = sym_[o] == n_ - 1; cnt_[o] = esc ? 0x100u : 0u; unseen_[o] = esc ? 0 : 1; if (esc) esc_slot_ = o; }
    resum();
  }
  int decode() {                                                    // -1: corrupt stream
    int o = take(cnt_, bcnt_, cnt_total_);
    if (o < 0) return -1;
    if (o == esc_slot_) {
      // the escape symbol: the novel symbol follows in the distribution of the unseen markers (:1590-1600); once the
      // last unseen symbol is about to go the escape symbol disappears with it (:1648)
      cnt_[o] += 0x100u; bcnt_[o >> 4] += 0x100u; cnt_total_ += 0x100u;
      if (unseen_total_ == 1) { cnt_total_ -= cnt_[o]; bcnt_[o >> 4] -= cnt_[o]; cnt_[o] = 0; }
      if (cnt_total_ >= 0xFF00u) halve();
      uint32_t ut = unseen_total_;
      if (ut == 0) return -1;
      o = take_unseen(ut);
      if (o < 0) return -1;
      unseen_[o] = 0; bun_[o >> 4]--; unseen_total_--;
    }
    cnt_[o] += 0x100u; bcnt_[o >> 4] += 0x100u; cnt_total_ += 0x100u;
    if (cnt_total_ >= 0xFF00u) halve();
    return sym_[o];
  }
 private:
  // slot whose cumulative count interval holds the coder's target; commits the step with the count BEFORE the update
  int take(const uint32_t* c, const uint32_t* bc, uint32_t tot) {
    if (tot == 0) return -1;
    const uint32_t t = d_.target(tot);
    uint32_t lt = 0; int b = 0;
    const int nbuck = (n_ + 15) >> 4;
    while (b < nbuck - 1 && lt + bc[b] <= t) lt += bc[b++];
    int o = b << 4;
    while (o < n_ - 1 && lt + c[o] <= t) lt += c[o++];
    d_.commit(c[o], lt, tot);
    return o;
  }
  int take_unseen(uint32_t tot) {
    const uint32_t t = d_.target(tot);
    uint32_t lt = 0; int b = 0;
    const int nbuck = (n_ + 15) >> 4;
    while (b < nbuck - 1 && lt + bun_[b] <= t) lt += bun_[b++];
    int o = b << 4;
    while (o < n_ - 1 && lt + unseen_[o] <= t) lt += unseen_[o++];
    d_.commit(unseen_[o], lt, tot);
    return unseen_[o] ? o : -1;
  }
  void halve() {                                                    // rescale (:1623-1646)
    bool any_unseen = false;
    for (int o = 0; o < n_; o++) {
      if (o == esc_slot_) continue;
      if (unseen_[o]) { any_unseen = true; continue; }
      cnt_[o] >>= 1;
      if (cnt_[o] == 0) { unseen_[o] = 1; any_unseen = true; }      // a count that halves to nothing makes the symbol novel again
    }
    uint32_t e = cnt_[esc_slot_] >> 1;
    cnt_[esc_slot_] = any_unseen ? (e ? e : 1u) : 0u;
    resum();
  }
  void resum() {
    const int nbuck = (n_ + 15) >> 4;
    cnt_total_ = unseen_total_ = 0;
    for (int b = 0; b < nbuck; b++) bcnt_[b] = bun_[b] = 0;
    for (int o = 0; o < n_; o++) { bcnt_[o >> 4] += cnt_[o]; bun_[o >> 4] += unseen_[o]; cnt_total_ += cnt_[o]; unseen_total_ += unseen_[o]; }
  }
  RangeDecoder& d_;
  int n_, esc_slot_ = 0;                                            // n_ = coded symbols incl. the escape symbol (the last one)
  uint16_t sym_[272];
  uint32_t cnt_[272], unseen_[272], bcnt_[17], bun_[17];
  uint32_t cnt_total_ = 0, unseen_total_ = 0;
};

// DefSumModel, decode side (:1327-1459): 8-bit total, counts folded in only every `quota_` symbols.  Lookup of the symbol
// from the 8-bit target is a direct 256-entry table rebuilt at every fold; the escape distribution is uniform over the
// symbols whose count is still zero.
class DeferredSumDecoder {
 public:
  DeferredSumDecoder(RangeDecoder& d, int size) : d_(d), esc_(size) {
    memset(cum_, 0, sizeof cum_); memset(pend_, 0, sizeof pend_);
    cum_[esc_ + 1] = 256;                                           // everything on the escape symbol at first
    for (int i = 0; i <= esc_; i++) zero_rank_[i] = (uint16_t)i;
    for (int i = 0; i < 256; i++) by_target_[i] = (uint16_t)esc_;
    for (int i = 0; i < 304; i++) by_zero_rank_[i] = (uint16_t)(i < esc_ ? i : 0);
  }
  int decode() {
    int sym = by_target_[d_.target_pow2(8)];
    d_.commit((uint32_t)cum_[sym + 1] - cum_[sym], cum_[sym], 256);
    note(sym);
    if (sym != esc_) return sym;
    const uint32_t nzero = zero_rank_[esc_];
    if (nzero == 0) return -1;
    sym = by_zero_rank_[d_.target(nzero)];
    d_.commit((uint32_t)zero_rank_[sym + 1] - zero_rank_[sym], zero_rank_[sym], nzero);
    note(sym);
    return sym;
  }
 private:
  void note(int sym) {                                              // :1398-1429
    if (sym == esc_ && (pend_[sym] >= 40 || seen_ >= quota_ - 1)) return;
    pend_[sym]++; seen_++;
    if (seen_ < quota_) return;
    int total = 0, zeros = 0, odd = 0;
    for (int i = 0; i <= esc_; i++) {                               // halve the old share, add what arrived since
      const int c = ((cum_[i + 1] - cum_[i]) >> 1) + pend_[i];
      cum_[i] = (uint16_t)total; zero_rank_[i] = (uint16_t)zeros;
      if (c) { total += c; odd += c & 1; } else zeros++;
    }
    cum_[esc_ + 1] = (uint16_t)total;
    quota_ = 256 - (total - odd) / 2;
    memset(pend_, 0, sizeof pend_);
    pend_[esc_] = 1; seen_ = 1;
    int t = 0, z = 0;
    for (int i = 0; i <= esc_; i++) {
      for (; t < cum_[i + 1]; t++) by_target_[t] = (uint16_t)i;
      const int ze = i + 1 <= esc_ ? zero_rank_[i + 1] : 0;         // the reference's escape[] has esc_+1 entries
      for (; z < ze; z++) by_zero_rank_[z] = (uint16_t)i;
    }
  }
  RangeDecoder& d_;
  int esc_;                                                         // number of ordinary symbols = index of the escape symbol
  uint16_t cum_[304], zero_rank_[304], pend_[304], by_target_[256], by_zero_rank_[304];
  int seen_ = 0, quota_ = 128;
};

struct BwtcBlocks {                     // output of the serial stage: the BWT columns of all blocks, back to back
  std::vector<uint8_t> cols;
  std::vector<uint32_t> lens, pidx;
  int level = 0;
};

// header, per-block flag / length / primary index / use-tree, symbol decode, RLE2 + MTF inverse (:1827-1913)
int bwtc_entropy_decode(const uint8_t* in, size_t n, BwtcBlocks& B) {
  if (n < 4 || in[0] != 'b' || in[1] != 'w' || in[2] != 't' || in[3] != 'c') return CJS_E_BAD_MAGIC;     // :559-565
  size_t p = 4;
  for (;;) { if (p >= n) return CJS_E_DATA_ERROR; if (in[p++] & 0x80) break; }                          // readUnsignedNumber :621-633
  RangeDecoder d(in, n, p);
  d.begin();
  const uint32_t lv = d.target_pow2(8); d.commit(1, lv, 256);                                          // decodeByte :1830
  if (lv < 1 || lv > 9) return CJS_E_DATA_ERROR;
  B.level = (int)lv;
  const bool fast = lv <= 5;
  const uint32_t bs = lv * 100000u;
  static const bool dbg = getenv("CJS_DEBUG") != nullptr;
  for (;;) {
    const uint32_t flag = d.target(3); d.commit(1, flag, 3);
    uint32_t length;
    if (flag == 0) length = bs;
    else if (flag == 1) { length = d.log_distance((int)bs); if (length > bs) return CJS_E_DATA_ERROR; }
    else break;
    const uint32_t pi = d.log_distance((int)bs);
    uint16_t tree[512]; memset(tree, 0, sizeof tree); tree[0] = 1;                                     // use-tree :1859-1874
    for (int i = 1; i < 512; i++) {
      const int parent = i >> 1, full = 1 << (9 - fls32((uint32_t)i));
      if (tree[parent] == 0 || tree[parent] == full * 2) tree[i] = tree[parent] >> 1;
      else if (i >= 256) tree[i] = (uint16_t)d.bit();
      else { const uint32_t v = d.target(3); d.commit(1, v, 3); tree[i] = (uint16_t)(v == 2 ? (uint32_t)full : v); }
    }
    uint8_t order[256]; int asz = 0;
    for (int i = 0; i < 256; i++) if (tree[256 + i]) order[asz++] = (uint8_t)i;
    if (d.exhausted()) return CJS_E_DATA_ERROR;
    // the block is stored at its decoded length (a forged header cannot make the host reserve a full block per 13 bits)
    const size_t base = B.cols.size();
    B.cols.resize(base + length);
    uint8_t* b = B.cols.data() + base;
    uint64_t run = 1; uint32_t i = 0; bool bad = false;
    {
      RankModelDecoder* rm = fast ? nullptr : new RankModelDecoder(d, asz + 1);
      DeferredSumDecoder* dm = fast ? new DeferredSumDecoder(d, asz + 1) : nullptr;
      while (i < length) {                                                                             // :1888-1903
        const int c = fast ? dm->decode() : rm->decode();
        if (c < 0 || d.exhausted()) { bad = true; break; }
        if (c <= 1) {                                                // RUNA / RUNB: bijective base-2 digits of a zero run
          const uint64_t add = run << c;
          if (i + add > length) { bad = true; break; }
          memset(b + i, 0, (size_t)add); i += (uint32_t)add; run *= 2;
        } else { run = 1; if (c - 1 >= asz) { bad = true; break; } b[i++] = (uint8_t)(c - 1); }
      }
      delete rm; delete dm;
    }
    if (dbg) fprintf(stderr, "[cjs bwtc dec] block %zu: length %u pidx %u asz %d decoded %u bad %d inpos %zu/%zu\n", B.lens.size(), length, pi, asz, i, (int)bad, d.consumed(), n);
    if (bad || pi > length) return CJS_E_DATA_ERROR;
    for (i = 0; i < length; i++) {                                                                     // MTF decode :1905-1913
      const int j = b[i]; const uint8_t c = order[j];
      b[i] = c;
      if (j) { memmove(order + 1, order, (size_t)j); order[0] = c; }
    }
    if (length == 0) { B.cols.resize(base); continue; }             // an empty block contributes nothing (unbwtransform of 0 bytes)
    B.lens.push_back(length); B.pidx.push_back(pi);
  }
  return 0;
}

}  // namespace

// Stage-level entry point (host logic only, no device needed): the serial entropy stage of BWTC.decompressFile.
// cols receives the BWT columns of all non-empty blocks back to back (malloc'd, cjs_free); lens/pidx per block.
extern "C" long cjs_stage_bwtc_entropy_decode(const uint8_t* in, size_t n, uint8_t** cols, size_t* cols_n, uint32_t* lens, uint32_t* pidx, long cap,
                                              int* level) {
  if (!cols || !cols_n) return CJS_E_INVALID_ARG;
  *cols = nullptr; *cols_n = 0;
  CJS_GUARD_BEGIN
  BwtcBlocks B;
  const int rc = bwtc_entropy_decode(in, n, B);
  if (rc) return (long)rc;
  if (level) *level = B.level;
  uint8_t* c = (uint8_t*)malloc(B.cols.size() ? B.cols.size() : 1);
  if (!c) return (long)CJS_E_OUT_OF_MEMORY;
  if (!B.cols.empty()) memcpy(c, B.cols.data(), B.cols.size());
  for (size_t k = 0; k < B.lens.size() && (long)k < cap; k++) { if (lens) lens[k] = B.lens[k]; if (pidx) pidx[k] = B.pidx[k]; }
  *cols = c; *cols_n = B.cols.size();
  return (long)B.lens.size();
  CJS_GUARD_END((long)CJS_E_OUT_OF_MEMORY, (long)CJS_E_HIP)
}

extern "C" int cjs_bwtc_decompress(const uint8_t* in, size_t n, uint8_t** out, size_t* out_n, const cjs_opts* opts) {
  if (!out || !out_n) return CJS_E_INVALID_ARG;
  *out = nullptr; *out_n = 0;
  clear_detail();
  CJS_GUARD_BEGIN
  CJS_TRY(select_device(opts));
  BwtcBlocks B;
  CJS_TRY(bwtc_entropy_decode(in, n, B));
  const uint32_t nb = (uint32_t)B.lens.size();
  const uint64_t total = B.cols.size();
  uint8_t* host = (uint8_t*)malloc(total ? (size_t)total : 1);
  if (!host) return CJS_E_OUT_OF_MEMORY;
  if (nb) {
    // n <= 1 blocks: unbwtransform copies (:1149-1152); handled by the same kernels (a 1-element chain)
    uint8_t *d_T = nullptr, *d_out = nullptr; hipStream_t s = nullptr;
    int rc = 0;
    if (hipMalloc((void**)&d_T, (size_t)total + 64) != hipSuccess) rc = CJS_E_OUT_OF_MEMORY;
    if (!rc && hipMalloc((void**)&d_out, (size_t)total + 64) != hipSuccess) rc = CJS_E_OUT_OF_MEMORY;
    if (!rc && hipStreamCreate(&s) != hipSuccess) rc = CJS_E_HIP;
    if (!rc && hipMemcpyAsync(d_T, B.cols.data(), (size_t)total, hipMemcpyHostToDevice, s) != hipSuccess) rc = CJS_E_HIP;
    if (!rc) rc = ibwt_sentinel_run(s, d_T, (uint32_t)B.level * 100000u, nb, B.lens.data(), B.pidx.data(), d_out);
    if (!rc && total && hipMemcpy(host, d_out, (size_t)total, hipMemcpyDeviceToHost) != hipSuccess) rc = CJS_E_HIP;
    if (s) (void)hipStreamDestroy(s);
    if (d_T) (void)hipFree(d_T);
    if (d_out) (void)hipFree(d_out);
    if (rc) { free(host); return rc; }
  }
  *out = host; *out_n = (size_t)total;
  return 0;
  CJS_GUARD_END(CJS_E_OUT_OF_MEMORY, CJS_E_HIP)
}
