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
#include <chrono>
#include <mutex>
#include <vector>

namespace cjs {
int select_device(const cjs_opts* opts);
int ibwt_sentinel_run(hipStream_t s, const uint8_t* d_T, uint32_t stride, uint32_t nb, const uint32_t* lens, const uint32_t* pidx, uint8_t* d_out);
}
using namespace cjs;

namespace cjs {

constexpr uint32_t F_MAX = 0xFF00u, F_INC = 0x0100u;
constexpr uint64_t STEP_SHIFT_FLAG = 1ull << 63;   // step is encodeShift(sy, lt, shift) instead of encodeFreq

// ---------------------------------------------------------------- FenwickModel, one wave per block
// tree lives in LDS; lane l owns level l of the leaf->root path.
struct Fen {
  uint32_t* tree; int num_syms;
};
// path sum + update (J/BWTC_joined_.js:1547-1562).  Returns lt (all lanes) and tot (old root).
// lane l owns level l of the path; the <= 10 per-level contributions are summed with scalar readlanes.
__device__ __forceinline__ void fen_path(const Fen& f, int leaf, uint32_t update, uint32_t& lt, uint32_t& tot) {
  const int lane = lane_id();
  const int node = lane < 16 ? leaf >> lane : 0;       // level `lane` of the path (0 when above the root)
  uint32_t contrib = 0;
  if (node > 1 && (node & 1)) contrib = f.tree[node - 1];
  const uint32_t root = f.tree[1];
  uint32_t sum = 0;
#pragma unroll
  for (int l = 0; l < 10; l++) sum += (uint32_t)__builtin_amdgcn_readlane((int)contrib, l);   // 2*num_syms <= 516 < 2^10
  lt = sum;
  tot = root;
  if (node >= 1) f.tree[node] += update;
  __builtin_amdgcn_wave_barrier();
}
__device__ void fen_sum_tree(const Fen& f) {           // _sumTree (:1655-1661), wave-parallel by index batches / levels
  const int lane = lane_id();
  int hi = f.num_syms - 1;
  while (hi > 127) {                                   // no dependency inside a 64-wide batch when hi > 126
    const int i = hi - lane;
    if (i > 127) f.tree[i] = f.tree[2 * i] + f.tree[2 * i + 1];
    __builtin_amdgcn_wave_barrier();
    hi -= 64;
    if (hi < 127) hi = 127;
  }
  for (int top = 64; top >= 1; top >>= 1) {            // levels [top, 2*top)
    const int i = top + lane;
    if (lane < top && i <= f.num_syms - 1) f.tree[i] = f.tree[2 * i] + f.tree[2 * i + 1];
    __builtin_amdgcn_wave_barrier();
  }
}
__device__ void fen_rescale(const Fen& f) {            // _rescale (:1623-1654)
  const int lane = lane_id();
  bool esc_here = false;
  for (int i = lane; i < f.num_syms - 1; i += 64) {
    uint32_t prob = f.tree[f.num_syms + i];
    if (prob & 0xFFFFu) { esc_here = true; continue; }
    prob = (prob & 0xFFFEFFFEu) >> 1;
    if (prob == 0) { prob = 1u; esc_here = true; }
    f.tree[f.num_syms + i] = prob;
  }
  const bool no_escape = __ballot(esc_here) == 0ull;
  if (lane == 0) {
    uint32_t prob = f.tree[2 * f.num_syms - 1];
    prob = (prob & 0xFFFEFFFEu) >> 1;
    if (no_escape) prob = 0; else if (prob == 0) prob = 1u << 16;
    f.tree[2 * f.num_syms - 1] = prob;
  }
  __builtin_amdgcn_wave_barrier();
  fen_sum_tree(f);
}
// one coder step of encode() (:1530-1571): plain symbol, or symbol in the escape distribution (esc_ctx)
__device__ __forceinline__ void fen_step(const Fen& f, int symbol, bool esc_ctx, uint64_t* out, uint32_t& n) {
  const int leaf = f.num_syms + symbol;
  const uint32_t sy_raw = f.tree[leaf];
  uint32_t mask = 0xFFFF0000u; int shift = 16;
  uint32_t update = F_INC << 16;
  if (esc_ctx) { mask = 0x0000FFFFu; update -= 1u; shift = 0; }
  else if (symbol == f.num_syms - 1 && (f.tree[1] & 0xFFFFu) == 1u) update = 0u - sy_raw;     // last escape: zero it out
  uint32_t lt, tot;
  fen_path(f, leaf, update, lt, tot);
  if (lane_id() == 0)
    out[n] = (uint64_t)((sy_raw & mask) >> shift) | ((uint64_t)((lt & mask) >> shift) << 16) | ((uint64_t)((tot & mask) >> shift) << 32);
  n++;
  if ((f.tree[1] >> 16) >= F_MAX) fen_rescale(f);
}
// encode(symbol): a symbol whose own count is still zero is announced by the escape symbol first (:1537-1541)
__device__ __forceinline__ void fen_encode(const Fen& f, int symbol, uint64_t* out, uint32_t& n) {
  const bool esc = (f.tree[f.num_syms + symbol] & 0xFFFF0000u) == 0;
  if (esc) fen_step(f, f.num_syms - 1, false, out, n);
  fen_step(f, symbol, esc, out, n);
}

__global__ __launch_bounds__(64) void bwtc_fenwick(MtfBufs mb, uint64_t* __restrict__ steps, size_t step_stride, uint32_t* __restrict__ nsteps) {
  __shared__ uint32_t tree[520];
  const uint32_t blk = blockIdx.x;
  const uint32_t asz = mb.asz[blk], nsym = mb.npos[blk] - 1;      // drop bzip2's EOB
  const uint16_t* A = mb.A + (size_t)blk * mb.a_stride;
  uint64_t* out = steps + (size_t)blk * step_stride;
  const int lane = lane_id();
  Fen f{tree, (int)asz + 2};
  const int size = (int)asz + 1;
  for (int i = lane; i < 2 * f.num_syms; i += 64) tree[i] = 0;
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < size; i += 64) tree[f.num_syms + i] = 1u;
  if (lane == 0) tree[f.num_syms + size] = F_INC << 16;
  __builtin_amdgcn_wave_barrier();
  fen_sum_tree(f);
  uint32_t n = 0;
  for (uint32_t base = 0; base < nsym; base += 64) {
    const uint32_t mine = base + lane < nsym ? A[base + lane] : 0u;
    const uint32_t cnt = nsym - base < 64 ? nsym - base : 64;
    for (uint32_t j = 0; j < cnt; j++) {
      const int sym = __builtin_amdgcn_readlane((int)mine, (int)j);
      fen_encode(f, sym, out, n);
    }
  }
  if (lane == 0) nsteps[blk] = n;
}

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

__global__ __launch_bounds__(64) void bwtc_fenwick_par(MtfBufs mb, uint64_t* __restrict__ steps, size_t step_stride, uint32_t* __restrict__ nsteps, int force_serial) {
  __shared__ uint32_t leaf[328], cum[328];
  const uint32_t blk = blockIdx.x;
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
  for (int i = lane; i < 328; i += 64) { leaf[i] = i == e ? (F_INC << 16) : i < ns ? 1u : 0u; cum[i] = 0; }
  __builtin_amdgcn_wave_barrier();
  uint32_t n = 0;
  const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;       // lanes < lane
  for (uint32_t base = 0; base < nsym; base += 64) {
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
      const bool cut = active && (through >= kstar || (escj && (root_lo - Eb == 1u || (force_serial & 2))));
      const uint64_t cutm = __ballot(cut);
      const uint32_t c = (force_serial & 1) ? start : cutm ? (uint32_t)__builtin_ctzll(cutm) : cnt;                       // [start, c) in parallel, c one at a time
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
  }
  if (lane == 0) nsteps[blk] = n;
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

struct HostCoder {                                      // RangeCoder encode side (J/BWTC_joined_.js:40-153)
  std::vector<uint8_t>& out;                            // bytes [0, len) are output; the vector is kept larger (reserve())
  uint8_t* data = nullptr;
  size_t len = 0;
  uint32_t low = 0, range = 0x80000000u, help = 0, bytecount = 0;
  int buffer = 0;
  explicit HostCoder(std::vector<uint8_t>& o) : out(o) { len = o.size(); reserve(4096); }
  void reserve(size_t extra) {                          // room for `extra` more bytes (a coder step emits at most 3)
    if (len + extra > out.size()) out.resize(std::max(out.size() * 2, len + extra));
    data = out.data();
  }
  inline void emit(uint8_t b) { data[len++] = b; }
  void start(int c, uint32_t initlen) { low = 0; range = 0x80000000u; buffer = c; help = 0; bytecount = initlen; }
  inline void normalize() {
    while (range <= 0x00800000u) {
      if (low < (0xFFu << 23)) { emit((uint8_t)buffer); for (; help; help--) emit(0xFF); buffer = (low >> 23) & 0xFF; }
      else if (low & 0x80000000u) { emit((uint8_t)(buffer + 1)); for (; help; help--) emit(0x00); buffer = (low >> 23) & 0xFF; }
      else help++;
      range <<= 8; low = (low << 8) & 0x7FFFFFFFu; bytecount++;
    }
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
  inline void freq(uint32_t sy, uint32_t lt, uint32_t tot) {
    normalize();
    const uint32_t r = range / tot, tmp = r * lt;
    low += tmp;
    if (lt + sy < tot) range = r * sy; else range -= tmp;
  }
  // same arithmetic with the division replaced by a multiply with a 64-bit reciprocal: the quotient sits on the serial
  // (low, range) chain, the reciprocal (looked up by tot, which comes from the step list) does not
  inline void freq_rcp(uint32_t sy, uint32_t lt, uint32_t tot, const uint64_t* rcp) {
    normalize_step();
    // rcp[tot] = floor(2^64 / tot) + 1: the high half of range * rcp is floor(range / tot) exactly (range < 2^32, tot < 2^17:
    // range * (rcp*tot - 2^64) < 2^49 < 2^64), so no correction step sits on the chain
    const uint32_t r = (uint32_t)(((unsigned __int128)range * rcp[tot]) >> 64);
    const uint32_t tmp = r * lt;
    low += tmp;
    range = (lt + sy < tot) ? r * sy : range - tmp;
  }
  inline void shift(uint32_t sy, uint32_t lt, int sh) {
    normalize();
    const uint32_t r = range >> sh, tmp = r * lt;
    low += tmp;
    if ((lt + sy) >> sh) range -= tmp; else range = r * sy;
  }
  void finish() {
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

extern "C" int cjs_bwtc_compress(const uint8_t* in, size_t n, int level, uint8_t** out, size_t* out_n, const cjs_opts* opts) {
  if (!out || !out_n) return CJS_E_INVALID_ARG;
  *out = nullptr; *out_n = 0;
  CJS_TRY(select_device(opts));
  if (level < 1 || level > 9) level = 9;                             // J/BWTC_joined_.js:1702-1705
  const bool fast = level <= 5;
  const uint32_t bs = (uint32_t)level * 100000u;
  const uint32_t nb = (uint32_t)((n + bs - 1) / bs);
  std::vector<uint8_t> o;
  o.reserve(n / 3 + 64);
  o.push_back('b'); o.push_back('w'); o.push_back('t'); o.push_back('c');
  uint8_t vb[12]; int nv = 0;                                        // writeUnsignedNumber(size+1) :605-620
  { uint64_t v = (uint64_t)n + 1; do { vb[nv++] = (uint8_t)(v & 0x7F); v >>= 7; } while (v); vb[0] |= 0x80; }
  for (int i = nv - 1; i >= 1; i--) o.push_back(vb[i]);
  HostCoder coder(o);
  coder.start(vb[0], 1);                                             // :1700 (the last varint byte is the coder's first byte)
  coder.shift(1, (uint32_t)level, 8);                                // encodeByte(level) :1706
  int rc = 0;
  const auto T0 = std::chrono::steady_clock::now();
  auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
  if (nb) {
    const uint32_t n_last = (uint32_t)(n - (size_t)(nb - 1) * bs);
    Arena arena;
    const size_t elems = (size_t)nb * bs;
    const size_t a_stride = MtfWork::a_stride_for(bs), step_stride = 2 * a_stride;
    CJS_TRY(arena.init(BwtWork::bytes_needed(elems) + MtfWork::bytes_needed(nb, bs) + 2 * (elems + 512) + 8 * (size_t)nb * step_stride +
                       16 * (size_t)nb + 65536));
    BwtWork bw; MtfWork mw;
    rc = bw.carve(arena, elems);
    if (!rc) rc = mw.carve(arena, nb, bs);
    uint8_t* d_T = arena.take<uint8_t>(elems);
    uint8_t* d_U = arena.take<uint8_t>(elems);
    uint32_t* d_pidx = arena.take<uint32_t>(nb);
    uint32_t* d_len = arena.take<uint32_t>(nb);
    uint32_t* d_nsteps = arena.take<uint32_t>(nb);
    uint64_t* d_steps = arena.take<uint64_t>((size_t)nb * step_stride);
    if (!rc && !d_steps) rc = CJS_E_OUT_OF_MEMORY;
    hipStream_t s = nullptr;
    if (!rc && hipStreamCreate(&s) != hipSuccess) rc = CJS_E_HIP;
    std::vector<uint32_t> lens(nb, bs); lens[nb - 1] = n_last;
    if (!rc && hipMemcpyAsync(d_T, in, n, hipMemcpyHostToDevice, s) != hipSuccess) rc = CJS_E_HIP;
    if (!rc && hipMemcpyAsync(d_len, lens.data(), 4 * (size_t)nb, hipMemcpyHostToDevice, s) != hipSuccess) rc = CJS_E_HIP;
    if (!rc) rc = bwt_run(s, bw, d_T, nb, bs, n_last, false, d_U, d_pidx, nullptr);
    if (!rc) rc = mtf_run(s, mw, d_U, nb, d_len);
    if (!rc) {
      if (fast) hipLaunchKernelGGL(bwtc_defsum, dim3(nb), dim3(64), 0, s, mw.b, d_steps, step_stride, d_nsteps);
      else if (getenv("CJS_BWTC_SERIAL_MODEL")) hipLaunchKernelGGL(bwtc_fenwick, dim3(nb), dim3(64), 0, s, mw.b, d_steps, step_stride, d_nsteps);
      else hipLaunchKernelGGL(bwtc_fenwick_par, dim3(nb), dim3(64), 0, s, mw.b, d_steps, step_stride, d_nsteps, getenv("CJS_BWTC_FORCE_SERIAL") ? atoi(getenv("CJS_BWTC_FORCE_SERIAL")) : 0);
      if (hipGetLastError() != hipSuccess) rc = CJS_E_HIP;
      if (!rc && !fast && getenv("CJS_BWTC_CHECK")) {               // debug: the one-symbol-at-a-time kernel must give the same steps
        uint64_t* d_ref = nullptr; uint32_t* d_nref = nullptr;
        if (hipMalloc((void**)&d_ref, 8 * (size_t)nb * step_stride) == hipSuccess && hipMalloc((void**)&d_nref, 4 * (size_t)nb) == hipSuccess) {
          hipLaunchKernelGGL(bwtc_fenwick, dim3(nb), dim3(64), 0, s, mw.b, d_ref, step_stride, d_nref);
          std::vector<uint32_t> na(nb), nr(nb);
          (void)hipMemcpyAsync(na.data(), d_nsteps, 4 * (size_t)nb, hipMemcpyDeviceToHost, s);
          (void)hipMemcpyAsync(nr.data(), d_nref, 4 * (size_t)nb, hipMemcpyDeviceToHost, s);
          (void)hipStreamSynchronize(s);
          for (uint32_t k = 0; k < nb; k++) {
            const uint32_t m = na[k] < nr[k] ? na[k] : nr[k];
            std::vector<uint64_t> a(m), r(m);
            if (m) { (void)hipMemcpy(a.data(), d_steps + (size_t)k * step_stride, 8 * (size_t)m, hipMemcpyDeviceToHost); (void)hipMemcpy(r.data(), d_ref + (size_t)k * step_stride, 8 * (size_t)m, hipMemcpyDeviceToHost); }
            uint32_t i = 0; while (i < m && a[i] == r[i]) i++;
            if (i < m || na[k] != nr[k]) {
              fprintf(stderr, "[cjs bwtc check] block %u: steps %u vs %u, first difference at %u", k, na[k], nr[k], i);
              for (uint32_t q = i > 2 ? i - 2 : 0; q < i + 3 && q < m; q++) fprintf(stderr, "  [%u] %llx | %llx", q, (unsigned long long)a[q], (unsigned long long)r[q]);
              fprintf(stderr, "\n");
              break;
            }
          }
        }
        if (d_ref) (void)hipFree(d_ref);
        if (d_nref) (void)hipFree(d_nref);
      }
    }
    std::vector<uint32_t> h_pidx(nb), h_asz(nb), h_nsteps(nb);
    std::vector<uint8_t> h_alist((size_t)nb * 256);
    if (!rc && hipMemcpyAsync(h_pidx.data(), d_pidx, 4 * (size_t)nb, hipMemcpyDeviceToHost, s) != hipSuccess) rc = CJS_E_HIP;
    if (!rc && hipMemcpyAsync(h_asz.data(), mw.b.asz, 4 * (size_t)nb, hipMemcpyDeviceToHost, s) != hipSuccess) rc = CJS_E_HIP;
    if (!rc && hipMemcpyAsync(h_nsteps.data(), d_nsteps, 4 * (size_t)nb, hipMemcpyDeviceToHost, s) != hipSuccess) rc = CJS_E_HIP;
    if (!rc && hipMemcpyAsync(h_alist.data(), mw.b.alist, (size_t)nb * 256, hipMemcpyDeviceToHost, s) != hipSuccess) rc = CJS_E_HIP;
    if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = CJS_E_HIP;
    const double ms_gpu = since(T0);
    const auto T1 = std::chrono::steady_clock::now();
    // reciprocals floor(2^64 / tot) + 1 for every total a step can carry (17 bits); tot < 2 keeps the division
    static std::vector<uint64_t> rcp;
    static std::once_flag rcp_once;
    std::call_once(rcp_once, [] { rcp.assign(1u << 17, 0ull); for (uint32_t t = 2; t < (1u << 17); t++) rcp[t] = (uint64_t)(((unsigned __int128)1 << 64) / t) + 1; });
    // the steps of block k+1 travel (pinned buffer, copy stream) while block k goes through the coder
    uint64_t* h_buf[2] = {nullptr, nullptr};
    hipStream_t cs = nullptr; hipEvent_t cev[2] = {nullptr, nullptr};
    if (!rc && (hipHostMalloc((void**)&h_buf[0], 8 * step_stride) != hipSuccess || hipHostMalloc((void**)&h_buf[1], 8 * step_stride) != hipSuccess ||
                hipStreamCreate(&cs) != hipSuccess || hipEventCreateWithFlags(&cev[0], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&cev[1], hipEventDisableTiming) != hipSuccess)) rc = CJS_E_HIP;
    auto fetch = [&](uint32_t k) {
      if (h_nsteps[k] && hipMemcpyAsync(h_buf[k & 1], d_steps + (size_t)k * step_stride, 8 * (size_t)h_nsteps[k], hipMemcpyDeviceToHost, cs) != hipSuccess) return (int)CJS_E_HIP;
      return hipEventRecord(cev[k & 1], cs) == hipSuccess ? 0 : (int)CJS_E_HIP;
    };
    if (!rc && nb) rc = fetch(0);
    for (uint32_t k = 0; k < nb && !rc; k++) {
      if (k + 1 < nb) rc = fetch(k + 1);
      if (rc) break;
      coder.reserve(3 * (size_t)h_nsteps[k] + coder.help + 4096);      // everything this block can emit
      const uint32_t length = lens[k];
      if (length == bs) coder.freq(1, 0, 3);                         // "full size block" :1734
      else { coder.freq(1, 1, 3); logdist(coder, (int)bs, length); } // "short block" :1737-1738
      logdist(coder, (int)bs, h_pidx[k]);                            // :1742
      uint16_t tree[512]; memset(tree, 0, sizeof tree);              // use-tree :1744-1765
      for (uint32_t i = 0; i < h_asz[k]; i++) tree[256 + h_alist[(size_t)k * 256 + i]] = 1;
      for (int i = 255; i > 0; i--) tree[i] = (uint16_t)(tree[2 * i] + tree[2 * i + 1]);
      tree[0] = 1;
      for (int i = 1; i < 512; i++) {
        const int parent = i >> 1, full = 1 << (9 - fls32((uint32_t)i));
        if (tree[parent] == 0 || tree[parent] == full * 2) continue;
        if (i >= 256) coder.shift(1, tree[i] ? 1 : 0, 1);
        else coder.freq(1, tree[i] == 0 ? 0u : tree[i] == full ? 2u : 1u, 3);
      }
      if (hipEventSynchronize(cev[k & 1]) != hipSuccess) { rc = CJS_E_HIP; break; }
      const uint64_t* h_steps = h_buf[k & 1];
      const uint64_t* rc_tab = rcp.data();
      for (uint32_t i = 0; i < h_nsteps[k]; i++) {                    // the serial tail
        const uint64_t st = h_steps[i];
        const uint32_t sy = (uint32_t)(st & 0xFFFF), lt = (uint32_t)((st >> 16) & 0xFFFF), tot = (uint32_t)((st >> 32) & 0x1FFFF);
        if (st & STEP_SHIFT_FLAG) coder.shift(sy, lt, (int)tot);
        else if (tot >= 2) coder.freq_rcp(sy, lt, tot, rc_tab);
        else coder.freq(sy, lt, tot);
      }
    }
    if (getenv("CJS_DEBUG")) fprintf(stderr, "[cjs bwtc] workspace + H2D + BWT + MTF + model %.1f ms, range coder over the step lists (host, serial) %.1f ms\n", ms_gpu, since(T1));
    if (cs) (void)hipStreamSynchronize(cs);
    for (int q = 0; q < 2; q++) { if (cev[q]) (void)hipEventDestroy(cev[q]); if (h_buf[q]) (void)hipHostFree(h_buf[q]); }
    if (cs) (void)hipStreamDestroy(cs);
    if (s) (void)hipStreamDestroy(s);
    if (bw.h_counters) (void)hipHostFree(bw.h_counters);
    arena.destroy();
  }
  if (rc) return rc;
  coder.reserve(coder.help + 64);
  coder.freq(1, 2, 3);                                               // "no more blocks" :1823
  coder.finish();
  uint8_t* host = (uint8_t*)malloc(o.size() ? o.size() : 1);
  if (!host) return CJS_E_OUT_OF_MEMORY;
  memcpy(host, o.data(), o.size());
  *out = host; *out_n = o.size();
  return 0;
}


// ---------------------------------------------------------------- BWTC.decompressFile (J/BWTC_joined_.js:1827-1920)
// Range decoding and the adaptive model are one serial chain over the whole file (every decoded symbol feeds the
// model that decodes the next one), so that part runs on one host thread; the inverse BWT of all blocks
// (BWT.unbwtransform, n dependent gathers per block in the reference) runs on the GPU (decode.hip).
namespace {

struct HostDecoder {                                    // RangeCoder decode side (:159-238)
  const uint8_t* in; size_t n, pos;
  uint32_t low = 0, range = 0, help = 0; int32_t buffer = 0;
  int32_t read_byte() { return pos < n ? (int32_t)in[pos++] : -1; }
  void start() { buffer = read_byte(); low = (uint32_t)buffer >> 1; range = 1u << 7; }          // decodeStart(skipInitialRead)
  inline void normalize() {
    while (range <= 0x00800000u) {
      low = (low << 8) | (((uint32_t)buffer << 7) & 0xFF);
      buffer = read_byte();
      low |= (uint32_t)buffer >> 1;
      range <<= 8;
    }
  }
  inline uint32_t cul_freq(uint32_t tot) { normalize(); help = range / tot; const uint32_t t = help ? low / help : 0; return t >= tot ? tot - 1 : t; }
  inline uint32_t cul_shift(int sh) { normalize(); help = range >> sh; const uint32_t t = help ? low / help : 0; return (t >> sh) ? (1u << sh) - 1 : t; }
  inline void update(uint32_t sy, uint32_t lt, uint32_t tot) { const uint32_t tmp = help * lt; low -= tmp; if (lt + sy < tot) range = help * sy; else range -= tmp; }
  uint32_t bit() { const uint32_t t = cul_shift(1); update(1, t, 2); return t; }
  bool overrun() const { return pos >= n + 8; }
};
uint32_t nomodel_dec(HostDecoder& d, int bits) { uint32_t r = 0; for (int i = bits - 1; i >= 0; i--) { r <<= 1; if (d.bit()) r++; } return r; }
uint32_t logdist_dec(HostDecoder& d, int block_size) {                                          // :1254-1261
  const int lgbits = fls32((uint32_t)(1 + fls32((uint32_t)block_size - 1)) - 1);
  const uint32_t lg = nomodel_dec(d, lgbits);
  if (lg < 2) return lg;
  return (1u << (lg - 1)) + nomodel_dec(d, (int)lg - 1);
}

struct FenDec {                                         // FenwickModel decode side (:1572-1661)
  HostDecoder& d; int num_syms; std::vector<uint32_t> tree;
  FenDec(HostDecoder& dd, int size) : d(dd), num_syms(size + 1), tree((size_t)(size + 1) * 2, 0) {
    int i; for (i = 0; i < size; i++) tree[num_syms + i] = 1u;
    tree[num_syms + i] = 0x100u << 16; sum();
  }
  void sum() { for (int i = num_syms - 1; i > 0; i--) tree[i] = tree[2 * i] + tree[2 * i + 1]; }
  void rescale() {
    int i; bool no_escape = true; uint32_t prob;
    for (i = 0; i < num_syms - 1; i++) {
      prob = tree[num_syms + i];
      if (prob & 0xFFFFu) { no_escape = false; continue; }
      prob = (prob & 0xFFFEFFFEu) >> 1;
      if (prob == 0) { prob = 1u; no_escape = false; }
      tree[num_syms + i] = prob;
    }
    prob = tree[num_syms + i]; prob = (prob & 0xFFFEFFFEu) >> 1;
    if (no_escape) prob = 0; else if (prob == 0) prob = 1u << 16;
    tree[num_syms + i] = prob; sum();
  }
  int decode1(bool esc) {
    uint32_t mask = 0xFFFF0000u; int shift = 16; uint32_t upd = 0x100u << 16;
    if (esc) { mask = 0xFFFFu; upd -= 1u; shift = 0; }
    const uint32_t tot = (tree[1] & mask) >> shift;
    if (tot == 0) return -1;
    const uint32_t prob = d.cul_freq(tot);
    int i = 1; uint32_t lt = 0;
    while (i < num_syms) {
      tree[i] += upd;
      const uint32_t left = (tree[2 * i] & mask) >> shift;
      i *= 2;
      if (prob - lt >= left) { lt += left; i++; }
    }
    const int symbol = i - num_syms;
    const uint32_t sy = (tree[i] & mask) >> shift;
    tree[i] += upd;
    d.update(sy, lt, tot);
    if (symbol == num_syms - 1 && (tree[1] & 0xFFFFu) == 1u) { upd = 0u - tree[i]; while (i >= 1) { tree[i] += upd; i >>= 1; } }
    if ((tree[1] >> 16) >= 0xFF00u) rescale();
    return symbol;
  }
  int decode() { int s = decode1(false); if (s == num_syms - 1) s = decode1(true); return s; }
};

struct DsmDec {                                         // DefSumModel decode side (:1327-1459)
  HostDecoder& d; int ns; uint16_t prob[304], esc[304], upd[304], p2s[256], e2s[304]; int ucount = 0, uthresh = 128;
  DsmDec(HostDecoder& dd, int size) : d(dd), ns(size) {
    memset(prob, 0, sizeof prob); memset(esc, 0, sizeof esc); memset(upd, 0, sizeof upd);
    prob[ns + 1] = 256;
    for (int i = 0; i <= ns; i++) esc[i] = (uint16_t)i;
    for (int i = 0; i < 256; i++) p2s[i] = (uint16_t)ns;
    for (int i = 0; i < 304; i++) e2s[i] = (uint16_t)(i < ns ? i : 0);
  }
  void update(int symbol) {
    if (symbol == ns) { if (upd[symbol] >= 40) return; if (ucount >= uthresh - 1) return; }
    upd[symbol]++; ucount++;
    if (ucount < uthresh) return;
    int cum = 0, cum_esc = 0, odd = 0, i;
    esc[0] = 0; prob[0] = 0;
    for (i = 0; i < ns + 1; i++) {
      const int np = ((prob[i + 1] - prob[i]) >> 1) + upd[i];
      prob[i] = (uint16_t)cum; esc[i] = (uint16_t)cum_esc;
      if (np) { cum += np; odd += np & 1; } else cum_esc++;
    }
    prob[i] = (uint16_t)cum;
    uthresh = 256 - (cum - odd) / 2;
    for (i = 0; i < ns + 1; i++) upd[i] = 0;
    upd[ns] = 1; ucount = 1;
    int j = 0, k = 0;
    for (i = 0; i < ns + 1; i++) {
      for (; j < prob[i + 1]; j++) p2s[j] = (uint16_t)i;
      const int el = i + 1 <= ns ? esc[i + 1] : 0;       // escape[] has ns+1 entries in the reference
      for (; k < el; k++) e2s[k] = (uint16_t)i;
    }
  }
  int decode() {
    uint32_t p = d.cul_shift(8);
    int symbol = p2s[p];
    uint32_t lt = prob[symbol], sy = (uint32_t)prob[symbol + 1] - lt;
    d.update(sy, lt, 256); update(symbol);
    if (symbol != ns) return symbol;
    const uint32_t tot = esc[ns];
    if (tot == 0) return -1;
    p = d.cul_freq(tot);
    symbol = e2s[p];
    lt = esc[symbol]; sy = (uint32_t)esc[symbol + 1] - lt;
    d.update(sy, lt, tot); update(symbol);
    return symbol;
  }
};

}  // namespace

extern "C" int cjs_bwtc_decompress(const uint8_t* in, size_t n, uint8_t** out, size_t* out_n, const cjs_opts* opts) {
  if (!out || !out_n) return CJS_E_INVALID_ARG;
  *out = nullptr; *out_n = 0;
  CJS_TRY(select_device(opts));
  if (n < 4 || in[0] != 'b' || in[1] != 'w' || in[2] != 't' || in[3] != 'c') return CJS_E_BAD_MAGIC;     // :559-565
  size_t p = 4;
  for (;;) { if (p >= n) return CJS_E_DATA_ERROR; if (in[p++] & 0x80) break; }                          // readUnsignedNumber :621-633
  HostDecoder d{in, n, p};
  d.start();
  const uint32_t lv = d.cul_shift(8); d.update(1, lv, 256);                                            // decodeByte :1830
  if (lv < 1 || lv > 9) return CJS_E_DATA_ERROR;
  const bool fast = lv <= 5;
  const uint32_t bs = lv * 100000u;
  std::vector<uint8_t> T;                                // BWT columns of all blocks, stride bs
  std::vector<uint32_t> lens, pidx;
  for (;;) {
    const uint32_t ind = d.cul_freq(3); d.update(1, ind, 3);
    uint32_t length;
    if (ind == 0) length = bs;
    else if (ind == 1) { length = logdist_dec(d, (int)bs); if (length > bs) return CJS_E_DATA_ERROR; }
    else break;
    const uint32_t pi = logdist_dec(d, (int)bs);
    uint16_t tree[512]; memset(tree, 0, sizeof tree); tree[0] = 1;                                     // use-tree :1859-1874
    for (int i = 1; i < 512; i++) {
      const int parent = i >> 1, full = 1 << (9 - fls32((uint32_t)i));
      if (tree[parent] == 0 || tree[parent] == full * 2) tree[i] = tree[parent] >> 1;
      else if (i >= 256) tree[i] = (uint16_t)d.bit();
      else { const uint32_t v = d.cul_freq(3); d.update(1, v, 3); tree[i] = (uint16_t)(v == 2 ? (uint32_t)full : v); }
    }
    uint8_t M[256]; int asz = 0;
    for (int i = 0; i < 256; i++) if (tree[256 + i]) M[asz++] = (uint8_t)i;
    const size_t base = T.size();
    T.resize(base + bs);
    uint8_t* b = T.data() + base;
    FenDec* fm = fast ? nullptr : new FenDec(d, asz + 1);
    DsmDec* dm = fast ? new DsmDec(d, asz + 1) : nullptr;
    uint64_t val = 1; uint32_t i = 0; bool bad = false;
    while (i < length) {                                                                               // :1888-1903
      const int c = fast ? dm->decode() : fm->decode();
      if (c < 0 || d.overrun()) { bad = true; break; }
      if (c == 0) { if (i + val > length) { bad = true; break; } for (uint64_t j = 0; j < val; j++) b[i++] = 0; val *= 2; }
      else if (c == 1) { if (i + 2 * val > length) { bad = true; break; } for (uint64_t j = 0; j < 2 * val; j++) b[i++] = 0; val *= 2; }
      else { val = 1; if (c - 1 >= asz) { bad = true; break; } b[i++] = (uint8_t)(c - 1); }
    }
    delete fm; delete dm;
    if (getenv("CJS_DEBUG")) fprintf(stderr, "[cjs bwtc dec] block %zu: length %u pidx %u asz %d decoded %u bad %d inpos %zu/%zu\n", lens.size(), length, pi, asz, i, (int)bad, d.pos, n);
    if (bad) return CJS_E_DATA_ERROR;
    for (i = 0; i < length; i++) {                                                                     // MTF decode :1905-1913
      int j = b[i]; const uint8_t c = M[j];
      b[i] = c;
      for (; j > 0; j--) M[j] = M[j - 1];
      M[0] = c;
    }
    if (pi > length) return CJS_E_DATA_ERROR;
    lens.push_back(length); pidx.push_back(pi);
  }
  const uint32_t nb = (uint32_t)lens.size();
  uint64_t total = 0;
  for (uint32_t k = 0; k < nb; k++) total += lens[k];
  uint8_t* host = (uint8_t*)malloc(total ? total : 1);
  if (!host) return CJS_E_OUT_OF_MEMORY;
  if (nb) {
    // n <= 1 blocks: unbwtransform copies (:1149-1152); handled by the same kernels (a 1-element chain)
    uint8_t *d_T = nullptr, *d_out = nullptr; hipStream_t s = nullptr;
    int rc = 0;
    if (hipMalloc((void**)&d_T, T.size() + 64) != hipSuccess) rc = CJS_E_OUT_OF_MEMORY;
    if (!rc && hipMalloc((void**)&d_out, total + 64) != hipSuccess) rc = CJS_E_OUT_OF_MEMORY;
    if (!rc && hipStreamCreate(&s) != hipSuccess) rc = CJS_E_HIP;
    if (!rc && hipMemcpyAsync(d_T, T.data(), T.size(), hipMemcpyHostToDevice, s) != hipSuccess) rc = CJS_E_HIP;
    if (!rc) rc = ibwt_sentinel_run(s, d_T, bs, nb, lens.data(), pidx.data(), d_out);
    if (!rc && total && hipMemcpy(host, d_out, total, hipMemcpyDeviceToHost) != hipSuccess) rc = CJS_E_HIP;
    if (s) (void)hipStreamDestroy(s);
    if (d_T) (void)hipFree(d_T);
    if (d_out) (void)hipFree(d_out);
    if (rc) { free(host); return rc; }
  }
  *out = host; *out_n = (size_t)total;
  return 0;
}
