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
#include <vector>

namespace cjs { int select_device(const cjs_opts* opts); }
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
__device__ __forceinline__ void fen_path(const Fen& f, int leaf, uint32_t update, uint32_t& lt, uint32_t& tot) {
  const int lane = lane_id();
  const int node = lane < 16 ? leaf >> lane : 0;       // level `lane` of the path (0 when above the root)
  uint32_t contrib = 0;
  if (node > 1 && (node & 1)) contrib = f.tree[node - 1];
  const uint32_t root = f.tree[1];
  __builtin_amdgcn_wave_barrier();
  lt = wave_sum(contrib);
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
// encode(symbol) (:1530-1571).  Emits 1 or 2 coder steps into out[*n..]
__device__ void fen_encode(const Fen& f, int symbol, uint64_t* out, uint32_t& n, bool top_level = true) {
  const int leaf = f.num_syms + symbol;
  const uint32_t sy_raw = f.tree[leaf];
  uint32_t mask = 0xFFFF0000u; int shift = 16;
  uint32_t update = F_INC << 16;
  if ((sy_raw & 0xFFFF0000u) == 0) {                   // escape first, then code in the escape distribution
    if (top_level) fen_encode(f, f.num_syms - 1, out, n, false);
    mask = 0x0000FFFFu; update -= 1u; shift = 0;
  } else if (symbol == f.num_syms - 1 && (f.tree[1] & 0xFFFFu) == 1u) {
    update = 0u - f.tree[leaf];                        // last escape: zero it out
  }
  uint32_t lt, tot;
  fen_path(f, leaf, update, lt, tot);
  if (lane_id() == 0)
    out[n] = (uint64_t)((sy_raw & mask) >> shift) | ((uint64_t)((lt & mask) >> shift) << 16) | ((uint64_t)((tot & mask) >> shift) << 32);
  n++;
  if ((f.tree[1] >> 16) >= F_MAX) fen_rescale(f);
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
      const int sym = (int)__shfl(mine, (int)j, 64);
      fen_encode(f, sym, out, n);
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
  std::vector<uint8_t>& out;
  uint32_t low = 0, range = 0x80000000u, help = 0, bytecount = 0;
  int buffer = 0;
  explicit HostCoder(std::vector<uint8_t>& o) : out(o) {}
  void start(int c, uint32_t initlen) { low = 0; range = 0x80000000u; buffer = c; help = 0; bytecount = initlen; }
  inline void normalize() {
    while (range <= 0x00800000u) {
      if (low < (0xFFu << 23)) { out.push_back((uint8_t)buffer); for (; help; help--) out.push_back(0xFF); buffer = (low >> 23) & 0xFF; }
      else if (low & 0x80000000u) { out.push_back((uint8_t)(buffer + 1)); for (; help; help--) out.push_back(0x00); buffer = (low >> 23) & 0xFF; }
      else help++;
      range <<= 8; low = (low << 8) & 0x7FFFFFFFu; bytecount++;
    }
  }
  inline void freq(uint32_t sy, uint32_t lt, uint32_t tot) {
    normalize();
    const uint32_t r = range / tot, tmp = r * lt;
    low += tmp;
    if (lt + sy < tot) range = r * sy; else range -= tmp;
  }
  inline void shift(uint32_t sy, uint32_t lt, int sh) {
    normalize();
    const uint32_t r = range >> sh, tmp = r * lt;
    low += tmp;
    if ((lt + sy) >> sh) range -= tmp; else range = r * sy;
  }
  void finish() {
    normalize();
    bytecount += 5;
    uint32_t tmp = low >> 23;
    if ((low & 0x7FFFFFu) >= ((bytecount & 0xFFFFFFu) >> 1)) tmp++;
    if (tmp > 0xFF) { out.push_back((uint8_t)(buffer + 1)); for (; help; help--) out.push_back(0x00); }
    else { out.push_back((uint8_t)buffer); for (; help; help--) out.push_back(0xFF); }
    out.push_back((uint8_t)(tmp & 0xFF));
    out.push_back((uint8_t)((bytecount >> 16) & 0xFF)); out.push_back((uint8_t)((bytecount >> 8) & 0xFF)); out.push_back((uint8_t)(bytecount & 0xFF));
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
      else hipLaunchKernelGGL(bwtc_fenwick, dim3(nb), dim3(64), 0, s, mw.b, d_steps, step_stride, d_nsteps);
      if (hipGetLastError() != hipSuccess) rc = CJS_E_HIP;
    }
    std::vector<uint32_t> h_pidx(nb), h_asz(nb), h_nsteps(nb);
    std::vector<uint8_t> h_alist((size_t)nb * 256);
    if (!rc && hipMemcpyAsync(h_pidx.data(), d_pidx, 4 * (size_t)nb, hipMemcpyDeviceToHost, s) != hipSuccess) rc = CJS_E_HIP;
    if (!rc && hipMemcpyAsync(h_asz.data(), mw.b.asz, 4 * (size_t)nb, hipMemcpyDeviceToHost, s) != hipSuccess) rc = CJS_E_HIP;
    if (!rc && hipMemcpyAsync(h_nsteps.data(), d_nsteps, 4 * (size_t)nb, hipMemcpyDeviceToHost, s) != hipSuccess) rc = CJS_E_HIP;
    if (!rc && hipMemcpyAsync(h_alist.data(), mw.b.alist, (size_t)nb * 256, hipMemcpyDeviceToHost, s) != hipSuccess) rc = CJS_E_HIP;
    if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = CJS_E_HIP;
    std::vector<uint64_t> h_steps;
    for (uint32_t k = 0; k < nb && !rc; k++) {
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
      h_steps.resize(h_nsteps[k]);
      if (h_nsteps[k] && hipMemcpy(h_steps.data(), d_steps + (size_t)k * step_stride, 8 * (size_t)h_nsteps[k], hipMemcpyDeviceToHost) != hipSuccess) { rc = CJS_E_HIP; break; }
      for (uint32_t i = 0; i < h_nsteps[k]; i++) {                    // the serial tail
        const uint64_t st = h_steps[i];
        const uint32_t sy = (uint32_t)(st & 0xFFFF), lt = (uint32_t)((st >> 16) & 0xFFFF), tot = (uint32_t)((st >> 32) & 0x1FFFF);
        if (st & STEP_SHIFT_FLAG) coder.shift(sy, lt, (int)tot); else coder.freq(sy, lt, tot);
      }
    }
    if (s) (void)hipStreamDestroy(s);
    if (bw.h_counters) (void)hipHostFree(bw.h_counters);
    arena.destroy();
  }
  if (rc) return rc;
  coder.freq(1, 2, 3);                                               // "no more blocks" :1823
  coder.finish();
  uint8_t* host = (uint8_t*)malloc(o.size() ? o.size() : 1);
  if (!host) return CJS_E_OUT_OF_MEMORY;
  memcpy(host, o.data(), o.size());
  *out = host; *out_n = o.size();
  return 0;
}
