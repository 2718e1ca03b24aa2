// rle1.hip — bzip2 stage 0 on the GPU: RLE1, block boundaries and per-block CRC-32.
//
// Replaces readBlock (J/Bzip2_joined_.js:1954-1985) + CRC32.updateCRC (:1048-1079) + the block
// loop's use of them (:2233-2242).  The reference walks the input byte-serially; block k+1 starts
// where block k's RLE1 OUTPUT reached level*100000-19 bytes and the run-chunking state is reset at
// every block start.  Parallel formulation used here:
//   c[i]  = bytes emitted when input byte i is consumed, under the "global" chunking (chunks of
//           <= 255 equal bytes counted from the start of each maximal run):
//           offset-in-chunk d' = (i - runstart) % 255 : d'<3 -> 1 (literal), d'==3 -> 2 (literal +
//           count byte), d'>=4 -> 0 (absorbed into the count byte)
//   G     = exclusive prefix sum of c (per 4096-byte tile: G_tile, then in-tile scan)
//   a block that starts at s re-chunks only the run that contains s (closed form); after that
//   run the global chunking applies again, so its end is found by a search on G.
// Kernels: tile summaries -> tile scans -> serial-over-blocks walk by ONE workgroup (k-ary search
// over G_tile + in-tile scan; O(#blocks) steps) -> materialise RLE1 bytes -> CRC (slice + GF(2)
// combine).  Integer/byte work, HBM-bound streaming reads.
#include "cjs_internal.h"
#include "prims.hpp"
#include "rle1.h"

namespace cjs {

constexpr uint32_t RT = 4096;          // input bytes per tile
constexpr uint64_t NONE64 = ~0ull;

__device__ __forceinline__ uint32_t emitted_fresh(uint64_t m) {   // output bytes after consuming m bytes of one run (fresh chunking)
  const uint64_t q = m / 255; const uint32_t r = (uint32_t)(m % 255);
  return (uint32_t)(5 * q) + (r < 4 ? r : 5u);
}

// 16 consecutive input bytes of one lane as ONE 16-byte load when the address allows it (consecutive lanes then read
// consecutive 16-byte words: a wave covers 1 KiB per instruction instead of touching 16 lines sixteen times)
__device__ __forceinline__ void load16(const uint8_t* __restrict__ in, uint64_t N, uint64_t p0, uint8_t (&b)[16]) {
  if (p0 + 16 <= N && (((uintptr_t)(in + p0)) & 15u) == 0) {
    const uint4 v = *reinterpret_cast<const uint4*>(in + p0);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 16; j++) b[j] = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
  } else {
#pragma unroll
    for (int j = 0; j < 16; j++) b[j] = p0 + j < N ? in[p0 + j] : 0;
  }
}

// Per-thread view of ITEMS consecutive positions starting at p0: boundary flags and the run start
// governing the first position.  BLOCK threads cover BLOCK*ITEMS positions from tile_start.
// carry = start of the run that contains tile_start when tile_start itself is not a boundary.
template <int BLOCK, int ITEMS>
__device__ __forceinline__ void run_starts(const uint8_t* __restrict__ in, uint64_t N, uint64_t tile_start, uint64_t carry,
                                           uint32_t* smem /* BLOCK + 16 u32 */, uint8_t (&b)[ITEMS], uint32_t& bmask, uint64_t& rs_first) {
  const uint64_t p0 = tile_start + (uint64_t)threadIdx.x * ITEMS;
  uint8_t prev = 0;
  if (p0 > 0 && p0 - 1 < N) prev = in[p0 - 1];
  bmask = 0;
  uint32_t last = 0;     // (tile-relative index of my last boundary)+1
  if (ITEMS == 16) load16(in, N, p0, reinterpret_cast<uint8_t (&)[16]>(b));
  else {
#pragma unroll
    for (int j = 0; j < ITEMS; j++) b[j] = p0 + j < N ? in[p0 + j] : 0;
  }
#pragma unroll
  for (int j = 0; j < ITEMS; j++) {
    const uint64_t p = p0 + j;
    const bool bd = p < N && (p == 0 || b[j] != prev);
    if (bd) { bmask |= 1u << j; last = (uint32_t)(p - tile_start) + 1u; }
    prev = b[j];
  }
  const uint32_t im = block_incl_max<BLOCK>(last, smem);
  smem[16 + threadIdx.x] = im;
  __syncthreads();
  const uint32_t ex = threadIdx.x ? smem[16 + threadIdx.x - 1] : 0u;
  __syncthreads();
  rs_first = ex ? tile_start + ex - 1 : carry;
}

// ---- P1: per tile first / last boundary (global position + 1; 0 = none)
__global__ __launch_bounds__(256) void rle_tile_summary(const uint8_t* __restrict__ in, uint64_t N, uint64_t* __restrict__ fb, uint64_t* __restrict__ lb, uint32_t t0) {
  __shared__ uint32_t smin[4], smax[4];
  const uint32_t tile = t0 + blockIdx.x;
  const uint64_t tile_start = (uint64_t)tile * RT, p0 = tile_start + (uint64_t)threadIdx.x * 16;
  uint8_t prev = 0;
  if (p0 > 0 && p0 - 1 < N) prev = in[p0 - 1];
  uint32_t first = 0xFFFFFFFFu, last = 0;
  uint8_t b[16];
  load16(in, N, p0, b);
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const uint64_t p = p0 + j;
    if (p < N) {
      const uint8_t c = b[j];
      if (p == 0 || c != prev) { const uint32_t rel = (uint32_t)(p - tile_start); if (first == 0xFFFFFFFFu) first = rel; last = rel + 1; }
      prev = c;
    }
  }
  first = wave_min(first); last = wave_max(last);
  if (lane_id() == 0) { smin[wave_id()] = first; smax[wave_id()] = last; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t f = smin[0], l = smax[0];
    for (int i = 1; i < 4; i++) { f = smin[i] < f ? smin[i] : f; l = smax[i] > l ? smax[i] : l; }
    fb[tile] = f == 0xFFFFFFFFu ? 0 : tile_start + f + 1;
    lb[tile] = l ? tile_start + l : 0;     // (tile_start + l-1) + 1
  }
}

// ---- P2: run_start_in[t] = start of the run containing the tile's first byte (from earlier tiles);
//          next_bnd[t] = first boundary in later tiles (N if none): rle_scanb_* below (three-phase scans)

// ---- P3: per-tile emitted byte count under the global chunking, plus two small tables per 256-byte subtile
//          (used by the speculative boundary search): subpre = emitted bytes of the tile before the subtile,
//          dmod = chunk phase (offset in run mod 255) of the subtile's first byte
__global__ __launch_bounds__(256) void rle_tile_count(const uint8_t* __restrict__ in, uint64_t N, const uint64_t* __restrict__ run_start_in,
                                                      uint64_t* __restrict__ gt, uint16_t* __restrict__ subpre, uint8_t* __restrict__ dmod, uint32_t t0) {
  __shared__ uint32_t smem[256 + 16];
  const uint32_t tile = t0 + blockIdx.x;
  const uint64_t tile_start = (uint64_t)tile * RT;
  uint8_t b[16]; uint32_t bm; uint64_t rs;
  run_starts<256, 16>(in, N, tile_start, run_start_in[tile], smem, b, bm, rs);
  const uint64_t p0 = tile_start + (uint64_t)threadIdx.x * 16;
  if ((threadIdx.x & 15) == 0) dmod[(size_t)tile * 16 + (threadIdx.x >> 4)] = (uint8_t)(p0 < N ? (((bm & 1u) ? 0ull : p0 - rs) % 255) : 0);
  uint32_t cnt = 0;
  // offset in the run mod 255: one 64-bit remainder per thread, then +1 per byte (0 at a run start)
  uint32_t dp = p0 < N ? (uint32_t)((p0 - rs) % 255) : 0u;
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const uint64_t p = p0 + j;
    if (p < N) {
      if ((bm >> j) & 1u) dp = 0;
      cnt += dp < 3 ? 1u : dp == 3 ? 2u : 0u;
      dp = dp == 254 ? 0u : dp + 1u;
    }
  }
  uint32_t tot;
  const uint32_t ex = block_excl_sum<256>(cnt, smem, tot);
  if ((threadIdx.x & 15) == 0) subpre[(size_t)tile * 16 + (threadIdx.x >> 4)] = (uint16_t)ex;
  if (threadIdx.x == 0) gt[tile] = tot;
}

// ---- P4: exclusive prefix sum (u64) over tiles, one workgroup; gt[Tn] = total
// ---- the tile scans (P2, and the exclusive prefix of the per-tile emitted counts) as three-phase scans over chunks of
//      1024 tiles (reduce -> scan of the chunk aggregates -> apply), so that their time does not grow with the stream (a multi-GPU job replicates this pass on every rank).
//      agg: [0..nch) chunk max of lb, [nch..2nch) chunk max of the inverted fb, [2nch..3nch) chunk sum of gt
constexpr uint32_t SC = 1024;
__global__ __launch_bounds__(1024) void rle_scanb_reduce(const uint64_t* __restrict__ fb, const uint64_t* __restrict__ lb, uint32_t Tn,
                                                         unsigned long long* __restrict__ agg, uint32_t nch) {
  __shared__ unsigned long long sm[16];
  const uint32_t i = blockIdx.x * SC + threadIdx.x;
  const unsigned long long l = i < Tn ? lb[i] : 0ull, f = i < Tn ? fb[i] : 0ull;
  const unsigned long long ml = block_incl_max<1024>(l, sm);
  const unsigned long long mf = block_incl_max<1024>(f ? (NONE64 - f) : 0ull, sm);
  if (threadIdx.x == 1023) { agg[blockIdx.x] = ml; agg[nch + blockIdx.x] = mf; }
}
// single workgroup: exclusive prefix max of agg[0..nch), exclusive SUFFIX max of agg[nch..2nch)
__global__ __launch_bounds__(1024) void rle_scanb_mid(unsigned long long* __restrict__ agg, uint32_t nch) {
  __shared__ unsigned long long sm[16];
  __shared__ unsigned long long arr[1024];
  unsigned long long carry = 0;
  for (uint32_t base = 0; base < nch; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const unsigned long long v = i < nch ? agg[i] : 0ull;
    const unsigned long long im = block_incl_max<1024>(v, sm);
    arr[threadIdx.x] = im;
    __syncthreads();
    const unsigned long long prev = threadIdx.x ? arr[threadIdx.x - 1] : 0ull, cmx = arr[1023];
    __syncthreads();
    if (i < nch) agg[i] = prev > carry ? prev : carry;
    carry = cmx > carry ? cmx : carry;
  }
  unsigned long long scarry = 0;
  const uint32_t rounds = (nch + 1023) / 1024;
  for (uint32_t r = 0; r < rounds; r++) {
    const uint32_t base = (rounds - 1 - r) * 1024;
    const uint32_t i = base + (1023 - threadIdx.x);
    const unsigned long long v = i < nch ? agg[nch + i] : 0ull;
    const unsigned long long im = block_incl_max<1024>(v, sm);
    arr[threadIdx.x] = im;
    __syncthreads();
    const unsigned long long prev = threadIdx.x ? arr[threadIdx.x - 1] : 0ull, cmx = arr[1023];
    __syncthreads();
    if (i < nch) agg[nch + i] = prev > scarry ? prev : scarry;
    scarry = cmx > scarry ? cmx : scarry;
  }
}
// carry[0] = (last boundary in front of the scanned tile range) + 1, 0 = none; carry[1] = first boundary behind the range, N = none
// (a rank of a multi-GPU job scans only its share of the tiles: rle_probe finds the two)
__global__ __launch_bounds__(1024) void rle_scanb_apply(uint64_t* __restrict__ fb, uint64_t* __restrict__ lb, uint32_t Tn,
                                                        const unsigned long long* __restrict__ agg, uint32_t nch, const uint64_t* __restrict__ carry_io) {
  __shared__ unsigned long long sm[16];
  __shared__ unsigned long long arr[1024];
  {                                                              // run start of each tile's first byte: exclusive prefix max of lb
    const uint32_t i = blockIdx.x * SC + threadIdx.x;
    const unsigned long long carry = agg[blockIdx.x];
    const unsigned long long v = i < Tn ? lb[i] : 0ull;
    const unsigned long long im = block_incl_max<1024>(v, sm);
    arr[threadIdx.x] = im;
    __syncthreads();
    const unsigned long long prev = threadIdx.x ? arr[threadIdx.x - 1] : 0ull;
    __syncthreads();
    if (i < Tn) { unsigned long long e = prev > carry ? prev : carry; const unsigned long long c0 = carry_io[0]; e = e > c0 ? e : c0; lb[i] = e ? e - 1 : 0ull; }
  }
  {                                                              // next boundary after each tile: exclusive suffix min of fb
    const uint32_t i = blockIdx.x * SC + (1023 - threadIdx.x);
    const unsigned long long scarry = agg[nch + blockIdx.x];
    const unsigned long long f = i < Tn ? fb[i] : 0ull;
    const unsigned long long im = block_incl_max<1024>(f ? (NONE64 - f) : 0ull, sm);
    arr[threadIdx.x] = im;
    __syncthreads();
    const unsigned long long prev = threadIdx.x ? arr[threadIdx.x - 1] : 0ull;
    __syncthreads();
    if (i < Tn) { const unsigned long long e = prev > scarry ? prev : scarry; fb[i] = e ? (NONE64 - e) - 1 : carry_io[1]; }
  }
}
__global__ __launch_bounds__(1024) void rle_scanc_reduce(const uint64_t* __restrict__ gt, uint32_t Tn, unsigned long long* __restrict__ agg, uint32_t nch) {
  __shared__ unsigned long long sm[16];
  const uint32_t i = blockIdx.x * SC + threadIdx.x;
  const unsigned long long t = block_sum<1024>((unsigned long long)(i < Tn ? gt[i] : 0ull), sm);
  if (threadIdx.x == 0) agg[2 * nch + blockIdx.x] = t;
}
__global__ __launch_bounds__(1024) void rle_scanc_mid(unsigned long long* __restrict__ agg, uint32_t nch, uint64_t* __restrict__ gt, uint32_t Tn) {
  __shared__ unsigned long long sm[16];
  unsigned long long carry = 0;
  for (uint32_t base = 0; base < nch; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    unsigned long long v = i < nch ? agg[2 * nch + i] : 0ull, tot;
    const unsigned long long ex = block_excl_sum<1024>(v, sm, tot);
    if (i < nch) agg[2 * nch + i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) gt[Tn] = carry;
}
__global__ __launch_bounds__(1024) void rle_scanc_apply(uint64_t* __restrict__ gt, uint32_t Tn, const unsigned long long* __restrict__ agg, uint32_t nch) {
  __shared__ unsigned long long sm[16];
  const uint32_t i = blockIdx.x * SC + threadIdx.x;
  unsigned long long v = i < Tn ? gt[i] : 0ull, tot;
  const unsigned long long ex = block_excl_sum<1024>(v, sm, tot);
  if (i < Tn) gt[i] = agg[2 * nch + blockIdx.x] + ex;
}

// in-tile helper for the walk (1024 threads x 4 positions): inclusive prefix of c at each position.
// Returns via `firstpos` the smallest position p in the tile with gbase + incl(p) >= target (NONE64 if none),
// and via `upto_sum` the sum of c over positions < upto.
__device__ void walk_tile_eval(const uint8_t* __restrict__ in, uint64_t N, uint64_t tile_start, uint64_t carry, uint64_t gbase,
                               uint64_t target, uint64_t upto, uint32_t* smem, unsigned long long* sh64,
                               uint64_t& firstpos, uint32_t& upto_sum) {
  uint8_t b[4]; uint32_t bm; uint64_t rs;
  run_starts<1024, 4>(in, N, tile_start, carry, smem, b, bm, rs);
  const uint64_t p0 = tile_start + (uint64_t)threadIdx.x * 4;
  uint32_t c[4], cnt = 0, below = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint64_t p = p0 + j;
    c[j] = 0;
    if (p < N) {
      if ((bm >> j) & 1u) rs = p;
      const uint32_t dp = (uint32_t)((p - rs) % 255);
      c[j] = dp < 3 ? 1u : dp == 3 ? 2u : 0u;
      cnt += c[j];
      if (p < upto) below += c[j];
    }
  }
  uint32_t tot;
  uint32_t ex = block_excl_sum<1024>(cnt, smem, tot);
  upto_sum = block_sum<1024>(below, smem);
  if (threadIdx.x == 0) sh64[0] = NONE64;
  __syncthreads();
  unsigned long long mine = NONE64;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint64_t p = p0 + j;
    ex += c[j];
    if (p < N && mine == NONE64 && c[j] && gbase + ex >= target) mine = p;
  }
  if (mine != NONE64) atomicMin(&sh64[0], mine);
  __syncthreads();
  firstpos = sh64[0];
  __syncthreads();
}

// ---- W: block boundaries.  ONE workgroup, serial over blocks.
// Speculative boundary search: from a block start s0 that coincides with a run boundary (fresh chunking == global
// chunking), the next boundaries are where G reaches G(s0) + j*cap.  Group j (32 lanes) looks boundary j up
// (search on G_tile, then the subtile table, then the <= 256 bytes of the subtile).  A boundary is "clean" when G hits
// the target exactly and the next block again starts on a run boundary; blocks up to the first unclean boundary
// are exact; the next round then starts inside a run (see rle_walk).
struct SpecOut { uint64_t e; uint32_t clean; uint32_t eof; uint32_t len_last; };
// The search for ONE target is done by a group of 32 lanes: 32-ary search on G_tile (3 dependent loads instead of the
// 15 of a binary search), one ballot over the 16 subtile prefixes, and the 256-byte subtile scanned 8 bytes per lane with two
// 32-lane scans (last run start, emitted-byte prefix).  A round of the walk costs a handful of memory round trips.
__device__ SpecOut spec_boundary_group(const uint8_t* __restrict__ in, uint64_t N, uint32_t Tn, const uint64_t* __restrict__ gt,
                                       const uint16_t* __restrict__ subpre, const uint8_t* __restrict__ dmod, uint64_t target, uint64_t prev_target) {
  const uint32_t l = threadIdx.x & 31u, hs = threadIdx.x & 32u;        // lane in the group; the group's half of the wave ballot
  SpecOut o; o.e = N; o.clean = 0; o.eof = 0; o.len_last = 0;
  const uint64_t gtot = gt[Tn];
  if (gtot < target) { o.eof = 1; o.len_last = gtot > prev_target ? (uint32_t)(gtot - prev_target) : 0u; return o; }
  uint32_t lo = 0, hi = Tn;                       // gt[lo] < target <= gt[hi]
  while (hi - lo > 1) {
    const uint32_t step = (hi - lo + 31u) / 32u;
    const uint64_t idx = (uint64_t)lo + (uint64_t)l * step;
    const bool pred = idx < hi && gt[idx] < target;
    const uint32_t cnt = (uint32_t)__builtin_popcount((uint32_t)(__ballot(pred) >> hs));     // ones are a prefix (gt is monotone), lane 0 is one
    const uint32_t nlo = lo + (cnt - 1u) * step;
    hi = nlo + step < hi ? nlo + step : hi;
    lo = nlo;
  }
  const uint64_t need = target - gt[lo];          // 1 .. tile count
  const uint32_t sp = l < 16 ? (uint32_t)subpre[(size_t)lo * 16 + l] : 0u;
  const bool pq = l < 16 && (l == 0 || (uint64_t)sp < need);
  const uint32_t sb = (uint32_t)__builtin_popcount((uint32_t)(__ballot(pq) >> hs)) - 1u;
  const uint32_t acc0 = (uint32_t)__shfl((int)sp, (int)sb, 32);
  const uint32_t d0 = dmod[(size_t)lo * 16 + sb];
  const uint64_t p0 = (uint64_t)lo * RT + (uint64_t)sb * 256;
  const uint64_t pb = p0 + 8u * l;
  uint8_t b[8];
  if (pb + 8 <= N && (((uintptr_t)(in + pb)) & 7u) == 0) {
    const uint64_t v = *reinterpret_cast<const uint64_t*>(in + pb);
#pragma unroll
    for (int t = 0; t < 8; t++) b[t] = (uint8_t)(v >> (8 * t));
  } else {
#pragma unroll
    for (int t = 0; t < 8; t++) b[t] = pb + t < N ? in[pb + t] : 0;
  }
  const uint32_t prevb = (uint32_t)__shfl_up((int)b[7], 1, 32);
  uint32_t rsmask = 0; int my_last = -1;
#pragma unroll
  for (int t = 0; t < 8; t++) {
    const uint32_t i = 8u * l + t;
    const uint32_t pv = t ? b[t - 1] : prevb;
    if (i > 0 && pb + t < N && b[t] != pv) { rsmask |= 1u << t; my_last = (int)i; }
  }
  int incl = my_last;
#pragma unroll
  for (int d = 1; d < 32; d <<= 1) { const int x = __shfl_up(incl, d, 32); if ((int)l >= d && x > incl) incl = x; }
  int rs = __shfl_up(incl, 1, 32);
  if (l == 0) rs = -1;
  uint32_t cs[8], sum = 0;
#pragma unroll
  for (int t = 0; t < 8; t++) {
    const uint32_t i = 8u * l + t;
    if ((rsmask >> t) & 1u) rs = (int)i;
    uint32_t dm = rs >= 0 ? i - (uint32_t)rs : d0 + i;
    dm = dm >= 255u ? dm - 255u : dm;                                   // d0 <= 254, i <= 255: one subtract is "% 255"
    dm = dm >= 255u ? dm - 255u : dm;
    const uint32_t c = pb + t < N ? (dm < 3 ? 1u : dm == 3 ? 2u : 0u) : 0u;
    sum += c; cs[t] = sum;
  }
  uint32_t incl2 = sum;
#pragma unroll
  for (int d = 1; d < 32; d <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)incl2, d, 32); if ((int)l >= d) incl2 += x; }
  const uint32_t excl = incl2 - sum;
  const uint32_t rel = (uint32_t)(need - acc0);                          // >= 1
  const uint32_t hm = (uint32_t)(__ballot(incl2 >= rel) >> hs);
  if (!hm) { o.e = p0 + 256 < N ? p0 + 256 : N; return o; }             // not reached (tables inconsistent): unclean
  const int fl = __builtin_ctz(hm);
  int tt = 8; uint32_t accv = 0, cv = 0;
#pragma unroll
  for (int t = 7; t >= 0; t--) if (excl + cs[t] >= rel) { tt = t; accv = excl + cs[t]; cv = b[t]; }
  tt = __shfl(tt, fl, 32); accv = (uint32_t)__shfl((int)accv, fl, 32); cv = (uint32_t)__shfl((int)cv, fl, 32);
  const uint64_t pe = p0 + 8u * (uint32_t)fl + (uint32_t)tt;
  o.e = pe + 1;
  bool next_fresh = pe + 1 >= N || in[pe + 1] != (uint8_t)cv;
  if (!next_fresh) {
    // the next block starts inside a run -- but a run of at most three bytes is all literals under either chunking (the one
    // counted from the run start and the one that starts afresh with the block), so the boundaries behind it are where the
    // speculation expects them: as good as a block start on a run boundary.  (text: "ll", "ee" at a block end)
    const uint8_t b = (uint8_t)cv;
    const uint32_t left = (pe >= 1 && in[pe - 1] == b) ? ((pe >= 2 && in[pe - 2] == b) ? 2u : 1u) : 0u;
    const uint32_t right = 1u + ((pe + 2 < N && in[pe + 2] == b) ? ((pe + 3 < N && in[pe + 3] == b) ? 2u : 1u) : 0u);
    next_fresh = 1u + left + right <= 3u;
  }
  o.clean = (accv == rel) && next_fresh;
  return o;
}

constexpr uint32_t NSPEC = 32;        // boundaries speculated per round: one 32-lane group each
__device__ uint64_t g_walk_dbg[8];     // CJS_DEBUG: [0] speculative rounds [1] serial steps [2],[3] their 100 MHz ticks
__global__ __launch_bounds__(1024) void rle_walk(const uint8_t* __restrict__ in, uint64_t N, uint32_t cap, uint32_t Tn,
                                                 const uint64_t* __restrict__ run_start_in, const uint64_t* __restrict__ next_bnd,
                                                 const uint64_t* __restrict__ gt, const uint16_t* __restrict__ subpre, const uint8_t* __restrict__ dmod,
                                                 RleBlock* __restrict__ blocks, uint32_t max_blocks, uint32_t* __restrict__ nblocks_out) {
  __shared__ uint32_t smem[1024 + 16];
  __shared__ unsigned long long sh64[2];
  __shared__ unsigned long long spec_e[NSPEC];
  __shared__ uint32_t spec_first_bad;
  uint64_t s = 0;
  uint32_t k = 0;
  bool done = false;
  uint64_t dbg_n[2] = {0, 0}, dbg_t[2] = {0, 0}, dbg_p[3] = {0, 0, 0};
  while (s < N && k < max_blocks && !done) {
    const uint64_t t_in = wall_clock64();
    // A round starts at s.  If s is a run boundary, fresh chunking == global chunking from s on and boundary j is where
    // G reaches G(s) + (j+1)*cap.  If s lies inside a run, only that run is chunked differently: with r_end its end and
    // base = bytes it emits under fresh chunking, the same holds with G(s) replaced by G(r_end) - base - unless the
    // block fills up inside the run or the run reaches the end of the input (closed forms below).
    uint64_t G0, r_end0 = s, Gr0; uint32_t base0 = 0;
    if (s == 0 || in[s] != in[s - 1]) {
      // G(s): tile prefix + in-tile prefix (one cooperative tile evaluation)
      const uint32_t t0 = (uint32_t)(s / RT);
      uint64_t fp; uint32_t below;
      walk_tile_eval(in, N, (uint64_t)t0 * RT, run_start_in[t0], 0, NONE64, s, smem, sh64, fp, below);
      G0 = gt[t0] + below; Gr0 = G0;
    } else {
      // end of the run containing s
      const uint32_t ts = (uint32_t)(s / RT);
      const uint64_t tile_end = ((uint64_t)(ts + 1) * RT < N) ? (uint64_t)(ts + 1) * RT : N;
      if (threadIdx.x == 0) sh64[0] = NONE64;
      __syncthreads();
      {
        unsigned long long mine = NONE64;
        for (int j = 0; j < 4; j++) {
          const uint64_t p = (uint64_t)ts * RT + (uint64_t)threadIdx.x * 4 + j;
          if (p > s && p < tile_end && mine == NONE64 && in[p] != in[p - 1]) mine = p;
        }
        if (mine != NONE64) atomicMin(&sh64[0], mine);
      }
      __syncthreads();
      uint64_t r_end = sh64[0];
      __syncthreads();
      if (r_end == NONE64) r_end = next_bnd[ts];
      const uint64_t Lp = r_end - s;
      const uint64_t q = Lp / 255; const uint32_t rr = (uint32_t)(Lp % 255);
      const uint64_t g64 = 5 * q + (rr < 4 ? rr : 5u);
      if (g64 >= cap || r_end >= N) {
        uint64_t e; uint32_t len, base;
        if (g64 >= cap) {                       // the block fills up inside this run
          const uint32_t qq = cap / 5, rem = cap % 5;
          const uint64_t m = rem == 0 ? (uint64_t)255 * (qq - 1) + 4 : (uint64_t)255 * qq + rem;
          e = s + m; len = cap; base = cap;
        } else { base = (uint32_t)g64; e = N; len = base; }     // the run is the tail of the input
        if (threadIdx.x == 0) {
          RleBlock bd; bd.s = s; bd.e = e; bd.r_end = r_end; bd.Gr = 0; bd.len = len; bd.base = base;
          blocks[k] = bd;
        }
        k++;
        s = e;
        dbg_n[1]++; dbg_t[1] += wall_clock64() - t_in;
        if (len < cap) break;
        continue;
      }
      base0 = (uint32_t)g64; r_end0 = r_end;
      const uint32_t tr = (uint32_t)(r_end / RT);
      uint64_t fp; uint32_t below;
      walk_tile_eval(in, N, (uint64_t)tr * RT, run_start_in[tr], 0, NONE64, r_end, smem, sh64, fp, below);
      Gr0 = gt[tr] + below;
      G0 = Gr0 - base0;                         // >= one block's worth of emitted bytes: s > 0 here
    }
    {
      const uint64_t t_a = wall_clock64(); dbg_p[0] += t_a - t_in;
      const uint32_t j = threadIdx.x >> 5;               // group = speculated boundary
      if (threadIdx.x == 0) spec_first_bad = NSPEC;
      __syncthreads();
      SpecOut so = spec_boundary_group(in, N, Tn, gt, subpre, dmod, G0 + (uint64_t)(j + 1) * cap, G0 + (uint64_t)j * cap);
      const bool lead = (threadIdx.x & 31u) == 0;
      if (lead) spec_e[j] = so.e;
      const uint64_t t_b = wall_clock64(); dbg_p[1] += t_b - t_a;
      if (lead && (!so.clean || so.eof)) atomicMin(&spec_first_bad, j);
      __syncthreads();
      dbg_p[2] += wall_clock64() - t_b;
      const uint32_t m = spec_first_bad;                    // boundaries 0..m-1 are clean; boundary m is EOF or unclean
      // blocks k+j for j <= m (block j spans (boundary j-1, boundary j]); the one ending at boundary m is still exact
      if (lead && j <= m && k + j < max_blocks) {
        const uint64_t bs = j == 0 ? s : spec_e[j - 1];
        RleBlock bd; bd.s = bs; bd.r_end = bs; bd.base = 0; bd.Gr = G0 + (uint64_t)j * cap;
        if (j == 0) { bd.r_end = r_end0; bd.base = base0; bd.Gr = Gr0; }
        bool emit = true;
        if (so.eof) { bd.e = N; bd.len = so.len_last; emit = so.len_last > 0 && bs < N; }
        else { bd.e = so.e; bd.len = cap; }
        if (j < m || (j == m && m < NSPEC)) { if (emit) blocks[k + j] = bd; }
      }
      __syncthreads();
      // advance: count emitted blocks
      if (m < NSPEC) {
        // boundary m: EOF (stream ends) or unclean (the next round starts inside a run)
        const unsigned long long e_m = spec_e[m];
        // was block m emitted?  eof with zero length -> not
        const uint64_t tgt_prev = G0 + (uint64_t)m * cap;
        const bool eof_m = gt[Tn] < tgt_prev + cap;
        if (eof_m) { const bool has = gt[Tn] > tgt_prev && (m == 0 ? s : spec_e[m - 1]) < N; k += m + (has ? 1u : 0u); done = true; }
        else { k += m + 1; s = e_m; }
      } else { k += NSPEC; s = spec_e[NSPEC - 1]; }
      __syncthreads();
      dbg_n[0]++; dbg_t[0] += wall_clock64() - t_in;
    }
  }
  if (threadIdx.x == 0) { *nblocks_out = k; g_walk_dbg[0] = dbg_n[0]; g_walk_dbg[1] = dbg_n[1]; g_walk_dbg[2] = dbg_t[0]; g_walk_dbg[3] = dbg_t[1]; g_walk_dbg[4] = dbg_p[0]; g_walk_dbg[5] = dbg_p[1]; g_walk_dbg[6] = dbg_p[2]; }
}


// ---- multi-GPU jobs: a rank makes the tile tables of ITS share of the tiles only (tiles [t0, t1)).  What the tile scans
// would have carried in from the other tiles is found by looking at the input itself (every rank holds the stream): the last
// run boundary in front of tile t0 and the first one at or behind tile t1 -- nearly always inside the neighbouring tile; a
// giant run makes the probe walk on, 16 KiB per step.  carry[0] = (last boundary position) + 1 or 0, carry[1] = first
// boundary position or N (see rle_scanb_apply).  One workgroup.
__global__ __launch_bounds__(1024) void rle_probe(const uint8_t* __restrict__ in, uint64_t N, uint32_t t0, uint32_t t1, uint64_t* __restrict__ carry) {
  __shared__ unsigned long long sm[16];
  const uint64_t lo = (uint64_t)t0 * RT, hi = (uint64_t)t1 * RT < N ? (uint64_t)t1 * RT : N;
  unsigned long long back = 0;
  for (uint64_t wend = lo; wend > 0 && !back;) {                      // windows [wend - 16 KiB, wend), walking towards 0
    const uint64_t wbeg = wend > 16384 ? wend - 16384 : 0;
    const uint64_t p0 = wbeg + (uint64_t)threadIdx.x * 16;
    unsigned long long mine = 0;
    if (p0 < wend) {
      uint8_t b[16];
      load16(in, N, p0, b);
      uint8_t prev = p0 ? in[p0 - 1] : 0;
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const uint64_t p = p0 + j;
        if (p < wend && (p == 0 || b[j] != prev)) mine = p + 1;
        prev = b[j];
      }
    }
    back = block_incl_max<1024>(mine, sm);
    __syncthreads();
    if (threadIdx.x == 1023) sm[0] = back;
    __syncthreads();
    back = sm[0];
    __syncthreads();
    wend = wbeg;
  }
  unsigned long long fwd = N;
  for (uint64_t wbeg = hi; wbeg < N && fwd == N; wbeg += 16384) {
    const uint64_t p0 = wbeg + (uint64_t)threadIdx.x * 16;
    unsigned long long mine = 0;                                      // NONE64 - position of the first boundary, 0 = none
    if (p0 < N) {
      uint8_t b[16];
      load16(in, N, p0, b);
      uint8_t prev = p0 ? in[p0 - 1] : 0;
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const uint64_t p = p0 + j;
        if (p < N && !mine && (p == 0 || b[j] != prev)) mine = NONE64 - p;
        prev = b[j];
      }
    }
    const unsigned long long m = block_incl_max<1024>(mine, sm);
    __syncthreads();
    if (threadIdx.x == 1023) sm[0] = m;
    __syncthreads();
    if (sm[0]) fwd = NONE64 - sm[0];
    __syncthreads();
  }
  if (threadIdx.x == 0) { carry[0] = back; carry[1] = fwd; }
}
// the whole stream's tile tables from the ranks' shares, gathered back to back (share r = tiles [r*tpr, (r+1)*tpr) in the layout
// of Rle1Work::tile_share): one thread per tile
__global__ __launch_bounds__(256) void rle_tables_unpack(const uint8_t* __restrict__ recv, uint32_t tpr, uint32_t Tn, uint64_t* __restrict__ fb, uint64_t* __restrict__ lb,
                                                         uint64_t* __restrict__ gt, uint16_t* __restrict__ subpre, uint8_t* __restrict__ dmod) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= Tn) return;
  const uint32_t r = t / tpr, j = t - r * tpr;
  const uint8_t* base = recv + (size_t)r * Rle1Work::share_bytes(tpr);
  lb[t] = reinterpret_cast<const uint64_t*>(base)[j];
  fb[t] = reinterpret_cast<const uint64_t*>(base + (size_t)8 * tpr)[j];
  gt[t] = reinterpret_cast<const uint64_t*>(base + (size_t)16 * tpr)[j];
  const uint4* sp = reinterpret_cast<const uint4*>(base + (size_t)24 * tpr) + (size_t)2 * j;
  uint4* dp = reinterpret_cast<uint4*>(subpre + (size_t)t * 16);
  dp[0] = sp[0]; dp[1] = sp[1];
  reinterpret_cast<uint4*>(dmod + (size_t)t * 16)[0] = (reinterpret_cast<const uint4*>(base + (size_t)56 * tpr))[j];
}

// ---- R: materialise the RLE1 bytes of every block (grid = input tiles)
__global__ __launch_bounds__(256) void rle_materialize(const uint8_t* __restrict__ in, uint64_t N, uint32_t cap,
                                                       const uint64_t* __restrict__ run_start_in, const uint64_t* __restrict__ next_bnd,
                                                       const uint64_t* __restrict__ gt, const RleBlock* __restrict__ blocks,
                                                       const uint32_t* __restrict__ nblocks_p, uint32_t first, uint32_t count,
                                                       uint8_t* __restrict__ out) {
  __shared__ uint32_t smem[256 + 16];
  __shared__ uint32_t nb_first[256];
  const uint32_t nblocks = *nblocks_p < first + count ? *nblocks_p : first + count;   // blocks [first, nblocks)
  const uint64_t tile_start = (uint64_t)blockIdx.x * RT;
  if (nblocks <= first || tile_start >= blocks[nblocks - 1].e || tile_start + RT <= blocks[first].s) return;
  uint8_t b[16]; uint32_t bm; uint64_t rs;
  run_starts<256, 16>(in, N, tile_start, run_start_in[blockIdx.x], smem, b, bm, rs);
  const uint64_t p0 = tile_start + (uint64_t)threadIdx.x * 16;
  // global c and its exclusive prefix inside the tile
  uint32_t dpg[16], cnt = 0;
  {
    uint32_t dp = p0 < N ? (uint32_t)((p0 - rs) % 255) : 0u;      // offset in the run mod 255: one 64-bit remainder per thread
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint64_t p = p0 + j;
      dpg[j] = 255;
      if (p < N) {
        if ((bm >> j) & 1u) dp = 0;
        dpg[j] = dp;
        cnt += dp < 3 ? 1u : dp == 3 ? 2u : 0u;
        dp = dp == 254 ? 0u : dp + 1u;
      }
    }
  }
  uint32_t tot;
  const uint32_t exc = block_excl_sum<256>(cnt, smem, tot);
  // next boundary after each position: tile-relative first boundary per thread, exclusive suffix-min over threads
  const uint32_t myfirst = bm ? (uint32_t)threadIdx.x * 16u + (uint32_t)__builtin_ctz(bm) : 0xFFFFu;
  {
    const uint32_t q = 255 - threadIdx.x;                 // reversed thread order
    nb_first[q] = myfirst == 0xFFFFu ? 0u : 0x10000u - myfirst;
    __syncthreads();
    const uint32_t v = nb_first[threadIdx.x];
    const uint32_t im = block_incl_max<256>(v, smem);
    __syncthreads();
    nb_first[threadIdx.x] = im;
    __syncthreads();
  }
  const uint32_t rq = 255 - threadIdx.x;
  const uint32_t sfx = rq ? nb_first[rq - 1] : 0u;        // max over later threads of (0x10000 - first)
  const uint64_t next_after_me = sfx ? tile_start + (0x10000u - sfx) : next_bnd[blockIdx.x];
  // blocks overlapping this tile: first block with e > tile_start
  uint32_t lo = first, hi = nblocks - 1;
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (blocks[mid].e > tile_start) hi = mid; else lo = mid + 1; }
  const uint64_t tile_end = tile_start + RT;
  {
    // A tile strictly inside one block, behind the block's re-chunked first run: its output is one contiguous byte range
    // of the block.  It is put together in LDS and written with aligned 32-bit stores (the general path below stores byte
    // by byte, sixteen scattered bytes per lane: 0.26 ms per 100 MB, most of it waiting for partial-line writes).
    __shared__ __attribute__((aligned(16))) uint8_t stage[RT + RT / 4 + 64];
    const RleBlock bd = blocks[lo];
    const uint64_t off0 = (uint64_t)bd.base + (gt[blockIdx.x] - bd.Gr);
    const bool fast = tile_start >= bd.s && tile_start >= bd.r_end && tile_end < bd.e && tile_end <= N && off0 + tot + 1 < cap;
    if (fast) {
      uint32_t run = exc;
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const uint32_t dp = dpg[j];
        if (dp < 4) {
          stage[run] = b[j];
          if (dp == 3) {
            uint64_t re = next_after_me;               // run end = next boundary after p
            const uint32_t later = (j < 15) ? (bm >> (j + 1)) : 0u;
            if (later) re = p0 + j + 1 + (uint32_t)__builtin_ctz(later);
            uint64_t follow = re - (p0 + j + 1);
            if (follow > 251) follow = 251;
            stage[run + 1] = (uint8_t)follow;
          }
        }
        run += dp < 3 ? 1u : dp == 3 ? 2u : 0u;
      }
      __syncthreads();
      uint8_t* dst = out + (size_t)(lo - first) * cap + off0;
      const uint32_t head = (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u) < tot ? (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u) : tot;
      const uint32_t n4 = (tot - head) >> 2, tail0 = head + 4u * n4;
      if (threadIdx.x < head) dst[threadIdx.x] = stage[threadIdx.x];
      const uint32_t* st32 = reinterpret_cast<const uint32_t*>(stage);
      uint32_t* d32 = reinterpret_cast<uint32_t*>(dst + head);
      for (uint32_t i = threadIdx.x; i < n4; i += 256) {
        const uint32_t sb = head + 4u * i, w0 = st32[sb >> 2], w1 = st32[(sb >> 2) + 1];
        d32[i] = __builtin_amdgcn_alignbyte(w1, w0, sb & 3u);
      }
      if (threadIdx.x < tot - tail0) dst[tail0 + threadIdx.x] = stage[tail0 + threadIdx.x];
      return;
    }
  }
  for (uint32_t k = lo; k < nblocks; k++) {
    const RleBlock bd = blocks[k];
    if (bd.s >= tile_end) break;
    uint8_t* o = out + (size_t)(k - first) * cap;
    uint32_t run = exc;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint64_t p = p0 + j;
      const uint32_t cg = dpg[j] < 3 ? 1u : dpg[j] == 3 ? 2u : 0u;
      if (p < N && p >= bd.s && p < bd.e) {
        uint32_t dp; uint64_t off;
        if (p < bd.r_end) { const uint64_t d = p - bd.s; dp = (uint32_t)(d % 255); off = 5 * (d / 255) + (dp < 4 ? dp : 4); }
        else { dp = dpg[j]; off = (uint64_t)bd.base + (gt[blockIdx.x] + run - bd.Gr); }
        if (dp < 4 && off < cap) {
          o[off] = b[j];
          if (dp == 3 && off + 1 < cap) {
            // run end = next boundary after p
            uint64_t re = next_after_me;
            const uint32_t later = (j < 15) ? (bm >> (j + 1)) : 0u;
            if (later) re = p + 1 + (uint32_t)__builtin_ctz(later);
            uint64_t follow = re - (p + 1);
            if (follow > 251) follow = 251;
            if (off + 2 == cap) follow = 0;                    // count byte fills the block: stays 0 (Q2)
            o[off + 1] = (uint8_t)follow;
          }
        }
      }
      run += (p < N) ? cg : 0u;
    }
  }
}

// ---- CRC-32 (bzip2: MSB-first, poly 0x04c11db7, init ~0, final ~) of input [s,e) per block
// The CRC register after a string is linear in the string: CRC0(A|B) = CRC0(A) * x^(8|B|) + CRC0(B) in GF(2)[x]/P
// (CRC0 = zero initial value).  Every thread takes 64 bytes, multiplies its CRC0 by x^(8 * bytes behind its chunk)
// and the workgroup XORs the products; the powers come from tables built at compile time.
constexpr uint32_t CRC_SEG = 16384;   // bytes per segment: 256 threads x 64 bytes; segments are 16 KiB-aligned ADDRESS windows
constexpr uint32_t cgf_mul(uint32_t a, uint32_t b) {                          // a*b mod P, bit k = x^k
  uint32_t r = 0;
  for (int i = 31; i >= 0; i--) {
    r = (r << 1) ^ ((r & 0x80000000u) ? 0x04c11db7u : 0u);
    if ((b >> i) & 1u) r ^= a;
  }
  return r;
}
struct CrcTables {
  uint32_t tab[4][256];    // slicing-by-4: tab[k][i] = CRC0 of byte i followed by k zero bytes
  uint32_t pw64[256];      // x^(8*64*q)
  uint32_t px[64];         // x^(8*r)
};
constexpr CrcTables make_crc_tables() {
  CrcTables t{};
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t c = i << 24;
    for (int k = 0; k < 8; k++) c = (c & 0x80000000u) ? (c << 1) ^ 0x04c11db7u : (c << 1);
    t.tab[0][i] = c;
  }
  for (int k = 1; k < 4; k++)
    for (uint32_t i = 0; i < 256; i++) { const uint32_t p = t.tab[k - 1][i]; t.tab[k][i] = (p << 8) ^ t.tab[0][p >> 24]; }
  t.px[0] = 1u;
  for (int r = 1; r < 64; r++) t.px[r] = cgf_mul(t.px[r - 1], 0x100u);
  const uint32_t x64 = cgf_mul(t.px[63], 0x100u);
  t.pw64[0] = 1u;
  for (int q = 1; q < 256; q++) t.pw64[q] = cgf_mul(t.pw64[q - 1], x64);
  return t;
}
__device__ const CrcTables g_crc_tables = make_crc_tables();

__device__ __forceinline__ uint32_t gf_mul(uint32_t a, uint32_t b) {
  uint32_t r = 0;
#pragma unroll 4
  for (int i = 31; i >= 0; i--) {
    r = (r << 1) ^ ((r & 0x80000000u) ? 0x04c11db7u : 0u);
    if ((b >> i) & 1u) r ^= a;
  }
  return r;
}
__device__ uint32_t gf_xpow8(uint64_t nbytes) {                              // x^(8*nbytes) mod P
  uint32_t result = 1u, base = 0x100u;                                       // x^8
  while (nbytes) {
    if (nbytes & 1) result = gf_mul(result, base);
    base = gf_mul(base, base);
    nbytes >>= 1;
  }
  return result;
}
// x^(8*n) for n < 16384 from the tables
__device__ __forceinline__ uint32_t gf_xpow8_small(uint32_t n, const uint32_t* pw64, const uint32_t* px) {
  const uint32_t q = n >> 6, r = n & 63u;
  return r ? gf_mul(pw64[q], px[r]) : pw64[q];
}

__global__ __launch_bounds__(256) void rle_crc_partial(const uint8_t* __restrict__ in, const RleBlock* __restrict__ blocks,
                                                       const uint32_t* __restrict__ nblocks_p, uint32_t max_segs,
                                                       uint32_t first, uint32_t* __restrict__ seg_crc) {
  __shared__ uint32_t tab[4][256];
  __shared__ uint32_t pw64[256], px[64];
  __shared__ uint32_t seg[CRC_SEG / 4 + CRC_SEG / 64];      // dword d of the window lives at d + d/16 (bank-conflict-free 64-byte strides)
  __shared__ uint32_t part[4];
  const uint32_t k = first + blockIdx.y;
  if (k >= *nblocks_p) return;
  const RleBlock bd = blocks[k];
  const int tid = threadIdx.x;
  // absolute byte addresses; windows are aligned in the address space so that every staging load is a 16-byte aligned one
  const uint64_t A0 = (uint64_t)(uintptr_t)in, S = A0 + bd.s, E = A0 + bd.e;
  const uint64_t W0 = S & ~(uint64_t)(CRC_SEG - 1);
  if (W0 + (uint64_t)blockIdx.x * CRC_SEG >= E) return;
  for (int j = 0; j < 4; j++) tab[j][tid] = g_crc_tables.tab[j][tid];
  pw64[tid] = g_crc_tables.pw64[tid];
  if (tid < 64) px[tid] = g_crc_tables.px[tid];
  for (uint32_t sgi = blockIdx.x; W0 + (uint64_t)sgi * CRC_SEG < E; sgi += gridDim.x) {
    const uint64_t W = W0 + (uint64_t)sgi * CRC_SEG;
    const uint64_t lo = W > S ? W : S, hi = W + CRC_SEG < E ? W + CRC_SEG : E;      // bytes of the block in this window
    __syncthreads();
    for (int j = 0; j < 4; j++) {
      const uint32_t c = (uint32_t)j * 256u + tid;                 // 16-byte chunk of the window
      const uint64_t ca = W + (uint64_t)c * 16;
      if (ca + 16 > lo && ca < hi) {
        const uint4 v = *(const uint4*)(uintptr_t)ca;              // may include bytes outside [lo,hi): never used
        const uint32_t d = c * 4u, o = d + (d >> 4);
        seg[o] = v.x; seg[o + 1] = v.y; seg[o + 2] = v.z; seg[o + 3] = v.w;
      }
    }
    __syncthreads();
    const uint64_t ca = W + (uint64_t)tid * 64, cb = ca + 64;
    const uint64_t a = ca > lo ? ca : lo, b = cb < hi ? cb : hi;
    uint32_t crc = 0;
    if (a < b) {
      const uint32_t o = (uint32_t)tid * 17u;
      if (b - a == 64) {
#pragma unroll 4
        for (int i = 0; i < 16; i++) {
          const uint32_t x = crc ^ __builtin_bswap32(seg[o + i]);
          crc = tab[3][x >> 24] ^ tab[2][(x >> 16) & 0xff] ^ tab[1][(x >> 8) & 0xff] ^ tab[0][x & 0xff];
        }
      } else {
        for (uint64_t p = a; p < b; p++) {
          const uint32_t off = (uint32_t)(p - ca);
          const uint32_t byte = (seg[o + (off >> 2)] >> (8u * (off & 3u))) & 0xffu;
          crc = (crc << 8) ^ tab[0][((crc >> 24) ^ byte) & 0xff];
        }
      }
      const uint32_t behind = (uint32_t)(hi - b);
      if (behind) crc = gf_mul(crc, gf_xpow8_small(behind, pw64, px));
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) crc ^= __shfl_xor(crc, m, 64);
    if ((tid & 63) == 0) part[tid >> 6] = crc;
    __syncthreads();
    if (tid == 0) seg_crc[(size_t)blockIdx.y * max_segs + sgi] = part[0] ^ part[1] ^ part[2] ^ part[3];
  }
}

__global__ void rle_crc_final(const RleBlock* __restrict__ blocks, const uint32_t* __restrict__ nblocks_p, uint32_t max_segs,
                              uint32_t first, uint32_t count, const uint8_t* __restrict__ in, const uint32_t* __restrict__ seg_crc,
                              uint32_t* __restrict__ block_crc) {
  const uint32_t rel = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t k = first + rel;
  if (rel >= count || k >= *nblocks_p) return;
  const RleBlock bd = blocks[k];
  const uint64_t n = bd.e - bd.s;
  const uint64_t A0 = (uint64_t)(uintptr_t)in, S = A0 + bd.s, E = A0 + bd.e;
  const uint64_t W0 = S & ~(uint64_t)(CRC_SEG - 1);
  const uint32_t xfull = gf_mul(g_crc_tables.pw64[255], g_crc_tables.pw64[1]);     // x^(8*16384)
  uint32_t crc = 0;
  for (uint32_t sgi = 0; W0 + (uint64_t)sgi * CRC_SEG < E; sgi++) {
    const uint64_t W = W0 + (uint64_t)sgi * CRC_SEG;
    const uint64_t lo = W > S ? W : S, hi = W + CRC_SEG < E ? W + CRC_SEG : E;
    const uint32_t l = (uint32_t)(hi - lo);
    crc = gf_mul(crc, l == CRC_SEG ? xfull : gf_xpow8_small(l, g_crc_tables.pw64, g_crc_tables.px)) ^ seg_crc[(size_t)rel * max_segs + sgi];
  }
  crc ^= gf_mul(0xFFFFFFFFu, gf_xpow8(n));     // init = ~0 carried through n bytes
  block_crc[k] = ~crc;
}

__global__ void rle_block_lens(const RleBlock* __restrict__ blocks, uint32_t* nblocks_p, uint32_t* __restrict__ block_len) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x, nb = nblocks_p[0];
  if (k < nb) block_len[k] = blocks[k].len;
  if (k + 1 == nb) nblocks_p[1] = blocks[k].len;      // length of the last block: goes to the host with the count
}

// ------------------------------------------------------------------------------------------
size_t Rle1Work::bytes_needed(size_t max_in, uint32_t cap, size_t range_blocks) {
  const size_t Tn = (max_in + RT - 1) / RT + 2;
  const size_t maxb = max_blocks_for(max_in, cap);
  const size_t segs = max_segs_for(cap);
  size_t b = 0;
  auto add = [&](size_t n) { b += (n + 255) & ~(size_t)255; };
  add(Tn * 8); add(Tn * 8); add((Tn + 1) * 8); add(Tn * 16 * 2); add(Tn * 16); add(3 * (Tn / 1024 + 2) * 8);
  add(maxb * sizeof(RleBlock)); add(maxb * 4); add(maxb * 4); add(64);
  add((range_blocks ? range_blocks : maxb) * segs * 4);
  return b + 4096;
}
int Rle1Work::carve(Arena& a, size_t max_in_, uint32_t cap_, size_t range_blocks_) {
  max_in = max_in_; cap = cap_;
  const size_t Tn = (max_in + RT - 1) / RT + 2;
  max_blocks = (uint32_t)max_blocks_for(max_in, cap);
  max_segs = (uint32_t)max_segs_for(cap);
  fb = a.take<uint64_t>(Tn); lb = a.take<uint64_t>(Tn); gt = a.take<uint64_t>(Tn + 1);
  agg = a.take<unsigned long long>(3 * (Tn / 1024 + 2));
  subpre = a.take<uint16_t>(Tn * 16); dmod = a.take<uint8_t>(Tn * 16);
  blocks = a.take<RleBlock>(max_blocks); block_len = a.take<uint32_t>(max_blocks); block_crc = a.take<uint32_t>(max_blocks);
  nblocks = a.take<uint32_t>(16);
  range_blocks = (uint32_t)(range_blocks_ ? range_blocks_ : max_blocks);
  seg_crc = a.take<uint32_t>((size_t)range_blocks * max_segs);
  return seg_crc ? 0 : CJS_E_OUT_OF_MEMORY;
}

// CRC-32 of arbitrary byte ranges [s,e) of d_data (used by the decoder for the per-block output CRCs)
int crc_ranges(hipStream_t s, const uint8_t* d_data, const RleBlock* d_blocks, const uint32_t* d_nblocks, uint32_t count, uint32_t max_segs,
               uint32_t* d_seg_crc, uint32_t* d_crc_out) {
  if (!count) return 0;
  hipLaunchKernelGGL(rle_crc_partial, dim3(64, count), dim3(256), 0, s, d_data, d_blocks, d_nblocks, max_segs, 0u, d_seg_crc);
  hipLaunchKernelGGL(rle_crc_final, dim3((count + 63) / 64), dim3(64), 0, s, d_blocks, d_nblocks, max_segs, 0u, count, d_data, d_seg_crc, d_crc_out);
  CJS_HIP_TRY(hipGetLastError());
  return 0;
}

// Stage 0a, first half: the tile tables (run start carried into each tile, next boundary behind it, emitted bytes under the
// global chunking, subtile tables) of tiles [t0, t1) of d_in[0..N).  `share` null: into the workspace's own arrays (absolute
// tile index); else into a share buffer of Rle1Work::share_bytes(tpr) bytes that a multi-GPU job exchanges (tile t0 first).
int rle1_tiles(hipStream_t s, Rle1Work& w, const uint8_t* d_in, uint64_t N, uint32_t t0, uint32_t t1, uint8_t* share, uint32_t tpr) {
  if (N > w.max_in) return CJS_E_INVALID_ARG;
  const uint32_t Tn = (uint32_t)((N + RT - 1) / RT);
  if (t1 > Tn) t1 = Tn;
  if (t0 >= t1) return 0;
  uint64_t *fb = w.fb, *lb = w.lb, *gt = w.gt; uint16_t* subpre = w.subpre; uint8_t* dmod = w.dmod;
  if (share) {                                       // kernels index by absolute tile: tile t0 is entry 0 of the share
    if (t1 - t0 > tpr) return CJS_E_INVALID_ARG;
    lb = reinterpret_cast<uint64_t*>(share) - t0; fb = reinterpret_cast<uint64_t*>(share + (size_t)8 * tpr) - t0;
    gt = reinterpret_cast<uint64_t*>(share + (size_t)16 * tpr) - t0;
    subpre = reinterpret_cast<uint16_t*>(share + (size_t)24 * tpr) - (size_t)t0 * 16; dmod = share + (size_t)56 * tpr - (size_t)t0 * 16;
  }
  const uint32_t Tl = t1 - t0, nch = (Tl + SC - 1) / SC;
  uint64_t* carry = reinterpret_cast<uint64_t*>(w.nblocks + 4);      // two u64 behind the block count
  hipLaunchKernelGGL(rle_tile_summary, dim3(Tl), dim3(256), 0, s, d_in, N, fb, lb, t0);
  hipLaunchKernelGGL(rle_probe, dim3(1), dim3(1024), 0, s, d_in, N, t0, t1, carry);
  hipLaunchKernelGGL(rle_scanb_reduce, dim3(nch), dim3(1024), 0, s, fb + t0, lb + t0, Tl, w.agg, nch);
  hipLaunchKernelGGL(rle_scanb_mid, dim3(1), dim3(1024), 0, s, w.agg, nch);
  hipLaunchKernelGGL(rle_scanb_apply, dim3(nch), dim3(1024), 0, s, fb + t0, lb + t0, Tl, w.agg, nch, carry);
  hipLaunchKernelGGL(rle_tile_count, dim3(Tl), dim3(256), 0, s, d_in, N, lb, gt, subpre, dmod, t0);
  CJS_HIP_TRY(hipGetLastError());
  return 0;
}
// the gathered shares of all ranks (rank-major, share r = tiles [r*tpr, (r+1)*tpr)) -> the workspace's arrays
int rle1_tables_from_shares(hipStream_t s, Rle1Work& w, uint64_t N, const uint8_t* d_recv, uint32_t tpr) {
  const uint32_t Tn = (uint32_t)((N + RT - 1) / RT);
  if (!Tn) return 0;
  hipLaunchKernelGGL(rle_tables_unpack, dim3((Tn + 255) / 256), dim3(256), 0, s, d_recv, tpr, Tn, w.fb, w.lb, w.gt, w.subpre, w.dmod);
  CJS_HIP_TRY(hipGetLastError());
  return 0;
}
// Stage 0a, second half: prefix of the emitted bytes over the tiles and the boundary walk.  Leaves descriptors / lengths on the
// device, returns the number of blocks in *nblocks_host (syncs the stream).
int rle1_walk_run(hipStream_t s, Rle1Work& w, const uint8_t* d_in, uint64_t N, uint32_t* nblocks_host, uint32_t* last_len_host) {
  if (N > w.max_in) return CJS_E_INVALID_ARG;
  if (N == 0) { *nblocks_host = 0; if (last_len_host) *last_len_host = 0; CJS_HIP_TRY(hipMemsetAsync(w.nblocks, 0, 4, s)); return 0; }
  const uint32_t Tn = (uint32_t)((N + RT - 1) / RT);
  const uint32_t nch = (Tn + SC - 1) / SC;
  hipLaunchKernelGGL(rle_scanc_reduce, dim3(nch), dim3(1024), 0, s, w.gt, Tn, w.agg, nch);
  hipLaunchKernelGGL(rle_scanc_mid, dim3(1), dim3(1024), 0, s, w.agg, nch, w.gt, Tn);
  hipLaunchKernelGGL(rle_scanc_apply, dim3(nch), dim3(1024), 0, s, w.gt, Tn, w.agg, nch);
  hipLaunchKernelGGL(rle_walk, dim3(1), dim3(1024), 0, s, d_in, N, w.cap, Tn, w.lb, w.fb, w.gt, w.subpre, w.dmod, w.blocks, w.max_blocks, w.nblocks);
  hipLaunchKernelGGL(rle_block_lens, dim3((w.max_blocks + 255) / 256), dim3(256), 0, s, w.blocks, w.nblocks, w.block_len);
  CJS_HIP_TRY(hipGetLastError());
  if (!w.h_n) CJS_HIP_TRY(hipHostMalloc((void**)&w.h_n, 16));
  CJS_HIP_TRY(hipMemcpyAsync(w.h_n, w.nblocks, 8, hipMemcpyDeviceToHost, s));
  CJS_HIP_TRY(hipStreamSynchronize(s));
  *nblocks_host = w.h_n[0];
  if (last_len_host) *last_len_host = w.h_n[0] ? w.h_n[1] : 0u;
  if (getenv("CJS_DEBUG")) {
    uint64_t d[8];
    if (hipMemcpyFromSymbol(d, HIP_SYMBOL(g_walk_dbg), sizeof d) == hipSuccess)
      fprintf(stderr, "[cjs rle] boundary walk: %llu speculative rounds %.1f us, %llu serial steps %.1f us, %u blocks; thread 0 of the rounds: tile eval %.1f us, own search %.1f us, wait for the slowest lane %.1f us\n", (unsigned long long)d[0], d[2] / 100.0,
              (unsigned long long)d[1], d[3] / 100.0, w.h_n[0], d[4] / 100.0, d[5] / 100.0, d[6] / 100.0);
  }
  return 0;
}
// Stage 0a: block boundaries of the whole stream d_in[0..N) on one GPU
int rle1_run(hipStream_t s, Rle1Work& w, const uint8_t* d_in, uint64_t N, uint32_t* nblocks_host, uint32_t* last_len_host) {
  if (N > w.max_in) return CJS_E_INVALID_ARG;
  if (N) CJS_TRY(rle1_tiles(s, w, d_in, N, 0u, (uint32_t)((N + RT - 1) / RT), nullptr, 0u));
  return rle1_walk_run(s, w, d_in, N, nblocks_host, last_len_host);
}

// Stage 0b: RLE1 bytes (d_blocks, block k at (k-first)*cap) and CRCs (block_crc[k], absolute) of blocks [first, first+count)
int rle1_finish(hipStream_t s, Rle1Work& w, const uint8_t* d_in, uint64_t N, uint32_t first, uint32_t count, uint8_t* d_blocks,
                hipStream_t side, hipEvent_t ev_fork, hipEvent_t ev_join) {
  if (N == 0 || count == 0) return 0;
  if (count > w.range_blocks) return CJS_E_INVALID_ARG;
  const uint32_t Tn = (uint32_t)((N + RT - 1) / RT);
  // the block CRCs read only the input and the block table: with a side stream they run beside the suffix sort
  // (ev_join is recorded at their end; the caller makes the packing stage wait for it)
  hipStream_t cs = s;
  if (side) {
    CJS_HIP_TRY(hipEventRecord(ev_fork, s));
    CJS_HIP_TRY(hipStreamWaitEvent(side, ev_fork, 0));
    cs = side;
  }
  hipLaunchKernelGGL(rle_materialize, dim3(Tn), dim3(256), 0, s, d_in, N, w.cap, w.lb, w.fb, w.gt, w.blocks, w.nblocks, first, count, d_blocks);
  hipLaunchKernelGGL(rle_crc_partial, dim3(64, count), dim3(256), 0, cs, d_in, w.blocks, w.nblocks, w.max_segs, first, w.seg_crc);
  hipLaunchKernelGGL(rle_crc_final, dim3((count + 63) / 64), dim3(64), 0, cs, w.blocks, w.nblocks, w.max_segs, first, count, d_in, w.seg_crc, w.block_crc);
  CJS_HIP_TRY(hipGetLastError());
  if (side) CJS_HIP_TRY(hipEventRecord(ev_join, side));
  return 0;
}

}  // namespace cjs
