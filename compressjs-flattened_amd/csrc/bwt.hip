// bwt.hip — batched suffix sorting / Burrows-Wheeler transform for gfx950.
//
// Replaces BWT.bwtransform2 (cyclic; J/Bzip2_joined_.js:928-971 -> SA_IS :730-857) and
// BWT.bwtransform (sentinel; J/BWTC_joined_.js:1125-1145) for ALL blocks of a batch at once.
// The reference runs SA-IS (serial induced sorting) per block.  A suffix array is unique, so
// any correct construction gives the same BWT; here it is prefix doubling (Manber-Myers /
// Larsson-Sadakane) expressed as data-parallel passes over all blocks' suffixes together:
//
//   round 0 : key = (block id, first 4 bytes)                      -> groups of depth 4
//   round r : key = (group ordinal, rank[i + h]) for suffixes in unresolved groups only,
//             h = 4, 8, 16, ...                                    -> depth doubles
//   each round: LSD radix sort (8-bit digits, LDS-staged buckets, wave64 match-any ranking)
//               -> regroup (flags + tile scan + apply) -> compaction of the still-unresolved set.
//
// Integer sort/scan only (no MFMA); HBM-bound on the radix scatter passes.
#include "cjs_internal.h"
#include "prims.hpp"
#include <stdlib.h>
#include <string.h>

namespace cjs {

struct Geom { uint32_t nb, stride, n_last; };
__device__ __forceinline__ uint32_t blk_len(const Geom& g, uint32_t blk) { return blk == g.nb - 1 ? g.n_last : g.stride; }

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with a private 4 MiB L2),
// so workgroup w = 8*j + x works on tile x*ceil(T/8) + j: every XCD walks ONE contiguous range of the
// suffix array, i.e. one block at a time, and that block's rank array (3.6 MB) stays in its L2 while the
// kernel scatters / gathers ranks at random positions of it.  Speed only; any mapping is correct.
__device__ __forceinline__ uint32_t xcd_tile(uint32_t wg, uint32_t T) {
  const uint32_t per = (T + 7u) >> 3;
  return (wg & 7u) * per + (wg >> 3);
}
__host__ __device__ __forceinline__ uint32_t xcd_grid(uint32_t T) { return ((T + 7u) >> 3) << 3; }

// ------------------------------------------------------------------------------------------
// LSD radix sort pass: histogram -> per-bin scan over tiles -> stable scatter
// ------------------------------------------------------------------------------------------
// Segments: the array is nseg runs of `stride` elements (the last one n_last) that are sorted independently in the
// same launches (round 1 of the suffix sort: one segment per block, so the block id needs no digit passes of its
// own).  Tiles never straddle segments: segment s owns tiles [s*tps, (s+1)*tps).  A plain sort is one segment.
struct SegGeom { uint32_t nseg, stride, n_last, tps; };
struct TileRef { uint32_t seg, off, nvalid; uint64_t base; };
__device__ __forceinline__ TileRef tile_ref(const SegGeom& sg, uint32_t tile) {
  TileRef t;
  t.seg = tile / sg.tps;
  t.off = (tile - t.seg * sg.tps) * RS_TILE;
  const uint32_t sn = t.seg + 1 == sg.nseg ? sg.n_last : sg.stride;
  t.nvalid = t.off < sn ? (sn - t.off < RS_TILE ? sn - t.off : RS_TILE) : 0u;
  t.base = (uint64_t)t.seg * sg.stride + t.off;
  return t;
}
// Key source of the first pass of round 1: keys are made on the fly from the block bytes (no key array is ever
// written for them).  key = leading nsym symbols | block parity above them (adjacent blocks must not compare equal);
// cyclic: bytes, wrapping; sentinel: 9-bit symbols byte+1, 0 = past the end.  value = position in the block.
struct GenSrc { const uint8_t* T; int cyclic, nsym, packed; };
// packed records (cyclic round 1): ONE u64 per suffix = 5 bytes (bits 63..24) | block parity (bit 20) | position in the block
// (bits 19..0): a radix pass moves 8 B per suffix each way instead of 12, and there is no value array.  The first phase sorts by
// bytes 2..6 of the suffix (packed == 2, the only packed form)
constexpr int PK_SHIFT = 20, PK_KEY_LO = 24;
constexpr uint32_t PK_POS_MASK = (1u << PK_SHIFT) - 1u;
// records of the two-phase sort after bwt_phase2_records: byte0 . byte1 (63..48) | rank of the class of bytes 2..6 (47..28) | the byte IN
// FRONT of the suffix (27..20) | position (19..0).  The group key is key >> PK2_GSHIFT; block boundaries are taken from the slot
// number (no parity bit).  The byte in front is what the BWT emits for the suffix: it rides along (later in the top byte of val[])
// so that the regroup kernels write BWT bytes without gathering them from the text.
constexpr int PK2_GSHIFT = 28, PK2_PREV_SHIFT = 20, VAL_PREV_SHIFT = 24;
constexpr uint32_t GEN_PAD = 8;
// stages the tile's bytes (+GEN_PAD lookahead) in LDS; returns the byte offset of the tile's first byte inside tb (< 4):
// tiles that do not touch the end of their block are copied as aligned 32-bit words from the aligned-down address
__device__ __forceinline__ uint32_t gen_stage(const GenSrc& gs, const SegGeom& sg, const TileRef& t, uint8_t* tb) {
  const uint32_t sn = t.seg + 1 == sg.nseg ? sg.n_last : sg.stride;
  const uint8_t* src = gs.T + (size_t)t.seg * sg.stride;
  if (t.off + RS_TILE + GEN_PAD <= sn) {
    const uintptr_t a = (uintptr_t)(src + t.off);
    const uint32_t* al = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
    uint32_t* tw = reinterpret_cast<uint32_t*>(tb);
    for (uint32_t i = threadIdx.x; i < (RS_TILE + GEN_PAD) / 4 + 1; i += 256) tw[i] = al[i];
    return (uint32_t)(a & 3);
  }
  for (uint32_t i = threadIdx.x; i < RS_TILE + GEN_PAD; i += 256) {
    uint32_t p = t.off + i;
    if (p >= sn) p = gs.cyclic ? p % sn : sn - 1;      // sentinel: never used (masked by position)
    tb[i] = t.nvalid ? src[p] : 0;
  }
  return 0;
}
__device__ __forceinline__ uint64_t gen_key(const GenSrc& gs, const SegGeom& sg, const TileRef& t, const uint8_t* tb, uint32_t tb0, uint32_t loc) {
  // the 8 bytes at tb[tb0 + loc ..] from three aligned LDS words
  const uint32_t* tw = reinterpret_cast<const uint32_t*>(tb) + ((tb0 + loc) >> 2);
  const uint32_t a0 = tw[0], a1 = tw[1], a2 = tw[2], sh = (tb0 + loc) & 3u;
  const uint32_t lo = __builtin_amdgcn_alignbyte(a1, a0, sh), hi = __builtin_amdgcn_alignbyte(a2, a1, sh);   // byte j of (hi:lo) = tb[loc + j]
  uint64_t k = 0;
  if (gs.cyclic) {
    k = ((uint64_t)__builtin_bswap32(lo) << 24) | (uint64_t)(__builtin_bswap32(hi) >> 8);     // 7 bytes, first byte on top
    if (gs.packed) return ((k & 0xFFFFFFFFFFull) << PK_KEY_LO) | ((uint64_t)(t.seg & 1u) << PK_SHIFT) | (uint64_t)(t.off + loc);  // bytes 2..6 (two-phase sort)
    k >>= 8 * (7 - gs.nsym);
    k |= (uint64_t)(t.seg & 1u) << (8 * gs.nsym);
  } else {
    const uint32_t sn = t.seg + 1 == sg.nseg ? sg.n_last : sg.stride;
    const uint64_t both = ((uint64_t)hi << 32) | lo;
    for (int j = 0; j < gs.nsym; j++) k = (k << 9) | (t.off + loc + j < sn ? (uint32_t)((both >> (8 * j)) & 0xFFu) + 1u : 0u);
    k |= (uint64_t)(t.seg & 1u) << (9 * gs.nsym);
  }
  return k;
}

// Per-tile digit counts, tile-major: hist[tile * 256 + digit] (one coalesced 1 KB row per workgroup; the digit-major
// layout of round 1 cost a 64-byte memory transaction per 4-byte counter on both sides).
// LDS atomics of one instruction that meet in one address OR in one bank are done one after the other, and text digits
// are skewed (a fifth of the lanes carry a space; a pass over a sorted byte has all 64 lanes on one counter): each wave
// counts into HR = 8 copies of its histogram picked by lane & 7 and laid out digit * 8 + copy, so the copies of a digit
// sit in eight different banks.  100 M keys, MI355X: 237 us (text digit) / 344 us (sorted digit) with one copy per wave;
// copies 1 KB apart (same bank) 141 / 190 us with two and slower again with more; interleaved copies 135 us for both,
// against 125 us for the same loads without any atomic.
constexpr int HR = 8;
template <typename K, bool GEN>
__global__ __launch_bounds__(256) void rs_hist(const K* __restrict__ keys, SegGeom sg, GenSrc gs, int shift,
                                               uint32_t* __restrict__ hist, uint32_t T) {
  __shared__ uint32_t h[4 * 256 * HR];
  __shared__ __attribute__((aligned(16))) uint8_t tb[GEN ? RS_TILE + GEN_PAD + 16 : 16];
  const int tid = threadIdx.x;
  uint32_t* hw = h + (tid >> 6) * 256 * HR + (tid & (HR - 1));          // this lane's copy in this wave's histogram
#pragma unroll
  for (int i = 0; i < 4 * HR; i++) h[i * 256 + tid] = 0;
  const uint32_t tile = blockIdx.x;
  const TileRef t = tile_ref(sg, tile);
  uint32_t tb0 = 0;
  if (GEN) tb0 = gen_stage(gs, sg, t, tb);
  __syncthreads();
  if (GEN) {
#pragma unroll 4
    for (int it = 0; it < 16; it++) {
      const uint32_t loc = (uint32_t)it * 256 + tid;
      if (loc < t.nvalid) atomicAdd(&hw[((uint32_t)(gen_key(gs, sg, t, tb, tb0, loc) >> shift) & 255u) * HR], 1u);
    }
  } else if (t.nvalid) {
    // all sixteen loads are issued before the first atomic (written as one loop the compiler waits for each load in turn)
    K k[16];
#pragma unroll
    for (int it = 0; it < 16; it++) {
      const uint32_t loc = (uint32_t)it * 256 + tid;
      k[it] = keys[t.base + (loc < t.nvalid ? loc : t.nvalid - 1u)];
    }
#pragma unroll
    for (int it = 0; it < 16; it++)
      if ((uint32_t)it * 256 + tid < t.nvalid) atomicAdd(&hw[((uint32_t)((uint64_t)k[it] >> shift) & 255u) * HR], 1u);
  }
  __syncthreads();
  uint32_t sum = 0;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int r = 0; r < HR; r++) sum += h[i * 256 * HR + tid * HR + r];
  hist[(size_t)tile * 256 + tid] = sum;
}

// The same counts from a byte per key: the scatter pass in front left the NEXT digit of every key it moved in dig[] (same
// index as the key), so this pass reads 1 B per key instead of 8.  Aligned 32-bit loads over the tile's byte range.
// With delta > 0 the bytes are the block TEXT (cyclic form): the first pass's digit of the suffix at position p is the text byte
// at p + delta, so a tile's digits are the text bytes [off + delta, off + delta + nvalid) of its block -- no keys are built for
// the count (rs_hist<GEN>: 130 us); the one tile per block whose range wraps around the block end counts byte by byte.
constexpr int HB = 4;   // histogram copies per wave (the zeroing and folding of 4 * 256 * HB counters is most of this kernel's LDS traffic; 59-82 us per 100 M keys with 4, 72-85 with 8, 75-130 with 2)
__global__ __launch_bounds__(256) void rs_hist_bytes(const uint8_t* __restrict__ dig, SegGeom sg, uint32_t* __restrict__ hist, uint32_t delta) {
  __shared__ uint32_t h[4 * 256 * HB];
  const int tid = threadIdx.x;
  uint32_t* hw = h + (tid >> 6) * 256 * HB + (tid & (HB - 1));
#pragma unroll
  for (int i = 0; i < 4 * HB; i++) h[i * 256 + tid] = 0;
  const uint32_t tile = blockIdx.x;
  const TileRef t = tile_ref(sg, tile);
  const uint32_t sn = t.seg + 1 == sg.nseg ? sg.n_last : sg.stride;
  const bool wraps = delta && t.nvalid && t.off + delta + t.nvalid > sn;         // (wave-uniform)
  if (wraps) {
    __syncthreads();
    const uint8_t* tx = dig + (size_t)t.seg * sg.stride;
    for (uint32_t e = tid; e < t.nvalid; e += 256) atomicAdd(&hw[(uint32_t)tx[(t.off + e + delta) % sn] * HB], 1u);
  }
  const uintptr_t a0 = (uintptr_t)(dig + t.base + delta), a1 = a0 + (wraps ? 0u : t.nvalid);
  const uint32_t* al = reinterpret_cast<const uint32_t*>(a0 & ~(uintptr_t)3);
  uint32_t wv[5];
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const uint32_t wi = (uint32_t)j * 256u + tid;
    wv[j] = (t.nvalid && (uintptr_t)(al + wi) < a1) ? al[wi] : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const uintptr_t wa = (uintptr_t)(al + ((uint32_t)j * 256u + tid));
#pragma unroll
    for (int b = 0; b < 4; b++)
      if (wa + b >= a0 && wa + b < a1) atomicAdd(&hw[((wv[j] >> (8 * b)) & 255u) * HB], 1u);
  }
  __syncthreads();
  uint32_t sum = 0;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int r = 0; r < HB; r++) sum += h[i * 256 * HB + tid * HB + r];
  hist[(size_t)tile * 256 + tid] = sum;
}

// one workgroup per segment: exclusive scan over the segment's tiles of every digit's count (thread = digit; the rows are
// read coalesced and the loads of a batch are independent, only the running sums are a chain); digit totals -> bintot
__global__ __launch_bounds__(256) void rs_scan_bins(uint32_t* __restrict__ hist, uint32_t tps, uint32_t* __restrict__ bintot) {
  uint32_t* p = hist + (size_t)blockIdx.x * tps * 256 + threadIdx.x;
  uint32_t carry = 0;
  uint32_t i = 0;
  for (; i + 8 <= tps; i += 8) {
    uint32_t v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = p[(size_t)(i + j) * 256];
#pragma unroll
    for (int j = 0; j < 8; j++) { p[(size_t)(i + j) * 256] = carry; carry += v[j]; }
  }
  for (; i < tps; i++) { const uint32_t v = p[(size_t)i * 256]; p[(size_t)i * 256] = carry; carry += v; }
  bintot[(size_t)blockIdx.x * 256 + threadIdx.x] = carry;
}
// plain (one-segment) sorts have up to tens of thousands of tiles: three-phase scan, chunks of SB_CHUNK tiles
constexpr uint32_t SB_CHUNK = 64;
__global__ __launch_bounds__(256) void rs_scan_chunk_sum(const uint32_t* __restrict__ hist, uint32_t T, uint32_t* __restrict__ csum) {
  const uint32_t t0 = blockIdx.x * SB_CHUNK, t1 = t0 + SB_CHUNK < T ? t0 + SB_CHUNK : T;
  uint32_t acc = 0;
  for (uint32_t t = t0; t < t1; t++) acc += hist[(size_t)t * 256 + threadIdx.x];
  csum[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void rs_scan_chunk_mid(uint32_t* __restrict__ csum, uint32_t nch, uint32_t* __restrict__ bintot) {
  uint32_t carry = 0;
  for (uint32_t c = 0; c < nch; c++) { const uint32_t v = csum[(size_t)c * 256 + threadIdx.x]; csum[(size_t)c * 256 + threadIdx.x] = carry; carry += v; }
  bintot[threadIdx.x] = carry;
}
__global__ __launch_bounds__(256) void rs_scan_chunk_apply(uint32_t* __restrict__ hist, uint32_t T, const uint32_t* __restrict__ csum) {
  const uint32_t t0 = blockIdx.x * SB_CHUNK, t1 = t0 + SB_CHUNK < T ? t0 + SB_CHUNK : T;
  uint32_t carry = csum[(size_t)blockIdx.x * 256 + threadIdx.x];
  for (uint32_t t = t0; t < t1; t++) { const uint32_t v = hist[(size_t)t * 256 + threadIdx.x]; hist[(size_t)t * 256 + threadIdx.x] = carry; carry += v; }
}

// lanes of the wave that carry the same 8-bit digit.  Per bit: m = the bit spread over a word (v_bfe_i32), one ballot, and
// the lanes whose bit differs from mine are ballot ^ m, folded into the running OR by one v_bitop3 per half (q | (m ^ bal) =
// table 0xde): four VALU instructions per bit (the select form took nine).
__device__ __forceinline__ uint64_t match_any8(uint32_t d) {
  uint32_t qlo = 0, qhi = 0;
#pragma unroll
  for (int b = 0; b < 8; b++) {
    const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)d, b, 1);      // 0 or ~0
    const uint64_t bal = __ballot(m != 0);
    qlo = __builtin_amdgcn_bitop3_b32(m, qlo, (uint32_t)bal, 0xde);
    qhi = __builtin_amdgcn_bitop3_b32(m, qhi, (uint32_t)(bal >> 32), 0xde);
  }
  return ~(((uint64_t)qhi << 32) | qlo);
}

// One ranking step of a wave: the number of keys with this lane's digit that the wave has seen before this lane's key (earlier
// steps, then lower lanes of this step); wc = the wave's 256 running digit counts.
// (Measured and dropped: taking the peer mask out of LDS instead of eight ballots -- every lane ORs its lane bit into the
// digit's 64-bit word, reads it back, the first lane clears it: 15 VALU + 5 DS instructions instead of ~70 VALU, bit-exact,
// but same-address LDS atomics are done one lane after the other and text digits put a dozen lanes on one word:
// rs_scatter 345 vs 320 us, tile sorter 1.07 ms both ways.)
__device__ __forceinline__ uint32_t rank_step(uint32_t d, uint32_t* __restrict__ wc) {
  const uint64_t peers = match_any8(d);
  const uint32_t prior = wc[d];
  const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
  __builtin_amdgcn_wave_barrier();
  if (r == 0) wc[d] = prior + (uint32_t)__popcll(peers);
  __builtin_amdgcn_wave_barrier();
  return prior + r;
}

template <typename K, bool GEN, bool NOVAL>
__global__ __launch_bounds__(256) void rs_scatter(const K* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                  K* __restrict__ kout, uint32_t* __restrict__ vout, SegGeom sg, GenSrc gs, int shift,
                                                  const uint32_t* __restrict__ hist, uint32_t T, const uint32_t* __restrict__ bintot,
                                                  uint8_t* __restrict__ dig /* next digit of every key, at the key's new index (or null) */) {
  __shared__ __attribute__((aligned(16))) K skey[RS_TILE + 2];
  __shared__ uint32_t sval[NOVAL ? 1 : RS_TILE];
  __shared__ uint32_t wcnt[4][256];
  __shared__ uint32_t goff[256];
  __shared__ uint32_t sm[4];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const uint32_t tile = blockIdx.x;
  const TileRef t = tile_ref(sg, tile);
  const uint64_t base = t.base;
  const uint32_t nvalid = t.nvalid;
  for (int i = tid; i < 1024; i += 256) (&wcnt[0][0])[i] = 0;
  K k[16];
  uint32_t v[16];
  uint32_t rk[16];
  if (GEN) {
    uint8_t* tb = reinterpret_cast<uint8_t*>(skey);     // skey is not written before the ranking is done
    const uint32_t tb0 = gen_stage(gs, sg, t, tb);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16; s++) {
      const uint32_t loc = (uint32_t)w * 1024u + (uint32_t)s * 64u + lane;
      const bool ok = loc < nvalid;
      k[s] = ok ? (K)gen_key(gs, sg, t, tb, tb0, loc) : (K)~(K)0;
      v[s] = ok ? t.off + loc : 0u;
    }
  } else {
#pragma unroll
    for (int s = 0; s < 16; s++) {
      const uint32_t loc = (uint32_t)w * 1024u + (uint32_t)s * 64u + lane;
      const bool ok = loc < nvalid;
      k[s] = ok ? kin[base + loc] : (K)~(K)0;           // (non-temporal loads here: no change, 12.69 vs 12.69 ms per step)
      v[s] = (ok && !NOVAL) ? vin[base + loc] : 0u;
    }
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 16; s++) rk[s] = rank_step((uint32_t)(k[s] >> shift) & 255u, wcnt[w]);
  __syncthreads();
  {
    const uint32_t c0 = wcnt[0][tid], c1 = wcnt[1][tid], c2 = wcnt[2][tid], c3 = wcnt[3][tid];
    uint32_t total;
    const uint32_t ex = block_excl_sum<256>(c0 + c1 + c2 + c3, sm, total);
    uint32_t tot2;
    const uint32_t binbase = block_excl_sum<256>(bintot[(size_t)t.seg * 256 + tid], sm, tot2);
    wcnt[0][tid] = ex; wcnt[1][tid] = ex + c0; wcnt[2][tid] = ex + c0 + c1; wcnt[3][tid] = ex + c0 + c1 + c2;
    goff[tid] = t.seg * sg.stride + binbase + hist[(size_t)tile * 256 + tid] - ex;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 16; s++) {
    const uint32_t d = (uint32_t)(k[s] >> shift) & 255u;
    const uint32_t p = wcnt[w][d] + rk[s];
    skey[p] = k[s];
    if (!NOVAL) sval[p] = v[s];
  }
  __syncthreads();
#pragma unroll 4
  for (int it = 0; it < 16; it++) {
    const uint32_t j = (uint32_t)it * 256u + tid;
    if (j < nvalid) {
      const K kk = skey[j];
      const uint32_t dst = goff[(uint32_t)(kk >> shift) & 255u] + j;
      kout[dst] = kk;
      if (!NOVAL) vout[dst] = sval[j];
      if (dig) dig[dst] = (uint8_t)((uint64_t)kk >> (shift + 8));
    }
  }
}

// ------------------------------------------------------------------------------------------
// suffix-sort kernels
// ------------------------------------------------------------------------------------------
// round 0 keys: (block id, first nsym symbols).  cyclic: bytes wrap; sentinel: 9-bit symbols, 0 = past the end
__global__ __launch_bounds__(256) void bwt_init_keys(const uint8_t* __restrict__ T, Geom g, int cyclic, int nsym, uint32_t M,
                                                     uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
  for (uint64_t a = (uint64_t)blockIdx.x * 256 + threadIdx.x; a < M; a += (uint64_t)gridDim.x * 256) {
    const uint32_t blk = (uint32_t)(a / g.stride), i = (uint32_t)(a - (uint64_t)blk * g.stride), n = blk_len(g, blk);
    const uint8_t* t = T + (size_t)blk * g.stride;
    uint64_t k = 0;
    if (cyclic) {
      uint32_t x = i;
      for (int j = 0; j < nsym; j++) { k = (k << 8) | t[x]; if (++x == n) x = 0; }
      k |= (uint64_t)blk << (8 * nsym);
    } else {
      for (int j = 0; j < nsym; j++) { const uint32_t x = i + j; k = (k << 9) | (x < n ? (uint32_t)t[x] + 1u : 0u); }
      k |= (uint64_t)blk << (9 * nsym);
    }
    key[a] = k; val[a] = i;                // slot a of round 1 is sorted position a: no pos[] yet
  }
}

// Two-sweep scheduling of the round-1 rank scatter: a block's 3.6 MB rank region does not survive in a 4 MiB L2 next to
// the streamed arrays (PMC: ~64 B of memory traffic per 4-byte store), so the tiles are walked twice, storing first the ranks
// of the lower half of every block's text positions, then the upper half (1.8 MB per sweep and block; the tiles of a block
// stay on one XCD, see xcd_tile).
// PMC: WRITE_SIZE of the kernel 6.66 -> 2.67 GB per 100 MB step; its time 1.64 -> 1.35 ms (the second sweep re-reads and
// re-derives the group structure).  The same scheduling applied to the gather / scatter of rounds >= 2 over the compacted
// arrays (per-block start table from a binary search) was measured SLOWER (gather 0.84 -> 1.07 ms, scatter 1.24 -> 1.84 ms
// in round 2) and is not kept.
struct HalfMap { uint32_t halves, stride; };      // halves 2: the two sweeps are the launches SWEEP 1 and SWEEP 2 of bwt_apply

// round r>=1 keys: (group ordinal, rank of suffix i+h)
__global__ __launch_bounds__(256) void bwt_gather_keys(Geom g, int cyclic, uint32_t A, uint32_t h, const uint32_t* __restrict__ R,
                                                       const uint32_t* __restrict__ val, const uint32_t* __restrict__ pos,
                                                       const uint32_t* __restrict__ gord, uint64_t* __restrict__ key, uint32_t T) {
  const uint32_t tile = xcd_tile(blockIdx.x, T);
  if (tile >= T) return;
  const uint64_t base = (uint64_t)tile * RS_TILE;
  const uint32_t nvalid = (uint32_t)((uint64_t)A - base < RS_TILE ? (uint64_t)A - base : RS_TILE);
  // three batches of sixteen independent loads each (streams, then the rank gathers, then the stores): as one loop the
  // compiler waits for every gather before it issues the next
  uint32_t p16[16], v16[16], g16[16], kk[16];
#pragma unroll
  for (int it = 0; it < 16; it++) {
    const uint32_t e = (uint32_t)it * 256 + threadIdx.x;
    const uint64_t a = base + (e < nvalid ? e : nvalid - 1u);
    p16[it] = __builtin_nontemporal_load(pos + a); v16[it] = __builtin_nontemporal_load(val + a); g16[it] = __builtin_nontemporal_load(gord + a);
  }
#pragma unroll
  for (int it = 0; it < 16; it++) {
    const uint32_t blk = p16[it] / g.stride, n = blk_len(g, blk);
    uint32_t j = (v16[it] & PK_POS_MASK) + h;  // (both below 2^31; the top byte of val[] may carry the byte in front of the suffix)
    if (cyclic) { if (j >= n) j %= n; }
    const bool past = !cyclic && j >= n;
    kk[it] = R[(size_t)blk * g.stride + (past ? 0 : j)] + 1u;
    if (past) kk[it] = 0u;
  }
#pragma unroll
  for (int it = 0; it < 16; it++) {
    const uint32_t e = (uint32_t)it * 256 + threadIdx.x;
    if (e < nvalid) __builtin_nontemporal_store(((uint64_t)g16[it] << 20) | kk[it], key + base + e);
  }
}

// per 4096-tile: #surviving elements, #surviving group heads, (last new-head index)+1
__global__ __launch_bounds__(256) void bwt_flags(const uint64_t* __restrict__ key, uint32_t A, uint32_t* __restrict__ tile_cnt, uint32_t T, int gshift,
                                                 uint32_t* __restrict__ big_flag, uint32_t first_stride /* round 1: block stride (a block start is a head), else 0 */) {
  __shared__ uint64_t sk[RS_TILE + 2];
  __shared__ uint32_t sm[4];
  __shared__ uint32_t hword[64];          // per 64-slot word: holds a new group head
  const uint64_t base = (uint64_t)blockIdx.x * RS_TILE;
  const uint32_t nvalid = (uint32_t)((uint64_t)A - base < RS_TILE ? (uint64_t)A - base : RS_TILE);
  uint32_t surv = 0, heads = 0, last = 0;
  const uint64_t bnd = first_stride ? (base + first_stride - 1u) / first_stride * first_stride : ~0ull;      // the block start in [base, base + tile]
  // the tile's keys go through LDS so that every global load is issued up front (neighbours come from LDS, not from three
  // loads per element that the compiler waits for one by one)
  uint64_t k[16];
#pragma unroll
  for (int it = 0; it < 16; it++) {
    const uint32_t loc = (uint32_t)it * 256 + threadIdx.x;
    k[it] = key[base + (loc < nvalid ? loc : nvalid - 1u)] >> gshift;            // packed round-1 records: the low bits are the position
  }
  if (threadIdx.x == 0) sk[0] = base ? key[base - 1] >> gshift : 0;
  if (threadIdx.x == 64) sk[RS_TILE + 1] = base + RS_TILE < A ? key[base + RS_TILE] >> gshift : 0;
#pragma unroll
  for (int it = 0; it < 16; it++) sk[1u + (uint32_t)it * 256 + threadIdx.x] = k[it];
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 16; it++) {
    const uint32_t loc = (uint32_t)it * 256 + threadIdx.x;
    const uint64_t a = base + loc;
    if (loc < nvalid) {
      const bool nh = a == 0 || a == bnd || sk[loc] != k[it];
      const bool nx = a + 1 == A || a + 1 == bnd || sk[loc + 2] != k[it];
      const bool single = nh && nx;
      surv += !single;
      heads += nh && !single;
      if (nh) last = (uint32_t)a + 1u;
    }
    const uint64_t hb = __ballot(loc < nvalid && (base + loc == 0 || base + loc == bnd || sk[loc] != k[it]));
    if (lane_id() == 0) hword[it * 4 + wave_id()] = hb != 0;
  }
  __syncthreads();
  // a group of more than 1023 slots covers a whole aligned run of 512 slots: when every such run holds a head, no group is too
  // large for the tile sorters of the next round (the host then skips their deferral bookkeeping)
  if (threadIdx.x < 8 && base + (threadIdx.x + 1u) * 512u <= A) {
    uint32_t any = 0;
    for (int j = 0; j < 8; j++) any |= hword[threadIdx.x * 8 + j];
    if (!any) atomicOr(big_flag, 1u);
  }
  surv = block_sum<256>(surv, sm);
  heads = block_sum<256>(heads, sm);
  last = wave_max(last);
  if (lane_id() == 0) sm[wave_id()] = last;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t m = sm[0]; for (int i = 1; i < 4; i++) m = sm[i] > m ? sm[i] : m;
    tile_cnt[blockIdx.x] = surv; tile_cnt[T + blockIdx.x] = heads; tile_cnt[2 * (size_t)T + blockIdx.x] = m;
  }
}

// single workgroup: exclusive sums of [0],[1]; exclusive prefix-max of [2]; totals -> counters[0..1]
// (eight consecutive tiles per thread, one round of workgroup scans per 8 K tiles: 78 instead of 44 us for the 24 K tiles of
// round 1 -- the strided accesses cost more than the barriers saved)
__global__ __launch_bounds__(1024) void bwt_scan_tiles(uint32_t* __restrict__ tile_cnt, uint32_t T, uint32_t* __restrict__ counters,
                                                      uint32_t* __restrict__ host_mirror /* pinned host memory, read after the stream sync */) {
  __shared__ uint32_t sm[16];
  __shared__ uint32_t mx[1024];
  uint32_t c0 = 0, c1 = 0, cm = 0;
  for (uint32_t base = 0; base < T; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const bool ok = i < T;
    uint32_t t0, t1;
    const uint32_t v0 = ok ? tile_cnt[i] : 0u, v1 = ok ? tile_cnt[T + i] : 0u, v2 = ok ? tile_cnt[2 * (size_t)T + i] : 0u;
    const uint32_t e0 = block_excl_sum<1024>(v0, sm, t0);
    const uint32_t e1 = block_excl_sum<1024>(v1, sm, t1);
    const uint32_t im = block_incl_max<1024>(v2, sm);
    mx[threadIdx.x] = im;
    __syncthreads();
    const uint32_t prev = threadIdx.x ? mx[threadIdx.x - 1] : 0u;
    const uint32_t chunk_max = mx[1023];
    __syncthreads();
    if (ok) {
      tile_cnt[i] = c0 + e0; tile_cnt[T + i] = c1 + e1;
      tile_cnt[2 * (size_t)T + i] = prev > cm ? prev : cm;
    }
    c0 += t0; c1 += t1; cm = chunk_max > cm ? chunk_max : cm;
  }
  if (threadIdx.x == 0) { counters[0] = c0; counters[1] = c1; host_mirror[0] = c0; host_mirror[1] = c1; host_mirror[4] = counters[4]; counters[4] = 0; }      // [4]: a group may exceed 1023 slots (bwt_flags / sweep 1)
}

// Wave-uniform: global index of the first slot of the class (key >> shift) that runs into slot `base` from the left, inside
// [seg0, base] (the block is sorted, seg0 = its first slot).  Looks at the 64 slots in front of `base` -- nearly always
// enough -- and otherwise makes a 64-way search for the lower bound of the class in the block.  Needs all 64 lanes.
__device__ __forceinline__ uint64_t class_head_before(const uint64_t* __restrict__ key, uint64_t seg0, uint64_t base, int shift, int lane) {
  if (base == seg0) return base;
  const uint64_t target = key[base] >> shift;
  const uint64_t off = base - seg0;
  const uint32_t back = off < 64u ? (uint32_t)off : 64u;                // slots in front of base, inside the block
  const bool eq = (uint32_t)lane < back && (key[base - 1u - (uint32_t)lane] >> shift) == target;
  const uint64_t ne = ~__ballot(eq);
  const uint32_t m = ne ? (uint32_t)__builtin_ctzll(ne) : 64u;          // slots base-1 .. base-m carry the class
  if (m < 64u || off <= 64u) return base - m;
  uint64_t lo = seg0, hi = base - 64u;                                  // key[hi] is known to carry the class
  while (hi > lo) {
    const uint64_t len = hi - lo, step = (len + 63u) / 64u, idx = lo + (uint64_t)lane * step;
    const bool hit = idx >= hi || (key[idx] >> shift) == target;
    const uint64_t hits = __ballot(hit);
    const uint32_t j = hits ? (uint32_t)__builtin_ctzll(hits) : 64u;    // first probe inside the class (64: none, it starts behind the last probe)
    if (!j) { hi = lo; break; }
    const uint64_t nhi = lo + (uint64_t)j * step;
    lo += (uint64_t)(j - 1u) * step + 1u;
    if (j < 64u && nhi < hi) hi = nhi;
  }
  return hi;
}

// regroup: new ranks -> R (scattered 4-byte stores), singletons -> SA, survivors compacted into the next
// active arrays.  Fully lane-striped: the per-element prefix quantities come from 4096-bit masks
// (wave ballots) + a 64-word scan, so every global access of a wave touches consecutive addresses.
// SWEEP 0: one launch behind a counting pass (bwt_flags + bwt_scan_tiles).
// SWEEP 1 / 2 (round 1, packed records; see HalfMap): two launches with the tile scan between them, and NO counting pass
// in front -- sweep 1 stores the ranks of the lower half-blocks and leaves the tile counts that bwt_flags would have made
// (it finds the one class head it cannot see by search: class_head_before); sweep 2 stores the upper half and, with the
// scanned counts, everything else.
template <bool FIRST, bool PACKED, int SWEEP = 0>      // FIRST: round 1 - slot a is sorted position a, and there is no previous grouping
__global__ __launch_bounds__(256) void bwt_apply(const uint64_t* __restrict__ key, const uint32_t* __restrict__ val,
                                                 const uint32_t* __restrict__ pos, uint32_t A, Geom g,
                                                 uint32_t* tile_cnt, uint32_t T,
                                                 uint32_t* __restrict__ R, uint32_t* __restrict__ SA,
                                                 uint32_t* __restrict__ nval, uint32_t* __restrict__ npos, uint32_t* __restrict__ ngord, HalfMap hm_,
                                                 const uint8_t* __restrict__ Tx, uint8_t* __restrict__ U, uint32_t* __restrict__ big_flag,
                                                 int gs1 /* FIRST: group key = key >> gs1 */, int carried /* the byte in front of a suffix rides in its record / val */) {
  __shared__ uint64_t sk[RS_TILE + 2];
  __shared__ uint64_t m_nh[64], m_sg[64], m_oh[64];
  __shared__ uint32_t wp_s[64], wp_h[64], wp_head[64];
  __shared__ uint32_t carry_s;
  const uint32_t tile = xcd_tile(blockIdx.x, T), half = SWEEP == 2 ? 1u : 0u;
  if (tile >= T) return;
  const uint32_t hsplit = hm_.stride >> 1;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const uint64_t base = (uint64_t)tile * RS_TILE;
  const uint32_t nvalid = (uint32_t)((uint64_t)A - base < RS_TILE ? (uint64_t)A - base : RS_TILE);
  // every streaming load of the tile is issued up front (one loop of load + LDS store makes the compiler wait per load)
  uint32_t sbase = 0, hbase = 0, carry = 0;
  if (SWEEP != 1) { sbase = tile_cnt[tile]; hbase = tile_cnt[T + tile]; carry = tile_cnt[2 * (size_t)T + tile]; }
  else if (w == 0) {
    const uint64_t seg0 = (uint64_t)((uint32_t)base / g.stride) * g.stride;     // (round 1: slot = sorted position)
    const uint32_t c = (uint32_t)class_head_before(key, seg0, base, gs1, lane) + 1u;
    if (lane == 0) carry_s = c;
  }
  uint32_t p16[FIRST ? 1 : 16], v16[PACKED ? 1 : 16];
  {
    uint64_t k16[16];
#pragma unroll
    for (int it = 0; it < 16; it++) {
      const uint32_t e = (uint32_t)it * 256u + tid;
      k16[it] = __builtin_nontemporal_load(key + base + (e < nvalid ? e : nvalid - 1u));     // streamed once: keep the L2 for R
    }
    if (!FIRST) {
#pragma unroll
      for (int it = 0; it < 16; it++) { const uint32_t e = (uint32_t)it * 256u + tid; p16[it] = __builtin_nontemporal_load(pos + base + (e < nvalid ? e : nvalid - 1u)); }
    }
    if (!PACKED) {
#pragma unroll
      for (int it = 0; it < 16; it++) { const uint32_t e = (uint32_t)it * 256u + tid; v16[it] = __builtin_nontemporal_load(val + base + (e < nvalid ? e : nvalid - 1u)); }
    }
    if (tid == 0) sk[0] = base ? key[base - 1] : ~0ull;
    if (tid == 64) sk[RS_TILE + 1] = base + RS_TILE < A ? key[base + RS_TILE] : ~0ull;
#pragma unroll
    for (int it = 0; it < 16; it++) {
      const uint32_t e = (uint32_t)it * 256u + tid;
      sk[e + 1] = e < nvalid ? k16[it] : ~0ull;
    }
  }
  __syncthreads();
  const int GS = FIRST ? gs1 : 0;                // packed records: the group key sits above the position (and carried byte) bits
  const uint64_t bnd = FIRST ? (base + g.stride - 1u) / g.stride * g.stride : ~0ull;      // round 1: a block start is a head (slot = sorted position)
#pragma unroll 4
  for (int it = 0; it < 16; it++) {
    const uint32_t e = (uint32_t)it * 256u + tid;
    const uint64_t a = base + e;
    const bool ok = e < nvalid;
    const uint64_t k = sk[e + 1] >> GS;
    const bool nh = ok && (a == 0 || a == bnd || (sk[e] >> GS) != k);
    const bool nx = a + 1 == A || a + 1 == bnd || (sk[e + 2] >> GS) != k;
    const uint64_t mnh = __ballot(nh), msg = __ballot(nh && nx);
    if (lane == 0) { m_nh[it * 4 + w] = mnh; m_sg[it * 4 + w] = msg; }
    if (!FIRST) {
      const bool oh = ok && (a == 0 || (sk[e] >> 20) != (sk[e + 1] >> 20));      // head of a group of the previous round
      const uint64_t moh = __ballot(oh);
      if (lane == 0) m_oh[it * 4 + w] = moh;
    }
  }
  __syncthreads();
  if (w == 0) {                          // 64 mask words, one per lane
    const uint64_t mh = m_nh[lane], ms = m_sg[lane];
    const uint32_t first = (uint32_t)lane * 64u;
    const uint32_t nv = nvalid > first ? (nvalid - first < 64u ? nvalid - first : 64u) : 0u;
    const uint64_t vm = nv == 64 ? ~0ull : ((1ull << nv) - 1ull);
    const uint32_t sv = (uint32_t)__popcll(vm & ~ms), hd = (uint32_t)__popcll(mh & ~ms);
    const uint32_t lastrel = mh ? first + 63u - (uint32_t)__builtin_clzll(mh) + 1u : 0u;
    const uint32_t is = wave_incl_sum(sv), ih = wave_incl_sum(hd), im = wave_incl_max(lastrel);
    uint32_t em = __shfl_up(im, 1, 64);
    if (lane == 0) em = 0;
    wp_s[lane] = is - sv; wp_h[lane] = ih - hd; wp_head[lane] = em;
    if (SWEEP == 1 && lane == 63) {          // what bwt_flags counts: survivors, surviving heads, (last head index) + 1
      tile_cnt[tile] = is; tile_cnt[T + tile] = ih; tile_cnt[2 * (size_t)T + tile] = im ? (uint32_t)base + im : 0u;
    }
    if (SWEEP == 1) {                        // (and its test for groups of more than 1023 slots: an aligned run of 512 slots without a head)
      const uint64_t nz = __ballot(mh != 0);
      bool bad = false;
#pragma unroll
      for (int b = 0; b < 8; b++) bad |= base + (uint64_t)(b + 1) * 512u <= A && ((nz >> (8 * b)) & 0xFFull) == 0;
      if (bad && lane == 0) atomicOr(big_flag, 1u);
    }
  }
  __syncthreads();
  if (SWEEP == 1) carry = carry_s;
  const uint64_t lt = (1ull << lane) - 1ull, le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
  // BWT bytes of the suffixes this tile resolves (see the store below): all sixteen gathers are issued before the loop
  uint32_t ub[16];
#pragma unroll
  for (int it = 0; it < 16; it++) ub[it] = 0u;
  const bool emits = U != nullptr && SWEEP != 1;
  if (emits && !carried) {
#pragma unroll
    for (int it = 0; it < 16; it++) {
      const uint32_t e = (uint32_t)it * 256u + tid;
      const bool sing = e < nvalid && ((m_sg[it * 4 + w] >> lane) & 1ull);
      const uint32_t p = FIRST ? (uint32_t)(base + e) : p16[FIRST ? 0 : it];
      const uint32_t vv = (PACKED ? (uint32_t)sk[e + 1] : v16[PACKED ? 0 : it]) & PK_POS_MASK;
      const uint32_t blk = p / g.stride;
      ub[it] = sing ? (uint32_t)Tx[(size_t)blk * g.stride + (vv ? vv - 1u : blk_len(g, blk) - 1u)] : 0u;
    }
  }
#pragma unroll
  for (int it = 0; it < 16; it++) {
    const uint32_t e = (uint32_t)it * 256u + tid;
    if (e < nvalid) {
      const uint64_t a = base + e;
      const int wi = it * 4 + w;
      const uint64_t mh = m_nh[wi], ms = m_sg[wi];
      const uint64_t hm = mh & le;
      uint32_t head_a;                    // global index of the governing group head
      bool keeps_rank = false;            // the head also led the group of the previous round: R already holds this rank
      if (hm || wp_head[wi]) {
        const uint32_t hrel = hm ? (uint32_t)wi * 64u + 63u - (uint32_t)__builtin_clzll(hm) : wp_head[wi] - 1u;
        head_a = (uint32_t)base + hrel;
        if (!FIRST) keeps_rank = (m_oh[hrel >> 6] >> (hrel & 63u)) & 1ull;
      } else {
        head_a = carry - 1u;
        if (!FIRST) keeps_rank = head_a == 0 || (key[head_a] >> 20) != (key[head_a - 1] >> 20);
      }
      const uint32_t p = FIRST ? (uint32_t)a : p16[FIRST ? 0 : it];
      const uint32_t vraw = PACKED ? (uint32_t)sk[e + 1] : v16[PACKED ? 0 : it];
      const uint32_t vv = vraw & PK_POS_MASK;                    // the suffix (positions are below 2^20)
      const uint32_t pb = PACKED ? (vraw >> PK2_PREV_SHIFT) & 0xFFu : vraw >> VAL_PREV_SHIFT;      // its carried byte, if any
      const uint32_t blk = p / g.stride;
      const uint32_t head_pos = p - ((uint32_t)a - head_a);
      if (!keeps_rank && (SWEEP == 0 || (uint32_t)(vv >= hsplit) == half)) R[(size_t)blk * g.stride + vv] = head_pos - blk * g.stride;
      if (SWEEP == 1) continue;
      if ((ms >> lane) & 1ull) {
        // a resolved suffix: its BWT byte goes straight to the output row (cyclic form: U[p] = T[suffix - 1], p in sorted-position
        // order) -- no suffix array is written, and the byte gathers overlap the rest of the regrouping instead of making a pass
        // of their own at the end; the sentinel form shifts the rows by the primary index, which is only known at the end: SA
        if (U) U[p] = (uint8_t)(carried ? pb : ub[it]);
        else __builtin_nontemporal_store(vv, SA + p);
      } else {
        const uint32_t so = sbase + wp_s[wi] + (uint32_t)__popcll(~ms & lt);
        const uint32_t ho = hbase + wp_h[wi] + (uint32_t)__popcll(mh & ~ms & le);
        __builtin_nontemporal_store(carried ? (vv | (pb << VAL_PREV_SHIFT)) : vv, nval + so); __builtin_nontemporal_store(p, npos + so); __builtin_nontemporal_store(ho - 1u, ngord + so);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Rounds >= 2: most unresolved groups are tiny, and the array is already ordered by group, so only the order
// INSIDE each group is missing.  bwt_tile_sort sorts every group of <= 1024 suffixes in LDS: a tile owns the
// groups whose head lies in its nominal 3072-slot range and loads a 4096-slot window (1024 slots of slack behind
// the nominal range, so every owned group is complete).  Tiles whose owned groups all have <= 64 members rank
// every member by counting inside its group; other tiles bitonic-sort the window on (head slot, rank key, suffix):
// slots that are not owned carry their own slot number and stay where they are.
// Groups of > 1024 suffixes keep their defer flag (preset to 1 by the host) and go through the global radix
// passes (compact -> sort -> scatter back).
// ------------------------------------------------------------------------------------------
constexpr uint32_t TS_WIN = 4096, TS_MAXGRP = 1024, TS_NOM = TS_WIN - TS_MAXGRP, TS_TINY = 64, TS_GT = 2048;

// Fused rank gather: the tile sorters of rounds >= 2 make the round's keys themselves -- group
// ordinal from gord[], rank of suffix val + h out of R -- instead of reading back a key array that bwt_gather_keys wrote: the
// random rank fetches (bound by the number of 64-byte transactions) then overlap the VALU-bound sorting of the other
// workgroups of the CU, and 16 B per suffix of key traffic are gone.  Windows overlap, so every slot has ONE window that
// fetches its rank and writes its key: the window that owns its group, else the window whose nominal range holds it.  The
// only group a window cannot see whole is the one that runs into it from the left; whether the window before owns that one
// (<= TS_MAXGRP members, all inside its 4096 slots) follows from the 1024 ordinals in front of the window.
struct TsGather { const uint32_t* R; const uint32_t* pos; const uint32_t* gord; uint32_t h; int cyclic; Geom g; };
// slot of step s of a thread: 64 consecutive slots per wave instruction in both layouts (word of a slot = slot >> 6)
template <bool WMAP> __device__ __forceinline__ uint32_t ts_slot(int s, int tid) {
  return WMAP ? (uint32_t)(tid >> 6) * 1024u + (uint32_t)s * 64u + (uint32_t)(tid & 63) : (uint32_t)s * 256u + (uint32_t)tid;
}
// Loads of the fused form (all issued before the first use): ordinals, suffixes and sorted positions of the window, and the
// position (+1) of the last ordinal in front of the window that differs from the window's first one (this thread's four).
template <bool WMAP>
__device__ __forceinline__ uint32_t ts_fused_load(const TsGather& tg, const uint32_t* __restrict__ val, uint64_t wb, uint32_t L,
                                                  uint32_t (&go)[16], uint32_t (&pv)[16], uint32_t (&pp)[16], uint32_t& gprev) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int s = 0; s < 16; s++) { const uint32_t x = ts_slot<WMAP>(s, tid); go[s] = tg.gord[wb + (x < L ? x : L - 1u)]; }
#pragma unroll
  for (int s = 0; s < 16; s++) { const uint32_t x = ts_slot<WMAP>(s, tid); pv[s] = val[wb + (x < L ? x : L - 1u)]; }
#pragma unroll
  for (int s = 0; s < 16; s++) { const uint32_t x = ts_slot<WMAP>(s, tid); pp[s] = tg.pos[wb + (x < L ? x : L - 1u)]; }
  uint32_t li = 0;
  gprev = 0xFFFFFFFFu;
  if (wb) {                                       // wb is a multiple of TS_NOM >= TS_MAXGRP
    const uint32_t* q = tg.gord + wb - TS_MAXGRP + (uint32_t)tid * 4u;
    const uint32_t q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], g0 = tg.gord[wb];
    gprev = tg.gord[wb - 1];
    li = q3 != g0 ? (uint32_t)tid * 4u + 4u : q2 != g0 ? (uint32_t)tid * 4u + 3u : q1 != g0 ? (uint32_t)tid * 4u + 2u : q0 != g0 ? (uint32_t)tid * 4u + 1u : 0u;
  }
#pragma unroll
  for (int s = 0; s < 16; s++) if (ts_slot<WMAP>(s, tid) >= L) go[s] = 0xFFFFFFFFu;
  return li;
}
// slots [0, result) of the window belong to a group that the window before owns (hm / wnext of this window are known)
__device__ __forceinline__ uint32_t ts_prev_end(uint64_t wb, uint32_t L, bool at_end, uint32_t li_max, const uint64_t* hm, const int32_t* wnext) {
  const uint64_t m0 = hm[0];
  if (!wb || (m0 & 1ull)) return 0u;              // the first slot starts a group
  const uint32_t sb = TS_MAXGRP - li_max;         // members in front of the window (TS_MAXGRP: or more)
  const uint32_t fh = m0 ? (uint32_t)__builtin_ctzll(m0) : (uint32_t)wnext[0];
  const uint32_t e = fh <= L ? fh : (at_end ? L : TS_WIN + 1u);
  return sb + e <= TS_MAXGRP ? e : 0u;
}
// rank keys of the slots in `need` (bit s = step s): R[suffix + h] + 1 of the suffix's block, 0 past the end (sentinel form)
__device__ __forceinline__ void ts_fused_gather(const TsGather& tg, uint32_t need, const uint32_t (&pv)[16], uint32_t (&pp)[16], uint32_t (&rk20)[16]) {
  uint32_t past = 0;
#pragma unroll
  for (int s = 0; s < 16; s++) {
    const uint32_t blk = pp[s] / tg.g.stride, n = blk_len(tg.g, blk);
    uint32_t j = (pv[s] & PK_POS_MASK) + tg.h;       // (the top byte of val[] may carry the byte in front of the suffix)
    if (tg.cyclic) { if (j >= n) j %= n; }
    else if (j >= n) { j = 0; past |= 1u << s; }
    pp[s] = blk * tg.g.stride + j;               // (positions are 32 bits wide: bwt_run refuses more)
  }
#pragma unroll
  for (int s = 0; s < 16; s++) rk20[s] = tg.R[((need >> s) & 1u) ? pp[s] : pp[0]];
#pragma unroll
  for (int s = 0; s < 16; s++) rk20[s] = ((past >> s) & 1u) ? 0u : rk20[s] + 1u;
}

__device__ __forceinline__ void ts_ce(uint64_t& a, uint64_t& b, bool up) {
  const bool sw = (a > b) == up;
  const uint64_t x = sw ? b : a, y = sw ? a : b;
  a = x; b = y;
}
__global__ __launch_bounds__(256) void bwt_tile_sort(uint64_t* __restrict__ key, uint32_t* __restrict__ val, uint32_t A,
                                                     uint8_t* __restrict__ dflag, TsGather tg, uint32_t Tt) {
  __shared__ uint64_t sk[TS_WIN];
  __shared__ uint64_t hm[64];                    // head mask of the window
  __shared__ int32_t wlast[64], wnext[64];       // last head before word / first head after word (window slot, -1 / TS_WIN+1 = none)
  uint32_t* gk = (uint32_t*)sk;                  // group ordinals of the window slots (only until the heads are known)
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  __shared__ uint32_t lired[4];
  const uint32_t win = xcd_tile(blockIdx.x, Tt);            // windows of one block on one XCD: its ranks stay in that L2
  if (win >= Tt) return;
  const uint64_t wb = (uint64_t)win * TS_NOM;
  const uint64_t we = (wb + TS_WIN < A) ? wb + TS_WIN : A;
  const uint32_t L = (uint32_t)(we - wb);
  uint64_t r[16];
  uint32_t pv[16], go[16], pp[16];          // go: group ordinal of slot it*256+tid
  uint32_t gprev;
  // every global load of the window (ordinals, suffixes, positions) is issued before the first LDS store: written as one loop
  // the compiler waits for each load in turn, sixteen memory round trips instead of one
  {
    const uint32_t li = wave_max(ts_fused_load<false>(tg, val, wb, L, go, pv, pp, gprev));
    if (lane == 0) lired[w] = li;
  }
#pragma unroll
  for (int it = 0; it < 16; it++) gk[(uint32_t)it * 256u + tid] = go[it];
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 16; it++) {
    const uint32_t x = (uint32_t)it * 256u + tid;
    const bool head = x < L && (wb + x == 0 || gk[x] != (x ? gk[x - 1] : gprev));
    const uint64_t m = __ballot(head);
    if (lane == 0) hm[it * 4 + w] = m;
  }
  __syncthreads();
  if (w == 0) {      // per mask word: last head strictly before the word, first head strictly after it
    const uint64_t m = hm[lane];
    const int lastin = m ? lane * 64 + 63 - (int)__builtin_clzll(m) : -1;
    const int firstin = m ? lane * 64 + (int)__builtin_ctzll(m) : (int)TS_WIN + 1;
    int il = wave_incl_max(lastin);
    int el = __shfl_up(il, 1, 64); if (lane == 0) el = -1;
    // suffix min of firstin: max-scan on the reversed lane order of the negated value
    int neg = -firstin;
    int rv = __shfl(neg, 63 - lane, 64);
    int ir = wave_incl_max(rv);
    int er = __shfl_up(ir, 1, 64); if (lane == 0) er = -((int)TS_WIN + 1);
    const int en = -__shfl(er, 63 - lane, 64);
    wlast[lane] = el; wnext[lane] = en;
  }
  __syncthreads();
  __shared__ uint64_t mm[64];                    // mask of the owned slots in groups of > TS_TINY members
  __shared__ uint32_t mpre[65];
  int any_medium = 0;
  uint32_t hx16[16];          // head slot (low 16 bits) | group size (high 16 bits) of owned slots, 0xFFFFFFFF otherwise
  uint32_t prev_end = 0, needm = 0;
  {
    const uint32_t a01 = lired[0] > lired[1] ? lired[0] : lired[1], a23 = lired[2] > lired[3] ? lired[2] : lired[3];
    prev_end = ts_prev_end(wb, L, we == A, a01 > a23 ? a01 : a23, hm, wnext);
  }
#pragma unroll
  for (int it = 0; it < 16; it++) {
    const uint32_t x = (uint32_t)it * 256u + tid;
    const int wi = it * 4 + w;
    const uint64_t m = hm[wi];
    const uint64_t le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    const uint64_t gt = lane == 63 ? 0ull : ~((2ull << lane) - 1ull);
    int hx = (m & le) ? wi * 64 + 63 - (int)__builtin_clzll(m & le) : wlast[wi];            // head slot of my group (-1: before the window)
    int nx = (m & gt) ? wi * 64 + (int)__builtin_ctzll(m & gt) : wnext[wi];                  // next head after me
    if (nx > (int)L) nx = (we == A) ? (int)L : (int)TS_WIN + 1;                              // the array end closes the last group
    const bool valid = x < L;
    const bool small = valid && hx >= 0 && nx <= (int)L && (uint32_t)(nx - hx) <= TS_MAXGRP;
    const bool owned = small && (uint32_t)hx < TS_NOM;
    const bool medium = owned && (nx - hx) > (int)TS_TINY;
    if (owned && dflag) dflag[wb + x] = 0;
    hx16[it] = owned ? ((uint32_t)hx | ((uint32_t)(nx - hx) << 16)) : 0xFFFFFFFFu;
    const uint64_t mb = __ballot(medium);
    if (lane == 0) mm[wi] = mb;
    any_medium |= medium ? 1 : 0;
    needm |= (valid && x >= prev_end && (owned || x < TS_NOM)) ? 1u << it : 0u;
  }
  {
    uint32_t rk20[16];
    ts_fused_gather(tg, needm, pv, pp, rk20);
#pragma unroll
    for (int it = 0; it < 16; it++) {
      const uint32_t x = (uint32_t)it * 256u + tid;
      const bool owned = hx16[it] != 0xFFFFFFFFu;
      if (((needm >> it) & 1u) && !owned) key[wb + x] = ((uint64_t)go[it] << 20) | rk20[it];      // not sorted here: the key as gathered
      r[it] = owned ? (((uint64_t)(hx16[it] & 0xFFFFu) << 52) | ((uint64_t)rk20[it] << 32) | pv[it])
                    : (((uint64_t)x << 52) | (0xFFFFFull << 32));
    }
  }
  __syncthreads();            // gk (aliasing sk) is dead from here
  // groups of <= TS_TINY members: every member counts the smaller members of its group.  Inside a group only the
  // 20-bit rank key matters (equal keys stay one group, any order): 32-bit words (key << 12 | slot) are compared
  uint32_t* sk32 = reinterpret_cast<uint32_t*>(sk);
#pragma unroll
  for (int it = 0; it < 16; it++) { const uint32_t x = (uint32_t)it * 256u + tid; sk32[x] = ((uint32_t)(r[it] >> 32) << 12) | x; }
  any_medium = __syncthreads_or(any_medium);
#pragma unroll
  for (int it = 0; it < 16; it++) {
    const bool own = hx16[it] != 0xFFFFFFFFu && (hx16[it] >> 16) <= TS_TINY;
    const uint32_t hx = own ? (hx16[it] & 0xFFFFu) : 0u, g = own ? (hx16[it] >> 16) : 0u;
    const uint32_t gmax = wave_max(g);
    if (gmax == 0) continue;
    const uint64_t mine = r[it];
    const uint32_t mine32 = ((uint32_t)(mine >> 32) << 12) | ((uint32_t)it * 256u + tid);
    uint32_t rank = 0;
    for (uint32_t d = 0; d < gmax; d++) rank += (d < g && sk32[hx + d] < mine32) ? 1u : 0u;
    if (own) {
      key[wb + hx + rank] = ((uint64_t)go[it] << 20) | ((mine >> 32) & 0xFFFFFull);
      val[wb + hx + rank] = (uint32_t)mine;
    }
  }
  if (!any_medium) return;
  // larger groups: compact their members (the order of slots is kept), bitonic-sort the compacted array on
  // (head slot, rank key, suffix), and hand the c-th element to the c-th compacted slot
  if (w == 0) {
    const uint32_t pc = (uint32_t)__builtin_popcountll(mm[lane]);
    const uint32_t inc = wave_incl_sum(pc);
    mpre[lane] = inc - pc;
    if (lane == 63) mpre[64] = inc;
  }
  __syncthreads();            // also: all counting reads of sk are done
  const uint32_t Mt = mpre[64];
  uint32_t P = 2;
  while (P < Mt) P <<= 1;
  uint32_t cix[16];
#pragma unroll
  for (int it = 0; it < 16; it++) {
    const int wi = it * 4 + w;
    const uint64_t mb = mm[wi];
    const bool medium = (mb >> lane) & 1ull;
    cix[it] = medium ? mpre[wi] + (uint32_t)__builtin_popcountll(mb & ((1ull << lane) - 1ull)) : 0xFFFFFFFFu;
    if (medium) sk[cix[it]] = r[it];
  }
  for (uint32_t c = Mt + tid; c < P; c += 256) sk[c] = ~0ull;
  __syncthreads();
  for (uint32_t kk = 2; kk <= P; kk <<= 1) {
    uint32_t j = kk >> 1;
    if (__builtin_popcount(kk - 1) & 1) {      // odd number of steps in this merge phase: one single step first
      for (uint32_t t = tid; t < (P >> 1); t += 256) {
        const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const uint32_t l = i | j;
        const uint64_t a = sk[i], b = sk[l];
        const bool up = (i & kk) == 0;
        if ((a > b) == up) { sk[i] = b; sk[l] = a; }
      }
      __syncthreads();
      j >>= 1;
    }
    for (; j > 0; j >>= 2) {                   // steps j and j/2 in one LDS round trip, 4 elements per thread
      const uint32_t hj = j >> 1;
      for (uint32_t t = tid; t < (P >> 2); t += 256) {
        const uint32_t i0 = ((t & ~(hj - 1)) << 2) | (t & (hj - 1));
        const bool up = (i0 & kk) == 0;
        uint64_t e0 = sk[i0], e1 = sk[i0 | hj], e2 = sk[i0 | j], e3 = sk[i0 | j | hj];
        ts_ce(e0, e2, up); ts_ce(e1, e3, up); ts_ce(e0, e1, up); ts_ce(e2, e3, up);
        sk[i0] = e0; sk[i0 | hj] = e1; sk[i0 | j] = e2; sk[i0 | j | hj] = e3;
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int it = 0; it < 16; it++) {
    if (cix[it] != 0xFFFFFFFFu) {
      const uint32_t x = (uint32_t)it * 256u + tid;
      const uint64_t e = sk[cix[it]];
      key[wb + x] = ((uint64_t)go[it] << 20) | ((e >> 32) & 0xFFFFFull);
      val[wb + x] = (uint32_t)e;
    }
  }
}
// LDS radix version of the tile sorter.  Every slot of the window gets a 32-bit composite (head slot of its group << 20 |
// 20-bit rank key); slots that are not owned carry (own slot << 20).  A stable LSD radix sort of the whole window on that
// composite (4 passes of 8 bits, wave64 match-any ranking, one LDS staging array) is then a permutation INSIDE every owned
// group: exactly h slots carry a composite below (h << 20), so the members of the group headed at h land on [h, h + size).
// The cost does not depend on the group sizes (the counting / bitonic version above degrades with them).
__global__ __launch_bounds__(256) void bwt_tile_sort_radix(uint64_t* __restrict__ key, uint32_t* __restrict__ val, uint32_t A,
                                                           uint8_t* __restrict__ dflag, TsGather tg, uint32_t Tt) {
  __shared__ uint64_t se[TS_WIN];                // (composite << 32) | suffix: staging of a pass
  __shared__ uint64_t hm[64];
  __shared__ int32_t wlast[64], wnext[64];
  __shared__ uint32_t wcnt[4][256];
  __shared__ uint32_t sm[4];
  uint32_t* gk = (uint32_t*)se;                  // group ordinals of the window slots (only until the heads are known)
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  __shared__ uint32_t lired[4];
  const uint32_t win = xcd_tile(blockIdx.x, Tt);            // windows of one block on one XCD: its ranks stay in that L2
  if (win >= Tt) return;
  const uint64_t wb = (uint64_t)win * TS_NOM;
  const uint64_t we = (wb + TS_WIN < A) ? wb + TS_WIN : A;
  const uint32_t L = (uint32_t)(we - wb);
  // slot of (wave w, step s, lane): w * 1024 + s * 64 + lane -- array order = (w, s, lane) order, which the stable ranking needs
  uint32_t rk20[16], go[16], pv[16], pp[16];
  uint32_t gprev;
  {
    const uint32_t li = wave_max(ts_fused_load<true>(tg, val, wb, L, go, pv, pp, gprev));
    if (lane == 0) lired[w] = li;
#pragma unroll
    for (int s = 0; s < 16; s++) gk[ts_slot<true>(s, tid)] = go[s];
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 16; s++) {
    const uint32_t x = (uint32_t)w * 1024u + (uint32_t)s * 64u + lane;
    const bool head = x < L && (wb + x == 0 || gk[x] != (x ? gk[x - 1] : gprev));
    const uint64_t m = __ballot(head);
    if (lane == 0) hm[w * 16 + s] = m;
  }
  __syncthreads();
  if (w == 0) {      // per mask word: last head strictly before the word, first head strictly after it
    const uint64_t m = hm[lane];
    const int lastin = m ? lane * 64 + 63 - (int)__builtin_clzll(m) : -1;
    const int firstin = m ? lane * 64 + (int)__builtin_ctzll(m) : (int)TS_WIN + 1;
    int il = wave_incl_max(lastin);
    int el = __shfl_up(il, 1, 64); if (lane == 0) el = -1;
    int neg = -firstin;
    int rv = __shfl(neg, 63 - lane, 64);
    int ir = wave_incl_max(rv);
    int er = __shfl_up(ir, 1, 64); if (lane == 0) er = -((int)TS_WIN + 1);
    const int en = -__shfl(er, 63 - lane, 64);
    wlast[lane] = el; wnext[lane] = en;
  }
  __syncthreads();
  uint32_t prev_end = 0;
  {
    const uint32_t a01 = lired[0] > lired[1] ? lired[0] : lired[1], a23 = lired[2] > lired[3] ? lired[2] : lired[3];
    prev_end = ts_prev_end(wb, L, we == A, a01 > a23 ? a01 : a23, hm, wnext);
  }
  uint32_t comp[16];
  uint32_t ownm = 0, needm = 0;
#pragma unroll
  for (int s = 0; s < 16; s++) {
    const uint32_t x = (uint32_t)w * 1024u + (uint32_t)s * 64u + lane;
    const int wi = w * 16 + s;
    const uint64_t m = hm[wi];
    const uint64_t le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    const uint64_t gt = lane == 63 ? 0ull : ~((2ull << lane) - 1ull);
    const int hx = (m & le) ? wi * 64 + 63 - (int)__builtin_clzll(m & le) : wlast[wi];            // head slot of my group (-1: before the window)
    int nx = (m & gt) ? wi * 64 + (int)__builtin_ctzll(m & gt) : wnext[wi];                        // next head after me
    if (nx > (int)L) nx = (we == A) ? (int)L : (int)TS_WIN + 1;                                    // the array end closes the last group
    const bool owned = x < L && hx >= 0 && nx <= (int)L && (uint32_t)(nx - hx) <= TS_MAXGRP && (uint32_t)hx < TS_NOM;
    if (owned && dflag) dflag[wb + x] = 0;
    ownm |= owned ? 1u << s : 0u;
    needm |= (x < L && x >= prev_end && (owned || x < TS_NOM)) ? 1u << s : 0u; comp[s] = (uint32_t)hx;
  }
  {
    ts_fused_gather(tg, needm, pv, pp, rk20);
#pragma unroll
    for (int s = 0; s < 16; s++) {
      const uint32_t x = (uint32_t)w * 1024u + (uint32_t)s * 64u + lane;
      const bool owned = (ownm >> s) & 1u;
      if (((needm >> s) & 1u) && !owned) key[wb + x] = ((uint64_t)go[s] << 20) | rk20[s];      // not sorted here: the key as gathered
      comp[s] = owned ? ((comp[s] << 20) | rk20[s]) : (x << 20);
    }
  }
#pragma unroll
  for (int s = 0; s < 16; s++) if (!((ownm >> s) & 1u)) pv[s] = 0u;
  if (!__syncthreads_or((int)ownm)) return;      // also: gk (aliasing se) is dead from here
#pragma unroll 1
  for (int shift = 0; shift < 32; shift += 8) {
    for (int i = tid; i < 1024; i += 256) (&wcnt[0][0])[i] = 0;
    __syncthreads();
    uint32_t rk[16];
#pragma unroll
    for (int s = 0; s < 16; s++) rk[s] = rank_step((comp[s] >> shift) & 255u, wcnt[w]);
    __syncthreads();
    {
      const uint32_t c0 = wcnt[0][tid], c1 = wcnt[1][tid], c2 = wcnt[2][tid], c3 = wcnt[3][tid];
      uint32_t total;
      const uint32_t ex = block_excl_sum<256>(c0 + c1 + c2 + c3, sm, total);
      wcnt[0][tid] = ex; wcnt[1][tid] = ex + c0; wcnt[2][tid] = ex + c0 + c1; wcnt[3][tid] = ex + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16; s++) {
      const uint32_t d = (comp[s] >> shift) & 255u;
      se[wcnt[w][d] + rk[s]] = ((uint64_t)comp[s] << 32) | pv[s];
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16; s++) {
      const uint64_t e = se[(uint32_t)w * 1024u + (uint32_t)s * 64u + lane];
      comp[s] = (uint32_t)(e >> 32); pv[s] = (uint32_t)e;
    }
    __syncthreads();
  }
#pragma unroll
  for (int s = 0; s < 16; s++) {
    if ((ownm >> s) & 1u) {
      const uint32_t x = (uint32_t)w * 1024u + (uint32_t)s * 64u + lane;
      key[wb + x] = ((uint64_t)go[s] << 20) | (comp[s] & 0xFFFFFu);
      val[wb + x] = pv[s];
    }
  }
}
// single workgroup: exclusive scan of n counters in place, total -> *total
__global__ __launch_bounds__(1024) void scan_u32_single(uint32_t* __restrict__ arr, uint32_t n, uint32_t* __restrict__ total, uint32_t* __restrict__ host_total) {
  __shared__ uint32_t sm[16];
  uint32_t carry = 0;
  for (uint32_t base = 0; base < n; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    uint32_t v = i < n ? arr[i] : 0u, tot;
    const uint32_t ex = block_excl_sum<1024>(v, sm, tot);
    if (i < n) arr[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) { *total = carry; *host_total = carry; }
}
// Groups too large for the tile sorters (dflag = 1 on all their slots) are compacted, sorted by the global radix passes and
// put back.  Their group ordinals take ~23 bits of the sort key; numbered densely among themselves (there are at most
// A / 1025 of them) they take ~11, which saves two of the six passes: a slot starts a deferred group when the slot in front of
// it is not deferred or carries another ordinal.  Per 2048-slot tile: deferred slots -> tcount, deferred group heads -> hcount.
__device__ __forceinline__ void defer_flags(uint32_t A, const uint8_t* __restrict__ dflag, const uint64_t* __restrict__ key, uint64_t a0,
                                            uint32_t& f, uint32_t& heads, uint64_t (&k8)[8]) {
  f = 0; heads = 0;
  if (a0 >= A) return;
  if (a0 + 8 <= A) {
    const uint64_t w = *(const uint64_t*)(dflag + a0);
#pragma unroll
    for (int j = 0; j < 8; j++) f |= (uint32_t)((w >> (8 * j)) & 1ull) << j;
  } else for (int j = 0; j < 8; j++) if (a0 + j < A && dflag[a0 + j]) f |= 1u << j;
  if (!f) return;
#pragma unroll
  for (int j = 0; j < 8; j++) k8[j] = a0 + j < A ? key[a0 + j] : 0ull;
  const bool pf = a0 && dflag[a0 - 1];
  uint64_t prev = pf ? key[a0 - 1] >> 20 : ~0ull;
  bool prev_def = pf;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const bool d = (f >> j) & 1u;
    if (d && (!prev_def || prev != (k8[j] >> 20))) heads |= 1u << j;
    prev_def = d; prev = k8[j] >> 20;
  }
}
__global__ __launch_bounds__(256) void bwt_defer_count(uint32_t A, const uint8_t* __restrict__ dflag, const uint64_t* __restrict__ key,
                                                       uint32_t* __restrict__ tcount, uint32_t* __restrict__ hcount) {
  __shared__ uint32_t sm[4];
  const uint64_t a0 = (uint64_t)blockIdx.x * TS_GT + (uint32_t)threadIdx.x * 8u;
  uint32_t f, heads; uint64_t k8[8];
  defer_flags(A, dflag, key, a0, f, heads, k8);
  const uint32_t cnt = block_sum<256>((uint32_t)__builtin_popcount(f), sm);
  const uint32_t hc = block_sum<256>((uint32_t)__builtin_popcount(heads), sm);
  if (threadIdx.x == 0) { tcount[blockIdx.x] = cnt; hcount[blockIdx.x] = hc; }
}
__global__ __launch_bounds__(256) void bwt_defer_gather(const uint64_t* __restrict__ key, const uint32_t* __restrict__ val, uint32_t A,
                                                        const uint8_t* __restrict__ dflag, const uint32_t* __restrict__ tcount, const uint32_t* __restrict__ hcount,
                                                        uint64_t* __restrict__ dk, uint32_t* __restrict__ dv, uint32_t* __restrict__ dpos) {
  __shared__ uint32_t sm[4];
  const uint64_t a0 = (uint64_t)blockIdx.x * TS_GT + (uint32_t)threadIdx.x * 8u;
  uint32_t f, heads; uint64_t k8[8];
  defer_flags(A, dflag, key, a0, f, heads, k8);
  uint32_t tot;
  uint32_t o = tcount[blockIdx.x] + block_excl_sum<256>((uint32_t)__builtin_popcount(f), sm, tot);
  uint32_t hid = hcount[blockIdx.x] + block_excl_sum<256>((uint32_t)__builtin_popcount(heads), sm, tot);      // deferred heads in front of this thread's slots
#pragma unroll
  for (int j = 0; j < 8; j++) if ((f >> j) & 1u) {
    hid += (heads >> j) & 1u;
    dk[o] = ((uint64_t)(hid - 1u) << 20) | (k8[j] & 0xFFFFFull); dv[o] = val[a0 + j]; dpos[o] = (uint32_t)(a0 + j); o++;
  }
}
__global__ __launch_bounds__(256) void bwt_defer_scatter(uint32_t D, const uint64_t* __restrict__ dk, const uint32_t* __restrict__ dv,
                                                         const uint32_t* __restrict__ dpos, uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
  for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < D; j += (uint64_t)gridDim.x * 256) {
    const uint32_t a = dpos[j];          // a slot of the same group: its ordinal stays, the rank key is the sorted one
    key[a] = (key[a] & ~0xFFFFFull) | (dk[j] & 0xFFFFFull); val[a] = dv[j];
  }
}

__global__ __launch_bounds__(256) void bwt_flush_active(uint32_t A, const uint32_t* __restrict__ val, const uint32_t* __restrict__ pos,
                                                        uint32_t* __restrict__ SA, Geom g, const uint8_t* __restrict__ Tx, uint8_t* __restrict__ U, int carried) {
  for (uint64_t a = (uint64_t)blockIdx.x * 256 + threadIdx.x; a < A; a += (uint64_t)gridDim.x * 256) {
    const uint32_t p = pos[a], v = val[a] & PK_POS_MASK;
    if (U && carried) U[p] = (uint8_t)(val[a] >> VAL_PREV_SHIFT);
    else if (U) { const uint32_t blk = p / g.stride; U[p] = Tx[(size_t)blk * g.stride + (v ? v - 1u : blk_len(g, blk) - 1u)]; }
    else SA[p] = v;
  }
}

// Two-phase round 1 (cyclic, packed records): seven bytes of depth at 8 bytes per record.  Phase 1 sorted every block by
// bytes 2..6; here every sorted slot learns r1 = block-local slot of the head of its (bytes 2..6) class and becomes the
// phase-2 record  byte0 . byte1 (bits 63..48) | r1 (47..28) | parity (20) | position (19..0); two more stable passes on the
// top 16 bits then order by (byte0, byte1, class of bytes 2..6), and key >> 20 is the 7-byte group key.
// Tiles are the tiles of the segmented radix passes (one segment per block), so this kernel also leaves the per-tile
// histogram of byte1 -- the digit of the first phase-2 pass -- in hist[], and that pass needs no rs_hist of its own.
// No flag / scan passes in front either: the only class head a tile cannot see is the one of the class that runs into it,
// and the block is sorted, so wave 0 finds it by looking at the 64 slots in front of the tile (nearly always enough) and
// otherwise by a 64-way search for the lower bound of the tile's first key in the block.
__global__ __launch_bounds__(256) void bwt_phase2_records(const uint64_t* __restrict__ key, SegGeom sg, const uint8_t* __restrict__ T,
                                                          uint32_t* __restrict__ hist, uint64_t* __restrict__ out) {
  __shared__ uint64_t sk[RS_TILE + 1];
  __shared__ uint64_t m_nh[64];
  __shared__ uint32_t wp_head[64];
  constexpr int PH = 2;                          // histogram copies per wave (the 32 KB of sk[] limit the residency already)
  __shared__ uint32_t h[4 * 256 * PH];
  __shared__ uint32_t carry_s;
  const uint32_t tile = blockIdx.x;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const TileRef t = tile_ref(sg, tile);
  if (!t.nvalid) { hist[(size_t)tile * 256 + tid] = 0; return; }
  const uint64_t base = t.base, seg0 = (uint64_t)t.seg * sg.stride;
  const uint32_t nvalid = t.nvalid, n = t.seg + 1 == sg.nseg ? sg.n_last : sg.stride;
  uint64_t k16[16];                              // all loads first (see bwt_apply)
#pragma unroll
  for (int it = 0; it < 16; it++) {
    const uint32_t e = (uint32_t)it * 256u + tid;
    k16[it] = __builtin_nontemporal_load(key + base + (e < nvalid ? e : nvalid - 1u));
  }
  if (tid == 64) sk[0] = t.off ? key[base - 1] : ~0ull;
  if (w == 0) {
    // head of the class of the tile's first slot (global index + 1, as bwt_scan_tiles would have carried it)
    const uint32_t carry = (uint32_t)class_head_before(key, seg0, base, PK_SHIFT, lane) + 1u;
    if (lane == 0) carry_s = carry;
  }
  // the two leading bytes of every record's suffix: gathers from the block text (L2), independent of everything below
  uint32_t b01[16];
  {
    const uint8_t* tx = T + seg0;
#pragma unroll
    for (int it = 0; it < 16; it++) {
      const uint32_t p = (uint32_t)k16[it] & PK_POS_MASK;
      b01[it] = ((uint32_t)tx[p] << 8) | tx[p + 1 < n ? p + 1 : 0];          // (one unaligned 16-bit load instead: 720 -> 900 us)
      b01[it] |= (uint32_t)tx[p ? p - 1 : n - 1] << 16;                     // the byte in front (nearly always the line of tx[p]): carried from here on
    }
  }
#pragma unroll
  for (int i = 0; i < 4 * PH; i++) h[i * 256 + tid] = 0;
#pragma unroll
  for (int it = 0; it < 16; it++) {
    const uint32_t e = (uint32_t)it * 256u + tid;
    sk[e + 1] = e < nvalid ? k16[it] : ~0ull;
  }
  __syncthreads();
  uint32_t* hw = h + w * 256 * PH + (tid & (PH - 1));      // (bank-interleaved copies as in rs_hist)
#pragma unroll
  for (int it = 0; it < 16; it++) {
    const uint32_t e = (uint32_t)it * 256u + tid;
    const bool nh = e < nvalid && (t.off + e == 0 || (sk[e] >> PK_SHIFT) != (sk[e + 1] >> PK_SHIFT));
    const uint64_t mnh = __ballot(nh);
    if (lane == 0) m_nh[it * 4 + w] = mnh;
    if (e < nvalid) atomicAdd(&hw[(b01[it] & 255u) * PH], 1u);
  }
  __syncthreads();
  if (w == 0) {
    const uint64_t mh = m_nh[lane];
    const uint32_t lastrel = mh ? (uint32_t)lane * 64u + 63u - (uint32_t)__builtin_clzll(mh) + 1u : 0u;
    const uint32_t im = wave_incl_max(lastrel);
    uint32_t em = __shfl_up(im, 1, 64);
    if (lane == 0) em = 0;
    wp_head[lane] = em;
  }
  {
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int r = 0; r < PH; r++) sum += h[i * 256 * PH + tid * PH + r];
    hist[(size_t)tile * 256 + tid] = sum;
  }
  __syncthreads();
  const uint32_t carry = carry_s;
  const uint64_t le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
#pragma unroll
  for (int it = 0; it < 16; it++) {
    const uint32_t e = (uint32_t)it * 256u + tid;
    if (e < nvalid) {
      const int wi = it * 4 + w;
      const uint64_t hm = m_nh[wi] & le;
      const uint32_t head_a = hm ? (uint32_t)base + (uint32_t)wi * 64u + 63u - (uint32_t)__builtin_clzll(hm) : wp_head[wi] ? (uint32_t)base + wp_head[wi] - 1u : carry - 1u;
      const uint64_t k = k16[it];
      const uint32_t r1 = head_a - (uint32_t)seg0;
      __builtin_nontemporal_store(((uint64_t)(b01[it] & 0xFFFFu) << 48) | ((uint64_t)r1 << PK2_GSHIFT) | ((uint64_t)(b01[it] >> 16) << PK2_PREV_SHIFT) | (k & PK_POS_MASK), out + base + e);
    }
  }
}

// primary index.  cyclic: last row of the group of rotation 0 (equal rotations are ordered by
// descending start, J/Bzip2_joined_.js:957-968); sentinel: (row of suffix 0)+1
__global__ __launch_bounds__(256) void bwt_pidx(Geom g, int cyclic, int leftover, const uint32_t* __restrict__ R, uint32_t* __restrict__ pidx) {
  __shared__ uint32_t sm[4];
  const uint32_t blk = blockIdx.x, n = blk_len(g, blk);
  const uint32_t* r = R + (size_t)blk * g.stride;
  const uint32_t r0 = r[0];
  if (!cyclic) { if (threadIdx.x == 0) pidx[blk] = r0 + 1u; return; }
  if (!leftover) { if (threadIdx.x == 0) pidx[blk] = r0; return; }     // every rotation is unique
  uint32_t cnt = 0;
  for (uint32_t i = threadIdx.x; i < n; i += 256) cnt += r[i] == r0;
  cnt = block_sum<256>(cnt, sm);
  if (threadIdx.x == 0) pidx[blk] = r0 + cnt - 1u;
}

// BWT bytes of the sentinel form (BWTC) from the finished suffix array: rows shift by the primary index, which is only known at
// the end, so this form keeps a suffix array and an emit pass (the cyclic form writes its bytes from the regroup kernels).
// Tiles of 4096 positions, dealt to the XCDs in contiguous ranges: a block's text (the random-access side) is then
// fetched into one L2 instead of all eight; the suffix array streams through non-temporally.
__global__ __launch_bounds__(256) void bwt_emit_sentinel(const uint8_t* __restrict__ T, Geom g, uint32_t M,
                                                         const uint32_t* __restrict__ SA, const uint32_t* __restrict__ R, uint8_t* __restrict__ U, uint32_t Tn) {
  const uint32_t tile = xcd_tile(blockIdx.x, Tn);
  if (tile >= Tn) return;
#pragma unroll 4
  for (int it = 0; it < 16; it++) {
    const uint64_t a = (uint64_t)tile * RS_TILE + (uint32_t)it * 256 + threadIdx.x;
    if (a >= M) break;
    const uint32_t blk = (uint32_t)(a / g.stride), p = (uint32_t)(a - (uint64_t)blk * g.stride), n = blk_len(g, blk);
    const uint8_t* t = T + (size_t)blk * g.stride;
    uint8_t* u = U + (size_t)blk * g.stride;
    const uint32_t s = __builtin_nontemporal_load(SA + a);
    const uint32_t p0 = R[(size_t)blk * g.stride];
    if (p == p0) u[0] = t[n - 1];
    else u[p < p0 ? p + 1 : p] = t[s - 1];
  }
}

// ------------------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------------------
static int bits_for(uint64_t x) { int b = 0; while (x) { b++; x >>= 1; } return b; }
size_t BwtWork::bytes_needed(size_t cap) {
  const size_t T = hist_tiles_for(cap);
  size_t b = 0;
  auto add = [&](size_t n) { b += (n + 255) & ~(size_t)255; };
  add(cap * 8); add(cap * 8); add(cap * 4); add(cap * 4); add(cap * 4); add(cap * 4); add(cap * 4);  // key x2, val x2, pos x2, gord
  add(cap * 4); add(cap * 4); add(cap);           // R, SA, dflag
  add(hist_words(T) * 4); add(256 * segs_for(cap) * 4); add(3 * T * 4); add(64); add(16 * 256 * 4);
  return b + 4096;
}
int BwtWork::carve(Arena& a, size_t cap_) {
  cap = cap_;
  const size_t T = hist_tiles_for(cap);
  hist_tiles = (uint32_t)T; bintot_segs = (uint32_t)segs_for(cap);
  key[0] = a.take<uint64_t>(cap); key[1] = a.take<uint64_t>(cap);
  val[0] = a.take<uint32_t>(cap); val[1] = a.take<uint32_t>(cap);
  pos[0] = a.take<uint32_t>(cap); pos[1] = a.take<uint32_t>(cap);
  gord = a.take<uint32_t>(cap);
  R = a.take<uint32_t>(cap); SA = a.take<uint32_t>(cap); dflag = a.take<uint8_t>(cap);
  hist = a.take<uint32_t>(hist_words(T)); bintot = a.take<uint32_t>(256 * (size_t)bintot_segs);
  tile_cnt = a.take<uint32_t>(3 * T); counters = a.take<uint32_t>(16);
  ghist = a.take<uint32_t>(16 * 256);
  if (!counters || !ghist) return CJS_E_OUT_OF_MEMORY;
  if (!h_counters) CJS_HIP_TRY(hipHostMalloc((void**)&h_counters, 64));
  if (!ev_scan) CJS_HIP_TRY(hipEventCreateWithFlags(&ev_scan, hipEventDisableTiming));
  return 0;
}

template <typename K>
static int radix_passes(hipStream_t s, BwtWork& w, K* k0, uint32_t* v0, K* k1, uint32_t* v1, int& cur, uint32_t n, int lo_bit, int hi_bit,
                        LaunchTimes* lt, const SegGeom* seg = nullptr, const GenSrc* gen = nullptr, bool noval = false,
                        bool first_hist_ready = false,         // first_hist_ready: w.hist already holds the tile counts of the first digit
                        uint8_t* dig = nullptr) {              // dig: byte per key for the next pass's counts (see rs_hist_bytes)
  const uint32_t T1 = (n + RS_TILE - 1) / RS_TILE;
  const SegGeom sg = seg ? *seg : SegGeom{1u, n, n, T1};
  const uint32_t T = sg.nseg * sg.tps;
  if (T > w.hist_tiles || sg.nseg > w.bintot_segs) return CJS_E_INVALID_ARG;
  K* kk[2] = {k0, k1}; uint32_t* vv[2] = {v0, v1};
  const GenSrc g0{nullptr, 0, 0, 0};
  for (int shift = lo_bit; shift < hi_bit; shift += 8) {
    const bool first_gen = gen && shift == lo_bit;        // the first pass makes its keys from the block bytes
    const bool dig_out = dig && shift + 8 < hi_bit;          // a pass follows: leave its digits
    if (first_hist_ready && shift == lo_bit) {}
    else if (dig && shift != lo_bit) hipLaunchKernelGGL(rs_hist_bytes, dim3(T), dim3(256), 0, s, dig, sg, w.hist, 0u);
    else if (first_gen && gen->cyclic && gen->packed && shift == PK_KEY_LO)       // packed cyclic sort: the first digit is the text byte at suffix + 6
      hipLaunchKernelGGL(rs_hist_bytes, dim3(T), dim3(256), 0, s, gen->T, sg, w.hist, 6u);
    else if (first_gen) hipLaunchKernelGGL((rs_hist<K, true>), dim3(T), dim3(256), 0, s, kk[cur], sg, *gen, shift, w.hist, T);
    else hipLaunchKernelGGL((rs_hist<K, false>), dim3(T), dim3(256), 0, s, kk[cur], sg, g0, shift, w.hist, T);
    if (sg.tps <= 4 * SB_CHUNK) hipLaunchKernelGGL(rs_scan_bins, dim3(sg.nseg), dim3(256), 0, s, w.hist, sg.tps, w.bintot);
    else {                                    // one long segment (nseg > 1 with long segments: still correct, one launch per segment)
      for (uint32_t sgi = 0; sgi < sg.nseg; sgi++) {
        uint32_t* hseg = w.hist + (size_t)sgi * sg.tps * 256;
        const uint32_t nch = (sg.tps + SB_CHUNK - 1) / SB_CHUNK;
        uint32_t* csum = w.hist + (size_t)w.hist_tiles * 256;          // behind the per-tile rows (BwtWork::hist_words)
        hipLaunchKernelGGL(rs_scan_chunk_sum, dim3(nch), dim3(256), 0, s, hseg, sg.tps, csum);
        hipLaunchKernelGGL(rs_scan_chunk_mid, dim3(1), dim3(256), 0, s, csum, nch, w.bintot + (size_t)sgi * 256);
        hipLaunchKernelGGL(rs_scan_chunk_apply, dim3(nch), dim3(256), 0, s, hseg, sg.tps, csum);
      }
    }
    if (lt) lt->begin(s, n);
#define RS_SCATTER(GEN_, NOVAL_, G_) hipLaunchKernelGGL((rs_scatter<K, GEN_, NOVAL_>), dim3(T), dim3(256), 0, s, kk[cur], vv[cur], kk[1 - cur], vv[1 - cur], sg, G_, shift, w.hist, T, w.bintot, dig_out ? dig : nullptr)
    if (noval) { if (first_gen) RS_SCATTER(true, true, *gen); else RS_SCATTER(false, true, g0); }
    else { if (first_gen) RS_SCATTER(true, false, *gen); else RS_SCATTER(false, false, g0); }
#undef RS_SCATTER
    if (lt) lt->end(s);
    cur = 1 - cur;
  }
  CJS_HIP_TRY(hipGetLastError());
  return 0;
}

// 32-bit-key variant for other stages (decode.hip: stable partition of the BWT bytes = LF vector build)
template <typename K>
int radix_passes_public(hipStream_t s, BwtWork& w, K* k0, uint32_t* v0, K* k1, uint32_t* v1, int& cur, uint32_t n, int lo_bit, int hi_bit) {
  return radix_passes<K>(s, w, k0, v0, k1, v1, cur, n, lo_bit, hi_bit, nullptr);
}
template int radix_passes_public<uint32_t>(hipStream_t, BwtWork&, uint32_t*, uint32_t*, uint32_t*, uint32_t*, int&, uint32_t, int, int);
// ... the same over nseg runs of `stride` elements, each sorted by itself (decode.hip: one run per block)
template <typename K>
int radix_pass_segments_public(hipStream_t s, BwtWork& w, K* k0, uint32_t* v0, K* k1, uint32_t* v1, int& cur, uint32_t nseg, uint32_t stride, int lo_bit, int hi_bit, bool noval, bool first_hist_ready) {
  const SegGeom sg{nseg, stride, stride, (stride + RS_TILE - 1) / RS_TILE};
  return radix_passes<K>(s, w, k0, v0, k1, v1, cur, nseg * stride, lo_bit, hi_bit, nullptr, &sg, nullptr, noval, first_hist_ready);
}
template int radix_pass_segments_public<uint32_t>(hipStream_t, BwtWork&, uint32_t*, uint32_t*, uint32_t*, uint32_t*, int&, uint32_t, uint32_t, int, int, bool, bool);

// Which tile sorter: the LDS radix version costs the same whatever the groups look like (18 ps per suffix), the counting /
// bitonic version is cheaper once the groups are tiny (round 2 of the bench text, 7.7 suffixes per group: 1.51 vs 1.82 ms;
// round 3, 3.5 per group: 0.67 vs 0.64 ms; later rounds up to 2x in favour of counting).
static void launch_tile_sort(hipStream_t s, uint32_t Tt, uint64_t* key, uint32_t* val, uint32_t A, uint8_t* dflag, uint32_t ngroups, const TsGather& tg) {
  if (ngroups && A / ngroups >= 5) hipLaunchKernelGGL(bwt_tile_sort_radix, dim3(xcd_grid(Tt)), dim3(256), 0, s, key, val, A, dflag, tg, Tt);
  else hipLaunchKernelGGL(bwt_tile_sort, dim3(xcd_grid(Tt)), dim3(256), 0, s, key, val, A, dflag, tg, Tt);
}
// the tile sorters take the round (small remainders: whole-array radix passes on keys that bwt_gather_keys writes first)
static bool tile_sorted_round(uint32_t A) { return A >= 2 * TS_WIN; }
// One sort of a round >= 2: in-LDS tile sort of the small groups + global radix passes for the large ones.
// Works in place on (key[c], val[c]); only the whole-array fallback flips c.
static int sort_round(hipStream_t s, BwtWork& w, int& c, int pc, uint32_t A, int bits, LaunchTimes* lt, uint32_t ngroups, const TsGather& tg) {
  if (!tile_sorted_round(A)) return radix_passes<uint64_t>(s, w, w.key[0], w.val[0], w.key[1], w.val[1], c, A, 0, bits, lt);      // (bwt_gather_keys made the keys)
  const uint32_t Tt = (A + TS_NOM - 1) / TS_NOM, Tg = (A + TS_GT - 1) / TS_GT;
  uint8_t* dflag = w.dflag;
  uint32_t* tcount = w.tile_cnt;                        // 3*cap/4096 entries >= cap/2048
  if (w.no_large_groups) {                              // groups only ever split: once none exceeds TS_MAXGRP, none will
    launch_tile_sort(s, Tt, w.key[c], w.val[c], A, nullptr, ngroups, tg);
    CJS_HIP_TRY(hipGetLastError());
    return 0;
  }
  dev_fill(s, dflag, 1, A);
  launch_tile_sort(s, Tt, w.key[c], w.val[c], A, dflag, ngroups, tg);
  uint32_t* hcount = w.hist;                            // (free until the radix passes below, which come after the last reader of hcount)
  hipLaunchKernelGGL(bwt_defer_count, dim3(Tg), dim3(256), 0, s, A, dflag, w.key[c], tcount, hcount);
  hipLaunchKernelGGL(scan_u32_single, dim3(1), dim3(1024), 0, s, hcount, Tg, w.counters + 3, w.h_counters + 3);      // deferred groups
  hipLaunchKernelGGL(scan_u32_single, dim3(1), dim3(1024), 0, s, tcount, Tg, w.counters + 2, w.h_counters + 2);      // the kernel writes the pinned mirror itself
  CJS_HIP_TRY(hipStreamSynchronize(s));
  const uint32_t D = w.h_counters[2];
  if (getenv("CJS_DEBUG")) fprintf(stderr, "[cjs bwt]   tile sort: %u of %u suffixes in groups > %u\n", D, A, TS_MAXGRP);
  if (D == 0) w.no_large_groups = true;
  if (D == 0) return 0;
  if ((size_t)D > w.cap / 2) return radix_passes<uint64_t>(s, w, w.key[0], w.val[0], w.key[1], w.val[1], c, A, 0, bits, lt);
  uint64_t* dk0 = w.key[1 - c]; uint64_t* dk1 = dk0 + w.cap / 2;
  uint32_t* dv0 = w.val[1 - c]; uint32_t* dv1 = dv0 + w.cap / 2;
  uint32_t* dpos = w.pos[1 - pc];
  hipLaunchKernelGGL(bwt_defer_gather, dim3(Tg), dim3(256), 0, s, w.key[c], w.val[c], A, dflag, tcount, hcount, dk0, dv0, dpos);
  int cur = 0;
  const uint32_t ndg = w.h_counters[3];                 // deferred groups: dense ordinals 0 .. ndg-1 above the 20-bit rank key
  CJS_TRY((radix_passes<uint64_t>(s, w, dk0, dv0, dk1, dv1, cur, D, 0, 20 + bits_for(ndg ? ndg - 1 : 0), lt)));
  hipLaunchKernelGGL(bwt_defer_scatter, dim3((D + 255) / 256 < 8192u ? (D + 255) / 256 : 8192u), dim3(256), 0, s, D, cur ? dk1 : dk0, cur ? dv1 : dv0, dpos,
                     w.key[c], w.val[c]);
  CJS_HIP_TRY(hipGetLastError());
  return 0;
}

int bwt_run(hipStream_t s, BwtWork& w, const uint8_t* d_T, uint32_t nb, uint32_t stride, uint32_t n_last,
            bool cyclic, uint8_t* d_U, uint32_t* d_pidx, cjs_stats* stats, bool resolve_stats) {
  if (nb == 0) return 0;
  const uint64_t M64 = (uint64_t)(nb - 1) * stride + n_last;
  if (M64 > w.cap || M64 >= 0xFFFFF000ull) return CJS_E_INVALID_ARG;
  if (stride > (1u << 20) - 2) return CJS_E_INVALID_ARG;          // ranks must fit 20 bits
  const uint32_t M = (uint32_t)M64;
  const Geom g{nb, stride, n_last};
  const uint32_t max_n = nb > 1 ? stride : n_last;
  const int grid_lin = (int)((M + 255) / 256 < 65535u * 16u ? (M + 255) / 256 : 65535u * 16u);
  LaunchTimes& lt = w.lt; lt.reset(); lt.enabled = stats != nullptr; lt.min_elems = M;      // the roofline is priced on the full-size scatter passes

  int c = 0, pc = 0;        // current key/val buffer, current pos buffer
  dev_fill(s, w.counters, 0, 64);
  // round 1 sorts by the leading symbols.  Segmented (normal case): one segment per block, keys generated from the
  // block bytes in the first pass, 7 symbols + the block parity.  Fallback (more tiles/segments than the workspace
  // was carved for, e.g. many tiny blocks): keys materialised with the block id on top, sorted as one array.
  const int sym_bits = cyclic ? 8 : 9;
  const uint32_t tps = ((nb > 1 ? stride : n_last) + RS_TILE - 1) / RS_TILE;
  const bool segmented = (uint64_t)nb * tps <= w.hist_tiles && nb <= w.bintot_segs && getenv("CJS_NO_SEGMENTED_SORT") == nullptr;
  const int blk_bits = bits_for(nb - 1);
  int nsym = segmented ? 7 : (64 - blk_bits) / sym_bits;
  if (nsym > 7) nsym = 7;
  // packed round-1 records (5 bytes + position in one u64, no value array) for cyclic, segmented sorts, as a two-phase sort:
  // bytes 2..6 first, then bytes 0..1 with the class of bytes 2..6 carried as a rank (depth 7 at 8 bytes per record; same-box
  // A/Bs in profiles/r02_*: the seven passes + the record rebuild cost more than five passes, but 61 M instead of 83.5 M suffixes
  // stay unresolved behind them and every suffix-round costs ~55 ps).  The scatter passes leave the next digit of every key as a
  // byte for the next pass's counts.  Everything else (sentinel form, unsegmented fallback): (u64 key, u32 value) records.
  const bool packed = segmented && cyclic;
  if (packed) nsym = 7;
  const SegGeom sg{nb, stride, n_last, tps};
  const GenSrc gen{d_T, cyclic ? 1 : 0, nsym, packed ? 2 : 0};
  if (!segmented) hipLaunchKernelGGL(bwt_init_keys, dim3(grid_lin), dim3(256), 0, s, d_T, g, (int)cyclic, nsym, M, w.key[0], w.val[0]);
  uint32_t A = M, h = (uint32_t)nsym, rounds = 0, ngroups = 0;
  w.no_large_groups = false;
  int bits = nsym * sym_bits + (segmented ? 0 : blk_bits);
  // two-sweep scheduling of the round-1 rank scatter (see HalfMap): pays only with the packed records (the second sweep of the
  // 12-byte key + value form re-reads more than the merged stores save: 2.15 vs 1.64 ms)
  const bool sweeps = nb >= 8 && packed;
  TsGather tg{nullptr, nullptr, nullptr, 0u, 0, g};
  // cyclic form: the regroup kernels write the BWT bytes of the suffixes they resolve themselves; the sentinel form keeps a suffix array
  const uint8_t* dT = cyclic ? d_T : nullptr;
  uint8_t* dU = cyclic ? d_U : nullptr;
  const int gs1 = packed ? PK2_GSHIFT : 0;                              // group key of the round-1 records = key >> gs1
  const int carried = packed ? 1 : 0;                                   // packed records (then val[]) carry the byte in front of their suffix
  for (;;) {
    if (rounds == 0) {
      if (packed) {
        CJS_TRY((radix_passes<uint64_t>(s, w, w.key[0], w.val[0], w.key[1], w.val[1], c, A, PK_KEY_LO, 64, &lt, &sg, &gen, true, false, w.dflag)));
        hipLaunchKernelGGL(bwt_phase2_records, dim3(sg.nseg * sg.tps), dim3(256), 0, s, w.key[c], sg, d_T, w.hist, w.key[1 - c]);
        c = 1 - c;
        CJS_TRY((radix_passes<uint64_t>(s, w, w.key[0], w.val[0], w.key[1], w.val[1], c, A, 48, 64, &lt, &sg, nullptr, true, true, w.dflag)));
      } else if (segmented) CJS_TRY((radix_passes<uint64_t>(s, w, w.key[0], w.val[0], w.key[1], w.val[1], c, A, 0, bits, &lt, &sg, &gen)));
      else CJS_TRY((radix_passes<uint64_t>(s, w, w.key[0], w.val[0], w.key[1], w.val[1], c, A, 0, bits, &lt)));
    } else CJS_TRY(sort_round(s, w, c, pc, A, bits, &lt, ngroups, tg));
    const uint32_t T = (A + RS_TILE - 1) / RS_TILE;
    if (rounds == 0 && sweeps) {                                       // two launches, no counting pass (see bwt_apply)
      const HalfMap hm{2u, stride};
      hipLaunchKernelGGL((bwt_apply<true, true, 1>), dim3(xcd_grid(T)), dim3(256), 0, s, w.key[c], w.val[c], w.pos[pc], A, g, w.tile_cnt, T, w.R, w.SA,
                         w.val[1 - c], w.pos[1 - pc], w.gord, hm, dT, dU, w.counters + 4, gs1, carried);
      hipLaunchKernelGGL(bwt_scan_tiles, dim3(1), dim3(1024), 0, s, w.tile_cnt, T, w.counters, w.h_counters);
      CJS_HIP_TRY(hipEventRecord(w.ev_scan, s));
      hipLaunchKernelGGL((bwt_apply<true, true, 2>), dim3(xcd_grid(T)), dim3(256), 0, s, w.key[c], w.val[c], w.pos[pc], A, g, w.tile_cnt, T, w.R, w.SA,
                         w.val[1 - c], w.pos[1 - pc], w.gord, hm, dT, dU, w.counters + 4, gs1, carried);
    } else {
    hipLaunchKernelGGL(bwt_flags, dim3(T), dim3(256), 0, s, w.key[c], A, w.tile_cnt, T, rounds == 0 ? gs1 : 0, w.counters + 4, rounds == 0 ? stride : 0u);
    hipLaunchKernelGGL(bwt_scan_tiles, dim3(1), dim3(1024), 0, s, w.tile_cnt, T, w.counters, w.h_counters);
    CJS_HIP_TRY(hipEventRecord(w.ev_scan, s));
    if (rounds == 0) {
      const HalfMap hm{1u, stride};
      const uint32_t grid = xcd_grid(T);
      if (packed) hipLaunchKernelGGL((bwt_apply<true, true>), dim3(grid), dim3(256), 0, s, w.key[c], w.val[c], w.pos[pc], A, g, w.tile_cnt, T, w.R, w.SA,
                                     w.val[1 - c], w.pos[1 - pc], w.gord, hm, dT, dU, w.counters + 4, gs1, carried);
      else hipLaunchKernelGGL((bwt_apply<true, false>), dim3(grid), dim3(256), 0, s, w.key[c], w.val[c], w.pos[pc], A, g, w.tile_cnt, T, w.R, w.SA,
                              w.val[1 - c], w.pos[1 - pc], w.gord, hm, dT, dU, w.counters + 4, gs1, carried);
    } else hipLaunchKernelGGL((bwt_apply<false, false>), dim3(xcd_grid(T)), dim3(256), 0, s, w.key[c], w.val[c], w.pos[pc], A, g, w.tile_cnt, T, w.R, w.SA,
                              w.val[1 - c], w.pos[1 - pc], w.gord, HalfMap{1u, stride}, dT, dU, w.counters + 4, gs1, carried);
    }
    // the host only needs the counters of the tile scan: it waits for THAT kernel and queues the next round behind the regroup
    // kernel while it runs (a stream synchronisation here left the GPU idle for ~20 us per round)
    CJS_HIP_TRY(hipEventSynchronize(w.ev_scan));
    rounds++;
    const uint32_t A2 = w.h_counters[0], NG = w.h_counters[1];
    if (w.h_counters[4] == 0) w.no_large_groups = true;       // every group of the new grouping fits the tile sorters
    if (getenv("CJS_DEBUG")) fprintf(stderr, "[cjs bwt] round %u h=%u A=%u bits=%d -> A'=%u groups=%u\n", rounds, h, A, bits, A2, NG);
    c = 1 - c; pc = 1 - pc;
    A = A2; ngroups = NG;
    if (A == 0) break;
    if (cyclic && h >= max_n) {     // only groups of equal rotations are left
      hipLaunchKernelGGL(bwt_flush_active, dim3((A + 255) / 256), dim3(256), 0, s, A, w.val[c], w.pos[pc], w.SA, g, dT, dU, carried);
      break;
    }
    if (rounds > 40) return CJS_E_HIP;    // cannot happen: depth doubles every round
    if (tile_sorted_round(A)) tg = TsGather{w.R, w.pos[pc], w.gord, h, (int)cyclic, g};       // the tile sorters fetch the ranks themselves
    else {
      const uint32_t Tg = (A + RS_TILE - 1) / RS_TILE;
      hipLaunchKernelGGL(bwt_gather_keys, dim3(xcd_grid(Tg)), dim3(256), 0, s, g, (int)cyclic, A, h, w.R, w.val[c], w.pos[pc], w.gord, w.key[c], Tg);
    }
    h = h < (1u << 29) ? h * 2 : h;
    bits = 20 + bits_for(NG ? NG - 1 : 0);
  }
  hipLaunchKernelGGL(bwt_pidx, dim3(nb), dim3(256), 0, s, g, (int)cyclic, (int)(A != 0), w.R, d_pidx);
  if (!cyclic) {
    const uint32_t Tn = (M + RS_TILE - 1) / RS_TILE;
    hipLaunchKernelGGL(bwt_emit_sentinel, dim3(xcd_grid(Tn)), dim3(256), 0, s, d_T, g, M, w.SA, w.R, d_U, Tn);
  }
  CJS_HIP_TRY(hipGetLastError());
  if (stats) stats->bwt_rounds = rounds;
  if (stats && resolve_stats) {                  // (else the caller resolves w.lt once its stream has drained)
    CJS_HIP_TRY(hipStreamSynchronize(s));
    lt.resolve(stats);
  }
  return 0;
}

}  // namespace cjs
