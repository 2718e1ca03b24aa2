// mtf.hip — symbol map, move-to-front and RLE2 (RUNA/RUNB) of the BWT output, plus symbol histogram.
//
// Replaces J/Bzip2_joined_.js:2064-2139 (used map, inline MTF with O(n*|alphabet|) search, zero-run
// coding, freq[]).  The reference walks U serially.  Parallel formulation:
//   * only RUN HEADS of U (U[p] != U[p-1]) can have a non-zero MTF rank; inside a run the rank is 0.
//     So: compact the heads (position, symbol), MTF only over the heads, and derive the RLE2 symbols
//     of each (head, run length) pair locally: [rank+1] followed by the bijective base-2 digits of
//     (run length - 1).
//   * MTF over heads is chunked (512 heads per chunk).  The list at a chunk start is "symbols ordered
//     by last occurrence before the chunk" = a rank-by-counting over <=256 keys, produced by one
//     workgroup per block that walks the chunks; the chunks are then replayed independently, one lane
//     per chunk with its list in LDS (padded rows, no bank aliasing between lanes).
#include "cjs_internal.h"
#include "prims.hpp"
#include "mtf.h"

namespace cjs {

#ifndef CJS_MTF_CHUNK
#define CJS_MTF_CHUNK 512
#endif
#ifndef CJS_MTF_CL_THREADS
#define CJS_MTF_CL_THREADS 256      /* ms_mtf 1.50 with 1024, 1.44 with 512, 1.40 with 256 (100 MB text, 512-head chunks; 256-head chunks: 1.46-1.56) */
#endif
constexpr int MTF_CHUNK = CJS_MTF_CHUNK;
constexpr int MTF_CL_THREADS = CJS_MTF_CL_THREADS;      // workgroup size of mtf_chunk_lists (a multiple of 256)

// ---- A: used-symbol list + run-head compaction.  Tiles of 16 KiB: count (16 bytes per thread: the workgroup scan and its
// barriers are per tile, at 4 bytes per thread they were most of the kernel: 98 -> 42 us) -> per-block scan -> write.
// tcnt[blk * tpb + tile] and the used flags (uflag[blk * 258 + byte], zeroed by the host) alias buffers that
// are not live yet (segkeys, freq).
constexpr uint32_t MT_TILE = 4096;          // heads per tile of the emit kernels
constexpr uint32_t HT_PER = 16, HT_TILE = 1024 * HT_PER;      // bytes per thread / per tile of the head kernels
// the heads of bytes p0 .. p0+3 of row u: flag mask, the four bytes; prev = byte p0-1
__device__ __forceinline__ uint32_t head_flags4(const uint8_t* __restrict__ u, bool row0, uint32_t p0, uint32_t n, uint32_t& mine) {
  mine = 0;
  if (p0 >= n) return 0u;
  // bytes p0-1 .. p0+3 from three aligned 32-bit words (block rows start at odd addresses: the lanes' own 4 bytes straddle
  // two words); the word before the very first byte of U is not touched
  const uintptr_t a = (uintptr_t)(u + p0);
  const uint32_t* al = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
  const uint32_t sh = (uint32_t)(a & 3);
  const bool first = row0 && p0 == 0;
  const uint32_t wm = first ? 0u : al[-1], w0 = al[0], w1 = sh ? al[1] : 0u;        // al[1] only when the 4 bytes straddle
  mine = __builtin_amdgcn_alignbyte(w1, w0, sh);                                   // bytes p0 .. p0+3
  uint32_t prev = __builtin_amdgcn_alignbyte(w0, wm, sh) >> 24;                      // byte p0-1
  uint32_t fm = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t p = p0 + j, c = (mine >> (8 * j)) & 0xFFu;
    if (p < n && (p == 0 || c != prev)) fm |= 1u << j;                              // every used byte value starts a run
    prev = c;
  }
  return fm;
}
// count pass: 16 bytes per thread (one workgroup scan per 16 KiB)
__global__ __launch_bounds__(1024) void mtf_head_count(const uint8_t* __restrict__ U, uint32_t stride, const uint32_t* __restrict__ blen,
                                                       uint32_t* __restrict__ tcnt, uint32_t tpb, uint32_t* __restrict__ uflag) {
  __shared__ uint32_t used[256];
  __shared__ uint32_t sm[16];
  const uint32_t blk = blockIdx.y, tile = blockIdx.x, n = blen[blk];
  const uint32_t base = tile * HT_TILE;
  if (base >= n) { if (threadIdx.x == 0) tcnt[(size_t)blk * tpb + tile] = 0; return; }
  const uint8_t* u = U + (size_t)blk * stride;
  if (threadIdx.x < 256) used[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t p0 = base + threadIdx.x * HT_PER;
  uint32_t m4[4] = {0, 0, 0, 0};               // bytes p0 .. p0+15
  uint32_t prev = 0, cnt = 0;
  if (p0 < n) {
    // bytes p0-1 .. p0+15 from aligned 32-bit words; the word before the very first byte of U is not touched, the word behind
    // the sixteen bytes only when they straddle it
    const uintptr_t a = (uintptr_t)(u + p0);
    const uint32_t* al = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(a & 3);
    const bool first = blk == 0 && p0 == 0;
    const uint32_t wm = first ? 0u : al[-1], w0 = al[0], w1 = al[1], w2 = al[2], w3 = al[3], w4 = sh ? al[4] : 0u;
    m4[0] = __builtin_amdgcn_alignbyte(w1, w0, sh); m4[1] = __builtin_amdgcn_alignbyte(w2, w1, sh);
    m4[2] = __builtin_amdgcn_alignbyte(w3, w2, sh); m4[3] = __builtin_amdgcn_alignbyte(w4, w3, sh);
    prev = __builtin_amdgcn_alignbyte(w0, wm, sh) >> 24;                              // byte p0-1
  }
#pragma unroll
  for (int j = 0; j < (int)HT_PER; j++) {
    const uint32_t p = p0 + j;
    const uint32_t c = p < n ? (m4[j >> 2] >> (8 * (j & 3))) & 0xFFu : 0u;
    if (p < n && (p == 0 || c != prev)) { cnt++; used[c] = 1; }
    prev = c;
  }
  const uint32_t tot = block_sum<1024>(cnt, sm);         // (barrier inside: used[] is complete)
  if (threadIdx.x == 0) tcnt[(size_t)blk * tpb + tile] = tot;
  if (threadIdx.x < 256 && used[threadIdx.x]) uflag[(size_t)blk * 258 + threadIdx.x] = 1;
}
// write pass: the same 16 KiB tiles in four rounds of 4 bytes per thread (a thread's stores go to consecutive slots: with
// sixteen bytes per thread the sixteen conditional stores took 240 us instead of 134)
__global__ __launch_bounds__(1024) void mtf_head_write(const uint8_t* __restrict__ U, uint32_t stride, const uint32_t* __restrict__ blen,
                                                       MtfBufs mb, const uint32_t* __restrict__ tcnt, uint32_t tpb) {
  __shared__ uint32_t sm[16];
  const uint32_t blk = blockIdx.y, tile = blockIdx.x, n = blen[blk];
  const uint32_t base = tile * HT_TILE;
  if (base >= n) return;
  const uint8_t* u = U + (size_t)blk * stride;
  uint32_t* hpos = mb.hpos + (size_t)blk * mb.hstride;
  uint8_t* hsym = mb.hsym + (size_t)blk * mb.hstride;
  uint32_t run = tcnt[(size_t)blk * tpb + tile];
  for (uint32_t r = 0; r < HT_TILE / 4096u; r++) {
    const uint32_t p0 = base + r * 4096u + threadIdx.x * 4u;
    if (base + r * 4096u >= n) break;
    uint32_t mine;
    const uint32_t fm = head_flags4(u, blk == 0, p0, n, mine);
    uint32_t tot;
    uint32_t o = run + block_excl_sum<1024>((uint32_t)__builtin_popcount(fm), sm, tot);
#pragma unroll
    for (int j = 0; j < 4; j++) if ((fm >> j) & 1u) { hpos[o] = p0 + j; hsym[o] = (uint8_t)(mine >> (8 * j)); o++; }
    run += tot;
  }
}
// per block: exclusive scan of the tile counts (tpb <= 1024), number of heads, used-symbol list
__global__ __launch_bounds__(1024) void mtf_head_scan(MtfBufs mb, uint32_t* __restrict__ tcnt, uint32_t tpb, const uint32_t* __restrict__ uflag) {
  __shared__ uint32_t sm[16];
  const uint32_t blk = blockIdx.x;
  uint32_t tot;
  const uint32_t v = threadIdx.x < tpb ? tcnt[(size_t)blk * tpb + threadIdx.x] : 0u;
  const uint32_t ex = block_excl_sum<1024>(v, sm, tot);
  if (threadIdx.x < tpb) tcnt[(size_t)blk * tpb + threadIdx.x] = ex;
  if (threadIdx.x == 0) mb.nheads[blk] = tot;
  const uint32_t f = threadIdx.x < 256 ? uflag[(size_t)blk * 258 + threadIdx.x] : 0u;
  uint32_t asz;
  const uint32_t ax = block_excl_sum<1024>(f, sm, asz);
  if (f) mb.alist[(size_t)blk * 256 + ax] = (uint8_t)threadIdx.x;
  if (threadIdx.x == 0) mb.asz[blk] = asz;
}

// ---- C1: MTF list at the start of every chunk of 512 heads.
// The list before a chunk = used symbols ordered by (last occurrence before the chunk, descending), never-seen
// symbols after them in ascending byte order.  Three kernels so that the serial chain is 32 chunks long, not ~1000:
//   mtf_seg_last  : per segment of 32 chunks, last head index of every byte value inside the segment
//   mtf_seg_scan  : per block, running maximum over segments -> sort keys at every segment start
//   mtf_chunk_lists: per segment, walk its 32 chunks: rank-by-counting over the <=256 keys, then fold the chunk in
constexpr int MTF_SEG = 32;     // chunks per segment
__global__ __launch_bounds__(256) void mtf_seg_last(MtfBufs mb) {
  __shared__ int last[4][256];           // four copies by lane: a few symbols carry most heads (same-address LDS atomics are serial)
  const uint32_t blk = blockIdx.y, seg = blockIdx.x, H = mb.nheads[blk];
  const uint32_t h0 = seg * MTF_SEG * MTF_CHUNK;
  if (h0 >= H) return;
  const uint32_t h1 = h0 + MTF_SEG * MTF_CHUNK < H ? h0 + MTF_SEG * MTF_CHUNK : H;
  const uint8_t* hsym = mb.hsym + (size_t)blk * mb.hstride;
#pragma unroll
  for (int r = 0; r < 4; r++) last[r][threadIdx.x] = -1;
  __syncthreads();
  int* mine = last[threadIdx.x & 3];
  for (uint32_t h = h0 + threadIdx.x; h < h1; h += 256) atomicMax(&mine[hsym[h]], (int)h);
  __syncthreads();
  const int a01 = last[0][threadIdx.x] > last[1][threadIdx.x] ? last[0][threadIdx.x] : last[1][threadIdx.x];
  const int a23 = last[2][threadIdx.x] > last[3][threadIdx.x] ? last[2][threadIdx.x] : last[3][threadIdx.x];
  mb.segkeys[((size_t)blk * mb.seg_stride + seg) * 256 + threadIdx.x] = a01 > a23 ? a01 : a23;
}
__global__ __launch_bounds__(256) void mtf_seg_scan(MtfBufs mb) {
  __shared__ uint32_t used[256];
  const uint32_t blk = blockIdx.x, H = mb.nheads[blk], asz = mb.asz[blk];
  const uint32_t nseg = (H + MTF_SEG * MTF_CHUNK - 1) / (MTF_SEG * MTF_CHUNK);
  used[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x < asz) used[mb.alist[(size_t)blk * 256 + threadIdx.x]] = 1;
  __syncthreads();
  const int d = threadIdx.x;
  int key = used[d] ? 255 - d : -1000;                 // never seen: ascending byte order behind every seen symbol
  int* sk = mb.segkeys + (size_t)blk * mb.seg_stride * 256;
  for (uint32_t sgi = 0; sgi < nseg; sgi++) {
    const int l = sk[(size_t)sgi * 256 + d];            // last head index of d inside segment sgi (or -1)
    sk[(size_t)sgi * 256 + d] = key;                    // becomes: key of d at the START of segment sgi
    if (l >= 0) key = 256 + l;
  }
}
__global__ __launch_bounds__(MTF_CL_THREADS) void mtf_chunk_lists(MtfBufs mb) {
  __shared__ int keys[256];            // sort key of the u-th USED symbol (text uses ~100 of the 256 byte values: the
  __shared__ uint8_t symof[256];       // rank-by-counting is quadratic in the number of symbols that take part)
  __shared__ uint8_t uof[256];         // byte value -> used index
  __shared__ uint32_t part[4][256];
  const uint32_t blk = blockIdx.y, seg = blockIdx.x, H = mb.nheads[blk], nu = mb.asz[blk];
  const uint32_t nch = (H + MTF_CHUNK - 1) / MTF_CHUNK;
  const uint32_t c0 = seg * MTF_SEG;
  if (c0 >= nch) return;
  const uint32_t c1 = c0 + MTF_SEG < nch ? c0 + MTF_SEG : nch;
  const uint8_t* hsym = mb.hsym + (size_t)blk * mb.hstride;
  uint8_t* lists = mb.lists + (size_t)blk * mb.list_stride;
  if (threadIdx.x < nu) {
    const uint8_t sym = mb.alist[(size_t)blk * 256 + threadIdx.x];
    symof[threadIdx.x] = sym; uof[sym] = (uint8_t)threadIdx.x;
    keys[threadIdx.x] = mb.segkeys[((size_t)blk * mb.seg_stride + seg) * 256 + sym];
  }
  __syncthreads();
  constexpr uint32_t NQ = MTF_CL_THREADS / 256;
  const uint32_t d = threadIdx.x & 255u, q = threadIdx.x >> 8, per = (nu + NQ - 1u) / NQ;
  const uint32_t j0 = q * per < nu ? q * per : nu, j1 = j0 + per < nu ? j0 + per : nu;
  for (uint32_t c = c0; c < c1; c++) {
    if (d < nu) {
      const int kd = keys[d];
      uint32_t cnt = 0;
      for (uint32_t j = j0; j < j1; j++) cnt += keys[j] > kd;
      part[q][d] = cnt;
    }
    __syncthreads();
    if (threadIdx.x < nu) {
      uint32_t p = 0;
#pragma unroll
      for (uint32_t qq = 0; qq < NQ; qq++) p += part[qq][d];
      lists[(size_t)c * 256 + p] = symof[d];
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < MTF_CHUNK; i += MTF_CL_THREADS) {
      const uint32_t h = c * MTF_CHUNK + i;
      if (h < H) atomicMax(&keys[uof[hsym[h]]], (int)(256 + h));
    }
    __syncthreads();
  }
}

// ---- C2: replay each chunk from its start list; one lane per chunk.
// The first 16 list positions live in four registers (byte k of the 128-bit value = list position k): finding a
// symbol there is a SWAR zero-byte test and the move-to-front is a byte shift under a mask, no memory access.
// Positions >= 16 stay in LDS (row stride 260 B, no bank aliasing between lanes) and are only walked for the few
// heads whose rank is that large.
constexpr int LROW = 260;
__device__ __forceinline__ uint32_t zero_bytes(uint32_t v) { return (v - 0x01010101u) & ~v & 0x80808080u; }   // lowest set bit is exact
__device__ __forceinline__ uint32_t low_bytes_mask(int nb) { return nb <= 0 ? 0u : nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u); }
constexpr int MTF_RW = 16;                 // list words held in registers (4 positions each)
__global__ __launch_bounds__(256) void mtf_replay(uint32_t stride, MtfBufs mb) {
  __shared__ uint8_t L[256 * LROW];
  const uint32_t blk = blockIdx.y, H = mb.nheads[blk], asz = mb.asz[blk];
  const uint32_t nch = (H + MTF_CHUNK - 1) / MTF_CHUNK;
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x * 256 >= nch) return;
  if (c >= nch) return;
  const uint8_t* hsym = mb.hsym + (size_t)blk * mb.hstride;
  uint8_t* hrank = mb.hrank + (size_t)blk * mb.hstride;
  const uint8_t* lst = mb.lists + (size_t)blk * mb.list_stride + (size_t)c * 256;
  uint8_t* my = L + threadIdx.x * LROW;
  constexpr uint32_t NR = 4 * MTF_RW;      // positions in registers
  uint32_t l[MTF_RW];
#pragma unroll
  for (int k = 0; k < MTF_RW; k += 4) {    // rows of lists[] are 256-byte aligned; bytes >= asz are never matched first
    const uint4 f = *reinterpret_cast<const uint4*>(lst + 4 * k);
    l[k] = f.x; l[k + 1] = f.y; l[k + 2] = f.z; l[k + 3] = f.w;
  }
  for (uint32_t j = NR; j < asz; j++) my[j] = lst[j];
  const uint32_t h0 = c * MTF_CHUNK, h1 = h0 + MTF_CHUNK < H ? h0 + MTF_CHUNK : H;
  // 16 heads at a time: one 16-byte load of symbols, one 16-byte store of ranks per lane (hsym/hrank rows
  // are 256-byte aligned: h0 is a multiple of 256 and the per-block stride is padded to 16)
  uint4 sv_next = *reinterpret_cast<const uint4*>(hsym + h0);
  for (uint32_t hb = h0; hb < h1; hb += 16) {
    const uint4 sv = sv_next;
    if (hb + 16 < h1) sv_next = *reinterpret_cast<const uint4*>(hsym + hb + 16);      // in flight while these 16 heads are replayed
    uint32_t sw[4] = {sv.x, sv.y, sv.z, sv.w}, rw[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const uint32_t h = hb + q;
      uint32_t r = 0;
      if (h < h1) {
        const uint32_t s8 = (sw[q >> 2] >> (8 * (q & 3))) & 0xFFu;
        const uint32_t bc = s8 * 0x01010101u;
        r = NR;
#pragma unroll
        for (int k = MTF_RW - 1; k >= 0; k--) {
          const uint32_t z = zero_bytes(l[k] ^ bc);
          if (z) r = 4u * (uint32_t)k + ((uint32_t)__builtin_ctz(z) >> 3);
        }
        if (r == NR) {
          uint8_t prev = (uint8_t)(l[MTF_RW - 1] >> 24);      // falls out of the register part
          for (; r < asz; r++) {
            const uint8_t x = my[r];
            my[r] = prev;
            prev = x;
            if (x == (uint8_t)s8) break;
          }
        }
        // positions 0..min(r,NR-1) shift up by one, the symbol goes to the front
        const int nb = (int)(r < NR ? r : NR - 1) + 1;
        uint32_t carry_in = s8;
#pragma unroll
        for (int k = 0; k < MTF_RW; k++) {
          const uint32_t sh = (l[k] << 8) | carry_in;
          carry_in = l[k] >> 24;
          const uint32_t m = low_bytes_mask(nb - 4 * k);
          l[k] = (sh & m) | (l[k] & ~m);
        }
      }
      rw[q >> 2] |= r << (8 * (q & 3));
    }
    *reinterpret_cast<uint4*>(hrank + hb) = make_uint4(rw[0], rw[1], rw[2], rw[3]);
  }
}

// ---- D: RLE2 symbols + histogram.  Tiles of 16 Ki heads: count (four times 4 heads per thread, one workgroup reduction per tile) ->
// per-block scan -> write (four rounds of 4 heads per thread; + tile histogram folded into the block's freq[] with atomics;
// freq is zeroed by the host, the scan kernel adds the end-of-block symbol).
constexpr uint32_t ET_TILE = 4 * MT_TILE;
// the four heads h0 .. h0+3 (h0 a multiple of 4): literal flag, zero-run length behind the head, its digit count, rank; returns
// the number of RLE2 symbols they emit.  Ranks as one 32-bit load, positions as one 128-bit load (+ the next head's position):
// rows of hrank / hpos are 16-element aligned
__device__ __forceinline__ uint32_t emit_heads4(const uint32_t* __restrict__ hpos, const uint8_t* __restrict__ hrank, uint32_t h0, uint32_t H, uint32_t n,
                                                uint32_t (&lit)[4], uint32_t (&z)[4], uint32_t (&nd)[4], uint32_t (&rk)[4]) {
  uint32_t rk4 = 0, hp[5] = {0, 0, 0, 0, 0}, cnt = 0;
  if (h0 < H) {
    rk4 = *reinterpret_cast<const uint32_t*>(hrank + h0);
    const uint4 p4 = *reinterpret_cast<const uint4*>(hpos + h0);
    hp[0] = p4.x; hp[1] = p4.y; hp[2] = p4.z; hp[3] = p4.w;
    hp[4] = h0 + 4 < H ? hpos[h0 + 4] : n;
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t h = h0 + j;
    lit[j] = z[j] = nd[j] = rk[j] = 0;
    if (h < H) {
      const uint32_t r = (rk4 >> (8 * j)) & 0xFFu;
      const uint32_t len = (h + 1 < H ? hp[j + 1] : n) - hp[j];
      lit[j] = r != 0;
      rk[j] = r;
      z[j] = len - lit[j];
      nd[j] = z[j] ? 31u - (uint32_t)__builtin_clz(z[j] + 1u) : 0u;
      cnt += lit[j] + nd[j];
    }
  }
  return cnt;
}
__global__ __launch_bounds__(1024) void mtf_emit_count(const uint32_t* __restrict__ blen, MtfBufs mb, uint32_t* __restrict__ tcnt, uint32_t tpb) {
  __shared__ uint32_t sm[16];
  const uint32_t blk = blockIdx.y, tile = blockIdx.x, n = blen[blk], H = mb.nheads[blk];
  const uint32_t base = tile * ET_TILE;
  if (base >= H) { if (threadIdx.x == 0) tcnt[(size_t)blk * tpb + tile] = 0; return; }
  const uint32_t* hpos = mb.hpos + (size_t)blk * mb.hstride;
  const uint8_t* hrank = mb.hrank + (size_t)blk * mb.hstride;
  uint32_t cnt = 0;
#pragma unroll
  for (uint32_t q = 0; q < 4; q++) {
    uint32_t lit[4], z[4], nd[4], rk[4];
    cnt += emit_heads4(hpos, hrank, base + q * MT_TILE + threadIdx.x * 4u, H, n, lit, z, nd, rk);      // (16 consecutive heads per thread: 78 us, the loads of a wave 64 B apart)
  }
  cnt = block_sum<1024>(cnt, sm);
  if (threadIdx.x == 0) tcnt[(size_t)blk * tpb + tile] = cnt;
}
__global__ __launch_bounds__(1024) void mtf_emit_write(const uint32_t* __restrict__ blen, MtfBufs mb, const uint32_t* __restrict__ tcnt, uint32_t tpb) {
  // symbol counts of the tile: ranks 1-3 and the two run digits make up nearly all symbols, so the counters are kept in eight
  // copies (by lane) and the digits are summed per wave first -- one LDS atomic per same-address lane is done after the other
  __shared__ uint32_t freq[8 * 258];
  __shared__ uint32_t sm[16];
  const uint32_t blk = blockIdx.y, tile = blockIdx.x, n = blen[blk], H = mb.nheads[blk];
  const uint32_t base = tile * ET_TILE;
  if (base >= H) return;
  const uint32_t* hpos = mb.hpos + (size_t)blk * mb.hstride;
  const uint8_t* hrank = mb.hrank + (size_t)blk * mb.hstride;
  for (int i = threadIdx.x; i < 8 * 258; i += 1024) freq[i] = 0;
  __syncthreads();
  uint16_t* A = mb.A + (size_t)blk * mb.a_stride;
  uint32_t* rep = freq + (threadIdx.x & 7u) * 258u;
  uint32_t n1 = 0, n0 = 0;                              // RUNB / RUNA digits of this thread
  uint32_t run = tcnt[(size_t)blk * tpb + tile];
  for (uint32_t r = 0; r < ET_TILE / MT_TILE && base + r * MT_TILE < H; r++) {
    uint32_t lit[4], z[4], nd[4], rk[4];
    const uint32_t cnt = emit_heads4(hpos, hrank, base + r * MT_TILE + threadIdx.x * 4u, H, n, lit, z, nd, rk);
    uint32_t tot;
    uint32_t o = run + block_excl_sum<1024>(cnt, sm, tot);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (lit[j]) { A[o++] = (uint16_t)(rk[j] + 1); atomicAdd(&rep[rk[j] + 1], 1u); }
      const uint32_t v = z[j] + 1u;
      uint32_t nb1 = 0;
      for (uint32_t i = 0; i < nd[j]; i++) { const uint32_t bit = (v >> i) & 1u; A[o++] = (uint16_t)bit; nb1 += bit; }
      n1 += nb1; n0 += nd[j] - nb1;
    }
    run += tot;
  }
  n1 = wave_sum(n1); n0 = wave_sum(n0);
  if (lane_id() == 0) { if (n1) atomicAdd(&freq[1], n1); if (n0) atomicAdd(&freq[0], n0); }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < 258; i += 1024) {
    uint32_t f = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) f += freq[r * 258 + i];
    if (f) atomicAdd(&mb.freq[(size_t)blk * 258 + i], f);
  }
}
__global__ __launch_bounds__(1024) void mtf_emit_scan(MtfBufs mb, uint32_t* __restrict__ tcnt, uint32_t tpb) {
  __shared__ uint32_t sm[16];
  const uint32_t blk = blockIdx.x, asz = mb.asz[blk];
  uint32_t tot;
  const uint32_t v = threadIdx.x < tpb ? tcnt[(size_t)blk * tpb + threadIdx.x] : 0u;
  const uint32_t ex = block_excl_sum<1024>(v, sm, tot);
  if (threadIdx.x < tpb) tcnt[(size_t)blk * tpb + threadIdx.x] = ex;
  if (threadIdx.x == 0) {                               // end-of-block symbol (Bzip2:2137-2139)
    mb.A[(size_t)blk * mb.a_stride + tot] = (uint16_t)(asz + 1);
    mb.freq[(size_t)blk * 258 + asz + 1] = 1;
    mb.npos[blk] = tot + 1;
  }
}

// ------------------------------------------------------------------------------------------
size_t MtfWork::bytes_needed(size_t max_blocks, uint32_t stride) {
  size_t b = 0;
  auto add = [&](size_t n) { b += (n + 255) & ~(size_t)255; };
  const size_t ls = list_stride_for(stride), as = a_stride_for(stride), hs = hstride_for(stride);
  add(max_blocks * hs * 4); add(max_blocks * hs); add(max_blocks * hs); add(max_blocks * ls);
  add(max_blocks * as * 2); add(max_blocks * 258 * 4); add(max_blocks * 256);
  add(max_blocks * 4); add(max_blocks * 4); add(max_blocks * 4);
  add(max_blocks * seg_stride_for(stride) * 256 * 4);
  return b + 4096;
}
int MtfWork::carve(Arena& a, size_t max_blocks_, uint32_t stride_) {
  max_blocks = max_blocks_; stride = stride_;
  b.list_stride = list_stride_for(stride); b.a_stride = a_stride_for(stride); b.hstride = hstride_for(stride);
  b.hpos = a.take<uint32_t>(max_blocks * b.hstride); b.hsym = a.take<uint8_t>(max_blocks * b.hstride); b.hrank = a.take<uint8_t>(max_blocks * b.hstride);
  b.lists = a.take<uint8_t>(max_blocks * b.list_stride);
  b.A = a.take<uint16_t>(max_blocks * b.a_stride); b.freq = a.take<uint32_t>(max_blocks * 258); b.alist = a.take<uint8_t>(max_blocks * 256);
  b.asz = a.take<uint32_t>(max_blocks); b.nheads = a.take<uint32_t>(max_blocks); b.npos = a.take<uint32_t>(max_blocks);
  b.seg_stride = seg_stride_for(stride);
  b.segkeys = a.take<int>(max_blocks * b.seg_stride * 256);
  return b.segkeys ? 0 : CJS_E_OUT_OF_MEMORY;
}

int mtf_run(hipStream_t s, MtfWork& w, const uint8_t* d_U, uint32_t nb, const uint32_t* d_blen) {
  if (nb == 0) return 0;
  if (nb > w.max_blocks) return CJS_E_INVALID_ARG;
  const uint32_t max_chunks = (w.stride + MTF_CHUNK - 1) / MTF_CHUNK;
  // tile counters alias segkeys (dead before mtf_seg_last and after mtf_chunk_lists); used flags alias freq
  const uint32_t tpb = (w.stride + MT_TILE - 1) / MT_TILE;
  if (tpb > 1024 || (size_t)tpb > (size_t)w.b.seg_stride * 256) return CJS_E_INVALID_ARG;
  uint32_t* tcnt = reinterpret_cast<uint32_t*>(w.b.segkeys);
  dev_fill(s, w.b.freq, 0, (size_t)nb * 258 * 4);
  const uint32_t tph = (w.stride + HT_TILE - 1) / HT_TILE;           // tiles of the head kernels (<= tpb)
  hipLaunchKernelGGL(mtf_head_count, dim3(tph, nb), dim3(1024), 0, s, d_U, w.stride, d_blen, tcnt, tph, w.b.freq);
  hipLaunchKernelGGL(mtf_head_scan, dim3(nb), dim3(1024), 0, s, w.b, tcnt, tph, w.b.freq);
  hipLaunchKernelGGL(mtf_head_write, dim3(tph, nb), dim3(1024), 0, s, d_U, w.stride, d_blen, w.b, tcnt, tph);
  const uint32_t max_segs = (max_chunks + MTF_SEG - 1) / MTF_SEG;
  hipLaunchKernelGGL(mtf_seg_last, dim3(max_segs, nb), dim3(256), 0, s, w.b);
  hipLaunchKernelGGL(mtf_seg_scan, dim3(nb), dim3(256), 0, s, w.b);
  hipLaunchKernelGGL(mtf_chunk_lists, dim3(max_segs, nb), dim3(MTF_CL_THREADS), 0, s, w.b);
  hipLaunchKernelGGL(mtf_replay, dim3((max_chunks + 255) / 256, nb), dim3(256), 0, s, w.stride, w.b);
  dev_fill(s, w.b.freq, 0, (size_t)nb * 258 * 4);
  const uint32_t tpe = (w.stride + ET_TILE - 1) / ET_TILE;           // tiles of the emit kernels
  hipLaunchKernelGGL(mtf_emit_count, dim3(tpe, nb), dim3(1024), 0, s, d_blen, w.b, tcnt, tpe);
  hipLaunchKernelGGL(mtf_emit_scan, dim3(nb), dim3(1024), 0, s, w.b, tcnt, tpe);
  hipLaunchKernelGGL(mtf_emit_write, dim3(tpe, nb), dim3(1024), 0, s, d_blen, w.b, tcnt, tpe);
  CJS_HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace cjs
