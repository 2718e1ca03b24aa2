// huff.hip — bzip2 entropy stage on the GPU: multi-table canonical Huffman construction
// (the reference's own heuristic, bit-exact), bit-length accounting and bit packing.
//
// Replaces J/Bzip2_joined_.js: StaticHuffman ctor :1866-1894 + HuffmanAllocator :1085-1301,
// assignSelectors :1989-2004, optimizeHuffmanGroups :2005-2054, table count :2150-2163,
// selector MTF/unary :2167-2182, StaticHuffman.emit/computeCanonical/encode :1896-1951,
// data loop :2189-2194, BitStream.writeBits :154-166.
// One workgroup per block: the work is tiny (<= 18 k groups of 50 symbols, <= 6 tables of <= 258
// symbols) but its logic is serial, so tables are built one per wave (rank sort by the wave,
// in-place length-limited allocator on lane 0 with the node array in LDS) and everything that is
// per-group or per-symbol is data-parallel across the 1024 lanes.  Bit packing: wave prefix-scan of
// code lengths -> bit offsets -> words assembled in LDS -> coalesced big-endian stores.
#include "cjs_internal.h"
#include "prims.hpp"
#include "huff.h"

namespace cjs {

constexpr int MAXSYM = 258;
constexpr int MAX_BITS = 20;
constexpr int GSZ = 50;
constexpr int PD_GROUPS = 80;            // groups per data-packing tile (4000 symbols)

// phase clock of workgroup 0 (CJS_DEBUG only): 100 MHz ticks at the marks of huff_block
__device__ uint64_t g_huff_clk[32];
__device__ uint64_t g_bt_clk[8];          // table-build phase clock of (block 0, wave 0), CJS_DEBUG
#define BT_MARK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_bt_clk[i] = wall_clock64(); } while (0)
#define HB_MARK(i) do { if (dbg && blockIdx.x == 0 && threadIdx.x == 0) g_huff_clk[i] = wall_clock64(); } while (0)

// ------------------------------------------------------------------ allocator (lane 0, LDS array)
// a[] entries below the root are extended parent pointers p or p + len (p < len), so "% len" is one conditional subtract
__device__ __forceinline__ int ha_mod(int x, int len) { return x >= len ? x - len : x; }
__device__ __forceinline__ int ha_first(const int* a, int len, int i, int nodes_to_move) {          // Bzip2:1135-1156
  const int limit = i;
  int k = len - 2;
  while (i >= nodes_to_move && ha_mod(a[i], len) > limit) { k = i; i -= (limit - i + 1); }
  if (i < nodes_to_move - 1) i = nodes_to_move - 1;
  while (k > i + 1) {
    const int mid = (i + k) >> 1;
    if (ha_mod(a[mid], len) > limit) k = mid; else i = mid;
  }
  return k;
}
// pass 1 (one lane): extended parent pointers (Bzip2:1162-1186); false when len < 3 (lengths written, nothing left to do)
__device__ __forceinline__ bool ha_pass1(int* a, int len) {
  if (len == 2) { a[1] = 1; a[0] = 1; return false; }
  if (len == 1) { a[0] = 1; return false; }
  a[0] += a[1];
  // the two queue fronts live in registers two deep (H = a[head], Hn = a[head+1], T = a[top], Tn = a[top+1]), so the
  // LDS reads that refill them overlap the comparisons instead of sitting on the critical path
  int head = 0, top = 2;
  int H = a[0], Hn = 0;
  int T = top < len ? a[top] : 0, Tn = top + 1 < len ? a[top + 1] : 0;
  for (int tail = 1; tail < len - 1; tail++) {
    int w;
    if (top >= len || H < T) { w = H; a[head++] = tail; H = Hn; if (head + 1 < tail) Hn = a[head + 1]; }
    else { w = T; top++; T = Tn; if (top + 1 < len) Tn = a[top + 1]; }
    if (top >= len || (head < tail && H < T)) { w += H; a[head++] = tail + len; H = Hn; if (head + 1 < tail) Hn = a[head + 1]; }
    else { w += T; top++; T = Tn; if (top + 1 < len) Tn = a[top + 1]; }
    a[tail] = w;
    if (head == tail) H = w; else if (head + 1 == tail) Hn = w;
  }
  return true;
}
// Passes 2 and 3 by the whole wave.  ha_first() is a search for the first node whose parent lies above `limit`, and parents
// are non-decreasing along the array, so it is ONE ballot per 64 nodes over a register copy of the parent pointers instead of
// ~10 dependent LDS reads of one lane (~60 ns each); the fills of pass 3 are one masked store.  Only nodes at or below `limit`
// are searched: pass 3 never overwrites those (it fills from the top down to `next` >= `first` > limit), and the node at `limit`
// always qualifies (its parent was created after it), which is the case in which the serial rule stays inside [lo, limit] too.
// Should it not qualify, the serial rule is applied as it stands.  All control values are wave-uniform.
__device__ __forceinline__ int haw_first(const int (&pm)[5], const int* a, int len, int limit, int lo) {
  const int lane = lane_id();
#pragma unroll
  for (int t = 0; t < 5; t++) {
    if (64 * t <= limit) {
      const int idx = lane + 64 * t;
      const uint64_t m = __ballot(idx >= lo && idx <= limit && pm[t] > limit);
      if (m) return 64 * t + (int)__builtin_ctzll(m);
    }
  }
  return ha_first(a, len, limit, lo);
}
__device__ __forceinline__ void haw_fill(int* a, int lo, int hi, int x) {          // a[lo..hi] = x
  const int lane = lane_id();
#pragma unroll
  for (int t = 0; t < 5; t++) { const int idx = lane + 64 * t; if (idx >= lo && idx <= hi) a[idx] = x; }
}
__device__ __forceinline__ void ha_passes23_wave(int* a, int len, int maxlen) {
  const int lane = lane_id();
  int pm[5];
#pragma unroll
  for (int t = 0; t < 5; t++) { const int idx = lane + 64 * t; pm[t] = idx < len ? ha_mod(a[idx], len) : 0; }
  const int root0 = __builtin_amdgcn_readfirstlane(pm[0]);                         // ha_mod(a[0], len)
  BT_MARK(2);
  // pass 2: nodes to relocate (Bzip2:1195-1204)
  int reloc = len - 2;
  for (int depth = 1; depth < maxlen - 1 && reloc > 1; depth++) reloc = haw_first(pm, a, len, reloc - 1, 0);
  BT_MARK(3);
  // pass 3
  if (root0 >= reloc) {                                                            // Bzip2:1211-1226
    int first = len - 2, next = len - 1;
    for (int depth = 1, avail = 2; avail > 0; depth++) {
      const int last = first;
      first = haw_first(pm, a, len, last - 1, 0);
      const int cnt = avail - (last - first);
      if (cnt > 0) { haw_fill(a, next - cnt + 1, next, depth); next -= cnt; }
      avail = (last - first) << 1;
    }
  } else {                                                                          // Bzip2:1235-1264
    const unsigned rm1 = (unsigned)(reloc - 1);
    const int insert_depth = maxlen - (rm1 ? 32 - __builtin_clz(rm1) : 0);      // Util.fls
    int first = len - 2, next = len - 1;
    int depth = insert_depth == 1 ? 2 : 1;
    int left = insert_depth == 1 ? reloc - 2 : reloc;
    for (int avail = depth << 1; avail > 0; depth++) {
      const int last = first;
      first = first <= reloc ? first : haw_first(pm, a, len, last - 1, reloc);
      int offset = 0;
      if (depth >= insert_depth) { offset = 1 << (depth - insert_depth); if (left < offset) offset = left; }
      else if (depth == insert_depth - 1) {
        offset = 1;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (__builtin_amdgcn_readfirstlane(a[first]) == last) first++;             // (the array as it stands: fills included, as in the serial form)
      }
      const int cnt = avail - (last - first + offset);
      if (cnt > 0) { haw_fill(a, next - cnt + 1, next, depth); next -= cnt; }
      left -= offset;
      avail = (last - first + offset) << 1;
    }
  }
}

// One wave builds one table: freq[0..n) -> lens[0..n).  key/work are per-table LDS scratch (n entries).
// (forced inline: as a call the LDS pointers are generic and every access a flat_load / flat_store.  Pass 1 of the allocator
// stays on one lane: a variant with the node array in the wave's registers -- v_readlane reads, scalar control flow -- ran its
// ~600 dependent steps no faster)
__device__ __forceinline__ void build_table_wave(const uint32_t* freq, uint8_t* lens, uint32_t* key, int* work, int n) {
  const int lane = lane_id();
  BT_MARK(0);
  for (int i = lane; i < n; i += 64) key[i] = (freq[i] << 9) | (uint32_t)i;       // Bzip2:1881-1883
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  int rk[5] = {0, 0, 0, 0, 0};
  uint32_t ki[5];
#pragma unroll
  for (int t = 0; t < 5; t++) { const int i = lane + 64 * t; ki[t] = i < n ? key[i] : 0u; }
#pragma unroll 6
  for (int k = 0; k < n; k++) {                 // one broadcast read per key, compared with all five of the lane's keys
    const uint32_t kv = key[k];
#pragma unroll
    for (int t = 0; t < 5; t++) rk[t] += kv < ki[t];
  }
  __builtin_amdgcn_wave_barrier();
  for (int t = 0, i = lane; t < 5; t++, i += 64) if (i < n) work[rk[t]] = (int)(freq[i]);   // sortedFreq
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  BT_MARK(1);
  int go = 0;
  if (lane == 0) go = ha_pass1(work, n) ? 1 : 0;
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  if (__builtin_amdgcn_readfirstlane(go)) ha_passes23_wave(work, __builtin_amdgcn_readfirstlane(n), MAX_BITS);
  BT_MARK(5);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  for (int t = 0, i = lane; t < 5; t++, i += 64) if (i < n) lens[i] = (uint8_t)work[rk[t]];
  BT_MARK(6);
}

struct HuffShared {
  uint32_t freq[6][260];
  uint32_t key[6][260];
  int work[6][260];
  uint8_t lens[6][264];
  uint32_t chist[1024];
  uint64_t packed[260];
  uint32_t counts[8];
  uint32_t misc[8];
  uint32_t sm[16];
};

// assignSelectors (Bzip2:1989-2004).  Cost of a 50-symbol group under all tables at once: the six code
// lengths of a symbol are packed as 10-bit fields of one u64 (50 * 20 < 1024, fields cannot overflow), so the
// inner loop is one LDS read + one 64-bit add per symbol.  Symbols are staged through LDS with coalesced
// loads (512 groups = 51,200 bytes per step); lane g then reads its 25 dwords at a 25-dword stride (no bank aliasing).
constexpr int AS_GROUPS = 512;
__device__ void assign_selectors(HuffShared& S, uint32_t* __restrict__ stage /* AS_GROUPS*25 dwords */, const uint16_t* __restrict__ A,
                                 uint32_t npos, uint32_t nsel, int ng, uint8_t* __restrict__ sel, uint16_t* __restrict__ bcost,
                                 uint32_t g_lo = 0, uint32_t g_hi = 0xFFFFFFFFu /* groups [g_lo, g_hi): a multiple of AS_GROUPS each */) {
  if (g_hi < nsel) nsel = g_hi;
  for (int s = threadIdx.x; s < MAXSYM; s += 1024) {
    uint64_t pk = 0;
    for (int j = 0; j < ng; j++) pk |= (uint64_t)S.lens[j][s] << (10 * j);
    S.packed[s] = pk;
  }
  const uint32_t* A32 = reinterpret_cast<const uint32_t*>(A);          // A rows are 16-byte aligned (a_stride % 8 == 0)
  const uint32_t ndw = (npos + 1) / 2;
  constexpr int NPRE = (AS_GROUPS * 25 + 1023) / 1024;                 // 13 dwords per lane and step
  uint32_t pre[NPRE];                                                  // next step's symbols: loaded while this step is costed
#pragma unroll
  for (int j = 0; j < NPRE; j++) { const uint32_t i = (uint32_t)j * 1024u + threadIdx.x; pre[j] = (i < AS_GROUPS * 25 && g_lo * 25 + i < ndw) ? A32[g_lo * 25 + i] : 0u; }
  const uint32_t half = threadIdx.x & 1u, gl = threadIdx.x >> 1;       // two lanes per group: dwords [0,13) and [13,25)
  for (uint32_t g0 = g_lo; g0 < nsel; g0 += AS_GROUPS) {
    __syncthreads();                                                   // previous step's readers are done (first step: packed[] is complete)
#pragma unroll
    for (int j = 0; j < NPRE; j++) { const uint32_t i = (uint32_t)j * 1024u + threadIdx.x; if (i < AS_GROUPS * 25) stage[i] = pre[j]; }
    __syncthreads();
    {
      const uint32_t dw1 = (g0 + AS_GROUPS) * 25;
#pragma unroll
      for (int j = 0; j < NPRE; j++) {
        const uint32_t i = (uint32_t)j * 1024u + threadIdx.x;
        pre[j] = (g0 + AS_GROUPS < nsel && i < AS_GROUPS * 25 && dw1 + i < ndw) ? A32[dw1 + i] : 0u;
      }
    }
    const uint32_t g = g0 + gl;
    uint64_t acc = 0;
    if (g < nsel) {
      const uint32_t off = g * GSZ, cnt = npos - off < GSZ ? npos - off : GSZ;
      const uint32_t* my = stage + gl * 25;
      const uint32_t j0 = half ? 13u : 0u, j1 = half ? 25u : 13u;
      for (uint32_t j = j0; j < j1; j++) {
        const uint32_t w = my[j];
        if (2 * j < cnt) acc += S.packed[w & 0xFFFFu];
        if (2 * j + 1 < cnt) acc += S.packed[w >> 16];
      }
    }
    acc += __shfl_xor(acc, 1, 64);
    if (g < nsel && half == 0) {
      int best = 0; uint32_t bc = (uint32_t)(acc & 1023u);
#pragma unroll
      for (int j = 1; j < 6; j++) { const uint32_t cj = (uint32_t)((acc >> (10 * j)) & 1023u); if (j < ng && cj < bc) { best = j; bc = cj; } }
      sel[g] = (uint8_t)best; bcost[g] = (uint16_t)bc;
    }
  }
  __syncthreads();
}

// optimizeHuffmanGroups, the split of one refinement (Bzip2:2012-2042): the most used table's groups are cut at the median
// cost (stable inside the median bin, Q16); the upper half moves to the new table ng
__device__ void median_split(HuffShared& S, uint8_t* __restrict__ sel, const uint16_t* __restrict__ bcost, uint32_t nsel, int ng) {
  if (threadIdx.x < 8) S.counts[threadIdx.x] = 0;
  __syncthreads();
  {                                                     // groups per table: wave ballots (all groups aim at <= 6 counters)
    uint32_t c6[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t g0 = 0; g0 < nsel; g0 += 1024) {
      const uint32_t g = g0 + threadIdx.x;
      const uint32_t sv = g < nsel ? sel[g] : 255u;
#pragma unroll
      for (int j = 0; j < 6; j++) c6[j] += (uint32_t)__popcll(__ballot(sv == (uint32_t)j));
    }
    if (lane_id() == 0) {
#pragma unroll
      for (int j = 0; j < 6; j++) if (c6[j]) atomicAdd(&S.counts[j], c6[j]);
    }
  }
  S.chist[threadIdx.x] = 0;
  __syncthreads();
  int which = 0;
  for (int j = 1; j < ng; j++) if (S.counts[j] > S.counts[which]) which = j;      // indexOf(max): first
  const uint32_t nsp = S.counts[which], m = nsp >> 1;
  for (uint32_t g = threadIdx.x; g < nsel; g += 1024) if (sel[g] == which) atomicAdd(&S.chist[bcost[g] & 1023u], 1u);
  __syncthreads();
  {
    const uint32_t hv = S.chist[threadIdx.x];
    uint32_t tot;
    const uint32_t cum = block_excl_sum<1024>(hv, S.sm, tot);
    if (hv && cum <= m && m < cum + hv) { S.misc[0] = threadIdx.x; S.misc[1] = cum; }
    __syncthreads();
  }
  const uint32_t cstar = S.misc[0], cumstar = S.misc[1];
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nsel; base += 1024) {        // stable order inside the median cost bin (Q16)
    const uint32_t g = base + threadIdx.x;
    uint32_t f = 0, mine = 0xFFFFu;
    if (g < nsel && sel[g] == which) { mine = bcost[g]; f = mine == cstar; }
    uint32_t tot;
    const uint32_t ex = block_excl_sum<1024>(f, S.sm, tot);
    if (g < nsel && mine != 0xFFFFu) {
      if (mine > cstar || (f && cumstar + carry + ex >= m)) sel[g] = (uint8_t)ng;
    }
    carry += tot;
  }
}

// symbol counts per table under the current selectors (Bzip2:2043-2048) of symbols [i_lo, i_hi) (i_lo a multiple of 8) into
// eight replicas of [6][260] counters in `stage` (zeroed here): MTF output is dominated by a few symbols, the replicas (by
// lane) keep the same-address LDS atomics apart; the caller sums them
__device__ void count_symbols(uint32_t* __restrict__ stage, const uint16_t* __restrict__ A, const uint8_t* __restrict__ sel,
                              uint32_t npos, uint32_t i_lo, uint32_t i_hi) {
  for (int i = threadIdx.x; i < 8 * 6 * 260; i += 1024) stage[i] = 0;
  __syncthreads();
  {
    uint32_t* rep = stage + (threadIdx.x & 7u) * (6 * 260);
    const uint4* A128 = reinterpret_cast<const uint4*>(A);                          // 8 symbols per load, two loads in flight
    const uint32_t nv = (i_hi < npos ? i_hi : npos) / 8;
    for (uint32_t v = i_lo / 8 + threadIdx.x; v < nv; v += 2048) {
      const uint32_t v2 = v + 1024;
      const uint4 x = A128[v];
      const uint4 y = v2 < nv ? A128[v2] : make_uint4(0, 0, 0, 0);
      const uint32_t i0 = v * 8, i1 = v2 * 8;
      const uint32_t ga = i0 / GSZ, gb = (i0 + 7) / GSZ, gc = v2 < nv ? i1 / GSZ : 0u, gd = v2 < nv ? (i1 + 7) / GSZ : 0u;
      const uint32_t sa = sel[ga], sb = sel[gb], sc = sel[gc], sd = sel[gd];
      const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t e0 = i0 + 2 * q, e1 = e0 + 1;
        atomicAdd(&rep[(e0 / GSZ == ga ? sa : sb) * 260 + (xs[q] & 0xFFFFu)], 1u);
        atomicAdd(&rep[(e1 / GSZ == ga ? sa : sb) * 260 + (xs[q] >> 16)], 1u);
      }
      if (v2 < nv) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint32_t e0 = i1 + 2 * q, e1 = e0 + 1;
          atomicAdd(&rep[(e0 / GSZ == gc ? sc : sd) * 260 + (ys[q] & 0xFFFFu)], 1u);
          atomicAdd(&rep[(e1 / GSZ == gc ? sc : sd) * 260 + (ys[q] >> 16)], 1u);
        }
      }
    }
    if (i_hi >= npos) for (uint32_t i = nv * 8 + threadIdx.x; i < npos; i += 1024) atomicAdd(&rep[sel[i / GSZ] * 260 + A[i]], 1u);
  }
}

// bit accounting of a block whose tables (S.lens) and selectors are final: tile offsets of the symbol data, selector MTF
// positions, table bits, used-map bits, canonical codes, the block's bit length
__device__ void block_accounting(HuffShared& S, HuffBufs& hb, uint32_t blk, const uint8_t* __restrict__ sel, uint8_t* __restrict__ selj,
                                 const uint16_t* __restrict__ bcost, const uint8_t* __restrict__ alist_all, uint32_t nsel, uint32_t asz, int n, int ng) {
  // ---- bit accounting
  // data bits, and the bit offset of every 80-group tile inside the data (the tiles are packed in parallel)
  uint32_t data_bits = 0;
  {
    const uint32_t ntile = (nsel + PD_GROUPS - 1) / PD_GROUPS;                     // <= 226
    uint32_t tsum = 0;
    if (threadIdx.x < ntile) {
      const uint32_t g1 = (threadIdx.x + 1) * PD_GROUPS < nsel ? (threadIdx.x + 1) * PD_GROUPS : nsel;
      for (uint32_t g = threadIdx.x * PD_GROUPS; g < g1; g++) tsum += bcost[g];
    }
    const uint32_t tex = block_excl_sum<1024>(tsum, S.sm, data_bits);
    if (threadIdx.x < ntile) hb.tileoff[(size_t)blk * hb.tile_stride + threadIdx.x] = tex;
  }
  // selector MTF positions (Bzip2:2170-2182): j = #values more recent than the previous occurrence
  uint32_t sel_bits = 0;
  {
    uint32_t lastc[6] = {0, 0, 0, 0, 0, 0};                   // (last index+1) of each value in earlier tiles
    for (uint32_t base = 0; base < nsel; base += 1024) {
      const uint32_t g = base + threadIdx.x;
      const uint32_t sv = g < nsel ? sel[g] : 255u;
      uint32_t lastv[6];
#pragma unroll
      for (int v = 0; v < 6; v++) {
        const uint32_t mine = sv == (uint32_t)v ? g + 1 : 0u;
        const uint32_t im = block_incl_max<1024>(mine, S.sm);
        S.chist[threadIdx.x] = im;
        __syncthreads();
        const uint32_t ex = threadIdx.x ? S.chist[threadIdx.x - 1] : 0u;
        const uint32_t tmax = S.chist[1023];
        __syncthreads();
        lastv[v] = ex > lastc[v] ? ex : lastc[v];
        lastc[v] = tmax > lastc[v] ? tmax : lastc[v];
      }
      if (g < nsel) {
        uint32_t j = 0, mylast = 0;
#pragma unroll
        for (int v = 0; v < 6; v++) if ((uint32_t)v == sv) mylast = lastv[v];
#pragma unroll
        for (int v = 0; v < 6; v++) {
          if (v >= ng || (uint32_t)v == sv) continue;
          if (mylast) j += lastv[v] > mylast;                 // seen before: values touched since then
          else j += (lastv[v] != 0) || ((uint32_t)v < sv);    // first use: seen values + unseen smaller ones
        }
        selj[g] = (uint8_t)j;
        sel_bits += j + 1;
      }
    }
  }
  sel_bits = block_sum<1024>(sel_bits, S.sm);
  uint32_t tab_bits = 0;
  for (int i = threadIdx.x; i < ng * n; i += 1024) {                                // Bzip2:1926-1947
    const int t = i / n, s = i - t * n;
    const int cur = S.lens[t][s], prev = s ? S.lens[t][s - 1] : cur;
    tab_bits += 2u * (uint32_t)(cur > prev ? cur - prev : prev - cur) + 1u + (s == 0 ? 5u : 0u);
  }
  tab_bits = block_sum<1024>(tab_bits, S.sm);
  // used map: 16 + 16 per non-empty range (Bzip2:2071-2080)
  if (threadIdx.x < 8) S.counts[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x < asz) atomicOr(&S.counts[0], 1u << (alist_all[(size_t)blk * 256 + threadIdx.x] >> 4));
  __syncthreads();
  const uint32_t nranges = (uint32_t)__builtin_popcount(S.counts[0]);
  // canonical codes (Bzip2:1896-1916): code = first_code[len] + #{smaller symbols with the same len}
  uint8_t* glens = hb.lens + (size_t)blk * 6 * MAXSYM;
  uint32_t* gcodes = hb.codes + (size_t)blk * 6 * MAXSYM;
  for (int i = threadIdx.x; i < 6 * 32; i += 1024) (&S.key[0][0])[i] = 0;           // reuse key[] as len histograms [6][32]
  __syncthreads();
  uint32_t* lhist = &S.key[0][0];
  for (int i = threadIdx.x; i < ng * n; i += 1024) { const int t = i / n, s = i - t * n; atomicAdd(&lhist[t * 32 + S.lens[t][s]], 1u); }
  __syncthreads();
  if (threadIdx.x < (uint32_t)ng) {
    uint32_t* fc = (uint32_t*)&S.work[0][0] + threadIdx.x * 32;                     // first_code per length
    uint32_t code = 0;
    for (int l = 1; l <= MAX_BITS; l++) { code <<= 1; fc[l] = code; code += lhist[threadIdx.x * 32 + l]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ng * n; i += 1024) {
    const int t = i / n, s = i - t * n;
    const int l = S.lens[t][s];
    uint32_t r = 0;
    for (int k = 0; k < s; k++) r += S.lens[t][k] == l;
    gcodes[t * MAXSYM + s] = ((const uint32_t*)&S.work[0][0])[t * 32 + l] + r;
    glens[t * MAXSYM + s] = (uint8_t)l;
  }
  if (threadIdx.x == 0) {
    hb.ngroups[blk] = (uint32_t)ng;
    hb.bitlen[blk] = 80u + 25u + 16u + 16u * nranges + 18u + sel_bits + tab_bits + data_bits;
    hb.databits[blk] = data_bits;
  }
}

__global__ __launch_bounds__(1024) void huff_block(HuffBufs hb, const uint16_t* __restrict__ Aall, size_t a_stride,
                                                   const uint32_t* __restrict__ npos_all, const uint32_t* __restrict__ asz_all,
                                                   const uint32_t* __restrict__ freq_all, const uint8_t* __restrict__ alist_all, int dbg) {
  __shared__ HuffShared S;
  __shared__ uint32_t stage[AS_GROUPS * 25];
  const uint32_t blk = blockIdx.x;
  const uint32_t npos = npos_all[blk], asz = asz_all[blk];
  const int n = (int)asz + 2;
  const uint16_t* A = Aall + (size_t)blk * a_stride;
  uint8_t* sel = hb.sel + (size_t)blk * hb.sel_stride;
  uint8_t* selj = hb.selj + (size_t)blk * hb.sel_stride;
  uint16_t* bcost = hb.bcost + (size_t)blk * hb.sel_stride;
  const uint32_t nsel = (npos + GSZ - 1) / GSZ;
  const int target = npos >= 2400 ? 6 : npos >= 1200 ? 5 : npos >= 600 ? 4 : npos >= 200 ? 3 : 2;   // Bzip2:2150
  const int w = wave_id();

  HB_MARK(0);
  // initial tables: global frequencies and flat (Bzip2:2155-2157)
  for (int i = threadIdx.x; i < n; i += 1024) { S.freq[0][i] = freq_all[(size_t)blk * 258 + i]; S.freq[1][i] = 1; }
  __syncthreads();
  if (w < 2) build_table_wave(S.freq[w], S.lens[w], S.key[w], S.work[w], n);
  __syncthreads();
  int ng = 2;
  HB_MARK(1);
  while (ng < target) {                                                             // Bzip2:2012-2053
    assign_selectors(S, stage, A, npos, nsel, ng, sel, bcost);
    if (ng == 2) HB_MARK(2);
    median_split(S, sel, bcost, nsel, ng);
    if (ng == 2) HB_MARK(3);
    ng++;
    count_symbols(stage, A, sel, npos, 0u, npos);
    __syncthreads();
    for (int i = threadIdx.x; i < 6 * 260; i += 1024) {
      uint32_t f = 0;
#pragma unroll
      for (int r = 0; r < 8; r++) f += stage[r * (6 * 260) + i];
      (&S.freq[0][0])[i] = f;
    }
    __syncthreads();
    if (ng == 3) HB_MARK(4);
    if (w < ng) build_table_wave(S.freq[w], S.lens[w], S.key[w], S.work[w], n);
    __syncthreads();
    if (ng == 3) HB_MARK(5);
  }
  HB_MARK(6);
  assign_selectors(S, stage, A, npos, nsel, ng, sel, bcost);                        // Bzip2:2163
  __syncthreads();

  HB_MARK(7);
  block_accounting(S, hb, blk, sel, selj, bcost, alist_all, nsel, asz, n, ng);
  HB_MARK(8);
  HB_MARK(9);
}

// ------------------------------------------------------------------ the same refinement as a chain of kernels
// huff_block keeps one CU per block busy for ~0.9 ms while the other CUs idle (a -9 call of 100 MB has 112 blocks).  Its
// per-group and per-symbol phases (selector assignment, symbol counts) are data-parallel, so here they are kernels of
// their own with many workgroups per block; the per-block phases (median split, table builds, accounting) stay one
// workgroup per block.  State between the kernels: code lengths wl[blk][6][264], counts wfreq[blk][6][260].
// ng = tables in use when the kernel runs; a block takes part while ng <= / < its target (Bzip2:2150).
__device__ __forceinline__ int huff_target(uint32_t npos) { return npos >= 2400 ? 6 : npos >= 1200 ? 5 : npos >= 600 ? 4 : npos >= 200 ? 3 : 2; }
__device__ __forceinline__ void load_lens(HuffShared& S, const uint8_t* __restrict__ wl, int ng) {
  const uint32_t* src = reinterpret_cast<const uint32_t*>(wl);
  uint32_t* dst = reinterpret_cast<uint32_t*>(&S.lens[0][0]);
  for (int i = threadIdx.x; i < ng * 66; i += blockDim.x) dst[i] = src[i];
  __syncthreads();
}
__global__ __launch_bounds__(128) void hs_init(HuffBufs hb, const uint32_t* __restrict__ asz_all, const uint32_t* __restrict__ freq_all) {
  __shared__ uint32_t freq[2][260], key[2][260];
  __shared__ int work[2][260];
  __shared__ uint8_t lens[2][264];
  const uint32_t blk = blockIdx.x;
  const int n = (int)asz_all[blk] + 2, w = wave_id();
  for (int i = threadIdx.x; i < n; i += 128) { freq[0][i] = freq_all[(size_t)blk * 258 + i]; freq[1][i] = 1; }
  __syncthreads();
  build_table_wave(freq[w], lens[w], key[w], work[w], n);
  __syncthreads();
  uint8_t* wl = hb.wl + (size_t)blk * 6 * 264;
  for (int i = threadIdx.x; i < 2 * 264; i += 128) wl[i] = (&lens[0][0])[i];
}
constexpr uint32_t HS_STEPS = 4;       // 512-group steps per workgroup of hs_assign
__global__ __launch_bounds__(1024) void hs_assign(HuffBufs hb, const uint16_t* __restrict__ Aall, size_t a_stride, const uint32_t* __restrict__ npos_all, int ng) {
  __shared__ HuffShared S;
  __shared__ uint32_t stage[AS_GROUPS * 25];
  const uint32_t blk = blockIdx.y, npos = npos_all[blk], nsel = (npos + GSZ - 1) / GSZ;
  const uint32_t g_lo = blockIdx.x * (HS_STEPS * AS_GROUPS);
  if (ng > huff_target(npos) || g_lo >= nsel) return;
  load_lens(S, hb.wl + (size_t)blk * 6 * 264, ng);
  assign_selectors(S, stage, Aall + (size_t)blk * a_stride, npos, nsel, ng, hb.sel + (size_t)blk * hb.sel_stride, hb.bcost + (size_t)blk * hb.sel_stride,
                   g_lo, g_lo + HS_STEPS * AS_GROUPS);
}
__global__ __launch_bounds__(1024) void hs_split(HuffBufs hb, const uint32_t* __restrict__ npos_all, int ng) {
  __shared__ HuffShared S;
  const uint32_t blk = blockIdx.x, npos = npos_all[blk], nsel = (npos + GSZ - 1) / GSZ;
  if (ng >= huff_target(npos)) return;
  median_split(S, hb.sel + (size_t)blk * hb.sel_stride, hb.bcost + (size_t)blk * hb.sel_stride, nsel, ng);
  uint32_t* wf = hb.wfreq + (size_t)blk * 6 * 260;
  for (int i = threadIdx.x; i < 6 * 260; i += 1024) wf[i] = 0;
}
constexpr uint32_t HS_CNT = 131072;     // symbols per workgroup of hs_count
__global__ __launch_bounds__(1024) void hs_count(HuffBufs hb, const uint16_t* __restrict__ Aall, size_t a_stride, const uint32_t* __restrict__ npos_all, int ng) {
  __shared__ uint32_t stage[8 * 6 * 260];
  const uint32_t blk = blockIdx.y, npos = npos_all[blk];
  const uint32_t i_lo = blockIdx.x * HS_CNT;
  if (ng >= huff_target(npos) || i_lo >= npos) return;
  count_symbols(stage, Aall + (size_t)blk * a_stride, hb.sel + (size_t)blk * hb.sel_stride, npos, i_lo, i_lo + HS_CNT);
  __syncthreads();
  uint32_t* wf = hb.wfreq + (size_t)blk * 6 * 260;
  for (int i = threadIdx.x; i < 6 * 260; i += 1024) {
    uint32_t f = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) f += stage[r * (6 * 260) + i];
    if (f) atomicAdd(&wf[i], f);
  }
}
__global__ __launch_bounds__(384) void hs_build(HuffBufs hb, const uint32_t* __restrict__ npos_all, const uint32_t* __restrict__ asz_all, int ng /* before the new table */) {
  __shared__ uint32_t freq[6][260], key[6][260];
  __shared__ int work[6][260];
  __shared__ uint8_t lens[6][264];
  const uint32_t blk = blockIdx.x, npos = npos_all[blk];
  if (ng >= huff_target(npos)) return;
  const int n = (int)asz_all[blk] + 2, w = wave_id();
  const uint32_t* wf = hb.wfreq + (size_t)blk * 6 * 260;
  for (int i = threadIdx.x; i < 6 * 260; i += 384) (&freq[0][0])[i] = wf[i];
  __syncthreads();
  if (w <= ng) build_table_wave(freq[w], lens[w], key[w], work[w], n);
  __syncthreads();
  uint8_t* wl = hb.wl + (size_t)blk * 6 * 264;
  for (int i = threadIdx.x; i < (ng + 1) * 264; i += 384) wl[i] = (&lens[0][0])[i];
}
__global__ __launch_bounds__(1024) void hs_finish(HuffBufs hb, const uint32_t* __restrict__ npos_all, const uint32_t* __restrict__ asz_all,
                                                  const uint8_t* __restrict__ alist_all) {
  __shared__ HuffShared S;
  const uint32_t blk = blockIdx.x, npos = npos_all[blk], asz = asz_all[blk], nsel = (npos + GSZ - 1) / GSZ;
  const int ng = huff_target(npos);
  load_lens(S, hb.wl + (size_t)blk * 6 * 264, ng);
  block_accounting(S, hb, blk, hb.sel + (size_t)blk * hb.sel_stride, hb.selj + (size_t)blk * hb.sel_stride, hb.bcost + (size_t)blk * hb.sel_stride,
                   alist_all, nsel, asz, (int)asz + 2, ng);
}

// ------------------------------------------------------------------ bit offsets + stream CRC (one lane, tiny)
// also the output size check (no host round trip for it): too_small[0] = 1 stops the pack kernels before they write anything
__global__ void huff_offsets(HuffBufs hb, const uint32_t* __restrict__ block_crc, uint32_t nb, uint32_t first, uint32_t count,
                             uint64_t start_bit, uint32_t* __restrict__ stream_crc_out, uint64_t trailer_bits, uint64_t cap_bytes,
                             uint64_t* __restrict__ too_small, int crc_given, uint32_t crc_value) {
  if (threadIdx.x || blockIdx.x) return;
  (void)first;
  uint64_t bit = start_bit;                         // per-block buffers are indexed relative to `first`
  for (uint32_t k = 0; k < count; k++) { hb.bitoff[k] = bit; bit += hb.bitlen[k]; }
  hb.bitoff[count] = bit;
  too_small[0] = (bit + trailer_bits + 7) / 8 + 8 > cap_bytes ? 1u : 0u;
  uint32_t c = crc_value;                           // (a rank of a multi-GPU job is handed the fold over all ranks' blocks)
  if (!crc_given) { c = 0; for (uint32_t k = 0; k < nb; k++) c = ((c << 1) | (c >> 31)) ^ block_crc[k]; }      // Bzip2:2237
  *stream_crc_out = c;
}

// zeroes the part of the output the stream will occupy (the pack kernels OR their boundary words into it): the stream end is
// known on the device (bitoff[count]); the rest of the caller's buffer is left as it is
__global__ __launch_bounds__(256) void pack_zero_output(HuffBufs hb, uint32_t count, uint64_t trailer_bits, uint32_t* __restrict__ out32, uint64_t cap_bytes,
                                                        const uint64_t* __restrict__ too_small) {
  if (too_small[0]) return;
  uint64_t bytes = (hb.bitoff[count] + trailer_bits + 7) / 8 + 16;
  if (bytes > cap_bytes) bytes = cap_bytes;
  const uint64_t nw = bytes / 4;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nw; i += (uint64_t)gridDim.x * 256) out32[i] = 0;
}

// ------------------------------------------------------------------ bit packing
// Put `nbits` (<= 64) of `val` at absolute stream bit `bit` into LDS words indexed from word0.
__device__ __forceinline__ void put_bits(uint32_t* words, uint64_t word0, uint64_t bit, uint64_t val, uint32_t nbits) {
  if (!nbits) return;
  uint64_t w = (bit >> 5) - word0;
  uint32_t o = (uint32_t)(bit & 31);
  uint32_t left = nbits;
  while (left) {
    const uint32_t room = 32 - o, take = left < room ? left : room;
    const uint32_t chunk = (uint32_t)((val >> (left - take)) & (take == 32 ? 0xFFFFFFFFull : ((1ull << take) - 1ull)));
    atomicOr(&words[w], chunk << (room - take));
    left -= take; o = 0; w++;
  }
}

constexpr int PK_ITEMS = 4;                       // items per thread per tile
constexpr int PK_WORDS = (1024 * PK_ITEMS * 40) / 32 + 8;

struct Item { uint64_t val; uint32_t nbits; };

template <typename F>
__device__ void pack_phase(uint32_t count, F item_fn, uint32_t* words, uint32_t* sm, uint64_t& bit, uint32_t* __restrict__ out32) {
  for (uint32_t base = 0; base < count; base += 1024 * PK_ITEMS) {
    Item it[PK_ITEMS];
    uint32_t nb = 0;
#pragma unroll
    for (int j = 0; j < PK_ITEMS; j++) {
      const uint32_t i = base + threadIdx.x * PK_ITEMS + j;
      it[j].val = 0; it[j].nbits = 0;
      if (i < count) it[j] = item_fn(i);
      nb += it[j].nbits;
    }
    uint32_t tot;
    const uint32_t ex = block_excl_sum<1024>(nb, sm, tot);
    const uint64_t word0 = bit >> 5;
    const uint32_t nwords = (uint32_t)(((bit + tot + 31) >> 5) - word0);
    for (uint32_t i = threadIdx.x; i < nwords; i += 1024) words[i] = 0;
    __syncthreads();
    uint64_t b = bit + ex;
    {                                                   // the thread's items as one or two strings of <= 64 bits: fewer LDS atomics on shared words
      uint64_t pv = 0; uint32_t pn = 0;
#pragma unroll
      for (int j = 0; j < PK_ITEMS; j++) {
        if (pn + it[j].nbits > 64u) { put_bits(words, word0, b, pv, pn); b += pn; pv = 0; pn = 0; }
        pv = it[j].nbits >= 64u ? it[j].val : ((pv << it[j].nbits) | (it[j].val & ((1ull << it[j].nbits) - 1ull)));
        pn += it[j].nbits;
      }
      put_bits(words, word0, b, pv, pn);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nwords; i += 1024) {
      const uint32_t v = __builtin_bswap32(words[i]);
      if (i == 0 || i == nwords - 1) { if (v) atomicOr(&out32[word0 + i], v); }
      else out32[word0 + i] = v;
    }
    __syncthreads();
    bit += tot;
  }
}

__global__ __launch_bounds__(1024) void pack_block(HuffBufs hb, uint32_t first, const uint16_t* __restrict__ Aall, size_t a_stride,
                                                   const uint32_t* __restrict__ npos_all, const uint32_t* __restrict__ asz_all,
                                                   const uint8_t* __restrict__ alist_all, const uint32_t* __restrict__ block_crc,
                                                   const uint32_t* __restrict__ pidx_all, uint32_t* __restrict__ out32,
                                                   const uint64_t* __restrict__ too_small) {
  __shared__ uint32_t words[PK_WORDS];
  __shared__ uint32_t sm[16];
  __shared__ uint32_t used16[17];
  if (too_small[0]) return;
  __shared__ uint32_t ctab[6 * MAXSYM];
  __shared__ uint8_t ltab[6 * MAXSYM + 4];
  const uint32_t blk = blockIdx.x;                  // relative to `first`; block_crc is absolute
  const uint32_t npos = npos_all[blk], asz = asz_all[blk], ng = hb.ngroups[blk];
  const uint32_t n = asz + 2, nsel = (npos + GSZ - 1) / GSZ;
  (void)Aall; (void)a_stride;
  const uint8_t* selj = hb.selj + (size_t)blk * hb.sel_stride;
  for (uint32_t i = threadIdx.x; i < 6 * MAXSYM; i += 1024) { ctab[i] = hb.codes[(size_t)blk * 6 * MAXSYM + i]; ltab[i] = hb.lens[(size_t)blk * 6 * MAXSYM + i]; }
  if (threadIdx.x < 17) used16[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x < asz) {
    const uint32_t v = alist_all[(size_t)blk * 256 + threadIdx.x];
    atomicOr(&used16[1 + (v >> 4)], 0x8000u >> (v & 15));
    atomicOr(&used16[0], 0x8000u >> (v >> 4));
  }
  __syncthreads();
  uint64_t bit = hb.bitoff[blk];
  const uint32_t crc = block_crc[first + blk], pidx = pidx_all[blk];
  // header: magic(48) crc(32) rand(1)+pidx(24) coarse(16) fine(16 each) ngroups(3)+nsel(15)
  const uint32_t coarse = used16[0];
  pack_phase(22, [&](uint32_t i) -> Item {
    if (i == 0) return Item{0x314159265359ull, 48};
    if (i == 1) return Item{crc, 32};
    if (i == 2) return Item{pidx & 0xFFFFFFu, 25};
    if (i == 3) return Item{coarse, 16};
    if (i < 20) { const uint32_t r = i - 4; return (coarse & (0x8000u >> r)) ? Item{used16[1 + r], 16} : Item{0, 0}; }
    if (i == 20) return Item{((uint64_t)ng << 15) | nsel, 18};
    return Item{0, 0};
  }, words, sm, bit, out32);
  // selectors, MTF + unary (Bzip2:2171-2182)
  pack_phase(nsel, [&](uint32_t g) -> Item { const uint32_t j = selj[g]; return Item{((1ull << j) - 1ull) << 1, j + 1}; }, words, sm, bit, out32);
  // tables (Bzip2:1926-1947)
  pack_phase(ng * n, [&](uint32_t i) -> Item {
    const uint32_t t = i / n, s = i - t * n;
    const uint32_t cur = ltab[t * MAXSYM + s], prev = s ? ltab[t * MAXSYM + s - 1] : cur;
    const uint32_t d = cur > prev ? cur - prev : prev - cur;
    uint64_t v = 0;
    const uint64_t pat = cur > prev ? 2ull : 3ull;
    for (uint32_t k = 0; k < d; k++) v = (v << 2) | pat;
    v <<= 1;
    uint32_t nbits = 2 * d + 1;
    if (s == 0) { v |= (uint64_t)cur << 1; nbits = 6; }      // 5-bit start length, then '0'
    return Item{v, nbits};
  }, words, sm, bit, out32);
  // the symbol data follows: pack_data, one workgroup per 4000-symbol tile
}

// data (Bzip2:2189-2194): tile `blockIdx.x` of block `blockIdx.y`; its first bit = block start + block bits - data bits +
// the tile's offset from huff_block.  Interior words are plain stores, the two boundary words are OR-ed.
__global__ __launch_bounds__(1024) void pack_data(HuffBufs hb, const uint16_t* __restrict__ Aall, size_t a_stride, const uint32_t* __restrict__ npos_all,
                                                  uint32_t* __restrict__ out32, const uint64_t* __restrict__ too_small) {
  __shared__ uint32_t words[PK_WORDS];
  __shared__ uint32_t sm[16];
  __shared__ uint32_t ctab[6 * MAXSYM];
  __shared__ uint8_t ltab[6 * MAXSYM + 4];
  if (too_small[0]) return;
  const uint32_t blk = blockIdx.y, tile = blockIdx.x;
  const uint32_t npos = npos_all[blk], nsel = (npos + GSZ - 1) / GSZ;
  if (tile * PD_GROUPS >= nsel) return;
  const uint16_t* A = Aall + (size_t)blk * a_stride;
  const uint8_t* sel = hb.sel + (size_t)blk * hb.sel_stride;
  for (uint32_t i = threadIdx.x; i < 6 * MAXSYM; i += 1024) { ctab[i] = hb.codes[(size_t)blk * 6 * MAXSYM + i]; ltab[i] = hb.lens[(size_t)blk * 6 * MAXSYM + i]; }
  __syncthreads();
  uint64_t bit = hb.bitoff[blk] + hb.bitlen[blk] - hb.databits[blk] + hb.tileoff[(size_t)blk * hb.tile_stride + tile];
  const uint32_t i0 = tile * PD_GROUPS * GSZ;
  const uint32_t cnt = npos - i0 < (uint32_t)(PD_GROUPS * GSZ) ? npos - i0 : (uint32_t)(PD_GROUPS * GSZ);
  pack_phase(cnt, [&](uint32_t i) -> Item {
    const uint32_t t = sel[(i0 + i) / GSZ], s = A[i0 + i];
    return Item{ctab[t * MAXSYM + s], ltab[t * MAXSYM + s]};
  }, words, sm, bit, out32);
}

// stream header / trailer.  One lane.
// follow_magic: the blocks of the next rank of a multi-GPU job follow this fragment: the last word of the fragment is completed
// with the leading bits of what comes next in the stream -- always the 48-bit block magic -- so that the ranks' fragments are
// disjoint runs of whole 32-bit words of the one stream (the next rank leaves its first, partial word out of its fragment).
__global__ void pack_frame(HuffBufs hb, uint32_t nb_range_end, int level, int write_header, int write_trailer, int follow_magic,
                           const uint32_t* __restrict__ stream_crc, uint32_t* __restrict__ out32, uint64_t* __restrict__ total_bits) {
  if (threadIdx.x || blockIdx.x) return;
  if (total_bits[2]) { total_bits[0] = hb.bitoff[nb_range_end] + (write_trailer ? 80u : 0u); return; }      // output too small: nothing is written
  if (write_header) atomicOr(&out32[0], __builtin_bswap32(0x425a6830u + (uint32_t)level));   // 'B''Z''h''0'+level
  uint64_t bit = hb.bitoff[nb_range_end];
  if (follow_magic && !write_trailer && (bit & 31)) {
    const uint32_t room = 32 - (uint32_t)(bit & 31);
    atomicOr(&out32[bit >> 5], __builtin_bswap32((uint32_t)(0x314159265359ull >> (48 - room))));
  }
  if (write_trailer) {
    const uint64_t vals[2] = {0x177245385090ull, (uint64_t)*stream_crc};
    const uint32_t nbs[2] = {48, 32};
    for (int q = 0; q < 2; q++) {
      uint32_t left = nbs[q];
      while (left) {
        const uint32_t o = (uint32_t)(bit & 31), room = 32 - o, take = left < room ? left : room;
        const uint32_t chunk = (uint32_t)((vals[q] >> (left - take)) & (take == 32 ? 0xFFFFFFFFull : ((1ull << take) - 1ull)));
        atomicOr(&out32[bit >> 5], __builtin_bswap32(chunk << (room - take)));
        left -= take; bit += take;
      }
    }
  }
  *total_bits = bit;
}

// ------------------------------------------------------------------------------------------
size_t HuffWork::bytes_needed(size_t max_blocks, uint32_t stride) {
  size_t b = 0;
  auto add = [&](size_t n) { b += (n + 255) & ~(size_t)255; };
  const size_t ss = sel_stride_for(stride);
  add(max_blocks * ss); add(max_blocks * ss); add(max_blocks * ss * 2);
  add(max_blocks * 6 * MAXSYM); add(max_blocks * 6 * MAXSYM * 4);
  add(max_blocks * 4); add(max_blocks * 4); add((max_blocks + 1) * 8); add(64);
  add(max_blocks * 4); add(max_blocks * (ss / PD_GROUPS + 2) * 4);
  add(max_blocks * 6 * 264); add(max_blocks * 6 * 260 * 4);
  return b + 4096;
}
int HuffWork::carve(Arena& a, size_t max_blocks_, uint32_t stride) {
  max_blocks = max_blocks_;
  b.sel_stride = sel_stride_for(stride);
  b.sel = a.take<uint8_t>(max_blocks * b.sel_stride); b.selj = a.take<uint8_t>(max_blocks * b.sel_stride);
  b.bcost = a.take<uint16_t>(max_blocks * b.sel_stride);
  b.lens = a.take<uint8_t>(max_blocks * 6 * MAXSYM); b.codes = a.take<uint32_t>(max_blocks * 6 * MAXSYM);
  b.ngroups = a.take<uint32_t>(max_blocks); b.bitlen = a.take<uint32_t>(max_blocks); b.bitoff = a.take<uint64_t>(max_blocks + 1);
  scalars = a.take<uint64_t>(8);
  b.tile_stride = b.sel_stride / PD_GROUPS + 2;
  b.databits = a.take<uint32_t>(max_blocks); b.tileoff = a.take<uint32_t>(max_blocks * b.tile_stride);
  b.wl = a.take<uint8_t>(max_blocks * 6 * 264); b.wfreq = a.take<uint32_t>(max_blocks * 6 * 260);
  max_stride = stride;
  return (scalars && b.tileoff && b.wl && b.wfreq) ? 0 : CJS_E_OUT_OF_MEMORY;
}

int huff_tables_run(hipStream_t s, HuffWork& w, uint32_t nb, const uint16_t* d_A, size_t a_stride, const uint32_t* d_npos,
                    const uint32_t* d_asz, const uint32_t* d_freq, const uint8_t* d_alist) {
  if (nb == 0) return 0;
  static const bool dbg = getenv("CJS_DEBUG") != nullptr;
  // one workgroup per block (huff_block) keeps nb CUs busy; with fewer blocks than CUs the chain of kernels spreads the
  // data-parallel phases over the whole chip
  const bool split = nb >= 8 && nb <= 512 && (size_t)nb * w.max_stride >= ((size_t)8 << 20);
  if (split) {
    const uint32_t max_sel = (uint32_t)(((size_t)w.max_stride + 1 + GSZ - 1) / GSZ);
    const uint32_t ga = (max_sel + HS_STEPS * AS_GROUPS - 1) / (HS_STEPS * AS_GROUPS), gc = (w.max_stride + 1 + HS_CNT - 1) / HS_CNT;
    hipLaunchKernelGGL(hs_init, dim3(nb), dim3(128), 0, s, w.b, d_asz, d_freq);
    for (int ng = 2; ng <= 6; ng++) {
      hipLaunchKernelGGL(hs_assign, dim3(ga, nb), dim3(1024), 0, s, w.b, d_A, a_stride, d_npos, ng);
      if (ng == 6) break;
      hipLaunchKernelGGL(hs_split, dim3(nb), dim3(1024), 0, s, w.b, d_npos, ng);
      hipLaunchKernelGGL(hs_count, dim3(gc, nb), dim3(1024), 0, s, w.b, d_A, a_stride, d_npos, ng);
      hipLaunchKernelGGL(hs_build, dim3(nb), dim3(384), 0, s, w.b, d_npos, d_asz, ng);
    }
    hipLaunchKernelGGL(hs_finish, dim3(nb), dim3(1024), 0, s, w.b, d_npos, d_asz, d_alist);
    CJS_HIP_TRY(hipGetLastError());
    if (dbg) {
      uint64_t clk[8];
      CJS_HIP_TRY(hipStreamSynchronize(s));
      CJS_HIP_TRY(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_bt_clk), sizeof clk));
      fprintf(stderr, "[cjs huff] last table build of block 0 / wave 0: rank sort %.1f us, allocator pass 1 %.1f, pass 2 %.1f, pass 3 %.1f, lengths back %.1f us\n",
              (double)(clk[1] - clk[0]) / 100.0, (double)(clk[2] - clk[1]) / 100.0, (double)(clk[3] - clk[2]) / 100.0, (double)(clk[5] - clk[3]) / 100.0, (double)(clk[6] - clk[5]) / 100.0);
    }
    return 0;
  }
  hipLaunchKernelGGL(huff_block, dim3(nb), dim3(1024), 0, s, w.b, d_A, a_stride, d_npos, d_asz, d_freq, d_alist, dbg ? 1 : 0);
  CJS_HIP_TRY(hipGetLastError());
  if (dbg) {
    uint64_t clk[32];
    CJS_HIP_TRY(hipStreamSynchronize(s));
    CJS_HIP_TRY(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_huff_clk), sizeof clk));
    static const char* names[] = {"init tables", "assign(ng=2)", "median split(ng=2)", "freq count(ng=2->3)", "build tables(ng=3)",
                                  "rest of refinement", "final assign", "bit accounting, codes"};
    for (int i = 0; i < 8; i++) fprintf(stderr, "[cjs huff] wg0 %-36s %8.1f us\n", names[i], (double)(clk[i + 1] - clk[i]) / 100.0);
  }
  return 0;
}

int huff_pack_run(hipStream_t s, HuffWork& w, uint32_t nb_total, uint32_t first, uint32_t count, uint64_t start_bit, int level,
                  int write_header, int write_trailer, const uint16_t* d_A, size_t a_stride, const uint32_t* d_npos,
                  const uint32_t* d_asz, const uint8_t* d_alist, const uint32_t* d_block_crc, const uint32_t* d_pidx,
                  uint32_t* d_out32, size_t out_cap_bytes, const PackShard* ps) {
  uint32_t* stream_crc = (uint32_t*)(w.scalars + 1);
  hipLaunchKernelGGL(huff_offsets, dim3(1), dim3(1), 0, s, w.b, d_block_crc, nb_total, first, count, start_bit, stream_crc,
                     (uint64_t)(write_trailer ? 80 : 0), (uint64_t)out_cap_bytes, w.scalars + 2, ps ? 1 : 0, ps ? ps->stream_crc : 0u);
  hipLaunchKernelGGL(pack_zero_output, dim3(4096), dim3(256), 0, s, w.b, count, (uint64_t)(write_trailer ? 80 : 0), d_out32, (uint64_t)out_cap_bytes, w.scalars + 2);
  if (count) {
    hipLaunchKernelGGL(pack_block, dim3(count), dim3(1024), 0, s, w.b, first, d_A, a_stride, d_npos, d_asz, d_alist, d_block_crc, d_pidx, d_out32, w.scalars + 2);
    hipLaunchKernelGGL(pack_data, dim3((unsigned)(w.b.tile_stride - 1), count), dim3(1024), 0, s, w.b, d_A, a_stride, d_npos, d_out32, w.scalars + 2);
  }
  hipLaunchKernelGGL(pack_frame, dim3(1), dim3(1), 0, s, w.b, count, level, write_header, write_trailer, ps ? ps->follow_magic : 0, stream_crc, d_out32, w.scalars);
  CJS_HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace cjs
