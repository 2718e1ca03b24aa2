// decode.hip — Bzip2.decompressFile (Bunzip.decode, J/Bzip2_joined_.js:1769-1796) on the MI355X.
//
// The reference decodes block after block from one bit cursor.  Blocks carry no length field, so here:
//   1. bz_magic_scan     every bit offset of the stream is tested against the 48-bit block / end-of-stream
//                        magics (:1434-1439) -> candidate list (a few hundred entries);
//   2. bz_decode_block   one wave per candidate: header, selector list, code-length tables, then the
//                        bit-serial Huffman + RUNA/RUNB + MTF decode (:1456-1670) on lane 0 with
//                        10-bit direct lookup tables built by the whole wave from the reference's
//                        limit/base/permute semantics; yields the BWT bytes, their histogram, the end bit;
//   3. host              walks the chain 32 -> end(block0) -> end(block1) ... over the candidates (a false
//                        2^-48 candidate inside payload bits is simply never reached), folds CRCs;
//   4. inverse BWT       T vector by a stable radix pass keyed (block, byte) (:1677-1690), then the LF walk
//                        (:1732-1737) — n dependent gathers — made k-way parallel by splitter list ranking:
//                        every 128th slot is a splitter, lanes walk to the next splitter, one workgroup per block
//                        ranks the splitters (pointer jumping in LDS), lanes re-walk writing bytes at their final offsets;
//   5. RLE1 expansion    (:1738-1753) parsed in parallel: a count byte follows 4 equal literals; inside a
//                        stretch of equal bytes the literal/count phase has period 5 and the only carried
//                        state (does the stretch start with a count byte?) is a 2-state function scan;
//   6. CRC check         per block over the output bytes (rle1.hip's slice + GF(2) combine), :1756-1761.
#include "cjs_internal.h"
#include "prims.hpp"
#include "rle1.h"
#include <algorithm>
#include <chrono>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>

namespace cjs {
int select_device(const cjs_opts* opts);
template <typename K>
int radix_passes_public(hipStream_t s, BwtWork& w, K* k0, uint32_t* v0, K* k1, uint32_t* v1, int& cur, uint32_t n, int lo_bit, int hi_bit);
template <typename K>
int radix_pass_segments_public(hipStream_t s, BwtWork& w, K* k0, uint32_t* v0, K* k1, uint32_t* v1, int& cur, uint32_t nseg, uint32_t stride, int lo_bit, int hi_bit, bool noval, bool first_hist_ready);
int crc_ranges(hipStream_t s, const uint8_t* d_data, const RleBlock* d_blocks, const uint32_t* d_nblocks, uint32_t count, uint32_t max_segs,
               uint32_t* d_seg_crc, uint32_t* d_crc_out);
}
using namespace cjs;

namespace cjs {

constexpr uint64_t MAGIC_BLOCK = 0x314159265359ull, MAGIC_END = 0x177245385090ull;
// One splitter every SPL slots of the LF vector.  A walk ends where it lands on a splitter slot -- a 1-in-SPL chance per step --,
// so the segments are geometric and the walk kernels last as long as the LONGEST of a block's segments, ~SPL x ln(n / SPL) steps
// of dependent loads (1,100 at 128, 610 at 64): halving SPL halves them; the splitter chain of a block (14,064 nodes at level 9)
// still fits the ranking kernel's LDS.
constexpr int SPL = 64;

struct Cand { uint64_t bit; uint32_t kind; uint32_t pad; };      // kind 0 = block, 1 = end of stream; pad = row of the block candidate in the decode buffer (set by the host)
struct BlockOut {
  uint64_t end_bit;       // first bit after the block's EOB code
  uint32_t count;         // decoded BWT bytes (dbufCount)
  uint32_t orig;          // origPointer
  uint32_t crc;           // stored block CRC
  int32_t err;            // 0 or a CJS_E_* code
};

// ---------------------------------------------------------------- 1. magic scan
// `in` is addressed by absolute stream byte; bytes [byte0, byte1) are tested as candidate starts, reads stay below n
__global__ __launch_bounds__(256) void bz_magic_scan(const uint8_t* __restrict__ in, uint64_t byte0, uint64_t byte1, uint64_t n, Cand* __restrict__ out, uint32_t cap,
                                                     uint32_t* __restrict__ count) {
  const uint64_t byte = byte0 + (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (byte >= byte1 || byte + 6 > n) return;
  uint64_t w = 0;
  for (int i = 0; i < 7; i++) w = (w << 8) | (byte + i < n ? in[byte + i] : 0);      // 56 bits
  for (int b = 0; b < 8; b++) {
    if (byte * 8 + b + 48 > n * 8) break;
    const uint64_t v = (w >> (8 - b)) & 0xFFFFFFFFFFFFull;
    if (v == MAGIC_BLOCK || v == MAGIC_END) {
      const uint32_t idx = atomicAdd(count, 1u);
      if (idx < cap) { out[idx].bit = byte * 8 + b; out[idx].kind = v == MAGIC_END; out[idx].pad = 0; }
    }
  }
}

// ---------------------------------------------------------------- 2. block decode (one wave per candidate)
struct BitReader {
  const uint8_t* p; uint64_t nbits, pos;
  uint64_t win; uint64_t wbyte;                                      // cached big-endian window of bytes [wbyte, wbyte+8)
  __device__ __forceinline__ void refill() {
    wbyte = pos >> 3;
    const uint64_t nbytes = (nbits + 7) >> 3;
    uint64_t w = 0;
    if (wbyte + 8 <= nbytes) {
#pragma unroll
      for (int i = 0; i < 8; i++) w = (w << 8) | p[wbyte + i];
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) w = (w << 8) | (wbyte + i < nbytes ? p[wbyte + i] : 0);   // zeros past EOF (:149)
    }
    win = w;
  }
  __device__ __forceinline__ uint32_t peek(int k) {                  // next k <= 25 bits
    uint64_t off = pos - (wbyte << 3);
    if (pos < (wbyte << 3) || off + (uint64_t)k > 64) { refill(); off = pos - (wbyte << 3); }
    return (uint32_t)((win << off) >> (64 - k));
  }
  __device__ __forceinline__ void skip(int k) { pos += k; if (pos > nbits) pos = nbits; }
  __device__ __forceinline__ uint32_t get(int k) { const uint32_t v = peek(k); skip(k); return v; }
};

// Canonical code of one table in first-code form: the codes of length L are first[L] .. first[L] + cnt[L] - 1, handed out in
// symbol order (bysym[start[L] ..]), and first[L + 1] = (first[L] + cnt[L]) << 1.  A prefix j of L bits that is not a code of a
// shorter length satisfies j >= first[L], so "j - first[L] < cnt[L]" decides -- the same decisions as the reference's
// limit / base / permute walk (:1522-1581, :1605-1616) on every length table, complete or not.
struct DecShared {
  uint32_t first[6][22];           // first code of each length
  uint16_t cnt[6][22];             // symbols of each length
  uint16_t start[6][22];           // index of the first symbol of each length in bysym
  uint16_t bysym[6][260];          // symbols ordered by (length, symbol)
  uint8_t minlen[8], maxlen[8];
  uint8_t length[6][260];
  uint8_t sym_to_byte[256];
  // left-justified (20-bit) end of the codes of each length, for the branch-free length rule of bz_chain: the code that starts
  // with the 20 bits x has length 1 + #{L in 1..19 : x >= limp[L]} and exists iff x < limp[20]  (limp[L] = 0 below the shortest
  // length, = (first[L] + cnt[L]) << (20 - L) from there to the longest, then 2^20 up to L = 19; limp[20] = the longest's)
  uint32_t limp[6][21];
};

// big-endian 32-bit word `dw` of the stream, zeros past the end (the reference's reader yields zero bits there, :149)
__device__ __forceinline__ uint32_t load_be32(const uint8_t* in, uint64_t n, uint64_t dw) {
  const uint64_t b = dw * 4;
  if (b + 4 <= n) return __builtin_bswap32(*reinterpret_cast<const uint32_t*>(in + b));       // `in` is a hipMalloc'd copy: 4-byte aligned
  uint32_t v = 0;
  for (int i = 0; i < 4; i++) v = (v << 8) | (b + i < n ? in[b + i] : 0u);
  return v;
}
// 4096-bit register window on the stream: lane j holds words base+j (A) and base+64+j (B); all cursor state is wave-uniform
struct BitWin {
  const uint8_t* in; uint64_t n; uint64_t base; uint32_t A, B;
  __device__ __forceinline__ void init(uint64_t pos, int lane) { base = pos >> 5; A = load_be32(in, n, base + lane); B = load_be32(in, n, base + 64 + lane); }
  __device__ __forceinline__ void ensure(uint64_t pos, int lane) {        // afterwards (pos >> 5) - base < 64
    while ((pos >> 5) - base >= 64) {
      if ((pos >> 5) - base >= 128) { init(pos, lane); return; }
      A = B; base += 64; B = load_be32(in, n, base + 64 + lane);
    }
  }
  __device__ __forceinline__ uint32_t word(uint32_t k) const { return k < 64 ? __builtin_amdgcn_readlane(A, k) : __builtin_amdgcn_readlane(B, k - 64); }
  __device__ __forceinline__ uint32_t peek(uint64_t pos, int k) const {   // k <= 32 bits at pos (uniform), window must cover it
    const uint32_t d = (uint32_t)((pos >> 5) - base);
    const uint64_t w = ((uint64_t)word(d) << 32) | word(d + 1);
    return (uint32_t)((w << (pos & 31)) >> (64 - k));
  }
};

// Block header and code lengths (lane 0, serial), selector list and decode tables (whole wave).  Executed by ONE wave; the results
// are wave-uniform scalars.  Returns 0 or a CJS_E_* code.  `selp`: LDS scratch of 4096 words (the unary values, a nibble each).
__device__ uint64_t g_dec_clk[8];      // phase clock of candidate 0 (CJS_DEBUG): 100 MHz ticks
// a list of eight nibbles: nibble j to the front / the list x read at the positions y holds
__device__ __forceinline__ uint32_t nib_to_front(uint32_t st, uint32_t j) {
  const uint32_t val = (st >> (4u * j)) & 15u, low = st & ((1u << (4u * j)) - 1u);
  return (st & ~((1u << (4u * j + 4u)) - 1u)) | (low << 4) | val;
}
__device__ __forceinline__ uint32_t nib_compose(uint32_t x, uint32_t y) {
  uint32_t r = 0;
#pragma unroll
  for (int p = 0; p < 8; p++) r |= ((x >> (4u * ((y >> (4 * p)) & 15u))) & 15u) << (4 * p);
  return r;
}
__device__ int dec_prologue(DecShared& S, BitReader& r, uint32_t dbuf_size, uint32_t& crc, uint32_t& orig, uint32_t& sym_total,
                            uint32_t& group_count, uint32_t& n_sel, uint8_t* __restrict__ selectors /* global, room for 32768 */, uint32_t* __restrict__ selp) {
  int err = 0;
  sym_total = 0; group_count = 0; n_sel = 0; orig = 0;
  const int lane = lane_id();
  if (lane == 0) {                                           // header (:1440-1493)
    crc = r.get(16) << 16; crc |= r.get(16);
    if (r.get(1)) err = CJS_E_OBSOLETE_INPUT;
    orig = r.get(24);
    if (!err && orig > dbuf_size) err = CJS_E_DATA_ERROR;
    const uint32_t t = r.get(16);
    for (int i = 0; i < 256; i++) S.sym_to_byte[i] = 0;
    for (int i = 0; i < 16; i++) if (t & (1u << (15 - i))) {
      const uint32_t k = r.get(16);
      for (int j = 0; j < 16; j++) if (k & (1u << (15 - j))) S.sym_to_byte[sym_total++] = (uint8_t)(i * 16 + j);
    }
    group_count = r.get(3);
    if (!err && (group_count < 2 || group_count > 6)) err = CJS_E_DATA_ERROR;
    n_sel = r.get(15);
    if (!err && n_sel == 0) err = CJS_E_DATA_ERROR;
  }
  err = __builtin_amdgcn_readfirstlane(err);
  group_count = __builtin_amdgcn_readfirstlane(group_count); n_sel = __builtin_amdgcn_readfirstlane(n_sel); sym_total = __builtin_amdgcn_readfirstlane(sym_total);
  uint64_t pos = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)r.pos) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(r.pos >> 32)) << 32);
  if (!err) {
    // Selector list (:1487-1493): n_sel unary numbers (ones closed by a zero).  Every lane takes 32 bits of a 2048-bit stretch:
    // its zeros are selector ends, the ones in front of a zero (back to the previous zero, which may sit in the lane before) the
    // value.  A value may equal group_count (the reference tests the count BEFORE it reads on: j ones pass for j <= group_count,
    // and its list holds zeros behind the groups); one more is an error.
    for (uint32_t i = lane; i < (n_sel + 7) / 8; i += 64) selp[i] = 0;
    __builtin_amdgcn_wave_barrier();
    const uint64_t nbytes = (r.nbits + 7) >> 3;
    uint32_t done = 0, carry = 0; int bad = 0; uint64_t endpos = pos;
    while (done < n_sel) {
      const uint64_t bp = pos + 32u * (uint32_t)lane;
      const uint32_t w0 = load_be32(r.p, nbytes, bp >> 5), w1 = load_be32(r.p, nbytes, (bp >> 5) + 1);
      const uint32_t v = (uint32_t)(((((uint64_t)w0 << 32) | w1) << (bp & 31)) >> 32);
      const uint32_t nz = (uint32_t)__builtin_popcount(~v);
      const uint32_t incl = wave_incl_sum(nz), excl = incl - nz;
      const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      const uint32_t t1 = v == 0xFFFFFFFFu ? 32u : (uint32_t)__builtin_ctz(~v);      // ones at the end of my word (32: all of it -- more than any value)
      uint32_t pend = (uint32_t)__shfl_up((int)t1, 1, 64);
      if (lane == 0) pend = carry;
      uint32_t z = ~v, k = done + excl; int prevb = -1 - (int)pend;
      while (z && k < n_sel) {
        const int bidx = __builtin_clz(z);
        const uint32_t j = (uint32_t)(bidx - prevb - 1);
        if (j > group_count) bad = 1;
        else atomicOr(&selp[k >> 3], j << (4u * (k & 7u)));
        if (k + 1 == n_sel) endpos = bp + (uint32_t)bidx + 1u;
        z &= ~(0x80000000u >> bidx); prevb = bidx; k++;
      }
      if (done + total >= n_sel) {                                       // the lane that holds the last selector knows where the list ends
        const uint64_t m = __ballot(done + incl >= n_sel);
        const int l = (int)__builtin_ctzll(m);
        endpos = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)endpos, l) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(endpos >> 32), l) << 32);
        done = n_sel;
      } else { done += total; carry = (uint32_t)__builtin_amdgcn_readlane((int)t1, 63); pos += 2048; }
    }
    if (__ballot(bad != 0)) err = CJS_E_DATA_ERROR;
    __builtin_amdgcn_wave_barrier();
    if (!err) {
      // Move-to-front over the values, the list as nibbles of one register.  A stretch of the list acts on the positions as a
      // permutation whatever they hold: every lane composes its stretch's (from the identity), a scan composes those in front of each
      // lane, and a second walk from the list the lane really starts with writes the selectors.  (Position group_count holds the
      // zero of the reference's zero-initialised list: a value equal to the count reads it, and moves it.)
      const uint32_t cs = (((n_sel + 63u) >> 6) + 7u) & ~7u;             // values per lane: whole words of selp
      const uint32_t c0 = min((uint32_t)lane * cs, n_sel), c1 = min(c0 + cs, n_sel);
      uint32_t R = 0x76543210u, wv = 0;
      for (uint32_t i = c0; i < c1; i++) {
        if ((i & 7u) == 0) wv = selp[i >> 3];
        R = nib_to_front(R, wv & 15u); wv >>= 4;
      }
      uint32_t I = R;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)I, d, 64);
        if (lane >= d) I = nib_compose(up, I);
      }
      uint32_t E = (uint32_t)__shfl_up((int)I, 1, 64);
      if (lane == 0) E = 0x76543210u;
      uint32_t st = 0;
      for (uint32_t i = 0; i < group_count; i++) st |= i << (4u * i);
      st = nib_compose(st, E);
      for (uint32_t i = c0; i < c1; i++) {
        if ((i & 7u) == 0) wv = selp[i >> 3];
        const uint32_t j = wv & 15u; wv >>= 4;
        selectors[i] = (uint8_t)((st >> (4u * j)) & 15u);
        st = nib_to_front(st, j);
      }
    }
    pos = endpos > r.nbits ? r.nbits : endpos;
  }
  if (!err) {
    // Code lengths (:1500-1520): per table 5 bits, then per symbol a run of (1, direction) pairs closed by a 0.  Behind a 0 and behind
    // a direction bit stands a control bit, so in a run of ones the bits alternate control / direction from the run's first (a control
    // bit): what a lane's first bit is follows from the parity of the ones in front of it.  Every lane walks 32 bits of a 2048-bit
    // stretch twice: once for its count of symbol ends and its sum of steps, and -- with the sums of the lanes in front -- once more
    // to write the lengths.  The reference tests the running length wherever it has changed (and where a table starts): 1 .. 20.
    const uint64_t nbytes = (r.nbits + 7) >> 3;
    const uint32_t sym_count0 = sym_total + 2;
    int bad = 0;
    for (uint32_t g = 0; g < group_count && !__ballot(bad != 0); g++) {
      int cur0;
      { const uint32_t w0 = load_be32(r.p, nbytes, pos >> 5), w1 = load_be32(r.p, nbytes, (pos >> 5) + 1);
        cur0 = (int)((((((uint64_t)w0 << 32) | w1) << (pos & 31)) >> 59)); pos += 5; }
      if (cur0 < 1 || cur0 > 20) bad = 1;
      uint32_t done = 0, par_in = 0;
      for (;;) {
        const uint64_t bp = pos + 32u * (uint32_t)lane;
        const uint32_t w0 = load_be32(r.p, nbytes, bp >> 5), w1 = load_be32(r.p, nbytes, (bp >> 5) + 1);
        const uint32_t v = (uint32_t)(((((uint64_t)w0 << 32) | w1) << (bp & 31)) >> 32);
        const uint32_t t1 = v == 0xFFFFFFFFu ? 32u : (uint32_t)__builtin_ctz(~v);
        const uint64_t m_ao = __ballot(v == 0xFFFFFFFFu), m_par = __ballot((t1 & 1u) != 0);
        const uint64_t below = ~m_ao & ((1ull << lane) - 1ull);
        const uint32_t par = below ? (uint32_t)((m_par >> (63 - __builtin_clzll(below))) & 1ull) : par_in;      // 1: my first bit is a direction bit
        uint32_t state = par, ne = 0; int nd = 0;
        for (int b = 31; b >= 0; b--) {
          const uint32_t bit = (v >> b) & 1u;
          if (state) { nd += bit ? -1 : 1; state = 0; }
          else if (bit) state = 1;
          else ne++;
        }
        const uint32_t ie = wave_incl_sum(ne); const int id = wave_incl_sum(nd);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)ie, 63);
        uint32_t idx = done + ie - ne; int cur = cur0 + id - nd; uint64_t endpos = 0;
        state = par;
        for (int b = 31; b >= 0 && idx < sym_count0; b--) {
          const uint32_t bit = (v >> b) & 1u;
          if (state) { cur += bit ? -1 : 1; if (cur < 1 || cur > 20) bad = 1; state = 0; }
          else if (bit) state = 1;
          else { S.length[g][idx] = (uint8_t)cur; if (++idx == sym_count0) endpos = bp + (uint32_t)(32 - b); }
        }
        if (__ballot(bad != 0)) break;                                   // (the reference stops at the first length out of range; so must a stretch of ones)
        if (done + total >= sym_count0) {                                // the lane that wrote the last length knows where the table ends
          const int l = (int)__builtin_ctzll(__ballot(done + ie >= sym_count0));
          pos = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)endpos, l) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(endpos >> 32), l) << 32);
          break;
        }
        done += total; cur0 += __builtin_amdgcn_readlane(id, 63);
        if (~m_ao) par_in = (uint32_t)((m_par >> (63 - __builtin_clzll(~m_ao))) & 1ull);
        pos += 2048;
      }
    }
    if (__ballot(bad != 0)) err = CJS_E_DATA_ERROR;
  }
  if (lane == 0) r.pos = pos > r.nbits ? r.nbits : pos;
  // lane 0's results become wave-uniform scalars (readfirstlane, not a shuffle: the compiler must KNOW they are uniform,
  // or the whole symbol loop is compiled as divergent code under exec masks)
  err = __builtin_amdgcn_readfirstlane(err);
  sym_total = __builtin_amdgcn_readfirstlane(sym_total); group_count = __builtin_amdgcn_readfirstlane(group_count);
  n_sel = __builtin_amdgcn_readfirstlane(n_sel);
  const uint32_t sym_count = sym_total + 2;
  __builtin_amdgcn_wave_barrier();
  if (!err) {
    // canonical tables: lane g builds table g -- a counting sort of the symbols by length, then the first codes
    if ((uint32_t)lane < group_count) {
      const int g = lane;
      for (int i = 0; i < 22; i++) { S.cnt[g][i] = 0; S.first[g][i] = 0; S.start[g][i] = 0; }
      int mn = 20, mx = 1;
      for (uint32_t i = 0; i < sym_count; i++) { const int l = S.length[g][i]; S.cnt[g][l]++; mn = l < mn ? l : mn; mx = l > mx ? l : mx; }
      S.minlen[g] = (uint8_t)mn; S.maxlen[g] = (uint8_t)mx;
      uint32_t code = 0, at = 0; uint16_t fillp[22];
      for (int l = mn; l <= mx; l++) {
        S.first[g][l] = code; S.start[g][l] = (uint16_t)at; fillp[l] = (uint16_t)at;
        at += S.cnt[g][l];
        code = (code + S.cnt[g][l]) << 1;
      }
      for (uint32_t i = 0; i < sym_count; i++) S.bysym[g][fillp[S.length[g][i]]++] = (uint16_t)i;
      for (int l = 1; l <= 20; l++) {
        const uint32_t lj = l < mn ? 0u : l <= mx ? (S.first[g][l] + S.cnt[g][l]) << (20 - l) : (1u << 20);
        S.limp[g][l] = l == 20 ? (mx == 20 ? lj : (S.first[g][mx] + S.cnt[g][mx]) << (20 - mx)) : (l < mx ? lj : (l >= mn ? (1u << 20) : 0u));
      }
      S.limp[g][0] = 0;
    }
  }
  __builtin_amdgcn_wave_barrier();
  crc = __builtin_amdgcn_readfirstlane(crc);
  orig = __builtin_amdgcn_readfirstlane(orig);
  r.pos = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)r.pos) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(r.pos >> 32)) << 32);   // (the builtin returns int: no sign extension of a low half >= 2^31)
  return err;
}

// ---------------------------------------------------------------- 2b. block decode in stages
// The Huffman chain of a block looks serial -- the table changes every 50 symbols, so there is no self-synchronisation to exploit
// -- but the only thing one group of 50 symbols hands to the next is WHERE IT ENDS.  So:
//   bz_chain        one workgroup per candidate: header, selector list and code tables (wave 0, in 2048-bit stretches), then up to
//                   four groups per step: for every bit position i of a group's positions thread i looks up the length of the code
//                   that WOULD start there: next[i] = i + len.  Five rounds of pointer doubling in LDS (next^2 .. next^32) give
//                   next^50(0) = 2 + 16 + 32 for the first group; the groups behind it get tables of their own over positions
//                   placed ahead, built in the same rounds (chain_tables).  About ten LDS round trips per 200 symbols instead of
//                   50 x (lookup + hop) per group by one lone wave.
//   bz_group_syms   one lane per group, all groups of all candidates at once: the 50 symbols from the group's start; the first
//                   end-of-block symbol and the first undecodable code of the block by 64-bit atomic minima.
//   bz_sym_ops      one workgroup per candidate over the symbols in front of the end-of-block: RUNA/RUNB digits -> byte counts,
//                   running output offset, every rank symbol leaves as (rank, output offset).  No move-to-front, no output bytes.
//   bz_mtf_tiles    the move-to-front of 256 consecutive rank ops of a block, started from the identity list, by one wave
//                   (list as bytes across the lanes, shift by wave_shr DPP): op j becomes q_j = the slot of the TILE-START
//                   list it reads, and the tile leaves its permutation P_t.   All tiles of all blocks in parallel.
//   bz_mtf_chunk_perm + bz_mtf_compose  chain the tiles: start list L_(t+1)[p] = L_t[P_t[p]] (an LDS gather per tile), in chunks of 64 tiles.
//   bz_mtf_emit     one thread per op: byte = L_t[q_j] at its offset, and the zero-rank run behind it (the gap to the next
//                   op's offset) is filled with the same byte (long runs by the whole wave).
// Same results as the reference loop (:1597-1670): every way it can fail there is DATA_ERROR, so a block is good iff its first
// end-of-block symbol comes before its first undecodable code, the selectors do not run out first, and the bytes fit the block.
struct RowTab {                    // per candidate row, in global memory between the stages
  uint16_t fast[6][1024];          // (sym << 5) | len by the next 10 bits, 0 = not decodable within 10 bits
  uint32_t first[6][22];
  uint16_t cnt[6][22], start[6][22], bysym[6][260];
  uint8_t minlen[8], maxlen[8];
  uint32_t sym_total, group_count, n_sel, err;      // err: the header's verdict
  uint64_t data_bit;               // first bit of the symbol data
  uint32_t crc, orig;
  uint32_t ngroups_ok;             // groups whose start bit is known (the chain's extent)
  uint32_t pad;
  unsigned long long eob_key;      // min over end-of-block symbols of (symbol index << 32 | bit behind the code - data_bit); ~0 = none
  unsigned long long err_key;      // min over undecodable codes of (symbol index << 32); ~0 = none
};
constexpr uint32_t CH_T = 512;                  // threads of bz_chain
constexpr uint32_t CH_SPAN = 1024;              // bit positions of a group's span (50 codes of <= 20 bits)
constexpr uint32_t CH_ARR = CH_SPAN + 64;
constexpr uint32_t CH_NONE = CH_ARR - 1;         // next[] of a position where no code of the table starts (an entry of its own, like the positions behind a span)
constexpr uint32_t CH_WIN = CH_T - 64;          // positions of a table in a step's first attempt: one per thread, the last 64 map to themselves
constexpr uint32_t CH_ARR2 = CH_T + 64;         // the later groups' tables: CH_WIN positions, and 64 + 64 that map to themselves
constexpr uint32_t CH_NONE2 = CH_ARR2 - 1;
constexpr uint32_t CH_WORDS = 2048;             // 32-bit words of the stream kept in LDS (65536 bits: ~180 groups of text)
constexpr uint32_t GROUP_SYMS = 50;
constexpr uint32_t MAX_SELECTORS = 32768;


// bz_chain's tables: level lv (next^(2^lv)) of table t by byte offset.  A level is read by the round behind it only (the hops aside: 1, 4
// and 5), so four arrays hold the six: 0 and 3 share one, 2 and 5 another -- and the step behind writes its level 0 where this step's
// last lookups (level 5) do not read.
__host__ __device__ constexpr int ch_slot(int lv) { return lv == 1 ? 0 : lv == 4 ? 1 : (lv == 2 || lv == 5) ? 2 : 3; }
template <uint32_t N> __device__ __forceinline__ uint32_t ch_ld(uint16_t (*t)[N], int lv, uint32_t off) { return *reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(t[ch_slot(lv)]) + off); }
template <uint32_t N> __device__ __forceinline__ void ch_st(uint16_t (*t)[N], int lv, uint32_t i, uint32_t v) { t[ch_slot(lv)][i] = (uint16_t)v; }
// The 20 bits at bit o of the window
__device__ __forceinline__ uint32_t chain_bits(const uint32_t* wbuf, uint32_t o) {
  const uint32_t w0 = wbuf[o >> 5], w1 = wbuf[(o >> 5) + 1];
  return (uint32_t)(((((uint64_t)w0 << 32) | w1) << (o & 31)) >> 44);
}
// The position behind the code that starts at position i with the 20 bits x20 under table g (CH_NONE: no code starts there).  e is the
// entry of the 12-bit direct table: the length of a code of <= 12 bits, CH_NOCODE, or 0 for a longer code, whose length is 13 + the
// number of lengths 13 .. 19 whose left-justified codes all lie below x20 (well under a percent of ARBITRARY bit offsets: one wave in
// four holds one; with a 10-bit table nearly every wave did and paid the compares).
constexpr uint32_t CH_NOCODE = 0xFF;
__device__ __forceinline__ uint32_t chain_next(const DecShared& S, int g, uint32_t e, uint32_t x20, uint32_t i, uint32_t none) {
  if (e) return e == CH_NOCODE ? none : i + e;
  uint32_t len = 13;
#pragma unroll
  for (int l = 13; l <= 19; l++) len += x20 >= S.limp[g][l] ? 1u : 0u;
  return x20 < S.limp[g][20] ? i + len : none;
}
// One step's tables for NT groups (k on A from its known start, k + q on L[q - 1] from bit start[q - 1] of the step): next^1, then five
// rounds of doubling.  Tables hold BYTE OFFSETS (2 x position) into a level's array, and every position behind a table's last (>= 64 of
// them: a code is at most 20 bits) maps to itself, as does CH_NONE: a chain that has left its positions, or met one where no code
// starts, stays where it is without a compare -- a round is one gather and one store per table, all threads on all tables, no branch
// (the CU's scalar unit serves every wave's branches and address arithmetic: with one thread per position and table, and a branch
// around each, those instructions outnumbered the vector ones three to one and set the pace).  e[q]: where group k + q ends, in
// positions of its table (every lane the same value; CH_NONE / CH_NONE2 or beyond the table's positions: not known from this step).
// Group k's 50 codes are 2 + 16 + 32: the first two hops as soon as their table stands, beside the following round's gathers.
struct ChainLater { uint16_t (*t)[CH_ARR2]; int g; uint32_t start; };
template <int NT>
__device__ __forceinline__ void chain_tables(const DecShared& S, const uint8_t (*len12)[4096], const uint32_t* wbuf, uint16_t (*A)[CH_ARR], const ChainLater (&L)[3],
                                             uint32_t i, uint32_t o0, int g, uint32_t span, bool whole, uint32_t (&e)[4]) {
  // (the tables' loads side by side: two LDS latencies for all of them)
  uint32_t x[NT], en[NT], m[NT];
  x[0] = chain_bits(wbuf, o0 + i);
#pragma unroll
  for (int q = 1; q < NT; q++) x[q] = chain_bits(wbuf, o0 + L[q - 1].start + i);
  en[0] = len12[g][x[0] >> 8];
#pragma unroll
  for (int q = 1; q < NT; q++) en[q] = len12[L[q - 1].g][x[q] >> 8];
  m[0] = chain_next(S, g, en[0], x[0], i, CH_NONE);
  m[0] = 2 * (i < span ? m[0] : i);
#pragma unroll
  for (int q = 1; q < NT; q++) { m[q] = chain_next(S, L[q - 1].g, en[q], x[q], i, CH_NONE2); m[q] = 2 * (i < CH_WIN ? m[q] : i); }
  ch_st(A, 0, i, m[0]);
#pragma unroll
  for (int q = 1; q < NT; q++) ch_st(L[q - 1].t, 0, i, m[q]);
  __syncthreads();
  uint32_t hop = 0;
#pragma unroll
  for (int lv = 1; lv <= 4; lv++) {
    m[0] = ch_ld(A, lv - 1, m[0]);
#pragma unroll
    for (int q = 1; q < NT; q++) m[q] = ch_ld(L[q - 1].t, lv - 1, m[q]);
    ch_st(A, lv, i, m[0]);
#pragma unroll
    for (int q = 1; q < NT; q++) ch_st(L[q - 1].t, lv, i, m[q]);
    __syncthreads();
    if (lv == 1) hop = ch_ld(A, 1, 0);
    if (lv == 4) hop = ch_ld(A, 4, hop);
  }
  // The last round: A's next^32 (one hop is left for it), but the later tables' next^50 = next^2 . next^16 . next^32 outright: two more
  // gathers here (every position at once) instead of two more hops each behind the barrier (one after the other).
  m[0] = ch_ld(A, 4, m[0]);
#pragma unroll
  for (int q = 1; q < NT; q++) m[q] = ch_ld(L[q - 1].t, 4, m[q]);
  ch_st(A, 5, i, m[0]);
#pragma unroll
  for (int q = 1; q < NT; q++) m[q] = ch_ld(L[q - 1].t, 4, m[q]);
#pragma unroll
  for (int q = 1; q < NT; q++) m[q] = ch_ld(L[q - 1].t, 1, m[q]);
#pragma unroll
  for (int q = 1; q < NT; q++) ch_st(L[q - 1].t, 5, i, m[q]);
  __syncthreads();
  // Where the groups end.  A chain that has left its positions STAYS on the value it left with, and a value equal to the number of
  // positions may be such a stop in the middle of the group: only a value below it is the end of 50 codes for sure -- except on the
  // whole span, whose last position nothing but 50 codes of the longest length reach.  A group that starts outside its table's
  // positions is looked up at CH_NONE2, which maps to itself.
  e[0] = ch_ld(A, 5, hop) >> 1;
  e[1] = e[2] = e[3] = CH_NONE2;
  bool known = e[0] < span || (whole && e[0] == span);
  uint32_t at = e[0];                            // where the group in front ends, in bits of the step
#pragma unroll
  for (int q = 1; q < NT; q++) {
    const bool in = known && at >= L[q - 1].start && at - L[q - 1].start < CH_WIN;
    e[q] = ch_ld(L[q - 1].t, 5, 2 * (in ? at - L[q - 1].start : CH_NONE2)) >> 1;
    known = e[q] < CH_WIN;
    at = L[q - 1].start + e[q];
  }
}
// The same for group k alone on its whole span (two positions per thread)
__device__ __forceinline__ uint32_t chain_table_full(const DecShared& S, const uint8_t (*len12)[4096], const uint32_t* wbuf, uint16_t (*A)[CH_ARR], uint32_t i, uint32_t o0, int g, uint32_t span) {
  const uint32_t j = i + CH_T;
  const uint32_t x0 = chain_bits(wbuf, o0 + i), x1 = chain_bits(wbuf, o0 + j);
  uint32_t m0 = chain_next(S, g, len12[g][x0 >> 8], x0, i, CH_NONE), m1 = chain_next(S, g, len12[g][x1 >> 8], x1, j, CH_NONE);
  m0 = 2 * (i < span ? m0 : i); m1 = 2 * (j < span ? m1 : j);
  ch_st(A, 0, i, m0); ch_st(A, 0, j, m1);
  __syncthreads();
  uint32_t hop = 0;
#pragma unroll
  for (int lv = 1; lv <= 5; lv++) {
    m0 = ch_ld(A, lv - 1, m0); m1 = ch_ld(A, lv - 1, m1);
    ch_st(A, lv, i, m0); ch_st(A, lv, j, m1);
    __syncthreads();
    if (lv == 1) hop = ch_ld(A, 1, 0);
    if (lv == 4) hop = ch_ld(A, 4, hop);
  }
  return hop;
}

__global__ __launch_bounds__(CH_T) void bz_chain(const uint8_t* __restrict__ in, uint64_t n, const Cand* __restrict__ cands, uint32_t ncand, uint32_t dbuf_size,
                                                 RowTab* __restrict__ tabs, uint8_t* __restrict__ sel_all, uint32_t* __restrict__ gstart_all,
                                                 uint8_t* __restrict__ l0_all, BlockOut* __restrict__ outs, uint32_t row0) {
  __shared__ DecShared S;
  __shared__ uint32_t scratch[4 * CH_ARR / 2 + 12 * CH_ARR2 / 2 + CH_WORDS + 2];         // the prologue's selector values (4096 words), then the chain's arrays
  uint16_t (*A)[CH_ARR] = reinterpret_cast<uint16_t (*)[CH_ARR]>(scratch);                     // group k: next^(2^lv), lv = 0 .. 5, in four arrays (ch_slot)
  uint16_t (*B)[CH_ARR2] = reinterpret_cast<uint16_t (*)[CH_ARR2]>(scratch + 4 * CH_ARR / 2);                    // groups k + 1, k + 2, k + 3: four arrays each
  uint32_t* wbuf = scratch + 4 * CH_ARR / 2 + 12 * CH_ARR2 / 2;
  __shared__ uint8_t len12[6][4096];            // code length by the next 12 bits (chain_next)
  __shared__ uint64_t s_pos;
  __shared__ uint32_t s_hdr[8];
  const uint32_t c = blockIdx.x;
  if (c >= ncand) return;
  const int tid = threadIdx.x, lane = tid & 63;
  if (cands[c].kind != 0) {                     // end-of-stream candidate: nothing to decode
    if (tid == 0) { BlockOut bo; bo.end_bit = cands[c].bit + 48; bo.count = 0; bo.orig = 0; bo.crc = 0; bo.err = 0; outs[c] = bo; }
    return;
  }
  const uint32_t row = cands[c].pad - row0;     // row of this batch's scratch
  RowTab& T = tabs[row];
  uint8_t* sel = sel_all + (size_t)row * MAX_SELECTORS;
  uint32_t* gstart = gstart_all + (size_t)row * (MAX_SELECTORS + 1);
  const uint64_t t_k0 = wall_clock64();
  if (tid < 64) {                               // wave 0: header, selectors, code lengths, tables
    BitReader r{in, n * 8, cands[c].bit + 48, 0, ~0ull >> 4};
    uint32_t sym_total = 0, group_count = 0, n_sel = 0, orig = 0, crc = 0;
    const int err = dec_prologue(S, r, dbuf_size, crc, orig, sym_total, group_count, n_sel, sel, scratch);
    if (lane == 0) {
      s_hdr[0] = (uint32_t)err; s_hdr[1] = sym_total; s_hdr[2] = group_count; s_hdr[3] = n_sel; s_hdr[4] = crc; s_hdr[5] = orig;
      s_pos = r.pos;
    }
  }
  __threadfence_block();
  __syncthreads();
  const uint64_t t_hdr = wall_clock64();
  if (tid == 0 && blockIdx.x == 0) g_dec_clk[5] = t_hdr - t_k0;
  const int herr = (int)s_hdr[0];
  const uint32_t group_count = s_hdr[2], n_sel = s_hdr[3];
  const uint64_t data_bit = s_pos;
  // the tables go to global memory for the symbol stage
  // ... with the 10-bit direct tables: entry x = the code that is a prefix of the 10 bits x, if it has one of <= 10 bits
  if (!herr) {
    for (uint32_t e = tid; e < group_count * 1024u; e += CH_T) {
      const uint32_t t = e >> 10, x = e & 1023u;
      const int mn = S.minlen[t], mx = S.maxlen[t];
      uint16_t v = 0;
      for (int l = mn; l <= 10 && l <= mx; l++) {
        const uint32_t k = (x >> (10 - l)) - S.first[t][l];              // (not below first: no shorter code matched)
        if (k < S.cnt[t][l]) { v = (uint16_t)((S.bysym[t][S.start[t][l] + k] << 5) | l); break; }
      }
      T.fast[t][x] = v;
    }
  }
  for (uint32_t i = tid; i < 6 * 22; i += CH_T) { (&T.first[0][0])[i] = (&S.first[0][0])[i]; (&T.cnt[0][0])[i] = (&S.cnt[0][0])[i]; (&T.start[0][0])[i] = (&S.start[0][0])[i]; }
  for (uint32_t i = tid; i < 6 * 260; i += CH_T) (&T.bysym[0][0])[i] = (&S.bysym[0][0])[i];
  if (tid < 8) { T.minlen[tid] = S.minlen[tid]; T.maxlen[tid] = S.maxlen[tid]; }
  for (int i = tid; i < 256; i += CH_T) l0_all[(size_t)row * 256 + i] = S.sym_to_byte[i];
  // 12-bit direct length tables.  The length rule is 1 + #{L in 1..19 : x20 >= limp[L]}, and a limit of a length <= 12 has its low 8
  // bits clear: the 12 bits x decide those; with all 12 below x the code is longer (or there is none) and the entry is 0.
  if (!herr) {
    for (uint32_t e = tid; e < group_count * 4096u; e += CH_T) {
      const uint32_t t = e >> 12, x = e & 4095u;
      uint32_t c = 0;
#pragma unroll
      for (int l = 1; l <= 12; l++) c += x >= (S.limp[t][l] >> 8) ? 1u : 0u;
      len12[t][x] = (uint8_t)(c == 12 ? 0u : (x << 8) < S.limp[t][20] ? c + 1 : CH_NOCODE);
    }
  }
  // (the positions behind the last a thread writes, once for every level)
  if (tid < 64) { for (int a = 0; a < 4; a++) A[a][CH_SPAN + tid] = (uint16_t)(2 * (CH_SPAN + tid)); for (int a = 0; a < 12; a++) B[a][CH_T + tid] = (uint16_t)(2 * (CH_T + tid)); }
  __syncthreads();
  uint32_t ok_groups = 0;
  if (!herr) {
    // Nothing the step's first instructions need comes from memory: the tables' shortest / longest lengths sit in registers (5 / 10 bits
    // each: the longest, and 50 x the shortest), and lane j of every wave holds the selectors kb + 8 j .. kb + 8 j + 7 as nibbles (15:
    // none) -- written by wave 0 above, visible after the barrier; 512 selectors per fill.
    uint32_t maxp = 0; uint64_t minp = 0;
    for (int t = 0; t < 6; t++) { minp |= (uint64_t)(GROUP_SYMS * ((uint32_t)S.minlen[t] & 31u)) << (10 * t); maxp |= ((uint32_t)S.maxlen[t] & 31u) << (5 * t); }
    maxp = (uint32_t)__builtin_amdgcn_readfirstlane((int)maxp);
    minp = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)minp) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(minp >> 32)) << 32);
    uint32_t kb = 0, sw = 0, lp = 0;
    uint32_t rel = 0, o0 = CH_WORDS * 32;          // bits from data_bit / from the window's first word to the step's first bit (o0 past the window: fill it)
    for (uint32_t k = 0; k < n_sel;) {
      if (k == 0 || k - kb >= 512 - 16) {          // (uniform) the next 512 selectors
        kb = k & ~7u;
        const uint32_t at = kb + 8u * (uint32_t)lane;
        uint64_t v = ~0ull;
        if (at < n_sel) v = *reinterpret_cast<const uint64_t*>(sel + at);                  // (the row is 8-byte aligned; bytes behind the list: masked)
        if (at + 8 > n_sel && at < n_sel) v |= ~0ull << (8u * (n_sel - at));
        v = (v | (v >> 4)) & 0x00FF00FF00FF00FFull; v = (v | (v >> 8)) & 0x0000FFFF0000FFFFull; v = v | (v >> 16);
        sw = (uint32_t)v;
      }
      const uint32_t kj = (uint32_t)__builtin_amdgcn_readfirstlane((int)(k - kb));
      const uint64_t sn = (((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)sw, (int)((kj >> 3) + 1)) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)sw, (int)(kj >> 3))) >> (4u * (kj & 7u));
      // Up to FOUR groups per step.  The tables next^1 .. next^32 of a group do not depend on where the group starts, only on its
      // code table and on the bit positions they cover.  So beside group k's tables from its known start (A) the same threads build
      // those of group k + 1 under ITS code table over CH_WIN positions from the earliest bit it can start at (50 x the shortest code
      // of group k's table), and of groups k + 2 and k + 3 likewise -- in the same rounds, behind the same barriers.  When the hops
      // on A have found where group k ends, the later tables are looked up there, one after the other.  The span that is safe for
      // any 50 codes (50 x the longest) is about five times what 50 codes of text take (~210 bits): the first attempt works on CH_WIN
      // positions; a group k that leaves them (or meets a position where no code starts) is worked out again, alone, on its whole
      // span; a later group that leaves its table's positions, or starts in front of them, simply is the first group of the next step.
      const int g = (int)(sn & 15u);
      const uint32_t n1 = (uint32_t)(sn >> 4) & 15u, n2 = (uint32_t)(sn >> 8) & 15u, n3 = (uint32_t)(sn >> 12) & 15u;
      const int g1 = n1 < 6 ? (int)n1 : -1, g2 = n2 < 6 ? (int)n2 : -1, g3 = n3 < 6 ? (int)n3 : -1;
      const uint32_t full_span = min(GROUP_SYMS * ((maxp >> (5 * g)) & 31u), CH_SPAN);
      const uint32_t base1 = (uint32_t)(minp >> (10 * g)) & 1023u;                          // group k + 1 starts at or behind this offset
      const uint32_t base2 = base1 + ((uint32_t)(minp >> (10 * min(n1, 5u))) & 1023u);       // ... group k + 2 at or behind this one
      const uint32_t base3 = base2 + ((uint32_t)(minp >> (10 * min(n2, 5u))) & 1023u);       // ... and group k + 3 here (no group: not used)
      if (o0 + 2 * CH_SPAN + 128 > CH_WORDS * 32) {                                         // (uniform) refill the bit window
        __syncthreads();
        const uint64_t pos = data_bit + rel, wbase = pos >> 5;
        for (uint32_t i = tid; i < CH_WORDS + 2; i += CH_T) wbuf[i] = load_be32(in, n, wbase + i);
        o0 = (uint32_t)(pos & 31u);
        __syncthreads();
      }
      const uint32_t i = (uint32_t)tid;
      // (where the later tables start: the earliest bit the group can start at, or -- lp = the bits of the last group -- half a group in
      // front of where groups of that length would put it, if that is more: its CH_WIN positions must hold the group's start AND end)
      const ChainLater L[3] = {{B, g1, min(base1, CH_SPAN)},
                               {B + 4, g2, min(max(base2, lp * 3 / 2), 2 * CH_SPAN - CH_T)},
                               {B + 8, g3, min(max(base3, lp * 5 / 2), 2 * CH_SPAN - CH_T)}};
      uint32_t span = min(full_span, CH_WIN), e[4];
      if (g3 >= 0) chain_tables<4>(S, len12, wbuf, A, L, i, o0, g, span, span == full_span, e);
      else if (g2 >= 0) chain_tables<3>(S, len12, wbuf, A, L, i, o0, g, span, span == full_span, e);
      else if (g1 >= 0) chain_tables<2>(S, len12, wbuf, A, L, i, o0, g, span, span == full_span, e);
      else chain_tables<1>(S, len12, wbuf, A, L, i, o0, g, span, span == full_span, e);
      uint32_t e0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)e[0]);
      const uint32_t e1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)e[1]), e2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)e[2]), e3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)e[3]);
      bool ok0 = e0 < span || (span == full_span && e0 == span);
      if (!ok0 && span < full_span) {              // (uniform) group k alone on its whole span
        span = full_span;
        e0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ch_ld(A, 5, chain_table_full(S, len12, wbuf, A, i, o0, g, span))) >> 1;
        ok0 = e0 <= span;
      }
      if (tid == 0) gstart[k] = rel;
      ok_groups = k + 1;
      if (!ok0) break;                             // (uniform) a code of the group is undecodable: the symbol stage reports it -- or finds the end of the block in front of it
      // (a later group's end is known only if those in front of it are: chain_tables)
      uint32_t adv;
      if (e3 < CH_WIN) {                           // four groups
        if (tid == 0) { gstart[k + 1] = rel + e0; gstart[k + 2] = rel + L[0].start + e1; gstart[k + 3] = rel + L[1].start + e2; }
        ok_groups = k + 4;
        lp = L[2].start + e3 - (L[1].start + e2);
        adv = L[2].start + e3; k += 4;
      } else if (e2 < CH_WIN) {                    // three
        if (tid == 0) { gstart[k + 1] = rel + e0; gstart[k + 2] = rel + L[0].start + e1; }
        ok_groups = k + 3;
        lp = L[1].start + e2 - (L[0].start + e1);
        adv = L[1].start + e2; k += 3;
      } else if (e1 < CH_WIN) {                    // two
        if (tid == 0) gstart[k + 1] = rel + e0;
        ok_groups = k + 2;
        lp = L[0].start + e1 - e0;
        adv = L[0].start + e1; k += 2;
      } else { lp = e0; adv = e0; k += 1; }        // (the second group left its table's positions, met an undecodable position, or there was none: next step)
      rel += adv; o0 += adv;
    }
  }
  if (tid == 0 && blockIdx.x == 0) { g_dec_clk[6] = wall_clock64() - t_hdr; g_dec_clk[7] = ok_groups; }
  if (tid == 0) {
    T.sym_total = s_hdr[1]; T.group_count = group_count; T.n_sel = n_sel; T.err = (uint32_t)herr; T.data_bit = data_bit; T.crc = s_hdr[4]; T.orig = s_hdr[5];
    T.ngroups_ok = ok_groups; T.pad = 0; T.eob_key = ~0ull; T.err_key = ~0ull;
  }
}

// one lane per group of 50 symbols; the block's tables in LDS
__global__ __launch_bounds__(256) void bz_group_syms(const uint8_t* __restrict__ in, uint64_t n, RowTab* __restrict__ tabs, const uint8_t* __restrict__ sel_all,
                                                     const uint32_t* __restrict__ gstart_all, uint16_t* __restrict__ syms_all, uint32_t sym_stride, uint32_t sym_groups, uint32_t row0) {
  __shared__ uint16_t fast[6][1024];
  __shared__ uint32_t first[6][22];
  __shared__ uint16_t cnt[6][22], start[6][22], bysym[6][260];
  __shared__ uint8_t maxlen[8];
  const uint32_t row = row0 + blockIdx.y;
  RowTab& T = tabs[row];
  const uint32_t ng = T.err ? 0u : T.ngroups_ok;
  if (blockIdx.x * 256u >= ng) return;
  const int tid = threadIdx.x;
  for (uint32_t i = tid; i < 6 * 1024; i += 256) (&fast[0][0])[i] = (&T.fast[0][0])[i];
  for (uint32_t i = tid; i < 6 * 22; i += 256) { (&first[0][0])[i] = (&T.first[0][0])[i]; (&cnt[0][0])[i] = (&T.cnt[0][0])[i]; (&start[0][0])[i] = (&T.start[0][0])[i]; }
  for (uint32_t i = tid; i < 6 * 260; i += 256) (&bysym[0][0])[i] = (&T.bysym[0][0])[i];
  if (tid < 8) maxlen[tid] = T.maxlen[tid];
  __syncthreads();
  const uint32_t k = blockIdx.x * 256u + tid;
  if (k >= ng) return;
  const int g = sel_all[(size_t)row * MAX_SELECTORS + k];
  const uint32_t sym_total = T.sym_total;
  const uint64_t data_bit = T.data_bit;
  uint64_t pos = data_bit + gstart_all[(size_t)row * (MAX_SELECTORS + 1) + k];
  // symbol j of group k is stored at [j][k]: the lanes of a wave (64 groups) write one line, not 64
  uint16_t* syms = syms_all + (size_t)row * sym_groups * GROUP_SYMS + k;
  const int mx = maxlen[g];
  // a 64-bit window on the stream: a code is at most 20 bits, so the cursor crosses at most one word per symbol (one load every
  // four or five symbols of text instead of two per symbol)
  uint64_t wdw = pos >> 5;
  uint32_t w0 = load_be32(in, n, wdw), w1 = load_be32(in, n, wdw + 1);
  for (uint32_t j = 0; j < GROUP_SYMS; j++) {
    const uint64_t idx = (uint64_t)k * GROUP_SYMS + j;
    if ((pos >> 5) != wdw) { wdw++; w0 = w1; w1 = load_be32(in, n, wdw + 1); }
    const uint32_t x20 = (uint32_t)(((((uint64_t)w0 << 32) | w1) << (pos & 31)) >> 44);
    uint32_t e = fast[g][x20 >> 10], sym = 0, len = 0;
    if (e) { sym = e >> 5; len = e & 31u; }
    else {
      for (int i = 11; i <= mx; i++) {
        const uint32_t q = (x20 >> (20 - i)) - first[g][i];
        if (q < cnt[g][i]) { len = (uint32_t)i; sym = bysym[g][start[g][i] + q]; break; }
      }
    }
    if (!len || idx >= sym_stride) { atomicMin(&T.err_key, (unsigned long long)idx << 32); break; }      // no code starts here (or more symbols than any block has room for)
    pos += len;
    if (sym > sym_total) { atomicMin(&T.eob_key, ((unsigned long long)idx << 32) | (unsigned long long)(uint32_t)(pos - data_bit)); break; }      // end of block (:1640)
    syms[(size_t)j * sym_groups] = (uint16_t)sym;
  }
}

// One workgroup per row: the symbols in front of the end-of-block symbol, 4096 per tile, front to back.
//   RUNA (0) / RUNB (1) are the bijective base-2 digits of a zero-rank run (:1621-1638): digit d adds (sym + 1) << d bytes.  The
//   reference keeps the digit weight in an int32 that it shifts left: the 32nd digit of a run adds (sym + 1) * -2^31, leaves the
//   weight 0, and with it the run is forgotten (the flush at :1643 tests the weight); a 33rd digit starts a fresh run.  So the
//   digit of a run symbol is its position in the run mod 32, and a symbol with digit 31 takes back what the 31 in front of it added.
//   rank symbols (>= 2) emit one byte each and leave as op (rank - 1, output offset); the bytes must fit the block (:1647, :1663).
constexpr int SO_PT = 8;                       // symbols per thread (a tile is a dozen barriers whatever it holds: 0.58 / 0.47 / 0.87 ms per 100 MB with 4 / 8 / 16)
constexpr uint32_t SO_TILE = 1024 * SO_PT, SO_KG = SO_TILE / GROUP_SYMS + 2;      // a tile's symbols lie in SO_KG groups at most
__global__ __launch_bounds__(1024) void bz_sym_ops(RowTab* __restrict__ tabs, const Cand* __restrict__ cands, uint32_t ncand, const uint16_t* __restrict__ syms_all,
                                                   uint32_t sym_groups, uint32_t dbuf_size, uint8_t* __restrict__ ops_all, uint32_t* __restrict__ opoff_all,
                                                   uint32_t ops_stride, uint32_t* __restrict__ nops_all, BlockOut* __restrict__ outs, uint32_t row0, uint64_t nbits) {
  __shared__ uint16_t st[SO_TILE + 32];          // the tile's symbols behind the last 32 of the tile in front
  __shared__ unsigned long long sm64[16];
  __shared__ uint32_t sm[16];
  __shared__ uint32_t mx[1024];
  const uint32_t c = blockIdx.x;
  if (c >= ncand || cands[c].kind != 0) return;
  const uint32_t row = cands[c].pad - row0;
  const RowTab& T = tabs[row];
  const int tid = threadIdx.x;
  int err = (int)T.err;
  const unsigned long long ek = T.eob_key, xk = T.err_key;
  if (!err && (ek == ~0ull || xk < ek)) err = CJS_E_DATA_ERROR;      // no end of block in the selectors' reach, or an undecodable code in front of it
  const uint32_t nsym = err ? 0u : (uint32_t)(ek >> 32);
  const uint16_t* syms = syms_all + (size_t)row * sym_groups * GROUP_SYMS;      // [symbol of the group][group] (bz_group_syms)
  uint8_t* ops = ops_all + (size_t)row * ops_stride;
  uint32_t* opoff = opoff_all + (size_t)row * ops_stride;
  unsigned long long off = 0;                    // bytes so far
  uint32_t j0 = 0, last_nonrun = 0;              // ops so far; (index of the last rank symbol so far) + 1
  if (tid < 32) st[tid] = 2;                     // in front of the first symbol: not a run
  for (uint32_t base = 0; base < nsym && !err; base += SO_TILE) {
    __syncthreads();
    {                                             // along the groups, symbol by symbol of the group
      const uint32_t k0 = base / GROUP_SYMS;
      constexpr int LR = (GROUP_SYMS * SO_KG + 1023) / 1024;
      uint32_t at[LR]; uint16_t v[LR];            // (all of a thread's loads on their way before the first is stored; fetching the tile
                                                  // behind during this one's work as well: 0.44 -> 0.70 ms)
#pragma unroll
      for (int q = 0; q < LR; q++) {
        const uint32_t u = (uint32_t)tid + 1024u * (uint32_t)q, j = u / SO_KG, k = k0 + (u - SO_KG * j), i = k * GROUP_SYMS + j;
        const bool in = u < GROUP_SYMS * SO_KG && i >= base && i < base + SO_TILE;
        at[q] = in ? 32u + i - base : ~0u;
        v[q] = in && i < nsym ? syms[(size_t)j * sym_groups + k] : (uint16_t)2;
      }
#pragma unroll
      for (int q = 0; q < LR; q++) if (at[q] != ~0u) st[at[q]] = v[q];
    }
    __syncthreads();
    // position in the run: i - (index of the last rank symbol in front of i) - 1, by a max scan of (index + 1) of the rank symbols
    uint32_t lastb = 0;
#pragma unroll
    for (int q = 0; q < SO_PT; q++) { const uint32_t i = base + (uint32_t)tid * (uint32_t)SO_PT + q; if (i < nsym && st[32 + tid * SO_PT + q] >= 2) lastb = i + 1; }
    const uint32_t incl = block_incl_max<1024>(lastb, sm);
    mx[tid] = incl;
    __syncthreads();
    uint32_t prevnr = tid ? mx[tid - 1] : 0u;
    const uint32_t tile_last = mx[1023];
    __syncthreads();
    if (prevnr < last_nonrun) prevnr = last_nonrun;
    long long cb[SO_PT]; unsigned long long mine = 0; uint32_t nops = 0;
#pragma unroll
    for (int q = 0; q < SO_PT; q++) {
      const uint32_t i = base + (uint32_t)tid * (uint32_t)SO_PT + q, sy = st[32 + tid * SO_PT + q];
      cb[q] = 0;
      if (i < nsym) {
        if (sy >= 2) { cb[q] = 1; nops++; prevnr = i + 1; }
        else {
          const uint32_t d = (i - prevnr) & 31u;
          if (d < 31) cb[q] = (long long)(sy + 1u) << d;
          else { long long t = 0; for (uint32_t b = 1; b <= 31; b++) t += (long long)((uint32_t)st[32 + tid * SO_PT + q - b] + 1u) << (31u - b); cb[q] = -t; }
        }
      }
      mine += (unsigned long long)cb[q];
    }
    unsigned long long tot;
    unsigned long long ex = off + block_excl_sum<1024>(mine, sm64, tot);
    uint32_t ntot;
    uint32_t jx = j0 + block_excl_sum<1024>(nops, sm, ntot);
#pragma unroll
    for (int q = 0; q < SO_PT; q++) {
      const uint32_t i = base + (uint32_t)tid * (uint32_t)SO_PT + q, sy = st[32 + tid * SO_PT + q];
      if (i < nsym && sy >= 2) {
        if (jx < ops_stride - 1u && ex < 0xFFFFFFFFull) { ops[jx] = (uint8_t)(sy - 1u); opoff[jx] = (uint32_t)ex; }      // rank symbol s reads list slot s - 1 (:1664)
        jx++;
      }
      ex += (unsigned long long)cb[q];
    }
    off += tot; j0 += ntot;                       // (off may hold digits of a run that is still open: the byte limit is tested at the end)
    if (tile_last > last_nonrun) last_nonrun = tile_last;
    if (j0 >= ops_stride - 1u) err = CJS_E_DATA_ERROR;                           // (uniform) more rank symbols than the block has bytes
    __syncthreads();
    if (tid < 32) st[tid] = st[SO_TILE + tid];   // the last 32 symbols stay in front of the next tile
  }
  // Offsets at rank symbols never decrease, so every "fits the block" test of the reference (:1647 before a flush, :1663 before
  // a literal) passes iff the final byte count does
  if (!err && off > dbuf_size) err = CJS_E_DATA_ERROR;
  if (!err && T.orig >= off) err = CJS_E_DATA_ERROR;                            // :1677
  if (tid == 0) {
    if (err) { j0 = 0; off = 0; }
    opoff[j0] = (uint32_t)off;                   // the end-of-block pseudo op: where the output ends
    nops_all[row] = j0;
    BlockOut bo;
    const uint64_t endb = T.data_bit + (uint32_t)ek;
    bo.end_bit = err ? 0 : (endb > nbits ? nbits : endb);
    bo.count = err ? 0u : (uint32_t)off; bo.orig = T.orig; bo.crc = T.crc; bo.err = err;
    outs[c] = bo;
  }
}

constexpr uint32_t MT_TILE = 256;
// grid = (tile groups, rows of a slab): rows come in slabs of <= 65535 (grid.y), and a grid may not exceed 2^32 threads in all
__global__ __launch_bounds__(256) void bz_mtf_tiles(uint8_t* __restrict__ ops_all, uint32_t ops_stride, const uint32_t* __restrict__ nops_all,
                                                    uint8_t* __restrict__ pl_all, uint32_t tiles_per_row, uint32_t row0) {
  const uint32_t row = row0 + blockIdx.y, t = blockIdx.x * 4u + (threadIdx.x >> 6);
  const int lane = lane_id();
  const uint32_t nops = nops_all[row];
  if ((size_t)t * MT_TILE >= nops) return;
  const uint32_t len = nops - t * MT_TILE < MT_TILE ? nops - t * MT_TILE : MT_TILE;
  uint32_t* opw = reinterpret_cast<uint32_t*>(ops_all + (size_t)row * ops_stride + (size_t)t * MT_TILE);
  const uint32_t opsreg = opw[lane];
  uint32_t L = (uint32_t)(4 * lane) * 0x01010101u + 0x03020100u;      // identity list: slot 4*lane+b holds 4*lane+b
  uint32_t qreg = 0;
  const int lane4m1 = 4 * lane - 1;
#pragma unroll 1
  for (uint32_t j = 0; j < len; j++) {
    const uint32_t k = ((uint32_t)__builtin_amdgcn_readlane(opsreg, j >> 2) >> (8u * (j & 3u))) & 0xFFu;
    const uint32_t v = ((uint32_t)__builtin_amdgcn_readlane(L, k >> 2) >> (8u * (k & 3u))) & 0xFFu;
    const uint32_t up = __builtin_amdgcn_update_dpp(v << 24, L, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    const uint32_t shifted = __builtin_amdgcn_alignbit(L, up, 24);
    int nbm = (int)k - lane4m1;                                          // bytes of this lane that move one slot up
    nbm = nbm < 0 ? 0 : nbm > 4 ? 4 : nbm;
    const uint32_t m = (uint32_t)((1ull << (8 * nbm)) - 1ull);
    L = (shifted & m) | (L & ~m);
    qreg = (uint32_t)lane == (j >> 2) ? qreg | (v << (8u * (j & 3u))) : qreg;
  }
  opw[lane] = qreg;                                                      // (bytes behind len are scratch: the row has room for a whole tile)
  reinterpret_cast<uint32_t*>(pl_all + ((size_t)row * tiles_per_row + t) * MT_TILE)[lane] = L;
}

// The tiles of a row are chained in chunks of MC_TILES: first every chunk's product of permutations (from the identity), then -- chunk
// by chunk again, all at once -- the start list of the chunk (the row's initial list through the products of the chunks in front) and
// from it the start list of each of its tiles.  (One workgroup walking a row's ~1,300 tiles, two barriers each, took 0.48 ms per 100 MB.)
constexpr uint32_t MC_TILES = 64;
__global__ __launch_bounds__(256) void bz_mtf_chunk_perm(const uint32_t* __restrict__ nops_all, const uint8_t* __restrict__ pl_all, uint32_t tiles_per_row,
                                                         uint8_t* __restrict__ cperm_all, size_t cperm_stride) {
  __shared__ uint8_t sl[256];
  const uint32_t row = blockIdx.y, p = threadIdx.x;
  const uint32_t nt = (nops_all[row] + MT_TILE - 1) / MT_TILE, t0 = blockIdx.x * MC_TILES, t1 = min(t0 + MC_TILES, nt);
  if (t0 >= nt) return;
  const uint8_t* pl = pl_all + (size_t)row * tiles_per_row * MT_TILE;
  uint8_t cur = (uint8_t)p, idx = pl[(size_t)t0 * MT_TILE + p];
  for (uint32_t t = t0; t < t1; t++) {
    const uint8_t nidx = t + 1 < t1 ? pl[(size_t)(t + 1) * MT_TILE + p] : 0;      // the next permutation travels while this one is applied
    sl[p] = cur;
    __syncthreads();
    cur = sl[idx];
    __syncthreads();
    idx = nidx;
  }
  cperm_all[(size_t)row * cperm_stride + (size_t)blockIdx.x * 256 + p] = cur;
}
__global__ __launch_bounds__(256) void bz_mtf_compose(const uint32_t* __restrict__ nops_all, const uint8_t* __restrict__ l0_all, uint8_t* __restrict__ pl_all,
                                                      uint32_t tiles_per_row, const uint8_t* __restrict__ cperm_all, size_t cperm_stride) {
  __shared__ uint8_t sl[256];
  const uint32_t row = blockIdx.y, p = threadIdx.x;
  const uint32_t nt = (nops_all[row] + MT_TILE - 1) / MT_TILE, t0 = blockIdx.x * MC_TILES, t1 = min(t0 + MC_TILES, nt);
  if (t0 >= nt) return;
  uint8_t* pl = pl_all + (size_t)row * tiles_per_row * MT_TILE;
  const uint8_t* cp = cperm_all + (size_t)row * cperm_stride;
  uint8_t cur = l0_all[(size_t)row * 256 + p];
  uint8_t idx = blockIdx.x ? cp[p] : pl[(size_t)t0 * MT_TILE + p];
  for (uint32_t c = 0; c < blockIdx.x; c++) {                                      // the chunks in front
    const uint8_t nidx = c + 1 < blockIdx.x ? cp[(size_t)(c + 1) * 256 + p] : pl[(size_t)t0 * MT_TILE + p];
    sl[p] = cur;
    __syncthreads();
    cur = sl[idx];
    __syncthreads();
    idx = nidx;
  }
  for (uint32_t t = t0; t < t1; t++) {
    const uint8_t nidx = t + 1 < t1 ? pl[(size_t)(t + 1) * MT_TILE + p] : 0;
    sl[p] = cur;
    __syncthreads();
    pl[(size_t)t * MT_TILE + p] = cur;                                            // start list of tile t, in place of its permutation
    cur = sl[idx];
    __syncthreads();
    idx = nidx;
  }
}

__global__ __launch_bounds__(256) void bz_mtf_emit(const uint8_t* __restrict__ q_all, const uint32_t* __restrict__ opoff_all, uint32_t ops_stride,
                                                   const uint32_t* __restrict__ nops_all, const uint8_t* __restrict__ l0_all, const uint8_t* __restrict__ pl_all,
                                                   uint32_t tiles_per_row, uint32_t row0, uint8_t* __restrict__ tt_all, uint32_t dbuf_size) {
  const uint32_t row = row0 + blockIdx.y, t = blockIdx.x;
  const int lane = lane_id();
  const uint32_t nops = nops_all[row];
  if ((size_t)t * MT_TILE >= nops && t) return;
  const uint8_t* q = q_all + (size_t)row * ops_stride;
  const uint32_t* opoff = opoff_all + (size_t)row * ops_stride;
  uint8_t* tt = tt_all + (size_t)row * dbuf_size;
  const uint32_t j = t * MT_TILE + threadIdx.x;
  uint32_t o = 0, gap = 0, byte = 0;
  bool live = j < nops;
  if (live) {
    byte = pl_all[((size_t)row * tiles_per_row + t) * MT_TILE + q[j]];
    o = opoff[j];
    gap = opoff[j + 1] - o - 1u;                            // the zero-rank run behind this op repeats its byte (the list front)
    tt[o] = (uint8_t)byte;
    o++;
  }
  if (live && gap <= 16u) { for (uint32_t x = 0; x < gap; x++) tt[o + x] = (uint8_t)byte; live = false; }
  uint64_t mask = __ballot(live && gap > 16u);
  while (mask) {                                           // long runs: the whole wave fills
    const int l = (int)__builtin_ctzll(mask);
    mask &= mask - 1;
    const uint32_t jo = (uint32_t)__builtin_amdgcn_readlane((int)o, l), jl = (uint32_t)__builtin_amdgcn_readlane((int)gap, l);
    const uint32_t jb = (uint32_t)__builtin_amdgcn_readlane((int)byte, l);
    for (uint32_t x = lane; x < jl; x += 64) tt[jo + x] = (uint8_t)jb;
  }
  if (t == 0 && threadIdx.x < 64) {                        // the run in front of the first op repeats the initial list front
    const uint32_t pre = opoff[0], b0 = l0_all[(size_t)row * 256];
    for (uint32_t x = lane; x < pre; x += 64) tt[x] = (uint8_t)b0;
  }
}

// decoded rows of a batch -> their packed places (phase A with several batches)
struct RowDst { uint64_t dst; uint64_t count; };      // device address of the row's packed place, bytes to copy
__global__ __launch_bounds__(256) void bz_rows_pack(const uint8_t* __restrict__ rows, uint32_t stride, const RowDst* __restrict__ desc) {
  const RowDst d = desc[blockIdx.y];
  uint8_t* __restrict__ dst = reinterpret_cast<uint8_t*>(d.dst);
  const uint32_t cnt = (uint32_t)d.count;
  const uint8_t* __restrict__ src = rows + (size_t)blockIdx.y * stride;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < cnt; i += gridDim.x * 256) dst[i] = src[i];
}

// ---------------------------------------------------------------- 4. inverse BWT
struct IbBlock {            // per valid block, in stream order
  uint64_t tt;              // device address of the block's decoded BWT bytes
  uint32_t count;           // n
  uint32_t orig;
  uint32_t off;             // element offset of the block in the concatenated sort / LF arrays
  uint32_t woff;            // byte offset of the block in the walk's output (w)
  uint64_t out_off;         // byte offset of the block in the final output
  uint32_t out_len;
  uint32_t crc;
};

// keys (block << 8 | byte), vals = i
__global__ __launch_bounds__(256) void ib_make_keys(const IbBlock* __restrict__ blocks, uint32_t* __restrict__ key, uint32_t* __restrict__ val, uint32_t stride) {
  const IbBlock b = blocks[blockIdx.y];
  const uint8_t* __restrict__ tt = reinterpret_cast<const uint8_t*>(b.tt);
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < b.count; i += gridDim.x * 256) {
    key[b.off + i] = ((uint32_t)blockIdx.y << 8) | tt[i];
    val[b.off + i] = i;
  }
}
// For the segmented pass every block owns `stride` slots (one segment each), and a key is (i << 8) | byte -- the index rides on the
// key, there is no value array (a block has < 2^24 bytes); the slots behind the block's bytes hold the largest digit, which the stable
// sort leaves behind everything real.  One workgroup per radix tile (RS_TILE slots of a block's range): the keys, and the tile's count of
// every byte -- the row of the histogram the pass would otherwise read the keys again for (four copies per wave, interleaved: see
// rs_hist_bytes)
__global__ __launch_bounds__(256) void ib_make_keys_hist(const IbBlock* __restrict__ blocks, uint32_t* __restrict__ key, uint32_t stride, uint32_t tps,
                                                         uint32_t* __restrict__ hist) {
  constexpr int HC = 4;
  __shared__ uint32_t h[4 * 256 * HC];
  const IbBlock b = blocks[blockIdx.y];
  const uint8_t* __restrict__ tt = reinterpret_cast<const uint8_t*>(b.tt);
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4 * HC; i++) h[i * 256 + tid] = 0;
  __syncthreads();
  uint32_t* hw = h + (tid >> 6) * 256 * HC + (tid & (HC - 1));
  const uint32_t t0 = blockIdx.x * RS_TILE;
#pragma unroll 4
  for (uint32_t e = tid; e < RS_TILE; e += 256) {
    const uint32_t i = t0 + e;
    if (i < stride) {
      const uint32_t by = i < b.count ? tt[i] : 0xFFu;
      key[b.off + i] = i < b.count ? (i << 8) | by : 0xFFu;
      atomicAdd(&hw[by * HC], 1u);
    }
  }
  __syncthreads();
  uint32_t sum = 0;
#pragma unroll
  for (int w = 0; w < 4; w++)
#pragma unroll
    for (int r = 0; r < HC; r++) sum += h[w * 256 * HC + tid * HC + r];
  hist[((size_t)blockIdx.y * tps + blockIdx.x) * 256 + tid] = sum;
}
// after the stable sort: slot j of the block holds (T[j] << 8) | tt[j] == the reference's dbuf (:1686-1690):
// the pointer comes from the sorted order, the low byte is the j-th DECODED byte (not the sorted one)
// (on_key: the sorted array holds (i << 8) | sorted byte, see ib_make_keys)
__global__ __launch_bounds__(256) void ib_pack(const IbBlock* __restrict__ blocks, const uint32_t* __restrict__ val, uint32_t* __restrict__ dbuf, int on_key) {
  const IbBlock b = blocks[blockIdx.y];
  const uint8_t* __restrict__ tt = reinterpret_cast<const uint8_t*>(b.tt);
  for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < b.count; j += gridDim.x * 256)
    dbuf[b.off + j] = (on_key ? val[b.off + j] & 0xFFFFFF00u : val[b.off + j] << 8) | tt[j];
}
// sentinel variant (BWT.unbwtransform, J/BWTC_joined_.js:1147-1168): next(t) = LF[t] + C[T[t]] (+1 below pidx) = the stable
// sorted position of element t; slot t holds (next(t) << 8) | T[t].  b.orig carries pidx.
__global__ __launch_bounds__(256) void ib_pack_sentinel(const IbBlock* __restrict__ blocks, const uint32_t* __restrict__ val, uint32_t* __restrict__ dbuf) {
  const IbBlock b = blocks[blockIdx.y];
  const uint8_t* __restrict__ tt = reinterpret_cast<const uint8_t*>(b.tt);
  for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < b.count; j += gridDim.x * 256) {
    const uint32_t t = val[b.off + j];
    dbuf[b.off + t] = ((j + (j < b.orig ? 1u : 0u)) << 8) | tt[t];
  }
}
// splitters: slot j with j % SPL == 0, plus the start slot.  Walk until the next splitter.
__device__ __forceinline__ bool is_split(uint32_t j, uint32_t start) { return (j % SPL) == 0 || j == start; }
// The walks below are n dependent random 4-byte reads per block.  With every block's walkers spread over the chip each XCD's
// 4 MiB L2 sees all blocks' vectors (3.6 MB each at level 9) and every step is a 64-byte fetch from memory: 2.4 + 3.2 ms per
// 100 MB.  So (1) workgroup 8 j + x -- it runs on XCD x -- takes chunk x * ceil(T / 8) + j of the (block, 256 splitters) chunks:
// an XCD works through a contiguous range of blocks; (2) the launches ask for WALK_LDS bytes of LDS they never touch, which
// leaves five workgroups per CU: 40 K walkers per XCD, the splitters of about six blocks.  Measured (walk1 + walk2 per 100 MB):
// blocks spread over the chip 5.57 ms; XCD ranges with 0 / 30 / 60 / 100 KB of LDS asked for 4.71 / 4.06 / 4.11 / 5.51 ms -- fewer
// blocks in flight hit the L2 more often but leave too few walkers to hide what still misses (with a splitter every 64 slots:
// 0 / 16 / 30 / 45 / 60 KB 3.53 / 3.51 / 2.90 / 2.91 / 3.32 ms).
constexpr uint32_t WALK_T = 256, WALK_LDS = 30 * 1024;
__host__ __device__ __forceinline__ uint32_t walk_chunks(uint32_t max_count) { return ((max_count + SPL - 1) / SPL + 1 + WALK_T - 1) / WALK_T; }
__device__ __forceinline__ bool walk_item(uint32_t nblocks, uint32_t cpb, uint32_t& blk, uint32_t& sidx) {
  const uint32_t T = nblocks * cpb, per = (T + 7u) >> 3;
  const uint32_t t = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
  if (t >= T) return false;
  blk = t / cpb; sidx = (t - blk * cpb) * WALK_T + threadIdx.x;
  return true;
}
constexpr uint32_t SPL_END = 0xFFFFFFFEu;   // the chain left the block (sentinel variant: the row of the implicit end marker)
// `seg` (cyclic form): the walk also KEEPS what it reads -- the bytes of the first SEG_CAP slots it visits, SEG_CAP bytes per splitter,
// sixteen at a time -- and the slot it stands on after SEG_CAP steps (`resume`): once the splitters are ranked the bytes only have
// to be put in their places (ib_place), and a second walk (ib_walk2) is left for what the few long stretches hold behind SEG_CAP.
// (Both walks were the same 100 M random 4-byte reads, a 64-byte line each: about 1.1 ms per 100 MB apiece.)
constexpr uint32_t SEG_CAP = 384;           // a stretch is longer with probability e^-6
__global__ __launch_bounds__(WALK_T) void ib_walk1(const uint32_t* __restrict__ dbuf, const IbBlock* __restrict__ blocks, uint32_t nblocks, uint32_t cpb,
                                                   uint32_t spl_stride, uint32_t* __restrict__ spl_next, uint32_t* __restrict__ spl_steps, int sentinel,
                                                   uint8_t* __restrict__ seg, uint32_t* __restrict__ resume) {
  uint32_t blk, sidx;
  if (!walk_item(nblocks, cpb, blk, sidx)) return;
  const IbBlock b = blocks[blk];
  const uint32_t* d = dbuf + b.off;
  const uint32_t nspl = (b.count + SPL - 1) / SPL + 1;         // regular splitters + one slot for `start`
  if (sidx >= nspl) return;
  const uint32_t start = sentinel ? 0u : d[b.orig] >> 8;       // first slot visited by the loop (:1698-1700)
  uint32_t pos;
  if (sidx == nspl - 1) { pos = start; if ((start % SPL) == 0) { spl_steps[(size_t)blk * spl_stride + sidx] = 0; spl_next[(size_t)blk * spl_stride + sidx] = start / SPL; return; } }
  else pos = sidx * SPL;
  uint32_t steps = 0, cur = pos;
  if (seg) {
    uint4* sb = reinterpret_cast<uint4*>(seg + ((size_t)blk * spl_stride + sidx) * SEG_CAP);
    uint32_t acc[4] = {0u, 0u, 0u, 0u};
    do {
      const uint32_t e = d[cur];
      if (steps < SEG_CAP) {
        const uint32_t k = (steps >> 2) & 3u;
#pragma unroll
        for (int t = 0; t < 4; t++) if (k == (uint32_t)t) acc[t] = (acc[t] >> 8) | (e << 24);
        if ((steps & 15u) == 15u) sb[steps >> 4] = make_uint4(acc[0], acc[1], acc[2], acc[3]);
      }
      cur = e >> 8; steps++;
      if (steps == SEG_CAP) resume[(size_t)blk * spl_stride + sidx] = cur;
    } while (cur < b.count && !is_split(cur, start) && steps < b.count);
    if (steps < SEG_CAP && (steps & 15u)) {                    // the open piece: its words' bytes stand at the top
      const uint32_t k = (steps >> 2) & 3u, r = 8u * (4u - (steps & 3u));
#pragma unroll
      for (int t = 0; t < 4; t++) if (k == (uint32_t)t && (steps & 3u)) acc[t] >>= r;
      sb[steps >> 4] = make_uint4(acc[0], acc[1], acc[2], acc[3]);
    }
  } else {
    do { cur = d[cur] >> 8; steps++; } while (cur < b.count && !is_split(cur, start) && steps < b.count);
  }
  spl_steps[(size_t)blk * spl_stride + sidx] = steps;
  spl_next[(size_t)blk * spl_stride + sidx] = cur >= b.count ? SPL_END : (cur == start && (start % SPL) != 0) ? nspl - 1 : cur / SPL;
}
// the kept bytes of 64 stretches (splitters s0 .. s0 + 63 of block b, their ranks and lengths one per lane) to their places: four
// stretches in flight at a time, the lanes along the bytes
__device__ __forceinline__ void ib_place64(const IbBlock& b, size_t at0, uint32_t s0, uint32_t nspl, const uint32_t* __restrict__ spl_rank,
                                           const uint32_t* __restrict__ spl_steps, const uint8_t* __restrict__ seg, uint8_t* __restrict__ w) {
  const int lane = lane_id();
  uint32_t rank = 0, nbytes = 0;
  if (s0 + lane < nspl) {
    rank = spl_rank[at0 + lane];
    const uint32_t steps = spl_steps[at0 + lane];
    if (rank != 0xFFFFFFFFu && steps != 0 && rank < b.count) nbytes = min(min(steps, SEG_CAP), b.count - rank);
  }
  const uint8_t* src = seg + at0 * SEG_CAP;
  for (int i = 0; i < 64; i += 4) {
    uint32_t r[4], nb[4], mx = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { r[j] = (uint32_t)__builtin_amdgcn_readlane((int)rank, i + j); nb[j] = (uint32_t)__builtin_amdgcn_readlane((int)nbytes, i + j); mx = max(mx, nb[j]); }
    for (uint32_t o = lane; o < mx; o += 64) {
      uint8_t v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = o < nb[j] ? src[(size_t)(i + j) * SEG_CAP + o] : (uint8_t)0;
#pragma unroll
      for (int j = 0; j < 4; j++) if (o < nb[j]) w[r[j] + o] = v[j];
    }
  }
}
__global__ __launch_bounds__(256) void ib_place(const IbBlock* __restrict__ blocks, uint32_t spl_stride, const uint32_t* __restrict__ spl_rank,
                                                const uint32_t* __restrict__ spl_steps, const uint8_t* __restrict__ seg, uint8_t* __restrict__ wbuf) {
  const IbBlock b = blocks[blockIdx.y];
  const uint32_t nspl = (b.count + SPL - 1) / SPL + 1, s0 = (blockIdx.x * 4u + (uint32_t)wave_id()) * 64u;
  if (s0 < nspl) ib_place64(b, (size_t)blockIdx.y * spl_stride + s0, s0, nspl, spl_rank, spl_steps, seg, wbuf + b.woff);
}
// rank the splitter chain from `start`: spl_rank[s] = number of output positions before splitter s's segment.
// One workgroup per block.  The chain is a list of <= 14064 nodes (a cycle through the start node for a cyclic BWT): the
// edge back into the first node is cut and the suffix sums of the segment lengths come from pointer jumping in LDS
// (14 rounds) instead of 14000 dependent loads; rank = total - suffix sum.  If the total is not the block length the
// permutation has a short cycle (periodic block) and lane 0 walks the chain the slow way, as the reference would.
constexpr uint32_t IBR_MAX = 14080;      // >= 900000 / SPL + 2
constexpr int IBR_PT = (IBR_MAX + 1023) / 1024;      // nodes per thread
__global__ __launch_bounds__(1024) void ib_rank(const IbBlock* __restrict__ blocks, uint32_t nblocks, uint32_t spl_stride, const uint32_t* __restrict__ spl_next,
                                                const uint32_t* __restrict__ spl_steps, uint32_t* __restrict__ spl_rank, int32_t* __restrict__ err) {
  __shared__ uint32_t nxt[IBR_MAX], dst[IBR_MAX];
  __shared__ uint32_t s_total;
  const uint32_t k = blockIdx.x;
  if (k >= nblocks) return;
  const IbBlock b = blocks[k];
  const uint32_t nspl = (b.count + SPL - 1) / SPL + 1;
  const uint32_t* nx = spl_next + (size_t)k * spl_stride; const uint32_t* st = spl_steps + (size_t)k * spl_stride;
  uint32_t* rk = spl_rank + (size_t)k * spl_stride;
  bool fast = nspl <= IBR_MAX;
  if (fast) {
    const uint32_t entry = nspl - 1;
    const bool alias = st[entry] == 0;                 // start % SPL == 0: the start entry only points at the regular splitter
    const uint32_t head = alias ? nx[entry] : entry;
    for (uint32_t i = threadIdx.x; i < nspl; i += 1024) {
      uint32_t n = nx[i];
      if (n == head && !(alias && i == entry)) n = SPL_END;       // the edge that closes the cycle
      if (n != SPL_END && n >= nspl) n = SPL_END;
      nxt[i] = n; dst[i] = st[i];
    }
    __syncthreads();
    for (uint32_t span = 1; span < nspl; span <<= 1) {
      uint32_t nn[IBR_PT], dd[IBR_PT];
#pragma unroll
      for (int q = 0; q < IBR_PT; q++) {
        const uint32_t i = threadIdx.x + 1024u * q;
        nn[q] = SPL_END; dd[q] = 0;
        if (i < nspl) { const uint32_t n = nxt[i]; if (n != SPL_END) { nn[q] = nxt[n]; dd[q] = dst[n]; } else nn[q] = SPL_END; }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < IBR_PT; q++) {
        const uint32_t i = threadIdx.x + 1024u * q;
        if (i < nspl && nxt[i] != SPL_END) { dst[i] += dd[q]; nxt[i] = nn[q]; }
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) s_total = dst[entry];
    __syncthreads();
    const uint32_t total = s_total;
    if (total == b.count) {
      for (uint32_t i = threadIdx.x; i < nspl; i += 1024) rk[i] = total - dst[i];
      if (threadIdx.x == 0) err[k] = (int32_t)total;
      return;
    }
    fast = false;
  }
  if (threadIdx.x != 0) return;
  for (uint32_t i = 0; i < nspl; i++) rk[i] = 0xFFFFFFFFu;
  uint32_t cur = nspl - 1, done = 0;
  // the start splitter may alias a regular one (start % SPL == 0): its entry has steps 0 and points at it
  for (uint32_t guard = 0; guard <= nspl + 1 && done < b.count; guard++) {
    if (rk[cur] != 0xFFFFFFFFu && st[cur] != 0) break;           // back on a visited splitter: the permutation has a short cycle
    rk[cur] = done; done += st[cur]; cur = nx[cur];
    if (cur == SPL_END) break;
  }
  // done < count: the LF permutation has a short cycle (periodic block, e.g. "abab"): the reference keeps walking
  // round it for `count` steps (:1732), i.e. the byte sequence is periodic with period `done`
  err[k] = (int32_t)done;
}
__global__ __launch_bounds__(256) void ib_periodic_fill(const IbBlock* __restrict__ blocks, const int32_t* __restrict__ cyc, uint8_t* __restrict__ wbuf) {
  const IbBlock b = blocks[blockIdx.y];
  const uint32_t L = (uint32_t)cyc[blockIdx.y];
  if (L == 0 || L >= b.count) return;
  uint8_t* w = wbuf + b.woff;
  for (uint32_t r = L + blockIdx.x * 256 + threadIdx.x; r < b.count; r += gridDim.x * 256) w[r] = w[r % L];
}
// second walk: write the pre-RLE1 byte sequence w[0..n) of each block (w[r] = byte of the (r+1)-th visited slot)
__global__ __launch_bounds__(WALK_T) void ib_walk2(const uint32_t* __restrict__ dbuf, const IbBlock* __restrict__ blocks, uint32_t nblocks, uint32_t cpb,
                                                   uint32_t spl_stride, const uint32_t* __restrict__ spl_rank, const uint32_t* __restrict__ spl_steps,
                                                   uint8_t* __restrict__ wbuf, int sentinel, const uint32_t* __restrict__ resume, const uint8_t* __restrict__ seg) {
  uint32_t blk, sidx;
  if (!walk_item(nblocks, cpb, blk, sidx)) return;
  const IbBlock b = blocks[blk];
  const uint32_t* d = dbuf + b.off;
  uint8_t* w = wbuf + b.woff;
  const uint32_t nspl = (b.count + SPL - 1) / SPL + 1;
  if (sidx >= nspl) return;
  const uint32_t start = sentinel ? 0u : d[b.orig] >> 8;
  uint32_t rank = spl_rank[(size_t)blk * spl_stride + sidx], steps = spl_steps[(size_t)blk * spl_stride + sidx];
  if (rank == 0xFFFFFFFFu || steps == 0) return;
  uint32_t cur = sidx == nspl - 1 ? start : sidx * SPL;
  if (seg) {                                      // only what ib_walk1 did not keep: from the slot it stood on after SEG_CAP steps
    if (steps <= SEG_CAP) return;
    cur = resume[(size_t)blk * spl_stride + sidx]; rank += SEG_CAP; steps -= SEG_CAP;
    if (rank >= b.count) return;
  }
  // visiting order: position `rank` of the walk is slot `cur`; the loop outputs the byte of every visited slot (:1735-1736)
  if (sentinel) {
    for (uint32_t q = 0; q < steps && rank + q < b.count; q++) {
      const uint32_t e = d[cur];
      w[b.count - 1 - (rank + q)] = (uint8_t)(e & 0xFF);                         // unbwtransform fills U from the end (BWTC:1161)
      cur = e >> 8;
    }
    return;
  }
  // A lane's bytes are neighbours in w, the lanes' stretches are not: a byte per store is 64 lines per wave instruction, and the
  // stores, not the reads, set the pace.  Whole aligned 16-byte pieces of the stretch go out as such, the words and then the bytes in
  // front of the first and behind the last one by one (they share their pieces with the stretches of other lanes).
  const uint32_t r1 = min(rank + steps, b.count);
  uint32_t r = rank;
  auto word = [&]() { uint32_t acc = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) { const uint32_t e = d[cur]; acc = (acc >> 8) | (e << 24); cur = e >> 8; }
    return acc; };
  for (; r < r1 && ((b.woff + r) & 3u); r++) { const uint32_t e = d[cur]; w[r] = (uint8_t)e; cur = e >> 8; }
  for (; r + 4 <= r1 && ((b.woff + r) & 15u); r += 4) *reinterpret_cast<uint32_t*>(w + r) = word();
  for (; r + 16 <= r1; r += 16) { uint4 q; q.x = word(); q.y = word(); q.z = word(); q.w = word(); *reinterpret_cast<uint4*>(w + r) = q; }
  for (; r + 4 <= r1; r += 4) *reinterpret_cast<uint32_t*>(w + r) = word();
  for (; r < r1; r++) { const uint32_t e = d[cur]; w[r] = (uint8_t)e; cur = e >> 8; }
}

// ---------------------------------------------------------------- 5. RLE1 expansion
// Stretch functions on the carried bit c0 ("this stretch starts with a count byte"): next = ((L - c0) % 5 == 4)
//   L % 5 == 4 -> NOT-ish (c0=0 ->1, c0=1 -> 0), L % 5 == 0 -> identity, else const 0.  Encoded as 2 bits (f(0) | f(1) << 1).
__device__ __forceinline__ uint32_t stretch_fn(uint32_t L) { const uint32_t m = L % 5; return m == 4 ? 1u : (m == 0 ? 2u : 0u); }
__device__ __forceinline__ uint32_t fn_apply(uint32_t f, uint32_t c) { return (f >> c) & 1u; }
__device__ __forceinline__ uint32_t fn_compose(uint32_t first, uint32_t then) {     // x -> then(first(x))
  return fn_apply(then, fn_apply(first, 0)) | (fn_apply(then, fn_apply(first, 1)) << 1);
}
// One tile (UR_TILE = 1024 threads x UR_BPT bytes) of a block by one workgroup.  Carried from the tiles in front: start of the current stretch,
// c0 of the current stretch, output bytes so far (from the length pass below); the bytes go out.
constexpr int UR_BPT_C = 16;
// A thread's UR_BPT = 16 bytes w[p0 .. p0 + 16) and the byte in front of them (0 at the block's first byte), bytes behind the block's
// end as 0.  The address has any alignment (the same for the whole workgroup): two aligned 16-byte loads and a funnel shift (a byte
// per load was 17 instructions of 16 lines each; the walk's output buffer has 64 bytes of slack behind its last block).
__device__ __forceinline__ void ur_load(const uint8_t* __restrict__ w, uint32_t n, uint32_t p0, uint8_t (&c)[UR_BPT_C + 1]) {
  const uintptr_t A = (uintptr_t)(w + p0);
  const uint4* q = reinterpret_cast<const uint4*>(A & ~(uintptr_t)15);
  const uint32_t r = 8u * ((uint32_t)A & 3u);
  uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
  if (p0 < n) { lo = q[0]; hi = q[1]; }             // (at most 31 bytes behind the block's end)
  uint32_t d[4];
  switch (((uint32_t)A >> 2) & 3u) {              // (uniform)
    case 0: d[0] = __funnelshift_r(lo.x, lo.y, r); d[1] = __funnelshift_r(lo.y, lo.z, r); d[2] = __funnelshift_r(lo.z, lo.w, r); d[3] = __funnelshift_r(lo.w, hi.x, r); break;
    case 1: d[0] = __funnelshift_r(lo.y, lo.z, r); d[1] = __funnelshift_r(lo.z, lo.w, r); d[2] = __funnelshift_r(lo.w, hi.x, r); d[3] = __funnelshift_r(hi.x, hi.y, r); break;
    case 2: d[0] = __funnelshift_r(lo.z, lo.w, r); d[1] = __funnelshift_r(lo.w, hi.x, r); d[2] = __funnelshift_r(hi.x, hi.y, r); d[3] = __funnelshift_r(hi.y, hi.z, r); break;
    default: d[0] = __funnelshift_r(lo.w, hi.x, r); d[1] = __funnelshift_r(hi.x, hi.y, r); d[2] = __funnelshift_r(hi.y, hi.z, r); d[3] = __funnelshift_r(hi.z, hi.w, r); break;
  }
#pragma unroll
  for (int j = 0; j < UR_BPT_C; j++) c[j + 1] = p0 + j < n ? (uint8_t)(d[j >> 2] >> (8 * (j & 3))) : (uint8_t)0;
  uint32_t prev = (uint32_t)__shfl_up((int)(d[3] >> 24), 1, 64);                 // the lane in front holds the byte in front (beyond n: never looked at)
  if (lane_id() == 0) prev = (p0 > 0 && p0 - 1 < n) ? w[p0 - 1] : 0u;
  c[0] = (uint8_t)prev;
}
struct RleCarry { uint32_t cur_start, cur_c0, out_base, pad; };
static_assert(UR_BPT_C == 16, "ur_load");
constexpr int UR_BPT = UR_BPT_C;                    // bytes per thread: the ten-step function scan over the 1024 threads is most of a tile, whatever the bytes per thread (4 bytes: 1.05 ms per 100 MB for the length pass)
constexpr uint32_t UR_TILE = 1024 * UR_BPT;
constexpr uint32_t UR_STAGE = 24 * 1024;      // bytes of a tile's output put together in LDS (a tile of plain text makes UR_TILE and a few)
template <bool WRITE>
__device__ __forceinline__ void unrle1_tile(const uint8_t* __restrict__ w, uint32_t n, uint32_t base, RleCarry& cy, uint8_t* __restrict__ o,
                                            uint32_t* sm, uint32_t* fnarr, uint32_t* posarr, uint8_t* stage) {
  const uint32_t p0 = base + threadIdx.x * UR_BPT;
  uint8_t c[UR_BPT + 1];
  ur_load(w, n, p0, c);
  // stretch boundaries inside my positions
  uint32_t bmask = 0, lastb = 0;
#pragma unroll
  for (int j = 0; j < UR_BPT; j++) { const uint32_t p = p0 + j; if (p < n && p > 0 && c[j + 1] != c[j]) { bmask |= 1u << j; lastb = p + 1; } }
  // previous boundary before my first position: max-scan of (boundary position + 1), 0 = none in this tile
  const uint32_t im = block_incl_max<1024>(lastb, sm);
  posarr[threadIdx.x] = im;
  __syncthreads();
  const uint32_t exb = threadIdx.x ? posarr[threadIdx.x - 1] : 0u;
  const uint32_t tile_last = posarr[1023];
  __syncthreads();
  // per-thread function = composition of the stretch functions of the boundaries in my positions (in order);
  // a boundary at p closes the stretch [prev_start, p) of length p - prev_start
  uint32_t f = 2u;   // identity
  {
    uint32_t ps = exb ? exb - 1 : cy.cur_start;
#pragma unroll
    for (int j = 0; j < UR_BPT; j++) if ((bmask >> j) & 1u) { const uint32_t p = p0 + j; f = fn_compose(f, stretch_fn(p - ps)); ps = p; }
  }
  // exclusive scan of function composition across the 1024 threads: shuffles inside the waves, the sixteen wave totals through LDS
  // (two barriers; as a ten-step Hillis-Steele scan in LDS, twenty barriers, this was most of a tile)
  uint32_t incl = f;
  {
    const int lane = lane_id(), wv = wave_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t other = (uint32_t)__shfl_up((int)incl, d, 64);
      if (lane >= d) incl = fn_compose(other, incl);           // (the earlier threads' function first)
    }
    if (lane == 63) fnarr[wv] = incl;
    __syncthreads();
    if (wv == 0) {
      uint32_t t = lane < 16 ? fnarr[lane] : 2u;
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) {
        const uint32_t other = (uint32_t)__shfl_up((int)t, d, 64);
        if (lane >= d) t = fn_compose(other, t);
      }
      if (lane < 16) fnarr[16 + lane] = t;                     // inclusive over the waves
    }
    __syncthreads();
  }
  const uint32_t wprefix = wave_id() ? fnarr[16 + wave_id() - 1] : 2u;          // all earlier waves
  uint32_t fex = (uint32_t)__shfl_up((int)incl, 1, 64);
  fex = lane_id() ? fn_compose(wprefix, fex) : wprefix;                          // composition of all earlier threads' functions
  const uint32_t fall = fnarr[16 + 15];
  __syncthreads();
  // c0 of the stretch governing my first position
  uint32_t c0 = fn_apply(fex, cy.cur_c0);
  uint32_t ps = exb ? exb - 1 : cy.cur_start;
  uint32_t cnt = 0, is_cnt = 0;
#pragma unroll
  for (int j = 0; j < UR_BPT; j++) {
    const uint32_t p = p0 + j;
    if (p < n) {
      if ((bmask >> j) & 1u) { c0 = fn_apply(stretch_fn(p - ps), c0); ps = p; }
      const uint32_t q = ps + c0;                             // first literal of the stretch
      const uint32_t rel = p >= q ? p - q : 0u;
      const uint32_t count_byte = ((p == ps) & c0) | ((p >= q) & ((rel % 5u) == 4u));
      is_cnt |= count_byte << j;
      cnt += count_byte ? (uint32_t)c[j + 1] : 1u;
    }
  }
  uint32_t tot;
  uint32_t off = cy.out_base + block_excl_sum<1024>(cnt, sm, tot);
  if (WRITE) {
    // The tile's bytes are one stretch of the output, a thread's a few of them at an odd address: they are put together in LDS (at the
    // stretch's own alignment) and leave as aligned 16-byte pieces; a tile of long runs that does not fit goes out byte by byte.
    uint8_t* dst = o + cy.out_base;
    const uint32_t al = (uint32_t)((uintptr_t)dst & 15u);
    const bool staged = tot <= UR_STAGE;
    uint32_t rel = off - cy.out_base;
    if (staged) {                                  // (uniform)
#pragma unroll
      for (int j = 0; j < UR_BPT; j++) {
        if (p0 + j < n) {
          if ((is_cnt >> j) & 1u) { const uint32_t k = c[j + 1]; const uint8_t v = c[j]; for (uint32_t q = 0; q < k; q++) stage[al + rel + q] = v; rel += k; }
          else stage[al + rel++] = c[j + 1];
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < UR_BPT; j++) {
        if (p0 + j < n) {
          if ((is_cnt >> j) & 1u) { const uint32_t k = c[j + 1]; const uint8_t v = c[j]; for (uint32_t q = 0; q < k; q++) dst[rel + q] = v; rel += k; }
          else dst[rel++] = c[j + 1];
        }
      }
    }
    if (staged) {
      __syncthreads();
      const uint32_t lead = min((16u - al) & 15u, tot), body = (tot - lead) >> 4, tail = lead + (body << 4);
      if (threadIdx.x < lead) dst[threadIdx.x] = stage[al + threadIdx.x];
      for (uint32_t i = threadIdx.x; i < body; i += 1024) reinterpret_cast<uint4*>(dst + lead)[i] = reinterpret_cast<const uint4*>(stage + al + lead)[i];
      if (threadIdx.x < tot - tail) dst[tail + threadIdx.x] = stage[al + tail + threadIdx.x];
    }
  }
  cy.out_base += tot;
  if (tile_last) { cy.cur_c0 = fn_apply(fall, cy.cur_c0); cy.cur_start = tile_last - 1; }
}
// The length pass, every tile of every block at once, in three launches (one workgroup per block walking its tiles front to back took
// 0.63 ms per 100 MB).  A tile depends on what lies in front of it through three things only: where the stretch that runs into it
// started (the last boundary in front), the carried bit c0 of that stretch, and the bytes out so far.  So: (1) every tile's last
// boundary; (2) with the last boundary in front of it, every tile's function on c0 and its byte count for BOTH values of c0;
// (3) per block, one wave chains the functions and sums the counts: the state carried INTO every tile, for the write pass.
// The RleCarry slot of a tile holds the intermediate values: pad = last boundary + 1 (0: none), cur_c0 = function, cur_start /
// out_base = bytes for c0 = 0 / 1.
__global__ __launch_bounds__(1024) void unrle1_bounds(const uint8_t* __restrict__ wbuf, const IbBlock* __restrict__ blocks, RleCarry* __restrict__ carry, uint32_t tiles_per_block) {
  __shared__ uint32_t sm[16];
  const IbBlock b = blocks[blockIdx.y];
  const uint32_t base = blockIdx.x * UR_TILE;
  if (base >= b.count) return;
  const uint8_t* w = wbuf + b.woff;
  const uint32_t p0 = base + threadIdx.x * UR_BPT;
  uint8_t c[UR_BPT + 1];
  ur_load(w, b.count, p0, c);
  uint32_t lastb = 0;
#pragma unroll
  for (int j = 0; j < UR_BPT; j++) { const uint32_t p = p0 + j; if (p < b.count && p > 0 && c[j + 1] != c[j]) lastb = p + 1; }
  const uint32_t im = block_incl_max<1024>(lastb, sm);
  if (threadIdx.x == 1023) carry[(size_t)blockIdx.y * tiles_per_block + blockIdx.x].pad = im;
}
__global__ __launch_bounds__(1024) void unrle1_sums(const uint8_t* __restrict__ wbuf, const IbBlock* __restrict__ blocks, RleCarry* __restrict__ carry, uint32_t tiles_per_block) {
  __shared__ unsigned long long sm64[16];
  __shared__ uint32_t sm[16];
  __shared__ uint32_t fnarr[32];
  __shared__ uint32_t posarr[1024];
  __shared__ uint32_t s_start;
  const IbBlock b = blocks[blockIdx.y];
  const uint32_t base = blockIdx.x * UR_TILE, n = b.count;
  if (base >= n) return;
  const uint8_t* w = wbuf + b.woff;
  RleCarry* row = carry + (size_t)blockIdx.y * tiles_per_block;
  if (threadIdx.x == 0) {                         // the last boundary in front of the tile: as a rule in the tile in front
    uint32_t st = 0;
    for (uint32_t t = blockIdx.x; t-- > 0;) { const uint32_t lb = row[t].pad; if (lb) { st = lb - 1; break; } }
    s_start = st;
  }
  const uint32_t p0 = base + threadIdx.x * UR_BPT;
  uint8_t c[UR_BPT + 1];
  ur_load(w, n, p0, c);
  uint32_t bmask = 0, lastb = 0;
#pragma unroll
  for (int j = 0; j < UR_BPT; j++) { const uint32_t p = p0 + j; if (p < n && p > 0 && c[j + 1] != c[j]) { bmask |= 1u << j; lastb = p + 1; } }
  const uint32_t im = block_incl_max<1024>(lastb, sm);
  posarr[threadIdx.x] = im;
  __syncthreads();
  const uint32_t exb = threadIdx.x ? posarr[threadIdx.x - 1] : 0u;
  const uint32_t cur_start = s_start;
  uint32_t f = 2u;   // identity
  {
    uint32_t ps = exb ? exb - 1 : cur_start;
#pragma unroll
    for (int j = 0; j < UR_BPT; j++) if ((bmask >> j) & 1u) { const uint32_t p = p0 + j; f = fn_compose(f, stretch_fn(p - ps)); ps = p; }
  }
  uint32_t incl = f;
  {
    const int lane = lane_id(), wv = wave_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t other = (uint32_t)__shfl_up((int)incl, d, 64);
      if (lane >= d) incl = fn_compose(other, incl);
    }
    if (lane == 63) fnarr[wv] = incl;
    __syncthreads();
    if (wv == 0) {
      uint32_t t = lane < 16 ? fnarr[lane] : 2u;
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) {
        const uint32_t other = (uint32_t)__shfl_up((int)t, d, 64);
        if (lane >= d) t = fn_compose(other, t);
      }
      if (lane < 16) fnarr[16 + lane] = t;
    }
    __syncthreads();
  }
  const uint32_t wprefix = wave_id() ? fnarr[16 + wave_id() - 1] : 2u;
  uint32_t fex = (uint32_t)__shfl_up((int)incl, 1, 64);
  fex = lane_id() ? fn_compose(wprefix, fex) : wprefix;
  const uint32_t fall = fnarr[16 + 15];
  unsigned long long both = 0;                    // bytes out of my positions for c0 = 0 (low half) and c0 = 1 (high half) at the tile's start
#pragma unroll
  for (int v = 0; v < 2; v++) {
    uint32_t c0 = fn_apply(fex, (uint32_t)v), ps = exb ? exb - 1 : cur_start, cnt = 0;
#pragma unroll
    for (int j = 0; j < UR_BPT; j++) {
      const uint32_t p = p0 + j;
      if (p < n) {
        if ((bmask >> j) & 1u) { c0 = fn_apply(stretch_fn(p - ps), c0); ps = p; }
        const uint32_t q = ps + c0;
        const uint32_t rel = p >= q ? p - q : 0u;
        const uint32_t count_byte = ((p == ps) & c0) | ((p >= q) & ((rel % 5u) == 4u));
        cnt += count_byte ? (uint32_t)c[j + 1] : 1u;
      }
    }
    both |= (unsigned long long)cnt << (32 * v);
  }
  unsigned long long tot;
  block_excl_sum<1024>(both, sm64, tot);
  if (threadIdx.x == 0) { RleCarry& e = row[blockIdx.x]; e.cur_c0 = fall; e.cur_start = (uint32_t)tot; e.out_base = (uint32_t)(tot >> 32); }
}
__global__ __launch_bounds__(64) void unrle1_carries(IbBlock* __restrict__ blocks, RleCarry* __restrict__ carry, uint32_t tiles_per_block) {
  const IbBlock b = blocks[blockIdx.x];
  RleCarry* row = carry + (size_t)blockIdx.x * tiles_per_block;
  const uint32_t nt = (b.count + UR_TILE - 1) / UR_TILE;
  const int lane = lane_id();
  uint32_t prevb = 0, c0 = 0, out = 0;            // last boundary + 1 / carried bit / bytes out in front of the chunk of 64 tiles
  for (uint32_t t0 = 0; t0 < nt; t0 += 64) {
    const uint32_t t = t0 + lane;
    RleCarry e{0u, 2u, 0u, 0u};                   // (behind the last tile: no bytes, the identity, no boundary)
    if (t < nt) e = row[t];
    const uint32_t im = wave_incl_max(e.pad);
    uint32_t exm = (uint32_t)__shfl_up((int)im, 1, 64); if (lane == 0) exm = 0;
    exm = exm > prevb ? exm : prevb;
    uint32_t incl = e.cur_c0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t other = (uint32_t)__shfl_up((int)incl, d, 64);
      if (lane >= d) incl = fn_compose(other, incl);
    }
    uint32_t fex = (uint32_t)__shfl_up((int)incl, 1, 64); if (lane == 0) fex = 2u;
    const uint32_t c0_in = fn_apply(fex, c0);
    const uint32_t bytes = c0_in ? e.out_base : e.cur_start;
    const uint32_t isum = wave_incl_sum(bytes);
    if (t < nt) row[t] = RleCarry{exm ? exm - 1 : 0u, c0_in, out + isum - bytes, 0u};
    const uint32_t last_m = (uint32_t)__builtin_amdgcn_readlane((int)im, 63);
    prevb = last_m > prevb ? last_m : prevb;
    c0 = fn_apply((uint32_t)__builtin_amdgcn_readlane((int)incl, 63), c0);
    out += (uint32_t)__builtin_amdgcn_readlane((int)isum, 63);
  }
  if (lane == 0) blocks[blockIdx.x].out_len = out;
}
// write pass: every tile of every block by itself
__global__ __launch_bounds__(1024) void unrle1_write(const uint8_t* __restrict__ wbuf, const IbBlock* __restrict__ blocks, const RleCarry* __restrict__ carry,
                                                     uint32_t tiles_per_block, uint8_t* __restrict__ out) {
  __shared__ uint32_t sm[16];
  __shared__ uint32_t fnarr[1024];
  __shared__ uint32_t posarr[1024];
  const IbBlock b = blocks[blockIdx.y];
  const uint32_t base = blockIdx.x * UR_TILE;
  if (base >= b.count) return;
  RleCarry cy = carry[(size_t)blockIdx.y * tiles_per_block + blockIdx.x];
  __shared__ __attribute__((aligned(16))) uint8_t stage[UR_STAGE + 16];
  unrle1_tile<true>(wbuf + b.woff, b.count, base, cy, out + b.out_off, sm, fnarr, posarr, stage);
}

__global__ void ib_make_crc_ranges(const IbBlock* __restrict__ blocks, uint32_t nblocks, RleBlock* __restrict__ ranges, uint32_t* __restrict__ nb_dev) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k == 0) *nb_dev = nblocks;
  if (k >= nblocks) return;
  RleBlock r; r.s = blocks[k].out_off; r.e = blocks[k].out_off + blocks[k].out_len; r.r_end = 0; r.Gr = 0; r.len = 0; r.base = 0;
  ranges[k] = r;
}

}  // namespace cjs

// ---------------------------------------------------------------- inverse sentinel BWT of a batch (used by BWTC.decompressFile)
namespace cjs {
// d_T holds the blocks back to back (block k at the sum of the earlier lengths), max_len bounds every block length.
// Blocks are processed in slabs of <= 65535 (grid.y of the per-block kernels) and <= 2^28 elements (bounded scratch).
static int ibwt_sentinel_slab(hipStream_t s, const uint8_t* d_T, uint32_t max_len, uint32_t nb, const uint32_t* lens, const uint32_t* pidx, uint8_t* d_out) {
  std::vector<IbBlock> chain(nb);
  uint64_t M64 = 0;
  for (uint32_t k = 0; k < nb; k++) {
    IbBlock& b = chain[k];
    b.tt = (uint64_t)(uintptr_t)(d_T + M64);
    b.count = lens[k]; b.orig = pidx[k]; b.off = b.woff = (uint32_t)M64; b.out_off = M64; b.out_len = lens[k]; b.crc = 0;
    M64 += lens[k];
  }
  if (M64 >= 0xFFFFF000ull) return CJS_E_UNSUPPORTED;
  const uint32_t M = (uint32_t)M64;
  std::vector<void*> to_free;
  auto dmalloc = [&](void** p, size_t bytes) { if (hipMalloc(p, bytes ? bytes : 4) != hipSuccess) return CJS_E_OUT_OF_MEMORY; to_free.push_back(*p); return 0; };
  auto cleanup = [&]() { for (void* p : to_free) (void)hipFree(p); };
  IbBlock* d_blocks = nullptr; uint32_t *d_key0 = nullptr, *d_key1 = nullptr, *d_val0 = nullptr, *d_val1 = nullptr;
  uint32_t *d_snext = nullptr, *d_ssteps = nullptr, *d_srank = nullptr; int32_t* d_err = nullptr;
  const uint32_t spl_stride = max_len / SPL + 4;
  BwtWork sw;
  int rc = dmalloc((void**)&d_blocks, sizeof(IbBlock) * nb);
  if (!rc) rc = dmalloc((void**)&d_key0, 4 * (size_t)M + 64); if (!rc) rc = dmalloc((void**)&d_key1, 4 * (size_t)M + 64);
  if (!rc) rc = dmalloc((void**)&d_val0, 4 * (size_t)M + 64); if (!rc) rc = dmalloc((void**)&d_val1, 4 * (size_t)M + 64);
  if (!rc) rc = dmalloc((void**)&d_snext, 4 * (size_t)nb * spl_stride); if (!rc) rc = dmalloc((void**)&d_ssteps, 4 * (size_t)nb * spl_stride);
  if (!rc) rc = dmalloc((void**)&d_srank, 4 * (size_t)nb * spl_stride); if (!rc) rc = dmalloc((void**)&d_err, 4 * (size_t)nb);
  const size_t T = ((size_t)M + RS_TILE - 1) / RS_TILE + 1;
  if (!rc) rc = dmalloc((void**)&sw.hist, BwtWork::hist_words(T) * 4); if (!rc) rc = dmalloc((void**)&sw.bintot, 256 * 4);
  sw.hist_tiles = (uint32_t)T; sw.bintot_segs = 1;
  if (!rc && hipMemcpyAsync(d_blocks, chain.data(), sizeof(IbBlock) * nb, hipMemcpyHostToDevice, s) != hipSuccess) rc = CJS_E_HIP;
  if (rc) { cleanup(); return rc; }
  hipLaunchKernelGGL(ib_make_keys, dim3(64, nb), dim3(256), 0, s, d_blocks, d_key0, d_val0, 0u);
  int cur = 0;
  int kbits = 8; { uint32_t x = nb - 1; while (x) { kbits++; x >>= 1; } }
  rc = radix_passes_public<uint32_t>(s, sw, d_key0, d_val0, d_key1, d_val1, cur, M, 0, kbits);
  if (rc) { cleanup(); return rc; }
  uint32_t* sval = cur ? d_val1 : d_val0;
  uint32_t* d_dbuf = cur ? d_key0 : d_key1;
  hipLaunchKernelGGL(ib_pack_sentinel, dim3(64, nb), dim3(256), 0, s, d_blocks, sval, d_dbuf);
  const uint32_t cpb = walk_chunks(max_len), wgrid = ((nb * cpb + 7u) >> 3) << 3;
  hipLaunchKernelGGL(ib_walk1, dim3(wgrid), dim3(WALK_T), WALK_LDS, s, d_dbuf, d_blocks, nb, cpb, spl_stride, d_snext, d_ssteps, 1, nullptr, nullptr);
  hipLaunchKernelGGL(ib_rank, dim3(nb), dim3(1024), 0, s, d_blocks, nb, spl_stride, d_snext, d_ssteps, d_srank, d_err);
  hipLaunchKernelGGL(ib_walk2, dim3(wgrid), dim3(WALK_T), WALK_LDS, s, d_dbuf, d_blocks, nb, cpb, spl_stride, d_srank, d_ssteps, d_out, 1, nullptr, nullptr);
  std::vector<int32_t> errs(nb);
  if (hipGetLastError() != hipSuccess || hipMemcpyAsync(errs.data(), d_err, 4 * (size_t)nb, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) rc = CJS_E_HIP;
  cleanup();
  if (rc) return rc;
  // The chain visits n distinct rows and then re-enters at row pidx (the step the reference computes last and never
  // uses, BWTC:1163-1165), so the last segment may overshoot; a chain that closes before n rows is corrupt input.
  for (uint32_t k = 0; k < nb; k++) if ((uint32_t)errs[k] < lens[k]) {
    if (getenv("CJS_DEBUG")) fprintf(stderr, "[cjs ibwt] block %u: chain covers %d of %u\n", k, errs[k], lens[k]);
    return CJS_E_DATA_ERROR;
  }
  return 0;
}
int ibwt_sentinel_run(hipStream_t s, const uint8_t* d_T, uint32_t max_len, uint32_t nb, const uint32_t* lens, const uint32_t* pidx, uint8_t* d_out) {
  uint64_t base = 0;
  for (uint32_t k0 = 0; k0 < nb;) {
    uint32_t k1 = k0; uint64_t el = 0;
    while (k1 < nb && k1 - k0 < 65535u && (k1 == k0 || el + lens[k1] <= (1ull << 28))) el += lens[k1++];
    CJS_TRY(ibwt_sentinel_slab(s, d_T + base, max_len, k1 - k0, lens + k0, pidx + k0, d_out + base));
    base += el; k0 = k1;
  }
  return 0;
}
}  // namespace cjs

// ---------------------------------------------------------------- host driver
// mode 0: Bunzip.decode (:1769-1796); mode 1: Bunzip.table (:1823-1863) -> (bit position, size) per block, no bytes;
// mode 2: Bunzip.decodeBlock (:1797-1818) -> the single block whose magic starts at `at_bit`.
//
// The job is cut into per-device shares (SURVEY §8e "Bzip2 decompress"; cjs_opts.n_devices / CJS_DEVICES, one host thread
// per share; several shares may sit on one GPU):
//   A  per share   upload its byte range (+ one worst-case block of overlap), magic scan, speculative decode of every
//                  candidate that STARTS in the share
//   -  host        chain walk 32 -> end(block 0) -> end(block 1) ... over all shares' candidates: stream CRC fold,
//                  multistream restarts (each stream keeps its own level, :1787-1792)
//   B  per share   inverse BWT of the chain blocks it decoded, in batches (bounded scratch), RLE1 length pass
//   -  host        exclusive prefix sum of the decoded lengths -> output offsets
//   C  per share   RLE1 expansion + block CRCs per batch, D2H straight to the final offsets
// No data moves between devices; the exchanged quantities are (end bit, count, crc) per candidate and a length per block.
namespace {

constexpr uint64_t DEC_BATCH_ELEMS = 1ull << 28;      // BWT bytes per inverse-BWT batch (scratch ~ 21 B each)
constexpr uint32_t DEC_BATCH_BLOCKS = 65535;          // grid.y of the per-block kernels

struct DecShare {
  int device = 0, rc = 0;
  hipStream_t s = nullptr;
  std::vector<void*> bufs;
  uint64_t lo = 0, hi = 0;            // candidates starting in bytes [lo, hi) are this share's
  uint64_t up_lo = 0, up_hi = 0;      // uploaded byte range
  const uint8_t* d_in = nullptr;      // addressed by absolute byte: d_in[b] is valid for up_lo <= b < up_hi
  std::vector<Cand> cands;            // sorted by bit
  std::vector<BlockOut> bos;
  uint8_t* d_tt = nullptr;            // decoded BWT bytes of a one-batch share, tt_stride per row (several batches: packed segments)
  std::vector<uint64_t> tt_ptr;       // per candidate: device address of its decoded bytes
  size_t cand_base = 0;               // index of cands[0] in the job's candidate list
  // chain part
  size_t c0 = 0, c1 = 0;              // chain blocks [c0, c1) were decoded here
  uint8_t* d_w = nullptr;             // pre-RLE1 bytes of those blocks, contiguous in chain order
  RleCarry* d_carry = nullptr; uint32_t carry_tiles = 0;      // per chain block and UR_TILE-byte tile: the RLE1 expansion state at the tile start
  std::vector<uint64_t> ebase;        // element offset of block c0+i inside d_w (size c1-c0+1)
  double ms_a = 0, ms_b = 0, ms_c = 0;
  char detail[96] = {0};            // error detail found by this share's worker thread (the detail text is per calling thread)
  int take(void** p, size_t bytes) { *p = DevPool::take(bytes); if (!*p) return (int)CJS_E_OUT_OF_MEMORY; bufs.push_back(*p); return 0; }
  void drop(void* p) { for (size_t i = 0; i < bufs.size(); i++) if (bufs[i] == p) { bufs.erase(bufs.begin() + (long)i); break; } DevPool::give(p); }
  void release() {
    if (hipSetDevice(device) != hipSuccess) return;
    if (s) (void)hipStreamSynchronize(s);
    for (void* p : bufs) DevPool::give(p);
    bufs.clear();
    if (s) (void)hipStreamDestroy(s);
    s = nullptr;
  }
};

struct DecJob {
  const uint8_t* in = nullptr; size_t n = 0;
  uint32_t tt_stride = 0;             // = dbuf size of the largest level in the input
  int mode = 0;
  std::vector<IbBlock> chain;         // all valid blocks in stream order (cand = index local to the decoding share)
  std::vector<uint64_t> chain_bits;
  std::vector<uint64_t> out_off;      // size chain.size()+1
  uint8_t* host = nullptr;            // final output (mode 0 / 2)
  bool timing = false;
};

double ms_since(std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); }

// ---- phase A
void dec_phase_a(DecJob* J, DecShare* S) {
  const auto T0 = std::chrono::steady_clock::now();
  if (hipSetDevice(S->device) != hipSuccess || hipStreamCreate(&S->s) != hipSuccess) { S->rc = CJS_E_HIP; return; }
  hipStream_t s = S->s;
  const size_t up_n = (size_t)(S->up_hi - S->up_lo);
  uint8_t* d_raw = nullptr; Cand* d_cand = nullptr; uint32_t* d_count = nullptr;
  uint32_t cand_cap = (uint32_t)((S->hi - S->lo) / 64 + 1024);      // grown to the exact count if a file of tiny streams has more
  int rc = S->take((void**)&d_raw, up_n + 256 + 16);
  if (!rc) rc = S->take((void**)&d_cand, sizeof(Cand) * cand_cap);
  if (!rc) rc = S->take((void**)&d_count, 64);
  if (rc) { S->rc = rc; return; }
  // keep the dword phase of the stream: the decoders fetch aligned big-endian words by absolute word index
  uint8_t* d_al = d_raw + (S->up_lo & 3u);
  S->d_in = d_al - S->up_lo;
  if (hipMemcpyAsync(d_al, J->in + S->up_lo, up_n, hipMemcpyHostToDevice, s) != hipSuccess || hipMemsetAsync(d_count, 0, 64, s) != hipSuccess) { S->rc = CJS_E_HIP; return; }
  auto launch_scan = [&]() {                                            // (slabs: a grid may not exceed 2^32 threads)
    for (uint64_t b0 = S->lo; b0 < S->hi; b0 += 1ull << 31) {
      const uint64_t b1 = std::min<uint64_t>(S->hi, b0 + (1ull << 31));
      hipLaunchKernelGGL(bz_magic_scan, dim3((unsigned)((b1 - b0 + 255) / 256)), dim3(256), 0, s, S->d_in, b0, b1, S->up_hi, d_cand, cand_cap, d_count);
    }
  };
  launch_scan();
  uint32_t ncand = 0;
  if (hipMemcpyAsync(&ncand, d_count, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { S->rc = CJS_E_HIP; return; }
  if (ncand > cand_cap) {                                             // more magics than planned for (many tiny member streams): scan again with room for all
    S->drop(d_cand);
    cand_cap = ncand;
    if ((rc = S->take((void**)&d_cand, sizeof(Cand) * cand_cap)) != 0) { S->rc = rc; return; }
    if (hipMemsetAsync(d_count, 0, 64, s) != hipSuccess) { S->rc = CJS_E_HIP; return; }
    launch_scan();
    if (hipMemcpyAsync(&ncand, d_count, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { S->rc = CJS_E_HIP; return; }
    if (ncand > cand_cap) { S->rc = CJS_E_HIP; return; }
  }
  S->cands.resize(ncand);
  if (ncand && hipMemcpy(S->cands.data(), d_cand, sizeof(Cand) * ncand, hipMemcpyDeviceToHost) != hipSuccess) { S->rc = CJS_E_HIP; return; }
  std::sort(S->cands.begin(), S->cands.end(), [](const Cand& a, const Cand& b) { return a.bit < b.bit; });
  uint32_t nrows = 0;                                              // only block candidates get a row of the decode buffer
  for (auto& c : S->cands) c.pad = c.kind == 0 ? nrows++ : 0u;
  if (ncand && hipMemcpy(d_cand, S->cands.data(), sizeof(Cand) * ncand, hipMemcpyHostToDevice) != hipSuccess) { S->rc = CJS_E_HIP; return; }
  S->bos.resize(ncand);
  S->tt_ptr.assign(ncand, 0ull);
  if (!ncand) { S->ms_a = ms_since(T0); return; }
  // Block decode in three stages (Huffman chain per block -> (rank, offset) ops; move-to-front of all 256-op tiles in parallel;
  // emit), over BATCHES of rows: a row of scratch is sized for a whole block of the file's largest level (~7 x tt_stride bytes),
  // whatever the candidate turns out to hold, so a file of very many tiny member streams (or one stuffed with block magics)
  // must not get a row per candidate at once.  One batch (the usual case: <= ~2000 level-9 rows in 16 GiB: a 2^30-byte stream has 1,194) keeps its decoded
  // rows where they are; with several batches each batch's decoded bytes are packed into a buffer of their exact size and
  // the scratch rows are used again.
  BlockOut* d_bo = nullptr;
  const uint32_t dsz = J->tt_stride;
  const uint32_t ops_stride = (dsz + 256u + 255u) & ~255u, tiles_per_row = ops_stride / MT_TILE;
  static const uint64_t budget = getenv("CJS_DEC_ROW_BYTES") ? strtoull(getenv("CJS_DEC_ROW_BYTES"), nullptr, 10) : (16ull << 30);      // (tests shrink it)
  const uint32_t sym_stride = dsz + 4096u;                              // symbols in front of the end of block: each emits a byte (but for forgotten runs), so <= dsz
  const uint32_t group_tiles = (std::min<uint32_t>(MAX_SELECTORS, sym_stride / GROUP_SYMS + 1u) + 255u) / 256u;
  const uint32_t sym_groups = (sym_stride + GROUP_SYMS - 1) / GROUP_SYMS;
  const uint64_t per_row = (uint64_t)dsz + 6ull * ops_stride + 256 + 4 + sizeof(RowTab) + MAX_SELECTORS + 4ull * (MAX_SELECTORS + 1) + 2ull * sym_groups * GROUP_SYMS;
  const uint32_t nr = std::max<uint32_t>(1u, (uint32_t)std::min<uint64_t>(std::min<uint64_t>(nrows ? nrows : 1u, 65535u), std::max<uint64_t>(1ull, budget / per_row)));      // (<= grid.y)
  const bool single = nrows <= nr;
  uint8_t *d_ttb = nullptr, *d_ops = nullptr, *d_l0 = nullptr, *d_pl = nullptr, *d_sel = nullptr; uint32_t *d_opoff = nullptr, *d_nops = nullptr, *d_gstart = nullptr;
  RowDst* d_gdst = nullptr; RowTab* d_tabs = nullptr; uint16_t* d_syms = nullptr;
  rc = S->take((void**)&d_ttb, (size_t)nr * dsz);
  if (!rc) rc = S->take((void**)&d_bo, sizeof(BlockOut) * ncand);
  if (!rc) rc = S->take((void**)&d_ops, (size_t)nr * ops_stride);
  if (!rc) rc = S->take((void**)&d_opoff, 4 * (size_t)nr * ops_stride);
  if (!rc) rc = S->take((void**)&d_l0, (size_t)nr * 256);
  if (!rc) rc = S->take((void**)&d_pl, (size_t)nr * ops_stride);
  if (!rc) rc = S->take((void**)&d_nops, 4 * (size_t)nr);
  if (!rc) rc = S->take((void**)&d_tabs, sizeof(RowTab) * (size_t)nr);
  if (!rc) rc = S->take((void**)&d_sel, (size_t)MAX_SELECTORS * nr);
  if (!rc) rc = S->take((void**)&d_gstart, 4 * (size_t)(MAX_SELECTORS + 1) * nr);
  if (!rc) rc = S->take((void**)&d_syms, 2 * (size_t)sym_groups * GROUP_SYMS * nr);
  if (!rc && !single) rc = S->take((void**)&d_gdst, sizeof(RowDst) * (size_t)nr);
  if (rc) { S->rc = rc; return; }
  if (single) S->d_tt = d_ttb;
  std::vector<RowDst> gdst(single ? 0 : nr);
  for (uint32_t c0 = 0; c0 < ncand;) {
    // candidates [c0, c1): at most nr block candidates (rows r0 .. r0 + rows)
    uint32_t c1 = c0, rows = 0, r0 = 0;
    while (c1 < ncand && (S->cands[c1].kind != 0 || rows < nr)) { if (S->cands[c1].kind == 0) { if (!rows) r0 = S->cands[c1].pad; rows++; } c1++; }
    const uint32_t nc = c1 - c0;
    hipLaunchKernelGGL(bz_chain, dim3(nc), dim3(CH_T), 0, s, S->d_in, S->up_hi, d_cand + c0, nc, dsz, d_tabs, d_sel, d_gstart, d_l0, d_bo + c0, r0);
    if (rows) hipLaunchKernelGGL(bz_group_syms, dim3(group_tiles, rows), dim3(256), 0, s, S->d_in, S->up_hi, d_tabs, d_sel, d_gstart, d_syms, sym_stride, sym_groups, 0u);
    hipLaunchKernelGGL(bz_sym_ops, dim3(nc), dim3(1024), 0, s, d_tabs, d_cand + c0, nc, d_syms, sym_groups, dsz, d_ops, d_opoff, ops_stride, d_nops, d_bo + c0, r0, S->up_hi * 8);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(S->bos.data() + c0, d_bo + c0, sizeof(BlockOut) * nc, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) { S->rc = CJS_E_HIP; return; }
    uint32_t maxc = 0; uint64_t packed = 0;
    for (uint32_t c = c0; c < c1; c++) if (S->cands[c].kind == 0 && !S->bos[c].err) { maxc = std::max(maxc, S->bos[c].count); packed += ((uint64_t)S->bos[c].count + 15u) & ~15ull; }
    if (rows) {
      const uint32_t tiles_used = std::min<uint32_t>(maxc / MT_TILE + 1u, tiles_per_row);
      // slabs of rows: grid.y <= 65535 and grid.x * grid.y * 256 threads < 2^32 (a larger launch is cut short without an error)
      const uint32_t slab = std::min<uint32_t>(65535u, std::max<uint32_t>(1u, (1u << 23) / tiles_used));
      for (uint32_t q0 = 0; q0 < rows; q0 += slab)
        hipLaunchKernelGGL(bz_mtf_tiles, dim3((tiles_used + 3u) / 4u, std::min(slab, rows - q0)), dim3(256), 0, s, d_ops, ops_stride, d_nops, d_pl, tiles_per_row, q0);
      {                                                                // (the symbols are spent: their rows hold the chunks' products)
        const uint32_t chunks = (tiles_used + MC_TILES - 1) / MC_TILES;
        const size_t cstride = 2 * (size_t)sym_groups * GROUP_SYMS;
        hipLaunchKernelGGL(bz_mtf_chunk_perm, dim3(chunks, rows), dim3(256), 0, s, d_nops, d_pl, tiles_per_row, reinterpret_cast<uint8_t*>(d_syms), cstride);
        hipLaunchKernelGGL(bz_mtf_compose, dim3(chunks, rows), dim3(256), 0, s, d_nops, d_l0, d_pl, tiles_per_row, reinterpret_cast<const uint8_t*>(d_syms), cstride);
      }
      for (uint32_t q0 = 0; q0 < rows; q0 += slab)
        hipLaunchKernelGGL(bz_mtf_emit, dim3(tiles_used, std::min(slab, rows - q0)), dim3(256), 0, s, d_ops, d_opoff, ops_stride, d_nops, d_l0, d_pl, tiles_per_row, q0, d_ttb, dsz);
      if (hipGetLastError() != hipSuccess) { S->rc = CJS_E_HIP; return; }
      if (single) {
        for (uint32_t c = c0; c < c1; c++) if (S->cands[c].kind == 0) S->tt_ptr[c] = (uint64_t)(uintptr_t)(d_ttb + (size_t)(S->cands[c].pad - r0) * dsz);
      } else {
        uint8_t* seg = nullptr;                                        // (kept until the share is released)
        if ((rc = S->take((void**)&seg, (size_t)packed + 16)) != 0) { S->rc = rc; return; }
        uint64_t at = 0;
        for (uint32_t c = c0; c < c1; c++) if (S->cands[c].kind == 0) {
          const uint32_t row = S->cands[c].pad - r0, cnt = S->bos[c].err ? 0u : S->bos[c].count;
          gdst[row] = RowDst{(uint64_t)(uintptr_t)(seg + at), cnt};
          S->tt_ptr[c] = (uint64_t)(uintptr_t)(seg + at);
          at += ((uint64_t)cnt + 15u) & ~15ull;
        }
        if (hipMemcpyAsync(d_gdst, gdst.data(), sizeof(RowDst) * (size_t)rows, hipMemcpyHostToDevice, s) != hipSuccess) { S->rc = CJS_E_HIP; return; }
        hipLaunchKernelGGL(bz_rows_pack, dim3(16, rows), dim3(256), 0, s, d_ttb, dsz, d_gdst);
        if (hipGetLastError() != hipSuccess) { S->rc = CJS_E_HIP; return; }
      }
      if (hipStreamSynchronize(s) != hipSuccess) { S->rc = CJS_E_HIP; return; }
    }
    c0 = c1;
  }
  S->drop(d_ops); S->drop(d_opoff); S->drop(d_l0); S->drop(d_pl); S->drop(d_nops); S->drop(d_tabs); S->drop(d_sel); S->drop(d_gstart); S->drop(d_syms);
  if (!single) { S->drop(d_ttb); S->drop(d_gdst); }
  if (getenv("CJS_DEBUG")) {
    uint64_t clk[8];
    if (hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_dec_clk), sizeof clk) == hipSuccess) {
      fprintf(stderr, "[cjs dec] candidate 0: header + tables %.1f us, group chain %.1f us for %llu groups\n", clk[5] / 100.0, clk[6] / 100.0, (unsigned long long)clk[7]);
    }
    fprintf(stderr, "[cjs dec] share on device %d: bytes [%llu, %llu) uploaded [%llu, %llu) = %zu B, %u candidates\n", S->device, (unsigned long long)S->lo,
            (unsigned long long)S->hi, (unsigned long long)S->up_lo, (unsigned long long)S->up_hi, up_n, ncand);
  }
  S->drop(d_cand); S->drop(d_count); S->drop(d_bo);
  S->ms_a = ms_since(T0);
}

// scratch of one inverse-BWT batch
struct IbScratch {
  IbBlock* d_blocks = nullptr; uint32_t *key0 = nullptr, *key1 = nullptr, *val0 = nullptr, *val1 = nullptr;
  uint32_t *snext = nullptr, *ssteps = nullptr, *srank = nullptr, *resume = nullptr; int32_t* d_err = nullptr;
  uint8_t* seg = nullptr;      // ib_walk1's kept bytes: SEG_CAP per splitter
  BwtWork sw;
};

// batches of the share's chain blocks: [b0, b1) with <= DEC_BATCH_ELEMS elements and <= DEC_BATCH_BLOCKS blocks
size_t dec_next_batch(const DecJob* J, size_t b0, size_t c1) {
  static const uint64_t max_el = getenv("CJS_DEC_BATCH_ELEMS") ? strtoull(getenv("CJS_DEC_BATCH_ELEMS"), nullptr, 10) : DEC_BATCH_ELEMS;   // (tests shrink it)
  uint64_t el = 0; size_t b = b0;
  while (b < c1 && b - b0 < DEC_BATCH_BLOCKS && (b == b0 || el + J->chain[b].count <= max_el)) { el += J->chain[b].count; b++; }
  return b;
}

// ---- phase B: inverse BWT (T vector by a stable radix pass, splitter list ranking, second walk) + RLE1 length pass
void dec_phase_b(DecJob* J, DecShare* S) {
  if (S->c1 <= S->c0) return;
  const auto T0 = std::chrono::steady_clock::now();
  if (hipSetDevice(S->device) != hipSuccess) { S->rc = CJS_E_HIP; return; }
  hipStream_t s = S->s;
  const size_t nbk = S->c1 - S->c0;
  S->ebase.assign(nbk + 1, 0);
  for (size_t i = 0; i < nbk; i++) S->ebase[i + 1] = S->ebase[i] + J->chain[S->c0 + i].count;
  int rc = S->take((void**)&S->d_w, (size_t)S->ebase[nbk] + 64);
  S->carry_tiles = (J->tt_stride + UR_TILE - 1) / UR_TILE;              // RLE1 state carried into every tile: written by phase B, read by phase C
  if (!rc) rc = S->take((void**)&S->d_carry, sizeof(RleCarry) * (size_t)nbk * S->carry_tiles);
  if (rc) { S->rc = rc; return; }
  for (size_t b0 = S->c0; b0 < S->c1 && !rc;) {
    const size_t b1 = dec_next_batch(J, b0, S->c1);
    const uint32_t nb = (uint32_t)(b1 - b0);
    const uint64_t e0 = S->ebase[b0 - S->c0], M64 = S->ebase[b1 - S->c0] - e0;
    if (M64 >= 0xFFFFF000ull) { rc = CJS_E_UNSUPPORTED; break; }     // a single block list beyond the batch limit cannot happen (count <= 900000)
    // Blocks of (nearly) one size -- a stream's are, but for its last -- get a slot range of that size each and ONE pass of the sort,
    // segment by segment; otherwise the block number is sorted on too (one or two more passes over everything).
    uint32_t maxc = 0;
    for (size_t k = b0; k < b1; k++) maxc = std::max(maxc, J->chain[k].count);
    const uint32_t seg_stride = (maxc + 3u) & ~3u;
    const uint32_t spl_stride = maxc / SPL + 4;                      // (of this batch: a file of very many small blocks must not pay for the largest level's)
    const bool strided = (uint64_t)nb * seg_stride <= M64 + M64 / 4 && (uint64_t)nb * seg_stride < 0xFFFFF000ull;
    const uint32_t M = strided ? nb * seg_stride : (uint32_t)M64;
    for (size_t k = b0; k < b1; k++) {
      J->chain[k].woff = (uint32_t)(S->ebase[k - S->c0] - e0);
      J->chain[k].off = strided ? (uint32_t)(k - b0) * seg_stride : J->chain[k].woff;
    }
    IbScratch q;
    rc = S->take((void**)&q.d_blocks, sizeof(IbBlock) * nb);
    if (!rc) rc = S->take((void**)&q.key0, 4 * (size_t)M + 64); if (!rc) rc = S->take((void**)&q.key1, 4 * (size_t)M + 64);
    if (!strided) { if (!rc) rc = S->take((void**)&q.val0, 4 * (size_t)M + 64); if (!rc) rc = S->take((void**)&q.val1, 4 * (size_t)M + 64); }
    if (!rc) rc = S->take((void**)&q.snext, 4 * (size_t)nb * spl_stride); if (!rc) rc = S->take((void**)&q.ssteps, 4 * (size_t)nb * spl_stride);
    if (!rc) rc = S->take((void**)&q.srank, 4 * (size_t)nb * spl_stride); if (!rc) rc = S->take((void**)&q.d_err, 4 * (size_t)nb);
    if (!rc) rc = S->take((void**)&q.resume, 4 * (size_t)nb * spl_stride);
    // (the first walk's kept bytes: without them -- one large block among very many tiny ones would ask for SEG_CAP x 14,066 bytes for
    // each -- the second walk does all the work, as it does for the sentinel form)
    const uint64_t seg_bytes = (uint64_t)nb * spl_stride * SEG_CAP;
    if (!rc && seg_bytes <= (8ull << 30) && S->take((void**)&q.seg, (size_t)seg_bytes) != 0) q.seg = nullptr;
    const uint32_t tps = (seg_stride + RS_TILE - 1) / RS_TILE;
    const size_t T = strided ? (size_t)nb * tps + 1 : ((size_t)M + RS_TILE - 1) / RS_TILE + 1;
    if (!rc) rc = S->take((void**)&q.sw.hist, BwtWork::hist_words(T) * 4); if (!rc) rc = S->take((void**)&q.sw.bintot, 256 * 4 * (size_t)(strided ? nb : 1u));
    q.sw.hist_tiles = (uint32_t)T; q.sw.bintot_segs = strided ? nb : 1u;
    if (!rc && hipMemcpyAsync(q.d_blocks, J->chain.data() + b0, sizeof(IbBlock) * nb, hipMemcpyHostToDevice, s) != hipSuccess) rc = CJS_E_HIP;
    if (!rc && hipMemsetAsync(q.d_err, 0, 4 * (size_t)nb, s) != hipSuccess) rc = CJS_E_HIP;
    if (rc) break;
    uint8_t* d_wb = S->d_w + e0;
    if (strided) hipLaunchKernelGGL(ib_make_keys_hist, dim3(tps, nb), dim3(256), 0, s, q.d_blocks, q.key0, seg_stride, tps, q.sw.hist);
    else hipLaunchKernelGGL(ib_make_keys, dim3(64, nb), dim3(256), 0, s, q.d_blocks, q.key0, q.val0, 0u);
    int cur = 0;
    if (strided) rc = radix_pass_segments_public<uint32_t>(s, q.sw, q.key0, q.val0, q.key1, q.val1, cur, nb, seg_stride, 0, 8, true, true);
    else {
      int kbits = 8; { uint32_t x = nb - 1; while (x) { kbits++; x >>= 1; } }
      rc = radix_passes_public<uint32_t>(s, q.sw, q.key0, q.val0, q.key1, q.val1, cur, M, 0, kbits);
    }
    if (rc) break;
    const uint32_t* sval = strided ? (cur ? q.key1 : q.key0) : (cur ? q.val1 : q.val0);      // (strided: the sorted keys carry the indices)
    uint32_t* d_dbuf = cur ? q.key0 : q.key1;                      // the buffer the sort is not sitting in
    hipLaunchKernelGGL(ib_pack, dim3(64, nb), dim3(256), 0, s, q.d_blocks, sval, d_dbuf, strided ? 1 : 0);
    const uint32_t cpb = walk_chunks(maxc), wgrid = ((nb * cpb + 7u) >> 3) << 3;
    hipLaunchKernelGGL(ib_walk1, dim3(wgrid), dim3(WALK_T), WALK_LDS, s, d_dbuf, q.d_blocks, nb, cpb, spl_stride, q.snext, q.ssteps, 0, q.seg, q.resume);
    hipLaunchKernelGGL(ib_rank, dim3(nb), dim3(1024), 0, s, q.d_blocks, nb, spl_stride, q.snext, q.ssteps, q.srank, q.d_err);
    // (placing inside the second walk, whose workgroups are few per CU, was no faster than the two launches: 0.49 vs 0.21 + 0.26 ms)
    if (q.seg) hipLaunchKernelGGL(ib_place, dim3((spl_stride + 255) / 256, nb), dim3(256), 0, s, q.d_blocks, spl_stride, q.srank, q.ssteps, q.seg, d_wb);
    hipLaunchKernelGGL(ib_walk2, dim3(wgrid), dim3(WALK_T), WALK_LDS, s, d_dbuf, q.d_blocks, nb, cpb, spl_stride, q.srank, q.ssteps, d_wb, 0, q.resume, q.seg);
    hipLaunchKernelGGL(ib_periodic_fill, dim3(32, nb), dim3(256), 0, s, q.d_blocks, q.d_err, d_wb);
    {
      RleCarry* cr = S->d_carry + (size_t)(b0 - S->c0) * S->carry_tiles;
      hipLaunchKernelGGL(unrle1_bounds, dim3(S->carry_tiles, nb), dim3(1024), 0, s, d_wb, q.d_blocks, cr, S->carry_tiles);
      hipLaunchKernelGGL(unrle1_sums, dim3(S->carry_tiles, nb), dim3(1024), 0, s, d_wb, q.d_blocks, cr, S->carry_tiles);
      hipLaunchKernelGGL(unrle1_carries, dim3(nb), dim3(64), 0, s, q.d_blocks, cr, S->carry_tiles);
    }
    std::vector<int32_t> errs(nb);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(J->chain.data() + b0, q.d_blocks, sizeof(IbBlock) * nb, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(errs.data(), q.d_err, 4 * (size_t)nb, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { rc = CJS_E_HIP; break; }
    for (uint32_t k = 0; k < nb; k++) if (errs[k] <= 0) rc = CJS_E_DATA_ERROR;      // cannot happen: the walk makes >= 1 step
    S->drop(q.d_blocks); S->drop(q.key0); S->drop(q.key1); if (q.val0) S->drop(q.val0); if (q.val1) S->drop(q.val1); S->drop(q.snext); S->drop(q.ssteps);
    S->drop(q.srank); S->drop(q.d_err); S->drop(q.resume); if (q.seg) S->drop(q.seg); S->drop(q.sw.hist); S->drop(q.sw.bintot);
    b0 = b1;
  }
  if (S->d_tt) { S->drop(S->d_tt); S->d_tt = nullptr; }
  S->rc = rc;
  S->ms_b = ms_since(T0);
}

// ---- phase C: RLE1 expansion to the final byte offsets, block CRC check, D2H
void dec_phase_c(DecJob* J, DecShare* S) {
  if (S->c1 <= S->c0) return;
  const auto T0 = std::chrono::steady_clock::now();
  if (hipSetDevice(S->device) != hipSuccess) { S->rc = CJS_E_HIP; return; }
  hipStream_t s = S->s;
  int rc = 0;
  for (size_t b0 = S->c0; b0 < S->c1 && !rc;) {
    const size_t b1 = dec_next_batch(J, b0, S->c1);
    const uint32_t nb = (uint32_t)(b1 - b0);
    const uint64_t e0 = S->ebase[b0 - S->c0], o0 = J->out_off[b0], obytes = J->out_off[b1] - o0;
    std::vector<IbBlock> blk(J->chain.begin() + (long)b0, J->chain.begin() + (long)b1);
    uint32_t need_segs = 1;
    for (uint32_t k = 0; k < nb; k++) {
      blk[k].off = blk[k].woff = (uint32_t)(S->ebase[b0 + k - S->c0] - e0);
      blk[k].out_off = J->out_off[b0 + k] - o0;                   // inside the batch's output buffer
      const uint32_t sg = (uint32_t)((blk[k].out_len + 16383) / 16384 + 1);
      if (sg > need_segs) need_segs = sg;
    }
    IbBlock* d_blocks = nullptr; uint8_t* d_out = nullptr; RleBlock* d_ranges = nullptr; uint32_t *d_nb = nullptr, *d_seg = nullptr, *d_crc = nullptr;
    rc = S->take((void**)&d_blocks, sizeof(IbBlock) * nb);
    if (!rc) rc = S->take((void**)&d_out, (size_t)obytes + 64);
    if (!rc) rc = S->take((void**)&d_ranges, sizeof(RleBlock) * nb);
    if (!rc) rc = S->take((void**)&d_nb, 64);
    if (!rc) rc = S->take((void**)&d_seg, 4 * (size_t)nb * need_segs);
    if (!rc) rc = S->take((void**)&d_crc, 4 * (size_t)nb);
    if (!rc && hipMemcpyAsync(d_blocks, blk.data(), sizeof(IbBlock) * nb, hipMemcpyHostToDevice, s) != hipSuccess) rc = CJS_E_HIP;
    if (rc) break;
    hipLaunchKernelGGL(unrle1_write, dim3(S->carry_tiles, nb), dim3(1024), 0, s, S->d_w + e0, d_blocks, S->d_carry + (size_t)(b0 - S->c0) * S->carry_tiles, S->carry_tiles, d_out);
    hipLaunchKernelGGL(ib_make_crc_ranges, dim3((nb + 63) / 64), dim3(64), 0, s, d_blocks, nb, d_ranges, d_nb);
    rc = crc_ranges(s, d_out, d_ranges, d_nb, nb, need_segs, d_seg, d_crc);
    std::vector<uint32_t> crcs(nb);
    if (!rc && hipMemcpyAsync(crcs.data(), d_crc, 4 * (size_t)nb, hipMemcpyDeviceToHost, s) != hipSuccess) rc = CJS_E_HIP;
    if (!rc && J->host && obytes && hipMemcpyAsync(J->host + o0, d_out, (size_t)obytes, hipMemcpyDeviceToHost, s) != hipSuccess) rc = CJS_E_HIP;
    if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = CJS_E_HIP;
    if (!rc) for (uint32_t k = 0; k < nb; k++) if (crcs[k] != blk[k].crc) {                    // Bad block CRC (:1756-1761)
      snprintf(S->detail, sizeof S->detail, "Bad block CRC (got %x expected %x)", crcs[k], blk[k].crc);
      if (getenv("CJS_DEBUG")) fprintf(stderr, "[cjs dec] block %zu: Bad block CRC (got %08x expected %08x) out_len %u\n", b0 + k, crcs[k], blk[k].crc, blk[k].out_len);
      rc = CJS_E_DATA_ERROR; break;
    }
    S->drop(d_blocks); S->drop(d_out); S->drop(d_ranges); S->drop(d_nb); S->drop(d_seg); S->drop(d_crc);
    b0 = b1;
  }
  S->rc = rc;
  S->ms_c = ms_since(T0);
}

template <typename F>
int for_each_share(std::vector<DecShare>& sh, DecJob* J, F fn) {
  // nothing may leave a worker thread (std::terminate): an exception of a phase becomes the share's return code
  auto guarded = [fn](DecJob* j, DecShare* s) {
    try { fn(j, s); }
    catch (const std::bad_alloc&) { s->rc = CJS_E_OUT_OF_MEMORY; }
    catch (...) { s->rc = CJS_E_HIP; }
  };
  if (sh.size() == 1) guarded(J, &sh[0]);
  else {
    struct JoinAll { std::vector<std::thread> th; ~JoinAll() { for (auto& t : th) if (t.joinable()) t.join(); } } workers;      // joined on every path
    for (auto& x : sh) workers.th.emplace_back(guarded, J, &x);
  }
  for (auto& x : sh) if (x.rc) { if (x.detail[0]) set_detail("%s", x.detail); return x.rc; }
  return 0;
}

}  // namespace

static int bunzip_core(const uint8_t* in, size_t n, int multistream, int mode, uint64_t at_bit, uint8_t** out, size_t* out_n,
                       uint64_t* tab_pos, uint32_t* tab_size, long tab_cap, long* tab_n, const cjs_opts* opts) {
  if (out) *out = nullptr;
  if (out_n) *out_n = 0;
  if (tab_n) *tab_n = 0;
  clear_detail();
  CJS_TRY(select_device(opts));
  // _start_bunzip (:1408-1427)
  if (n < 4 || in[0] != 'B' || in[1] != 'Z' || in[2] != 'h') { set_detail("bad magic"); return CJS_E_NOT_BZIP_DATA; }
  int level = in[3] - '0';
  if (level < 1 || level > 9) { set_detail("level out of range"); return CJS_E_NOT_BZIP_DATA; }
  int ndev = 0, dev0 = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || hipGetDevice(&dev0) != hipSuccess) return CJS_E_NO_DEVICE;

  DecJob J; J.in = in; J.n = n; J.mode = mode;
  J.timing = getenv("CJS_DEBUG") != nullptr;
  // The scratch rows are sized for the largest level any member stream can have: a multistream file may change level
  // between members (:1787-1792), so every byte-aligned "BZh<d>" followed by a block or end-of-stream magic counts.
  int max_level = level;
  if (multistream && mode != 2) {
    for (const uint8_t* p = in + 4; p + 10 <= in + n && (p = (const uint8_t*)memchr(p, 'B', (size_t)(in + n - 9 - p))) != nullptr; p++) {
      if (p[1] != 'Z' || p[2] != 'h' || p[3] < '1' || p[3] > '9') continue;
      uint64_t m = 0; for (int i = 0; i < 6; i++) m = (m << 8) | p[4 + i];
      if ((m == MAGIC_BLOCK || m == MAGIC_END) && p[3] - '0' > max_level) max_level = p[3] - '0';
    }
  }
  J.tt_stride = 100000u * (uint32_t)max_level;

  // shares: contiguous byte ranges, one per requested device slot
  uint32_t nsh = (opts && opts->struct_size >= sizeof(cjs_opts)) ? opts->n_devices : 0;
  if (const char* e = getenv("CJS_DEVICES")) nsh = (uint32_t)atoi(e);
  if (nsh < 1 || mode == 2) nsh = 1;
  if (nsh > 64) nsh = 64;
  if ((size_t)nsh * 65536 > n) nsh = (uint32_t)(n / 65536 ? n / 65536 : 1);     // tiny inputs: one share
  const uint64_t overlap = (uint64_t)J.tt_stride * 5 / 2 + 65536;               // one block at 20 bits per symbol + tables
  std::vector<DecShare> sh(nsh);
  for (uint32_t i = 0; i < nsh; i++) {
    DecShare& S = sh[i];
    S.device = nsh == 1 ? dev0 : (int)(i % (uint32_t)ndev);
    S.lo = (uint64_t)n * i / nsh; S.hi = (uint64_t)n * (i + 1) / nsh;
    S.up_lo = S.lo & ~(uint64_t)255;
    S.up_hi = std::min<uint64_t>(n, S.hi + overlap);
  }
  if (mode == 2) {                                                             // one block: upload from its byte on
    sh[0].lo = std::min<uint64_t>(at_bit >> 3, n); sh[0].hi = std::min<uint64_t>(n, sh[0].lo + 1);
    sh[0].up_lo = sh[0].lo & ~(uint64_t)255; sh[0].up_hi = std::min<uint64_t>(n, sh[0].hi + overlap);
  }
  const auto T0 = std::chrono::steady_clock::now();
  auto release_all = [&]() { for (auto& x : sh) x.release(); (void)hipSetDevice(dev0); };
  struct ReleaseGuard { decltype(release_all)& f; ~ReleaseGuard() { f(); } } release_guard{release_all};      // also when an exception unwinds (release is idempotent)
  int rc = for_each_share(sh, &J, dec_phase_a);
  if (rc) { release_all(); return rc; }
  const double ms_a = ms_since(T0);

  // ---- chain walk over all shares' candidates (Bunzip.decode :1776-1794)
  std::vector<uint64_t> cbit; std::vector<uint32_t> cshare, clocal;
  for (uint32_t i = 0; i < nsh; i++) {
    sh[i].cand_base = cbit.size();
    for (size_t k = 0; k < sh[i].cands.size(); k++) { cbit.push_back(sh[i].cands[k].bit); cshare.push_back(i); clocal.push_back((uint32_t)k); }
  }
  auto find = [&](uint64_t bit) -> long {
    const auto it = std::lower_bound(cbit.begin(), cbit.end(), bit);
    return (it != cbit.end() && *it == bit) ? (long)(it - cbit.begin()) : -1;
  };
  auto read_bits = [&](uint64_t bit, int k) -> uint64_t { uint64_t v = 0; for (int i = 0; i < k; i++) { const uint64_t b = bit + i; v = (v << 1) | ((b >> 3) < n ? (in[b >> 3] >> (7 - (b & 7))) & 1u : 0u); } return v; };
  std::vector<uint32_t> chain_share;
  uint32_t dbuf_size = 100000u * (uint32_t)level;                  // of the member stream being walked
  auto take_block = [&](long ci, uint64_t bitpos) -> int {
    const DecShare& S = sh[cshare[(size_t)ci]];
    const BlockOut& bo = S.bos[clocal[(size_t)ci]];
    if (getenv("CJS_DEBUG")) fprintf(stderr, "[cjs dec] block at bit %llu: err %d count %u orig %u crc %08x end %llu\n", (unsigned long long)bitpos, bo.err, bo.count, bo.orig, bo.crc, (unsigned long long)bo.end_bit);
    if (bo.err != CJS_E_OBSOLETE_INPUT && bo.orig > dbuf_size) { set_detail("initial position out of bounds"); return CJS_E_DATA_ERROR; }   // :1449-1450
    if (bo.err) return bo.err;
    if (bo.count > dbuf_size) return CJS_E_DATA_ERROR;             // decoded with the largest level's limit: this stream's is lower (:1647,1663)
    IbBlock ib; ib.tt = S.tt_ptr[clocal[(size_t)ci]]; ib.count = bo.count; ib.orig = bo.orig; ib.off = 0; ib.woff = 0; ib.out_off = 0; ib.out_len = 0; ib.crc = bo.crc;
    J.chain.push_back(ib); J.chain_bits.push_back(bitpos); chain_share.push_back(cshare[(size_t)ci]);
    return 0;
  };
  uint64_t pos = 32; uint32_t stream_crc = 0;
  if (mode == 2) {                                               // reader.seekBit(pos); _get_next_block() (:1803-1805)
    const long ci = find(at_bit);
    if (ci < 0) rc = CJS_E_NOT_BZIP_DATA;
    else if (sh[cshare[(size_t)ci]].cands[clocal[(size_t)ci]].kind == 0) rc = take_block(ci, at_bit);
  } else for (;;) {
    if ((pos + 7) / 8 >= n) break;                               // inputStream.eof() (:1777)
    const long ci = find(pos);
    if (ci < 0) { rc = CJS_E_NOT_BZIP_DATA; break; }             // h !== WHOLEPI (:1438)
    const DecShare& S = sh[cshare[(size_t)ci]];
    if (S.cands[clocal[(size_t)ci]].kind == 0) {
      rc = take_block(ci, pos);
      if (rc) break;
      const BlockOut& bo = S.bos[clocal[(size_t)ci]];
      stream_crc = bo.crc ^ ((stream_crc << 1) | (stream_crc >> 31));
      pos = bo.end_bit;
    } else {
      const uint32_t target = (uint32_t)read_bits(pos + 48, 32);
      pos += 80;
      if ((pos + 7) / 8 > n) pos = (uint64_t)n * 8;
      if (getenv("CJS_DEBUG")) fprintf(stderr, "[cjs dec] end of stream at bit %llu: stream crc %08x stored %08x\n", (unsigned long long)pos - 80, stream_crc, target);
      if (mode == 0 && target != stream_crc) {                   // Bunzip.table ignores the stream crc (:1852)
        set_detail("Bad stream CRC (got %x expected %x)", stream_crc, target);
        rc = CJS_E_DATA_ERROR; break;
      }
      const uint64_t byte = (pos + 7) / 8;
      if (multistream && byte < n) {                            // _start_bunzip again, byte aligned (:1787-1792)
        if (byte + 4 > n || in[byte] != 'B' || in[byte + 1] != 'Z' || in[byte + 2] != 'h') { set_detail("bad magic"); rc = CJS_E_NOT_BZIP_DATA; break; }
        const int lv = in[byte + 3] - '0';
        if (lv < 1 || lv > 9) { set_detail("level out of range"); rc = CJS_E_NOT_BZIP_DATA; break; }
        dbuf_size = 100000u * (uint32_t)lv;
        if (dbuf_size > J.tt_stride) { rc = CJS_E_UNSUPPORTED; break; }        // cannot happen: the pre-scan saw this header
        pos = (byte + 4) * 8; stream_crc = 0;
      } else break;
    }
  }
  // The reference decodes block after block and checks every block's CRC before it reads on (:1756-1761), so an error met
  // by the walk (bad stream CRC, damaged later block, broken chain) is reported only if every block in front of it
  // passes its own CRC check: keep it pending and run the rest of the pipeline, without output, over the chain so far.
  const int pending_rc = rc;
  char pending_detail[192];
  snprintf(pending_detail, sizeof pending_detail, "%s", cjs_last_error_detail());
  clear_detail();
  rc = 0;
  const size_t nb = J.chain.size();
  if (nb == 0) {
    release_all();
    if (pending_rc) { set_detail("%s", pending_detail); return pending_rc; }
    if (out) { *out = (uint8_t*)malloc(1); if (!*out) return CJS_E_OUT_OF_MEMORY; }
    return 0;
  }
  {  // the chain is increasing in bit position, so every share owns one contiguous run of it
    size_t k = 0;
    for (uint32_t i = 0; i < nsh; i++) { sh[i].c0 = k; while (k < nb && chain_share[k] == i) k++; sh[i].c1 = k; }
    if (k != nb) { release_all(); return CJS_E_DATA_ERROR; }      // a chain that runs backwards: corrupt input
  }
  const auto T1 = std::chrono::steady_clock::now();
  rc = for_each_share(sh, &J, dec_phase_b);
  if (rc) { release_all(); return rc; }
  const double ms_b = ms_since(T1);
  J.out_off.assign(nb + 1, 0);
  for (size_t k = 0; k < nb; k++) J.out_off[k + 1] = J.out_off[k] + J.chain[k].out_len;
  const uint64_t total = J.out_off[nb];
  if (out && !pending_rc) { J.host = (uint8_t*)HostPool::take(total ? (size_t)total : 1); if (!J.host) { release_all(); return CJS_E_OUT_OF_MEMORY; } }
  const auto T2 = std::chrono::steady_clock::now();
  rc = for_each_share(sh, &J, dec_phase_c);
  const double ms_c = ms_since(T2);
  release_all();
  if (J.timing) {
    fprintf(stderr, "[cjs dec] %u share(s): upload + magic scan + block decode %.2f ms, inverse BWT + RLE1 lengths %.2f ms, RLE1 + CRC + D2H %.2f ms\n", nsh, ms_a, ms_b, ms_c);
    for (uint32_t i = 0; i < nsh; i++) fprintf(stderr, "[cjs dec]   share %u (device %d): %zu candidates, blocks [%zu, %zu): %.2f / %.2f / %.2f ms\n", i, sh[i].device,
                                               sh[i].cands.size(), sh[i].c0, sh[i].c1, sh[i].ms_a, sh[i].ms_b, sh[i].ms_c);
  }
  if (rc) { HostPool::give(J.host); return rc; }
  if (pending_rc) { set_detail("%s", pending_detail); return pending_rc; }
  if (tab_n) {
    *tab_n = (long)nb;
    for (size_t k = 0; k < nb && (long)k < tab_cap; k++) { tab_pos[k] = J.chain_bits[k]; tab_size[k] = J.chain[k].out_len; }
  }
  if (out) { *out = J.host; *out_n = (size_t)total; }
  return 0;
}

extern "C" int cjs_bzip2_decompress(const uint8_t* in, size_t n, int multistream, uint8_t** out, size_t* out_n, const cjs_opts* opts) {
  if (!out || !out_n) return CJS_E_INVALID_ARG;
  CJS_GUARD_BEGIN
  return bunzip_core(in, n, multistream, 0, 0, out, out_n, nullptr, nullptr, 0, nullptr, opts);
  CJS_GUARD_END(CJS_E_OUT_OF_MEMORY, CJS_E_HIP)
}
extern "C" int cjs_bzip2_decompress_block(const uint8_t* in, size_t n, uint64_t bitpos, uint8_t** out, size_t* out_n, const cjs_opts* opts) {
  if (!out || !out_n) return CJS_E_INVALID_ARG;
  CJS_GUARD_BEGIN
  return bunzip_core(in, n, 0, 2, bitpos, out, out_n, nullptr, nullptr, 0, nullptr, opts);
  CJS_GUARD_END(CJS_E_OUT_OF_MEMORY, CJS_E_HIP)
}
extern "C" long cjs_bzip2_table(const uint8_t* in, size_t n, int multistream, uint64_t* bitpos, uint32_t* size, long cap, const cjs_opts* opts) {
  CJS_GUARD_BEGIN
  long nbk = 0;
  const int rc = bunzip_core(in, n, multistream, 1, 0, nullptr, nullptr, bitpos, size, cap, &nbk, opts);
  return rc ? (long)rc : nbk;
  CJS_GUARD_END((long)CJS_E_OUT_OF_MEMORY, (long)CJS_E_HIP)
}
