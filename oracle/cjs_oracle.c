/* cjs_oracle.c — CPU restatement (plain C99) of the compressjs Bzip2 / BWTC hot path.
 * TEST INFRASTRUCTURE ONLY — see cjs_oracle.h for the rules and the parity status (PINNED).
 * J/ = /root/reference/ ; "Bzip2:" = J/Bzip2_joined_.js ; "BWTC:" = J/BWTC_joined_.js.
 */
#include "cjs_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ byte buffer */
typedef struct { uint8_t *p; size_t n, cap; int oom; } buf_t;
static void buf_put(buf_t *b, int byte) {
  if (b->n == b->cap) {
    size_t nc = b->cap ? b->cap * 2 : 16384;          /* Bzip2:266 growable from 16 KiB */
    uint8_t *np = (uint8_t *)realloc(b->p, nc);
    if (!np) { b->oom = 1; return; }
    b->p = np; b->cap = nc;
  }
  b->p[b->n++] = (uint8_t)byte;
}
void cjs_oracle_free(void *p) { free(p); }

/* ------------------------------------------------------------------ CRC32 (Bzip2:1013-1079) */
static uint32_t crc_table[256];
static int crc_ready = 0;
static void crc_init(void) {
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t c = i << 24;
    for (int k = 0; k < 8; k++) c = (c & 0x80000000u) ? (c << 1) ^ 0x04c11db7u : (c << 1);
    crc_table[i] = c;
  }
  crc_ready = 1;
}
static inline uint32_t crc_step(uint32_t crc, uint8_t v) { return (crc << 8) ^ crc_table[((crc >> 24) ^ v) & 0xff]; }
uint32_t cjs_oracle_crc32(const uint8_t *p, size_t n) {
  if (!crc_ready) crc_init();
  uint32_t c = 0xffffffffu;
  for (size_t i = 0; i < n; i++) c = crc_step(c, p[i]);
  return ~c;
}

/* ------------------------------------------------------------------ SA-IS (Bzip2:496-857 computes
 * the same suffix array with Yuta Mori's sais; a suffix array is unique, so this independent
 * textbook formulation (Nong/Zhang/Chan 2009, explicit sentinel) yields identical SA/BWT). */
#define TGET(i) ((t[(i) >> 3] >> ((i) & 7)) & 1)
#define TSET(i, b) (t[(i) >> 3] = (uint8_t)((b) ? (t[(i) >> 3] | (1u << ((i) & 7))) : (t[(i) >> 3] & ~(1u << ((i) & 7)))))
#define ISLMS(i) ((i) > 0 && TGET(i) && !TGET((i) - 1))

static void sa_buckets(const int32_t *s, int32_t *bkt, int n, int K, int end) {
  for (int i = 0; i < K; i++) bkt[i] = 0;
  for (int i = 0; i < n; i++) bkt[s[i]]++;
  int sum = 0;
  for (int i = 0; i < K; i++) { sum += bkt[i]; bkt[i] = end ? sum : sum - bkt[i]; }
}
static void sa_induce_l(const uint8_t *t, int32_t *SA, const int32_t *s, int32_t *bkt, int n, int K) {
  sa_buckets(s, bkt, n, K, 0);
  for (int i = 0; i < n; i++) {
    int j = SA[i] - 1;
    if (j >= 0 && !TGET(j)) SA[bkt[s[j]]++] = j;
  }
}
static void sa_induce_s(const uint8_t *t, int32_t *SA, const int32_t *s, int32_t *bkt, int n, int K) {
  sa_buckets(s, bkt, n, K, 1);
  for (int i = n - 1; i >= 0; i--) {
    int j = SA[i] - 1;
    if (j >= 0 && TGET(j)) SA[--bkt[s[j]]] = j;
  }
}
/* s[0..n) with s[n-1] == 0 the unique smallest sentinel; K = alphabet size incl. sentinel */
static int sa_is(const int32_t *s, int32_t *SA, int n, int K) {
  uint8_t *t = (uint8_t *)calloc((size_t)n / 8 + 1, 1);
  int32_t *bkt = (int32_t *)malloc(sizeof(int32_t) * (size_t)K);
  if (!t || !bkt) { free(t); free(bkt); return -1; }
  int i, j;
  TSET(n - 1, 1);
  if (n >= 2) TSET(n - 2, 0);
  for (i = n - 3; i >= 0; i--) TSET(i, (s[i] < s[i + 1] || (s[i] == s[i + 1] && TGET(i + 1))) ? 1 : 0);
  sa_buckets(s, bkt, n, K, 1);
  for (i = 0; i < n; i++) SA[i] = -1;
  for (i = 1; i < n; i++) if (ISLMS(i)) SA[--bkt[s[i]]] = i;
  sa_induce_l(t, SA, s, bkt, n, K);
  sa_induce_s(t, SA, s, bkt, n, K);
  int n1 = 0;
  for (i = 0; i < n; i++) if (ISLMS(SA[i])) SA[n1++] = SA[i];
  for (i = n1; i < n; i++) SA[i] = -1;
  int name = 0, prev = -1;
  for (i = 0; i < n1; i++) {
    int pos = SA[i], diff = 0;
    for (int d = 0; d < n; d++) {
      if (prev == -1 || s[pos + d] != s[prev + d] || TGET(pos + d) != TGET(prev + d)) { diff = 1; break; }
      else if (d > 0 && (ISLMS(pos + d) || ISLMS(prev + d))) break;
    }
    if (diff) { name++; prev = pos; }
    SA[n1 + pos / 2] = name - 1;
  }
  for (i = n - 1, j = n - 1; i >= n1; i--) if (SA[i] >= 0) SA[j--] = SA[i];
  int32_t *SA1 = SA, *s1 = SA + n - n1;
  int rc = 0;
  if (name < n1) rc = sa_is(s1, SA1, n1, name);
  else for (i = 0; i < n1; i++) SA1[s1[i]] = i;
  if (rc == 0) {
    sa_buckets(s, bkt, n, K, 1);
    for (i = 1, j = 0; i < n; i++) if (ISLMS(i)) s1[j++] = i;
    for (i = 0; i < n1; i++) SA1[i] = s1[SA1[i]];
    for (i = n1; i < n; i++) SA[i] = -1;
    for (i = n1 - 1; i >= 0; i--) { j = SA[i]; SA[i] = -1; SA[--bkt[s[j]]] = j; }
    sa_induce_l(t, SA, s, bkt, n, K);
    sa_induce_s(t, SA, s, bkt, n, K);
  }
  free(t); free(bkt);
  return rc;
}
/* suffix array of bytes T[0..n) (shorter-is-smaller order, = BWT.suffixsort Bzip2:862-876) */
static int suffix_array_bytes(const uint8_t *T, int n, int32_t *SA /* n entries */, int doubled) {
  int m = doubled ? 2 * n : n;
  int32_t *s = (int32_t *)malloc(sizeof(int32_t) * ((size_t)m + 1));
  int32_t *sa = (int32_t *)malloc(sizeof(int32_t) * ((size_t)m + 1));
  if (!s || !sa) { free(s); free(sa); return -1; }
  for (int i = 0; i < m; i++) s[i] = (int32_t)T[i >= n ? i - n : i] + 1;
  s[m] = 0;
  int rc = sa_is(s, sa, m + 1, 257);
  if (rc == 0) {
    if (!doubled) memcpy(SA, sa + 1, sizeof(int32_t) * (size_t)n);
    else { int j = 0; for (int i = 1; i <= m; i++) if (sa[i] < n) SA[j++] = sa[i]; }
  }
  free(s); free(sa);
  return rc;
}
int cjs_oracle_suffix_array(const uint8_t *T, int n, int32_t *SA) {
  if (n <= 0) return 0;
  return suffix_array_bytes(T, n, SA, 0);
}
/* cyclic BWT (Bzip2:928-971): suffix array of T.T filtered to starts < n; pidx = row of start 0 */
int cjs_oracle_bwt_cyclic(const uint8_t *T, int n, uint8_t *U) {
  if (n <= 1) { if (n == 1) U[0] = T[0]; return 0; }
  int32_t *SA = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  if (!SA) return -1;
  if (suffix_array_bytes(T, n, SA, 1) != 0) { free(SA); return -1; }
  int pidx = 0;
  for (int j = 0; j < n; j++) {
    int s = SA[j];
    if (s == 0) pidx = j;
    U[j] = T[s == 0 ? n - 1 : s - 1];
  }
  free(SA);
  return pidx;
}
/* sentinel BWT (BWTC:1125-1145 -> SA_IS(isbwt) computeBWT BWTC:922-967):
 * U = T[n-1] . (T[SA[i]-1] for SA[i] != 0); returns (index of suffix 0)+1 */
int cjs_oracle_bwt_sentinel(const uint8_t *T, int n, uint8_t *U) {
  if (n <= 1) { if (n == 1) U[0] = T[0]; return n; }
  int32_t *SA = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  if (!SA) return -1;
  if (suffix_array_bytes(T, n, SA, 0) != 0) { free(SA); return -1; }
  int pidx = 0, j = 1;
  U[0] = T[n - 1];
  for (int i = 0; i < n; i++) {
    if (SA[i] == 0) pidx = i; else U[j++] = T[SA[i] - 1];
  }
  free(SA);
  return pidx + 1;
}

/* ------------------------------------------------------------------ HuffmanAllocator (Bzip2:1085-1301) */
static int ha_first(const int32_t *a, int len, int i, int nodes_to_move) {      /* Bzip2:1135-1156 */
  int limit = i, k = len - 2;
  while (i >= nodes_to_move && (a[i] % len) > limit) { k = i; i -= (limit - i + 1); }
  if (i < nodes_to_move - 1) i = nodes_to_move - 1;
  while (k > i + 1) {
    int mid = (i + k) >> 1;
    if ((a[mid] % len) > limit) k = mid; else i = mid;
  }
  return k;
}
static void ha_parent_pointers(int32_t *a, int len) {                           /* Bzip2:1162-1186 */
  a[0] += a[1];
  int head = 0, tail = 1, top = 2;
  for (; tail < len - 1; tail++) {
    int32_t w;
    if (top >= len || a[head] < a[top]) { w = a[head]; a[head++] = tail; }
    else w = a[top++];
    if (top >= len || (head < tail && a[head] < a[top])) { w += a[head]; a[head++] = tail + len; }
    else w += a[top++];
    a[tail] = w;
  }
}
static int ha_nodes_to_relocate(const int32_t *a, int len, int maxlen) {        /* Bzip2:1195-1204 */
  int cur = len - 2;
  for (int depth = 1; depth < maxlen - 1 && cur > 1; depth++) cur = ha_first(a, len, cur - 1, 0);
  return cur;
}
static void ha_lengths(int32_t *a, int len) {                                   /* Bzip2:1211-1226 */
  int first = len - 2, next = len - 1;
  for (int depth = 1, avail = 2; avail > 0; depth++) {
    int last = first;
    first = ha_first(a, len, last - 1, 0);
    for (int i = avail - (last - first); i > 0; i--) a[next--] = depth;
    avail = (last - first) << 1;
  }
}
static void ha_lengths_reloc(int32_t *a, int len, int nodes_to_move, int insert_depth) { /* Bzip2:1235-1264 */
  int first = len - 2, next = len - 1;
  int depth = (insert_depth == 1) ? 2 : 1;
  int left = (insert_depth == 1) ? nodes_to_move - 2 : nodes_to_move;
  for (int avail = depth << 1; avail > 0; depth++) {
    int last = first;
    first = (first <= nodes_to_move) ? first : ha_first(a, len, last - 1, nodes_to_move);
    int offset = 0;
    if (depth >= insert_depth) {
      offset = 1 << (depth - insert_depth);
      if (left < offset) offset = left;
    } else if (depth == insert_depth - 1) {
      offset = 1;
      if (a[first] == last) first++;
    }
    for (int i = avail - (last - first + offset); i > 0; i--) a[next--] = depth;
    left -= offset;
    avail = (last - first + offset) << 1;
  }
}
static int fls32(uint32_t v) { int r = 0; while (v) { r++; v >>= 1; } return r; }    /* Util.fls Bzip2:470-486 */
void cjs_oracle_huff_alloc(int32_t *a, int len, int maxlen) {                   /* Bzip2:1275-1298 */
  if (len == 2) { a[1] = 1; a[0] = 1; return; }
  if (len == 1) { a[0] = 1; return; }
  if (len <= 0) return;
  ha_parent_pointers(a, len);
  int reloc = ha_nodes_to_relocate(a, len, maxlen);
  if ((a[0] % len) >= reloc) ha_lengths(a, len);
  else ha_lengths_reloc(a, len, reloc, maxlen - fls32((uint32_t)(reloc - 1)));
}
static int cmp_i32(const void *x, const void *y) { int32_t a = *(const int32_t *)x, b = *(const int32_t *)y; return (a > b) - (a < b); }
void cjs_oracle_huff_lengths(const uint32_t *freq, int alphabet, uint8_t *len) { /* StaticHuffman Bzip2:1866-1894 */
  int32_t merged[260], sorted[260];
  for (int i = 0; i < alphabet; i++) merged[i] = (int32_t)((freq[i] << 9) | (uint32_t)i);
  qsort(merged, (size_t)alphabet, sizeof(int32_t), cmp_i32);      /* keys are distinct */
  for (int i = 0; i < alphabet; i++) sorted[i] = merged[i] >> 9;
  cjs_oracle_huff_alloc(sorted, alphabet, 20);
  for (int i = 0; i < alphabet; i++) len[merged[i] & 0x1ff] = (uint8_t)sorted[i];
}

/* ------------------------------------------------------------------ bit writer (BitStream Bzip2:109-166) */
typedef struct { buf_t *b; uint32_t acc; int nacc; } bitw_t;
static void bw_bits(bitw_t *w, int n, uint64_t v) {
  for (int i = n - 1; i >= 0; i--) {
    w->acc = (w->acc << 1) | (uint32_t)((v >> i) & 1);
    if (++w->nacc == 8) { buf_put(w->b, (int)(w->acc & 0xff)); w->acc = 0; w->nacc = 0; }
  }
}
static void bw_flush(bitw_t *w) { while (w->nacc) bw_bits(w, 1, 0); }

/* ------------------------------------------------------------------ readBlock: RLE1 + CRC (Bzip2:1954-1985) */
int cjs_oracle_rle1_block(const uint8_t *in, size_t n, size_t *cursor, uint8_t *block, int cap, uint32_t *crc_out) {
  if (!crc_ready) crc_init();
  int pos = 0, last = -1, run = 0;
  uint32_t crc = 0xffffffffu;
  size_t c = *cursor;
  while (pos < cap) {
    if (run == 4) { block[pos++] = 0; if (pos >= cap) break; }
    if (c >= n) break;
    int ch = in[c++];
    crc = crc_step(crc, (uint8_t)ch);
    if (ch != last) { last = ch; run = 1; }
    else {
      run++;
      if (run > 4) {
        if (run < 256) { block[pos - 1]++; continue; }
        run = 1;
      }
    }
    block[pos++] = (uint8_t)ch;
  }
  *cursor = c;
  *crc_out = ~crc;
  return pos;
}

/* ------------------------------------------------------------------ MTF + RLE2 (Bzip2:2064-2139) */
int cjs_oracle_mtf_rle2(const uint8_t *U, const uint8_t *block, int n, uint16_t *A, uint32_t *freq, int *alphabet_size) {
  uint8_t used[256], M[256];
  memset(used, 0, sizeof used);
  for (int i = 0; i < n; i++) used[block[i]] = 1;
  int asz = 0;
  for (int i = 0; i < 256; i++) if (used[i]) M[asz++] = (uint8_t)i;
  int eob = asz + 1, pos = 0;
  for (int i = 0; i <= eob; i++) freq[i] = 0;
  uint32_t run = 0;
#define EMIT(c) do { A[pos++] = (uint16_t)(c); freq[c]++; } while (0)
#define EMIT_RUN() do { while (run) { if (run & 1) { EMIT(0); run -= 1; } else { EMIT(1); run -= 2; } run >>= 1; } } while (0)
  for (int i = 0; i < n; i++) {
    uint8_t c = U[i];
    int j = 0;
    while (M[j] != c) j++;
    for (int k = j; k > 0; k--) M[k] = M[k - 1];
    M[0] = c;
    if (j == 0) run++;
    else { EMIT_RUN(); EMIT(j + 1); run = 0; }
  }
  EMIT_RUN();
  EMIT(eob);
#undef EMIT
#undef EMIT_RUN
  *alphabet_size = asz;
  return pos;
}

/* ------------------------------------------------------------------ Huffman groups (Bzip2:1989-2054, 2147-2163) */
static int group_cost(const uint8_t *len, const uint16_t *A, int off, int cnt) { /* Bzip2:1918-1924 */
  int c = 0;
  for (int i = 0; i < cnt; i++) c += len[A[off + i]];
  return c;
}
static void assign_selectors(uint8_t *sel, int nsel, const uint8_t *lens /*[6][258]*/, int ngroups, const uint16_t *A, int pos) {
  for (int g = 0; g < nsel; g++) {                                              /* Bzip2:1989-2004 */
    int off = g * 50, cnt = pos - off < 50 ? pos - off : 50;
    int best = 0, best_cost = group_cost(lens, A, off, cnt);
    for (int j = 1; j < ngroups; j++) {
      int c = group_cost(lens + j * 258, A, off, cnt);
      if (c < best_cost) { best = j; best_cost = c; }
    }
    sel[g] = (uint8_t)best;
  }
}
typedef struct { int32_t cost, index; } split_t;
static int cmp_split(const void *x, const void *y) {   /* stable order: cost asc, then index asc (Q16) */
  const split_t *a = (const split_t *)x, *b = (const split_t *)y;
  if (a->cost != b->cost) return (a->cost > b->cost) - (a->cost < b->cost);
  return (a->index > b->index) - (a->index < b->index);
}
int cjs_oracle_huff_groups(const uint16_t *A, int pos, int alphabet_size, uint8_t *sel, uint8_t *lens) {
  int asz2 = alphabet_size + 2;               /* RUNA, RUNB, ranks.., EOB (Q7) */
  int target = pos >= 2400 ? 6 : pos >= 1200 ? 5 : pos >= 600 ? 4 : pos >= 200 ? 3 : 2;   /* Bzip2:2150 */
  int nsel = (pos + 49) / 50;
  uint32_t freq[6][258];
  memset(freq, 0, sizeof freq);
  for (int i = 0; i < pos; i++) freq[0][A[i]]++;
  cjs_oracle_huff_lengths(freq[0], asz2, lens);                       /* global table Bzip2:2155 */
  for (int i = 0; i < asz2; i++) freq[1][i] = 1;
  cjs_oracle_huff_lengths(freq[1], asz2, lens + 258);                 /* flat table Bzip2:2156-2157 */
  int ng = 2;
  split_t *splits = (split_t *)malloc(sizeof(split_t) * (size_t)(nsel > 0 ? nsel : 1));
  while (ng < target) {                                               /* optimizeHuffmanGroups Bzip2:2012-2053 */
    assign_selectors(sel, nsel, lens, ng, A, pos);
    int counts[6] = {0, 0, 0, 0, 0, 0};
    for (int g = 0; g < nsel; g++) counts[sel[g]]++;
    int which = 0;
    for (int j = 1; j < ng; j++) if (counts[j] > counts[which]) which = j;   /* indexOf(max): first */
    int ns = 0;
    for (int g = 0; g < nsel; g++) {
      if (sel[g] != which) continue;
      int off = g * 50, cnt = pos - off < 50 ? pos - off : 50;
      splits[ns].index = g;
      splits[ns].cost = group_cost(lens + which * 258, A, off, cnt);
      ns++;
    }
    qsort(splits, (size_t)ns, sizeof(split_t), cmp_split);
    for (int i = ns >> 1; i < ns; i++) sel[splits[i].index] = (uint8_t)ng;
    ng++;
    memset(freq, 0, sizeof freq);
    for (int i = 0; i < pos; i++) freq[sel[i / 50]][A[i]]++;
    for (int j = 0; j < ng; j++) cjs_oracle_huff_lengths(freq[j], asz2, lens + j * 258);
  }
  free(splits);
  assign_selectors(sel, nsel, lens, ng, A, pos);                      /* Bzip2:2163 */
  return ng;
}

/* ------------------------------------------------------------------ compressBlock (Bzip2:2056-2196) */
static int compress_block(const uint8_t *block, int n, bitw_t *w) {
  uint8_t *U = (uint8_t *)malloc((size_t)n);
  uint16_t *A = (uint16_t *)malloc(sizeof(uint16_t) * ((size_t)n + 1));
  uint8_t *sel = (uint8_t *)malloc((size_t)n / 50 + 2);
  if (!U || !A || !sel) { free(U); free(A); free(sel); return CJSO_OUT_OF_MEMORY; }
  int pidx = cjs_oracle_bwt_cyclic(block, n, U);
  if (pidx < 0) { free(U); free(A); free(sel); return CJSO_OUT_OF_MEMORY; }
  bw_bits(w, 1, 0);
  bw_bits(w, 24, (uint64_t)pidx);
  uint8_t used[256]; memset(used, 0, sizeof used);
  for (int i = 0; i < n; i++) used[block[i]] = 1;
  for (int i = 0; i < 16; i++) { int any = 0; for (int j = 0; j < 16; j++) any |= used[i * 16 + j]; bw_bits(w, 1, (uint64_t)any); }
  for (int i = 0; i < 16; i++) {
    int any = 0; for (int j = 0; j < 16; j++) any |= used[i * 16 + j];
    if (any) for (int j = 0; j < 16; j++) bw_bits(w, 1, used[i * 16 + j]);
  }
  uint32_t freq[258]; int asz;
  int pos = cjs_oracle_mtf_rle2(U, block, n, A, freq, &asz);
  uint8_t lens[6 * 258];
  int ng = cjs_oracle_huff_groups(A, pos, asz, sel, lens);
  int nsel = (pos + 49) / 50, asz2 = asz + 2;
  bw_bits(w, 3, (uint64_t)ng);
  bw_bits(w, 15, (uint64_t)nsel);
  uint8_t M[6];
  for (int i = 0; i < ng; i++) M[i] = (uint8_t)i;
  for (int g = 0; g < nsel; g++) {                                   /* Bzip2:2171-2182 */
    int j = 0; while (M[j] != sel[g]) j++;
    for (int k = j; k > 0; k--) M[k] = M[k - 1];
    M[0] = sel[g];
    for (; j > 0; j--) bw_bits(w, 1, 1);
    bw_bits(w, 1, 0);
  }
  uint32_t code[6][258];
  for (int t = 0; t < ng; t++) {
    const uint8_t *L = lens + t * 258;
    int cur = L[0];                                                  /* emit Bzip2:1926-1947 */
    bw_bits(w, 5, (uint64_t)cur);
    for (int i = 0; i < asz2; i++) {
      int l = L[i];
      while (cur < l) { bw_bits(w, 2, 2); cur++; }
      while (cur > l) { bw_bits(w, 2, 3); cur--; }
      bw_bits(w, 1, 0);
    }
    uint32_t c = 0; int prev = 0;                                    /* computeCanonical Bzip2:1896-1916 */
    for (int l = 0; l <= 20; l++)
      for (int s = 0; s < asz2; s++)
        if (L[s] == l) { c <<= (l - prev); code[t][s] = c++; prev = l; }
  }
  for (int i = 0; i < pos; i++) {                                    /* Bzip2:2189-2194 */
    int t = sel[i / 50];
    bw_bits(w, lens[t * 258 + A[i]], code[t][A[i]]);
  }
  free(U); free(A); free(sel);
  return w->b->oom ? CJSO_OUT_OF_MEMORY : 0;
}

/* Block loop of Bzip2.compressFile (Bzip2:2199-2249).  framed: 'BZh'+level header and trailer are written and
 * all blocks are emitted; otherwise only blocks [first, first+count) are emitted as a bare bit string that
 * starts at bit 0 (what one rank of a sharded job produces).  crcs (optional) receives every block's CRC. */
static int bzip2_blocks(const uint8_t *in, size_t n, int level, int framed, long first, long count, buf_t *b,
                        uint64_t *out_bits, uint32_t *crcs, long crc_cap, long *total_blocks) {
  if (level < 1 || level > 9) return CJSO_BAD_LEVEL;                 /* Bzip2:2208 */
  int block_size = level * 100000 - 19;                              /* Bzip2:2212-2220 */
  bitw_t w = {b, 0, 0};
  if (framed) { buf_put(b, 'B'); buf_put(b, 'Z'); buf_put(b, 'h'); buf_put(b, '0' + level); }
  uint8_t *block = (uint8_t *)malloc((size_t)block_size);
  if (!block) return CJSO_OUT_OF_MEMORY;
  uint32_t stream_crc = 0;
  size_t cursor = 0;
  int length, rc = 0;
  long k = 0;
  do {                                                               /* Bzip2:2233-2242 */
    uint32_t crc;
    length = cjs_oracle_rle1_block(in, n, &cursor, block, block_size, &crc);
    if (length > 0) {
      stream_crc = ((stream_crc << 1) | (stream_crc >> 31)) ^ crc;
      if (crcs && k < crc_cap) crcs[k] = crc;
      if (framed || (k >= first && (count < 0 || k < first + count))) {
        bw_bits(&w, 48, 0x314159265359ull);
        bw_bits(&w, 32, crc);
        rc = compress_block(block, length, &w);
        if (rc) break;
      }
      k++;
    }
  } while (length == block_size);
  free(block);
  if (rc) return rc;
  if (total_blocks) *total_blocks = k;
  if (framed) { bw_bits(&w, 48, 0x177245385090ull); bw_bits(&w, 32, stream_crc); }
  if (out_bits) *out_bits = (uint64_t)b->n * 8 + (uint64_t)w.nacc;
  bw_flush(&w);
  return b->oom ? CJSO_OUT_OF_MEMORY : 0;
}
int cjs_oracle_bzip2_compress(const uint8_t *in, size_t n, int level, uint8_t **out, size_t *out_n) {
  buf_t b = {0, 0, 0, 0};
  int rc = bzip2_blocks(in, n, level, 1, 0, -1, &b, 0, 0, 0, 0);
  if (rc) { free(b.p); return rc; }
  *out = b.p; *out_n = b.n;
  return 0;
}
int cjs_oracle_bzip2_compress_range(const uint8_t *in, size_t n, int level, long first, long count, uint8_t **out,
                                    uint64_t *out_bits, uint32_t *crcs, long crc_cap, long *total_blocks) {
  buf_t b = {0, 0, 0, 0};
  int rc = bzip2_blocks(in, n, level, 0, first, count, &b, out_bits, crcs, crc_cap, total_blocks);
  if (rc) { free(b.p); return rc; }
  *out = b.p ? b.p : (uint8_t *)malloc(1);
  return 0;
}

/* ------------------------------------------------------------------ Bunzip (Bzip2:1393-1863) */
typedef struct { const uint8_t *p; size_t n; uint64_t bit; } bitr_t;      /* bits past EOF read as 0 (Bzip2:149) */
static uint64_t br_bits(bitr_t *r, int n) {
  uint64_t v = 0;
  for (int i = 0; i < n; i++) {
    size_t byte = (size_t)(r->bit >> 3);
    int b = 0;
    if (byte < r->n) { b = (r->p[byte] >> (7 - (int)(r->bit & 7))) & 1; r->bit++; }
    /* past EOF: readBit returns EOF without consuming; position stays at the end */
    v = (v << 1) | (uint64_t)b;
  }
  return v;
}
static size_t br_bytes_consumed(const bitr_t *r) { return (size_t)((r->bit + 7) >> 3); }

typedef struct {
  uint16_t permute[258]; uint32_t limit[22]; uint32_t base[21]; int min_len, max_len;
} hgroup_t;

typedef struct {
  bitr_t r; int dbuf_size; uint32_t *dbuf; uint32_t stream_crc, target_block_crc;
  /* write state */
  uint32_t write_pos; int write_current; int write_count; int write_run;
} bunzip_t;

static int bunzip_start(bunzip_t *bz, const uint8_t *p, size_t n, size_t at) {   /* Bzip2:1408-1427 */
  if (at + 4 > n || p[at] != 'B' || p[at + 1] != 'Z' || p[at + 2] != 'h') return CJSO_NOT_BZIP_DATA;
  int level = p[at + 3] - '0';
  if (level < 1 || level > 9) return CJSO_NOT_BZIP_DATA;
  bz->r.p = p; bz->r.n = n; bz->r.bit = (uint64_t)(at + 4) * 8;
  bz->dbuf_size = 100000 * level;
  bz->stream_crc = 0;
  return 0;
}
/* returns 1 = block ready, 0 = end-of-stream marker, <0 error */
static int bunzip_next_block(bunzip_t *bz) {                                      /* Bzip2:1428-1709 */
  bitr_t *r = &bz->r;
  uint64_t h = br_bits(r, 48);
  if (h == 0x177245385090ull) return 0;
  if (h != 0x314159265359ull) return CJSO_NOT_BZIP_DATA;
  bz->target_block_crc = (uint32_t)br_bits(r, 32);
  bz->stream_crc = bz->target_block_crc ^ ((bz->stream_crc << 1) | (bz->stream_crc >> 31));
  if (br_bits(r, 1)) return CJSO_OBSOLETE_INPUT;
  uint32_t orig = (uint32_t)br_bits(r, 24);
  if ((int64_t)orig > bz->dbuf_size) return CJSO_DATA_ERROR;
  uint32_t t = (uint32_t)br_bits(r, 16);
  uint8_t sym_to_byte[256]; int sym_total = 0;
  memset(sym_to_byte, 0, sizeof sym_to_byte);
  for (int i = 0; i < 16; i++) if (t & (1u << (15 - i))) {
    uint32_t k = (uint32_t)br_bits(r, 16);
    for (int j = 0; j < 16; j++) if (k & (1u << (15 - j))) sym_to_byte[sym_total++] = (uint8_t)(i * 16 + j);
  }
  int group_count = (int)br_bits(r, 3);
  if (group_count < 2 || group_count > 6) return CJSO_DATA_ERROR;
  int n_sel = (int)br_bits(r, 15);
  if (n_sel == 0) return CJSO_DATA_ERROR;
  uint8_t mtf[256]; memset(mtf, 0, sizeof mtf);
  for (int i = 0; i < group_count; i++) mtf[i] = (uint8_t)i;
  uint8_t *selectors = (uint8_t *)malloc((size_t)n_sel);
  if (!selectors) return CJSO_OUT_OF_MEMORY;
  for (int i = 0; i < n_sel; i++) {
    int j;
    for (j = 0; br_bits(r, 1); j++) if (j >= group_count) { free(selectors); return CJSO_DATA_ERROR; }
    uint8_t v = mtf[j];
    for (int k = j; k > 0; k--) mtf[k] = mtf[k - 1];
    mtf[0] = v;
    selectors[i] = v;
  }
  int sym_count = sym_total + 2;
  hgroup_t groups[6];
  for (int j = 0; j < group_count; j++) {
    uint8_t length[258]; uint16_t temp[21];
    int tt = (int)br_bits(r, 5);
    for (int i = 0; i < sym_count; i++) {
      for (;;) {
        if (tt < 1 || tt > 20) { free(selectors); return CJSO_DATA_ERROR; }
        if (!br_bits(r, 1)) break;
        if (!br_bits(r, 1)) tt++; else tt--;
      }
      length[i] = (uint8_t)tt;
    }
    int min_len = length[0], max_len = length[0];
    for (int i = 1; i < sym_count; i++) {
      if (length[i] > max_len) max_len = length[i];
      else if (length[i] < min_len) min_len = length[i];
    }
    hgroup_t *g = &groups[j];
    memset(g, 0, sizeof *g);
    g->min_len = min_len; g->max_len = max_len;
    int pp = 0;
    memset(temp, 0, sizeof temp);
    for (int i = min_len; i <= max_len; i++)
      for (int s = 0; s < sym_count; s++) if (length[s] == i) g->permute[pp++] = (uint16_t)s;
    for (int i = 0; i < sym_count; i++) temp[length[i]]++;
    int64_t p2 = 0, t2 = 0;
    for (int i = min_len; i < max_len; i++) {
      p2 += temp[i];
      g->limit[i] = (uint32_t)(p2 - 1);
      p2 <<= 1;
      t2 += temp[i];
      g->base[i + 1] = (uint32_t)(p2 - t2);
    }
    g->limit[max_len] = (uint32_t)(p2 + temp[max_len] - 1);
    g->base[min_len] = 0;
  }
  uint32_t byte_count[256]; memset(byte_count, 0, sizeof byte_count);
  for (int i = 0; i < 256; i++) mtf[i] = (uint8_t)i;
  int32_t run_pos = 0; int64_t run_t = 0;
  int dbuf_count = 0, selector = 0, sym_left = 0;
  uint32_t *dbuf = bz->dbuf;
  hgroup_t *hg = 0;
  int rc = 0;
  for (;;) {
    if (!(sym_left--)) {
      sym_left = 49;
      if (selector >= n_sel) { rc = CJSO_DATA_ERROR; break; }
      int s = selectors[selector++];
      if (s >= group_count) { rc = CJSO_DATA_ERROR; break; }
      hg = &groups[s];
    }
    int i = hg->min_len;
    int64_t j = (int64_t)br_bits(r, i);
    for (;; i++) {
      if (i > hg->max_len) { rc = CJSO_DATA_ERROR; break; }
      if (j <= (int64_t)hg->limit[i]) break;
      j = (j << 1) | (int64_t)br_bits(r, 1);
    }
    if (rc) break;
    j -= (int64_t)hg->base[i];
    if (j < 0 || j >= 258) { rc = CJSO_DATA_ERROR; break; }
    int next_sym = hg->permute[j];
    if (next_sym == 0 || next_sym == 1) {
      if (!run_pos) { run_pos = 1; run_t = 0; }
      run_t += (next_sym == 0) ? (int64_t)run_pos : 2 * (int64_t)run_pos;
      run_pos = (int32_t)((uint32_t)run_pos << 1);
      continue;
    }
    if (run_pos) {
      run_pos = 0;
      if ((int64_t)dbuf_count + run_t > bz->dbuf_size) { rc = CJSO_DATA_ERROR; break; }
      uint8_t uc = sym_to_byte[mtf[0]];
      byte_count[uc] += (uint32_t)run_t;
      while (run_t-- > 0) dbuf[dbuf_count++] = uc;
    }
    if (next_sym > sym_total) break;
    if (dbuf_count >= bz->dbuf_size) { rc = CJSO_DATA_ERROR; break; }
    int k = next_sym - 1;
    uint8_t v = mtf[k];
    for (; k > 0; k--) mtf[k] = mtf[k - 1];
    mtf[0] = v;
    uint8_t uc = sym_to_byte[v];
    byte_count[uc]++;
    dbuf[dbuf_count++] = uc;
  }
  free(selectors);
  if (rc) return rc;
  if ((int64_t)orig >= dbuf_count) return CJSO_DATA_ERROR;
  uint32_t sum = 0;
  for (int i = 0; i < 256; i++) { uint32_t k = sum + byte_count[i]; byte_count[i] = sum; sum = k; }
  for (int i = 0; i < dbuf_count; i++) {
    uint8_t uc = (uint8_t)(dbuf[i] & 0xff);
    dbuf[byte_count[uc]] |= ((uint32_t)i << 8);
    byte_count[uc]++;
  }
  uint32_t pos = 0; int current = 0, run = 0;
  if (dbuf_count) { pos = dbuf[orig]; current = (int)(pos & 0xff); pos >>= 8; run = -1; }
  bz->write_pos = pos; bz->write_current = current; bz->write_count = dbuf_count; bz->write_run = run;
  return 1;
}
/* Bzip2:1716-1763; out may be NULL (table mode) -> only counts */
static int bunzip_read(bunzip_t *bz, buf_t *out, size_t *count) {
  if (!crc_ready) crc_init();
  uint32_t pos = bz->write_pos; int current = bz->write_current, run = bz->write_run;
  int left = bz->write_count;
  uint32_t crc = 0xffffffffu;
  const uint32_t *dbuf = bz->dbuf;
  while (left) {
    left--;
    int previous = current;
    pos = dbuf[pos];
    current = (int)(pos & 0xff);
    pos >>= 8;
    int copies, outbyte;
    if (run++ == 3) { copies = current; outbyte = previous; current = -1; }
    else { copies = 1; outbyte = current; }
    for (int c = 0; c < copies; c++) {
      crc = crc_step(crc, (uint8_t)outbyte);
      if (out) buf_put(out, outbyte);
    }
    *count += (size_t)copies;
    if (current != previous) run = 0;
  }
  bz->write_count = 0;
  if (~crc != bz->target_block_crc) return CJSO_DATA_ERROR;
  return 0;
}
static int bunzip_decode(const uint8_t *in, size_t n, int multistream, buf_t *out,
                         uint64_t *tab_pos, uint32_t *tab_size, long tab_cap, long *tab_n) {
  bunzip_t bz; memset(&bz, 0, sizeof bz);
  int rc = bunzip_start(&bz, in, n, 0);
  if (rc) return rc;
  bz.dbuf = (uint32_t *)malloc(sizeof(uint32_t) * 900000u);
  if (!bz.dbuf) return CJSO_OUT_OF_MEMORY;
  long nblocks = 0;
  for (;;) {                                                         /* Bzip2:1776-1794, 1842-1862 */
    if (br_bytes_consumed(&bz.r) >= n) break;
    uint64_t position = bz.r.bit;
    int more = bunzip_next_block(&bz);
    if (more < 0) { rc = more; break; }
    if (more) {
      size_t cnt = 0;
      rc = bunzip_read(&bz, out, &cnt);
      if (tab_n) {
        if (nblocks < tab_cap) { tab_pos[nblocks] = position; tab_size[nblocks] = (uint32_t)cnt; }
        nblocks++;
        if (rc == CJSO_DATA_ERROR && !out) rc = rc; /* table mode still checks the block CRC (Bzip2:1849) */
      }
      if (rc) break;
    } else {
      uint32_t target = (uint32_t)br_bits(&bz.r, 32);
      if (!tab_n && target != bz.stream_crc) { rc = CJSO_DATA_ERROR; break; }
      if (multistream && br_bytes_consumed(&bz.r) < n) {
        int level_before = bz.dbuf_size;
        rc = bunzip_start(&bz, in, n, br_bytes_consumed(&bz.r));
        (void)level_before;
        if (rc) break;
      } else break;
    }
  }
  free(bz.dbuf);
  if (tab_n) *tab_n = nblocks;
  if (!rc && out && out->oom) rc = CJSO_OUT_OF_MEMORY;
  return rc;
}
int cjs_oracle_bzip2_decompress(const uint8_t *in, size_t n, int multistream, uint8_t **out, size_t *out_n) {
  buf_t b = {0, 0, 0, 0};
  int rc = bunzip_decode(in, n, multistream, &b, 0, 0, 0, 0);
  if (rc) { free(b.p); return rc; }
  *out = b.p ? b.p : (uint8_t *)malloc(1); *out_n = b.n;
  return 0;
}
long cjs_oracle_bzip2_table(const uint8_t *in, size_t n, int multistream, uint64_t *bitpos, uint32_t *size, long cap) {
  long nb = 0;
  int rc = bunzip_decode(in, n, multistream, 0, bitpos, size, cap, &nb);
  return rc ? rc : nb;
}
int cjs_oracle_bzip2_decompress_block(const uint8_t *in, size_t n, uint64_t bitpos, uint8_t **out, size_t *out_n) {
  bunzip_t bz; memset(&bz, 0, sizeof bz);                            /* Bzip2:1797-1818 */
  int rc = bunzip_start(&bz, in, n, 0);
  if (rc) return rc;
  bz.dbuf = (uint32_t *)malloc(sizeof(uint32_t) * 900000u);
  if (!bz.dbuf) return CJSO_OUT_OF_MEMORY;
  bz.r.bit = bitpos;
  buf_t b = {0, 0, 0, 0};
  int more = bunzip_next_block(&bz);
  if (more < 0) rc = more;
  else if (more) { size_t cnt = 0; rc = bunzip_read(&bz, &b, &cnt); }
  free(bz.dbuf);
  if (rc) { free(b.p); return rc; }
  *out = b.p ? b.p : (uint8_t *)malloc(1); *out_n = b.n;
  return 0;
}

/* ------------------------------------------------------------------ RangeCoder (BWTC:16-252) */
#define RC_TOP 0x80000000u
#define RC_BOTTOM 0x00800000u
#define RC_SHIFT 23
#define RC_EXTRA 7
typedef struct {
  uint32_t low, range; int32_t buffer; uint32_t help; uint32_t bytecount;
  buf_t *out;                       /* encoder */
  const uint8_t *in; size_t in_n, in_pos;   /* decoder */
} rc_t;
static void rc_enc_normalize(rc_t *rc) {                              /* BWTC:51-73 */
  while (rc->range <= RC_BOTTOM) {
    if (rc->low < (0xFFu << RC_SHIFT)) {
      buf_put(rc->out, rc->buffer & 0xff);
      for (; rc->help; rc->help--) buf_put(rc->out, 0xFF);
      rc->buffer = (int32_t)((rc->low >> RC_SHIFT) & 0xFF);
    } else if (rc->low & RC_TOP) {
      buf_put(rc->out, (rc->buffer + 1) & 0xff);
      for (; rc->help; rc->help--) buf_put(rc->out, 0x00);
      rc->buffer = (int32_t)((rc->low >> RC_SHIFT) & 0xFF);
    } else rc->help++;
    rc->range <<= 8;
    rc->low = (rc->low << 8) & (RC_TOP - 1);
    rc->bytecount++;
  }
}
static void rc_encode_start(rc_t *rc, int c, uint32_t initlen) { rc->low = 0; rc->range = RC_TOP; rc->buffer = c; rc->help = 0; rc->bytecount = initlen; }
static void rc_encode_freq(rc_t *rc, uint32_t sy, uint32_t lt, uint32_t tot) {   /* BWTC:92-102 */
  rc_enc_normalize(rc);
  uint32_t r = rc->range / tot;
  uint32_t tmp = r * lt;
  rc->low += tmp;
  if (lt + sy < tot) rc->range = r * sy; else rc->range -= tmp;
}
static void rc_encode_shift(rc_t *rc, uint32_t sy, uint32_t lt, int shift) {    /* BWTC:103-113 */
  rc_enc_normalize(rc);
  uint32_t r = rc->range >> shift;
  uint32_t tmp = r * lt;
  rc->low += tmp;
  if ((lt + sy) >> shift) rc->range -= tmp; else rc->range = r * sy;
}
static uint32_t rc_encode_finish(rc_t *rc) {                                     /* BWTC:129-153 */
  rc_enc_normalize(rc);
  rc->bytecount += 5;
  uint32_t tmp = rc->low >> RC_SHIFT;
  if ((rc->low & (RC_BOTTOM - 1)) >= ((rc->bytecount & 0xFFFFFF) >> 1)) tmp++;
  if (tmp > 0xFF) { buf_put(rc->out, (rc->buffer + 1) & 0xff); for (; rc->help; rc->help--) buf_put(rc->out, 0x00); }
  else { buf_put(rc->out, rc->buffer & 0xff); for (; rc->help; rc->help--) buf_put(rc->out, 0xFF); }
  buf_put(rc->out, (int)(tmp & 0xFF));
  buf_put(rc->out, (int)((rc->bytecount >> 16) & 0xFF));
  buf_put(rc->out, (int)((rc->bytecount >> 8) & 0xFF));
  buf_put(rc->out, (int)(rc->bytecount & 0xFF));
  return rc->bytecount;
}
static int32_t rc_read_byte(rc_t *rc) { return rc->in_pos < rc->in_n ? (int32_t)rc->in[rc->in_pos++] : -1; }
static void rc_decode_start(rc_t *rc) {                                          /* BWTC:159-168 (skipInitialRead) */
  rc->buffer = rc_read_byte(rc);
  rc->low = (uint32_t)rc->buffer >> (8 - RC_EXTRA);
  rc->range = 1u << RC_EXTRA;
}
static void rc_dec_normalize(rc_t *rc) {                                         /* BWTC:170-179 */
  while (rc->range <= RC_BOTTOM) {
    rc->low = (rc->low << 8) | (((uint32_t)rc->buffer << RC_EXTRA) & 0xFF);
    rc->buffer = rc_read_byte(rc);
    rc->low |= (uint32_t)rc->buffer >> (8 - RC_EXTRA);
    rc->range <<= 8;
  }
}
static uint32_t rc_decode_culfreq(rc_t *rc, uint32_t tot) {                      /* BWTC:186-191 */
  rc_dec_normalize(rc);
  rc->help = rc->range / tot;
  uint32_t tmp = rc->help ? rc->low / rc->help : 0;
  return tmp >= tot ? tot - 1 : tmp;
}
static uint32_t rc_decode_culshift(rc_t *rc, int shift) {                        /* BWTC:192-198 */
  rc_dec_normalize(rc);
  rc->help = rc->range >> shift;
  uint32_t tmp = rc->help ? rc->low / rc->help : 0;
  return (tmp >> shift) ? (1u << shift) - 1 : tmp;
}
static void rc_decode_update(rc_t *rc, uint32_t sy, uint32_t lt, uint32_t tot) { /* BWTC:205-213 */
  uint32_t tmp = rc->help * lt;
  rc->low -= tmp;
  if (lt + sy < tot) rc->range = rc->help * sy; else rc->range -= tmp;
}
static uint32_t rc_decode_bit(rc_t *rc) { uint32_t t = rc_decode_culshift(rc, 1); rc_decode_update(rc, 1, t, 2); return t; }

/* NoModel (BWTC:1274-1295) + LogDistanceModel (BWTC:1224-1261) over raw coder bits */
static void nomodel_encode(rc_t *rc, int bits, uint32_t sym) { for (int i = bits - 1; i >= 0; i--) rc_encode_shift(rc, 1, (sym >> i) & 1, 1); }
static uint32_t nomodel_decode(rc_t *rc, int bits) { uint32_t r = 0; for (int i = bits - 1; i >= 0; i--) { r <<= 1; if (rc_decode_bit(rc)) r++; } return r; }
static void logdist_encode(rc_t *rc, int block_size, uint32_t d) {
  int lgbits = fls32((uint32_t)(1 + fls32((uint32_t)block_size - 1)) - 1);      /* NoModel(size=1+bits) */
  if (d < 2) { nomodel_encode(rc, lgbits, d); return; }
  int lg = fls32(d);
  nomodel_encode(rc, lgbits, (uint32_t)lg);
  nomodel_encode(rc, lg - 1, d & ((1u << (lg - 1)) - 1));
}
static uint32_t logdist_decode(rc_t *rc, int block_size) {
  int lgbits = fls32((uint32_t)(1 + fls32((uint32_t)block_size - 1)) - 1);
  uint32_t lg = nomodel_decode(rc, lgbits);
  if (lg < 2) return lg;
  uint32_t rest = nomodel_decode(rc, (int)lg - 1);
  return (1u << (lg - 1)) + rest;
}

/* FenwickModel (BWTC:1496-1661) */
typedef struct { rc_t *rc; int num_syms; uint32_t *tree; uint32_t increment, max_prob; } fen_t;
static void fen_sum(fen_t *m) { for (int i = m->num_syms - 1; i > 0; i--) m->tree[i] = m->tree[2 * i] + m->tree[2 * i + 1]; }
static int fen_init(fen_t *m, rc_t *rc, int size, uint32_t max_prob, uint32_t increment) {
  m->rc = rc; m->num_syms = size + 1; m->increment = increment; m->max_prob = max_prob;
  m->tree = (uint32_t *)calloc((size_t)m->num_syms * 2, sizeof(uint32_t));
  if (!m->tree) return -1;
  int i;
  for (i = 0; i < size; i++) m->tree[m->num_syms + i] = 1u;
  m->tree[m->num_syms + i] = increment << 16;
  fen_sum(m);
  return 0;
}
static void fen_rescale(fen_t *m) {                                              /* BWTC:1623-1654 */
  int i, no_escape = 1;
  uint32_t prob;
  for (i = 0; i < m->num_syms - 1; i++) {
    prob = m->tree[m->num_syms + i];
    if (prob & 0xFFFFu) { no_escape = 0; continue; }
    prob = (prob & 0xFFFEFFFEu) >> 1;
    if (prob == 0) { prob = 1u; no_escape = 0; }
    m->tree[m->num_syms + i] = prob;
  }
  prob = m->tree[m->num_syms + i];
  prob = (prob & 0xFFFEFFFEu) >> 1;
  if (no_escape) prob = 0; else if (prob == 0) prob = 1u << 16;
  m->tree[m->num_syms + i] = prob;
  fen_sum(m);
}
static void fen_encode(fen_t *m, int symbol) {                                   /* BWTC:1530-1571 */
  int i = m->num_syms + symbol;
  uint32_t sy = m->tree[i], mask = 0xFFFF0000u; int shift = 16;
  uint32_t update = m->increment << 16;
  if ((sy & 0xFFFF0000u) == 0) { fen_encode(m, m->num_syms - 1); mask = 0x0000FFFFu; update -= 1u; shift = 0; }
  else if (symbol == m->num_syms - 1 && (m->tree[1] & 0xFFFFu) == 1) update = 0u - m->tree[i];
  uint32_t lt = 0;
  while (i > 1) {
    int parent = i >> 1;
    if (i & 1) lt += m->tree[2 * parent];
    m->tree[i] += update;
    i = parent;
  }
  uint32_t tot = m->tree[1];
  m->tree[1] += update;
  rc_encode_freq(m->rc, (sy & mask) >> shift, (lt & mask) >> shift, (tot & mask) >> shift);
  if ((m->tree[1] >> 16) >= m->max_prob) fen_rescale(m);
}
static int fen_decode1(fen_t *m, int is_escape) {                                /* BWTC:1572-1614 */
  uint32_t mask = 0xFFFF0000u; int shift = 16;
  uint32_t update = m->increment << 16;
  if (is_escape) { mask = 0xFFFFu; update -= 1u; shift = 0; }
  uint32_t tot = (m->tree[1] & mask) >> shift;
  uint32_t prob = rc_decode_culfreq(m->rc, tot);
  int i = 1; uint32_t lt = 0;
  while (i < m->num_syms) {
    m->tree[i] += update;
    uint32_t left = (m->tree[2 * i] & mask) >> shift;
    i *= 2;
    if (prob - lt >= left) { lt += left; i++; }
  }
  int symbol = i - m->num_syms;
  uint32_t sy = (m->tree[i] & mask) >> shift;
  m->tree[i] += update;
  rc_decode_update(m->rc, sy, lt, tot);
  if (symbol == m->num_syms - 1 && (m->tree[1] & 0xFFFFu) == 1) {
    update = 0u - m->tree[i];
    while (i >= 1) { m->tree[i] += update; i >>= 1; }
  }
  if ((m->tree[1] >> 16) >= m->max_prob) fen_rescale(m);
  return symbol;
}
static int fen_decode(fen_t *m) { int s = fen_decode1(m, 0); if (s == m->num_syms - 1) s = fen_decode1(m, 1); return s; }

/* DefSumModel (BWTC:1327-1459) */
typedef struct { rc_t *rc; int num_syms; uint16_t prob[304], escape[304], update[304]; int update_count, update_thresh;
                 uint16_t prob_to_sym[256], esc_prob_to_sym[304]; int is_decoder; } dsm_t;
static void dsm_init(dsm_t *m, rc_t *rc, int size, int is_decoder) {
  memset(m, 0, sizeof *m);
  m->rc = rc; m->num_syms = size; m->is_decoder = is_decoder;
  m->prob[size + 1] = 256;
  for (int i = 0; i <= size; i++) m->escape[i] = (uint16_t)i;
  m->update_count = 0; m->update_thresh = 256 - 128;
  if (is_decoder) {
    for (int i = 0; i < 256; i++) m->prob_to_sym[i] = (uint16_t)size;
    for (int i = 0; i < size; i++) m->esc_prob_to_sym[i] = (uint16_t)i;
  }
}
static void dsm_update(dsm_t *m, int symbol) {                                   /* BWTC:1359-1421 */
  if (symbol == m->num_syms) {
    if (m->update[symbol] >= 40) return;
    if (m->update_count >= m->update_thresh - 1) return;
  }
  m->update[symbol]++;
  m->update_count++;
  if (m->update_count < m->update_thresh) return;
  int cum = 0, cum_esc = 0, odd = 0, i;
  m->escape[0] = 0; m->prob[0] = 0;
  for (i = 0; i < m->num_syms + 1; i++) {
    int np = ((m->prob[i + 1] - m->prob[i]) >> 1) + m->update[i];
    if (np) { m->prob[i] = (uint16_t)cum; cum += np; if (np & 1) odd++; m->escape[i] = (uint16_t)cum_esc; }
    else { m->prob[i] = (uint16_t)cum; m->escape[i] = (uint16_t)cum_esc; cum_esc++; }
  }
  m->prob[i] = (uint16_t)cum;
  m->update_thresh = 256 - (cum - odd) / 2;
  for (i = 0; i < m->num_syms + 1; i++) m->update[i] = 0;
  m->update[m->num_syms] = 1;
  m->update_count = 1;
  if (!m->is_decoder) return;
  int j = 0, k = 0;
  for (i = 0; i < m->num_syms + 1; i++) {
    int pl = m->prob[i + 1];
    for (; j < pl; j++) m->prob_to_sym[j] = (uint16_t)i;
    /* escape[] has num_syms+1 entries in the reference; index num_syms+1 reads undefined -> loop no-op */
    int el = (i + 1 <= m->num_syms) ? m->escape[i + 1] : 0;
    for (; k < el; k++) m->esc_prob_to_sym[k] = (uint16_t)i;
  }
}
static void dsm_encode(dsm_t *m, int symbol) {                                   /* BWTC:1422-1439 */
  uint32_t lt = m->prob[symbol], sy = (uint32_t)m->prob[symbol + 1] - lt;
  if (sy) { rc_encode_shift(m->rc, sy, lt, 8); dsm_update(m, symbol); return; }
  dsm_encode(m, m->num_syms);
  lt = m->escape[symbol]; sy = (uint32_t)m->escape[symbol + 1] - lt;
  rc_encode_freq(m->rc, sy, lt, m->escape[m->num_syms]);
  dsm_update(m, symbol);
}
static int dsm_decode(dsm_t *m) {                                                /* BWTC:1440-1459 */
  uint32_t prob = rc_decode_culshift(m->rc, 8);
  int symbol = m->prob_to_sym[prob];
  uint32_t lt = m->prob[symbol], sy = (uint32_t)m->prob[symbol + 1] - lt;
  rc_decode_update(m->rc, sy, lt, 256);
  dsm_update(m, symbol);
  if (symbol != m->num_syms) return symbol;
  uint32_t tot = m->escape[m->num_syms];
  prob = rc_decode_culfreq(m->rc, tot);
  symbol = m->esc_prob_to_sym[prob];
  lt = m->escape[symbol]; sy = (uint32_t)m->escape[symbol + 1] - lt;
  rc_decode_update(m->rc, sy, lt, tot);
  dsm_update(m, symbol);
  return symbol;
}

/* BWTC.compressFile (BWTC:1698-1825) + Util.compressFileHelper (BWTC:516-553) */
int cjs_oracle_bwtc_compress(const uint8_t *in, size_t n, int level, uint8_t **out, size_t *out_n) {
  buf_t b = {0, 0, 0, 0};
  buf_put(&b, 'b'); buf_put(&b, 'w'); buf_put(&b, 't'); buf_put(&b, 'c');
  uint8_t vb[12]; int nv = 0;                                         /* writeUnsignedNumber BWTC:605-620 */
  uint64_t v = (uint64_t)n + 1;
  do { vb[nv++] = (uint8_t)(v & 0x7F); v >>= 7; } while (v);
  vb[0] |= 0x80;
  for (int i = nv - 1; i >= 1; i--) buf_put(&b, vb[i]);
  rc_t rc; memset(&rc, 0, sizeof rc); rc.out = &b;
  rc_encode_start(&rc, vb[0], 1);
  if (level < 1 || level > 9) level = 9;                             /* W2 */
  rc_encode_shift(&rc, 1, (uint32_t)level, 8);
  int fast = level <= 5;
  int block_size = level * 100000;
  uint8_t *U = (uint8_t *)malloc((size_t)block_size);
  if (!U) { free(b.p); return CJSO_OUT_OF_MEMORY; }
  size_t cursor = 0;
  int length, rcode = 0;
  do {
    length = (int)(n - cursor < (size_t)block_size ? n - cursor : (size_t)block_size);
    if (length == 0) break;
    const uint8_t *blk = in + cursor; cursor += (size_t)length;
    if (length == block_size) rc_encode_freq(&rc, 1, 0, 3);
    else { rc_encode_freq(&rc, 1, 1, 3); logdist_encode(&rc, block_size, (uint32_t)length); }
    int pidx = cjs_oracle_bwt_sentinel(blk, length, U);
    if (pidx < 0) { rcode = CJSO_OUT_OF_MEMORY; break; }
    logdist_encode(&rc, block_size, (uint32_t)pidx);
    uint16_t tree[512]; memset(tree, 0, sizeof tree);                /* use-tree BWTC:1744-1765 */
    for (int i = 0; i < length; i++) tree[256 + U[i]] = 1;
    for (int i = 255; i > 0; i--) tree[i] = (uint16_t)(tree[2 * i] + tree[2 * i + 1]);
    tree[0] = 1;
    for (int i = 1; i < 512; i++) {
      int parent = i >> 1, full = 1 << (9 - fls32((uint32_t)i));
      if (tree[parent] == 0 || tree[parent] == full * 2) continue;
      if (i >= 256) rc_encode_shift(&rc, 1, tree[i] ? 1 : 0, 1);
      else { uint32_t vv = tree[i] == 0 ? 0 : tree[i] == full ? 2 : 1; rc_encode_freq(&rc, 1, vv, 3); }
    }
    uint8_t M[256]; int asz = 0;
    for (int i = 0; i < 256; i++) if (tree[256 + i]) M[asz++] = (uint8_t)i;
    for (int i = 0; i < length; i++) {                               /* MTF BWTC:1775-1789 */
      uint8_t c = U[i]; int j = 0;
      while (M[j] != c) j++;
      U[i] = (uint8_t)j;
      for (; j > 0; j--) M[j] = M[j - 1];
      M[0] = c;
    }
    fen_t fm; dsm_t *dm = 0; fm.tree = 0;
    if (fast) { dm = (dsm_t *)malloc(sizeof(dsm_t)); dsm_init(dm, &rc, asz + 1, 0); }
    else if (fen_init(&fm, &rc, asz + 1, 0xFF00, 0x100)) { rcode = CJSO_OUT_OF_MEMORY; break; }
#define MENC(s) do { if (fast) dsm_encode(dm, (s)); else fen_encode(&fm, (s)); } while (0)
    uint32_t run = 0;
    for (int i = 0; i < length; i++) {                               /* RLE2 BWTC:1794-1819 */
      int c = U[i];
      if (c == 0) run++;
      else {
        while (run) { if (run & 1) { MENC(0); run -= 1; } else { MENC(1); run -= 2; } run >>= 1; }
        MENC(c + 1);
      }
    }
    while (run) { if (run & 1) { MENC(0); run -= 1; } else { MENC(1); run -= 2; } run >>= 1; }
#undef MENC
    free(fm.tree); free(dm);
  } while (length == block_size);
  free(U);
  if (rcode) { free(b.p); return rcode; }
  rc_encode_freq(&rc, 1, 2, 3);
  rc_encode_finish(&rc);
  if (b.oom) { free(b.p); return CJSO_OUT_OF_MEMORY; }
  *out = b.p; *out_n = b.n;
  return 0;
}

/* BWT.unbwtransform (BWTC:1147-1168) */
static void unbwt_sentinel(const uint8_t *T, uint8_t *U, uint32_t *LF, int n, int pidx) {
  uint32_t C[256]; memset(C, 0, sizeof C);
  for (int i = 0; i < n; i++) LF[i] = C[T[i]]++;
  uint32_t t = 0;
  for (int i = 0; i < 256; i++) { t += C[i]; C[i] = t - C[i]; }
  t = 0;
  for (int i = n - 1; i >= 0; i--) {
    U[i] = T[t];
    t = LF[t] + C[U[i]];
    t += (t < (uint32_t)pidx) ? 1 : 0;
  }
}
/* BWTC.decompressFile (BWTC:1827-1920) + Util.decompressFileHelper (BWTC:554-577) */
int cjs_oracle_bwtc_decompress(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n) {
  if (n < 4 || in[0] != 'b' || in[1] != 'w' || in[2] != 't' || in[3] != 'c') return CJSO_BAD_MAGIC;
  size_t p = 4;
  for (;;) { if (p >= n) return CJSO_DATA_ERROR; uint8_t c = in[p++]; if (c & 0x80) break; }   /* readUnsignedNumber */
  rc_t rc; memset(&rc, 0, sizeof rc); rc.in = in; rc.in_n = n; rc.in_pos = p;
  rc_decode_start(&rc);
  uint32_t lv = rc_decode_culshift(&rc, 8); rc_decode_update(&rc, 1, lv, 256);
  if (lv < 1 || lv > 9) return CJSO_DATA_ERROR;
  int fast = lv <= 5, block_size = (int)lv * 100000;
  uint8_t *blk = (uint8_t *)malloc((size_t)block_size + 2), *U = (uint8_t *)malloc((size_t)block_size);
  uint32_t *LF = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)block_size);
  buf_t b = {0, 0, 0, 0};
  int rcode = 0;
  if (!blk || !U || !LF) rcode = CJSO_OUT_OF_MEMORY;
  while (!rcode) {
    uint32_t ind = rc_decode_culfreq(&rc, 3); rc_decode_update(&rc, 1, ind, 3);
    int length;
    if (ind == 0) length = block_size;
    else if (ind == 1) { length = (int)logdist_decode(&rc, block_size); if (length > block_size) { rcode = CJSO_DATA_ERROR; break; } }
    else break;
    int pidx = (int)logdist_decode(&rc, block_size);
    uint16_t tree[512]; memset(tree, 0, sizeof tree); tree[0] = 1;
    for (int i = 1; i < 512; i++) {
      int parent = i >> 1, full = 1 << (9 - fls32((uint32_t)i));
      if (tree[parent] == 0 || tree[parent] == full * 2) tree[i] = tree[parent] >> 1;
      else if (i >= 256) tree[i] = (uint16_t)rc_decode_bit(&rc);
      else { uint32_t vv = rc_decode_culfreq(&rc, 3); rc_decode_update(&rc, 1, vv, 3); tree[i] = (uint16_t)(vv == 2 ? full : vv); }
    }
    uint8_t M[256]; int asz = 0;
    for (int i = 0; i < 256; i++) if (tree[256 + i]) M[asz++] = (uint8_t)i;
    fen_t fm; dsm_t *dm = 0; fm.tree = 0;
    if (fast) { dm = (dsm_t *)malloc(sizeof(dsm_t)); dsm_init(dm, &rc, asz + 1, 1); }
    else if (fen_init(&fm, &rc, asz + 1, 0xFF00, 0x100)) { rcode = CJSO_OUT_OF_MEMORY; break; }
    int64_t val = 1; int i = 0, bad = 0;
    while (i < length) {
      int c = fast ? dsm_decode(dm) : fen_decode(&fm);
      if (rc.in_pos > n + 8) { bad = 1; break; }
      if (c == 0) { if (i + val > length) { bad = 1; break; } for (int64_t j = 0; j < val; j++) blk[i++] = 0; val *= 2; }
      else if (c == 1) { if (i + 2 * val > length) { bad = 1; break; } for (int64_t j = 0; j < val; j++) { blk[i++] = 0; blk[i++] = 0; } val *= 2; }
      else { val = 1; if (c - 1 >= asz) { bad = 1; break; } blk[i++] = (uint8_t)(c - 1); }
    }
    free(fm.tree); free(dm);
    if (bad) { rcode = CJSO_DATA_ERROR; break; }
    for (i = 0; i < length; i++) {
      int j = blk[i]; uint8_t c = M[j];
      blk[i] = c;
      for (; j > 0; j--) M[j] = M[j - 1];
      M[0] = c;
    }
    unbwt_sentinel(blk, U, LF, length, pidx);
    for (i = 0; i < length; i++) buf_put(&b, U[i]);
  }
  free(blk); free(U); free(LF);
  if (rcode) { free(b.p); return rcode; }
  if (b.oom) { free(b.p); return CJSO_OUT_OF_MEMORY; }
  *out = b.p ? b.p : (uint8_t *)malloc(1); *out_n = b.n;
  return 0;
}
