/* textgen.c — deterministic synthetic "enwik8-shaped" byte stream.
 *
 * TEST / BENCH INFRASTRUCTURE, not product code.  Integer-only arithmetic (no libm) so
 * the stream is bit-identical on every host: the golden fixtures under tests/golden/ were
 * cut by feeding the output of this generator to the reference JS under Node.
 *
 * Shape: Zipf-distributed vocabulary of 32768 pseudo-words, wiki/XML page framing,
 * 1-12 paragraphs per page, 5 % wiki markup, paragraph / page repeat injection (gives the
 * long-LCP tail a suffix sorter sees on real Wikipedia text), rare rule lines and long
 * space runs so bzip2's RLE1 (runs >= 4, 255-chunking) stays exercised.
 *
 * FROZEN: any change here invalidates tests/golden/manifest.json (generator_version).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define TG_VERSION 1
#define VOCAB 32768
#define PARA_RING 600
#define PAGE_RING 64
#define GUIDE_BITS 12

typedef struct {
  uint32_t s;
  char *words;            /* VOCAB * 16 bytes, NUL padded */
  uint8_t wlen[VOCAB];
  uint64_t cum[VOCAB + 1];
  uint32_t guide[(1u << GUIDE_BITS) + 1];
  uint64_t total;
  /* repeat rings: copies of recent paragraphs / page bodies */
  char *para[PARA_RING]; uint32_t para_len[PARA_RING]; uint32_t para_n;
  char *page[PAGE_RING]; uint32_t page_len[PAGE_RING]; uint32_t page_n;
  /* output */
  uint8_t *out; size_t cap, pos;
  /* scratch for the page body being built */
  char *body; size_t body_cap, body_len;
  uint32_t page_id;
} tg_t;

static inline uint32_t rnd(tg_t *g) {
  uint32_t s = g->s;
  s ^= s << 13; s ^= s >> 17; s ^= s << 5;
  g->s = s;
  return s;
}
static inline uint32_t rndn(tg_t *g, uint32_t n) { /* uniform in [0,n) */
  return (uint32_t)(((uint64_t)rnd(g) * n) >> 32);
}

static void build_vocab(tg_t *g) {
  static const char letters[] = "etaoinshrdlcumwfgypbvkjxqz";
  g->words = (char *)calloc(VOCAB, 16);
  for (int r = 0; r < VOCAB; r++) {
    int len = 2 + rndn(g, 4) + rndn(g, 4);
    if (r >= 2000) len += rndn(g, 3);
    char *w = g->words + (size_t)r * 16;
    for (int k = 0; k < len; k++) {
      uint32_t u = rnd(g) >> 16;             /* 16 bits */
      uint32_t v = (u * u) >> 16;            /* skew towards frequent letters */
      w[k] = letters[(26 * v) >> 16];
    }
    g->wlen[r] = (uint8_t)len;
  }
  /* Zipf(s=1) weights, integer */
  uint64_t c = 0;
  for (int r = 0; r < VOCAB; r++) { g->cum[r] = c; c += 4000000000ull / (uint64_t)(r + 1); }
  g->cum[VOCAB] = c; g->total = c;
  /* guide table: first rank whose cum range can contain x with top GUIDE_BITS bits */
  int r = 0;
  for (uint32_t q = 0; q <= (1u << GUIDE_BITS); q++) {
    uint64_t x = (g->total >> GUIDE_BITS) * q;
    while (r + 1 < VOCAB && g->cum[r + 1] <= x) r++;
    g->guide[q] = (uint32_t)r;
  }
}

static inline int zipf(tg_t *g) {
  uint64_t x = (((uint64_t)rnd(g) << 20) ^ (uint64_t)rnd(g)) % g->total;
  uint64_t step = g->total >> GUIDE_BITS;
  uint32_t q = (uint32_t)(x / step);
  if (q > (1u << GUIDE_BITS)) q = 1u << GUIDE_BITS;
  int r = (int)g->guide[q];
  while (r + 1 < VOCAB && g->cum[r + 1] <= x) r++;
  return r;
}

static void body_reserve(tg_t *g, size_t extra) {
  if (g->body_len + extra + 64 > g->body_cap) {
    g->body_cap = (g->body_len + extra + 64) * 2;
    g->body = (char *)realloc(g->body, g->body_cap);
  }
}
static void body_put(tg_t *g, const char *s, size_t n) {
  body_reserve(g, n);
  memcpy(g->body + g->body_len, s, n);
  g->body_len += n;
}
static void body_str(tg_t *g, const char *s) { body_put(g, s, strlen(s)); }
static void body_word(tg_t *g) {
  int r = zipf(g);
  body_put(g, g->words + (size_t)r * 16, g->wlen[r]);
}
static void body_num(tg_t *g, uint32_t v) {
  char t[16]; int n = snprintf(t, sizeof t, "%u", v);
  body_put(g, t, (size_t)n);
}
static void body_fill(tg_t *g, char c, size_t n) {
  body_reserve(g, n);
  memset(g->body + g->body_len, c, n);
  g->body_len += n;
}

static void gen_sentence(tg_t *g) {
  int nw = 4 + rndn(g, 22);
  for (int k = 0; k < nw; k++) {
    if (k) body_put(g, " ", 1);
    uint32_t m = rndn(g, 100);
    if (m < 5) {
      switch (rndn(g, 6)) {
        case 0: body_str(g, "[["); body_word(g); body_str(g, "]]"); break;
        case 1: body_str(g, "[["); body_word(g); body_str(g, "|"); body_word(g); body_str(g, "]]"); break;
        case 2: body_str(g, "''"); body_word(g); body_str(g, "''"); break;
        case 3: body_str(g, "&quot;"); body_word(g); body_str(g, "&quot;"); break;
        case 4: body_str(g, "{{"); body_word(g); body_str(g, "}}"); break;
        default: body_num(g, rndn(g, 3000)); break;
      }
    } else {
      size_t at = g->body_len;
      body_word(g);
      if (k == 0) g->body[at] = (char)(g->body[at] - 32);   /* capitalise */
    }
    if (k + 1 < nw && rndn(g, 12) == 0) body_put(g, ",", 1);
  }
  body_put(g, ". ", 2);
}

static void ring_store(char **ring, uint32_t *lens, uint32_t *count, uint32_t cap,
                       const char *s, size_t n) {
  uint32_t slot = *count % cap;
  free(ring[slot]);
  ring[slot] = (char *)malloc(n ? n : 1);
  memcpy(ring[slot], s, n);
  lens[slot] = (uint32_t)n;
  (*count)++;
}

static void gen_paragraph(tg_t *g) {
  if (g->para_n > 0 && rndn(g, 12) == 0) {        /* verbatim repeat of a recent paragraph */
    uint32_t have = g->para_n < PARA_RING ? g->para_n : PARA_RING;
    uint32_t k = rndn(g, have);
    body_put(g, g->para[k], g->para_len[k]);
    return;
  }
  size_t start = g->body_len;
  uint32_t m = rndn(g, 2048);
  if (m == 0) {                                     /* rare long run (RLE1 255-chunking) */
    body_fill(g, ' ', 200 + rndn(g, 400));
  } else if (m < 32) {                              /* rule line, run of 4..15 */
    body_fill(g, '-', 4 + rndn(g, 12));
    body_put(g, "\n", 1);
  } else if (m < 96) {                              /* heading */
    body_str(g, "=="); body_word(g); body_put(g, " ", 1); body_word(g); body_str(g, "==\n");
  } else if (m < 128) {                             /* indented line */
    body_fill(g, ' ', 4 + rndn(g, 5));
  }
  int ns = 1 + rndn(g, 7);
  for (int k = 0; k < ns; k++) gen_sentence(g);
  body_put(g, "\n\n", 2);
  ring_store(g->para, g->para_len, &g->para_n, PARA_RING, g->body + start, g->body_len - start);
}

static void out_put(tg_t *g, const char *s, size_t n) {
  if (g->pos >= g->cap) return;
  if (n > g->cap - g->pos) n = g->cap - g->pos;
  memcpy(g->out + g->pos, s, n);
  g->pos += n;
}

static void gen_page(tg_t *g) {
  char hdr[512];
  g->page_id += 1 + rndn(g, 7);
  /* title words */
  g->body_len = 0;
  body_word(g); g->body[0] = (char)(g->body[0] - 32);
  if (rndn(g, 2)) { body_put(g, " ", 1); body_word(g); }
  int tl = (int)g->body_len; if (tl > 100) tl = 100;
  int n = snprintf(hdr, sizeof hdr,
                   "  <page>\n    <title>%.*s</title>\n    <id>%u</id>\n    <revision>\n      <id>%u</id>\n"
                   "      <timestamp>2006-%02u-%02uT%02u:%02u:%02uZ</timestamp>\n"
                   "      <text xml:space=\"preserve\">",
                   tl, g->body, g->page_id, 15898000u + rndn(g, 60000000u),
                   1 + rndn(g, 12), 1 + rndn(g, 28), rndn(g, 24), rndn(g, 60), rndn(g, 60));
  out_put(g, hdr, (size_t)n);
  g->body_len = 0;
  if (g->page_n > 0 && rndn(g, 100) == 0) {         /* verbatim repeat of an earlier page body */
    uint32_t have = g->page_n < PAGE_RING ? g->page_n : PAGE_RING;
    uint32_t k = rndn(g, have);
    body_put(g, g->page[k], g->page_len[k]);
  } else {
    int np = 1 + rndn(g, 12);
    for (int k = 0; k < np; k++) gen_paragraph(g);
    ring_store(g->page, g->page_len, &g->page_n, PAGE_RING, g->body, g->body_len);
  }
  out_put(g, g->body, g->body_len);
  static const char tail[] = "</text>\n    </revision>\n  </page>\n";
  out_put(g, tail, sizeof tail - 1);
}

/* Fill out[0..n) with the stream for `seed`.  Returns TG_VERSION. */
int cjs_textgen(uint8_t *out, size_t n, uint32_t seed) {
  tg_t *g = (tg_t *)calloc(1, sizeof(tg_t));
  g->s = seed ? seed : 1u;
  g->out = out; g->cap = n; g->pos = 0;
  build_vocab(g);
  static const char head[] = "<mediawiki xml:lang=\"en\">\n  <siteinfo>\n    <sitename>Synthetic</sitename>\n  </siteinfo>\n";
  out_put(g, head, sizeof head - 1);
  while (g->pos < g->cap) gen_page(g);
  for (int i = 0; i < PARA_RING; i++) free(g->para[i]);
  for (int i = 0; i < PAGE_RING; i++) free(g->page[i]);
  free(g->body); free(g->words); free(g);
  return TG_VERSION;
}

int cjs_textgen_version(void) { return TG_VERSION; }

#ifdef TEXTGEN_MAIN
int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: textgen <bytes> <seed> > out\n"); return 2; }
  size_t n = (size_t)strtoull(argv[1], 0, 10);
  uint32_t seed = (uint32_t)strtoul(argv[2], 0, 10);
  uint8_t *buf = (uint8_t *)malloc(n ? n : 1);
  cjs_textgen(buf, n, seed);
  fwrite(buf, 1, n, stdout);
  free(buf);
  return 0;
}
#endif
