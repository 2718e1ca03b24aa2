/* cjs_oracle.h — CPU restatement of the compressjs Bzip2 / BWTC block-sorting path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, loaded by or called from
 * the product library (compressjs-flattened_amd/csrc, libcjs_hip.so) or the JS fronts.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and there
 * only as the checker / reported CPU baseline.
 *
 * Parity status: PINNED.  Every entry point below is checked (tests/test_oracle.py) against
 *   - the reference's own known answers (cyclic-BWT KATs NPM/test/bwtest.js:39-79, allocator
 *     KATs NPM/test/huffman.js:15-76, decoder goldens sample0-4.bz2, .bzt tables, block dumps)
 *   - outputs of the reference JS itself run under Node in the build container, committed as
 *     tests/golden/*.json by tests/golden/make_golden.js (length + sha256 of every stream).
 *
 * Citations: J/ = /root/reference/ (Bzip2_joined_.js, BWTC_joined_.js).
 */
#ifndef CJS_ORACLE_H
#define CJS_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* error codes: J/Bzip2_joined_.js:1365-1375 */
#define CJSO_OK 0
#define CJSO_NOT_BZIP_DATA (-2)
#define CJSO_DATA_ERROR (-5)
#define CJSO_OUT_OF_MEMORY (-6)
#define CJSO_OBSOLETE_INPUT (-7)
#define CJSO_BAD_LEVEL (-20)     /* Error('Invalid block size multiplier') J/Bzip2_joined_.js:2208 */
#define CJSO_BAD_MAGIC (-21)     /* Error("Bad magic") J/BWTC_joined_.js:559-565 */

/* whole streams */
int cjs_oracle_bzip2_compress(const uint8_t *in, size_t n, int level, uint8_t **out, size_t *out_n);
/* blocks [first, first+count) only, as a bare bit string from bit 0 (no header / trailer); crcs[] gets every block's CRC */
int cjs_oracle_bzip2_compress_range(const uint8_t *in, size_t n, int level, long first, long count, uint8_t **out,
                                    uint64_t *out_bits, uint32_t *crcs, long crc_cap, long *total_blocks);
int cjs_oracle_bzip2_decompress(const uint8_t *in, size_t n, int multistream, uint8_t **out, size_t *out_n);
int cjs_oracle_bwtc_compress(const uint8_t *in, size_t n, int level, uint8_t **out, size_t *out_n);
int cjs_oracle_bwtc_decompress(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n);
/* Bzip2.table: fills pos[]/size[] (up to cap entries), returns the number of blocks or <0 */
long cjs_oracle_bzip2_table(const uint8_t *in, size_t n, int multistream, uint64_t *bitpos, uint32_t *size, long cap);
int cjs_oracle_bzip2_decompress_block(const uint8_t *in, size_t n, uint64_t bitpos, uint8_t **out, size_t *out_n);
void cjs_oracle_free(void *p);

/* stages (for stage-level parity of the HIP kernels) */
uint32_t cjs_oracle_crc32(const uint8_t *p, size_t n);                           /* J/Bzip2:1048-1079 */
int cjs_oracle_suffix_array(const uint8_t *T, int n, int32_t *SA);               /* J/Bzip2:862-876 */
int cjs_oracle_bwt_cyclic(const uint8_t *T, int n, uint8_t *U);                  /* J/Bzip2:928-971, returns pidx */
int cjs_oracle_bwt_sentinel(const uint8_t *T, int n, uint8_t *U);                /* J/BWTC:1125-1145, returns pidx */
void cjs_oracle_huff_alloc(int32_t *arr, int n, int maxlen);                     /* J/Bzip2:1275-1298 */
void cjs_oracle_huff_lengths(const uint32_t *freq, int alphabet, uint8_t *len);  /* J/Bzip2:1866-1894 */
/* readBlock: consumes input from in[*cursor..n), fills block[0..cap), returns length; crc out */
int cjs_oracle_rle1_block(const uint8_t *in, size_t n, size_t *cursor, uint8_t *block, int cap, uint32_t *crc);
/* MTF+RLE2 of one block: U = BWT bytes, block = RLE1 bytes (for the used map). Returns pos (symbols incl. EOB) */
int cjs_oracle_mtf_rle2(const uint8_t *U, const uint8_t *block, int n, uint16_t *A, uint32_t *freq, int *alphabet_size);
/* selectors + tables for one block (optimizeHuffmanGroups + final assignSelectors).
 * lengths is [6][258]; returns number of tables; selectors has ceil(pos/50) entries */
int cjs_oracle_huff_groups(const uint16_t *A, int pos, int alphabet_size, uint8_t *selectors, uint8_t *lengths);

#ifdef __cplusplus
}
#endif
#endif
