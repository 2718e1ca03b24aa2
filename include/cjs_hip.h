/* cjs_hip.h — C ABI of the MI355X-native block-sorting core (libcjs_hip.so).
 *
 * This is the drop-in boundary for the compressjs Bzip2 / BWTC hot path: exactly what an FFI
 * binding of the reference's per-algorithm `compressFile` / `decompressFile` would call
 * (N-API shim: compressjs-flattened_amd/js/cjs_napi.cc; ctypes: tests/support.py).
 * Plain pointers and sizes only.  Functions never throw; they return 0 or a negative code.
 * Every entry point needs a HIP device: without one the call fails with CJS_E_NO_DEVICE
 * (there is NO CPU fallback in this library).
 *
 * J/ = /root/reference/ (reference source, cited for parity checks).
 */
#ifndef CJS_HIP_H
#define CJS_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- return codes.  -1..-8 keep the reference's Err table (J/Bzip2_joined_.js:1365-1375) */
#define CJS_OK 0
#define CJS_E_NOT_BZIP_DATA (-2)   /* TypeError "Not bzip data[: bad magic|level out of range]" */
#define CJS_E_DATA_ERROR (-5)      /* TypeError "Data error[: Bad block CRC ...]" */
#define CJS_E_OUT_OF_MEMORY (-6)
#define CJS_E_OBSOLETE_INPUT (-7)  /* randomised blocks */
#define CJS_E_BAD_LEVEL (-20)      /* Error('Invalid block size multiplier') J/Bzip2_joined_.js:2208 */
#define CJS_E_BAD_MAGIC (-21)      /* Error("Bad magic") J/BWTC_joined_.js:559-565 */
#define CJS_E_NO_DEVICE (-30)      /* no HIP device / HIP runtime error */
#define CJS_E_HIP (-31)
#define CJS_E_INVALID_ARG (-32)
#define CJS_E_OUTPUT_TOO_SMALL (-33)
#define CJS_E_UNSUPPORTED (-34)

/* ---- options (all optional; pass NULL for defaults) */
typedef struct cjs_stats {
  double ms_total;        /* device time of the whole call (hipEvents on the work stream) */
  double ms_rle1;         /* RLE1 + CRC + block boundaries */
  double ms_bwt;          /* suffix sort + BWT emit */
  double ms_mtf;          /* MTF + RLE2 + histogram */
  double ms_huff;         /* Huffman table construction / optimisation */
  double ms_pack;         /* bit packing + stream assembly */
  double ms_bwt_dominant; /* average launch duration of the dominant kernel (radix scatter) */
  uint64_t bwt_dominant_launches;
  uint64_t bwt_dominant_bytes;   /* algorithmic bytes moved by those launches */
  uint64_t blocks;
  uint64_t bytes_in, bytes_out;
  uint32_t bwt_rounds;
  uint32_t reserved;      /* written as 0 */
} cjs_stats;               /* OUT only: cleared and filled by the call, never read */
/* cjs_bwtc_compress fills the same struct with wall-clock times of its two halves: ms_total = whole call, ms_bwt =
 * longest GPU batch (workspace + H2D + BWT + MTF + model), ms_mtf = time until the first step list reached the host,
 * ms_pack = serial range coder over the step lists (host), ms_rle1 = time the coder spent waiting for the GPU. */

typedef struct cjs_opts {
  uint32_t struct_size;   /* sizeof(cjs_opts) */
  int32_t device;         /* HIP device ordinal; -1 = current device */
  uint32_t n_devices;     /* host-buffer entry points: shard blocks over this many GPUs (0/1 = one) */
  uint32_t flags;         /* CJS_FLAG_* */
  cjs_stats *stats;       /* optional out */
} cjs_opts;
/* cjs_bwtc_compress: the input came from a stream without a known size, so the header carries varint(0) instead of
 * varint(size+1) (Util.compressFileHelper, J/BWTC_joined_.js:529-543) */
#define CJS_FLAG_SIZE_UNKNOWN 1u
/* cjs_bzip2_compress with opts->stats: no stream synchronisation between the stages -- ms_rle1 / ms_bwt / ms_mtf / ms_huff /
 * ms_pack stay 0, the rest is filled (whole-call events, dominant-kernel events, counts).  Device-resident entry points:
 * cjs_ctx_set_stage_times(ctx, 0). */
#define CJS_FLAG_NO_STAGE_TIMES 2u

/* ---- host-buffer entry points (what the JS fronts bind).
 * cjs_bzip2_compress   replaces Bzip2.compressFile    J/Bzip2_joined_.js:2199-2249
 * cjs_bzip2_decompress replaces Bzip2.decompressFile  J/Bzip2_joined_.js:1769-1796
 * cjs_bwtc_compress    replaces BWTC.compressFile     J/BWTC_joined_.js:1698-1825
 * cjs_bwtc_decompress  replaces BWTC.decompressFile   J/BWTC_joined_.js:1827-1920
 * `*out` is allocated by the library (plain malloc, or for results of 1 MiB and more a cached pinned host buffer: see
 * cjs_trim); release it with cjs_free and with nothing else.  level: 1..9 (bzip2: else
 * CJS_E_BAD_LEVEL; bwtc: else 9, J/BWTC_joined_.js:1702-1705). */
int cjs_bzip2_compress(const uint8_t *in, size_t n, int level, uint8_t **out, size_t *out_n, const cjs_opts *opts);
int cjs_bzip2_decompress(const uint8_t *in, size_t n, int multistream, uint8_t **out, size_t *out_n, const cjs_opts *opts);
int cjs_bwtc_compress(const uint8_t *in, size_t n, int level, uint8_t **out, size_t *out_n, const cjs_opts *opts);
int cjs_bwtc_decompress(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n, const cjs_opts *opts);
/* Bzip2.table (J/Bzip2_joined_.js:1823-1863): fills (bit position, uncompressed size) of up to `cap` blocks,
 * returns the number of blocks or a negative code.  Bzip2.decompressBlock (:1797-1818): the single block
 * whose 48-bit magic starts at bit `bitpos`. */
long cjs_bzip2_table(const uint8_t *in, size_t n, int multistream, uint64_t *bitpos, uint32_t *size, long cap, const cjs_opts *opts);
int cjs_bzip2_decompress_block(const uint8_t *in, size_t n, uint64_t bitpos, uint8_t **out, size_t *out_n, const cjs_opts *opts);
void cjs_free(void *p);
/* Memory kept between calls (allocating and freeing multi-GB scratch costs more than compressing 100 MB):
 * cjs_bzip2_compress keeps its per-device workspace (~70 B per input byte of the largest call so far) and staging buffers;
 * cjs_bzip2_decompress / _table / _decompress_block keep their device scratch buffers (~25 B per output byte) in a
 * per-device pool; result buffers given back with cjs_free stay pinned for the next result (at most CJS_PINNED_RESULT_MB
 * megabytes of idle ones, default 2048; 0 = results are plain malloc).  cjs_trim() returns all of it to the driver;
 * environment CJS_NO_CTX_CACHE=1: never keep device memory. */
void cjs_trim(void);
const char *cjs_strerror(int code);
/* Detail text of the most recent FAILED call on the calling thread, "" if it had none: the reference's optDetail
 * of _throw(status, optDetail) (J/Bzip2_joined_.js:1385-1391), e.g. "bad magic", "level out of range",
 * "initial position out of bounds", "Bad block CRC (got 1a2b3c4d expected 5e6f7081)", "Bad stream CRC (got .. expected ..)"
 * (:1413,1417,1450,1757,1783).  A front appends it to cjs_strerror(code) after ": ".  Valid until the thread's next call. */
const char *cjs_last_error_detail(void);
int cjs_device_count(void);
const char *cjs_version(void);

/* ---- device-resident pipeline (input already in HBM, output left in HBM): what bench.py times.
 * A context owns the per-GPU workspace (sized for max_input bytes at `level`) and one stream. */
typedef struct cjs_ctx cjs_ctx;
int cjs_ctx_create(cjs_ctx **ctx, int device, size_t max_input, int level);
/* as above, but the per-block workspace (suffix sorter, MTF, Huffman) is sized for at most
 * max_range_blocks blocks per call: for cjs_bzip2_compress_device_range on a replicated stream */
int cjs_ctx_create_sharded(cjs_ctx **ctx, int device, size_t max_input, long max_range_blocks, int level);
void cjs_ctx_destroy(cjs_ctx *ctx);
/* Calls on this context that are given a cjs_stats: on != 0 (default) synchronise the stream between the stages and fill the
 * per-stage times; on == 0 only record events (whole call, every full-size launch of the dominant kernel) -- what bench.py
 * sets for its timed loop. */
void cjs_ctx_set_stage_times(cjs_ctx *ctx, int on);
/* d_in/d_out are device pointers; d_out has out_cap bytes; *out_n receives the stream length.
 * Synchronous on return (the context stream has drained). */
int cjs_bzip2_compress_device(cjs_ctx *ctx, const uint8_t *d_in, size_t n, int level,
                              uint8_t *d_out, size_t out_cap, size_t *out_n, cjs_stats *stats);
/* Sharded variant for one-process-per-GPU jobs: compress only blocks [first, first+count) of the
 * stream held (replicated) in d_in, writing the block bit-strings from bit 0 of d_out WITHOUT the
 * 'BZh' header / trailer.  Returns the bit length and the per-block CRCs so the ranks can fold the
 * stream CRC and bit offsets (host side, a few bytes per rank; no data-path collective).
 * count = -1 means "to the end".  *total_blocks receives the number of blocks of the stream. */
int cjs_bzip2_compress_device_range(cjs_ctx *ctx, const uint8_t *d_in, size_t n, int level,
                                    long first_block, long count, uint8_t *d_out, size_t out_cap,
                                    uint64_t *out_bits, uint32_t *block_crcs, long crc_cap,
                                    long *total_blocks, cjs_stats *stats);

/* ---- one process (or thread) per GPU: the three phases of a multi-GPU Bzip2.compressFile (replaces the block
 * loop J/Bzip2_joined_.js:2233-2247 for `world` GPUs).  Every rank holds the stream in its GPU's memory.  The library calls no
 * collective: between the phases the CALLER exchanges two small tables with whatever transport it owns (bench.py:
 * torch.distributed all_gather over RCCL; cjs_bzip2_compress with n_devices > 1: worker threads and host memory).
 *   1. cjs_bzip2_shard_tiles : boundary tables of this rank's share of the 4 KiB input tiles -> d_share
 *                              (cjs_bzip2_shard_share_bytes(n, world) bytes, the same for every rank);
 *      exchange: all-gather of the shares, rank order, back to back -> d_shares (world x share bytes);
 *   2. cjs_bzip2_shard_blocks: block boundaries of the stream (replicated: serial by the format, Q1-Q3), then this rank's
 *                              contiguous range of blocks through RLE1 / CRC / BWT / MTF / Huffman tables -> *meta;
 *                              world == 1 may pass d_shares = NULL (no phase 1);
 *      exchange: all-gather of the metas (32 bytes per rank);
 *   3. cjs_bzip2_shard_pack  : the rank's blocks at their FINAL bit offset.  The ranks' fragments are disjoint runs of whole
 *                              32-bit words of the one .bz2 stream: bytes [frag_off, frag_off + frag_len) of d_out are stream
 *                              bytes [stream_off, stream_off + frag_len); rank 0 writes 'BZh<level>', the last rank with blocks
 *                              the trailer and the combined CRC.  The word two ranks share is completed by the earlier one
 *                              (what follows is always the 48-bit block magic).  *stream_len (optional) = length of the stream.
 * Each call is synchronous (the context's stream has drained on return). */
typedef struct cjs_shard_meta {
  uint64_t bits;          /* bit length of this rank's blocks, without header / trailer */
  uint64_t total_blocks;  /* blocks of the whole stream (must agree between the ranks) */
  uint64_t first_block;   /* this rank's contiguous range: [first_block, first_block + blocks) */
  uint32_t blocks;
  uint32_t crc_fold;      /* the range's block CRCs folded from 0: c = rol1(c) ^ crc (J/Bzip2_joined_.js:2237) */
} cjs_shard_meta;
size_t cjs_bzip2_shard_share_bytes(size_t n, int world);
int cjs_bzip2_shard_tiles(cjs_ctx *ctx, const uint8_t *d_in, size_t n, int rank, int world, void *d_share);
int cjs_bzip2_shard_blocks(cjs_ctx *ctx, const uint8_t *d_in, size_t n, int level, int rank, int world, const void *d_shares,
                           cjs_shard_meta *meta, cjs_stats *stats);
int cjs_bzip2_shard_pack(cjs_ctx *ctx, int level, int rank, int world, const cjs_shard_meta *metas, uint8_t *d_out, size_t out_cap,
                         size_t *frag_off, size_t *frag_len, uint64_t *stream_off, uint64_t *stream_len);

/* ---- stage-level entry points (host buffers; used by the parity tests to localise a mismatch).
 * in = nb consecutive blocks of block_len bytes (last one may be shorter).
 * cjs_stage_bwt: cyclic!=0 -> BWT.bwtransform2 semantics (J/Bzip2_joined_.js:928-971, Q4),
 *                cyclic==0 -> BWT.bwtransform semantics (J/BWTC_joined_.js:1125-1145). */
int cjs_stage_bwt(const uint8_t *in, size_t n, int block_len, int cyclic, uint8_t *out, int32_t *pidx, const cjs_opts *opts);
/* readBlock (J/Bzip2_joined_.js:1954-1985) for the whole stream: RLE1 bytes of all blocks,
 * concatenated with stride = level*100000-19; per block length / crc / input start */
int cjs_stage_rle1(const uint8_t *in, size_t n, int level, uint8_t *blocks, size_t blocks_cap,
                   uint32_t *block_len, uint32_t *block_crc, uint64_t *block_start, long cap, long *nblocks, const cjs_opts *opts);
/* MTF + RLE2 (J/Bzip2_joined_.js:2064-2139) for nb blocks: U and block bytes with the same layout as
 * cjs_stage_bwt; A = u16 symbols with stride block_len+1 */
int cjs_stage_mtf(const uint8_t *U, const uint8_t *blocks, size_t n, int block_len, uint16_t *A, uint32_t *npos,
                  uint32_t *freq /* nb*258 */, uint32_t *alphabet /* nb */, const cjs_opts *opts);
/* Huffman tables + selectors (J/Bzip2_joined_.js:1989-2054,2147-2163) for one symbol stream */
int cjs_stage_huff(const uint16_t *A, uint32_t npos, uint32_t alphabet, uint8_t *selectors, uint8_t *lengths /* 6*258 */,
                   uint32_t *ngroups, const cjs_opts *opts);

/* Serial entropy stage of BWTC.decompressFile (J/BWTC_joined_.js:1827-1913): range decoder + adaptive model + RLE2 + MTF
 * inverse, i.e. everything before BWT.unbwtransform.  Host logic only (the one entry point that needs no device; the chain
 * is serial by the format).  *cols receives the BWT columns of all non-empty blocks back to back (malloc'd, cjs_free);
 * lens[k] / pidx[k] for up to cap blocks; *level = the stream's level.  Returns the number of blocks or a negative code. */
long cjs_stage_bwtc_entropy_decode(const uint8_t *in, size_t n, uint8_t **cols, size_t *cols_n, uint32_t *lens, uint32_t *pidx,
                                   long cap, int *level);

#ifdef __cplusplus
}
#endif
#endif
