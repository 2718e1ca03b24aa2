// rle1.h — stage 0 (RLE1 + block boundaries + CRC) workspace and entry point.
#pragma once
#include "cjs_internal.h"

namespace cjs {

struct RleBlock {
  uint64_t s, e;       // input range consumed by the block
  uint64_t r_end;      // end of the run that contains s (block-fresh chunking applies to [s, r_end))
  uint64_t Gr;         // global emitted-byte prefix at r_end
  uint32_t len;        // RLE1 output length (== cap for every block but possibly the last)
  uint32_t base;       // bytes emitted by the first (re-chunked) run
};

struct Rle1Work {
  size_t max_in = 0;
  uint32_t cap = 0, max_blocks = 0, max_segs = 0, range_blocks = 0;
  uint32_t* h_n = nullptr;       // pinned host scalar
  uint64_t *fb = nullptr, *lb = nullptr, *gt = nullptr;
  unsigned long long* agg = nullptr;     // chunk aggregates of the three-phase tile scans
  uint16_t* subpre = nullptr;    // [tiles][16] emitted bytes of the tile before each 256-byte subtile
  uint8_t* dmod = nullptr;       // [tiles][16] chunk phase of each subtile's first byte
  RleBlock* blocks = nullptr;
  uint32_t *block_len = nullptr, *block_crc = nullptr, *nblocks = nullptr, *seg_crc = nullptr;
  static size_t max_blocks_for(size_t max_in, uint32_t cap) { return max_in / ((size_t)cap * 4 / 5) + 2; }
  static size_t max_segs_for(uint32_t cap) { return ((size_t)cap * 51 + 16383) / 16384 + 1; }
  // multi-GPU jobs: every rank makes the tile tables of `tiles_per_rank` consecutive 4 KiB tiles and the ranks exchange them.
  // Layout of one share (tpr tiles): lb u64[tpr] | fb u64[tpr] | gt u64[tpr] | subpre u16[tpr][16] | dmod u8[tpr][16]
  static uint32_t tiles_for(uint64_t n) { return (uint32_t)((n + 4095) / 4096); }
  static uint32_t tiles_per_rank(uint64_t n, uint32_t world) { const uint32_t t = (tiles_for(n) + world - 1) / (world ? world : 1); return (t + 3u) & ~3u; }
  __host__ __device__ static size_t share_bytes(uint32_t tpr) { return (size_t)72 * tpr; }
  // range_blocks = max number of blocks materialised / CRC'd per call (0 = all blocks of the stream)
  static size_t bytes_needed(size_t max_in, uint32_t cap, size_t range_blocks = 0);
  int carve(Arena& a, size_t max_in, uint32_t cap, size_t range_blocks = 0);
  void release() { if (h_n) (void)hipHostFree(h_n); h_n = nullptr; }
};

int rle1_run(hipStream_t s, Rle1Work& w, const uint8_t* d_in, uint64_t N, uint32_t* nblocks_host, uint32_t* last_len_host = nullptr);
// the two halves of rle1_run for jobs that shard the tile passes over ranks (see rle1.hip)
int rle1_tiles(hipStream_t s, Rle1Work& w, const uint8_t* d_in, uint64_t N, uint32_t t0, uint32_t t1, uint8_t* share, uint32_t tpr);
int rle1_tables_from_shares(hipStream_t s, Rle1Work& w, uint64_t N, const uint8_t* d_recv, uint32_t tpr);
int rle1_walk_run(hipStream_t s, Rle1Work& w, const uint8_t* d_in, uint64_t N, uint32_t* nblocks_host, uint32_t* last_len_host = nullptr);
int rle1_finish(hipStream_t s, Rle1Work& w, const uint8_t* d_in, uint64_t N, uint32_t first, uint32_t count, uint8_t* d_blocks,
                hipStream_t side = nullptr, hipEvent_t ev_fork = nullptr, hipEvent_t ev_join = nullptr);

}  // namespace cjs
