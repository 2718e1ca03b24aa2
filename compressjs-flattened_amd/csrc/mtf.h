// mtf.h — MTF / RLE2 stage buffers (per block, stride-addressed) and entry point.
#pragma once
#include "cjs_internal.h"

namespace cjs {

struct MtfBufs {
  uint32_t* hpos;     // [nb][stride]  run-head positions in U
  uint8_t* hsym;      // [nb][stride]  run-head symbols
  uint8_t* hrank;     // [nb][stride]  MTF rank of each head
  uint8_t* lists;     // [nb][list_stride] MTF list at the start of every 256-head chunk
  uint16_t* A;        // [nb][a_stride] MTF/RLE2 symbols incl. EOB
  uint32_t* freq;     // [nb][258]
  uint8_t* alist;     // [nb][256] used byte values, ascending
  uint32_t *asz, *nheads, *npos;   // [nb]
  int* segkeys;       // [nb][seg_stride][256] last-occurrence sort keys at every 32-chunk segment start
  size_t seg_stride;
  size_t list_stride, a_stride, hstride;   // hstride: per-block stride of hpos/hsym/hrank (multiple of 16)
};

struct MtfWork {
  size_t max_blocks = 0;
  uint32_t stride = 0;
  MtfBufs b{};
  static size_t list_stride_for(uint32_t stride) { return ((size_t)(stride + 255) / 256 + 1) * 256; }
  static size_t seg_stride_for(uint32_t stride) { return ((size_t)stride + 8191) / 8192 + 1; }
  static size_t hstride_for(uint32_t stride) { return ((size_t)stride + 15) & ~(size_t)15; }
  static size_t a_stride_for(uint32_t stride) { return ((size_t)stride + 2 + 7) & ~(size_t)7; }
  static size_t bytes_needed(size_t max_blocks, uint32_t stride);
  int carve(Arena& a, size_t max_blocks, uint32_t stride);
  // the same workspace seen from block `first` on (`count` blocks): a piece of a call that is worked on by itself
  MtfWork view(size_t first, size_t count) const {
    MtfWork v = *this;
    v.max_blocks = count;
    v.b.hpos += first * b.hstride; v.b.hsym += first * b.hstride; v.b.hrank += first * b.hstride;
    v.b.lists += first * b.list_stride; v.b.A += first * b.a_stride; v.b.freq += first * 258; v.b.alist += first * 256;
    v.b.asz += first; v.b.nheads += first; v.b.npos += first;
    v.b.segkeys += first * b.seg_stride * 256;
    return v;
  }
};

// d_U: BWT bytes, block k at k*stride with length d_blen[k]
int mtf_run(hipStream_t s, MtfWork& w, const uint8_t* d_U, uint32_t nb, const uint32_t* d_blen);

}  // namespace cjs
