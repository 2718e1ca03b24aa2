// huff.h — Huffman table construction + bit packing stage buffers and entry points.
#pragma once
#include "cjs_internal.h"

namespace cjs {

struct HuffBufs {
  uint8_t* sel;       // [nb][sel_stride] table index per group of 50 symbols
  uint8_t* selj;      // [nb][sel_stride] MTF position of each selector (unary-coded in the stream)
  uint16_t* bcost;    // [nb][sel_stride] cost of each group under its selected table
  uint8_t* lens;      // [nb][6][258]
  uint32_t* codes;    // [nb][6][258] canonical codes
  uint32_t* ngroups;  // [nb]
  uint32_t* bitlen;   // [nb] bits of the block incl. its 48-bit magic and CRC
  uint64_t* bitoff;   // [nb+1] absolute bit offset of each block in the output
  uint32_t* databits; // [nb] bits of the block's symbol data (the last part of the block)
  uint32_t* tileoff;  // [nb][tile_stride] bit offset of every 80-group (4000-symbol) tile inside the symbol data
  uint8_t* wl;        // [nb][6][264] code lengths between the kernels of the split refinement
  uint32_t* wfreq;    // [nb][6][260] symbol counts per table, same
  size_t sel_stride, tile_stride;
};

struct HuffWork {
  size_t max_blocks = 0;
  uint32_t max_stride = 0;
  HuffBufs b{};
  uint64_t* scalars = nullptr;   // [0] total bits, [1] (u32) stream crc, [2] output too small
  static size_t sel_stride_for(uint32_t stride) { return (((size_t)stride + 1 + 49) / 50 + 63) & ~(size_t)63; }
  static size_t bytes_needed(size_t max_blocks, uint32_t stride);
  int carve(Arena& a, size_t max_blocks, uint32_t stride);
  // the table-construction buffers seen from block `first` on (`count` blocks); bitoff / scalars belong to the packing of a whole call
  HuffWork view(size_t first, size_t count) const {
    HuffWork v = *this;
    v.max_blocks = count;
    v.b.sel += first * b.sel_stride; v.b.selj += first * b.sel_stride; v.b.bcost += first * b.sel_stride;
    v.b.lens += first * 6 * 258; v.b.codes += first * 6 * 258;
    v.b.ngroups += first; v.b.bitlen += first; v.b.databits += first; v.b.tileoff += first * b.tile_stride;
    v.b.wl += first * 6 * 264; v.b.wfreq += first * 6 * 260;
    return v;
  }
};

int huff_tables_run(hipStream_t s, HuffWork& w, uint32_t nb, const uint16_t* d_A, size_t a_stride, const uint32_t* d_npos,
                    const uint32_t* d_asz, const uint32_t* d_freq, const uint8_t* d_alist);
// a rank of a multi-GPU job: the stream CRC folded over ALL ranks' blocks (the trailer writer needs it), and whether the next
// rank's blocks follow this fragment (pack_frame then completes the fragment's last word with the leading bits of the block magic)
struct PackShard { uint32_t stream_crc; int follow_magic; };
// Packs blocks [first, first+count) starting at absolute bit `start_bit` of d_out32 (the part the stream occupies is
// zeroed here, the bits in front of start_bit included).
int huff_pack_run(hipStream_t s, HuffWork& w, uint32_t nb_total, uint32_t first, uint32_t count, uint64_t start_bit, int level,
                  int write_header, int write_trailer, const uint16_t* d_A, size_t a_stride, const uint32_t* d_npos,
                  const uint32_t* d_asz, const uint8_t* d_alist, const uint32_t* d_block_crc, const uint32_t* d_pidx,
                  uint32_t* d_out32, size_t out_cap_bytes, const PackShard* ps = nullptr);    // scalars[2] = 1 and nothing written if the stream does not fit

}  // namespace cjs
