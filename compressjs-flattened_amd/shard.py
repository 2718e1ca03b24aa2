"""shard.py — host-side logic of the multi-GPU bzip2 path: deal contiguous block ranges to
ranks, fold the per-block CRCs into the stream CRC (J/Bzip2_joined_.js:2237) and funnel-shift the ranks'
bare bit strings into one .bz2 stream.  No collective on the data path: ranks exchange only
(bit length, block CRCs)."""
import numpy as np


def plan_ranges(total_blocks, world):
    """contiguous (first, count) per rank; ranks beyond the last block get count 0"""
    share = -(-total_blocks // world) if total_blocks else 0
    out = []
    for r in range(world):
        first = min(r * share, total_blocks)
        out.append((first, min(share, total_blocks - first)))
    return out


def fold_stream_crc(block_crcs):
    c = 0
    for b in block_crcs:
        c = (((c << 1) | (c >> 31)) & 0xFFFFFFFF) ^ int(b)
    return c


def _put_bits(buf, bitpos, value, nbits):
    for i in range(nbits):
        if (value >> (nbits - 1 - i)) & 1:
            buf[(bitpos + i) >> 3] |= 0x80 >> ((bitpos + i) & 7)


def assemble(level, parts, block_crcs):
    """parts: [(uint8 array, nbits)] in rank order -> complete stream ('BZh'+level, blocks, trailer)"""
    total_bits = 32 + sum(nb for _, nb in parts) + 80
    out = np.zeros((total_bits + 7) // 8 + 1, dtype=np.uint8)
    out[:4] = np.frombuffer(b"BZh%d" % level, dtype=np.uint8)
    pos = 32
    for data, nbits in parts:
        nbytes = (nbits + 7) // 8
        src = np.asarray(data[:nbytes], dtype=np.uint8)
        if nbits & 7:                      # clear the padding bits of the last byte
            src = src.copy()
            src[-1] &= (0xFF << (8 - (nbits & 7))) & 0xFF
        s = pos & 7
        o = pos >> 3
        if s == 0:
            out[o:o + nbytes] |= src
        else:                              # funnel shift by s bits
            out[o:o + nbytes] |= src >> s
            out[o + 1:o + 1 + nbytes] |= (src << (8 - s)).astype(np.uint8)
        pos += nbits
    _put_bits(out, pos, 0x177245385090, 48)
    _put_bits(out, pos + 48, fold_stream_crc(block_crcs), 32)
    return out[: (total_bits + 7) // 8]
