"""shard.py — host-side logic of the multi-GPU bzip2 path, mirrored in Python for the launcher side of a
one-process-per-GPU job and for the CPU tests: deal contiguous block ranges to ranks, chain the ranks' CRC folds into the
stream CRC (J/Bzip2_joined_.js:2237), and the layout of the ranks' fragments -- each rank packs its blocks at their FINAL bit
offset, so the fragments are disjoint runs of whole 32-bit words of the one .bz2 stream (pipeline.hip: shard_layout /
shard_pack_core).  No collective on the data path: ranks exchange only (bit length, block count, CRC fold)."""
import numpy as np

BLOCK_MAGIC = 0x314159265359
END_MAGIC = 0x177245385090


def plan_ranges(total_blocks, world):
    """contiguous (first, count) per rank; ranks beyond the last block get count 0"""
    share = -(-total_blocks // world) if total_blocks else 0
    out = []
    for r in range(world):
        first = min(r * share, total_blocks)
        out.append((first, min(share, total_blocks - first)))
    return out


def fold_stream_crc(block_crcs, start=0):
    c = start
    for b in block_crcs:
        c = (((c << 1) | (c >> 31)) & 0xFFFFFFFF) ^ int(b)
    return c


def chain_folds(metas):
    """metas: [(bits, blocks, crc_fold)] in rank order -> stream CRC: a rank's fold enters as rol(c, blocks) ^ fold"""
    c = 0
    for _, blocks, fold in metas:
        rot = blocks & 31
        c = ((((c << rot) | (c >> (32 - rot))) & 0xFFFFFFFF) if rot else c) ^ fold
    return c


def layout(metas):
    """-> (per rank (start_bit, stream_off, frag_len), writer rank, stream length in bytes); metas as in chain_folds"""
    world = len(metas)
    total = 32 + sum(m[0] for m in metas)
    writer = max([r for r in range(world) if metas[r][1]] or [0])
    stream_len = (total + 80 + 7) // 8
    out, start = [], 32
    for r, (bits, blocks, _) in enumerate(metas):
        if r and not blocks:
            out.append((start, stream_len, 0))
        else:
            lo = 0 if r == 0 else ((start + 31) // 32) * 4
            hi = stream_len if r == writer else ((start + bits + 31) // 32) * 4
            out.append((start, lo, hi - lo))
        start += bits
    return out, writer, stream_len


def _put_bits(buf, bitpos, value, nbits):
    for i in range(nbits):
        if (value >> (nbits - 1 - i)) & 1:
            buf[(bitpos + i) >> 3] |= 0x80 >> ((bitpos + i) & 7)


def _or_bits(out, pos, src, nbits):
    """out bits [pos, pos + nbits) |= first nbits bits of src (uint8 array, MSB first)"""
    nbytes = (nbits + 7) // 8
    src = np.asarray(src[:nbytes], dtype=np.uint8).copy()
    if nbits & 7:                      # clear the padding bits of the last byte
        src[-1] &= (0xFF << (8 - (nbits & 7))) & 0xFF
    s, o = pos & 7, pos >> 3
    if s == 0:
        out[o:o + nbytes] |= src
    else:                              # funnel shift by s bits
        out[o:o + nbytes] |= src >> s
        out[o + 1:o + 1 + nbytes] |= (src << (8 - s)).astype(np.uint8)


def fragment(level, metas, rank, bitstring):
    """What rank `rank` leaves in its HBM: its bare bit string (uint8, from bit 0) placed at its final offset, cut to whole
    words; rank 0 adds 'BZh<level>', the writer the trailer, every other rank the leading bits of the next block magic."""
    lay, writer, stream_len = layout(metas)
    start, lo, n = lay[rank]
    if n == 0:
        return lo, np.empty(0, dtype=np.uint8)
    bits = metas[rank][0]
    buf = np.zeros(stream_len + 16, dtype=np.uint8)
    if rank == 0:
        buf[:4] = np.frombuffer(b"BZh%d" % level, dtype=np.uint8)
    _or_bits(buf, start, bitstring, bits)
    end = start + bits
    if rank == writer:
        _put_bits(buf, end, END_MAGIC, 48)
        _put_bits(buf, end + 48, chain_folds(metas), 32)
    elif end & 31:
        room = 32 - (end & 31)
        _put_bits(buf, end, BLOCK_MAGIC >> (48 - room), room)
    return lo, buf[lo:lo + n].copy()       # (a rank's first, partial word belongs to the rank in front: lo is the next word)


def concat(frags, stream_len):
    """[(stream_off, uint8 array)] in rank order -> the stream; fragments must tile [0, stream_len) exactly"""
    pos, chunks = 0, []
    for off, b in frags:
        if len(b):
            if off != pos:
                raise ValueError("fragment at %d, expected %d" % (off, pos))
            chunks.append(b)
            pos += len(b)
    if pos != stream_len:
        raise ValueError("fragments cover %d of %d bytes" % (pos, stream_len))
    return np.concatenate(chunks) if chunks else np.empty(0, dtype=np.uint8)


def assemble(level, parts, block_crcs):
    """parts: [(uint8 array, nbits)] in rank order -> complete stream ('BZh'+level, blocks, trailer); the funnel-shift form"""
    total_bits = 32 + sum(nb for _, nb in parts) + 80
    out = np.zeros((total_bits + 7) // 8 + 1, dtype=np.uint8)
    out[:4] = np.frombuffer(b"BZh%d" % level, dtype=np.uint8)
    pos = 32
    for data, nbits in parts:
        _or_bits(out, pos, data, nbits)
        pos += nbits
    _put_bits(out, pos, END_MAGIC, 48)
    _put_bits(out, pos + 48, fold_stream_crc(block_crcs), 32)
    return out[: (total_bits + 7) // 8]
