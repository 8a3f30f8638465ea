#!/usr/bin/env python3
"""tests/golden/spec_bytes.bam, assembled byte by byte from the SAM/BAM specification (SAMv1.pdf section 4.1 BGZF,
4.2 the BAM format) -- NOT through tests/bamtools.py, whose writer shares its author's understanding of the format
with bam_reader.cpp.  Every field below is spelled out with the spec's name; the DEFLATE payloads are STORED blocks
(RFC 1951 section 3.2.4: BFINAL/BTYPE byte 0x01, LEN, NLEN, the bytes), so no compressor is involved and the file is
exactly the bytes written here.  The only computed values are the CRC-32s and the lengths.

Records (reference of the expected output: detect.rs:290-338, anchor length 10):
  1 unmapped, QNAME with ':'                     -> sent as read 1, fragment id with '_' (detect.rs:315-318)
  2 mapped, CIGAR 12S18M                          -> leading soft clip >= 10: sent as read 2 (detect.rs:305-310)
  3 mapped, CIGAR 20M10S, every 4-bit base code   -> trailing soft clip >= 10: sent as read 3; "=ACMGRSVTWYHKDBN"
  4 mapped, CIGAR 30M                             -> dropped (aligned at full length)
  5 mapped, CIGAR 9S21M                           -> dropped (soft clip shorter than the anchor)
  6 unmapped, 19 bases                            -> dropped (shorter than two anchors, detect.rs:311); odd l_seq
  7 mapped, CIGAR 5H11S19M                        -> the soft clip behind a hard clip counts: sent as read 4
  8 unmapped, 21 bases (odd l_seq: a padding nibble) -> sent as read 5
The BGZF block boundary falls inside record 4; the file ends with the 28-byte EOF marker block.
"""
import os
import struct
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
CODES = "=ACMGRSVTWYHKDBN"                     # SAMv1 4.2: seq nibble -> base


def u8(v): return struct.pack("<B", v)
def u16(v): return struct.pack("<H", v)
def u32(v): return struct.pack("<I", v)
def i32(v): return struct.pack("<i", v)


def seq_nibbles(s):
    """4-bit packed sequence, high nibble first; an odd length leaves the last low nibble 0 (SAMv1 4.2)"""
    codes = [CODES.index(c) for c in s]
    if len(codes) & 1:
        codes.append(0)
    return bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(codes), 2))


def record(qname, flag, seq, cigar=(), ref_id=-1, pos=-1):
    """One alignment record, SAMv1 4.2 (all integers little endian)"""
    name = qname + b"\0"
    ops = b"".join(u32((n << 4) | "MIDNSHP=X".index(op)) for n, op in cigar)      # op_len << 4 | op
    body = b"".join([
        i32(ref_id),                # refID
        i32(pos),                   # pos (0-based)
        u8(len(name)),              # l_read_name (with NUL)
        u8(0),                      # mapq
        u16(4680),                  # bin (reg2bin(-1, 0) for unplaced reads; unused by the reader)
        u16(len(cigar)),            # n_cigar_op
        u16(flag),                  # flag
        u32(len(seq)),              # l_seq
        i32(-1),                    # next_refID
        i32(-1),                    # next_pos
        i32(0),                     # tlen
        name,                       # read_name
        ops,                        # cigar
        seq_nibbles(seq),           # seq
        b"\xff" * len(seq),         # qual (0xff = absent)
    ])
    return u32(len(body)) + body    # block_size


def bgzf_block(data):
    """SAMv1 4.1: a gzip member with the BC extra subfield; payload = one STORED deflate block"""
    cdata = b"\x01" + u16(len(data)) + u16(len(data) ^ 0xffff) + data
    total = 12 + 6 + len(cdata) + 8
    return b"".join([
        b"\x1f\x8b",                # ID1 ID2
        b"\x08",                    # CM = deflate
        b"\x04",                    # FLG = FEXTRA
        u32(0),                     # MTIME
        b"\x00",                    # XFL
        b"\xff",                    # OS = unknown
        u16(6),                     # XLEN
        b"BC", u16(2),              # SI1 SI2 SLEN
        u16(total - 1),             # BSIZE = total block size - 1
        cdata,
        u32(zlib.crc32(data) & 0xffffffff),   # CRC32 of the uncompressed data
        u32(len(data)),             # ISIZE
    ])


EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")     # SAMv1 4.1.2

IUPAC = "=ACMGRSVTWYHKDBN" + "ACGTACGTACGTAC"                                         # 30 bases
READS = [
    (b"q:1:a", 4, "ACGTACGTACGTACGTACGTAC", ()),
    (b"clip_lead", 0, "TTTTTTTTTTTTGGGGGGGGGGGGGGGGGG", ((12, "S"), (18, "M"))),
    (b"clip:trail", 16, IUPAC, ((20, "M"), (10, "S"))),
    (b"full", 0, "ACACACACACACACACACACACACACACAC", ((30, "M"),)),
    (b"short_clip", 0, "GAGAGAGAGAGAGAGAGAGAGAGAGAGAGA", ((9, "S"), (21, "M"))),
    (b"tiny", 4, "ACGTACGTACGTACGTACG", ()),
    (b"hard:soft", 0, "CCCCCCCCCCCAAAAAAAAAAAAAAAAAAA", ((5, "H"), (11, "S"), (19, "M"))),
    (b"odd", 4, "GATTACAGATTACAGATTACA", ()),
]

def main():
    header_text = b"@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:chr1\tLN:1000\n"
    stream = b"".join([
        b"BAM\x01",                 # magic
        u32(len(header_text)), header_text,      # l_text, text
        u32(1),                     # n_ref
        u32(5), b"chr1\0", u32(1000),            # l_name, name, l_ref
    ])
    recs = [record(q, f, s, c, ref_id=-1 if f & 4 else 0, pos=-1 if f & 4 else 100) for q, f, s, c in READS]
    stream += b"".join(recs)
    cut = len(stream) - len(b"".join(recs[3:])) + 17          # inside record 4
    bam = bgzf_block(stream[:cut]) + bgzf_block(stream[cut:]) + EOF_BLOCK
    open(os.path.join(HERE, "spec_bytes.bam"), "wb").write(bam)
    print("spec_bytes.bam: %d bytes, %d records, block boundary at uncompressed offset %d" % (len(bam), len(recs), cut))


if __name__ == "__main__":
    main()
