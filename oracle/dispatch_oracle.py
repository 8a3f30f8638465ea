"""CPU oracle for the BAM front-end -- TEST INFRASTRUCTURE ONLY (see detect_oracle.py's header).

Restates `dispatch_reads_to_bowtie` (/root/reference/src/detect.rs:290-338) and `read_bam_record`
(src/common.rs:61-67) on top of an independent pure-Python BGZF/BAM reader (SAMv1 section 4).
Third-party behaviour restated (crate rust-htslib 0.38, Cargo.toml:16, not vendored):
  Record::is_unmapped   = flag & 0x4
  Record::seq().as_bytes() decodes 4-bit codes with "=ACMGRSVTWYHKDBN"
  Record::qname()        = read name without the trailing NUL
  CigarStringView::leading_softclips / trailing_softclips = the soft clip at that end, looking through
  one hard clip (H then S counts, S alone counts, anything else is 0).
PARITY UNPINNED by the reference (no fixtures); pinned by hand-written cases in tests/test_bam_frontend.py.
"""
import struct
import zlib
from typing import Iterator, List, Tuple

NIB = b"=ACMGRSVTWYHKDBN"


def bgzf_decompress(data: bytes) -> bytes:
    out = []
    o = 0
    while o < len(data):
        if data[o:o + 4] != b"\x1f\x8b\x08\x04":
            raise ValueError("not BGZF")
        xlen = struct.unpack_from("<H", data, o + 10)[0]
        extra = data[o + 12:o + 12 + xlen]
        bsize = None
        e = 0
        while e + 4 <= xlen:
            si1, si2, slen = extra[e], extra[e + 1], struct.unpack_from("<H", extra, e + 2)[0]
            if si1 == 66 and si2 == 67 and slen == 2:
                bsize = struct.unpack_from("<H", extra, e + 4)[0]
            e += 4 + slen
        clen = bsize - xlen - 19
        comp = data[o + 12 + xlen:o + 12 + xlen + clen]
        crc, isize = struct.unpack_from("<II", data, o + 12 + xlen + clen)
        raw = zlib.decompress(comp, -15) if isize else b""
        assert len(raw) == isize and (zlib.crc32(raw) & 0xffffffff) == crc
        out.append(raw)
        o += bsize + 1
    return b"".join(out)


def bam_records(raw: bytes) -> Iterator[Tuple[int, bytes, List[int], bytes]]:
    """-> (flag, qname, cigar ops as len<<4|op, decoded seq) per record"""
    assert raw[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<I", raw, 4)[0]
    o = 8 + l_text
    n_ref = struct.unpack_from("<I", raw, o)[0]
    o += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from("<I", raw, o)[0]
        o += 4 + l_name + 4
    while o < len(raw):
        block_size = struct.unpack_from("<I", raw, o)[0]
        p = o + 4
        l_read_name = raw[p + 8]
        n_cigar, flag = struct.unpack_from("<HH", raw, p + 12)
        l_seq = struct.unpack_from("<I", raw, p + 16)[0]
        q = p + 32
        qname = raw[q:q + l_read_name - 1]
        q += l_read_name
        cigar = list(struct.unpack_from("<%dI" % n_cigar, raw, q))
        q += 4 * n_cigar
        seq = bytes(NIB[(raw[q + (i >> 1)] >> (0 if i & 1 else 4)) & 15] for i in range(l_seq))
        yield flag, qname, cigar, seq
        o += 4 + block_size


def leading_softclips(cigar: List[int]) -> int:
    if not cigar:
        return 0
    if cigar[0] & 15 == 4:
        return cigar[0] >> 4
    if cigar[0] & 15 == 5 and len(cigar) > 1 and cigar[1] & 15 == 4:
        return cigar[1] >> 4
    return 0


def trailing_softclips(cigar: List[int]) -> int:
    return leading_softclips(cigar[::-1])


def dispatch_reads_to_bowtie(bam_bytes: bytes, anchor_len: int) -> bytes:
    """What detect.rs:290-338 writes to bowtie's stdin for this BAM file."""
    out = []
    num_reads_sent = 0
    for flag, qname, cigar, seq in bam_records(bgzf_decompress(bam_bytes)):
        if not (flag & 4):                                                       # detect.rs:305-310
            if leading_softclips(cigar) < anchor_len and trailing_softclips(cigar) < anchor_len:
                continue
        if len(seq) < anchor_len * 2:                                            # detect.rs:311
            continue
        frag_id = qname.replace(b":", b"_")                                      # detect.rs:315-318
        num_reads_sent += 1
        out.append(b">5p:%d:\n%s" % (num_reads_sent, seq[:anchor_len]))          # detect.rs:326-327
        out.append(b"\n>3p:%d:%s:%s:\n%s\n" % (num_reads_sent, frag_id, seq, seq[len(seq) - anchor_len:]))
    return b"".join(out)
