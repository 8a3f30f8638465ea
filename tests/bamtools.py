"""Test helpers: a minimal BAM writer and a stand-in for the Bowtie1 binary (absent from the image)."""
import os
import stat
import struct
import sys
import zlib

NIB = b"=ACMGRSVTWYHKDBN"
OPS = "MIDNSHP=X"


def _bgzf_block(data: bytes) -> bytes:
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    bsize = len(comp) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


def write_bam(path, refs, reads, block_bytes=700):
    """refs: [(name, length)]; reads: dicts with qname, flag, seq, cigar ("20S80M", "" for none), ref, pos.
    Small BGZF blocks on purpose: records straddle block boundaries."""
    text = b"@HD\tVN:1.6\tSO:unsorted\n" + b"".join(b"@SQ\tSN:%s\tLN:%d\n" % (n.encode(), l) for n, l in refs)
    raw = bytearray(b"BAM\x01" + struct.pack("<I", len(text)) + text + struct.pack("<I", len(refs)))
    for n, l in refs:
        raw += struct.pack("<I", len(n) + 1) + n.encode() + b"\x00" + struct.pack("<I", l)
    for r in reads:
        seq = r["seq"]
        cig = []
        num = ""
        for ch in r.get("cigar", ""):
            if ch.isdigit():
                num += ch
            else:
                cig.append((int(num) << 4) | OPS.index(ch))
                num = ""
        packed = bytearray((len(seq) + 1) // 2)
        for i, b in enumerate(seq):
            packed[i >> 1] |= NIB.index(b) << (0 if i & 1 else 4)
        qn = r["qname"] + b"\x00"
        body = struct.pack("<iiBBHHHIiii", r.get("ref", -1), r.get("pos", -1), len(qn), r.get("mapq", 0), 4680, len(cig),
                           r["flag"], len(seq), -1, -1, 0)
        body += qn + b"".join(struct.pack("<I", c) for c in cig) + bytes(packed) + b"\xff" * len(seq)
        raw += struct.pack("<I", len(body)) + body
    out = b"".join(_bgzf_block(bytes(raw[i:i + block_bytes])) for i in range(0, len(raw), block_bytes))
    out += _bgzf_block(b"")                                       # the BGZF end-of-file marker
    open(path, "wb").write(out)
    return out


FAKE_BOWTIE = r'''#!/usr/bin/env python3
"""Stand-in for `bowtie -f -p1 -v0 -m1 -B1 --suppress 5,6,7,8 <genome> -` (detect.rs:61) on tiny test
genomes: exact matches of each FASTA record on both strands, printed only when unique, in input order."""
import sys
args = sys.argv[1:]
assert args[:7] == ["-f", "-p1", "-v0", "-m1", "-B1", "--suppress", "5,6,7,8"] and args[8] == "-", args
genome = {}
name = None
for line in open(args[7] + ".fa", "rb"):
    line = line.rstrip()
    if line.startswith(b">"):
        name = line[1:].split()[0]; genome[name] = []
    else:
        genome[name].append(line)
genome = {k: b"".join(v).upper() for k, v in genome.items()}
comp = bytes.maketrans(b"ACGT", b"TGCA")
out = sys.stdout.buffer
rec = None
def align(name, seq):
    hits = []
    if set(seq) <= set(b"ACGT"):
        rc = seq.translate(comp)[::-1]
        for c, g in genome.items():
            for strand, s in ((b"+", seq), (b"-", rc)):
                i = g.find(s)
                while i >= 0:
                    hits.append((strand, c, i + 1)); i = g.find(s, i + 1)
                if strand == b"+" and rc == seq:
                    break
    if len(hits) == 1:
        out.write(b"%s\t%s\t%s\t%d\n" % (name, hits[0][0], hits[0][1], hits[0][2]))
for line in sys.stdin.buffer:
    line = line.rstrip(b"\n")
    if line.startswith(b">"):
        rec = line[1:]
    elif rec is not None:
        align(rec, line); rec = None
out.flush()
'''


def install_fake_bowtie(directory):
    p = os.path.join(directory, "bowtie")
    open(p, "w").write(FAKE_BOWTIE.replace("#!/usr/bin/env python3", "#!" + sys.executable))
    os.chmod(p, os.stat(p).st_mode | stat.S_IEXEC)
    return p
