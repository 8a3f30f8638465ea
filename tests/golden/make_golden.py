#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ (committed together with this script).

The reference (annalam/breakfast) ships NO tests, fixtures or golden vectors and cannot be built
here, so these are authored for this repository:

  kat_survey   the known-answer vector of SURVEY.md section 8c, whose expected per-read values and
               stdout were derived BY HAND from detect.rs (not from either oracle); both oracles
               must reproduce it.
  hz_*         one fixture per bit-exactness hazard of SURVEY.md section 7 (H1-H10).  Expected
               stdout = the output both independent restatements (oracle/detect_oracle.py, pure
               Python, and oracle/detect_oracle.cpp) agree on; the script asserts that each
               fixture really exercises its hazard.

A fixture is data only: <name>.fa (genome FASTA), <name>.bowtie (bowtie stdout lines),
<name>.args (anchor_len max_frag_len min_evidence), <name>.sv (expected stdout).
Run:  python tests/golden/make_golden.py
"""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import detect_oracle as po  # noqa: E402
import oracle_lib as ol  # noqa: E402


def rand_seq(rng, n):
    return bytes(rng.choice(b"ACGT") for _ in range(n))


def split_read(genome, c1, x1, s1, c2, y1, s2, L, b, A, flip=False):
    """A read whose 5' b bases end on genome[c1][x1] (strand s1) and whose remaining L-b bases start
    on genome[c2][y1] (strand s2), plus the (strand, contig, 1-based pos) bowtie reports per anchor."""
    g1, g2 = genome[c1], genome[c2]
    left = g1[x1 - b + 1:x1 + 1] if s1 else po.revcomp(g1[x1:x1 + b])
    right = g2[y1:y1 + L - b] if s2 else po.revcomp(g2[y1 - (L - b) + 1:y1 + 1])
    seq = left + right
    assert len(seq) == L
    pos5 = x1 - b + 2 if s1 else x1 + b - A + 1
    pos3 = y1 + L - A - b + 1 if s2 else y1 - L + b + 2
    if s1 is False:
        # '-' 5' part: read[k] = comp(g1[x1 + b - 1 - k]) ends on x1
        left = po.revcomp(g1[x1:x1 + b])
    h5, h3 = (s1, c1, pos5), (s2, c2, pos3)
    if flip:
        seq = po.revcomp(seq)
        h5, h3 = (not s2, c2, pos3), (not s1, c1, pos5)
    return seq, h5, h3


class Fixture:
    def __init__(self, name, genome, args):
        self.name, self.genome, self.args = name, genome, args
        self.lines = []
        self.n = 0

    def add(self, frag, seq, h5, h3, drop5=False, drop3=False):
        self.n += 1
        if not drop5:
            self.lines.append(b"5p:%d:\t%s\t%s\t%d" % (self.n, b"+" if h5[0] else b"-", h5[1], h5[2]))
        if not drop3:
            self.lines.append(b"3p:%d:%s:%s:\t%s\t%s\t%d" % (self.n, frag, seq, b"+" if h3[0] else b"-", h3[1], h3[2]))

    def fasta(self):
        out = b""
        for n, s in self.genome.items():
            out += b">" + n + b" synthetic\n"
            out += b"".join(s[i:i + 50] + b"\n" for i in range(0, len(s), 50))
        return out

    def run(self):
        A, F, E = self.args
        fa = self.fasta()
        sv_py = po.detect(fa, self.lines, A, F, E)
        text = b"".join(l + b"\n" for l in self.lines)
        sv_cc = ol.detect_text(ol.OracleGenome.from_fasta(fa), text, A, F, E).sv
        assert sv_py == sv_cc, "the two restatements disagree on fixture %s" % self.name
        return sv_py

    def evidence(self):
        return po.consume_lines(po.parse_fasta(self.fasta()), self.lines, self.args[0])

    def write(self, sv=None):
        sv = self.run() if sv is None else sv
        base = os.path.join(HERE, self.name)
        open(base + ".fa", "wb").write(self.fasta())
        open(base + ".bowtie", "wb").write(b"".join(l + b"\n" for l in self.lines))
        open(base + ".args", "w").write("%d %d %d\n" % self.args)
        open(base + ".sv", "wb").write(sv)
        print("%-22s %3d lines  %2d rows" % (self.name, len(self.lines), sv.count(b"\n") - 1))
        return sv


def kat_survey():
    g = {b"chr1": b"GTCTTTTGGGAATCCCAAGTGCCGCAAGTTATCGGTGCCGCGTGTCCTACCATGACCAGCGAAAGCGTTGAAACGGGATT",
         b"chr2": b"CTCAGGACTGTCTCCTACGTCCACGTTCTAAAATGATGGAGTACCCAGGGCTTGCGCTTAACTTTTGACATTCTCGACAG"}
    f = Fixture("kat_survey", g, (8, 5000, 2))
    f.lines = [b"5p:1:\t+\tchr1\t27", b"3p:1:fragX:AGTTATCGGTGCCGAACTGATGGA:\t+\tchr2\t33",
               b"5p:2:\t-\tchr2\t37", b"3p:2:fragX:GTACTCCATCATTTCGGCACCGAT:\t-\tchr1\t31",
               b"3p:3:fragY:TTATCGGTGCCGAAATGATGGAGT:\t+\tchr2\t35",
               b"5p:4:\t+\tchr1\t29", b"3p:4:fragZ:TTAGCGGTGCCGAAATGATGGAGT:\t+\tchr2\t35"]
    expected = (po.SV_HEADER +
                b"chr1\t+\t40\t\tchr2\t+\t31\t\tAGTTATCGGTGCCG|AAcTGATGGA;TTAgCGGTGCCG|AAATGATGGAGT\t\t\n")
    # hand-derived per-read values
    ev = f.evidence()
    assert [(e.read_num, e.swapped, e.bp, e.pos, e.mpos) for e in ev] == \
        [(1, False, 14, 40, 31), (2, True, 10, 40, 31), (4, False, 12, 40, 31)]
    assert [e.sequence for e in ev] == [b"AGTTATCGGTGCCG|AAcTGATGGA", b"ATCGGTGCCG|AAATGATGGAGTAC",
                                        b"TTAgCGGTGCCG|AAATGATGGAGT"]
    assert all(e.signature == b"CGGTGCCG|AAATGATG" for e in ev)
    assert f.run() == expected, "oracles do not reproduce the hand-checked known-answer vector"
    f.write(expected)


def base_genome(seed, names=(b"chr1", b"chr2"), n=600):
    rng = random.Random(seed)
    return {nm: rand_seq(rng, n) for nm in names}, rng


def hz_tie_lowest_k():
    # H3: microhomology at the junction -> several split points tie; the lowest must win
    g, rng = base_genome(11)
    c1, c2 = bytearray(g[b"chr1"]), bytearray(g[b"chr2"])
    c1[301:305] = c2[200:204]                     # bases right of x1=300 equal the first right bases
    c2[199] = ord("A") if c1[300] != ord("A") else ord("C")   # ... and the base left of y1 differs
    g[b"chr2"] = bytes(c2)
    g[b"chr1"] = bytes(c1)
    f = Fixture("hz_tie_lowest_k", g, (10, 5000, 2))
    for i, b in enumerate((20, 25, 30)):
        f.add(b"t%d" % i, *split_read(g, b"chr1", 300, True, b"chr2", 200, True, 50, b, 10))
    ev = f.evidence()
    assert all(e.bp == b for e, b in zip(ev, (20, 25, 30))), [e.bp for e in ev]   # first minimum = true split
    # shifting the junction right by up to 4 gives the same mismatch count
    f.write()


def hz_len_2a():
    # H3: L == 2A leaves exactly one candidate, bp = A
    g, _ = base_genome(12)
    f = Fixture("hz_len_2a", g, (12, 5000, 2))
    for i in range(3):
        f.add(b"m%d" % i, *split_read(g, b"chr1", 250, True, b"chr2", 111, False, 24, 12, 12, flip=(i == 1)))
    assert all(e.bp == 12 for e in f.evidence())
    f.write()


def hz_contig_ends():
    # H5: windows running off both contig ends ('N' fill), on both strands
    g, _ = base_genome(13, n=120)
    f = Fixture("hz_contig_ends", g, (8, 5000, 1))
    L = 40
    # left window of a '+' anchor runs past the END of chr1; right window ('+') starts before chr2[0]
    f.add(b"e0", *split_read(g, b"chr1", 110, True, b"chr2", 5, True, L, 15, 8))
    f.add(b"e1", *split_read(g, b"chr1", 110, True, b"chr2", 5, True, L, 20, 8, flip=True))
    # '-' strands: left window runs before chr1[0], right window runs past the end of chr2
    f.add(b"e2", *split_read(g, b"chr1", 6, False, b"chr2", 112, False, L, 14, 8))
    f.add(b"e3", *split_read(g, b"chr1", 6, False, b"chr2", 112, False, L, 22, 8, flip=True))
    f.write()


def hz_wrapped_coords():
    # H5: rpos = mpos + A - (L - bp) underflows usize -> printed as 2^64 - x
    g, _ = base_genome(14, n=200)
    f = Fixture("hz_wrapped_coords", g, (8, 5000, 1))
    # 3p anchor at offset 1 on '+': the right window starts 32 bases before the contig ('N' fill).
    # 12 inserted bases that all mismatch the left flank make every split in [20, 31] tie -> bp = 20,
    # rpos = 1 + 8 - (40 - 20) = -11 -> wraps
    ins = bytes((b"A" if c != ord("A") else b"C")[0] for c in g[b"chr1"][70:82])
    seq = g[b"chr1"][50:70] + ins + g[b"chr2"][0:8]
    f.add(b"w0", seq, (True, b"chr1", 51), (True, b"chr2", 1))
    f.add(b"w1", seq, (True, b"chr1", 51), (True, b"chr2", 1))
    # '-' left anchor at offset 1: the left window is 'N' beyond the anchor; read N's there match it,
    # so bp = 12 and lpos = 1 + 8 - 12 = -3 -> wraps
    seq2 = po.revcomp(g[b"chr1"][0:8]) + b"NNNN" + g[b"chr2"][104:132]
    f.add(b"w2", seq2, (False, b"chr1", 1), (True, b"chr2", 125))
    f.add(b"w3", seq2, (False, b"chr1", 1), (True, b"chr2", 125))
    ev = f.evidence()
    assert [e.bp for e in ev] == [20, 20, 12, 12], [e.bp for e in ev]
    assert ev[0].mpos == 2**64 - 11 and ev[2].pos == 2**64 - 3
    sv = f.run()
    assert b"1844674407370955" in sv, "expected a wrapped coordinate in the output"
    f.write(sv)


def hz_ref_alphabet():
    # H1: lowercase, N and IUPAC reference bases never match an uppercase read base; signatures with
    # such bytes compare case-sensitively
    g, _ = base_genome(15)
    c1, c2 = bytearray(g[b"chr1"]), bytearray(g[b"chr2"])
    c1[280:290] = bytes(c1[280:290]).lower()       # soft-masked stretch inside the left flank
    c1[297:299] = b"NN"                            # N inside the signature window
    c2[203:204] = b"R"                             # IUPAC inside the right signature
    g[b"chr1"], g[b"chr2"] = bytes(c1), bytes(c2)
    f = Fixture("hz_ref_alphabet", g, (10, 5000, 2))
    for i, b in enumerate((22, 28, 31)):
        f.add(b"a%d" % i, *split_read(g, b"chr1", 300, True, b"chr2", 200, True, 60, b, 10, flip=(i == 2)))
    # same junction on an unmasked copy of the flank must NOT share the signature
    g2 = dict(g)
    ev = f.evidence()
    assert any(b"N" in e.signature for e in ev) and any(b"R" in e.signature for e in ev)
    assert any(c in e.sequence for e in ev for c in b"acgt")
    f.write()


def hz_read_n_iupac():
    # H2: reads containing N / IUPAC codes, also through the revcomp of a swapped read (R<->Y, M<->K)
    g, _ = base_genome(16)
    f = Fixture("hz_read_n_iupac", g, (10, 5000, 2))
    for i, b in enumerate((20, 26, 33)):
        seq, h5, h3 = split_read(g, b"chr2", 300, True, b"chr1", 200, False, 60, b, 10, flip=(i != 1))
        s = bytearray(seq)
        s[15] = ord("N"); s[30] = ord("R"); s[31] = ord("M"); s[40] = ord("=")
        f.add(b"n%d" % i, bytes(s), h5, h3)
    ev = f.evidence()
    assert any(e.swapped for e in ev) and any(not e.swapped for e in ev)
    assert any(b"y" in e.sequence or b"Y" in e.sequence for e in ev)      # revcomp turned R into Y
    f.write()


def hz_same_chr_swap():
    # H4: same contig, 5p anchor at the higher offset -> swapped; equal offsets -> not swapped
    g, _ = base_genome(17)
    f = Fixture("hz_same_chr_swap", g, (10, 5000, 1))
    f.add(b"s0", *split_read(g, b"chr1", 400, True, b"chr1", 100, True, 50, 25, 10))   # pos > mpos
    f.add(b"s1", *split_read(g, b"chr1", 100, True, b"chr1", 400, True, 50, 25, 10))   # pos < mpos
    seq, h5, h3 = split_read(g, b"chr1", 150, True, b"chr1", 300, True, 50, 20, 10)
    f.add(b"s2", seq, h5, (h3[0], h3[1], h5[2]))                                       # pos == mpos
    ev = f.evidence()
    assert [e.swapped for e in ev] == [True, False, False]
    f.write()


def hz_contig_order():
    # H4: byte-wise contig order chr1 < chr10 < chr2 decides the swap and the output order
    g, _ = base_genome(18, names=(b"chr2", b"chr10", b"chr1"), n=500)
    f = Fixture("hz_contig_order", g, (10, 5000, 2))
    for i in range(2):
        f.add(b"o%d" % i, *split_read(g, b"chr2", 300, True, b"chr10", 200, True, 50, 20 + i, 10))
    for i in range(2):
        f.add(b"p%d" % i, *split_read(g, b"chr10", 100, False, b"chr1", 250, True, 50, 22 + i, 10))
    for i in range(2):
        f.add(b"q%d" % i, *split_read(g, b"chr1", 50, True, b"chr2", 50, True, 50, 24 + i, 10))
    sv = f.run()
    rows = [l.split(b"\t")[0] + b"/" + l.split(b"\t")[4] for l in sv.split(b"\n")[1:-1]]
    # rows sort by (chr, lpos): chr1@50 before chr1@250; chr10 < chr2 byte-wise
    assert rows == [b"chr1/chr2", b"chr1/chr10", b"chr10/chr2"], rows
    f.write(sv)


def hz_two_clusters():
    # H6 (i): `reported[s]` is not checked when adding -> one read can be a member of two clusters.
    # Tandem repeats (period 40) on both sides give reads A, B, X one signature at different positions:
    #   lpos: A < B < X (40 apart);  rpos: A = 0, B = +8000, X = +4000  (max_frag_len 5000)
    # seed A takes X (not B: rpos too far); B is unreported, becomes a seed and takes X again.
    g, _ = base_genome(19, n=30000)
    c1, c2 = bytearray(g[b"chr1"]), bytearray(g[b"chr2"])
    for c, start in ((c1, 1000), (c2, 280)):
        unit = c[start:start + 40]
        for k in range(1, 260):
            c[start + 40 * k:start + 40 * k + 40] = unit
    g[b"chr1"], g[b"chr2"] = bytes(c1), bytes(c2)
    f = Fixture("hz_two_clusters", g, (10, 5000, 2))
    for name, dx, dy in ((b"X", 80, 4000), (b"A", 0, 0), (b"B", 40, 8000)):
        f.add(name, *split_read(g, b"chr1", 1039 + dx, True, b"chr2", 300 + dy, True, 50, 25, 10))
    ev = f.evidence()
    assert len({e.signature for e in ev}) == 1
    sv = f.run()
    rows = sv.split(b"\n")[1:-1]
    assert len(rows) == 2, rows
    x_junction = ev[0].sequence
    assert all(x_junction in r for r in rows), "X must support both clusters"
    f.write(sv)


def hz_dedup_min_evidence():
    # H6 (iii) + H7: a 2-read cluster whose reads share the QNAME passes the first min_evidence test
    # and fails the second; a 3-read cluster survives with 2; representative = longest shorter flank
    g, _ = base_genome(20)
    f = Fixture("hz_dedup_min_evidence", g, (10, 5000, 2))
    f.add(b"fragA", *split_read(g, b"chr1", 300, True, b"chr2", 200, True, 60, 20, 10))
    f.add(b"fragA", *split_read(g, b"chr1", 300, True, b"chr2", 200, True, 60, 30, 10, flip=True))
    # second junction: fragB twice (second has the better score: not the first member), fragC once
    f.add(b"fragB", *split_read(g, b"chr1", 100, True, b"chr2", 400, False, 60, 12, 10))
    f.add(b"fragC", *split_read(g, b"chr1", 100, True, b"chr2", 400, False, 60, 40, 10))
    f.add(b"fragB", *split_read(g, b"chr1", 100, True, b"chr2", 400, False, 60, 29, 10, flip=True))
    # equal scores -> the first stays (strict >)
    f.add(b"fragD", *split_read(g, b"chr1", 450, False, b"chr2", 80, True, 60, 25, 10))
    f.add(b"fragD", *split_read(g, b"chr1", 450, False, b"chr2", 80, True, 60, 35, 10))
    f.add(b"fragE", *split_read(g, b"chr1", 450, False, b"chr2", 80, True, 60, 15, 10))
    sv = f.run()
    rows = sv.split(b"\n")[1:-1]
    assert len(rows) == 2, rows
    f.write(sv)


def hz_mixed_lengths():
    # H9: reads of different lengths in one run (and one longer than 256 bases)
    g, _ = base_genome(21, n=2000)
    f = Fixture("hz_mixed_lengths", g, (10, 5000, 2))
    for i, (L, b) in enumerate(((40, 15), (75, 30), (150, 100), (300, 130), (64, 32), (65, 40), (128, 64), (129, 20))):
        f.add(b"l%d" % i, *split_read(g, b"chr1", 1000, True, b"chr2", 900, True, L, b, 10, flip=(i % 3 == 1)))
    assert len({e.full_len for e in f.evidence()}) == 8
    f.write()


def hz_pairing():
    # detect.rs:79-87: a 3p line is used only with the 5p line of the same read number; a 5p line
    # without its 3p line is simply overwritten; frag ids containing '_' and contigs with ':' pass
    g, _ = base_genome(22, names=(b"chr1", b"HLA:A*01"), n=500)
    f = Fixture("hz_pairing", g, (10, 5000, 1))
    r = split_read(g, b"chr1", 300, True, b"HLA:A*01", 200, True, 50, 20, 10)
    f.add(b"k0", *r, drop3=True)
    f.add(b"k1", *r, drop5=True)           # read number 2: remembered 5p is number 1 -> skipped
    f.add(b"k_2", *r)
    f.add(b"k3", *r, drop5=True)           # remembered 5p is number 3 -> skipped
    f.add(b"k4", *split_read(g, b"chr1", 300, True, b"HLA:A*01", 200, True, 50, 27, 10))
    assert len(f.evidence()) == 2
    f.write()


def hz_empty():
    g, _ = base_genome(23)
    f = Fixture("hz_empty", g, (30, 5000, 2))
    assert f.run() == po.SV_HEADER
    f.write()


def hz_min_evidence_1(E=1):
    # min_evidence = 1: singletons are printed; frag-id duplicates still collapse.
    # min_evidence = 0 (same input): `cluster.len() < 0` is never true (detect.rs:229,231), so the reference
    # prints exactly what it prints for 1 -- asserted here.
    g, _ = base_genome(24)
    f = Fixture("hz_min_evidence_%d" % E, g, (10, 100, E))
    f.add(b"x0", *split_read(g, b"chr1", 300, True, b"chr2", 200, True, 50, 20, 10))
    f.add(b"x1", *split_read(g, b"chr2", 100, False, b"chr1", 50, False, 50, 30, 10))
    f.add(b"x0", *split_read(g, b"chr1", 300, True, b"chr2", 200, True, 50, 24, 10))
    # same class but rpos further than max_frag_len=100 apart -> separate clusters
    f.add(b"x3", *split_read(g, b"chr2", 100, False, b"chr1", 350, False, 50, 30, 10))
    sv = f.write()
    if E == 0:
        assert sv == open(os.path.join(HERE, "hz_min_evidence_1.sv"), "rb").read()


if __name__ == "__main__":
    kat_survey()
    hz_tie_lowest_k()
    hz_len_2a()
    hz_contig_ends()
    hz_wrapped_coords()
    hz_ref_alphabet()
    hz_read_n_iupac()
    hz_same_chr_swap()
    hz_contig_order()
    hz_two_clusters()
    hz_dedup_min_evidence()
    hz_mixed_lengths()
    hz_pairing()
    hz_empty()
    hz_min_evidence_1()
    hz_min_evidence_1(E=0)
