"""CPU tests of `breakfast-hip filter` / `annotate` (text_tools.cpp = src/filter.rs, src/annotate.rs; SURVEY.md
section 8f rows 3-4) against hand-derived expectations and the pure-Python restatements in oracle/.
The reference has no tests or fixtures for these tools: parity is unpinned by the reference."""
import gzip
import os
import random
import subprocess

import pytest

import annotate_oracle as ao
import filter_oracle as fo
from conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "breakfast_amd", "csrc", "breakfast-hip")
HEADER = (b"CHROM\tSTRAND\tPOSITION\tNEARBY FEATURES\tCHROM\tSTRAND\tPOSITION\tNEARBY FEATURES\t"
          b"SUPPORTING READS\tSIGNATURE\tNOTES")


def row(c1, s1, p1, c2, s2, p2, reads, sig=b"", notes=b""):
    return b"\t".join([c1, s1, b"%d" % p1, b"", c2, s2, b"%d" % p2, b"", b";".join(reads), sig, notes])


def run_filter(cwd, sheet="sheet.tsv", *opts):
    return subprocess.run([EXE, "filter", *opts, sheet], cwd=cwd, capture_output=True, timeout=60)


def rnd_read(rng, n=41):
    return bytes(rng.choice(b"ACGTacgt") for _ in range(n // 2)) + b"|" + bytes(rng.choice(b"ACGTacgt") for _ in range(n - n // 2 - 1))


# ---- filter ---------------------------------------------------------------------------------------

def test_filter_by_hand(tmp_path):
    rng = random.Random(1)
    r = [rnd_read(rng) for _ in range(8)]
    dup_tail = r[0][:19] + b"TT" + r[1][21:]           # same last 20 bases as r[1], different head
    dup_head = r[2][:25] + b"GGGG" + r[3][29:]         # same first 20 bases as r[2], different tail
    tumor = [
        row(b"chr1", b"+", 1000, b"chr2", b"-", 500, r[:5]),                        # 5 reads, no control evidence -> kept, reads sorted
        row(b"chr1", b"+", 1000, b"chr1", b"+", 1005, r[:6]),                       # same chr+strand, 5 apart < 10 -> dropped
        row(b"chr1", b"+", 1000, b"chr1", b"-", 1005, r[:4]),                       # adjacent but strands differ -> kept
        row(b"chr3", b"-", 77, b"chr4", b"-", 99, r[:3]),                           # 3 reads < min-reads 4 -> dropped
        row(b"chr5", b"+", 10, b"chr6", b"+", 20, [r[1], dup_tail, r[2], dup_head, r[4]]),   # 5 -> 3 after merging -> dropped
        row(b"chr7", b"+", 10, b"chr8", b"+", 20, r[:8], b"sig", b"note"),          # control holds 1 read: needs >= 50 -> dropped
        row(b"chr9", b"+", 10, b"chr9", b"+", 20, r[:4], b"", b"extra\tcolumns"),   # exactly min-distance apart -> kept, 12 columns survive
    ]
    normal = [row(b"chr7", b"+", 10, b"chr8", b"+", 20, r[:1])]
    other = [row(b"chr1", b"+", 1000, b"chr2", b"-", 501, r[:6])]                    # a different rearrangement (pos differs)
    (tmp_path / "T1.sv").write_bytes(b"\n".join([HEADER] + tumor) + b"\n")
    (tmp_path / "N1.sv").write_bytes(b"\n".join([HEADER] + normal) + b"\n")
    (tmp_path / "T2.sv").write_bytes(b"\n".join([HEADER] + other) + b"\n")
    (tmp_path / "sheet.tsv").write_bytes(b"SAMPLE\tPATIENT\tTYPE\tCOMMENT\nT1\tP1\ttumor\tx\n\nN1 \tP1\t WBC\ty\nT2\tP2\tTumor\n")
    p = run_filter(tmp_path)
    assert p.returncode == 0, p.stderr.decode()
    got = (tmp_path / "annotated" / "T1.sv").read_bytes()
    want = b"\n".join([HEADER,
                       row(b"chr1", b"+", 1000, b"chr2", b"-", 500, sorted(r[:5])),
                       row(b"chr1", b"+", 1000, b"chr1", b"-", 1005, sorted(r[:4])),
                       row(b"chr9", b"+", 10, b"chr9", b"+", 20, sorted(r[:4]), b"", b"extra\tcolumns")]) + b"\n"
    assert got == want
    # the germline sample of P1 has only T2 (another patient) as control; its one-read call is < min-reads
    assert (tmp_path / "annotated" / "N1.sv").read_bytes() == HEADER + b"\n"
    assert (tmp_path / "annotated" / "T2.sv").read_bytes() == b"\n".join([HEADER, row(b"chr1", b"+", 1000, b"chr2", b"-", 501, sorted(r[:6]))]) + b"\n"
    # the oracle agrees with the hand-derived expectation
    ref = fo.filter_cohort(str(tmp_path / "sheet.tsv"), cwd=str(tmp_path))
    assert ref[os.path.join("annotated", "T1.sv")] == want
    assert b"Output files will be written in directory annotated" in p.stderr


def test_filter_options_and_keep_duplicates(tmp_path):
    rng = random.Random(2)
    r = [rnd_read(rng) for _ in range(3)]
    same = [r[0], r[0], r[1], r[1], r[2]]
    (tmp_path / "A.sv").write_bytes(b"\n".join([HEADER, row(b"chr1", b"+", 5, b"chr2", b"+", 9, same)]) + b"\n")
    (tmp_path / "B.sv").write_bytes(HEADER + b"\n")
    (tmp_path / "sheet.tsv").write_bytes(b"PATIENT\tSAMPLE\tTYPE\nP1\tA\ttumor\nP2\tB\tgDNA\n")
    p = run_filter(tmp_path, "sheet.tsv", "--keep-duplicates", "--min-reads=5", "--output-dir=out/deep")
    assert p.returncode == 0, p.stderr.decode()
    assert (tmp_path / "out" / "deep" / "A.sv").read_bytes() == b"\n".join([HEADER, row(b"chr1", b"+", 5, b"chr2", b"+", 9, same)]) + b"\n"
    p = run_filter(tmp_path, "sheet.tsv", "--min-reads", "3", "--output-dir=o2")           # docopt's "--opt VALUE" form
    assert p.returncode == 0, p.stderr.decode()
    assert (tmp_path / "o2" / "A.sv").read_bytes() == b"\n".join([HEADER, row(b"chr1", b"+", 5, b"chr2", b"+", 9, sorted(r))]) + b"\n"


def random_cohort(tmp_path, seed):
    rng = random.Random(seed)
    pool = [rnd_read(rng, rng.choice([41, 60, 101])) for _ in range(30)]
    # near-duplicates sharing a head or a tail
    for _ in range(15):
        a, b = rng.choice(pool), rng.choice(pool)
        pool.append(a[:20] + b[20:] if rng.random() < 0.5 else b[:-20] + a[-20:])
    sites = [(rng.choice([b"chr1", b"chr2", b"chr10"]), rng.choice([b"+", b"-"]), rng.randrange(1, 300),
              rng.choice([b"chr1", b"chr2", b"chr10"]), rng.choice([b"+", b"-"]), rng.randrange(1, 300)) for _ in range(60)]
    samples = []
    for pi in range(3):
        for si, ty in enumerate(rng.sample(["tumor", "WBC", "cfDNA", "gdna", "Germline"], rng.randrange(1, 4))):
            name = "P%dS%d" % (pi, si)
            samples.append((name, "P%d" % pi, ty))
            rows = []
            for _ in range(rng.randrange(0, 14)):
                s = rng.choice(sites)
                k = rng.choice([1, 2, 4, 6, 60, 120])
                rows.append(row(*s, [rng.choice(pool) for _ in range(k)], b"", rng.choice([b"", b"n"])))
            (tmp_path / (name + ".sv")).write_bytes(b"\n".join([HEADER] + rows) + (b"\n" if rng.random() < 0.8 or not rows else b""))
    sheet = b"TYPE\tX\tSAMPLE\tPATIENT\n" + b"".join(b"%s\tx\t%s\t%s\n" % (t.encode(), n.encode(), p.encode()) for n, p, t in samples)
    (tmp_path / "sheet.tsv").write_bytes(sheet)
    return [n for n, _, _ in samples]


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("opts", [(), ("--keep-duplicates",), ("--min-reads=2", "--min-distance=40")])
def test_filter_matches_oracle_on_random_cohorts(tmp_path, seed, opts):
    names = random_cohort(tmp_path, seed)
    kw = dict(keep_duplicates="--keep-duplicates" in opts)
    for o in opts:
        if o.startswith("--min-reads="):
            kw["min_reads"] = int(o.split("=")[1])
        if o.startswith("--min-distance="):
            kw["min_distance"] = int(o.split("=")[1])
    ref = fo.filter_cohort(str(tmp_path / "sheet.tsv"), cwd=str(tmp_path), **kw)
    p = run_filter(tmp_path, "sheet.tsv", *opts)
    assert p.returncode == 0, p.stderr.decode()
    assert len(ref) == len(names)
    kept = 0
    for path, want in ref.items():
        got = (tmp_path / path).read_bytes()
        assert got == want, path
        kept += want.count(b"\n") - 1
    assert kept > 0


@pytest.mark.parametrize("sheet,files,msg", [
    (b"SAMPLE\tPATIENT\nA\tP\n", {}, b"ERROR: Required headers PATIENT, SAMPLE and TYPE"),
    (b"SAMPLE\tPATIENT\tTYPE\nA\tP\n", {}, b"ERROR: Sample sheet line has too few columns"),
    (b"SAMPLE\tPATIENT\tTYPE\nA\t \ttumor\n", {}, b"ERROR: Missing PATIENT, SAMPLE or TYPE column"),
    (b"SAMPLE\tPATIENT\tTYPE\nA\tP\ttumor\n", {}, b"ERROR: Cannot open file A.sv for reading."),
    (b"SAMPLE\tPATIENT\tTYPE\nA\tP\ttumor\n", {"A.sv": HEADER + b"\n"}, b"ERROR: No negative controls found for sample A."),
    (b"SAMPLE\tPATIENT\tTYPE\nA\tP\ttumor\nB\tQ\ttumor\n", {"A.sv": HEADER + b"\nchr1\t+\t5\n", "B.sv": HEADER + b"\n"},
     b"ERROR: File A.sv has a line with too few columns"),
    (b"SAMPLE\tPATIENT\tTYPE\nA\tP\ttumor\nB\tQ\ttumor\n",
     {"A.sv": HEADER + b"\n" + row(b"chr1", b"+", 5, b"chr2", b"+", 9, [b"ACGT|ACGT"]) + b"\n", "B.sv": HEADER + b"\n"},
     b"ERROR: Supporting read shorter than 20 bases"),
])
def test_filter_errors(tmp_path, sheet, files, msg):
    (tmp_path / "sheet.tsv").write_bytes(sheet)
    for n, b in files.items():
        (tmp_path / n).write_bytes(b)
    p = run_filter(tmp_path)
    assert p.returncode == 255
    assert p.stderr[p.stderr.index(b"ERROR"):].startswith(msg), p.stderr


def test_filter_reads_gz_sample_sheet_and_stdin(tmp_path):
    names = random_cohort(tmp_path, 11)
    ref = fo.filter_cohort(str(tmp_path / "sheet.tsv"), cwd=str(tmp_path))
    sheet = (tmp_path / "sheet.tsv").read_bytes()
    (tmp_path / "sheet.tsv.gz").write_bytes(gzip.compress(sheet))
    p = run_filter(tmp_path, "sheet.tsv.gz", "--output-dir=gz")
    assert p.returncode == 0, p.stderr.decode()
    p2 = subprocess.run([EXE, "filter", "--output-dir=si", "-"], cwd=tmp_path, input=sheet, capture_output=True, timeout=60)
    assert p2.returncode == 0, p2.stderr.decode()
    for path, want in ref.items():
        base = os.path.basename(path)
        assert (tmp_path / "gz" / base).read_bytes() == want
        assert (tmp_path / "si" / base).read_bytes() == want
    assert len(names) == len(ref)


# ---- annotate -------------------------------------------------------------------------------------

def test_annotate_by_hand(tmp_path):
    bed = (b"chr1\t999\t2000\tGENE_A\tignored\n"      # 1-based [1000, 2000]
           b"chr1\t2500\t2600\tGENE_B\n"              # [2501, 2600]
           b"chr2\t0\t10\tGENE_C\n"                   # [1, 10]
           b"chr1\t102100\t102200\tFAR\n"             # [102101, 102200]: 100101 from 2000 -> too far; 100000 from 2101
           b"chr1\t1499\t1501\tGENE_D\n")             # [1500, 1501] inside GENE_A: distance 0 too, BED order breaks the tie
    sv = b"\n".join([HEADER,
                     row(b"chr1", b"+", 1500, b"chr2", b"-", 110, [b"ACGT|ACGT"], b"S", b"N"),
                     b"# a comment line is skipped",
                     row(b"scaffold7", b"+", 5, b"chr1", b"+", 5, [b"AC|GT"]),            # does not start with "chr": skipped
                     row(b"chr1", b"-", 2101, b"chrUn", b"+", 7, [b"AC|GT"], b"", b""),
                     b"chr2\t+\t0011\told\tchr1\t-\t+2000\told\tR\tS\tN\tdropped column"]) + b"\n"
    (tmp_path / "in.sv").write_bytes(sv)
    (tmp_path / "f.bed").write_bytes(bed)
    p = subprocess.run([EXE, "annotate", str(tmp_path / "in.sv"), str(tmp_path / "f.bed")], capture_output=True, timeout=60)
    assert p.returncode == 0, p.stderr.decode()
    want = b"\n".join([
        HEADER,
        b"chr1\t+\t1500\tGENE_A (0), GENE_D (0), GENE_B (1001)\tchr2\t-\t110\tGENE_C (100)\tACGT|ACGT\tS\tN",
        b"chr1\t-\t2101\tGENE_A (101), GENE_B (400), GENE_D (600), FAR (100000)\tchrUn\t+\t7\t\tAC|GT\t\t",
        b"chr2\t+\t11\tGENE_C (1)\tchr1\t-\t2000\tGENE_A (0), GENE_D (499), GENE_B (501)\tR\tS\tN"]) + b"\n"
    assert p.stdout == want
    assert ao.annotate(str(tmp_path / "in.sv"), str(tmp_path / "f.bed")) == want


@pytest.mark.parametrize("seed", range(5))
def test_annotate_matches_oracle_on_random_input(tmp_path, seed):
    rng = random.Random(100 + seed)
    chroms = [b"chr1", b"chr2", b"chrX", b"chr10"]
    bed = []
    for i in range(200):
        s = rng.randrange(0, 1_000_000)
        bed.append(b"%s\t%d\t%d\tG%d" % (rng.choice(chroms), s, s + rng.randrange(0, 50_000), i))
    rows = [row(rng.choice(chroms + [b"contig9"]), rng.choice([b"+", b"-"]), rng.randrange(0, 1_200_000), rng.choice(chroms),
                rng.choice([b"+", b"-"]), rng.randrange(0, 1_200_000), [b"ACGT|acgt"], b"s", b"n") for _ in range(150)]
    (tmp_path / "in.sv").write_bytes(b"\n".join([HEADER] + rows) + b"\n")
    (tmp_path / "f.bed").write_bytes(b"\n".join(bed) + b"\n")
    p = subprocess.run([EXE, "annotate", str(tmp_path / "in.sv"), str(tmp_path / "f.bed")], capture_output=True, timeout=60)
    assert p.returncode == 0, p.stderr.decode()
    assert p.stdout == ao.annotate(str(tmp_path / "in.sv"), str(tmp_path / "f.bed"))
    assert p.stdout.count(b"(") > 100


def test_annotate_errors(tmp_path):
    (tmp_path / "in.sv").write_bytes(HEADER + b"\n")
    (tmp_path / "bad.bed").write_bytes(b"chr1\t5\t9\n")
    p = subprocess.run([EXE, "annotate", str(tmp_path / "in.sv"), str(tmp_path / "bad.bed")], capture_output=True, timeout=60)
    assert p.returncode == 255 and b"ERROR: Feature BED file has too few columns on this line:" in p.stderr
    p = subprocess.run([EXE, "annotate", str(tmp_path / "in.sv")], capture_output=True, timeout=60)
    assert p.returncode == 255 and p.stderr.startswith(b"ERROR: Invalid arguments.")
    p = subprocess.run([EXE, "annotate", str(tmp_path / "nope.sv"), str(tmp_path / "bad.bed")], capture_output=True, timeout=60)
    assert p.returncode == 255 and b"ERROR: Cannot open file" in p.stderr


# ---- the chain on detect's own output format -------------------------------------------------------

def test_detect_output_feeds_filter_and_annotate(tmp_path):
    """The golden .sv of the survey's known-answer vector (what detect prints) goes through filter and
    annotate: a regression on the wire format (columns 0,1,2,4,5,6,8 parsed back, filter.rs:139-146)."""
    _, _, _, sv = load_golden("kat_survey")
    (tmp_path / "T.sv").write_bytes(sv)
    (tmp_path / "N.sv").write_bytes(sv.split(b"\n")[0] + b"\n")
    (tmp_path / "O.sv").write_bytes(sv.split(b"\n")[0] + b"\n")
    # N (germline of the same patient) needs a control of its own: a sample of another patient
    (tmp_path / "sheet.tsv").write_bytes(b"PATIENT\tSAMPLE\tTYPE\nP\tT\ttumor\nP\tN\twbc\nQ\tO\ttumor\n")
    p = run_filter(tmp_path, "sheet.tsv", "--min-reads=2")
    assert p.returncode == 0, p.stderr.decode()
    got = (tmp_path / "annotated" / "T.sv").read_bytes()
    assert got == fo.filter_cohort(str(tmp_path / "sheet.tsv"), min_reads=2, cwd=str(tmp_path))[os.path.join("annotated", "T.sv")]
    lines = got.split(b"\n")
    assert len(lines) == 3 and lines[1].startswith(b"chr1\t+\t40\t\tchr2\t+\t31\t\t")
    assert lines[1].split(b"\t")[8] == b"AGTTATCGGTGCCG|AAcTGATGGA;TTAgCGGTGCCG|AAATGATGGAGT"     # already sorted
    (tmp_path / "genes.bed").write_bytes(b"chr1\t10\t30\tLEFT\nchr2\t30\t50\tRIGHT\n")
    p = subprocess.run([EXE, "annotate", "annotated/T.sv", "genes.bed"], cwd=tmp_path, capture_output=True, timeout=60)
    assert p.returncode == 0, p.stderr.decode()
    assert p.stdout.split(b"\n")[1] == (b"chr1\t+\t40\tLEFT (10)\tchr2\t+\t31\tRIGHT (0)\t"
                                       b"AGTTATCGGTGCCG|AAcTGATGGA;TTAgCGGTGCCG|AAATGATGGAGT\t\t")
