"""`breakfast-hip detect` on the GPU -> `filter` -> `annotate`, the reference's whole tool chain (src/main.rs:25-27), in
one test the driver's GPU run executes: detect's stdout of a tumor sample and of its matched normal go through the
cohort filter (filter.rs:83-156) and the BED annotation (annotate.rs:52-92); every stage is compared with its oracle
(detect_oracle for detect.rs:76-287, filter_oracle.py, annotate_oracle.py) fed with the same files."""
import os
import subprocess

import numpy as np
import pytest

import annotate_oracle as ao
import detect_oracle as po
import filter_oracle as fo

from breakfast_amd import engine, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "breakfast_amd", "csrc", "breakfast-hip")


def hits_text(b, names):
    return synth.bowtie_lines(b, names)


def test_detect_filter_annotate_chain(tmp_path):
    A = 20
    contigs = [("chr1", 90000), ("chr10", 60000), ("chr2", 70000), ("scaffold_9", 30000)]
    g = synth.make_genome(contigs, seed=81)
    prefix = str(tmp_path / "g")
    open(prefix + ".fa", "wb").write(g.fasta())
    # tumor: 60 junctions at coverage ~30; normal: reads of the first junctions only, a few each, so that some
    # tumor calls meet the 50x negative-control rule of filter.rs:120-128 and others fail it
    spec = synth.ReadSpec(n_reads=4000, read_len=100, anchor_len=A, n_events=60, singleton_frac=0.2, mate_frac=0.3,
                          seed=83, margin=300)
    tumor = synth.make_reads(g, spec)
    spec_n = synth.ReadSpec(n_reads=300, read_len=100, anchor_len=A, n_events=60, singleton_frac=0.5, mate_frac=0.0,
                            seed=83, margin=300)
    spec_n.read_seed = 991                      # same junction table (seed), other reads
    normal = synth.make_reads(g, spec_n)
    svs = {}
    for name, b in (("T1", tumor), ("N1", normal), ("T2", tumor.slice(0, 50))):
        hits = str(tmp_path / (name + ".hits"))
        open(hits, "wb").write(hits_text(b, g.names))
        p = subprocess.run([EXE, "detect", "--anchor-len=%d" % A, "--anchor-hits=" + hits, "unused.bam", prefix],
                           capture_output=True, timeout=120)
        assert p.returncode == 0, p.stderr.decode()
        want = po.detect(g.fasta(), open(hits, "rb").read().split(b"\n")[:-1], A, 5000, 2)
        assert p.stdout == want, "detect stage differs from the oracle for %s" % name
        (tmp_path / (name + ".sv")).write_bytes(p.stdout)
        svs[name] = p.stdout
    assert svs["T1"].count(b"\n") > 40 and svs["N1"].count(b"\n") > 3
    (tmp_path / "sheet.tsv").write_bytes(b"SAMPLE\tPATIENT\tTYPE\nT1\tP1\ttumor\nN1\tP1\tnormal\nT2\tP2\ttumor\n")
    p = subprocess.run([EXE, "filter", "--min-reads=3", "sheet.tsv"], cwd=str(tmp_path), capture_output=True, timeout=120)
    assert p.returncode == 0, p.stderr.decode()
    ref = fo.filter_cohort(str(tmp_path / "sheet.tsv"), min_reads=3, cwd=str(tmp_path))
    kept = 0
    for name in ("T1", "N1", "T2"):
        got = (tmp_path / "annotated" / (name + ".sv")).read_bytes()
        assert got == ref[os.path.join("annotated", name + ".sv")], "filter stage differs from the oracle for %s" % name
        kept += got.count(b"\n") - 1
    t1 = (tmp_path / "annotated" / "T1.sv").read_bytes()
    assert 5 < t1.count(b"\n") - 1 < svs["T1"].count(b"\n") - 1          # the control removed some calls, not all
    # annotate: genes tiled over the contigs; scaffold_9 rows are not annotated (annotate.rs:53)
    rng = np.random.default_rng(5)
    bed = []
    for cname, clen in contigs[:3]:
        for k, s in enumerate(sorted(rng.integers(0, clen - 2000, size=12).tolist())):
            bed.append(b"%s\t%d\t%d\tGENE_%s_%d" % (cname.encode(), s, s + int(rng.integers(1, 2000)), cname.encode(), k))
    (tmp_path / "genes.bed").write_bytes(b"\n".join(bed) + b"\n")
    sv_path = str(tmp_path / "annotated" / "T1.sv")
    p = subprocess.run([EXE, "annotate", sv_path, str(tmp_path / "genes.bed")], capture_output=True, timeout=120)
    assert p.returncode == 0, p.stderr.decode()
    assert p.stdout == ao.annotate(sv_path, str(tmp_path / "genes.bed"))
    assert p.stdout.count(b"GENE_") > 10
