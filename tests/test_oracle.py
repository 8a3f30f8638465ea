"""CPU tests of the oracle itself: both restatements against the golden vectors and each other."""
import numpy as np
import pytest

import detect_oracle as po
import oracle_lib as ol
from conftest import golden_names, load_golden

from breakfast_amd import synth


@pytest.mark.parametrize("name", golden_names())
def test_python_oracle_reproduces_golden(name):
    fa, lines, (A, F, E), sv = load_golden(name)
    assert po.detect(fa, lines.split(b"\n")[:-1], A, F, E) == sv


@pytest.mark.parametrize("name", golden_names())
def test_cpp_oracle_reproduces_golden(name):
    fa, lines, (A, F, E), sv = load_golden(name)
    run = ol.detect_text(ol.OracleGenome.from_fasta(fa), lines, A, F, E)
    assert run.sv == sv


def test_known_answer_vector_per_read():
    # hand-derived per-read values
    fa, lines, (A, F, E), _ = load_golden("kat_survey")
    run = ol.detect_text(ol.OracleGenome.from_fasta(fa), lines, A, F, E, keep=True)
    assert run.n_evidence == 3
    assert run.bp.tolist() == [14, 10, 12]
    assert (run.flags & 4).astype(bool).tolist() == [False, True, False]
    assert run.lpos.tolist() == [40, 40, 40] and run.rpos.tolist() == [31, 31, 31]
    assert [bytes(s) for s in run.signature] == [b"CGGTGCCG|AAATGATG"] * 3
    assert run.junction_of(1) == b"ATCGGTGCCG|AAATGATGGAGTAC"


def test_complement_table():
    # bio 0.39 alphabets::dna: identity except the IUPAC letters, both cases
    for a, b in zip(b"AGCTYRWSKMDVHBN", b"TCGARYWSMKHBDVN"):
        assert po.complement(a) == b and po.complement(a + 32) == b + 32
    for c in b"=*-.XUxu0 ":
        assert po.complement(c) == c
    assert po.revcomp(b"ACGTNRYacgt=") == b"=acgtRYNACGT"


def test_fasta_parsing_rules():
    g = po.parse_fasta(b">c1 desc here\nACgt \r\nNN\n>c2\tx\n\nTT\n>c1\nGG\n")
    assert g == {b"c1": b"GG", b"c2": b"TT"}          # id = first word; last duplicate wins
    assert po.parse_fasta(b">only\nAC\nGT") == {b"only": b"ACGT"}


@pytest.mark.parametrize("cfg", [0, 1, 3, 4])
def test_restatements_agree_on_synthetic(cfg):
    contigs, gk, spec = synth.config(cfg, scale=0.0002 if cfg else 0.01, genome_scale=0.0005)
    spec.n_reads = min(spec.n_reads, 1500)
    spec.n_events = min(spec.n_events, 40)
    g = synth.make_genome(contigs, **gk)
    b = synth.make_reads(g, spec)
    lines = synth.bowtie_lines(b, g.names)
    A = spec.anchor_len
    og = ol.OracleGenome(g.names, g.seqs)
    text = ol.detect_text(og, lines, A, 5000, 2, keep=True)
    batch = ol.detect_batch(og, g.names, b, A, 5000, 2, keep=True)
    assert text.sv == batch.sv and np.array_equal(text.bp, batch.bp)
    assert po.detect(g.fasta(), lines.split(b"\n")[:-1], A, 5000, 2) == text.sv
    assert text.n_clusters > 0


def test_synthetic_hits_are_self_consistent():
    # error-free reads: the scan must find a split with zero mismatches at or before the simulated one
    contigs, gk, spec = synth.config(1, scale=0.0001, genome_scale=0.0005)
    spec.subst_rate = 0.0
    g = synth.make_genome(contigs, **gk)
    b = synth.make_reads(g, spec)
    og = ol.OracleGenome(g.names, g.seqs)
    run = ol.detect_batch(og, g.names, b, spec.anchor_len, 5000, 2, keep=True)
    L, A = spec.read_len, spec.anchor_len
    for i in range(run.n_evidence):
        j = run.junction_of(i)
        if not j.replace(b"|", b"").isupper():
            # only possible when the presented split is L-A, one past the last candidate (detect.rs:149)
            assert run.bp[i] >= L - A - 8, (i, j)
