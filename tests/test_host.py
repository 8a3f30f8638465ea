"""CPU tests of the host-side mirror of the consume loop (detect.rs:76-103) and the generator."""
import numpy as np

import detect_oracle as po
from conftest import load_golden

from breakfast_amd import engine, synth


def test_pairing_rule_matches_the_reference_loop():
    fa, lines, (A, F, E), _ = load_golden("hz_pairing")
    g = po.parse_fasta(fa)
    names = list(g.keys())
    b = engine.parse_bowtie_lines(lines.split(b"\n")[:-1], {n: i for i, n in enumerate(names)})
    ev = po.consume_lines(g, lines.split(b"\n")[:-1], A)
    assert b.n == len(ev) == 2
    assert [b.frag(i) for i in range(b.n)] == [e.frag_id for e in ev]


def test_parser_keeps_input_order_and_fields():
    fa, lines, _, _ = load_golden("kat_survey")
    b = engine.parse_bowtie_lines(lines.split(b"\n")[:-1], {b"chr1": 0, b"chr2": 1})
    assert b.n == 3                                   # read 3 has no 5p line (detect.rs:85-87)
    assert [b.frag(i) for i in range(3)] == [b"fragX", b"fragX", b"fragZ"]
    assert b.seq(1) == b"GTACTCCATCATTTCGGCACCGAT"
    h = b.hits[1]
    assert (int(h["pos5"]), int(h["pos3"]), int(h["contig5"]), int(h["contig3"]), int(h["strands"])) == (37, 31, 1, 0, 0)


def test_batch_roundtrips_through_bowtie_text():
    contigs, gk, spec = synth.config(1, scale=0.00003, genome_scale=0.0005)
    g = synth.make_genome(contigs, **gk)
    b = synth.make_reads(g, spec)
    text = synth.bowtie_lines(b, g.names)
    b2 = engine.parse_bowtie_lines(text.split(b"\n")[:-1], {n: i for i, n in enumerate(g.names)})
    assert b2.n == b.n
    assert np.array_equal(b2.seq_bytes, b.seq_bytes) and np.array_equal(b2.frag_bytes, b.frag_bytes)
    for f in ("pos5", "pos3", "contig5", "contig3", "strands"):
        assert np.array_equal(b2.hits[f], b.hits[f])


def test_batch_slice():
    contigs, gk, spec = synth.config(1, scale=0.00002, genome_scale=0.0005)
    g = synth.make_genome(contigs, **gk)
    b = synth.make_reads(g, spec)
    s = b.slice(10, 30)
    assert s.n == 20 and s.seq(0) == b.seq(10) and s.frag(19) == b.frag(29)
    assert int(s.seq_off[0]) == 0 and int(s.frag_off[0]) == 0
