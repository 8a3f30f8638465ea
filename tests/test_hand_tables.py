"""Both restatements of detect.rs (oracle/detect_oracle.py, oracle/detect_oracle.cpp) and the GPU against per-read
tables derived without either of them (tests/golden/derive_tables.py -> tests/golden/hand_tables.json: whole-slice
flanks, position-by-position mismatch vectors, the closed form of mm[k]; one row entirely by hand).

The reference holds no vectors, so parity stays "unpinned by the reference"; these tables are what this repository
can add: a third derivation that shares no code with the two oracles it checks.
"""
import json
import os

import numpy as np
import pytest

import detect_oracle as po
import oracle_lib as ol
from conftest import GOLDEN, load_golden

TABLES = json.load(open(os.path.join(GOLDEN, "hand_tables.json")))
SV_HEADER = (b"CHROM\tSTRAND\tPOSITION\tNEARBY FEATURES\tCHROM\tSTRAND\tPOSITION\tNEARBY FEATURES\t"
             b"SUPPORTING READS\tSIGNATURE\tNOTES\n")


def test_tables_are_what_the_script_derives():
    """The committed JSON is the script's output (so the literal numbers cannot drift from their derivation)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("derive_tables", os.path.join(GOLDEN, "derive_tables.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert {n: mod.derive(n) for n in mod.FIXTURES} == TABLES
    hand = mod.HAND_ROW
    row = TABLES[hand["fixture"]]["reads"][hand["read"]]
    assert all(row[k] == hand[k] for k in ("swapped", "bp", "lpos", "rpos", "junction", "signature"))
    assert row["mm"][:len(hand["mm"])] == hand["mm"]


@pytest.mark.parametrize("name", sorted(TABLES))
def test_python_restatement_against_the_tables(name):
    fa, lines, (A, F, E), _ = load_golden(name)
    ev = po.consume_lines(po.parse_fasta(fa), lines.split(b"\n")[:-1], A)
    rows = TABLES[name]["reads"]
    assert len(ev) == len(rows)
    for e, r in zip(ev, rows):
        assert (e.swapped, e.bp, e.pos, e.mpos) == (r["swapped"], r["bp"], r["lpos"], r["rpos"])
        assert (e.strand, e.mstrand, e.chr, e.mchr) == (r["strand"], r["mstrand"], r["chr"].encode(), r["mchr"].encode())
        assert e.sequence == r["junction"].encode() and e.signature == r["signature"].encode()
        assert min(r["mm"]) == r["mm"][0 if r["bp"] == A else r["bp"] - A] and r["mm"].index(min(r["mm"])) == r["bp"] - A


@pytest.mark.parametrize("name", sorted(TABLES))
def test_cpp_restatement_against_the_tables(name):
    fa, lines, (A, F, E), _ = load_golden(name)
    run = ol.detect_text(ol.OracleGenome.from_fasta(fa), lines, A, F, E, keep=True)
    rows = TABLES[name]["reads"]
    assert run.n_evidence == len(rows)
    assert run.bp.tolist() == [r["bp"] for r in rows]
    assert (run.flags & 4).astype(bool).tolist() == [r["swapped"] for r in rows]
    assert (run.flags & 1).astype(bool).tolist() == [r["strand"] for r in rows]
    assert (run.flags & 2).astype(bool).tolist() == [r["mstrand"] for r in rows]
    assert run.lpos.tolist() == [r["lpos"] for r in rows] and run.rpos.tolist() == [r["rpos"] for r in rows]
    assert [bytes(s) for s in run.signature] == [r["signature"].encode() for r in rows]
    assert [run.junction_of(i) for i in range(len(rows))] == [r["junction"].encode() for r in rows]


def expected_sv(name):
    """A fixture's .sv written out from its table's clusters (seed coordinates as usize, surviving reads): detect.rs:233-240."""
    t = TABLES[name]
    out = [SV_HEADER]
    for c in t["clusters"]:
        s = t["reads"][c["seed"]]
        out.append(b"%s\t%s\t%d\t\t%s\t%s\t%d\t\t%s\t\t\n" % (
            s["chr"].encode(), b"+" if s["strand"] else b"-", s["lpos"], s["mchr"].encode(), b"+" if s["mstrand"] else b"-", s["rpos"],
            b";".join(t["reads"][k]["junction"].encode() for k in c["kept"])))
    return b"".join(out)


@pytest.mark.parametrize("name", sorted(TABLES))
def test_clusters_against_the_tables(name):
    """Which reads form a printed cluster, in which order, and which survive remove_frag_id_duplicates (detect.rs:193-241,
    247-287) -- derived as a fixed point of the compatibility predicate, from fragment ids and scores, not by either
    restatement's loops: the committed .sv and both restatements must print exactly that.  Covers the quirks: a read an
    earlier seed reported joins a later seed's cluster (hz_two_clusters), coordinates printed and sorted as wrapped usize
    (hz_wrapped_coords), unpaired lines (hz_pairing), min_evidence 0 / 1 / 2 with the test applied twice."""
    fa, lines, (A, F, E), sv = load_golden(name)
    want = expected_sv(name)
    assert sv == want
    assert po.detect(fa, lines.split(b"\n")[:-1], A, F, E) == want
    assert ol.detect_text(ol.OracleGenome.from_fasta(fa), lines, A, F, E).sv == want


def sv_from_table(t):
    out = [SV_HEADER]
    for c in t["clusters"]:
        s = t["reads"][c["seed"]]
        out.append(b"%s\t%s\t%d\t\t%s\t%s\t%d\t\t%s\t\t\n" % (
            s["chr"].encode(), b"+" if s["strand"] else b"-", s["lpos"], s["mchr"].encode(), b"+" if s["mstrand"] else b"-", s["rpos"],
            b";".join(t["reads"][k]["junction"].encode() for k in c["kept"])))
    return b"".join(out)


def random_case(seed, tmp_path):
    """A small synthetic case with everything the path distinguishes: both strands, same-contig and cross-contig events,
    mates that share a fragment id, piles on a hot junction, N runs and soft-masked stretches in the genome, N in reads,
    contigs whose names sort unlike their order -> (fasta path, bowtie path, A, F, E)."""
    from breakfast_amd import synth
    rng = np.random.default_rng(seed)
    contigs = [("chr2", 30000), ("chr10", 26000), ("chrX", 22000)]
    g = synth.make_genome(contigs, seed=1000 + seed, end_n=int(rng.integers(0, 40)),
                          n_runs=[(0, 9000, 400), (2, 5000, 120)], lower_runs=[(1, 8000, 5000)])
    A = int(rng.choice([12, 20]))
    L = int(rng.choice([2 * A + 8, 75, 100]))
    spec = synth.ReadSpec(n_reads=int(rng.integers(60, 220)), read_len=L, anchor_len=A, n_events=int(rng.integers(3, 12)),
                          singleton_frac=0.25, mate_frac=0.5, hot_frac=0.2, hot_events=1, subst_rate=0.02, n_rate=0.003,
                          seed=2000 + seed, margin=int(rng.choice([0, 300])))
    b = synth.make_reads(g, spec)
    fa, bt = tmp_path / ("r%d.fa" % seed), tmp_path / ("r%d.bowtie" % seed)
    fa.write_bytes(g.fasta())
    bt.write_bytes(synth.bowtie_lines_slow(b, g.names))
    return str(fa), str(bt), A, int(rng.choice([300, 5000])), int(rng.choice([1, 2, 3]))


@pytest.mark.parametrize("seed", range(64))
def test_independent_derivation_on_random_cases(seed, tmp_path):
    """The derivation of tests/golden/derive_tables.py (flanks as slices, closed-form mm[k], clusters as a fixed point,
    dedup from ids and scores; no code shared with oracle/) against BOTH restatements on random cases, not only on the
    twelve fixtures: per-read rows and the printed table."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("derive_tables", os.path.join(GOLDEN, "derive_tables.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fa, bt, A, F, E = random_case(seed, tmp_path)
    t = mod.derive_files(fa, bt, A, F, E)
    assert len(t["reads"]) >= 50
    fa_bytes, lines = open(fa, "rb").read(), open(bt, "rb").read()
    ev = po.consume_lines(po.parse_fasta(fa_bytes), lines.split(b"\n")[:-1], A)
    assert len(ev) == len(t["reads"])
    for e, r in zip(ev, t["reads"]):
        assert (e.swapped, e.bp, e.pos, e.mpos, e.sequence, e.signature) == (
            r["swapped"], r["bp"], r["lpos"], r["rpos"], r["junction"].encode(), r["signature"].encode())
    want = sv_from_table(t)
    assert po.detect(fa_bytes, lines.split(b"\n")[:-1], A, F, E) == want
    assert ol.detect_text(ol.OracleGenome.from_fasta(fa_bytes), lines, A, F, E).sv == want


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(100, 132))
def test_gpu_against_the_independent_derivation_on_random_cases(seed, tmp_path):
    """The GPU path against the third derivation directly (no restatement in between): the printed table of random cases,
    packed kernel, byte-exact kernel and contig groups."""
    import importlib.util
    from breakfast_amd import engine
    spec = importlib.util.spec_from_file_location("derive_tables", os.path.join(GOLDEN, "derive_tables.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fa, bt, A, F, E = random_case(seed, tmp_path)
    want = sv_from_table(mod.derive_files(fa, bt, A, F, E))
    g = po.parse_fasta(open(fa, "rb").read())
    lines = open(bt, "rb").read().split(b"\n")[:-1]
    for flags in (0, engine.BF_FLAG_FORCE_EXACT, engine.BF_FLAG_GROUPS_ALWAYS, engine.BF_FLAG_GROUPS_ALWAYS | engine.BF_FLAG_FORCE_EXACT,
                  engine.BF_FLAG_SORTED_DEDUP, engine.BF_FLAG_WEAK_DEDUP_HASH | engine.BF_FLAG_GROUPS_ALWAYS, engine.BF_FLAG_ONE_STREAM):
        assert engine.detect(g, lines, A, F, E, flags=flags) == want, flags
        try:                                             # the reads as 4-bit codes (lower-case reads are outside BAM's alphabet)
            assert engine.detect(g, lines, A, F, E, flags=flags, packed=True) == want, ("packed", flags)
        except ValueError:
            pass


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(TABLES))
def test_gpu_against_the_tables(name):
    from breakfast_amd import engine
    fa, lines, (A, F, E), sv = load_golden(name)
    rows = TABLES[name]["reads"]
    g = po.parse_fasta(fa)
    names = list(g.keys())
    for flags in (0, engine.BF_FLAG_FORCE_EXACT):
        eng = engine.Engine(A, F, E, flags=flags | engine.BF_FLAG_KEEP_EVIDENCE)
        try:
            eng.set_genome(names, [np.frombuffer(g[n], np.uint8) for n in names])
            b = engine.parse_bowtie_lines(lines.split(b"\n")[:-1], eng.contig_id)
            eng.submit(b)
            res = eng.finish()
            ev = eng.evidence()
            assert ev["bp"].tolist() == [r["bp"] for r in rows]
            assert ((ev["flags"] & 4) != 0).tolist() == [r["swapped"] for r in rows]
            assert ((ev["flags"] & 1) != 0).tolist() == [r["strand"] for r in rows]
            assert ((ev["flags"] & 2) != 0).tolist() == [r["mstrand"] for r in rows]
            assert ev["lpos"].astype(np.int64).astype(np.uint64).tolist() == [r["lpos"] for r in rows]
            assert ev["rpos"].astype(np.int64).astype(np.uint64).tolist() == [r["rpos"] for r in rows]
            by, off = eng.render_junctions(np.arange(b.n, dtype=np.uint32))
            text = by.tobytes()
            assert [text[int(off[i]):int(off[i + 1])] for i in range(b.n)] == [r["junction"].encode() for r in rows]
            assert eng.format_sv(res) == expected_sv(name)
        finally:
            eng.close()
