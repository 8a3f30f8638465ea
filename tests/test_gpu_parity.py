"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, bit-exact.

Integer/byte work only: every comparison is equality, no tolerance.
"""
import numpy as np
import pytest

import detect_oracle as po
import oracle_lib as ol
from conftest import golden_names, load_golden

from breakfast_amd import engine, synth
from breakfast_amd.engine import (BF_FLAG_FORCE_EXACT, BF_FLAG_KEEP_EVIDENCE, BF_FLAG_SORTED_DEDUP, BF_FLAG_WEAK_DEDUP_HASH,
                                  BF_FLAG_GROUPS_ALWAYS, BF_FLAG_ONE_GROUP)

pytestmark = pytest.mark.gpu


def run_gpu_text(fa, lines, A, F, E, flags=0, packed=False):
    return engine.detect(po.parse_fasta(fa), lines.split(b"\n")[:-1], A, F, E, flags=flags, packed=packed)


@pytest.mark.parametrize("name", golden_names())
def test_golden_fixture(name):
    fa, lines, (A, F, E), sv = load_golden(name)
    assert run_gpu_text(fa, lines, A, F, E) == sv


@pytest.mark.parametrize("flags", [0, BF_FLAG_FORCE_EXACT, BF_FLAG_GROUPS_ALWAYS])
@pytest.mark.parametrize("name", golden_names())
def test_golden_fixture_with_reads_as_4bit_codes(name, flags):
    # bf_submit_packed: the reads as BAM 4-bit base codes (odd lengths, '=' and IUPAC letters, reads above 256 bases)
    fa, lines, (A, F, E), sv = load_golden(name)
    try:
        out = run_gpu_text(fa, lines, A, F, E, flags=flags, packed=True)
    except ValueError:
        # hz_ref_alphabet's reads were cut out of a soft-masked genome: lower-case letters are outside BAM's
        # alphabet, such reads go through bf_submit as ASCII (engine.pack_batch refuses them)
        assert name == "hz_ref_alphabet"
        pytest.skip("reads with bytes outside =ACMGRSVTWYHKDBN are handed over as ASCII")
    assert out == sv


@pytest.mark.parametrize("name", golden_names())
def test_golden_fixture_in_contig_groups(name):
    # the cluster stage a few contig ranks at a time (what inputs of 4M reads and more get): same bytes
    fa, lines, (A, F, E), sv = load_golden(name)
    assert run_gpu_text(fa, lines, A, F, E, flags=BF_FLAG_GROUPS_ALWAYS) == sv


@pytest.mark.parametrize("name", golden_names())
def test_golden_fixture_exact_kernel(name):
    # the byte-exact kernel alone must also reproduce every fixture
    fa, lines, (A, F, E), sv = load_golden(name)
    assert run_gpu_text(fa, lines, A, F, E, flags=BF_FLAG_FORCE_EXACT) == sv


def pack_sig(sig17: np.ndarray) -> np.ndarray:
    """oracle signature bytes -> the 32-bit packing of the kernels (plain ACGT signatures)."""
    s = np.concatenate([sig17[:, :8], sig17[:, 9:]], axis=1).astype(np.uint32)
    code = (s >> 1) & 3
    out = np.zeros(s.shape[0], np.uint32)
    for k in range(8):
        out |= ((code[:, k] & 1) << k) | ((code[:, k] >> 1) << (8 + k))
        out |= ((code[:, 8 + k] & 1) << (16 + k)) | ((code[:, 8 + k] >> 1) << (24 + k))
    return out


def check_against_oracle(g, b, A, F=5000, E=2, flags=0, expect_exact=None, packed=False):
    og = ol.OracleGenome(g.names, g.seqs)
    ref = ol.detect_batch(og, g.names, b, A, F, E, keep=True)
    eng = engine.Engine(A, F, E, flags=flags | BF_FLAG_KEEP_EVIDENCE)
    try:
        eng.set_genome(g.names, g.seqs)
        eng.submit(engine.pack_batch(b) if packed else b)
        # what bf_finish itself brings to the host: rows, records, one offset per cluster; no read table
        bare = eng.finish(reads=False)
        assert bare.reads.shape[0] == 0
        assert eng.format_sv(bare) == ref.sv, "stdout differs (records found through cluster_junction_off)"
        res = eng.finish()
        assert bare.n_reads == res.reads.shape[0]
        if res.reads.shape[0]:
            first = res.clusters["read_begin"].astype(np.int64)
            assert np.array_equal(res.reads["junction_off"][first], bare.cluster_junction_off[:-1])
            assert int(bare.cluster_junction_off[-1]) == res.junction_bytes.size
            assert np.array_equal(bare.junction_bytes, res.junction_bytes)
        ev = eng.evidence()
        n = b.n
        assert res.n_evidence == ref.n_evidence == n
        # per-read evidence (detect.rs:105-187)
        assert np.array_equal(ev["bp"].astype(np.uint32), ref.bp), "breakpoint offsets differ"
        assert np.array_equal((ev["flags"] & 7).astype(np.uint8), ref.flags), "strand/swap flags differ"
        assert np.array_equal(ev["lpos"].astype(np.int64).astype(np.uint64), ref.lpos), "left positions differ"
        assert np.array_equal(ev["rpos"].astype(np.int64).astype(np.uint64), ref.rpos), "right positions differ"
        plain = np.all(np.isin(np.delete(ref.signature, 8, axis=1), np.frombuffer(b"ACGT", np.uint8)), axis=1)
        sigx = (ev["flags"] & engine.BF_EV_SIGX) != 0
        assert np.array_equal(~plain, sigx), "non-ACGT signature flag differs"
        assert np.array_equal((ev["class_key"] & 0xffffffff).astype(np.uint32)[plain], pack_sig(ref.signature)[plain])
        if expect_exact is not None:
            n_exact = int(((ev["flags"] & engine.BF_EV_EXACT) != 0).sum())
            assert expect_exact(n_exact, n), "unexpected share of reads on the exact path: %d of %d" % (n_exact, n)
        # case-marked junction strings of ALL reads (detect.rs:155-162), rendered on the GPU
        by, off = eng.render_junctions(np.arange(n, dtype=np.uint32))
        assert np.array_equal(off, ref.junction_off)
        assert np.array_equal(by, ref.junction), "junction strings differ"
        # sort + cluster + dedup + print (detect.rs:190-241, 247-287)
        sv = eng.format_sv(res)
        assert sv == ref.sv, "stdout differs"
        assert res.clusters.shape[0] == ref.n_clusters
        return res, ref
    finally:
        eng.close()


@pytest.mark.parametrize("cfg,scale,gscale", [(0, 0.05, 0.02), (1, 0.004, 0.002), (3, 0.001, 0.002), (4, 0.002, 0.002)])
def test_scaled_baseline_config(cfg, scale, gscale):
    contigs, gk, spec = synth.config(cfg, scale=scale, genome_scale=gscale)
    g = synth.make_genome(contigs, **gk)
    b = synth.make_reads(g, spec)
    res, ref = check_against_oracle(g, b, spec.anchor_len)
    assert ref.n_clusters > 0


@pytest.mark.parametrize("flags", [0, BF_FLAG_FORCE_EXACT])
@pytest.mark.parametrize("cfg,scale,gscale", [(0, 0.05, 0.02), (1, 0.004, 0.002), (3, 0.001, 0.002), (4, 0.002, 0.002)])
def test_scaled_baseline_config_with_reads_as_4bit_codes(cfg, scale, gscale, flags):
    """Every per-read value, every junction string and the .sv with the reads resident as 4-bit codes
    (bf_submit_packed): configs[4] brings 1 % N bases, the exact kernel reads its bytes through the code table."""
    contigs, gk, spec = synth.config(cfg, scale=scale, genome_scale=gscale)
    g = synth.make_genome(contigs, **gk)
    b = synth.make_reads(g, spec)
    b.seq_bytes = b.seq_bytes & np.uint8(0xDF)     # a sequencer's reads are upper case whatever the reference's case is
    res, ref = check_against_oracle(g, b, spec.anchor_len, flags=flags, packed=True)
    assert ref.n_clusters > 0


def test_long_and_odd_reads_as_4bit_codes():
    contigs = [("c%d" % i, 120000) for i in range(3)]
    g = synth.make_genome(contigs, seed=19, lower_runs=[(1, 30000, 20000)])
    for L in (51, 129, 257, 701):
        spec = synth.ReadSpec(n_reads=900, read_len=L, anchor_len=25, n_events=15, singleton_frac=0.3, seed=300 + L,
                              margin=1300, subst_rate=0.02, n_rate=0.01)
        b = synth.make_reads(g, spec)
        b.seq_bytes = b.seq_bytes & np.uint8(0xDF)
        check_against_oracle(g, b, 25, packed=True)


def test_one_form_of_reads_per_ctx_and_packed_validation():
    contigs, gk, spec = synth.config(1, scale=0.001, genome_scale=0.002)
    g = synth.make_genome(contigs, **gk)
    b = synth.make_reads(g, spec)
    pb = engine.pack_batch(b)
    eng = engine.Engine(spec.anchor_len, 5000, 2)
    try:
        eng.set_genome(g.names, g.seqs)
        eng.submit(pb)
        with pytest.raises(engine.BreakfastError) as e:
            eng.submit(b)                                   # ASCII after 4-bit
        assert e.value.code == -4
        ref = ol.detect_batch(ol.OracleGenome(g.names, g.seqs), g.names, b, spec.anchor_len, 5000, 2)
        assert eng.format_sv(eng.finish()) == ref.sv
        eng.reset()
        eng.submit(b)                                       # after a reset either form is fine
        with pytest.raises(engine.BreakfastError) as e:
            eng.submit(pb)
        assert e.value.code == -4
        assert eng.format_sv(eng.finish()) == ref.sv
        eng.reset()
        bad = engine.PackedBatch(pb.seq_nibbles, pb.seq_off, pb.seq_len.copy(), pb.frag_bytes, pb.frag_off, pb.hits)
        bad.seq_len[7] += 2                                 # bases and bytes disagree
        with pytest.raises(engine.BreakfastError) as e:
            eng.submit(bad)
        assert e.value.code == -1 and "read 7 " in str(e.value)
        eng.submit(pb)
        assert eng.format_sv(eng.finish()) == ref.sv
    finally:
        eng.close()
    with pytest.raises(ValueError):
        engine.pack_batch(engine.make_batch([b"ACGTacgt" * 6], [b"f"], [(1, 100, 0, 0, 3)]))


@pytest.mark.parametrize("cfg,scale,gscale", [(0, 0.05, 0.02), (1, 0.004, 0.002), (3, 0.001, 0.002)])
def test_scaled_baseline_config_sorted_dedup(cfg, scale, gscale):
    # fragment-id dedup by the sorted-run kernels (the path of clusters above 256 members) instead of the in-wave form
    contigs, gk, spec = synth.config(cfg, scale=scale, genome_scale=gscale)
    g = synth.make_genome(contigs, **gk)
    b = synth.make_reads(g, spec)
    check_against_oracle(g, b, spec.anchor_len, flags=BF_FLAG_SORTED_DEDUP)


@pytest.mark.parametrize("cfg,scale,gscale", [(0, 0.05, 0.02), (1, 0.004, 0.002), (3, 0.001, 0.002), (4, 0.002, 0.002)])
def test_scaled_baseline_config_in_contig_groups(cfg, scale, gscale):
    """Sort / sweep / dedup / order / emit / render per group of whole contig ranks, results appended
    (api.hip, clusterise_in_groups); also together with the sorted-run dedup."""
    contigs, gk, spec = synth.config(cfg, scale=scale, genome_scale=gscale)
    g = synth.make_genome(contigs, **gk)
    b = synth.make_reads(g, spec)
    res, ref = check_against_oracle(g, b, spec.anchor_len, flags=BF_FLAG_GROUPS_ALWAYS)
    if cfg:                                                       # (configs[0] is one contig)
        assert len(set(res.clusters["contig"].tolist())) > 2      # several groups had something to report
    check_against_oracle(g, b, spec.anchor_len, flags=BF_FLAG_GROUPS_ALWAYS | BF_FLAG_SORTED_DEDUP)
    # one contig: nothing to cut, the one-go path takes over
    g1 = synth.make_genome([("only", 400000)], seed=5)
    spec1 = synth.ReadSpec(n_reads=4000, read_len=100, anchor_len=20, n_events=40, singleton_frac=0.3, mate_frac=0.0, seed=9, margin=300)
    check_against_oracle(g1, synth.make_reads(g1, spec1), 20, flags=BF_FLAG_GROUPS_ALWAYS)


@pytest.mark.parametrize("n_reads,n_events", [(3000, 40), (6000, 30), (9000, 12)])
def test_dedup_over_cluster_sizes(n_reads, n_events):
    """Clusters of ~75 (one round of the in-wave dedup), ~200 (four rounds) and ~750 members (above 256: the
    sorted-run kernels take over), 40 % of the reads sharing their fragment id with a mate at the same junction."""
    contigs = [("chr1", 200000), ("chr2", 150000)]
    g = synth.make_genome(contigs, seed=5)
    spec = synth.ReadSpec(n_reads=n_reads, read_len=100, anchor_len=20, n_events=n_events, singleton_frac=0.0, mate_frac=0.4,
                          seed=n_reads, margin=300)
    b = synth.make_reads(g, spec)
    res, ref = check_against_oracle(g, b, 20)
    sizes = np.diff(np.stack([res.clusters["read_begin"], res.clusters["read_end"]], 1).astype(np.int64), axis=1)
    assert sizes.max() >= {3000: 30, 6000: 80, 9000: 300}[n_reads]
    check_against_oracle(g, b, 20, flags=BF_FLAG_SORTED_DEDUP)
    # three hash bits: different fragment ids meet in one hash group, the byte check fails, the wave redoes its cluster
    check_against_oracle(g, b, 20, flags=BF_FLAG_WEAK_DEDUP_HASH)


@pytest.mark.parametrize("bits", [None, "20", "9", "1"])
def test_class_order_chunks_that_overflow(bits, monkeypatch):
    """Class partition + chunk sort: signatures with thousands of records (chunks too large for LDS are set aside and
    sorted by the radix sort), beside thousands of small ones; and the same with 2^20, 512 and 2 buckets (test hook):
    with two buckets EVERY chunk overflows."""
    if bits:
        monkeypatch.setenv("BREAKFAST_HIP_BUCKET_BITS", bits)
    contigs = [("chr1", 300000), ("chr10", 200000), ("chr2", 250000)]
    g = synth.make_genome(contigs, seed=15)
    heavy = synth.make_reads(g, synth.ReadSpec(n_reads=12000, read_len=100, anchor_len=20, n_events=5, singleton_frac=0.0, mate_frac=0.3,
                                               seed=77, margin=300))
    light = synth.make_reads(g, synth.ReadSpec(n_reads=20000, read_len=100, anchor_len=20, n_events=2500, singleton_frac=0.4, mate_frac=0.2,
                                               seed=78, margin=300, frag_base=12000))
    import test_gpu_submit as tgs
    b = tgs.concat([heavy.slice(0, 5000), light, heavy.slice(5000, heavy.n)])
    res, ref = check_against_oracle(g, b, 20)
    sizes = (res.clusters["read_end"] - res.clusters["read_begin"]).astype(np.int64)
    assert sizes.max() > 1024
    check_against_oracle(g, b, 20, flags=BF_FLAG_GROUPS_ALWAYS)
    check_against_oracle(g, b, 20, flags=BF_FLAG_FORCE_EXACT)


@pytest.mark.parametrize("bits", ["", "12", "8", "7"])
def test_buckets_sorted_by_one_wave_in_registers(bits, monkeypatch):
    """The chunk sort's register form: a bucket (the records of a few signatures) of up to 64, 128 or 256 records is sorted
    by one wave, one, two or four records a lane, on one 64-bit key whose low bits hold the record's position in the bucket.
    Junctions with ~40, ~100 and ~200 reads each (every size class, ties on lpos broken by input order), with 65536, 4096
    and 256 buckets (fewer buckets = more signatures per bucket: 8 bits leave exactly the room a position needs), and with
    128 buckets (no room: the LDS network takes every chunk).  Against the oracle, in one go and in contig groups."""
    if bits:
        monkeypatch.setenv("BREAKFAST_HIP_BUCKET_BITS", bits)
    contigs = [("chr1", 300000), ("chr10", 200000), ("chr2", 250000)]
    g = synth.make_genome(contigs, seed=25)
    parts, base = [], 0
    for n_reads, n_events, seed in ((4000, 100, 81), (6000, 60, 82), (8000, 40, 83), (3000, 1500, 84)):
        parts.append(synth.make_reads(g, synth.ReadSpec(n_reads=n_reads, read_len=100, anchor_len=20, n_events=n_events, singleton_frac=0.0,
                                                        mate_frac=0.3, seed=seed, margin=300, frag_base=base)))
        base += n_reads
    import test_gpu_submit as tgs
    pieces = []                                                    # the size classes interleaved in input order
    for k in range(16):
        pieces += [p.slice(k * p.n // 16, (k + 1) * p.n // 16) for p in parts]
    b = tgs.concat(pieces)
    res, ref = check_against_oracle(g, b, 20)
    sizes = (res.clusters["read_end"] - res.clusters["read_begin"]).astype(np.int64)
    assert sizes.max() > 128 and (sizes < 64).any()
    check_against_oracle(g, b, 20, flags=BF_FLAG_GROUPS_ALWAYS)


def test_scaled_config_exact_kernel_only():
    contigs, gk, spec = synth.config(1, scale=0.0005, genome_scale=0.002)
    g = synth.make_genome(contigs, **gk)
    b = synth.make_reads(g, spec)
    check_against_oracle(g, b, spec.anchor_len, flags=BF_FLAG_FORCE_EXACT, expect_exact=lambda x, n: x == n)


def test_packed_path_is_the_one_that_runs():
    # uppercase ACGT genome, junctions far from N runs: (almost) everything must take the packed kernel
    contigs, gk, spec = synth.config(1, scale=0.002, genome_scale=0.002)
    spec.margin = 10000 + 2 * spec.read_len + 128      # whole windows clear of the N runs and their 64-base blocks
    contigs = [(n, max(l, 60000)) for n, l in contigs]
    g = synth.make_genome(contigs, **gk)
    b = synth.make_reads(g, spec)
    check_against_oracle(g, b, spec.anchor_len, expect_exact=lambda x, n: x == 0)


def test_soft_masked_and_n_reference():
    # lowercase + N runs under the junctions (UCSC-style soft-masked genomes): the packed kernel keeps these
    # reads (a reference byte outside ACGT differs from every ACGT read base) and builds raw-byte signatures
    contigs = [("chrA", 300000), ("chrB", 200000)]
    g = synth.make_genome(contigs, seed=5, n_runs=[(0, 50000, 3000), (1, 1000, 500)],
                          lower_runs=[(0, 100000, 100000), (1, 100000, 50000)])
    spec = synth.ReadSpec(n_reads=6000, read_len=100, anchor_len=30, n_events=60, singleton_frac=0.3, seed=77, margin=400)
    b = synth.make_reads(g, spec)
    upper = synth.Batch(b.seq_bytes & np.uint8(0xDF), b.seq_off, b.frag_bytes, b.frag_off, b.hits)   # BAM reads are upper case
    # (a few generated reads come out of the N runs and carry N themselves: those need the bytes)
    check_against_oracle(g, upper, 30, expect_exact=lambda x, n: x < n // 20)
    # the generator's own reads copy the genome's case: a lowercase (or N) read base facing a masked
    # reference base needs the bytes -> exact kernel
    check_against_oracle(g, b, 30, expect_exact=lambda x, n: 0 < x < n)
    spec = synth.ReadSpec(n_reads=6000, read_len=100, anchor_len=30, n_events=60, singleton_frac=0.3, seed=78, n_rate=0.02, margin=400)
    b = synth.make_reads(g, spec)
    upper = synth.Batch(np.where(b.seq_bytes == ord("n"), b.seq_bytes, b.seq_bytes & np.uint8(0xDF)).astype(np.uint8), b.seq_off,
                        b.frag_bytes, b.frag_off, b.hits)
    check_against_oracle(g, upper, 30, expect_exact=lambda x, n: 0 < x < n // 2)


def test_heavily_soft_masked_genome():
    # half of the genome in lowercase runs of 50..3000 bases, windows straddling run boundaries, every
    # strand combination; the packed kernel and the byte-exact kernel
    rng = np.random.Generator(np.random.PCG64(17))
    contigs = [("chr1", 250000), ("chr2", 150000), ("chrM", 40000)]
    g = synth.make_genome(contigs, seed=15, end_n=500)
    for sq in g.seqs:
        for _ in range(sq.shape[0] // 2000):
            a = int(rng.integers(0, sq.shape[0] - 3000))
            sq[a:a + int(rng.integers(50, 3000))] |= 0x20
    g.seqs[0][100000:100040] = np.frombuffer(b"RYKMSWBDHV" * 4, np.uint8)          # IUPAC codes in the reference
    for L, A in ((100, 30), (151, 20), (250, 30)):
        spec = synth.ReadSpec(n_reads=5000, read_len=L, anchor_len=A, n_events=80, singleton_frac=0.3, deletion_frac=0.3,
                              seed=300 + L, margin=800, subst_rate=0.01)
        b = synth.make_reads(g, spec)
        b = synth.Batch(b.seq_bytes & np.uint8(0xDF), b.seq_off, b.frag_bytes, b.frag_off, b.hits)     # upper-case reads
        # (reads drawn from the IUPAC stretch carry those codes themselves and go to the exact kernel)
        res, ref = check_against_oracle(g, b, A, expect_exact=lambda x, n: x < n // 50)
        assert ref.n_clusters > 20
        check_against_oracle(g, b, A, flags=BF_FLAG_FORCE_EXACT)


def test_mixed_read_lengths_and_batches():
    # reads of several lengths submitted in several batches (one longer than the packed path's 256)
    contigs = [("c%d" % i, 80000) for i in range(3)]
    g = synth.make_genome(contigs, seed=9)
    parts = []
    for i, L in enumerate((64, 100, 151, 250, 300)):
        spec = synth.ReadSpec(n_reads=700, read_len=L, anchor_len=20, n_events=12, singleton_frac=0.3, seed=100 + i,
                              margin=400)
        parts.append(synth.make_reads(g, spec))
    seqs, frags, hits = [], [], []
    for p in parts:
        for i in range(p.n):
            h = p.hits[i]
            seqs.append(p.seq(i)); frags.append(p.frag(i))
            hits.append((int(h["pos5"]), int(h["pos3"]), int(h["contig5"]), int(h["contig3"]), int(h["strands"])))
    order = np.random.Generator(np.random.PCG64(3)).permutation(len(seqs))
    whole = engine.make_batch([seqs[i] for i in order], [frags[i] for i in order], [hits[i] for i in order])
    og = ol.OracleGenome(g.names, g.seqs)
    ref = ol.detect_batch(og, g.names, whole, 20, 5000, 2)
    eng = engine.Engine(20, 5000, 2)
    try:
        eng.set_genome(g.names, g.seqs)
        for a in range(0, whole.n, 1000):
            eng.submit(whole.slice(a, min(whole.n, a + 1000)))
        assert eng.format_sv(eng.finish()) == ref.sv
        assert eng.format_sv(eng.finish()) == ref.sv          # idempotent on resident reads
        eng.reset()
        assert eng.format_sv(eng.finish()) == po.SV_HEADER    # empty input: header only
    finally:
        eng.close()


def test_error_behaviour():
    g = synth.make_genome([("c0", 50000)], seed=2)
    eng = engine.Engine(30, 5000, 2)
    try:
        with pytest.raises(engine.BreakfastError):            # submit before genome
            eng.submit(engine.make_batch([b"A" * 60], [b"f"], [(1, 100, 0, 0, 3)]))
        eng.set_genome(g.names, g.seqs)
        with pytest.raises(engine.BreakfastError) as e:       # unknown contig id (reference: panic)
            eng.submit(engine.make_batch([b"A" * 60], [b"f"], [(1, 100, 0, 5, 3)]))
        assert "contig" in str(e.value)
        with pytest.raises(engine.BreakfastError) as e:       # read shorter than 2 * anchor_len
            eng.submit(engine.make_batch([b"A" * 59], [b"f"], [(1, 100, 0, 0, 3)]))
        assert e.value.code == -5
        with pytest.raises(engine.BreakfastError):            # duplicate contig names
            eng.set_genome([b"x", b"x"], [b"ACGT" * 100, b"ACGT" * 100])
    finally:
        eng.close()


def test_read_table_on_request():
    """bf_finish leaves the read table on the device; bf_result_reads completes the result it is given, and only
    the result of this ctx's last bf_finish."""
    import ctypes as C
    contigs, gk, spec = synth.config(1, scale=0.002, genome_scale=0.002)
    g = synth.make_genome(contigs, **gk)
    b = synth.make_reads(g, spec)
    eng = engine.Engine(spec.anchor_len, 5000, 2)
    try:
        eng.set_genome(g.names, g.seqs)
        r = engine.BfResult()
        assert eng.lib.bf_result_reads(eng.h, C.byref(r)) == -4            # nothing clustered yet
        eng.submit(b)
        r = eng.finish_raw()
        assert r.n_reads > 0 and not r.reads and r.cluster_junction_off
        stale = engine.BfResult(r.n_evidence, r.n_clusters, r.n_reads + 1, r.clusters, None, r.junction_bytes,
                                r.junction_size, r.cluster_junction_off)
        assert eng.lib.bf_result_reads(eng.h, C.byref(stale)) == -4        # not this ctx's result
        assert eng.lib.bf_result_reads(eng.h, None) == -1
        for _ in range(2):                                                  # asking twice is fine
            assert eng.lib.bf_result_reads(eng.h, C.byref(r)) == 0 and r.reads
        res = eng._wrap(r)
        # the table agrees with a walk over the records: offsets, and the input read behind every string
        off = 0
        for k in range(res.n_reads):
            assert int(res.reads[k]["junction_off"]) == off
            L, bp, sw, size = engine.junction_header(res.junction_bytes, off)
            assert L == int(b.seq_off[int(res.reads[k]["index"]) + 1] - b.seq_off[int(res.reads[k]["index"])])
            off += size
        assert off == res.junction_bytes.size
        eng.reset()
        assert eng.lib.bf_result_reads(eng.h, C.byref(r)) in (0, -4)       # never a crash after a reset
    finally:
        eng.close()


@pytest.mark.parametrize("form", ["file", "file --ascii-reads", "pipe", "pipe --packed-reads", "file --devices=0,0"])
@pytest.mark.parametrize("name", ["kat_survey", "hz_contig_order", "hz_ref_alphabet", "hz_pairing", "hz_wrapped_coords",
                                  "hz_read_n_iupac", "hz_mixed_lengths", "hz_dedup_min_evidence", "hz_empty"])
def test_cli_driver_reproduces_golden(name, form, tmp_path):
    """`breakfast-hip detect` (the C++ host driver over the C ABI): same argv shape and stdout as the reference.
    A hit list in a file goes to the GPU as 4-bit base codes, and once more as ASCII when a read is outside BAM's
    alphabet (hz_ref_alphabet's lower-case reads); a pipe is taken as ASCII unless --packed-reads promises the alphabet."""
    import os
    import subprocess
    from conftest import GOLDEN, ROOT
    exe = os.path.join(ROOT, "breakfast_amd", "csrc", "breakfast-hip")
    assert os.path.exists(exe), "build the CLI driver first (__graft_entry__.build)"
    fa, lines, (A, F, E), sv = load_golden(name)
    prefix = os.path.join(GOLDEN, name)          # the driver opens <genome>.fa like detect.rs:48
    source, *flags = form.split()
    reads = [l.split(b":")[3] for l in lines.split(b"\n") if l.startswith(b"3p:") and l.count(b":") >= 4]
    lower = any(c not in b"=ACMGRSVTWYHKDBN" for r in reads for c in r)
    p = subprocess.run([exe, "detect", "--anchor-len=%d" % A, "--max-frag-len=%d" % F, "--min-evidence=%d" % E] + flags +
                       ["--anchor-hits=" + ("%s.bowtie" % prefix if source == "file" else "-"), "unused.bam", prefix],
                       input=lines if source == "pipe" else None, capture_output=True, timeout=120)
    if lower and form == "pipe --packed-reads":                    # the promise does not hold: an error, never a wrong table
        assert p.returncode == 255 and b"outside BAM's base alphabet" in p.stderr
        return
    assert p.returncode == 0, p.stderr.decode()
    assert p.stdout == sv
    assert b"Found " in p.stderr and b"rearrangement supporting reads." in p.stderr
    assert p.stderr.count(b"Reading reference genome into memory...") == 1


@pytest.mark.parametrize("form", ["", "--devices=0,0"])
@pytest.mark.parametrize("name", ["kat_survey", "hz_ref_alphabet", "hz_wrapped_coords", "hz_read_n_iupac"])
def test_cli_driver_with_a_genome_image(name, form, tmp_path):
    """`breakfast-hip index <genome>` writes <genome>.bf2bit beside <genome>.fa; `detect` then hands the image to the GPU
    instead of reading the FASTA (--timing names the phase), with the same stdout; an image whose FASTA has changed since
    is left aside with a warning, and the FASTA is read."""
    import os
    import shutil
    import subprocess
    from conftest import GOLDEN, ROOT
    exe = os.path.join(ROOT, "breakfast_amd", "csrc", "breakfast-hip")
    fa, lines, (A, F, E), sv = load_golden(name)
    prefix = str(tmp_path / "genome")
    shutil.copy(os.path.join(GOLDEN, name + ".fa"), prefix + ".fa")
    argv = [exe, "detect", "--timing", "--anchor-len=%d" % A, "--max-frag-len=%d" % F, "--min-evidence=%d" % E] + form.split() + \
           ["--anchor-hits=%s.bowtie" % os.path.join(GOLDEN, name), "unused.bam", prefix]
    p = subprocess.run(argv, capture_output=True, timeout=120)
    assert p.returncode == 0 and p.stdout == sv and b"image_open" not in p.stderr, p.stderr.decode()
    q = subprocess.run([exe, "index", prefix], capture_output=True, timeout=120)
    assert q.returncode == 0 and q.stdout == b"", q.stderr.decode()
    assert sorted(os.listdir(tmp_path)) == ["genome.bf2bit", "genome.fa"]
    p = subprocess.run(argv, capture_output=True, timeout=120)
    assert p.returncode == 0 and p.stdout == sv and b"image_open" in p.stderr, p.stderr.decode()
    assert b"fasta_scan" not in p.stderr and b"fasta_load" not in p.stderr
    p = subprocess.run(argv, capture_output=True, timeout=120, env=dict(os.environ, BREAKFAST_HIP_NO_IMAGE="1"))
    assert p.returncode == 0 and p.stdout == sv and b"image_open" not in p.stderr
    st = os.stat(prefix + ".fa")
    os.utime(prefix + ".fa", ns=(st.st_atime_ns, st.st_mtime_ns + 1000))            # the FASTA was touched after the image was made
    p = subprocess.run(argv, capture_output=True, timeout=120)
    assert p.returncode == 0 and p.stdout == sv and b"image_open" not in p.stderr and b"WARNING: " in p.stderr and b"stale" in p.stderr
    q = subprocess.run([exe, "index", str(tmp_path / "absent")], capture_output=True, timeout=120)
    assert q.returncode == 255 and q.stderr.startswith(b"ERROR: ")


@pytest.mark.parametrize("slices", ["1", "3", "8"])
def test_cli_driver_writes_into_a_file_a_pipe_and_an_appended_file(slices, tmp_path):
    """The table is rendered slice by slice while a thread writes the slice before: into a pipe, behind what a file already
    holds, at the end of a file opened for appending.  Same bytes every way, whatever the number of slices.  (Writing the
    slices at their places from several threads gained nothing -- buffered writes to one file take turns in the kernel --
    and rendering straight into the mapped file took five times as long: both measured and dropped)"""
    import os
    import subprocess
    from conftest import GOLDEN, ROOT
    exe = os.path.join(ROOT, "breakfast_amd", "csrc", "breakfast-hip")
    name = "hz_dedup_min_evidence"
    fa, lines, (A, F, E), sv = load_golden(name)
    prefix = os.path.join(GOLDEN, name)
    argv = [exe, "detect", "--anchor-len=%d" % A, "--max-frag-len=%d" % F, "--min-evidence=%d" % E,
            "--anchor-hits=%s.bowtie" % prefix, "unused.bam", prefix]
    env = dict(os.environ, BREAKFAST_HIP_SV_SLICES=slices)
    assert subprocess.run(argv, capture_output=True, timeout=120, env=env).stdout == sv                    # a pipe
    for mode in ("wb", "ab"):
        path = tmp_path / ("out_" + mode)
        path.write_bytes(b"")
        with open(path, mode) as f:
            f.write(b"# what was there before\n"); f.flush()
            assert subprocess.run(argv, stdout=f, stderr=subprocess.PIPE, timeout=120, env=env).returncode == 0
            f.write(b"# what came after\n")
        assert path.read_bytes() == b"# what was there before\n" + sv + b"# what came after\n", mode


def test_cli_driver_error_convention(tmp_path):
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "breakfast_amd", "csrc", "breakfast-hip")
    p = subprocess.run([exe, "detect", "--anchor-hits=/dev/null", "x.bam", "/nonexistent/genome"], capture_output=True, timeout=120)
    assert p.returncode == 255 and p.stderr.startswith(b"ERROR: ")      # common.rs:8-13: exit(-1)


def test_cli_driver_one_process_and_detached_agree(tmp_path):
    """`detect` hands its status to a waiting parent once the table is written and frees the GPU behind it;
    BREAKFAST_HIP_NO_DETACH=1 keeps one process.  Same bytes, same status, same stderr either way, and a hit list
    that fails late (a contig the genome does not have, after the GPU came up) is exit(-1) in both."""
    import os
    import subprocess
    from conftest import GOLDEN, ROOT
    exe = os.path.join(ROOT, "breakfast_amd", "csrc", "breakfast-hip")
    fa, lines, (A, F, E), sv = load_golden("kat_survey")
    prefix = os.path.join(GOLDEN, "kat_survey")
    bad = tmp_path / "bad.bowtie"
    bad.write_bytes(lines + b"5p:999999:\t+\tno_such_contig\t27\n3p:999999:fragZ:AGTTATCGGTGCCGAACTGATGGA:\t+\tchr2\t33\n")
    outs = []
    for env in ({}, {"BREAKFAST_HIP_NO_DETACH": "1"}):
        e = dict(os.environ, **env)
        argv = [exe, "detect", "--anchor-len=%d" % A, "--max-frag-len=%d" % F, "--min-evidence=%d" % E]
        good = subprocess.run(argv + ["--anchor-hits=%s.bowtie" % prefix, "unused.bam", prefix], capture_output=True,
                              timeout=120, env=e)
        assert good.returncode == 0 and good.stdout == sv, good.stderr.decode()
        late = subprocess.run(argv + ["--anchor-hits=%s" % bad, "unused.bam", prefix], capture_output=True, timeout=120, env=e)
        assert late.returncode == 255 and late.stderr.count(b"ERROR: ") == 1, late.stderr.decode()
        outs.append((good.stderr, late.stderr, late.stdout))
    assert outs[0] == outs[1]


def _frag_hash32(b: bytes) -> int:
    """low 32 bits of read_meta_kernel's fragment-id hash (FNV-1a/64 + avalanche)."""
    M = (1 << 64) - 1
    h = 0xcbf29ce484222325
    for c in b:
        h = ((h ^ c) * 0x100000001b3) & M
    h ^= h >> 32
    h = (h * 0xd6e8feb86659fd93) & M
    h ^= h >> 32
    return h & 0xffffffff


def test_dedup_survives_hash_collisions():
    """Two DIFFERENT fragment ids whose 32 sort-key hash bits collide must stay two fragments (groups are
    decided on the id bytes, detect.rs:257), while equal ids still collapse."""
    seen = {}
    pair = None
    i = 0
    while pair is None:
        name = b"frag%d" % i
        h = _frag_hash32(name)
        if h in seen:
            pair = (seen[h], name)
        seen[h] = name
        i += 1
    a_id, b_id = pair
    assert a_id != b_id and _frag_hash32(a_id) == _frag_hash32(b_id)
    import random
    sys_path_golden = __import__("os").path.join(__import__("conftest").GOLDEN)
    import sys
    sys.path.insert(0, sys_path_golden)
    import make_golden as mg
    g, _ = mg.base_genome(77)
    f = mg.Fixture("tmp_collision", g, (10, 5000, 2))
    # one junction: a_id twice (collapses to its best), b_id twice (collapses), c once -> 3 reads survive
    for frag, b in ((a_id, 20), (b_id, 22), (a_id, 30), (b"other", 26), (b_id, 28)):
        f.add(frag, *mg.split_read(g, b"chr1", 300, True, b"chr2", 200, True, 60, b, 10))
    expected = f.run()
    assert expected.count(b";") == 2
    text = b"".join(l + b"\n" for l in f.lines)
    assert run_gpu_text(f.fasta(), text, 10, 5000, 2) == expected
    assert run_gpu_text(f.fasta(), text, 10, 5000, 2, flags=BF_FLAG_SORTED_DEDUP) == expected
    assert run_gpu_text(f.fasta(), text, 10, 5000, 2, flags=BF_FLAG_WEAK_DEDUP_HASH) == expected


def test_long_reads_render_in_several_tiles():
    # reads longer than one render tile (256 output bytes) and than the packed path (exact kernel)
    contigs = [("c%d" % i, 120000) for i in range(3)]
    g = synth.make_genome(contigs, seed=19)
    for L in (257, 700, 1100):
        spec = synth.ReadSpec(n_reads=900, read_len=L, anchor_len=25, n_events=15, singleton_frac=0.3, seed=200 + L,
                              margin=1300, subst_rate=0.02)
        b = synth.make_reads(g, spec)
        check_against_oracle(g, b, 25)


def test_full_size_headline_config_is_byte_identical():
    """BASELINE.json configs[1] at FULL size (10M reads x 150 bp, 3.09 Gb genome): the whole .sv
    (about 600 MB) against the C++ oracle, plus idempotence of a second pass on the resident reads."""
    import hashlib
    contigs, gk, spec = synth.config(1)
    spec.margin = 10000 + 2 * spec.read_len + 128
    g = synth.make_genome(contigs, **gk)
    b = synth.make_reads(g, spec)
    eng = engine.Engine(spec.anchor_len, 5000, 2)
    try:
        eng.set_genome(g.names, g.seqs)
        for a in range(0, b.n, 1 << 21):
            eng.submit(b.slice(a, min(b.n, a + (1 << 21))))
        res = eng.finish(copy=False)
        sv = eng.format_sv(res)
        h1 = hashlib.sha256(sv).hexdigest()
        n_clusters = res.clusters.shape[0]
        res2 = eng.finish(copy=False)
        assert hashlib.sha256(eng.format_sv(res2)).hexdigest() == h1
    finally:
        eng.close()
    # the same input as 4-bit reads whose bases stay with the caller (BF_FLAG_READS_ON_HOST: records name the read)
    eng = engine.Engine(spec.anchor_len, 5000, 2, flags=engine.BF_FLAG_READS_ON_HOST)
    try:
        eng.set_genome(g.names, g.seqs)
        for a in range(0, b.n, 3000001):
            eng.submit(engine.pack_batch(b.slice(a, min(b.n, a + 3000001))))
        res = eng.finish(copy=False)
        assert res.junction_bytes.size * 4 < len(sv)
        assert hashlib.sha256(eng.format_sv(res)).hexdigest() == h1
    finally:
        eng.close()
    ref = ol.detect_batch(ol.OracleGenome(g.names, g.seqs), g.names, b, spec.anchor_len, 5000, 2)
    assert n_clusters == ref.n_clusters > 100000
    assert len(sv) == len(ref.sv) and sv == ref.sv


def test_segments_that_need_a_walk():
    """Several junctions sharing ONE breakpoint class (same contigs, strands and 8+8 signature bases, made by
    copying the flanks around the genome): the sweep's fast path (whole segment = one cluster) must not
    fire; short segments take the one-lane walk, long ones the one-wave walk; left positions within
    max_frag_len with right positions far apart exercise the `continue` branch and reads that join a
    later seed's cluster although an earlier seed already reported them (detect.rs:215-226)."""
    rng = np.random.Generator(np.random.PCG64(123))
    A, L = 20, 100
    g = synth.make_genome([("c0", 400000), ("c1", 300000)], seed=31)
    s0 = g.seqs[0]
    seqs, frags, hits = [], [], []

    def plant(src, dsts, contig):
        for d in dsts:
            g.seqs[contig][d - 300:d + 300] = g.seqs[contig][src - 300:src + 300]

    def add_reads(x1, y1, c2, count, tag):
        for i in range(count):
            b = int(rng.integers(A, L - A))
            read = np.concatenate([g.seqs[0][x1 - b + 1:x1 + 1], g.seqs[c2][y1:y1 + L - b]]).copy()
            if rng.random() < 0.3:                                   # a substitution outside the anchors
                k = int(rng.integers(A, L - A))
                read[k] = ord("A") if read[k] != ord("A") else ord("C")
            seqs.append(read.tobytes())
            frags.append(b"%s_%d" % (tag, i // 2 if i % 5 == 0 else 100000 + i))
            hits.append((x1 - b + 2, y1 + L - A - b + 1, 0, c2, 3))

    # class 1 (long segment): left copies 20 kb apart (beyond -> separate clusters), 400 reads each
    plant(50000, [70000, 90000], 0); plant(250000, [270000, 290000], 0)
    for k in range(3):
        add_reads(50000 + 20000 * k, 250000 + 20000 * k, 0, 400, b"big%d" % k)
    # class 2 (short segment): left copies 150 bases apart (within max_frag_len), right copies 30 kb apart
    plant(120000, [120150, 120300], 0); plant(100000, [130000, 160000], 1)
    for k in range(3):
        add_reads(120000 + 150 * k, 100000 + 30000 * k, 1, 12, b"small%d" % k)
    # class 3: same left copies, right copies 2 kb apart: everything within the seed's windows -> fast path
    plant(200000, [202000], 1)
    for k in range(2):
        add_reads(120000 + 150 * k + 7, 200000 + 2000 * k, 1, 20, b"fast%d" % k)
    order = rng.permutation(len(seqs))
    b = engine.make_batch([seqs[i] for i in order], [frags[i] for i in order], [hits[i] for i in order])
    res, ref = check_against_oracle(g, b, A)
    assert ref.n_clusters >= 7
    # a smaller window turns the 150-base neighbours into separate clusters as well
    check_against_oracle(g, b, A, F=100)


@pytest.mark.parametrize("E,F", [(1, 5000), (0, 5000), (1, 0), (3, 50), (50, 5000)])
def test_min_evidence_and_window_extremes(E, F):
    """--min-evidence 1 (every read seeds or joins a reported cluster: as many clusters as classes), 0 (what the
    reference prints for 1: `len() < 0` is never true, detect.rs:229), a zero --max-frag-len window, and a threshold
    no cluster reaches (header only)."""
    contigs, gk, spec = synth.config(1, scale=0.001, genome_scale=0.002)
    g = synth.make_genome(contigs, **gk)
    b = synth.make_reads(g, spec)
    res, ref = check_against_oracle(g, b, spec.anchor_len, F=F, E=E)
    if E <= 1:
        assert ref.n_clusters > 1000


def test_single_read_and_two_reads():
    contigs, gk, spec = synth.config(1, scale=0.001, genome_scale=0.002)
    g = synth.make_genome(contigs, **gk)
    b = synth.make_reads(g, spec)
    for n in (1, 2, 3):
        check_against_oracle(g, b.slice(0, n), spec.anchor_len, E=1)


def test_classes_that_share_a_signature():
    """Different breakpoint classes with the SAME 8+8 signature bases (the flanks copied onto another contig):
    the class keys are sorted on their signature bits only, so such classes share a segment, their records
    interleaved by left position; every membership test has to compare the whole key (detect.rs:218-223).
    Short segments (one-lane walk), long ones (one-wave walk) and the member enumeration all see the mix."""
    rng = np.random.Generator(np.random.PCG64(321))
    A, L = 20, 100
    g = synth.make_genome([("c0", 300000), ("c1", 300000), ("c2", 300000)], seed=37)
    seqs, frags, hits = [], [], []

    def add_reads(cl, x1, cr, y1, count, tag):
        for i in range(count):
            b = int(rng.integers(A, L - A))
            read = np.concatenate([g.seqs[cl][x1 - b + 1:x1 + 1], g.seqs[cr][y1:y1 + L - b]]).copy()
            if rng.random() < 0.3:
                k = int(rng.integers(A, L - A))
                read[k] = ord("A") if read[k] != ord("A") else ord("C")
            seqs.append(read.tobytes())
            frags.append(b"%s_%d" % (tag, i // 2 if i % 4 == 0 else 100000 + i))
            hits.append((x1 - b + 2, y1 + L - A - b + 1, cl, cr, 3))

    for n_reads, x, y in ((14, 50000, 80000), (500, 150000, 200000)):
        # the left flank of c0:x also at c2:x+10, the right flank of c1:y also at c1:y+40000 and c2:y
        g.seqs[2][x + 10 - 300:x + 10 + 300] = g.seqs[0][x - 300:x + 300]
        g.seqs[1][y + 40000 - 300:y + 40000 + 300] = g.seqs[1][y - 300:y + 300]
        g.seqs[2][y - 300:y + 300] = g.seqs[1][y - 300:y + 300]
        add_reads(0, x, 1, y, n_reads, b"a%d" % n_reads)               # class (c0, c1)
        add_reads(2, x + 10, 1, y, n_reads, b"b%d" % n_reads)          # class (c2, c1): same signature, other left contig
        add_reads(0, x, 2, y, n_reads, b"c%d" % n_reads)               # class (c0, c2): other right contig
        add_reads(0, x, 1, y + 40000, n_reads // 2, b"d%d" % n_reads)  # class (c0, c1) again, right position far away
    order = rng.permutation(len(seqs))
    b = engine.make_batch([seqs[i] for i in order], [frags[i] for i in order], [hits[i] for i in order])
    res, ref = check_against_oracle(g, b, A)
    assert ref.n_clusters >= 8
    check_against_oracle(g, b, A, flags=BF_FLAG_SORTED_DEDUP)


def case_reads_printed_twice(pile_on_last_contig, n_pile=400):
    """detect.rs:215-226 lets a read that an earlier seed reported join a later seed's cluster as well, so the printed reads
    can outnumber the evidence: two seeds whose right positions lie too far apart for one cluster, and a pile of reads
    within reach of both; ordinary junctions on another contig, so that there are groups to cut -> (genome, batch, A, F)."""
    rng = np.random.Generator(np.random.PCG64(99))
    A, L, F = 20, 100, 5000
    g = synth.make_genome([("c0", 300000), ("c1", 300000), ("c2", 300000), ("c3", 300000)], seed=41)
    cp, co = (2, 0) if pile_on_last_contig else (0, 2)     # left contigs of the pile and of the ordinary junctions (right: + 1)
    x0, y0 = 100000, 150000
    for dx in (1000, 2000):                                 # the left flank twice more, the right flank twice more
        g.seqs[cp][x0 + dx - 300:x0 + dx + 300] = g.seqs[cp][x0 - 300:x0 + 300]
    for dy in (6000, 3000):
        g.seqs[cp + 1][y0 + dy - 300:y0 + dy + 300] = g.seqs[cp + 1][y0 - 300:y0 + 300]
    seqs, frags, hits = [], [], []

    def add_reads(cl, x1, y1, count, tag, b=50):
        for i in range(count):
            read = np.concatenate([g.seqs[cl][x1 - b + 1:x1 + 1], g.seqs[cl + 1][y1:y1 + L - b]]).copy()
            seqs.append(read.tobytes())
            frags.append(b"%s_%d" % (tag, i))
            hits.append((x1 - b + 2, y1 + L - A - b + 1, cl, cl + 1, 3))
    add_reads(cp, x0, y0, 3, b"s1")                         # seed 1 and company
    add_reads(cp, x0 + 1000, y0 + 6000, 3, b"s2")           # seed 2: right positions 6000 from seed 1's, no common cluster
    add_reads(cp, x0 + 2000, y0 + 3000, n_pile, b"both")    # within reach of both seeds: printed twice
    for k in range(40):
        add_reads(co, 20000 + 1000 * k, 30000 + 1000 * k, 4, b"o%d" % k, b=45)
    order = rng.permutation(len(seqs))
    return g, engine.make_batch([seqs[i] for i in order], [frags[i] for i in order], [hits[i] for i in order]), A, F


@pytest.mark.parametrize("pile_on_last_contig", [False, True])
def test_reads_printed_in_two_clusters_outgrow_the_evidence(pile_on_last_contig):
    """More reads print than there is evidence.  Contig groups size their per-read arrays for the evidence and their record
    buffers from ALL reads' record sizes before the first group is clustered: the arrays must grow (with earlier groups'
    entries in them when the pile is in the last group), and the stage must fall back to one go for the records."""
    g, b, A, F = case_reads_printed_twice(pile_on_last_contig)
    res, ref = check_against_oracle(g, b, A, F=F)
    assert res.reads.shape[0] > b.n                         # more printed reads than evidence
    for flags in (BF_FLAG_GROUPS_ALWAYS, BF_FLAG_GROUPS_ALWAYS | BF_FLAG_FORCE_EXACT):
        check_against_oracle(g, b, A, F=F, flags=flags)
    check_against_oracle(g, b, A, F=F, flags=BF_FLAG_GROUPS_ALWAYS, packed=True)


def test_print_order_ties_and_the_fallback_to_the_full_sort():
    """The print order is (contig, left position, input index of the seed).  The short sort orders on contig and position
    and puts seeds that share both into input order run by run: 5 clusters at one left position (a run the fix-up handles),
    90 clusters at another (longer than it handles: the ctx falls back to the two full sorts, for this call and for good),
    and anchors far beyond their contig's end (a left position outside the short sort's mapped range: fallback as well)."""
    rng = np.random.Generator(np.random.PCG64(99))
    A, L = 20, 100
    g = synth.make_genome([("c0", 400000), ("c1", 400000)], seed=41)

    def case(n_partners, x, bogus=False):
        seqs, frags, hits = [], [], []
        for j in range(n_partners):                               # one left breakpoint, many right partners: one class each
            y = 5000 + 4000 * j
            while g.seqs[1][y - 1] == g.seqs[0][x]:               # no shared base before the junction: the split scan's
                y += 1                                            # lowest-k rule (detect.rs:143-146) leaves lpos at x
            for i in range(3):
                b = int(rng.integers(A + 5, L - A - 5))
                read = np.concatenate([g.seqs[0][x - b + 1:x + 1], g.seqs[1][y:y + L - b]])
                seqs.append(read.tobytes()); frags.append(b"p%d_%d" % (j, i))
                hits.append((x - b + 2, y + L - A - b + 1, 0, 1, 3))
        for i in range(3 if bogus else 0):                        # anchors 300 kb past the end of c0: windows of N, one cluster
            seqs.append(g.seqs[1][1000:1000 + L].tobytes()); frags.append(b"far_%d" % i)
            hits.append((700000, 1000 + L - A + 1, 0, 1, 3))
        order = rng.permutation(len(seqs))
        return engine.make_batch([seqs[i] for i in order], [frags[i] for i in order], [hits[i] for i in order])

    for flags in (0, BF_FLAG_GROUPS_ALWAYS):
        res, ref = check_against_oracle(g, case(5, 150000), A, flags=flags)
        assert ref.n_clusters == 5 and len(set(res.clusters["lpos"].tolist())) == 1
        res, ref = check_against_oracle(g, case(90, 250000), A, flags=flags)
        assert ref.n_clusters == 90
        res, ref = check_against_oracle(g, case(4, 100000, bogus=True), A, flags=flags)
        assert ref.n_clusters == 5
    # one ctx through all three: the fallback is sticky and the results stay right
    og = ol.OracleGenome(g.names, g.seqs)
    eng = engine.Engine(A, 5000, 2)
    try:
        eng.set_genome(g.names, g.seqs)
        for b in (case(5, 150000), case(90, 250000), case(5, 150000), case(4, 100000, bogus=True), case(7, 120000)):
            eng.reset()
            eng.submit(b)
            assert eng.format_sv(eng.finish()) == ol.detect_batch(og, g.names, b, A, 5000, 2).sv
    finally:
        eng.close()


def test_format_sv_size_query_truncation_and_threads():
    """bf_format_sv: NULL buffer = size query; a short buffer receives a prefix and the full size is still
    returned; a result large enough for the multi-threaded fill equals the oracle's text."""
    import ctypes as C
    contigs, gk, spec = synth.config(1, scale=0.02, genome_scale=0.01)          # ~12 MB of .sv: several fill threads
    spec.margin = 10000 + 2 * spec.read_len + 128
    g = synth.make_genome(contigs, **gk)
    b = synth.make_reads(g, spec)
    ref = ol.detect_batch(ol.OracleGenome(g.names, g.seqs), g.names, b, spec.anchor_len, 5000, 2)
    eng = engine.Engine(spec.anchor_len, 5000, 2)
    try:
        eng.set_genome(g.names, g.seqs)
        eng.submit(b)
        eng.finish(copy=False)
        r = eng._last
        n = eng.lib.bf_format_sv(eng.h, C.byref(r), None, 0)
        assert n == len(ref.sv) > (8 << 20)
        full = C.create_string_buffer(n)
        assert eng.lib.bf_format_sv(eng.h, C.byref(r), full, n) == n
        assert full.raw == ref.sv
        short = C.create_string_buffer(b"\xee" * 1000, 1000)
        assert eng.lib.bf_format_sv(eng.h, C.byref(r), short, 777) == n
        assert short.raw[:777] == ref.sv[:777] and short.raw[777:] == b"\xee" * 223
    finally:
        eng.close()
