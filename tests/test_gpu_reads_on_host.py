"""BF_FLAG_READS_ON_HOST: junction records that name the read instead of carrying its bases (record kind 3,
include/breakfast_hip.h "Junction records").  The caller keeps the batches it submitted; bf_format_sv and
bf_junction_text_of read the bases there.  Same bytes on stdout as without the flag, a quarter of the result bytes."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
from conftest import golden_names, load_golden
from test_gpu_parity import check_against_oracle, run_gpu_text

from breakfast_amd import engine, synth
from breakfast_amd.engine import BF_FLAG_FORCE_EXACT, BF_FLAG_GROUPS_ALWAYS, BF_FLAG_READS_ON_HOST

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("name", golden_names())
def test_golden_fixture(name, packed):
    fa, lines, (A, F, E), sv = load_golden(name)
    try:
        out = run_gpu_text(fa, lines, A, F, E, flags=BF_FLAG_READS_ON_HOST, packed=packed)
    except ValueError:
        assert name == "hz_ref_alphabet" and packed          # lower-case reads have no 4-bit form
        pytest.skip("reads with bytes outside =ACMGRSVTWYHKDBN are handed over as ASCII")
    assert out == sv


@pytest.mark.parametrize("flags", [0, BF_FLAG_FORCE_EXACT, BF_FLAG_GROUPS_ALWAYS])
@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("cfg,scale,gscale", [(0, 0.05, 0.02), (1, 0.004, 0.002), (3, 0.001, 0.002), (4, 0.002, 0.002)])
def test_scaled_baseline_config(cfg, scale, gscale, packed, flags):
    """every per-read value, every junction string (expanded from the caller's arrays) and the .sv"""
    contigs, gk, spec = synth.config(cfg, scale=scale, genome_scale=gscale)
    g = synth.make_genome(contigs, **gk)
    b = synth.make_reads(g, spec)
    if packed:
        b.seq_bytes = b.seq_bytes & np.uint8(0xDF)
    res, ref = check_against_oracle(g, b, spec.anchor_len, flags=flags | BF_FLAG_READS_ON_HOST, packed=packed)
    assert ref.n_clusters > 0


def test_long_reads_and_many_mismatches():
    """reads above 256 bases list their lower-case positions as u16; N bases and IUPAC letters print from the host copy"""
    contigs = [("c%d" % i, 120000) for i in range(3)]
    g = synth.make_genome(contigs, seed=19, lower_runs=[(1, 30000, 20000)])
    for L in (51, 129, 257, 701):
        spec = synth.ReadSpec(n_reads=900, read_len=L, anchor_len=25, n_events=15, singleton_frac=0.3, seed=300 + L,
                              margin=1300, subst_rate=0.05, n_rate=0.01)
        b = synth.make_reads(g, spec)
        check_against_oracle(g, b, 25, flags=BF_FLAG_READS_ON_HOST)
        b.seq_bytes = b.seq_bytes & np.uint8(0xDF)
        check_against_oracle(g, b, 25, flags=BF_FLAG_READS_ON_HOST, packed=True)


def test_records_name_their_reads_and_are_small():
    contigs, gk, spec = synth.config(1, scale=0.004, genome_scale=0.002)
    g = synth.make_genome(contigs, **gk)
    b = synth.make_reads(g, spec)
    ref = ol.detect_batch(ol.OracleGenome(g.names, g.seqs), g.names, b, spec.anchor_len, 5000, 2, keep=True)
    sizes = {}
    for flags in (0, BF_FLAG_READS_ON_HOST):
        eng = engine.Engine(spec.anchor_len, 5000, 2, flags=flags)
        try:
            eng.set_genome(g.names, g.seqs)
            half = b.n // 2
            eng.submit(b.slice(0, half))                      # two batches: the second one's reads are numbered on
            eng.submit(engine.Batch(b.seq_bytes, b.seq_off[half:], b.frag_bytes, b.frag_off[half:], b.hits[half:]))
            res = eng.finish()
            assert eng.format_sv(res) == ref.sv
            sizes[flags] = res.junction_bytes.size
            if not flags:
                continue
            lib = eng.lib
            lib.bf_junction_record_bytes.restype = C.c_int64
            for k in range(res.reads.shape[0]):
                off = int(res.reads["junction_off"][k])
                head, read = np.frombuffer(res.junction_bytes[off:off + 8].tobytes(), "<u4")
                assert head & 3 == 3 and int(read) == int(res.reads["index"][k])       # one ctx: index = resident number
                i = int(read)
                want = bytes(ref.junction[int(ref.junction_off[i]):int(ref.junction_off[i + 1])])
                assert eng.junction_text_of(res.junction_bytes, off) == want
                with pytest.raises(engine.BreakfastError):                           # no ctx: no bases
                    engine.junction_text(res.junction_bytes, off)
                stored = lib.bf_junction_record_bytes(C.c_void_p(res.junction_bytes.ctypes.data + off), res.junction_bytes.size - off)
                nxt = int(res.reads["junction_off"][k + 1]) if k + 1 < res.reads.shape[0] else res.junction_bytes.size
                assert stored == nxt - off
            # after a reset the ctx has forgotten the arrays: old records do not expand, new reads do
            eng.reset()
            with pytest.raises(engine.BreakfastError):
                eng.junction_text_of(res.junction_bytes, 0)
            eng.submit(b)
            assert eng.format_sv(eng.finish()) == ref.sv
        finally:
            eng.close()
    assert sizes[BF_FLAG_READS_ON_HOST] * 3 < sizes[0]


def test_not_with_more_than_one_rank():
    eng = engine.Engine(20, 5000, 2, flags=BF_FLAG_READS_ON_HOST)
    try:
        eng.set_genome([b"c"], [np.frombuffer(b"ACGT" * 100, np.uint8)])
        with pytest.raises(engine.BreakfastError):
            eng.shard_init_rccl(0, 1, 0, engine.Engine.shard_unique_id())
        with pytest.raises(engine.BreakfastError):
            eng.shard_init_host(0, 1, 0, "/bf_test_roh", 1 << 16)
    finally:
        eng.close()


def test_mutated_reference_records_are_refused_or_stay_in_bounds():
    """bf_junction_text_of on reference records with flipped bytes: an error or a string of some resident read's length,
    never a write past the buffer (a guard band behind it survives), never a read past the caller's arrays."""
    contigs, gk, spec = synth.config(1, scale=0.002, genome_scale=0.002)
    g = synth.make_genome(contigs, **gk)
    b = synth.make_reads(g, spec)
    eng = engine.Engine(spec.anchor_len, 5000, 2, flags=BF_FLAG_READS_ON_HOST)
    try:
        eng.set_genome(g.names, g.seqs)
        eng.submit(b)
        res = eng.finish()
        lib = eng.lib
        rng = np.random.default_rng(3)
        offs = res.reads["junction_off"].astype(np.int64)
        ends = np.append(offs[1:], res.junction_bytes.size)
        n_ok = n_bad = 0
        for trial in range(3000):
            k = int(rng.integers(0, offs.size))
            rec = res.junction_bytes[offs[k]:ends[k]].copy()
            for _ in range(int(rng.integers(1, 3))):
                rec[int(rng.integers(0, rec.size))] = int(rng.integers(0, 256))
            if trial % 5 == 0:
                rec = rec[:int(rng.integers(0, rec.size))].copy()
            ptr = rec.ctypes.data if rec.size else None
            n = lib.bf_junction_text_of(eng.h, ptr, rec.size, None, 0)
            if n < 0:
                n_bad += 1
                continue
            n_ok += 1
            assert 1 <= n <= 65536
            buf = np.full(n + 64, 0xEE, np.uint8)
            assert lib.bf_junction_text_of(eng.h, ptr, rec.size, buf.ctypes.data, n) == n
            assert np.all(buf[n:] == 0xEE)
        assert n_ok > 100 and n_bad > 100
    finally:
        eng.close()
