"""The host -> device hand-over (bf_submit): asynchronous, piece by piece, with the per-read kernels queued behind
every uploaded piece (the reference consumes reads as they arrive, detect.rs:76-188).  However a read set is cut
into batches and pieces, pinned or pageable, the .sv must be the oracle's, byte for byte.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol

from breakfast_amd import engine, synth
from breakfast_amd.engine import BF_FLAG_FORCE_EXACT, BF_FLAG_KEEP_EVIDENCE

pytestmark = pytest.mark.gpu


def case(n_reads=24000, L=120, seed=5, n_frac=0.0, upper=False):
    contigs = [("chr1", 300000), ("chr10", 200000), ("chr2", 250000)]
    g = synth.make_genome(contigs, seed=seed, n_runs=[(0, 5000, 2000)], lower_runs=[(2, 100000, 30000)])
    spec = synth.ReadSpec(n_reads=n_reads, read_len=L, anchor_len=20, n_events=250, singleton_frac=0.3, mate_frac=0.4,
                          seed=seed + 1, margin=300, subst_rate=0.01)
    b = synth.make_reads(g, spec)
    if upper:                                   # reads cut out of the soft-masked stretch are lower case: not BAM's alphabet
        b.seq_bytes = b.seq_bytes & np.uint8(0xDF)
    if n_frac:
        rng = np.random.default_rng(seed)
        sb = b.seq_bytes.copy()
        pos = rng.integers(0, sb.size, size=int(sb.size * n_frac))
        sb[pos] = ord("N")
        b.seq_bytes = sb
    return g, b


def oracle_sv(g, b, A=20, F=5000, E=2):
    return ol.detect_batch(ol.OracleGenome(g.names, g.seqs), g.names, b, A, F, E).sv


def concat(batches):
    seq = np.concatenate([x.seq_bytes for x in batches])
    frag = np.concatenate([x.frag_bytes for x in batches])
    so, fo = [np.zeros(1, np.uint64)], [np.zeros(1, np.uint64)]
    for x in batches:
        so.append(x.seq_off[1:] - x.seq_off[0] + so[-1][-1])
        fo.append(x.frag_off[1:] - x.frag_off[0] + fo[-1][-1])
    return engine.Batch(seq, np.concatenate(so), frag, np.concatenate(fo), np.concatenate([x.hits for x in batches]))


def cuts_for(n, sizes):
    out, a, k = [], 0, 0
    while a < n:
        b = min(n, a + sizes[k % len(sizes)])
        out.append((a, b))
        a, k = b, k + 1
    return out


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("piece", [0, 1000, 4097])
@pytest.mark.parametrize("sizes", [(10 ** 9,), (1, 7, 5000, 300, 12000), (2500,)])
@pytest.mark.parametrize("pinned", [False, True])
def test_any_cut_into_batches_and_pieces_gives_the_oracle_bytes(piece, sizes, pinned, packed):
    g, b = case(L=121 if packed else 120, upper=packed)    # an odd length: a padding nibble per read
    ref = oracle_sv(g, b)
    eng = engine.Engine(20, 5000, 2, piece_reads=piece, host_threads=4)
    try:
        eng.set_genome(g.names, g.seqs)
        held = []
        for a, e in cuts_for(b.n, sizes):
            part = b.slice(a, e)
            if packed:
                part = engine.pack_batch(part)
            if pinned:
                part = eng.pin_batch(part)
                held.append(part)                       # pinned batches stay untouched until the hand-over is over
            eng.submit(part)
        res = eng.finish()
        assert res.n_evidence == b.n
        assert eng.format_sv(res) == ref
        # again without new reads: the whole path runs again on the resident reads
        assert eng.format_sv(eng.finish()) == ref
        # explicit stage calls: every read localised in one go
        eng.localise()
        assert eng.format_sv(eng.clusterise()) == ref
    finally:
        eng.close()


def test_evidence_of_reads_localised_behind_their_upload_equals_one_launch():
    g, b = case(n_frac=0.002)
    outs = []
    for piece, sizes in ((0, (10 ** 9,)), (777, (5000, 1, 9000))):
        eng = engine.Engine(20, 5000, 2, piece_reads=piece, flags=BF_FLAG_KEEP_EVIDENCE)
        try:
            eng.set_genome(g.names, g.seqs)
            for a, e in cuts_for(b.n, sizes):
                eng.submit(b.slice(a, e))
            res = eng.finish()
            ev = eng.evidence()
            by, off = eng.render_junctions(np.arange(b.n, dtype=np.uint32))
            outs.append((eng.format_sv(res), ev, by, off))
        finally:
            eng.close()
    assert outs[0][0] == outs[1][0] == oracle_sv(g, b)
    for k in ("class_key", "lpos", "rpos", "bp", "mask"):
        assert np.array_equal(outs[0][1][k], outs[1][1][k]), k
    assert np.array_equal(outs[0][1]["flags"] & 0x17, outs[1][1]["flags"] & 0x17)
    assert np.array_equal(outs[0][2], outs[1][2]) and np.array_equal(outs[0][3], outs[1][3])


@pytest.mark.parametrize("order", ["short_first", "long_first"])
def test_read_lengths_that_change_the_evidence_layout_between_pieces(order):
    """The evidence layout depends on the longest read seen so far (mask words per render record, the kernel
    instance): a later piece with longer reads makes the library start over; shorter ones do not."""
    g, b1 = case(n_reads=9000, L=100, seed=21)
    _, b2 = case(n_reads=9000, L=200, seed=21)        # same genome (same seed), other reads
    _, b3 = case(n_reads=3000, L=300, seed=21)        # longer than the packed kernel: exact path, text records
    parts = [b1, b2, b3] if order == "short_first" else [b3, b2, b1]
    whole = concat(parts)
    ref = oracle_sv(g, whole)
    eng = engine.Engine(20, 5000, 2, piece_reads=2000)
    try:
        eng.set_genome(g.names, g.seqs)
        for p in parts:
            eng.submit(p)
        assert eng.format_sv(eng.finish()) == ref
    finally:
        eng.close()


@pytest.mark.parametrize("packed", [False, True])
def test_finish_then_more_reads_then_finish(packed):
    """Appending after a bf_finish: only the new reads are localised, the exact kernel only sees its new list
    entries, and the result is the oracle's for the union (also with every read on the exact path).  With 4-bit
    batches the device rebuilds the offsets of every appended piece from its lengths, starting where the resident
    bytes end."""
    g, b = case(n_frac=0.003, upper=packed)
    for flags in (0, BF_FLAG_FORCE_EXACT):
        eng = engine.Engine(20, 5000, 2, piece_reads=3000, flags=flags)
        try:
            eng.set_genome(g.names, g.seqs)
            cut = [0, 7000, 7001, 15000, b.n]
            for a, e in zip(cut[:-1], cut[1:]):
                part = b.slice(a, e)
                eng.submit(engine.pack_batch(part) if packed else part)
                assert eng.format_sv(eng.finish()) == oracle_sv(g, b.slice(0, e))
        finally:
            eng.close()


def test_reset_keeps_the_memory_and_reserve_sizes_it():
    g, b = case()
    ref = oracle_sv(g, b)
    eng = engine.Engine(20, 5000, 2, piece_reads=5000)
    try:
        eng.set_genome(g.names, g.seqs)
        eng.reserve(b.n, int(b.seq_off[-1]), int(b.frag_off[-1]))
        _, bytes0 = eng.stats()
        pb = eng.pin_batch(b)
        for _ in range(3):
            eng.reset()
            eng.submit(pb)
            assert eng.format_sv(eng.finish()) == ref
        eng.reset()
        half = b.slice(0, b.n // 2)
        eng.submit(half)
        assert eng.format_sv(eng.finish()) == oracle_sv(g, half)
        assert eng.stats()[0] == half.n
    finally:
        eng.close()


def test_a_rejected_batch_leaves_the_ctx_as_it_was():
    g, b = case()
    good = b.slice(0, 10000)
    ref = oracle_sv(g, good)
    eng = engine.Engine(20, 5000, 2, piece_reads=1500, host_threads=3)
    try:
        eng.set_genome(g.names, g.seqs)
        eng.submit(good)
        bad = b.slice(10000, 20000)
        # an unknown contig id in the batch's last piece: earlier pieces were already uploaded and localised
        h = bad.hits.copy(); h["contig3"][9000] = 99
        with pytest.raises(engine.BreakfastError) as e:
            eng.submit(engine.Batch(bad.seq_bytes, bad.seq_off, bad.frag_bytes, bad.frag_off, h))
        assert e.value.code == -1 and "read 9000 " in str(e.value)
        # offsets that run backwards; an anchor position beyond 2^31; a read shorter than 2 * anchor_len
        so = bad.seq_off.copy(); so[5000] = so[4999] - 1
        with pytest.raises(engine.BreakfastError) as e:
            eng.submit(engine.Batch(bad.seq_bytes, so, bad.frag_bytes, bad.frag_off, bad.hits))
        assert e.value.code == -1 and "monotone" in str(e.value)
        h = bad.hits.copy(); h["pos5"][3] = 0x80000000
        with pytest.raises(engine.BreakfastError) as e:
            eng.submit(engine.Batch(bad.seq_bytes, bad.seq_off, bad.frag_bytes, bad.frag_off, h))
        assert e.value.code == -5
        so = bad.seq_off.copy(); so[8001:] -= 100            # read 8000 has 20 bases
        with pytest.raises(engine.BreakfastError) as e:
            eng.submit(engine.Batch(bad.seq_bytes, so, bad.frag_bytes, bad.frag_off, bad.hits))
        assert e.value.code == -5 and "shorter than 2 * anchor_len" in str(e.value)
        assert eng.stats()[0] == good.n
        assert eng.format_sv(eng.finish()) == ref
        # and the ctx still takes the good version of the batch
        eng.submit(bad)
        assert eng.format_sv(eng.finish()) == oracle_sv(g, b.slice(0, 20000))
    finally:
        eng.close()


def test_a_rejected_4bit_batch_leaves_the_ctx_as_it_was():
    """The same for bf_submit_packed: pieces of the rejected batch were uploaded, localised and had their offsets rebuilt
    on the device before the offending read was seen; nothing of it stays, and the next batch takes its place."""
    import dataclasses
    g, b = case(upper=True)
    good, bad = engine.pack_batch(b.slice(0, 10000)), engine.pack_batch(b.slice(10000, 20000))
    ref = oracle_sv(g, b.slice(0, 10000))
    eng = engine.Engine(20, 5000, 2, piece_reads=1500, host_threads=3)
    try:
        eng.set_genome(g.names, g.seqs)
        eng.submit(good)
        h = bad.hits.copy(); h["contig3"][9000] = 99                       # in the batch's last piece
        with pytest.raises(engine.BreakfastError) as e:
            eng.submit(dataclasses.replace(bad, hits=h))
        assert e.value.code == -1 and "read 9000 " in str(e.value)
        ln = bad.seq_len.copy(); ln[4000] += 2                             # a length that does not match its bytes
        with pytest.raises(engine.BreakfastError) as e:
            eng.submit(dataclasses.replace(bad, seq_len=ln))
        assert e.value.code == -1 and "read 4000 " in str(e.value)
        assert eng.stats()[0] == good.n
        assert eng.format_sv(eng.finish()) == ref
        eng.submit(bad)                                                    # and the ctx still takes the good version
        assert eng.format_sv(eng.finish()) == oracle_sv(g, b.slice(0, 20000))
        pinned = eng.pin_batch(engine.pack_batch(b.slice(20000, b.n)))     # ... and more, from bf_host_alloc memory
        eng.submit(pinned)
        assert eng.format_sv(eng.finish()) == oracle_sv(g, b)
    finally:
        eng.close()


@pytest.mark.parametrize("piece", [0, 1000, 4097])
@pytest.mark.parametrize("sizes", [(10 ** 9,), (1, 7, 5000, 300, 12000)])
@pytest.mark.parametrize("pinned", [False, True])
def test_reads_at_two_bits_a_base_give_the_oracle_bytes(piece, sizes, pinned):
    """bf_submit_packed2: two bits a base on PCIe, the bases that are none of ACGT as exceptions, expanded to the resident
    4-bit form behind every uploaded piece.  Odd read lengths (123: a last 2-bit byte with three bases, a padding nibble),
    N in one read in five hundred bases, every cut into batches and pieces, pinned or pageable; batches alternate between
    the 4-bit and the 2-bit call (one ctx, one resident form): the .sv is the oracle's, and the resident bytes are the ones
    bf_submit_packed would have left (same evidence, same junction strings, through both localise kernels)."""
    g, b = case(L=123, upper=True, n_frac=0.002)
    ref = oracle_sv(g, b)
    eng = engine.Engine(20, 5000, 2, piece_reads=piece, host_threads=4)
    try:
        eng.set_genome(g.names, g.seqs)
        held, n_exc = [], 0
        for k, (a, e) in enumerate(cuts_for(b.n, sizes)):
            part = engine.pack_batch(b.slice(a, e))
            if k % 3 != 2:
                part = engine.pack2_batch(part)
                n_exc += int(part.exc_read.size)
            if pinned:
                part = eng.pin_batch(part)
                held.append(part)
            eng.submit(part)
        assert n_exc > 100
        res = eng.finish()
        assert res.n_evidence == b.n
        assert eng.format_sv(res) == ref
        assert eng.format_sv(eng.finish()) == ref
    finally:
        eng.close()


def test_two_bit_reads_leave_the_evidence_of_four_bit_reads():
    g, b = case(L=150, upper=True, n_frac=0.003)
    pb = engine.pack_batch(b)
    outs = []
    for form, flags in ((pb, 0), (engine.pack2_batch(pb), 0), (engine.pack2_batch(pb), BF_FLAG_FORCE_EXACT)):
        eng = engine.Engine(20, 5000, 2, piece_reads=3001, host_threads=3, flags=flags | BF_FLAG_KEEP_EVIDENCE)
        try:
            eng.set_genome(g.names, g.seqs)
            eng.submit(form)
            res = eng.finish()
            ev = eng.evidence()
            by, off = eng.render_junctions(np.arange(b.n, dtype=np.uint32))
            outs.append((eng.format_sv(res), {k: ev[k].copy() for k in ("bp", "lpos", "rpos", "class_key", "mask")}, by.tobytes(), off.copy()))
        finally:
            eng.close()
    for o in outs[1:]:
        assert o[0] == outs[0][0] and o[2] == outs[0][2] and np.array_equal(o[3], outs[0][3])
        for k in ("bp", "lpos", "rpos", "mask"):
            assert np.array_equal(o[1][k], outs[0][1][k]), k
    assert np.array_equal(outs[1][1]["class_key"], outs[0][1]["class_key"])


def test_a_rejected_2bit_batch_leaves_the_ctx_as_it_was():
    import dataclasses
    g, b = case(upper=True, n_frac=0.002)
    good, bad = engine.pack2_batch(engine.pack_batch(b.slice(0, 10000))), engine.pack2_batch(engine.pack_batch(b.slice(10000, 20000)))
    assert bad.exc_read.size > 10
    eng = engine.Engine(20, 5000, 2, piece_reads=1500, host_threads=3)
    try:
        eng.set_genome(g.names, g.seqs)
        eng.submit(good)
        ln = bad.seq_len.copy(); ln[4000] += 4                             # a length that does not match its bytes
        with pytest.raises(engine.BreakfastError) as e:
            eng.submit(dataclasses.replace(bad, seq_len=ln))
        assert e.value.code == -1 and "read 4000 " in str(e.value)
        for field, j, value, what in (("exc_code", 3, 4, "code 4"), ("exc_code", 3, 16, "code 16"), ("exc_pos", 5, 60000, "names base"),
                                      ("exc_read", 2, 10000, "names base"), ("exc_read", 7, 0, "not ascending")):
            arr = getattr(bad, field).copy(); arr[j] = value
            with pytest.raises(engine.BreakfastError) as e:
                eng.submit(dataclasses.replace(bad, **{field: arr}))
            assert e.value.code == -1 and what in str(e.value), str(e.value)
        assert eng.stats()[0] == good.n
        assert eng.format_sv(eng.finish()) == oracle_sv(g, b.slice(0, 10000))
        eng.submit(bad)
        eng.submit(engine.pack_batch(b.slice(20000, b.n)))                 # a 4-bit batch behind two 2-bit ones
        assert eng.format_sv(eng.finish()) == oracle_sv(g, b)
        eng.reset()
        eng.submit(bad)                                                    # after a reset the 2-bit scratch starts over
        assert eng.format_sv(eng.finish()) == oracle_sv(g, b.slice(10000, 20000))
    finally:
        eng.close()
    roh = engine.Engine(20, 5000, 2, flags=engine.BF_FLAG_READS_ON_HOST)
    try:
        roh.set_genome(g.names, g.seqs)
        with pytest.raises(engine.BreakfastError) as e:
            roh.submit(good)
        assert e.value.code == -4 and "READS_ON_HOST" in str(e.value)
    finally:
        roh.close()


def test_pageable_batches_may_be_freed_on_return_and_pinned_ones_after_the_wait():
    g, b = case()
    ref = oracle_sv(g, b)
    eng = engine.Engine(20, 5000, 2, piece_reads=4000, host_threads=8)
    try:
        eng.set_genome(g.names, g.seqs)
        half = b.n // 2
        part = b.slice(0, half)
        arrs = [np.array(x, copy=True) for x in (part.seq_bytes, part.seq_off, part.frag_bytes, part.frag_off, part.hits)]
        eng.submit(engine.Batch(*arrs))
        for a in arrs:                                   # pageable: staged before bf_submit returned
            a.view(np.uint8)[...] = 0xAB
        pinned = eng.pin_batch(b.slice(half, b.n))
        eng.submit(pinned)
        eng.submit_wait()
        for a in (pinned.seq_bytes, pinned.frag_bytes):  # pinned: free for reuse once the uploads are through
            a[...] = 0x21
        pinned.hits.view(np.uint8)[...] = 0xFF
        assert eng.format_sv(eng.finish()) == ref
    finally:
        eng.close()


@pytest.mark.parametrize("fill", [False, True])
def test_genome_in_pieces_any_order_and_filled_in_place(fill):
    """bf_genome_begin / bf_genome_write | bf_genome_fill / bf_genome_end: pieces in any order, written by the caller's
    callback straight into the staging memory (fill) or copied from its buffer; same table as bf_genome_set."""
    contigs, gk, spec = synth.config(1, scale=0.002, genome_scale=0.002)
    g = synth.make_genome(contigs, **gk)
    b = synth.make_reads(g, spec)
    ref = ol.detect_batch(ol.OracleGenome(g.names, g.seqs), g.names, b, spec.anchor_len, 5000, 2)
    eng = engine.Engine(spec.anchor_len, 5000, 2)
    try:
        rng = np.random.default_rng(5)
        eng.set_genome_in_pieces(g.names, g.seqs, piece=37777, fill=fill, order=lambda n: rng.permutation(n))
        eng.submit(b)
        assert eng.format_sv(eng.finish()) == ref.sv
    finally:
        eng.close()


def genome_of_every_byte(rng, n):
    """ACGT with soft-masked stretches, N runs of every kind of length (across 64-base blocks and 4096-base waves of the
    pack kernel), isolated IUPAC letters in both cases, and bytes no FASTA should hold but the reference would keep."""
    s = rng.choice(np.frombuffer(b"ACGT", np.uint8), n)
    for _ in range(n // 3000 + 3):                                  # soft-masked stretches
        a = int(rng.integers(0, n)); m = int(rng.integers(1, 700))
        s[a:a + m] |= 0x20
    for m in (1, 2, 63, 64, 65, 127, 128, 129, 4095, 4096, 4097, 9000):     # runs of one byte
        for c in b"Nn":
            if m < n // 4:
                a = int(rng.integers(0, n - m)); s[a:a + m] = c
    for _ in range(n // 500 + 5):                                   # single letters and short mixed runs
        a = int(rng.integers(0, n)); m = int(rng.integers(1, 6))
        s[a:a + m] = rng.choice(np.frombuffer(b"RYKMSWBDHVNrykmswbdhvn*-.Xx", np.uint8), min(m, n - a))
    for a in (0, 63, 64, n - 1, n - 64):                            # block edges
        if 0 <= a < n: s[a] = ord("N")
    return s


def test_every_byte_of_the_genome_can_be_rebuilt():
    """The raw image is released after the pack (2-bit codes + exception plane + lower-case plane + a table of the runs of
    every other byte stay): bf_genome_read must give back the bytes bf_genome_write was given, whatever they are; and the
    byte-exact kernel and the raw signatures, which ask for such bytes, still agree with the oracle on such a genome."""
    rng = np.random.default_rng(77)
    lens = [200000, 1, 63, 64, 65, 4096, 4097, 131072 + 5, 70000]
    names = [b"c%02d" % i for i in range(len(lens))]
    seqs = [genome_of_every_byte(rng, n) for n in lens]
    seqs.append(np.full(10000, ord("N"), np.uint8)); names.append(b"allN")
    seqs.append(np.frombuffer(b"ACGT" * 2500, np.uint8).copy()); names.append(b"plain")
    eng = engine.Engine(20, 5000, 2)
    try:
        eng.set_genome_in_pieces(names, seqs, piece=33333, order=lambda n: rng.permutation(n))
        for c, s in enumerate(seqs):
            assert np.array_equal(eng.genome_read(c, 0, s.size), s), names[c]
        assert np.array_equal(eng.genome_read(0, 12345, 4099), seqs[0][12345:12345 + 4099])
        assert eng.genome_read(0, 200000, 0).size == 0
        with pytest.raises(engine.BreakfastError):
            eng.genome_read(0, 199999, 2)
        eng.set_genome(names[:3], seqs[:3])                           # a second genome in the same ctx
        assert np.array_equal(eng.genome_read(0, 0, seqs[0].size), seqs[0])
    finally:
        eng.close()
    # reads cut out of such a genome (upper-cased, as a sequencer gives them): windows over N runs, lower case and IUPAC
    big = [0, 7, 8]
    g = synth.Genome([names[i] for i in big], [seqs[i] for i in big])
    spec = synth.ReadSpec(n_reads=6000, read_len=100, anchor_len=20, n_events=150, singleton_frac=0.3, seed=9, margin=300)
    b = synth.make_reads(g, spec)
    low = (b.seq_bytes >= ord("a")) & (b.seq_bytes <= ord("z"))
    b.seq_bytes[low] &= np.uint8(0xDF)
    ref = ol.detect_batch(ol.OracleGenome(g.names, g.seqs), g.names, b, 20, 5000, 2)
    assert ref.n_clusters > 0
    for flags in (0, engine.BF_FLAG_FORCE_EXACT):
        e2 = engine.Engine(20, 5000, 2, flags=flags | engine.BF_FLAG_KEEP_EVIDENCE)
        try:
            e2.set_genome(g.names, g.seqs)
            e2.submit(b)
            assert e2.format_sv(e2.finish()) == ref.sv
            if flags == 0:
                ev = e2.evidence()
                assert ((ev["flags"] & engine.BF_EV_SIGX) != 0).sum() > 50, "no raw signatures built by the packed kernel"
                assert ((ev["flags"] & engine.BF_EV_EXACT) != 0).sum() < b.n
        finally:
            e2.close()


@pytest.mark.parametrize("begin_first", [False, True])
def test_genome_from_its_image(tmp_path, begin_first):
    """bf_genome_image_write -> bf_genome_image_open -> bf_genome_set_image: the resident genome is the one bf_genome_set
    builds from the bytes (every byte read back; the oracle's stdout on reads over N runs, lower case and IUPAC letters,
    packed kernel and byte-exact kernel), with 0.26 of the bytes on the wire."""
    import test_genome_image as tgi
    names, seqs = tgi.case_genome(seed=21)
    path = str(tmp_path / "g.bf2bit")
    engine.write_genome_image(path, names, seqs)
    img = engine.GenomeImage(path)
    assert img.upload_bytes < 0.4 * sum(s.size for s in seqs)          # (a genome dense in odd bytes; hg38: 0.27)
    big = [0, 7, 8, 9]
    g = synth.Genome([names[i] for i in big], [seqs[i] for i in big])
    spec = synth.ReadSpec(n_reads=6000, read_len=100, anchor_len=20, n_events=150, singleton_frac=0.3, seed=9, margin=300)
    b = synth.make_reads(g, spec)
    low = (b.seq_bytes >= ord("a")) & (b.seq_bytes <= ord("z"))
    b.seq_bytes[low] &= np.uint8(0xDF)
    b.hits["contig5"] = np.asarray(big, np.uint32)[b.hits["contig5"]]          # contig ids of the whole genome
    b.hits["contig3"] = np.asarray(big, np.uint32)[b.hits["contig3"]]
    ref = ol.detect_batch(ol.OracleGenome(names, seqs), names, b, 20, 5000, 2)
    assert ref.n_clusters > 0
    try:
        for flags in (0, engine.BF_FLAG_FORCE_EXACT):
            eng = engine.Engine(20, 5000, 2, flags=flags)
            try:
                if begin_first:
                    eng.set_genome(names[:2], seqs[:2])                        # another genome first: the ctx starts over
                eng.set_genome_image(img, begin_first=begin_first)
                for c, s_ in enumerate(seqs):
                    assert np.array_equal(eng.genome_read(c, 0, s_.size), s_), names[c]
                eng.submit(b)
                assert eng.format_sv(eng.finish()) == ref.sv
                eng.reset()
                eng.set_genome(names, seqs)                                    # and back to the bytes
                eng.submit(b)
                assert eng.format_sv(eng.finish()) == ref.sv
            finally:
                eng.close()
    finally:
        img.close()


def test_damaged_images_are_refused_by_the_device_checks(tmp_path):
    """Tables that pass the file checks (sizes, sections) but do not ascend, overlap or leave the genome."""
    import struct
    import test_genome_image as tgi
    names, seqs = tgi.case_genome(seed=22)
    path = str(tmp_path / "g.bf2bit")
    engine.write_genome_image(path, names, seqs)
    raw = bytearray(open(path, "rb").read())
    f = tgi.HEADER.unpack(bytes(raw[:192]))
    n_lower, n_other, off_lower_start, off_other_start = f[6], f[7], f[13], f[15]
    assert n_lower > 10 and n_other > 10

    def with_word(off, value):
        d = bytearray(raw); d[off:off + 8] = struct.pack("<Q", value); return bytes(d)
    first_other = struct.unpack("<Q", raw[off_other_start:off_other_start + 8])[0]
    cases = {"descending": with_word(off_lower_start + 8 * 5, 0), "outside": with_word(off_lower_start + 8 * (n_lower - 1), 1 << 40),
             "both_kinds": with_word(off_lower_start, first_other)}
    eng = engine.Engine(20, 5000, 2)
    try:
        for name, data in cases.items():
            p = str(tmp_path / name)
            open(p, "wb").write(data)
            img = engine.GenomeImage(p)
            with pytest.raises(engine.BreakfastError) as e:
                eng.set_genome_image(img)
            assert "damaged" in str(e.value), name
            img.close()
        img = engine.GenomeImage(path)                                         # the ctx is still good for a sound image
        eng.set_genome_image(img)
        assert np.array_equal(eng.genome_read(0, 0, seqs[0].size), seqs[0])
        img.close()
        # bf_genome_begin_image + bf_genome_end with nothing in between: a genome of 'A's (as unwritten bytes are after
        # bf_genome_begin), not the tables of the genome before
        n2 = 2
        eng._check(eng.lib.bf_genome_begin_image(eng.h, n2, (C.c_char_p * n2)(*names[:n2]), (C.c_uint64 * n2)(*[x.size for x in seqs[:n2]])))
        eng._check(eng.lib.bf_genome_end(eng.h))
        assert eng.genome_read(0, 0, seqs[0].size).tobytes() == b"A" * seqs[0].size
        with pytest.raises(engine.BreakfastError):                             # bytes after bf_genome_begin_image
            n = len(names)
            eng._check(eng.lib.bf_genome_begin_image(eng.h, n, (C.c_char_p * n)(*names), (C.c_uint64 * n)(*[x.size for x in seqs])))
            eng._check(eng.lib.bf_genome_write(eng.h, 0, 0, seqs[0].ctypes.data, 10))
    finally:
        eng.close()


def test_genome_fill_limits():
    eng = engine.Engine(20, 5000, 2)
    try:
        seq = np.frombuffer(b"ACGT" * (3 << 20), np.uint8)               # 12 MB
        with pytest.raises(engine.BreakfastError) as e:                   # a piece above BF_GENOME_FILL_MAX
            eng.set_genome_in_pieces([b"c"], [seq], piece=9 << 20, fill=True)
        assert "BF_GENOME_FILL_MAX" in str(e.value)
        eng.set_genome_in_pieces([b"c"], [seq], piece=8 << 20, fill=True)   # exactly the limit is fine
    finally:
        eng.close()
