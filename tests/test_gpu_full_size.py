"""BASELINE.json's configs at FULL size (round 3): what can break here and not at a few per cent of the size is index
width (2^32 boundaries in offsets x strides), eight contig groups, chunks set aside by the thousand, clusters far above
256 members.

The oracle finishes 0.4M reads a second, so 50M and 100M reads are checked through a cut that is EXACT, not a sample:
breakpoint classes never interact, a read's class holds its left contig after the reorient step
(detect.rs:108-113: the smaller of the two anchors' contig names), and the reference prints contig-major
(detect.rs:193-198).  So the reference run on only the reads whose left contig is chr21 or chr22 (about 3 % of them, in
their input order) prints exactly the chr21 and chr22 rows of the full run: those rows of the GPU's table must equal
the oracle's bytes.  On top of that, for the whole table: n_evidence, a second bf_finish on the resident reads returns
the same result arrays, BF_FLAG_READS_ON_HOST gives the same .sv (SHA-256), and for configs[2] two ranks through
bf_shard_finish (host transport, threads sharing the GPU) give it too.  configs[4] and configs[0] take the whole table.

BREAKFAST_FULL_SCALE (default 1.0) shrinks reads and genome for a quick rehearsal.
"""
import hashlib
import os
import threading
import time
import uuid
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import oracle_lib as ol

from breakfast_amd import engine, synth

pytestmark = pytest.mark.gpu

SCALE = float(os.environ.get("BREAKFAST_FULL_SCALE", "1.0"))
SV_HEADER = (b"CHROM\tSTRAND\tPOSITION\tNEARBY FEATURES\tCHROM\tSTRAND\tPOSITION\tNEARBY FEATURES\t"
             b"SUPPORTING READS\tSIGNATURE\tNOTES\n")


def log(*a):
    print("[full-size %s]" % time.strftime("%H:%M:%S"), *a, flush=True)


def sha(a) -> str:
    return hashlib.sha256(memoryview(np.ascontiguousarray(a))).hexdigest()


def left_contig(hits: np.ndarray, names) -> np.ndarray:
    """contig id of `chr` after detect.rs:108-113: the anchor contig whose NAME is smaller (byte-wise)."""
    order = sorted(range(len(names)), key=lambda i: names[i])
    rank = np.empty(len(names), np.int64)
    rank[order] = np.arange(len(names))
    c5, c3 = hits["contig5"].astype(np.int64), hits["contig3"].astype(np.int64)
    return np.where(rank[c5] <= rank[c3], c5, c3)


def take_reads(b: engine.Batch, keep: np.ndarray) -> engine.Batch:
    """the reads `keep` (ascending indices) of a batch whose reads and fragment ids have one length each"""
    n = b.n
    L, F = int(b.seq_off[1]), int(b.frag_off[1])
    m = keep.shape[0]
    return engine.Batch(b.seq_bytes.reshape(n, L)[keep].reshape(-1), np.arange(m + 1, dtype=np.uint64) * np.uint64(L),
                        b.frag_bytes.reshape(n, F)[keep].reshape(-1), np.arange(m + 1, dtype=np.uint64) * np.uint64(F), b.hits[keep])


def concat(parts):
    so, fo, s0, f0 = [np.zeros(1, np.uint64)], [np.zeros(1, np.uint64)], 0, 0
    for p in parts:
        so.append(p.seq_off[1:] + np.uint64(s0)); fo.append(p.frag_off[1:] + np.uint64(f0))
        s0 += int(p.seq_off[-1]); f0 += int(p.frag_off[-1])
    return engine.Batch(np.concatenate([p.seq_bytes for p in parts]), np.concatenate(so), np.concatenate([p.frag_bytes for p in parts]),
                        np.concatenate(fo), np.concatenate([p.hits for p in parts]))


def rows_of_contigs(eng, res, ids, names) -> bytes:
    """header + the rows of `res` whose left contig is one of `ids`, in print order (a contig's rows are contiguous)."""
    out = [SV_HEADER]
    contig = res.clusters["contig"]
    for c in sorted(ids, key=lambda i: names[i]):
        at = np.flatnonzero(contig == c)
        if at.size == 0:
            continue
        a, e = int(at[0]), int(at[-1]) + 1
        assert e - a == at.size, "rows of contig %r are not contiguous in print order" % names[c]
        part = engine.Result(res.n_evidence, res.clusters[a:e], np.zeros(0, engine.READ_DTYPE), res.junction_bytes,
                             n_reads=res.n_reads, cluster_junction_off=res.cluster_junction_off[a:e + 1])
        out.append(eng.format_sv(part)[len(SV_HEADER):])
    return b"".join(out)


def make_blocks(cfg):
    """-> genome, spec, list of ASCII blocks in input order (configs[2]: synth.BLOCKS blocks; others: one)."""
    contigs, gk, spec = synth.config(cfg, scale=SCALE, genome_scale=SCALE)
    if cfg and SCALE >= 0.05:
        spec.margin = 10000 + 2 * spec.read_len + 128             # as bench.py: junctions clear of the N runs at contig ends
    t = time.time()
    g = synth.make_genome(contigs, **gk)
    log("configs[%d]: genome %.2f Gb in %.0fs" % (cfg, sum(s.size for s in g.seqs) / 1e9, time.time() - t))
    t = time.time()
    if cfg == 2:
        with ThreadPoolExecutor(3) as ex:
            blocks = list(ex.map(lambda k: synth.make_reads(g, synth.block_spec(spec, k)), range(synth.BLOCKS)))
    else:
        blocks = [synth.make_reads(g, spec)]
    if cfg == 0:                                                   # the generator copies the genome's case (a soft-masked run); a
        blocks[0].seq_bytes &= np.uint8(0xDF)                      # sequencer's reads are upper case (the ASCII form: see the test)
    log("%d reads x %d bp in %d block(s), %.0fs" % (sum(b.n for b in blocks), spec.read_len, len(blocks), time.time() - t))
    return g, spec, blocks


def check_full(cfg, whole=False, shard=False, args=None):
    g, spec, blocks = make_blocks(cfg)
    A, F, E = args or (spec.anchor_len, 5000, 2)
    names = g.names
    n_total = sum(b.n for b in blocks)
    # the oracle's part: everything, or the reads whose left contig is chr21 / chr22
    if whole:
        cut_ids = list(range(len(names)))
        sub = concat(blocks) if len(blocks) > 1 else blocks[0]
    else:
        cut_ids = [i for i, nm in enumerate(names) if nm in (b"chr21", b"chr22")]
        assert len(cut_ids) == 2
        parts = []
        for b in blocks:
            lc = left_contig(b.hits, names)
            parts.append(take_reads(b, np.flatnonzero(np.isin(lc, cut_ids))))
        sub = concat(parts)
    t = time.time()
    ref = ol.detect_batch(ol.OracleGenome(g.names, g.seqs), names, sub, A, F, E)
    log("oracle: %d reads -> %d clusters, %d supporting reads, %.0fs" % (sub.n, ref.n_clusters, ref.n_support, time.time() - t))
    assert ref.n_clusters > (0 if cfg == 0 else 100 * min(SCALE, 1.0))
    t = time.time()
    packed = [engine.pack_batch(b) for b in blocks]
    del blocks, sub
    log("packed to 4-bit codes in %.0fs" % (time.time() - t))

    # ---- one GPU, bf_finish -------------------------------------------------------------------------------------
    eng = engine.Engine(A, F, E)
    try:
        eng.set_genome(g.names, g.seqs)
        t = time.time()
        for p in packed:
            eng.submit(p)
        res = eng.finish(copy=False)
        log("GPU: %d clusters, %d supporting reads, %.1f MB of records, first pass %.2fs" %
            (res.clusters.shape[0], res.n_reads, res.junction_bytes.size / 1e6, time.time() - t))
        assert res.n_evidence == n_total
        got = rows_of_contigs(eng, res, cut_ids, names)
        assert len(got) == len(ref.sv) and got == ref.sv, "the rows of the cut differ from the oracle's"
        del got
        sv = eng.format_sv(res, as_array=True)
        want = sha(sv)
        sv_len = sv.size
        if whole:
            assert sv_len == len(ref.sv)
        del sv
        keep = (res.clusters.copy(), res.junction_bytes.copy(), res.cluster_junction_off.copy())
        res2 = eng.finish(copy=False)                                 # the whole path again on the resident reads
        assert np.array_equal(res2.clusters, keep[0]) and np.array_equal(res2.cluster_junction_off, keep[2])
        assert np.array_equal(res2.junction_bytes, keep[1]), "a second bf_finish returned other records"
        del keep, res, res2
    finally:
        eng.close()
    log("cut identical to the oracle, second pass identical; .sv %d bytes, sha256 %s" % (sv_len, want[:16]))

    # ---- the caller keeps its batches (BF_FLAG_READS_ON_HOST): same table ---------------------------------------------
    eng = engine.Engine(A, F, E, flags=engine.BF_FLAG_READS_ON_HOST)
    try:
        eng.set_genome(g.names, g.seqs)
        for p in packed:
            eng.submit(p)
        res = eng.finish(copy=False)
        assert res.n_evidence == n_total
        assert sha(eng.format_sv(res, as_array=True)) == want, "BF_FLAG_READS_ON_HOST: another table"
        del res
    finally:
        eng.close()
    log("reads-on-host table identical")

    # ---- two ranks through bf_shard_finish (host transport, threads sharing the GPU) ----------------------------------
    if shard:
        world = 2
        name = "/bf_full_%s" % uuid.uuid4().hex[:12]
        per = [list(synth.rank_blocks(r, world, len(packed))) for r in range(world)]
        firsts = np.cumsum([0] + [p.n for p in packed])
        results, errors = [None] * world, []

        def work(rank):
            try:
                e2 = engine.Engine(A, F, E, device=0)
                try:
                    e2.set_genome(g.names, g.seqs)
                    e2.shard_init_host(rank, world, int(firsts[per[rank][0]]), name, 256 << 20)
                    for k in per[rank]:
                        e2.submit(packed[k])
                    results[rank] = e2.shard_finish()
                finally:
                    e2.close()
            except Exception as e:                                   # noqa: BLE001
                errors.append((rank, repr(e)))

        t = time.time()
        threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        for th in threads:
            th.start()
        for th in threads:
            th.join(timeout=600)
        assert not errors, errors
        assert sum(x.n_evidence for x in results) == sum(x.n_local for x in results) == n_total
        e3 = engine.Engine(A, F, E, device=0)
        try:
            e3.set_genome(g.names, g.seqs)
            got = sha(e3.shard_format_sv(results, as_array=True))
        finally:
            e3.close()
        assert got == want, "two ranks through bf_shard_finish: another table"
        log("two ranks through bf_shard_finish: table identical (%.0fs)" % (time.time() - t))


def test_config0_full_size_default_anchor_and_anchor_20():
    """configs[0]: 100k 2x100 bp reads on one chr22-sized contig with an N run and a soft-masked run, A = 30 (the
    reference's default) and A = 20, whole table."""
    check_full(0, whole=True)
    contigs, gk, spec = synth.config(0, scale=SCALE, genome_scale=SCALE)
    spec.anchor_len = 20
    g = synth.make_genome(contigs, **gk)
    b = synth.make_reads(g, spec)
    ref = ol.detect_batch(ol.OracleGenome(g.names, g.seqs), g.names, b, 20, 5000, 2)
    assert ref.n_clusters > 0
    for packed in (False, True):
        eng = engine.Engine(20, 5000, 2)
        try:
            eng.set_genome(g.names, g.seqs)
            if packed:
                b.seq_bytes = b.seq_bytes & np.uint8(0xDF)           # a sequencer's reads are upper case
                ref = ol.detect_batch(ol.OracleGenome(g.names, g.seqs), g.names, b, 20, 5000, 2)
            eng.submit(engine.pack_batch(b) if packed else b)
            assert eng.format_sv(eng.finish()) == ref.sv
        finally:
            eng.close()


def test_config4_full_size_whole_table():
    """configs[4]: 10M x 250 bp, A = 30, 5 % mismatches + 1 % N: the whole table against the oracle."""
    check_full(4, whole=True)


def test_config3_full_size_duplicate_heavy():
    """configs[3]: 50M reads, 60 % duplicates, 5000 junctions with thousands of reads each."""
    check_full(3)


def test_config2_full_size_one_gpu_and_two_ranks():
    """configs[2]: 100M reads as 16 blocks; one GPU (eight contig groups) and two ranks through bf_shard_finish."""
    check_full(2, shard=True)
