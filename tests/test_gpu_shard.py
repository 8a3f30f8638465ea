"""The sharded path inside the library (bf_shard_*, breakfast_amd/csrc/shard.hip): no Python between the stages.

  - RCCL transport with one rank (one GPU cannot host two RCCL ranks): the collectives really go through librccl.
  - host-staged transport with 2, 3 and 5 ranks sharing the test box's one GPU, as processes and as threads of one
    process; uneven and empty shards; a staging slot smaller than the exchange (several rounds).
Every case: the merged .sv (bf_shard_format_sv) equals the oracle's, byte for byte.
"""
import os
import pickle
import threading
import uuid

import numpy as np
import pytest
import torch.multiprocessing as mp

import oracle_lib as ol

from breakfast_amd import engine, synth

pytestmark = pytest.mark.gpu


def make_case(n_reads=30000, n_frac=0.0):
    contigs = [("chr1", 300000), ("chr10", 200000), ("chr2", 250000)]
    g = synth.make_genome(contigs, seed=61, n_runs=[(0, 5000, 2000)], lower_runs=[(2, 100000, 30000)])
    spec = synth.ReadSpec(n_reads=n_reads, read_len=120, anchor_len=20, n_events=300, singleton_frac=0.3, mate_frac=0.4,
                          seed=63, margin=300, n_rate=n_frac)
    return g, synth.make_reads(g, spec), 20, 5000, 2


def oracle_sv(g, b, A, F, E):
    return ol.detect_batch(ol.OracleGenome(g.names, g.seqs), g.names, b, A, F, E).sv


def run_rank(rank, world, cuts, name, slot_bytes, transport, uid, steps, out_dir, flags=0):
    g, b, A, F, E = make_case()
    lo, hi = cuts[rank], cuts[rank + 1]
    eng = engine.Engine(A, F, E, device=0, piece_reads=4000, flags=flags)
    try:
        eng.set_genome(g.names, g.seqs)
        if transport == "rccl":
            eng.shard_init_rccl(rank, world, lo, uid)
        else:
            eng.shard_init_host(rank, world, lo, name, slot_bytes)
        if hi > lo:
            eng.submit(b.slice(lo, hi))
        for _ in range(steps):                                   # idempotent on the resident reads
            res = eng.shard_finish(places=True)
        assert res.n_local == hi - lo
        pickle.dump(res, open(os.path.join(out_dir, "rank%d.pkl" % rank), "wb"))
    finally:
        eng.close()


def _proc(rank, world, cuts, name, slot_bytes, out_dir, flags=0):
    run_rank(rank, world, cuts, name, slot_bytes, "host", None, 2, out_dir, flags)


def check_places(results):
    """What crosses PCIe by itself (a source rank per read, block offsets per round and owner) against the tables that
    stay on the device (bf_shard_result_places): the records an owner asked of a source in a round are consecutive slots
    in result order, and the block offsets are those slots' record offsets."""
    W = len(results)
    for o, own in enumerate(results):
        assert np.array_equal(own.places["src_rank"], own.src_rank)
        gb = own.group_read_begin
        assert gb[0] == 0 and gb[-1] == own.src_rank.shape[0] and np.all(np.diff(gb.astype(np.int64)) >= 0)
        for g in range(gb.shape[0] - 1):
            for s, src in enumerate(results):
                lo, hi = int(src.block_off[g * W + o]), int(src.block_off[g * W + o + 1])
                mine = np.flatnonzero(own.src_rank[int(gb[g]):int(gb[g + 1])] == s) + int(gb[g])
                slots = own.places["slot"][mine].astype(np.int64)
                if mine.size == 0:
                    assert lo == hi
                    continue
                assert np.array_equal(slots, slots[0] + np.arange(slots.size))
                assert int(src.record_off[slots[0]]) == lo and int(src.record_off[slots[-1] + 1]) == hi
    for src in results:
        assert int(src.block_off[-1]) == src.record_bytes.size and np.all(np.diff(src.block_off.astype(np.int64)) >= 0)
        assert src.record_off.shape[0] == (src.n_records + 1 if src.n_records else 0)


def merged_sv(out_dir, world):
    g, b, A, F, E = make_case()
    results = [pickle.load(open(os.path.join(out_dir, "rank%d.pkl" % r), "rb")) for r in range(world)]
    assert sum(x.n_local for x in results) == sum(x.n_evidence for x in results) == b.n
    check_places(results)
    eng = engine.Engine(A, F, E, device=0)
    try:
        eng.set_genome(g.names, g.seqs)
        return eng.shard_format_sv(results), results
    finally:
        eng.close()


@pytest.mark.parametrize("flags", [0, engine.BF_FLAG_GROUPS_ALWAYS])
def test_one_rank_over_rccl(tmp_path, flags):
    g, b, A, F, E = make_case()
    uid = engine.Engine.shard_unique_id()
    assert len(uid) == 128
    run_rank(0, 1, [0, b.n], None, 0, "rccl", uid, 3, str(tmp_path), flags)
    sv, results = merged_sv(str(tmp_path), 1)
    assert sv == oracle_sv(g, b, A, F, E)
    # with one rank the places are simply the result order (group after group when the cluster stage runs in groups)
    assert np.array_equal(results[0].places["slot"], np.arange(results[0].places.shape[0], dtype=np.uint32))


@pytest.mark.parametrize("world,cuts_frac,slot_bytes", [
    (2, (0, 0.33, 1.0), 64 << 20),
    (3, (0, 0.25, 0.25, 1.0), 64 << 20),                 # an empty shard in the middle
    (5, (0, 0.1, 0.5, 0.5, 0.9, 1.0), 1 << 16),          # staging slots far smaller than the exchange: many rounds
])
@pytest.mark.parametrize("flags", [0, engine.BF_FLAG_GROUPS_ALWAYS])
def test_ranks_as_processes_over_the_host_transport(tmp_path, world, cuts_frac, slot_bytes, flags):
    """flags = BF_FLAG_GROUPS_ALWAYS: the owners' cluster stage a few contig ranks at a time, the sources rendering and
    copying a group's records while the next group is clustered (what large inputs do by themselves)."""
    g, b, A, F, E = make_case()
    cuts = [int(b.n * f) for f in cuts_frac]
    name = "/bf_test_%s" % uuid.uuid4().hex[:12]
    mp.spawn(_proc, args=(world, cuts, name, slot_bytes, str(tmp_path), flags), nprocs=world, join=True)
    sv, results = merged_sv(str(tmp_path), world)
    assert sv == oracle_sv(g, b, A, F, E)
    assert sum(x.clusters.shape[0] for x in results) == sv.count(b"\n") - 1
    assert sum(1 for x in results if x.clusters.shape[0]) >= 2            # the classes really spread over the owners


def test_ranks_as_threads_of_one_process(tmp_path):
    """What `breakfast-hip detect --gpus N` does: one thread and one ctx per rank in one process."""
    g, b, A, F, E = make_case()
    world = 3
    cuts = [0, b.n // 2, b.n // 2 + 1, b.n]
    name = "/bf_test_%s" % uuid.uuid4().hex[:12]
    errors = []

    def work(rank):
        try:
            run_rank(rank, world, cuts, name, 1 << 20, "host", None, 1, str(tmp_path))
        except Exception as e:                                   # noqa: BLE001
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=240)
    assert not errors, errors
    sv, _ = merged_sv(str(tmp_path), world)
    assert sv == oracle_sv(g, b, A, F, E)


def test_descending_input_ranges_are_refused():
    g, b, A, F, E = make_case(n_reads=2000)
    name = "/bf_test_%s" % uuid.uuid4().hex[:12]
    out = {}

    def work(rank):
        eng = engine.Engine(A, F, E, device=0)
        try:
            eng.set_genome(g.names, g.seqs)
            eng.shard_init_host(rank, 2, 1000 if rank == 0 else 0, name, 1 << 16)
            out[rank] = "joined"
        except engine.BreakfastError as e:
            out[rank] = str(e)
        finally:
            eng.close()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert all("ascending input ranges" in out[r] for r in range(2)), out


@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_command_line_sharded_over_one_gpu(tmp_path, devices):
    """`breakfast-hip detect --devices=0,0`: the C++ driver's sharded mode (threads, the library's sharded path, the
    host-staged transport because the ranks share the GPU), from files on disk to the .sv bytes."""
    import subprocess
    import detect_oracle as po
    g, b, A, F, E = make_case(n_reads=12000)
    prefix = str(tmp_path / "g")
    open(prefix + ".fa", "wb").write(g.fasta())
    hits = str(tmp_path / "hits.txt")
    text = synth.bowtie_lines(b, g.names)
    open(hits, "wb").write(text)
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "breakfast_amd", "csrc", "breakfast-hip")
    p = subprocess.run([exe, "detect", "--anchor-len=%d" % A, "--devices=" + devices, "--block-bytes=200000",
                        "--anchor-hits=" + hits, "unused.bam", prefix], capture_output=True, timeout=200)
    assert p.returncode == 0, p.stderr.decode()
    assert p.stdout == po.detect(g.fasta(), text.split(b"\n")[:-1], A, F, E)
    assert b"Found 12000 rearrangement supporting reads." in p.stderr


def run_as_threads(g, parts, bases, A, F=5000, E=2, flags=0, slot_bytes=8 << 20, name=None, delay=None, timeout=280):
    """One thread and one ctx per rank on the test box's one GPU (the host transport), rank r holding parts[r] (a Batch, a
    PackedBatch or None) whose first read is global read bases[r] -> the ranks' ShardResults."""
    world = len(parts)
    name = name or "/bf_test_%s" % uuid.uuid4().hex[:12]
    results, errors = [None] * world, []

    def work(rank):
        try:
            if delay:
                import time
                time.sleep(delay[rank])
            eng = engine.Engine(A, F, E, device=0, flags=flags)
            try:
                eng.set_genome(g.names, g.seqs)
                eng.shard_init_host(rank, world, int(bases[rank]), name, slot_bytes)
                if parts[rank] is not None and parts[rank].n:
                    eng.submit(parts[rank])
                results[rank] = eng.shard_finish()
            finally:
                eng.close()
        except Exception as e:                                   # noqa: BLE001
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=timeout)
    assert not errors, errors
    assert all(x is not None for x in results), "a rank did not finish"
    return results


def merged(g, results, A, F=5000, E=2):
    """the ranks' results as the reference's stdout: with a ctx, and by the ctx-free merge (what a process without a GPU
    runs) -- the same bytes"""
    eng = engine.Engine(A, F, E, device=0)
    try:
        eng.set_genome(g.names, g.seqs)
        sv = eng.shard_format_sv(results)
    finally:
        eng.close()
    assert engine.shard_format_sv(g.names, results) == sv
    return sv


@pytest.mark.parametrize("world", [2, 3])
def test_config2_scaled_over_ranks(world):
    """BASELINE configs[2] (100M reads as 16 blocks, sharded over 2/4/8 GPUs in contiguous input ranges), scaled to 400k reads
    on 1.5 % of the genome: rank r holds the blocks bench.py gives it, through bf_shard_finish; merged table == oracle."""
    contigs, gk, spec = synth.config(2, scale=0.004, genome_scale=0.015)
    spec.margin = 10000 + 2 * spec.read_len + 128
    g = synth.make_genome(contigs, **gk)
    blocks = [synth.make_reads(g, synth.block_spec(spec, k)) for k in range(synth.BLOCKS)]
    import test_gpu_submit as tgs
    whole = tgs.concat(blocks)
    ref = ol.detect_batch(ol.OracleGenome(g.names, g.seqs), g.names, whole, spec.anchor_len, 5000, 2)
    assert ref.n_clusters > 2000
    parts = [tgs.concat([blocks[k] for k in synth.rank_blocks(r, world)]) for r in range(world)]
    bases = [synth.block_first_read(spec, synth.rank_blocks(r, world)[0]) for r in range(world)]
    results = run_as_threads(g, parts, bases, spec.anchor_len)
    assert sum(x.n_local for x in results) == sum(x.n_evidence for x in results) == whole.n
    assert merged(g, results, spec.anchor_len) == ref.sv


@pytest.mark.parametrize("pile_on_last_contig", [False, True])
@pytest.mark.parametrize("flags", [0, engine.BF_FLAG_GROUPS_ALWAYS])
def test_more_printed_reads_than_evidence_over_ranks(pile_on_last_contig, flags):
    """A read reported by an earlier seed joins a later seed's cluster as well (detect.rs:215-226): an owner prints more
    reads than it holds evidence, in one go and in contig groups (whose per-read arrays were sized for the evidence)."""
    import test_gpu_parity as tgp
    g, b, A, F = tgp.case_reads_printed_twice(pile_on_last_contig)
    ref = ol.detect_batch(ol.OracleGenome(g.names, g.seqs), g.names, b, A, F, 2)
    cut = b.n // 3
    results = run_as_threads(g, [b.slice(0, cut), b.slice(cut, b.n)], [0, cut], A, F=F, flags=flags)
    assert sum(x.src_rank.size for x in results) > b.n
    assert merged(g, results, A, F=F) == ref.sv


@pytest.mark.parametrize("seed", range(200, 224))
def test_small_random_cases_over_ranks(seed, tmp_path):
    """Small random cases (60-220 reads on three contigs: owners with a handful of records, with none, with records on one
    contig only) over two and three ranks, in one go and in contig groups, against the oracle's table."""
    import detect_oracle as po
    import test_hand_tables as tht
    fa, bt, A, F, E = tht.random_case(seed, tmp_path)
    fa_bytes, lines = open(fa, "rb").read(), open(bt, "rb").read()
    want = ol.detect_text(ol.OracleGenome.from_fasta(fa_bytes), lines, A, F, E).sv
    gd = po.parse_fasta(fa_bytes)
    names = list(gd.keys())
    g = synth.Genome(names, [np.frombuffer(gd[n], np.uint8) for n in names])
    b = engine.parse_bowtie_lines(lines.split(b"\n")[:-1], {n: i for i, n in enumerate(names)})
    world = 2 + seed % 2
    cuts = [b.n * r // world for r in range(world + 1)]
    if seed % 5 == 0:
        cuts[1] = cuts[0]                                   # a rank without reads
    parts = [b.slice(cuts[r], cuts[r + 1]) if cuts[r + 1] > cuts[r] else None for r in range(world)]
    for flags in (0, engine.BF_FLAG_GROUPS_ALWAYS):
        results = run_as_threads(g, parts, cuts[:-1], A, F=F, E=E, flags=flags, timeout=100)
        assert merged(g, results, A, F=F, E=E) == want, flags


def odd_case(seed):
    """Genomes of one, two or seven contigs (names that sort unlike their order), windows that reach over contig ends,
    max_frag_len from 0 to 2^31 - 1, min_evidence 1 to 4, everything on one contig or spread -> (genome, batch, A, F, E)."""
    rng = np.random.default_rng(seed)
    n_contigs = int(rng.choice([1, 2, 7]))
    names = ["chr%d" % k for k in rng.permutation(20)[:n_contigs]]
    contigs = [(nm, int(rng.integers(3000, 30000))) for nm in names]
    g = synth.make_genome(contigs, seed=3000 + seed, end_n=int(rng.integers(0, 30)))
    A = int(rng.choice([12, 20]))
    L = int(rng.choice([2 * A + 4, 60, 100]))
    spec = synth.ReadSpec(n_reads=int(rng.integers(20, 300)), read_len=L, anchor_len=A, n_events=int(rng.integers(1, 10)),
                          singleton_frac=float(rng.choice([0.0, 0.3])), mate_frac=0.5, hot_frac=float(rng.choice([0.0, 0.5])), hot_events=1,
                          subst_rate=0.02, n_rate=0.003, seed=4000 + seed, margin=0)
    b = synth.make_reads(g, spec)
    return g, b, A, int(rng.choice([0, 1, 50, 5000, 2 ** 31 - 1])), int(rng.integers(1, 5))


@pytest.mark.parametrize("seed", range(40))
def test_odd_small_cases_on_one_gpu_and_over_ranks(seed):
    g, b, A, F, E = odd_case(seed)
    want = ol.detect_batch(ol.OracleGenome(g.names, g.seqs), g.names, b, A, F, E).sv
    for flags in (0, engine.BF_FLAG_GROUPS_ALWAYS, engine.BF_FLAG_FORCE_EXACT | engine.BF_FLAG_GROUPS_ALWAYS):
        eng = engine.Engine(A, F, E, flags=flags)
        try:
            eng.set_genome(g.names, g.seqs)
            eng.submit(b)
            assert eng.format_sv(eng.finish(reads=False)) == want, ("one GPU", flags)
        finally:
            eng.close()
    world = 2 + seed % 2
    cuts = [b.n * r // world for r in range(world + 1)]
    parts = [b.slice(cuts[r], cuts[r + 1]) if cuts[r + 1] > cuts[r] else None for r in range(world)]
    for flags in (0, engine.BF_FLAG_GROUPS_ALWAYS):
        results = run_as_threads(g, parts, cuts[:-1], A, F=F, E=E, flags=flags, timeout=100)
        assert merged(g, results, A, F=F, E=E) == want, ("ranks", flags)


def test_eight_ranks_uneven_empty_shards_and_a_hot_class():
    """World size 8 (what configs[2] shards over): owner = mix64(class) % 8, arrival order = input order, merged print
    order; uneven shards, two of them empty, one holding a single read; a third of the reads on two hot junctions, so
    two classes own most of the evidence and ownership is skewed."""
    contigs = [("chr1", 90000), ("chr10", 70000), ("chr2", 80000), ("chrX", 60000)]
    g = synth.make_genome(contigs, seed=51)
    spec = synth.ReadSpec(n_reads=1600, read_len=80, anchor_len=20, n_events=40, singleton_frac=0.2, mate_frac=0.6,
                          hot_frac=0.35, hot_events=2, seed=53, margin=300)
    b = synth.make_reads(g, spec)
    n = b.n
    ref = ol.detect_batch(ol.OracleGenome(g.names, g.seqs), g.names, b, 20, 5000, 2)
    assert ref.n_clusters > 10
    cuts = [0, n // 3, n // 3 + 1, n // 3 + 1, n // 2, (3 * n) // 4, (3 * n) // 4, (7 * n) // 8, n]
    parts = [b.slice(cuts[r], cuts[r + 1]) if cuts[r + 1] > cuts[r] else None for r in range(8)]
    for flags in (0, engine.BF_FLAG_GROUPS_ALWAYS):
        results = run_as_threads(g, parts, cuts[:8], 20, flags=flags, slot_bytes=1 << 16)
        owned = [x.n_evidence for x in results]
        assert sum(owned) == n and max(owned) > 3 * (n // 16), owned    # the hot classes make ownership uneven
        assert [x.n_local for x in results] == [cuts[r + 1] - cuts[r] for r in range(8)]
        assert merged(g, results, 20) == ref.sv


def test_the_pack_that_sorts_and_the_pack_that_scatters_agree(monkeypatch):
    """run_partition packs by count -> scan -> scatter; the round-2 form (sort (owner, index) pairs, gather the records) stays
    as the fallback for more than 64 ranks or records that do not fit LDS.  Both through five ranks with an empty shard and
    a hot class, reads with N (signatures that travel with their raw bytes): the merged table equals the oracle's."""
    contigs = [("chr1", 90000), ("chr10", 70000), ("chr2", 80000), ("chrX", 60000)]
    g = synth.make_genome(contigs, seed=61, n_runs=[(0, 30000, 400), (2, 40000, 700), (3, 20000, 300)])
    spec = synth.ReadSpec(n_reads=3000, read_len=100, anchor_len=20, n_events=60, singleton_frac=0.2, mate_frac=0.6,
                          hot_frac=0.3, hot_events=2, seed=67, margin=300)
    b = synth.make_reads(g, spec)
    n = b.n
    ref = ol.detect_batch(ol.OracleGenome(g.names, g.seqs), g.names, b, 20, 5000, 2)
    assert ref.n_clusters > 10
    cuts = [0, n // 4, n // 4, n // 2, n - 1, n]
    parts = [b.slice(cuts[r], cuts[r + 1]) if cuts[r + 1] > cuts[r] else None for r in range(5)]
    for sorted_pack in (False, True):
        if sorted_pack:
            monkeypatch.setenv("BREAKFAST_HIP_SORTED_PACK", "1")
        else:
            monkeypatch.delenv("BREAKFAST_HIP_SORTED_PACK", raising=False)
        results = run_as_threads(g, parts, cuts[:5], 20, slot_bytes=1 << 16)
        assert sum(x.n_evidence for x in results) == n
        assert merged(g, results, 20) == ref.sv


def test_two_ranks_at_half_a_million_reads():
    contigs, gk, spec = synth.config(1, scale=0.05, genome_scale=0.05)      # 500k reads, 154 Mb genome, 25 contigs
    spec.margin = 10000 + 2 * spec.read_len + 128
    g = synth.make_genome(contigs, **gk)
    b = synth.make_reads(g, spec)
    ref = ol.detect_batch(ol.OracleGenome(g.names, g.seqs), g.names, b, spec.anchor_len, 5000, 2)
    assert ref.n_clusters > 3000
    cuts = [0, (2 * b.n) // 5, b.n]
    results = run_as_threads(g, [b.slice(cuts[0], cuts[1]), b.slice(cuts[1], cuts[2])], cuts[:2], spec.anchor_len)
    assert merged(g, results, spec.anchor_len) == ref.sv


def test_scaffold_rich_genome_in_groups():
    """More contigs than one launch of the owners' rank histogram counts (4096 bins of LDS; the key has room for 16384
    contigs): the contig groups of the sharded path must still cut the imported records by left contig rank."""
    n_contigs = 5000
    contigs = [("scaf%05d" % i, 2500) for i in range(n_contigs)]
    g = synth.make_genome(contigs, seed=91)
    spec = synth.ReadSpec(n_reads=20000, read_len=100, anchor_len=20, n_events=600, singleton_frac=0.3, mate_frac=0.3,
                          deletion_frac=0.2, seed=93, margin=120)
    b = synth.make_reads(g, spec)
    ref = ol.detect_batch(ol.OracleGenome(g.names, g.seqs), g.names, b, 20, 5000, 2)
    assert ref.n_clusters > 300
    left = np.minimum(b.hits["contig5"], b.hits["contig3"])                  # (names sort like their numbers here)
    assert int(left.max()) > 4096                                            # reads whose rank lies in the second window
    cuts = [0, b.n // 2, b.n]
    results = run_as_threads(g, [b.slice(0, cuts[1]), b.slice(cuts[1], b.n)], cuts[:2], 20, flags=engine.BF_FLAG_GROUPS_ALWAYS)
    assert all(len(x.group_read_begin) == 3 for x in results)               # two groups each
    assert merged(g, results, 20) == ref.sv
    # and the one-GPU path on the same genome (its histogram comes from bf_submit's staging threads)
    eng = engine.Engine(20, 5000, 2, flags=engine.BF_FLAG_GROUPS_ALWAYS)
    try:
        eng.set_genome(g.names, g.seqs)
        eng.submit(b)
        assert eng.format_sv(eng.finish()) == ref.sv
    finally:
        eng.close()


def test_a_stale_segment_of_a_crashed_run_is_not_joined():
    """A segment of the same name and shape with `ready` set, left by a run that died: a rank that opens the name before
    rank 0 has replaced it must not settle there (it says hello and nobody answers), but find the new segment."""
    import struct
    g, b, A, F, E = make_case(n_reads=3000)
    ref = oracle_sv(g, b, A, F, E)
    name = "/bf_test_%s" % uuid.uuid4().hex[:12]
    slot = 1 << 16
    size = 4096 + 2 * 4096 * 8 + 2 * slot
    with open("/dev/shm" + name, "wb") as f:
        f.truncate(size)
        f.seek(0)
        f.write(struct.pack("<IIIIQQ", 0xbf5a4d31, 0, 0, 2, slot, 4096))     # ShmHeader: ready, arrived, sense, n_ranks, slot_bytes, ctl_words
    try:
        cuts = [0, b.n // 2, b.n]
        # rank 1 arrives first and finds the stale segment; rank 0 replaces it half a second later
        results = run_as_threads(g, [b.slice(0, cuts[1]), b.slice(cuts[1], b.n)], cuts[:2], A, slot_bytes=slot, name=name, delay=[0.6, 0.0])
        assert merged(g, results, A) == ref
    finally:
        try:
            os.unlink("/dev/shm" + name)
        except FileNotFoundError:
            pass


def test_two_ranks_with_owners_large_enough_for_two_groups(tmp_path):
    """From 4M records per owner on the owners cluster in two contig groups by themselves (the sources render and copy a
    group's records while the next one is clustered).  9M reads over two ranks (threads, host transport), the merged .sv
    against the single-GPU path's on the same reads -- which tests/test_gpu_parity.py holds against the oracle at this size."""
    import hashlib
    contigs, gk, spec = synth.config(1, scale=0.9, genome_scale=0.02)
    spec.margin = 10000 + 2 * spec.read_len + 128
    g = synth.make_genome(contigs, **gk)
    b = engine.pack_batch(synth.make_reads(g, spec))
    A = spec.anchor_len
    single = engine.Engine(A, 5000, 2)
    try:
        single.set_genome(g.names, g.seqs)
        single.submit(b)
        want = hashlib.sha256(single.format_sv(single.finish(copy=False))).hexdigest()
    finally:
        single.close()
    world, cuts = 2, [0, b.n * 2 // 5, b.n]
    name = "/bf_test_%s" % uuid.uuid4().hex[:12]
    results, errors = [None] * world, []

    def part(lo, hi):
        so, fo = b.seq_off, b.frag_off
        return engine.PackedBatch(b.seq_nibbles[int(so[lo]):int(so[hi])], so[lo:hi + 1] - so[lo], b.seq_len[lo:hi],
                                  b.frag_bytes[int(fo[lo]):int(fo[hi])], fo[lo:hi + 1] - fo[lo], b.hits[lo:hi])

    def work(rank):
        try:
            eng = engine.Engine(A, 5000, 2, device=0)
            try:
                eng.set_genome(g.names, g.seqs)
                eng.shard_init_host(rank, world, cuts[rank], name, 64 << 20)
                eng.submit(part(cuts[rank], cuts[rank + 1]))
                res = eng.shard_finish()
                assert len(res.group_read_begin) == 3, "expected two groups, got %d" % (len(res.group_read_begin) - 1)
                results[rank] = res
            finally:
                eng.close()
        except Exception as e:                                   # noqa: BLE001
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=280)
    assert not errors, errors
    assert sum(x.n_evidence for x in results) == b.n and min(x.n_evidence for x in results) >= 1 << 22
    eng = engine.Engine(A, 5000, 2, device=0)
    try:
        eng.set_genome(g.names, g.seqs)
        got = hashlib.sha256(eng.shard_format_sv(results)).hexdigest()
    finally:
        eng.close()
    assert got == want
