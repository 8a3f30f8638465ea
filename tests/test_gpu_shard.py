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
