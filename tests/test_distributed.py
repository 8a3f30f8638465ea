"""world_size-2 tests of the sharded path's orchestration on CPU (gloo).

The GPU engine cannot run here, so a stand-in stage engine built on the ORACLE plays the device
stages; what is under test is breakfast_amd.distributed: owner partitioning by class, count and
record all-to-alls, arrival order (stability), junction routing back to the source ranks and the
merge of per-owner cluster lists.  tests/test_gpu_distributed.py runs the same orchestration with
the real HIP engine.
"""
import os
import socket
import sys
import zlib

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import detect_oracle as po

from breakfast_amd import synth
from breakfast_amd.distributed import DistributedDetect, format_sv
from breakfast_amd.engine import CLUSTER_DTYPE, READ_DTYPE, Result, text_records


class OracleStage:
    """Stand-in for the device stages.  Every rank can see the whole input (the test generates it
    deterministically), so a wire record only has to carry the global read index."""

    def __init__(self, genome, contig_names, batch, lo, hi, A, F, E, bases_hint=None):
        self.g, self.names, self.b, self.lo, self.hi, self.A, self.F, self.E = genome, contig_names, batch, lo, hi, A, F, E
        self.ev_all = None

    def _evidence(self, i):
        h = self.b.hits[i]
        e = po.localise(self.g, self.A, self.b.frag(i), self.b.seq(i), bool(h["strands"] & 1), self.names[int(h["contig5"])],
                        int(h["pos5"]), bool(h["strands"] & 2), self.names[int(h["contig3"])], int(h["pos3"]))
        e.read_num = i
        return e

    def localise(self):
        self.local = [self._evidence(i) for i in range(self.lo, self.hi)]

    def exchange_local(self):
        return max((len(e.frag_id) for e in self.local), default=0), 0

    def exchange_pack_tensor(self, mf, sx, world):
        owner = np.array([zlib.crc32(repr((e.chr, e.strand, e.mchr, e.mstrand, e.signature)).encode()) % world
                          for e in self.local], np.int64)
        order = np.argsort(owner, kind="stable")
        counts = np.bincount(owner, minlength=world).astype(np.int64)
        gidx = (np.arange(self.lo, self.hi, dtype=np.int64))[order]
        return torch.from_numpy(gidx.view(np.uint8).copy()), counts, 8

    def exchange_import_tensor(self, recv, src_counts, bases):
        gidx = recv.numpy().view(np.int64)
        # arrival order must be ascending global index (stability argument of the sort)
        assert np.all(np.diff(gidx) > 0), "records did not arrive in input order"
        self.bases = bases.astype(np.int64)
        self.owned = [self._evidence(int(i)) for i in gidx]

    def clusterise(self):
        sv, clusters = po.cluster_and_format(self.owned, self.F, self.E)
        cid = {n: i for i, n in enumerate(self.names)}
        cl = np.zeros(len(clusters), CLUSTER_DTYPE)
        reads, sizes, src_rank, src_index = [], [], [], []
        for k, (seed, members) in enumerate(clusters):
            cl[k]["lpos"] = po._i64(seed.pos); cl[k]["rpos"] = po._i64(seed.mpos)
            cl[k]["seed_index"] = seed.read_num
            cl[k]["read_begin"] = len(reads); cl[k]["read_end"] = len(reads) + len(members)
            cl[k]["contig"] = cid[seed.chr]; cl[k]["mcontig"] = cid[seed.mchr]
            cl[k]["strand"] = seed.strand; cl[k]["mstrand"] = seed.mstrand
            for m in members:
                # the record size travels with the exchange record: a text record here (no packed renderer)
                reads.append((m.read_num, 0))
                sizes.append((8 + len(m.sequence) + 3) & ~3)
                r = int(np.searchsorted(self.bases, m.read_num, side="right") - 1)
                src_rank.append(r); src_index.append(m.read_num - int(self.bases[r]))
        rd = np.array(reads, READ_DTYPE) if reads else np.zeros(0, READ_DTYPE)
        return Result(len(self.owned), cl, rd, np.zeros(0, np.uint8), np.array(src_rank, np.uint32), np.array(src_index, np.uint32),
                      junction_sizes=np.array(sizes, np.uint32))

    def render_junctions(self, local_index):
        js = [self.local[int(i)].sequence for i in local_index]
        off = np.zeros(len(js) + 1, np.uint64)
        off[1:] = np.cumsum([len(j) for j in js])
        # junction strings travel as records (include/breakfast_hip.h); this stand-in has no packed renderer
        return text_records(np.frombuffer(b"".join(js), np.uint8).copy(), off, [self.local[int(i)].swapped for i in local_index])


def make_case(seed=0):
    contigs = [("chr1", 90000), ("chr10", 70000), ("chr2", 80000)]
    g = synth.make_genome(contigs, seed=41 + seed)
    spec = synth.ReadSpec(n_reads=900, read_len=80, anchor_len=20, n_events=15, singleton_frac=0.3, mate_frac=0.4,
                          seed=43 + seed, margin=300)
    return g, synth.make_reads(g, spec), 20, 5000, 2


def _worker(rank, world, port, out_path, uneven, chunks):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g, b, A, F, E = make_case()
        n = b.n
        cuts = [0, n // 5, n] if uneven else [0, n // 2, n]
        if world == 3:
            cuts = [0, n // 3, n // 3, n]           # an empty shard in the middle
        lo, hi = cuts[rank], cuts[rank + 1]
        stage = OracleStage(g.as_dict(), g.names, b, lo, hi, A, F, E)
        dd = DistributedDetect(stage, rank, world, lo, hi - lo, torch.device("cpu"))
        dd.junction_chunks, dd.chunk_threshold = chunks, 0
        res = dd.step()
        merged = dd.gather_result(res, g.names)
        if rank == 0:
            open(out_path, "wb").write(format_sv(merged, g.names))
    finally:
        dist.destroy_process_group()


def make_skewed_case():
    """configs[3]-style skew: a third of the reads pile onto two hot junctions (two classes own most of the
    evidence, so one or two of eight owners get most of it), many QNAME duplicates."""
    contigs = [("chr1", 90000), ("chr10", 70000), ("chr2", 80000), ("chrX", 60000)]
    g = synth.make_genome(contigs, seed=51)
    spec = synth.ReadSpec(n_reads=1600, read_len=80, anchor_len=20, n_events=40, singleton_frac=0.2, mate_frac=0.6,
                          hot_frac=0.35, hot_events=2, seed=53, margin=300)
    return g, synth.make_reads(g, spec), 20, 5000, 2


def _worker8(rank, world, port, out_path, stats_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        g, b, A, F, E = make_skewed_case()
        n = b.n
        # uneven shards, two of them empty (ranks 2 and 5), one holding a single read
        cuts = [0, n // 3, n // 3 + 1, n // 3 + 1, n // 2, (3 * n) // 4, (3 * n) // 4, (7 * n) // 8, n]
        lo, hi = cuts[rank], cuts[rank + 1]
        stage = OracleStage(g.as_dict(), g.names, b, lo, hi, A, F, E)
        dd = DistributedDetect(stage, rank, world, lo, hi - lo, torch.device("cpu"))
        dd.junction_chunks = 3
        res = dd.step()
        owned = torch.tensor([dd.stats.n_owned], dtype=torch.int64)
        allo = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allo, owned)
        merged = dd.gather_result(res, g.names)
        if rank == 0:
            open(out_path, "wb").write(format_sv(merged, g.names))
            open(stats_path, "w").write(" ".join(str(int(x.item())) for x in allo))
    finally:
        dist.destroy_process_group()


def test_eight_ranks_uneven_empty_shards_and_a_hot_class(tmp_path):
    """World size 8 (what BASELINE configs[2] shards over): owner = hash(class) % 8, arrival order = input order
    (asserted inside the stand-in's import), merged print order, with empty shards and skewed ownership."""
    g, b, A, F, E = make_skewed_case()
    expected = po.detect(g.fasta(), synth.bowtie_lines(b, g.names).split(b"\n")[:-1], A, F, E)
    assert expected.count(b"\n") > 10
    out, stats = str(tmp_path / "out.sv"), str(tmp_path / "owned.txt")
    mp.spawn(_worker8, args=(8, _free_port(), out, stats), nprocs=8, join=True)
    assert open(out, "rb").read() == expected
    owned = [int(x) for x in open(stats).read().split()]
    assert sum(owned) == b.n and max(owned) > 2 * (b.n // 8), owned     # the hot classes make ownership uneven


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("world,uneven,chunks", [(2, False, 1), (2, True, 3), (3, False, 4), (2, False, 1000)])
def test_sharded_detect_matches_single_process(tmp_path, world, uneven, chunks):
    g, b, A, F, E = make_case()
    expected = po.detect(g.fasta(), synth.bowtie_lines(b, g.names).split(b"\n")[:-1], A, F, E)
    assert expected.count(b"\n") > 5
    out = str(tmp_path / "out.sv")
    mp.spawn(_worker, args=(world, _free_port(), out, uneven, chunks), nprocs=world, join=True)
    assert open(out, "rb").read() == expected


def test_single_rank_path_is_identity():
    g, b, A, F, E = make_case(1)
    stage = OracleStage(g.as_dict(), g.names, b, 0, b.n, A, F, E)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        dd = DistributedDetect(stage, 0, 1, 0, b.n, torch.device("cpu"))
        res = dd.step()
        merged = dd.gather_result(res, g.names)
    finally:
        dist.destroy_process_group()
    expected = po.detect(g.fasta(), synth.bowtie_lines(b, g.names).split(b"\n")[:-1], A, F, E)
    assert format_sv(merged, g.names) == expected
