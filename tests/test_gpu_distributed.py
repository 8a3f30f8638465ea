"""2-rank run of the sharded path with the real HIP engine (both ranks share the one GPU of the test
box; records travel through gloo with CPU staging, the device stages are the product's)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as ol

from breakfast_amd import engine, synth
from breakfast_amd.distributed import DistributedDetect, HipStageEngine, format_sv

pytestmark = pytest.mark.gpu


def make_case():
    contigs = [("chr1", 300000), ("chr10", 200000), ("chr2", 250000)]
    g = synth.make_genome(contigs, seed=61, n_runs=[(0, 5000, 2000)], lower_runs=[(2, 100000, 30000)])
    spec = synth.ReadSpec(n_reads=30000, read_len=120, anchor_len=20, n_events=300, singleton_frac=0.3, mate_frac=0.4,
                          seed=63, margin=300)
    return g, synth.make_reads(g, spec), 20, 5000, 2


def _worker(rank, world, port, out_path, flags=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g, b, A, F, E = make_case()
        cuts = [0, b.n // 3, b.n] if world == 2 else [0, b.n // 4, b.n // 2, b.n]
        lo, hi = cuts[rank], cuts[rank + 1]
        dev = torch.device("cuda", 0)
        eng = engine.Engine(A, F, E, device=0, flags=flags)
        eng.set_genome(g.names, g.seqs)
        eng.set_rank(rank, world, lo)
        eng.submit(b.slice(lo, hi))
        dd = DistributedDetect(HipStageEngine(eng, dev), rank, world, lo, hi - lo, dev)
        dd.junction_chunks, dd.chunk_threshold = 3, 0       # junction records in three pieces even at this size
        for _ in range(2):                      # the step must be repeatable on resident reads
            res = dd.step()
        merged = dd.gather_result(res, g.names)
        if rank == 0:
            open(out_path, "wb").write(format_sv(merged, g.names))
        eng.close()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("world,flags", [(2, 0), (3, 0), (2, engine.BF_FLAG_DEVICE_ROUTING)])
def test_sharded_hip_detect_matches_oracle(tmp_path, world, flags):
    g, b, A, F, E = make_case()
    ref = ol.detect_batch(ol.OracleGenome(g.names, g.seqs), g.names, b, A, F, E)
    assert ref.n_clusters > 50
    out = str(tmp_path / "out.sv")
    mp.spawn(_worker, args=(world, _free_port(), out, flags), nprocs=world, join=True)
    assert open(out, "rb").read() == ref.sv


def test_device_render_matches_host_render_and_rejects_bad_indices():
    """bf_render_junctions_dev (device index list -> device text) against bf_render_junctions."""
    g, b, A, F, E = make_case()
    eng = engine.Engine(A, F, E, device=0)
    try:
        eng.set_genome(g.names, g.seqs)
        eng.submit(b)
        eng.localise()
        dev = torch.device("cuda", 0)
        stage = HipStageEngine(eng, dev)
        rng = np.random.default_rng(5)
        for n in (0, 1, 777, 5000):
            idx = rng.integers(0, b.n, size=n).astype(np.uint32)
            by_h, off_h = eng.render_junctions(idx)
            by_d, off_d = stage.render_junctions_tensor(torch.from_numpy(idx.astype(np.int32)).to(dev))
            # the device call hands out junction RECORDS, the host call their text
            rec, roff = by_d.cpu().numpy().tobytes(), off_d.cpu().numpy()
            text = [engine.junction_text(rec[int(roff[i]):int(roff[i + 1])]) for i in range(n)]
            assert [len(t) for t in text] == list(np.diff(off_h).astype(np.int64))
            assert b"".join(text) == by_h.tobytes()
        bad = torch.tensor([0, b.n, 3], dtype=torch.int32, device=dev)
        with pytest.raises(engine.BreakfastError) as ei:
            stage.render_junctions_tensor(bad)
        assert ei.value.code == -1
    finally:
        eng.close()


def _worker_nccl_single(rank, world, port, out_path):
    """World size 1 over the nccl (RCCL) backend: the collectives move device tensors and run on torch's
    stream, as they do on a multi-GPU node (one GPU cannot host two RCCL ranks)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        g, b, A, F, E = make_case()
        dev = torch.device("cuda", 0)
        eng = engine.Engine(A, F, E, device=0, flags=engine.BF_FLAG_DEVICE_ROUTING)
        eng.set_genome(g.names, g.seqs)
        eng.set_rank(0, 1, 0)
        eng.submit(b)
        dd = DistributedDetect(HipStageEngine(eng, dev), 0, 1, 0, b.n, dev)
        for _ in range(3):
            res = dd.step()
        merged = dd.gather_result(res, g.names)
        open(out_path, "wb").write(format_sv(merged, g.names))
        _check_layout_and_attach_calls(eng, dd.stage, res, g, b, A, F, E, dev)
        eng.close()
    finally:
        dist.destroy_process_group()


def _check_layout_and_attach_calls(eng, stage, res, g, b, A, F, E, dev):
    """The sharded result is laid out as bf_finish lays it out on one GPU -- with one rank it IS that result,
    byte for byte --, and bf_result_layout_dev / bf_result_attach_records_dev refuse what does not fit."""
    one = engine.Engine(A, F, E, device=0)
    one.set_genome(g.names, g.seqs)
    one.submit(b)
    r1 = one.finish()
    assert res.cluster_junction_off is not None
    assert np.array_equal(res.cluster_junction_off, r1.cluster_junction_off)
    assert np.array_equal(res.junction_bytes, r1.junction_bytes)
    assert np.array_equal(res.clusters, r1.clusters) and np.array_equal(res.reads, r1.reads)
    bare = one.finish(reads=False)
    assert one.format_sv(bare) == eng.format_sv(res)
    one.close()

    nr = res.n_reads
    sr, si, jl = stage.result_sources_tensors(res)
    rng = np.random.default_rng(3)

    def piece_records(r0, r1_, shuffled):
        pos = np.arange(r0, r1_, dtype=np.int64)
        if shuffled:
            rng.shuffle(pos)
        pos_t = torch.from_numpy(pos).to(dev)
        by, off = stage.render_junctions_tensor(si[pos_t])
        return by, pos_t

    # the same result again through the calls themselves: 5 pieces, records of a piece in shuffled order
    rb, pb = eng.result_layout_dev(5)
    assert list(rb) == [nr * c // 5 for c in range(6)] and int(pb[0]) == 0 and int(pb[-1]) == res.junction_bytes.size
    keep = []
    for c in (3, 0, 4, 1, 2):                                           # pieces in any order
        by, pos_t = piece_records(int(rb[c]), int(rb[c + 1]), shuffled=True)
        assert by.numel() == int(pb[c + 1] - pb[c])
        stage.attach_records(res, c, by, pos_t)
        keep.append((by, pos_t))
    again = eng.attach_finish()
    assert np.array_equal(again.junction_bytes, res.junction_bytes) and np.array_equal(again.reads, res.reads)
    assert np.array_equal(again.cluster_junction_off, res.cluster_junction_off)

    # the library's own request order (bf_result_requests_dev): with one rank, result order; records of a piece
    # then go in without positions
    req, cnt, gby, rb, pb = stage.result_requests(res, 3, 1)
    assert torch.equal(req, si) and cnt.shape == (3, 1)
    assert list(cnt[:, 0]) == [int(rb[c + 1] - rb[c]) for c in range(3)]
    assert list(gby[:, 0]) == [int(pb[c + 1] - pb[c]) for c in range(3)]
    keep = []
    for c in (2, 0, 1):
        by, off = stage.render_junctions_tensor(req[int(rb[c]):int(rb[c + 1])])
        with pytest.raises(engine.BreakfastError):          # positions are not accepted in this mode
            eng.attach_records_dev(c, by.data_ptr(), by.numel(), off.data_ptr())
        stage.attach_records(res, c, by, None)
        keep.append(by)
    again = eng.attach_finish()
    assert np.array_equal(again.junction_bytes, res.junction_bytes) and np.array_equal(again.reads, res.reads)

    # refused: wrong byte count, unknown piece, a piece twice, finish with a piece missing
    rb, pb = eng.result_layout_dev(2)
    by0, pos0 = piece_records(int(rb[0]), int(rb[1]), shuffled=False)
    for call, code in ((lambda: eng.attach_records_dev(0, by0.data_ptr(), by0.numel() - 4, pos0.data_ptr()), -1),
                       (lambda: eng.attach_records_dev(2, by0.data_ptr(), by0.numel(), pos0.data_ptr()), -1),
                       (lambda: eng.attach_finish(), -4)):
        with pytest.raises(engine.BreakfastError) as ei:
            call()
        assert ei.value.code == code
    stage.attach_records(res, 0, by0, pos0)
    with pytest.raises(engine.BreakfastError) as ei:
        eng.attach_records_dev(0, by0.data_ptr(), by0.numel(), pos0.data_ptr())
    assert ei.value.code == -4
    # positions outside the piece / records that do not add up to the announced sizes: caught at the end
    by1, pos1 = piece_records(int(rb[1]), int(rb[2]), shuffled=False)
    wrong = pos1.clone()
    wrong[0] = 0
    stage.attach_records(res, 1, by1, wrong)
    with pytest.raises(engine.BreakfastError) as ei:
        eng.attach_finish()
    assert ei.value.code == -1
    with pytest.raises(engine.BreakfastError):
        eng.result_layout_dev(65)


def test_sharded_path_over_rccl_with_one_rank(tmp_path):
    g, b, A, F, E = make_case()
    ref = ol.detect_batch(ol.OracleGenome(g.names, g.seqs), g.names, b, A, F, E)
    out = str(tmp_path / "out.sv")
    mp.spawn(_worker_nccl_single, args=(1, _free_port(), out), nprocs=1, join=True)
    assert open(out, "rb").read() == ref.sv


def _worker_scaled(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g, b, A = scaled_case()
        n = b.n
        cuts = [0, (2 * n) // 5, n]
        lo, hi = cuts[rank], cuts[rank + 1]
        dev = torch.device("cuda", 0)
        eng = engine.Engine(A, 5000, 2, device=0, flags=engine.BF_FLAG_DEVICE_ROUTING)
        eng.set_genome(g.names, g.seqs)
        eng.set_rank(rank, world, lo)
        eng.submit(b.slice(lo, hi))
        dd = DistributedDetect(HipStageEngine(eng, dev), rank, world, lo, hi - lo, dev)
        res = dd.step()
        merged = dd.gather_result(res, g.names)
        if rank == 0:
            open(out_path, "wb").write(format_sv(merged, g.names))
        eng.close()
    finally:
        dist.destroy_process_group()


def scaled_case():
    contigs, gk, spec = synth.config(1, scale=0.05, genome_scale=0.05)      # 500k reads, 154 Mb genome, 25 contigs
    spec.margin = 10000 + 2 * spec.read_len + 128
    g = synth.make_genome(contigs, **gk)
    return g, synth.make_reads(g, spec), spec.anchor_len


def config2_case():
    """BASELINE configs[2] (100M reads on hg38, sharded over 2/4/8 GPUs), scaled: 400k reads, 1.5 % of the genome."""
    contigs, gk, spec = synth.config(2, scale=0.004, genome_scale=0.015)
    spec.margin = 10000 + 2 * spec.read_len + 128
    g = synth.make_genome(contigs, **gk)
    return g, synth.make_reads(g, spec), spec.anchor_len


def _worker_config2(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g, b, A = config2_case()
        n = b.n
        cuts = [n * r // world for r in range(world + 1)]               # contiguous line ranges, as configs[2] says
        lo, hi = cuts[rank], cuts[rank + 1]
        dev = torch.device("cuda", 0)
        eng = engine.Engine(A, 5000, 2, device=0, flags=engine.BF_FLAG_DEVICE_ROUTING)
        eng.set_genome(g.names, g.seqs)
        eng.set_rank(rank, world, lo)
        eng.submit(b.slice(lo, hi))
        dd = DistributedDetect(HipStageEngine(eng, dev), rank, world, lo, hi - lo, dev)
        res = dd.step()
        merged = dd.gather_result(res, g.names)
        if rank == 0:
            open(out_path, "wb").write(format_sv(merged, g.names))
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_config2_scaled_over_ranks(tmp_path, world):
    g, b, A = config2_case()
    ref = ol.detect_batch(ol.OracleGenome(g.names, g.seqs), g.names, b, A, 5000, 2)
    assert ref.n_clusters > 2000
    out = str(tmp_path / "out.sv")
    mp.spawn(_worker_config2, args=(world, _free_port(), out), nprocs=world, join=True)
    assert open(out, "rb").read() == ref.sv


def test_two_ranks_at_half_a_million_reads(tmp_path):
    g, b, A = scaled_case()
    ref = ol.detect_batch(ol.OracleGenome(g.names, g.seqs), g.names, b, A, 5000, 2)
    assert ref.n_clusters > 3000
    out = str(tmp_path / "out.sv")
    mp.spawn(_worker_scaled, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out, "rb").read() == ref.sv
