"""The N > 1 path on CPU: world size 2 over gloo, of code that SHIPS.  What a multi-process run does
after bf_shard_finish is host work: every rank holds a bf_shard_result (its rows as an owner, its records as a source),
the results travel to the rank that prints, and the library's merge -- bf_shard_format_sv_names, pure host code, no GPU
and no ctx -- orders the owners' rows by the seed's rank in the reference's sort (detect.rs:193-198) and finds every
supporting read's record on its source rank.  The two results are fixtures made on an MI355X by
tests/golden/make_shard_fixture.py (two ranks through bf_shard_finish), the expected bytes are the oracle's."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN

from breakfast_amd import engine


def load_rank(tag, r):
    z = np.load(os.path.join(GOLDEN, "shard_ws2_%s.npz" % tag), allow_pickle=False)
    c = z["r%d_counts" % r]
    return [bytes(n) for n in z["names"]], engine.ShardResult(int(c[0]), int(c[1]), z["r%d_clusters" % r].view(engine.CLUSTER_DTYPE).reshape(-1),
                                                               z["r%d_src_rank" % r], z["r%d_group_read_begin" % r], int(c[2]),
                                                               z["r%d_block_off" % r], z["r%d_record_bytes" % r])


def golden_sv():
    return open(os.path.join(GOLDEN, "shard_ws2.expected_sv"), "rb").read()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, tag, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        names, mine = load_rank(tag, rank)                       # this rank's own result only
        box = [None] * world
        dist.all_gather_object(box, mine)                        # the results travel (pickled numpy arrays)
        sv = engine.shard_format_sv(names, box)                  # every rank can print
        open(os.path.join(out_dir, "rank%d.sv" % rank), "wb").write(sv)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tag", ["plain", "groups"])
def test_two_ranks_over_gloo_merge_to_the_reference_table(tmp_path, tag):
    mp.spawn(_worker, args=(2, _free_port(), tag, str(tmp_path)), nprocs=2, join=True)
    want = golden_sv()
    assert want.count(b"\n") > 10
    for r in range(2):
        assert open(str(tmp_path / ("rank%d.sv" % r)), "rb").read() == want


@pytest.mark.parametrize("tag", ["plain", "groups"])
def test_merge_properties(tag):
    names, r0 = load_rank(tag, 0)
    _, r1 = load_rank(tag, 1)
    want = golden_sv()
    assert engine.shard_format_sv(names, [r0, r1]) == want
    assert r0.clusters.shape[0] and r1.clusters.shape[0]                    # both ranks own rows
    assert set(np.unique(r0.src_rank)) == {0, 1}                            # ... whose reads live on both ranks
    if tag == "groups":
        assert len(r0.group_read_begin) == 3                                # two rounds
    # ranks in the wrong order: an owner's requests are no longer where the block offsets say -> refused, or at least not
    # the reference's table (records are self-describing, so a walk from a wrong offset may still parse)
    try:
        assert engine.shard_format_sv(names, [r1, r0]) != want
    except engine.BreakfastError:
        pass
    # a truncated record buffer is refused
    import copy
    cut = copy.copy(r1)
    cut.record_bytes = r1.record_bytes[:-8]
    with pytest.raises(engine.BreakfastError):
        engine.shard_format_sv(names, [r0, cut])
    # one rank alone (the other's rows and records missing) is refused as well
    with pytest.raises(engine.BreakfastError):
        engine.shard_format_sv(names, [r0])
