"""bf_render_junctions_dev_static -- the entry point that allocates nothing -- against bf_render_junctions_dev,
through the C ABI, bit-exact: offsets, total and every record byte, on two calls in a row (the second with a
smaller input: the static scratch array keeps the first call's contents), plus its limits.

Collected last (file name): a failure here must not hide the other GPU tests from `pytest -x`.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from breakfast_amd import engine, synth

pytestmark = pytest.mark.gpu


def make_case():
    contigs = [("chr1", 300000), ("chr10", 200000), ("chr2", 250000)]
    g = synth.make_genome(contigs, seed=71, n_runs=[(0, 5000, 2000)], lower_runs=[(2, 100000, 30000)])
    spec = synth.ReadSpec(n_reads=40000, read_len=130, anchor_len=20, n_events=300, singleton_frac=0.3, mate_frac=0.4,
                          seed=73, margin=300, subst_rate=0.02)
    b = synth.make_reads(g, spec)
    # a few reads with bases outside ACGT (exception entries in their records)
    sb = b.seq_bytes.copy()
    for i in range(0, b.n, 97):
        sb[int(b.seq_off[i]) + 50] = ord("N")
    b.seq_bytes = sb
    return g, b


def render(eng, call, idx_np, dev):
    """-> (offsets, record bytes, total) of one render call on a device index list"""
    n = int(idx_np.size)
    idx = torch.from_numpy(idx_np.astype(np.int32)).to(dev)
    off = torch.full((n + 1,), -1, dtype=torch.int64, device=dev)
    cap = n * eng.junction_record_bound() + 64
    out = torch.full((cap,), 0xEE, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    total = call(n, idx.data_ptr() if n else 0, off.data_ptr(), out.data_ptr(), cap)
    torch.cuda.synchronize()
    return off.cpu().numpy(), out.cpu().numpy(), total


def test_static_scratch_entry_point_matches_the_allocating_one():
    g, b = make_case()
    dev = torch.device("cuda", 0)
    eng = engine.Engine(20, 5000, 2, device=0)
    other = engine.Engine(20, 5000, 2, device=0)       # a second ctx on the same GPU shares the array
    try:
        for e in (eng, other):
            e.set_genome(g.names, g.seqs)
        eng.submit(b)
        eng.localise()
        other.submit(b.slice(0, 5000))
        other.localise()
        rng = np.random.default_rng(11)
        # two calls in a row, the second smaller; then empty, one entry, and a larger one again
        for n in (30000, 1234, 0, 1, 40000):
            idx = rng.integers(0, b.n, size=n).astype(np.uint32)
            off_a, by_a, tot_a = render(eng, eng.render_junctions_dev, idx, dev)
            off_s, by_s, tot_s = render(eng, eng.render_junctions_dev_static, idx, dev)
            assert tot_s == tot_a
            assert np.array_equal(off_s, off_a)
            assert np.array_equal(by_s, by_a)                      # record bytes AND the untouched tail
            if n:
                assert tot_a > 0 and int(off_a[-1]) == tot_a
            # size query (no output buffer): offsets and total only
            offq = torch.zeros(n + 1, dtype=torch.int64, device=dev)
            idx_t = torch.from_numpy(idx.astype(np.int32)).to(dev)
            assert eng.render_junctions_dev_static(n, idx_t.data_ptr() if n else 0, offq.data_ptr()) == tot_a
            torch.cuda.synchronize()
            assert np.array_equal(offq.cpu().numpy(), off_a)
            # the other ctx uses the same array in between
            idx2 = rng.integers(0, 5000, size=777).astype(np.uint32)
            o1 = render(other, other.render_junctions_dev, idx2, dev)
            o2 = render(other, other.render_junctions_dev_static, idx2, dev)
            assert o1[2] == o2[2] and np.array_equal(o1[0], o2[0]) and np.array_equal(o1[1], o2[1])
    finally:
        eng.close()
        other.close()


def test_static_scratch_entry_point_allocates_nothing():
    """What the entry point is for: after a warm-up call, calls of any size up to the limit leave the ctx's device memory
    (bf_stats: every hipMalloc of the library goes through one book-keeping function) and the GPU's free memory as they were."""
    g, b = make_case()
    dev = torch.device("cuda", 0)
    eng = engine.Engine(20, 5000, 2, device=0)
    try:
        eng.set_genome(g.names, g.seqs)
        eng.submit(b)
        eng.localise()
        rng = np.random.default_rng(12)
        sizes = (100, 40000, 7, 25000)
        # every torch tensor up front: torch's allocator must not move the GPU's free memory during the measured calls
        idx = {n: torch.from_numpy(rng.integers(0, b.n, size=n).astype(np.int32)).to(dev) for n in sizes}
        off = torch.zeros(max(sizes) + 1, dtype=torch.int64, device=dev)
        cap = max(sizes) * eng.junction_record_bound() + 64
        out = torch.zeros(cap, dtype=torch.uint8, device=dev)
        eng.render_junctions_dev_static(sizes[0], idx[sizes[0]].data_ptr(), off.data_ptr(), out.data_ptr(), cap)      # warm-up
        torch.cuda.synchronize()
        held = eng.stats()[1]
        free0 = torch.cuda.mem_get_info(dev)[0]
        for n in sizes[1:]:
            assert eng.render_junctions_dev_static(n, idx[n].data_ptr(), off.data_ptr(), out.data_ptr(), cap) > 0
            assert eng.stats()[1] == held, "the static entry point grew the ctx's device memory"
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(dev)[0] == free0, "the GPU's free memory changed across the static calls"
        # (the allocating entry point, for contrast, is allowed to: its buffers grow with the list)
    finally:
        eng.close()


def test_static_scratch_entry_point_limits_and_errors():
    g, b = make_case()
    dev = torch.device("cuda", 0)
    eng = engine.Engine(20, 5000, 2, device=0)
    try:
        eng.set_genome(g.names, g.seqs)
        eng.submit(b.slice(0, 2000))
        off = torch.zeros(8, dtype=torch.int64, device=dev)
        idx = torch.zeros(8, dtype=torch.int32, device=dev)
        with pytest.raises(engine.BreakfastError) as e:            # before bf_localise
            eng.render_junctions_dev_static(4, idx.data_ptr(), off.data_ptr())
        assert e.value.code == -4
        eng.localise()
        # more entries than the static array holds: refused before anything is touched
        with pytest.raises(engine.BreakfastError) as e:
            eng.render_junctions_dev_static((1 << 20) + 1, idx.data_ptr(), off.data_ptr())
        assert e.value.code == -5 and "static scratch" in str(e.value)
        # an index outside the resident reads is counted, never dereferenced -- and the next call starts clean
        bad = torch.tensor([0, 1, 2000, 7], dtype=torch.int32, device=dev)
        with pytest.raises(engine.BreakfastError) as e:
            eng.render_junctions_dev_static(4, bad.data_ptr(), off.data_ptr())
        assert e.value.code == -1 and "1 of 4" in str(e.value)
        good = np.array([0, 1, 1999, 7], np.uint32)
        a = render(eng, eng.render_junctions_dev, good, dev)
        s = render(eng, eng.render_junctions_dev_static, good, dev)
        assert a[2] == s[2] and np.array_equal(a[0], s[0]) and np.array_equal(a[1], s[1])
        # too small an output buffer
        out = torch.zeros(16, dtype=torch.uint8, device=dev)
        gi = torch.from_numpy(good.astype(np.int32)).to(dev)
        with pytest.raises(engine.BreakfastError) as e:
            eng.lib.bf_render_junctions_dev_static.restype = C.c_int
            eng.render_junctions_dev_static(4, gi.data_ptr(), off.data_ptr(), out.data_ptr(), 16)
        assert e.value.code == -1
    finally:
        eng.close()
