"""bf_render_junctions_dev (device index list -> junction RECORDS in device memory) against bf_render_junctions (host
index list -> TEXT), and its refusal of indices outside the resident reads."""
import numpy as np
import pytest
import torch

from breakfast_amd import engine, synth

pytestmark = pytest.mark.gpu


def test_device_render_matches_host_render_and_rejects_bad_indices():
    contigs = [("chr1", 300000), ("chr10", 200000), ("chr2", 250000)]
    g = synth.make_genome(contigs, seed=61, n_runs=[(0, 5000, 2000)], lower_runs=[(2, 100000, 30000)])
    spec = synth.ReadSpec(n_reads=30000, read_len=120, anchor_len=20, n_events=300, singleton_frac=0.3, mate_frac=0.4,
                          seed=63, margin=300)
    b = synth.make_reads(g, spec)
    dev = torch.device("cuda", 0)
    eng = engine.Engine(20, 5000, 2, device=0)
    try:
        eng.set_genome(g.names, g.seqs)
        eng.submit(b)
        eng.localise()
        rng = np.random.default_rng(5)
        cap = 5000 * eng.junction_record_bound() + 64
        out = torch.zeros(cap, dtype=torch.uint8, device=dev)
        for n in (0, 1, 777, 5000):
            idx = rng.integers(0, b.n, size=n).astype(np.uint32)
            by_h, off_h = eng.render_junctions(idx)
            idx_t = torch.from_numpy(idx.astype(np.int32)).to(dev)
            off_t = torch.zeros(n + 1, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            total = eng.render_junctions_dev(n, idx_t.data_ptr() if n else 0, off_t.data_ptr(), out.data_ptr(), cap)
            assert eng.render_junctions_dev(n, idx_t.data_ptr() if n else 0, off_t.data_ptr()) == total      # size query
            rec, roff = out.cpu().numpy().tobytes(), off_t.cpu().numpy()
            assert int(roff[-1]) == total
            # the device call hands out junction RECORDS, the host call their text
            text = [engine.junction_text(rec[int(roff[i]):int(roff[i + 1])]) for i in range(n)]
            assert [len(t) for t in text] == list(np.diff(off_h).astype(np.int64))
            assert b"".join(text) == by_h.tobytes()
        bad = torch.tensor([0, b.n, 3], dtype=torch.int32, device=dev)
        off_t = torch.zeros(4, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        with pytest.raises(engine.BreakfastError) as ei:
            eng.render_junctions_dev(3, bad.data_ptr(), off_t.data_ptr(), out.data_ptr(), cap)
        assert ei.value.code == -1
    finally:
        eng.close()
