"""Makes tests/golden/shard_ws2_{plain,groups}.npz + shard_ws2.expected_sv (needs an MI355X: run with gpurun, then copy the files
from gpurun_out/).  Two ranks (threads sharing the GPU, host transport) run bf_shard_finish on a small case; what each
rank's bf_shard_result holds is saved as plain arrays, and the oracle's stdout for the whole input beside them.  The
CPU test tests/test_shard_merge.py feeds the two results to the library's merge (bf_shard_format_sv_names) over gloo."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import oracle_lib as ol  # noqa: E402
from breakfast_amd import engine, synth  # noqa: E402
from test_gpu_shard import run_as_threads  # noqa: E402


def case():
    contigs = [("chr1", 60000), ("chr10", 50000), ("chr2", 40000), ("chrM", 30000)]
    g = synth.make_genome(contigs, seed=81, n_runs=[(0, 2000, 500)])
    spec = synth.ReadSpec(n_reads=2400, read_len=90, anchor_len=20, n_events=40, singleton_frac=0.3, mate_frac=0.4,
                          hot_frac=0.2, hot_events=2, seed=83, margin=200, subst_rate=0.01, n_rate=0.002)
    return g, synth.make_reads(g, spec)


def main(out_dir):
    os.makedirs(out_dir, exist_ok=True)
    g, b = case()
    ref = ol.detect_batch(ol.OracleGenome(g.names, g.seqs), g.names, b, 20, 5000, 2)
    open(os.path.join(out_dir, "shard_ws2.expected_sv"), "wb").write(ref.sv)
    cut = b.n * 2 // 5
    for tag, flags in (("plain", 0), ("groups", engine.BF_FLAG_GROUPS_ALWAYS)):
        res = run_as_threads(g, [b.slice(0, cut), b.slice(cut, b.n)], [0, cut], 20, flags=flags, slot_bytes=1 << 16)
        assert engine.shard_format_sv(g.names, res) == ref.sv
        arrays = {"names": np.array(g.names)}
        for r, x in enumerate(res):
            for f in ("clusters", "src_rank", "group_read_begin", "block_off", "record_bytes"):
                arrays["r%d_%s" % (r, f)] = getattr(x, f)
            arrays["r%d_counts" % r] = np.array([x.n_local, x.n_evidence, x.n_records], np.uint64)
        np.savez_compressed(os.path.join(out_dir, "shard_ws2_%s.npz" % tag), **arrays)
        print(tag, [(x.n_local, x.n_evidence, x.clusters.shape[0], x.src_rank.shape[0], len(x.group_read_begin) - 1) for x in res])
    print("oracle: %d clusters, %d bytes" % (ref.n_clusters, len(ref.sv)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else HERE)
