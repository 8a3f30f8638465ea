"""A fixed sample of tests/fuzz_gpu.py's random mid-size cases inside the GPU suite: every case through seven ways of
driving the library (ASCII / 4-bit / 2-bit reads, one go and contig groups, the exact path in pieces, reads kept on the
host, two and three ranks over the host transport), the whole .sv against the oracle's bytes.  The campaigns run by hand
(thousands of cases, the command line, multi-million-read cases) are in profiles/r03o_fuzz_summary.txt."""
import pytest

import fuzz_gpu as fz
from breakfast_amd import engine

pytestmark = pytest.mark.gpu

# seeds whose case the generator can place and the oracle finishes in under two seconds
SEEDS = [0, 1, 2, 3, 7, 8, 10, 11, 12, 14, 16, 18, 19, 21, 22, 24, 25, 26, 28, 30, 31, 32, 34, 35, 36, 37, 39, 40]


@pytest.mark.parametrize("seed", SEEDS)
def test_random_case_through_every_way(seed):
    g, b, A, F, E, what = fz.make_case(seed)
    want = fz.ol.detect_batch(fz.ol.OracleGenome(g.names, g.seqs), g.names, b, A, F, E).sv
    G, X, H = engine.BF_FLAG_GROUPS_ALWAYS, engine.BF_FLAG_FORCE_EXACT, engine.BF_FLAG_READS_ON_HOST
    assert fz.one_gpu(g, b, A, F, E, 0, "ascii") == want, ("ascii", what)
    assert fz.one_gpu(g, b, A, F, E, G, "packed") == want, ("packed, groups", what)
    assert fz.one_gpu(g, b, A, F, E, 0, "packed2") == want, ("packed2", what)
    assert fz.one_gpu(g, b, A, F, E, X | G, "pieces") == want, ("exact, groups, pieces", what)
    assert fz.one_gpu(g, b, A, F, E, H | G, "packed") == want, ("reads on host, groups", what)
    assert fz.over_ranks(g, b, A, F, E, 2, 0) == want, ("two ranks", what)
    assert fz.over_ranks(g, b, A, F, E, 3, G) == want, ("three ranks, groups", what)
