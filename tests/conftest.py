import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests are only skipped when the machine has no GPU at all; on a GPU box a missing or
    # broken HIP library makes them FAIL (there is no CPU fallback to hide behind).
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this machine")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def golden_names():
    return sorted(f[:-3] for f in os.listdir(GOLDEN) if f.endswith(".sv"))


def load_golden(name):
    base = os.path.join(GOLDEN, name)
    fa = open(base + ".fa", "rb").read()
    lines = open(base + ".bowtie", "rb").read()
    A, F, E = (int(x) for x in open(base + ".args").read().split())
    sv = open(base + ".sv", "rb").read()
    return fa, lines, (A, F, E), sv
