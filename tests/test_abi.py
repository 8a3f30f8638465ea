"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every declared symbol."""
import ctypes as C
import os
import re

import pytest

from breakfast_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "breakfast_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bf_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_all_exported():
    lib = engine.load_library()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "libbreakfast_hip.so does not export %s" % s
    assert sorted(engine.ABI_SYMBOLS) == syms, "engine.ABI_SYMBOLS out of date with the header"
    header = open(os.path.join(ROOT, "include", "breakfast_hip.h")).read()
    assert lib.bf_abi_version() == int(re.search(r"#define BF_ABI_VERSION (\d+)", header).group(1))


def test_struct_layouts_match_header():
    assert C.sizeof(engine.BfConfig) == 32
    assert C.sizeof(engine.BfBatch) == 48
    assert C.sizeof(engine.BfResult) == 64
    assert engine.HIT_DTYPE.itemsize == 16
    assert engine.CLUSTER_DTYPE.itemsize == 56
    assert engine.READ_DTYPE.itemsize == 16


def test_no_cpu_fallback_without_gpu():
    """Without a GPU the product must refuse to run, not fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this machine has a GPU")
    with pytest.raises(engine.BreakfastError) as e:
        engine.Engine(anchor_len=20)
    assert e.value.code == -2          # BF_ERR_NO_DEVICE
    assert "no CPU path" in str(e.value)


def test_config_validation_precedes_device_probe():
    with pytest.raises(engine.BreakfastError) as e:
        engine.Engine(anchor_len=7)    # the reference panics below 8 (detect.rs:166)
    assert e.value.code == -5


def test_product_does_not_import_the_oracle():
    """breakfast_amd/ (Python and C++/HIP sources) must never reference oracle/."""
    pkg = os.path.join(ROOT, "breakfast_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".inc")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                for word in ("detect_oracle", "dispatch_oracle", "filter_oracle", "annotate_oracle", "liboracle", "oracle_lib", "oracle/"):
                    assert word not in text, "%s mentions %s" % (f, word)
