"""The hand-written device primitives of the cluster stage (breakfast_amd/csrc/primitives.hip): stable LSD
radix sort of (key, value) pairs and the two chained scans, against torch/numpy on the same inputs.
Bit-exact (integers)."""
import ctypes as C

import numpy as np
import pytest
import torch

from breakfast_amd import engine

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 63, 64, 65, 4095, 4096, 4097, 8191, 100_003, 1_000_000, 3_333_333]


@pytest.fixture(scope="module")
def eng():
    e = engine.Engine(20, 5000, 2, device=0)
    vp = C.c_void_p
    for name in ("bfx_sort_pairs_u32", "bfx_sort_pairs_u64"):
        getattr(e.lib, name).argtypes = [vp, vp, vp, vp, vp, C.c_uint64, C.c_uint, C.c_uint]
    e.lib.bfx_scan_sum.argtypes = [vp, vp, vp, C.c_uint64]
    e.lib.bfx_scan_max.argtypes = [vp, vp, C.c_uint64]
    yield e
    e.close()


def _sort(eng, keys, begin, end):
    dev = torch.device("cuda", 0)
    n = keys.numel()
    vals = torch.arange(n, dtype=torch.int32, device=dev)
    kout, vout = torch.empty_like(keys), torch.empty_like(vals)
    fn = eng.lib.bfx_sort_pairs_u64 if keys.dtype == torch.int64 else eng.lib.bfx_sort_pairs_u32
    torch.cuda.synchronize()
    rc = fn(eng.h, keys.data_ptr(), kout.data_ptr(), vals.data_ptr(), vout.data_ptr(), n, begin, end)
    assert rc == 0, eng.lib.bf_last_error(eng.h)
    return kout, vout


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("bits", [(0, 64), (0, 45), (3, 20), (0, 1), (0, 8), (16, 33)])
def test_sort_pairs_u64_is_a_stable_sort_on_the_bit_range(eng, n, bits):
    begin, end = bits
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(n * 131 + begin * 7 + end)
    # few distinct values in part of the key: long runs of equal digits, stability matters
    keys = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device=dev, generator=g)
    keys = torch.where(torch.rand(n, device=dev, generator=g) < 0.5, keys & 0x0000FF00FF0000FF, keys)
    if end == 64:
        keys = keys | (torch.randint(0, 2, (n,), dtype=torch.int64, device=dev, generator=g) << 63)     # the top bit too
    keys0 = keys.clone()
    kout, vout = _sort(eng, keys, begin, end)
    assert torch.equal(keys, keys0), "the input must stay untouched"
    width = end - begin
    sub = (keys >> begin) & ((1 << width) - 1) if width < 64 else keys
    if width == 64:     # unsigned order of all 64 bits: flip the sign bit for torch's signed sort
        sub = keys ^ (-(1 << 63))
    order = torch.sort(sub, stable=True).indices
    assert torch.equal(vout.to(torch.int64), order)
    assert torch.equal(kout, keys[order])


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("bits", [(0, 32), (0, 3), (5, 29)])
def test_sort_pairs_u32(eng, n, bits):
    begin, end = bits
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(n * 17 + end)
    keys64 = torch.randint(0, 1 << 32, (n,), dtype=torch.int64, device=dev, generator=g)
    keys64 = torch.where(torch.rand(n, device=dev, generator=g) < 0.3, keys64 & 0xFF0000FF, keys64)
    keys = keys64.to(torch.uint32).view(torch.int32) if hasattr(torch, "uint32") else None
    if keys is None:
        keys = torch.from_numpy(keys64.cpu().numpy().astype(np.uint32).view(np.int32)).to(dev)
    kout, vout = _sort(eng, keys, begin, end)
    sub = (keys64 >> begin) & ((1 << (end - begin)) - 1)
    order = torch.sort(sub, stable=True).indices
    assert torch.equal(vout.to(torch.int64), order)
    assert torch.equal(kout, keys[order])


def test_sort_of_all_equal_and_presorted_keys(eng):
    dev = torch.device("cuda", 0)
    n = 1_234_567
    same = torch.full((n,), 0x1234_5678_9ABC, dtype=torch.int64, device=dev)
    kout, vout = _sort(eng, same, 0, 48)
    assert torch.equal(vout, torch.arange(n, dtype=torch.int32, device=dev)) and torch.equal(kout, same)
    up = torch.arange(n, dtype=torch.int64, device=dev)
    kout, vout = _sort(eng, up, 0, 21)
    assert torch.equal(kout, up)
    kout, vout = _sort(eng, up.flip(0).contiguous(), 0, 21)
    assert torch.equal(kout, up)


@pytest.mark.parametrize("n", [0] + SIZES + [10_000_001])
def test_scans(eng, n):
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(n + 5)
    x = torch.randint(0, 1 << 32, (n,), dtype=torch.int64, device=dev, generator=g)
    x32 = torch.from_numpy(x.cpu().numpy().astype(np.uint32).view(np.int32)).to(dev)
    out = torch.full((n + 1,), -1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    assert eng.lib.bfx_scan_sum(eng.h, x32.data_ptr(), out.data_ptr(), n) == 0
    want = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(x, 0, out=want[1:])
    assert torch.equal(out, want)
    # running maximum, in place; sparse non-zero entries like the segment-head marks it is used on
    y = torch.where(torch.rand(n, device=dev, generator=g) < 0.01, x, torch.zeros_like(x))
    y32 = torch.from_numpy(y.cpu().numpy().astype(np.uint32).view(np.int32)).to(dev)
    torch.cuda.synchronize()
    assert eng.lib.bfx_scan_max(eng.h, y32.data_ptr(), n) == 0
    got = torch.from_numpy(y32.cpu().numpy().view(np.uint32).astype(np.int64))
    assert torch.equal(got, torch.cummax(y, 0).values.cpu() if n else got)
