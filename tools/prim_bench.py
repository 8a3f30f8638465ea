#!/usr/bin/env python3
"""Times the device primitives (primitives.hip) on the cluster stage's shapes: us per call, wall clock
around synchronous test hooks, best of several repetitions.  python tools/prim_bench.py [n]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from breakfast_amd import engine  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    dev = torch.device("cuda", 0)
    e = engine.Engine(20, 5000, 2, device=0)
    vp = C.c_void_p
    for name in ("bfx_sort_pairs_u32", "bfx_sort_pairs_u64"):
        getattr(e.lib, name).argtypes = [vp, vp, vp, vp, vp, C.c_uint64, C.c_uint, C.c_uint]
    e.lib.bfx_scan_sum.argtypes = [vp, vp, vp, C.c_uint64]
    e.lib.bfx_scan_max.argtypes = [vp, vp, C.c_uint64]
    k64 = torch.randint(0, 1 << 45, (n,), dtype=torch.int64, device=dev)
    k32 = torch.randint(0, 1 << 28, (n,), dtype=torch.int32, device=dev)
    v = torch.arange(n, dtype=torch.int32, device=dev)
    ko64, ko32, vo = torch.empty_like(k64), torch.empty_like(k32), torch.empty_like(v)
    out = torch.empty(n + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    def best(fn, reps=7):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return min(ts) * 1e6

    cases = [
        ("sort u32 keys, 32 bits (4 passes)", lambda: e.lib.bfx_sort_pairs_u32(e.h, k32.data_ptr(), ko32.data_ptr(), v.data_ptr(), vo.data_ptr(), n, 0, 32)),
        ("sort u64 keys, 45 bits (6 passes)", lambda: e.lib.bfx_sort_pairs_u64(e.h, k64.data_ptr(), ko64.data_ptr(), v.data_ptr(), vo.data_ptr(), n, 0, 45)),
        ("sort u64 keys, 8 bits (1 pass)", lambda: e.lib.bfx_sort_pairs_u64(e.h, k64.data_ptr(), ko64.data_ptr(), v.data_ptr(), vo.data_ptr(), n, 0, 8)),
        ("exclusive sum scan u32 -> u64", lambda: e.lib.bfx_scan_sum(e.h, k32.data_ptr(), out.data_ptr(), n)),
        ("running max u32 in place", lambda: e.lib.bfx_scan_max(e.h, ko32.data_ptr(), n)),
        ("torch.sort int64 (rocPRIM) for comparison", lambda: (torch.sort(k64), torch.cuda.synchronize())),
        ("torch.cumsum int32->int64 for comparison", lambda: (torch.cumsum(k32, 0, dtype=torch.int64), torch.cuda.synchronize())),
    ]
    for name, fn in cases:
        print("%-45s %9.1f us  (n = %d)" % (name, best(fn), n), flush=True)
    e.close()


if __name__ == "__main__":
    main()
