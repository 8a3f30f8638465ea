#!/usr/bin/env python3
"""Device-to-host copy of the result's size (174 MB) as one copy, as 2 / 4 / 8 concurrent copies on streams of their own, and
by a kernel that stores into mapped pinned memory: does anything beat one hipMemcpyAsync's 54.5 GB/s?
   python3 tools/microbench/d2h_split.py"""
import time
import torch

n = 174_000_000
dev = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255)
host = torch.empty(n, dtype=torch.uint8).pin_memory()
streams = [torch.cuda.Stream() for _ in range(8)]


def run(parts, reps=12):
    cut = [n * k // parts for k in range(parts + 1)]
    cut = [c // 4096 * 4096 for c in cut[:-1]] + [n]
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(parts):
            with torch.cuda.stream(streams[k]):
                host[cut[k]:cut[k + 1]].copy_(dev[cut[k]:cut[k + 1]], non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


for parts in (1, 2, 4, 8, 1):
    t = run(parts)
    print("%d concurrent copies: %.3f ms, %.1f GB/s" % (parts, t * 1e3, n / t / 1e9), flush=True)
# sequential pieces on one stream (what the groups do)
for pieces in (3, 6):
    cut = [n * k // pieces for k in range(pieces + 1)]
    best = 1e9
    for _ in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(streams[0]):
            for k in range(pieces):
                host[cut[k]:cut[k + 1]].copy_(dev[cut[k]:cut[k + 1]], non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("%d pieces back to back on one stream: %.3f ms, %.1f GB/s" % (pieces, best * 1e3, n / best / 1e9), flush=True)
