"""Device-to-host copies right after a large host-to-device upload: are the first ones slower, and for how long?
   python tools/microbench/d2h_after_h2d.py"""
import time, torch
N = 174_000_000
x = torch.empty(N, dtype=torch.uint8, device="cuda")
h = torch.empty(N, dtype=torch.uint8).pin_memory()
big_h = torch.empty(2_000_000_000, dtype=torch.uint8).pin_memory()
big_d = torch.empty(2_000_000_000, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for k in range(3):                                   # warm: both directions once
    h.copy_(x, non_blocking=True); torch.cuda.synchronize()
for pause in (0.0, 0.0, 0.3):
    t = time.perf_counter(); big_d.copy_(big_h, non_blocking=True); torch.cuda.synchronize(); up = time.perf_counter() - t
    time.sleep(pause)
    t0 = time.perf_counter()
    out = []
    for i in range(60):
        t = time.perf_counter(); h.copy_(x, non_blocking=True); torch.cuda.synchronize(); out.append((time.perf_counter() - t) * 1e3)
    slow = [i for i, v in enumerate(out) if v > 3.5]
    print("upload of 2 GB in %.1f ms (%.1f GB/s), pause %.1f s, then 174 MB device-to-host copies, ms:" % (up * 1e3, 2.0 / up, pause), " ".join("%.2f" % v for v in out[:40]))
    print("   copies slower than 3.5 ms: %s (cumulative time at the last one: %.0f ms)" % (slow, sum(out[:slow[-1] + 1]) if slow else 0), flush=True)
