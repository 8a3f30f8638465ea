"""Does a chain of device-to-host copies that each WAIT FOR AN EVENT of a busy compute stream start out slower?
(the library's record copies do: tools/step_series.py)   python tools/microbench/d2h_series2.py"""
import time, torch
N = 174_000_000
x = torch.empty(N, dtype=torch.uint8, device="cuda")
h = torch.empty(N, dtype=torch.uint8).pin_memory()
y = torch.empty(1 << 26, dtype=torch.float32, device="cuda")
cs, ks = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
pieces = 8
step = N // pieces
for mode in ("events", "host waits", "events"):
    out = []
    for i in range(14):
        t = time.perf_counter()
        evs = []
        with torch.cuda.stream(ks):
            for p in range(pieces):
                y.add_(1.0); y.add_(1.0)                 # ~0.2 ms of kernels per piece
                e = torch.cuda.Event(); e.record(ks); evs.append(e)
        for p in range(pieces):
            if mode == "events":
                cs.wait_event(evs[p])
            else:
                evs[p].synchronize()
            with torch.cuda.stream(cs):
                h[p * step:(p + 1) * step].copy_(x[p * step:(p + 1) * step], non_blocking=True)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t) * 1e3)
    print("%-10s ms per round:" % mode, " ".join("%.2f" % v for v in out), flush=True)
