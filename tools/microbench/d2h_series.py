import time, torch
x = torch.empty(174_000_000, dtype=torch.uint8, device="cuda")
h = torch.empty(174_000_000, dtype=torch.uint8).pin_memory()
torch.cuda.synchronize()
out = []
for i in range(24):
    t = time.perf_counter(); h.copy_(x, non_blocking=True); torch.cuda.synchronize(); out.append((time.perf_counter() - t) * 1e3)
print("d2h ms:", " ".join("%.2f" % v for v in out))
# with kernels running beside the copies
y = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
s2 = torch.cuda.Stream()
out = []
for i in range(16):
    t = time.perf_counter()
    with torch.cuda.stream(s2):
        h.copy_(x, non_blocking=True)
    y.add_(1.0); y.add_(1.0)
    torch.cuda.synchronize(); out.append((time.perf_counter() - t) * 1e3)
print("d2h beside kernels ms:", " ".join("%.2f" % v for v in out))
