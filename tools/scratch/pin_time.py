import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
from breakfast_amd import engine
eng = engine.Engine(20, 5000, 2)
lib, h = eng.lib, eng.h
lib.bf_host_alloc.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
for mb in (1, 4, 16, 64):
    ps = []
    t = time.time()
    for i in range(16):
        p = C.c_void_p()
        assert lib.bf_host_alloc(h, mb << 20, C.byref(p)) == 0
        ps.append(p)
    dt = time.time() - t
    print("bf_host_alloc %3d MB x16: %.2f ms each, %.2f GB/s" % (mb, dt / 16 * 1e3, 16 * mb / 1024 / dt), flush=True)
    t = time.time()
    for p in ps: lib.bf_host_free(h, p)
    print("   free: %.2f ms each" % ((time.time() - t) / 16 * 1e3), flush=True)
sys.exit(0)
import threading
def work(n, mb, out):
    for i in range(n):
        p = C.c_void_p(); lib.bf_host_alloc(h, mb << 20, C.byref(p)); out.append(p)
