#!/usr/bin/env python3
"""Wall time of every single step of the headline workload after the hand-over (warm-up effects: which steps still grow buffers).
   python tools/step_series.py [n_steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from breakfast_amd import engine, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 14
contigs, gk, spec = synth.config(1)
g = synth.make_genome(contigs, **gk)
b = engine.pack_batch(synth.make_reads(g, spec))
eng = engine.Engine(spec.anchor_len, 5000, 2)
eng.set_genome(g.names, g.seqs)
eng.submit(b); eng.submit_wait(); eng.sync()
out = []
eng.kernel_times()
for i in range(n):
    if i == n // 2:
        time.sleep(1.0)                      # an idle second: do the first steps after it pay again?
    t = time.perf_counter(); eng.finish_raw(); eng.sync(); out.append((time.perf_counter() - t) * 1e3)
    kt = eng.kernel_times()
    print("step %2d: %.2f ms; kernels %.2f ms: %s" % (i, out[-1], sum(v[0] for v in kt.values()),
                                                     " ".join("%s %.3f" % (k, v[0]) for k, v in kt.items() if k in ("localise", "sort", "render", "d2h"))), flush=True)
print("ms per step:", " ".join("%.2f" % x for x in out))
print("device bytes:", eng.stats()[1])
