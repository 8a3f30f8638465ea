#!/usr/bin/env python3
"""The slow first steps of a ctx (tools/step_series.py) under other configurations of the cluster stage."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from breakfast_amd import engine, synth

contigs, gk, spec = synth.config(1)
g = synth.make_genome(contigs, **gk)
b = engine.pack_batch(synth.make_reads(g, spec))
configs = (("default", {}), ("one group", {"groups": 1}), ("six groups", {"groups": 6}), ("one stream", {"flags": engine.BF_FLAG_ONE_STREAM}),
           ("reads on host", {"flags": engine.BF_FLAG_READS_ON_HOST}))
if len(sys.argv) > 1 and sys.argv[1] == "same":
    configs = (("default", {}), ("default again", {}), ("and again", {}))
for label, kw in configs:
    t_create = time.perf_counter()
    eng = engine.Engine(spec.anchor_len, 5000, 2, **kw)
    eng.set_genome(g.names, g.seqs)
    t_genome = time.perf_counter()
    if label == "default again":
        time.sleep(0.5)                      # between the genome (whose raw bytes were just freed) and the reads
    eng.submit(b); eng.submit_wait(); eng.sync()
    t_submit = time.perf_counter()
    if label == "and again":
        time.sleep(0.5)
    out, fast_at = [], None
    for i in range(30):
        t = time.perf_counter(); eng.finish_raw(); eng.sync(); out.append((time.perf_counter() - t) * 1e3)
        if fast_at is None and i > 0 and out[-1] < 6.5:
            fast_at = (i, time.perf_counter() - t_create, time.perf_counter() - t_genome, time.perf_counter() - t_submit)
    kt = eng.kernel_times()
    print("%-14s ms per step:" % label, " ".join("%.2f" % x for x in out), flush=True)
    print("   first fast step %s: %.3f s after the ctx was created, %.3f s after the genome was in, %.3f s after the reads were in" % (fast_at or (-1, 0, 0, 0)), flush=True)
    eng.close()
