#!/usr/bin/env python3
"""The slow first steps after a hand-over (tools/step_series.py): do they pass with time alone (a pause after the upload)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from breakfast_amd import engine, synth

contigs, gk, spec = synth.config(1)
g = synth.make_genome(contigs, **gk)
b = engine.pack_batch(synth.make_reads(g, spec))
eng = engine.Engine(spec.anchor_len, 5000, 2)
eng.set_genome(g.names, g.seqs)
for pause in (0.0, 0.3, 0.0, 2.0):
    eng.reset()
    eng.submit(b); eng.submit_wait(); eng.sync()
    time.sleep(pause)
    out = []
    for i in range(12):
        t = time.perf_counter(); eng.finish_raw(); eng.sync(); out.append((time.perf_counter() - t) * 1e3)
    print("pause %.1f s after the hand-over, ms per step:" % pause, " ".join("%.2f" % x for x in out), flush=True)
