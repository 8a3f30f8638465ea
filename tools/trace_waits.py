#!/usr/bin/env python3
"""BREAKFAST_TRACE=1 output (stderr of a run) -> per bf_finish call: wall time, number of host waits, time inside them,
the longest ones.   python tools/trace_waits.py <stderr file>"""
import re
import sys

pat = re.compile(r"\[bf-trace pid=\d+ rank=\d+/\d+ t=([0-9.]+)\] (\w+) (\S+) (\S+)")
calls, cur, open_wait = [], None, None
for line in open(sys.argv[1]):
    m = pat.match(line)
    if not m:
        continue
    t, what, name, where = float(m.group(1)), m.group(2), m.group(3), m.group(4)
    if what == "enter" and name == "bf_finish":
        cur = {"t0": t, "waits": []}
    elif what == "leave" and name == "bf_finish" and cur:
        cur["t1"] = t; calls.append(cur); cur = None
    elif what == "wait" and cur is not None:
        open_wait = (t, name, where)
    elif what == "done" and cur is not None and open_wait:
        cur["waits"].append(((t - open_wait[0]) * 1e6, t - cur["t0"], open_wait[1], open_wait[2])); open_wait = None
for i, c in enumerate(calls):
    w = c["waits"]
    top = sorted(w, reverse=True)[:4]
    print("call %2d: %.2f ms, %d waits, %.2f ms inside them; longest: %s" %
          (i, (c["t1"] - c["t0"]) * 1e3, len(w), sum(x[0] for x in w) / 1e3, "; ".join("%.0f us %s@%s at %.2f ms" % (x[0], x[2], x[3], x[1] * 1e3) for x in top)))
