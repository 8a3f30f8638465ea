#!/usr/bin/env python3
"""Timeline of ONE step from a rocprofv3 kernel trace of the shipped (multi-stream) configuration:
   python tools/step_timeline.py <trace dir> [step index from the end, default 1]
Prints every dispatch of the step with its queue, start and end relative to the step's first kernel, and the idle gap
before it on its queue; then busy / idle time per queue.  A step starts at the read_meta_kernel launch with the largest grid."""
import csv
import glob
import sys


def short(name):
    return name.replace("void ", "").split("(")[0].strip().replace("bf::", "")[:60]


def main():
    d = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    meta = [(i, int(r["Grid_Size_X"])) for i, r in enumerate(rows) if "read_meta_kernel" in r["Kernel_Name"]]
    big = max(g for _, g in meta)
    starts = [i for i, g in meta if g == big]
    a = starts[-back]
    b = starts[-back + 1] if back > 1 else len(rows)
    step = rows[a:b]
    t0 = int(step[0]["Start_Timestamp"])
    last_end = {}
    busy = {}
    print("%8s %8s %7s %6s  q  kernel" % ("start", "end", "dur", "gap"))
    for r in step:
        q = r.get("Queue_Id", "?")
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        gap = s - last_end.get(q, s)
        last_end[q] = e
        busy[q] = busy.get(q, 0.0) + (e - s)
        print("%8.1f %8.1f %7.1f %6.1f  %s  %s" % (s, e, e - s, gap, q, short(r["Kernel_Name"])))
    end = max(last_end.values())
    print("step: %.1f us of dispatches; per queue busy: %s" % (end, {q: round(v, 1) for q, v in busy.items()}))


if __name__ == "__main__":
    main()
