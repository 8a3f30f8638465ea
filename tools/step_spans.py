#!/usr/bin/env python3
"""Per step of a rocprofv3 kernel trace (shipped configuration): span from the step's first kernel to its last dispatch's
end, time in the copy kernels, time in all other kernels, idle time on the copy queue inside the span.
   python tools/step_spans.py <trace dir>"""
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
meta = [(i, int(r["Grid_Size_X"])) for i, r in enumerate(rows) if "read_meta_kernel" in r["Kernel_Name"]]
big = max(g for _, g in meta)
starts = [i for i, g in meta if g == big] + [len(rows)]
for k in range(len(starts) - 1):
    step = rows[starts[k]:starts[k + 1]]
    t0 = int(step[0]["Start_Timestamp"])
    end = max(int(r["End_Timestamp"]) for r in step)
    copies = [r for r in step if "copyBuffer" in r["Kernel_Name"]]
    big_copies = sorted(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in copies), reverse=True)[:6]
    ct = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in copies) / 1e3
    kt = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step if "copyBuffer" not in r["Kernel_Name"]) / 1e3
    first_copy = min((int(r["Start_Timestamp"]) for r in copies if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 100000), default=t0)
    print("step %2d: span %7.1f us, kernels %7.1f, copies %7.1f (n=%d; first long copy starts at %6.1f; longest %s)" %
          (k, (end - t0) / 1e3, kt, ct, len(copies), (first_copy - t0) / 1e3, " ".join("%.0f" % x for x in big_copies)))
