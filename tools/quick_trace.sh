#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of a short one-stream bench run -> gpurun_out/<tag>/trace1, and the per-kernel
# averages of the steps on stdout.   bash tools/quick_trace.sh <tag>
set -u
tag=$1; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace1 -o t --output-format csv -- python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-h2h --one-stream > $out/trace1.log 2>&1 || { tail -5 $out/trace1.log; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/trace1/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
meta = [(i, int(r["Grid_Size_X"])) for i, r in enumerate(rows) if "read_meta_kernel" in r["Kernel_Name"]]
big = max(g for _, g in meta); starts = [i for i, g in meta if g == big]
steps = len(starts); rows = rows[starts[0]:]
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"].replace("void ", "").split("(")[0].replace("bf::", "")[:70]
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = 0
for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if us / steps >= 3: print("%-72s %5.1f x %7.1f us = %8.1f us/step" % (n, c / steps, us / c, us / steps))
    if "copyBuffer" not in n: tot += us / steps
print("kernels without copies: %.1f us/step over %d steps" % (tot, steps))
PY
