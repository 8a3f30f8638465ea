#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats output (…_kernel_stats.csv) -> a short markdown table under profiles/.

  python tools/kernel_stats_md.py <dir with *_kernel_stats.csv> <out.md> "<title>" "<command>"
"""
import csv
import glob
import shutil
import sys


def main():
    d, out, title, cmd = sys.argv[1:5]
    f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[0]
    shutil.copy(f, out[:-3] + ".csv")
    rows = list(csv.DictReader(open(f)))
    with open(out, "w") as o:
        o.write("# %s\n\nCommand: `%s`\n\n| kernel | calls | avg us | total ms | %% |\n|---|---|---|---|---|\n" % (title, cmd))
        for r in rows[:40]:
            o.write("| `%s` | %s | %.1f | %.2f | %s |\n" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                            float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))


if __name__ == "__main__":
    main()
