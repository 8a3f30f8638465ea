#!/usr/bin/env python3
"""Turns rocprofv3 --pmc passes into per-kernel HBM traffic per launch (profiles/*.json).

  python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [workload note]

FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes (TCC counter slots), both in KiB.
Per kernel only the launches with its largest grid count (the launches over the whole resident input).
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-byte requests at 64 bytes, so
the read side is doubled; calibrated in this repo on genome_pack_kernel (3.088 GB read: FETCH_SIZE
says 1.544 GB, TCC_EA0_RDREQ_128B x 128 B says 3.088 GB; every request of these kernels is a
128-byte one).  WRITE_SIZE is taken as is.
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    # a kernel that also ran on pieces of the input (the per-piece launches behind bf_submit's uploads) is reported
    # for its launches over the WHOLE input only: those with the kernel's largest grid
    out = {}
    for k, v in acc.items():
        g = max(x[0] for x in v)
        out[k] = [x[1] for x in v if x[0] == g]
    return out


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    rows = {}
    for k in fe:
        short = k.split("(")[0].replace("void ", "").strip()
        f = sum(fe[k]) / len(fe[k]) * 1024 * 2
        w = sum(wr.get(k, [0])) / max(len(wr.get(k, [0])), 1) * 1024
        rows[short] = {"launches": len(fe[k]), "fetch_bytes_per_launch": round(f), "write_bytes_per_launch": round(w),
                       "hbm_bytes_per_launch": round(f + w)}
    json.dump({"workload": note, "correction": "FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (separate passes)", "kernels": rows},
              open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
        print("%-60s %8.1f MB" % (k[:60], v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
