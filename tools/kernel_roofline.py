#!/usr/bin/env python3
"""Per-kernel roofline table of one step of the path (profiles/<tag>_kernel_roofline.md).

  python tools/kernel_roofline.py <tag> [bench.json]

Inputs, all made by tools/collect_profiles.sh <tag> on the headline workload (configs[1], 10M reads):
  gpurun_out/<tag>/trace1   rocprofv3 --kernel-trace of `bench.py --one-stream` (every kernel and copy of the library on
                            one stream, so a dispatch's duration is the kernel's own)
  gpurun_out/<tag>/pmc_fetch, pmc_write   the two PMC passes (FETCH_SIZE, WRITE_SIZE), same command with fewer steps
A "step" starts at a launch of read_meta_kernel over the whole resident input (its largest grid): everything before
the first one is the hand-over (per-piece launches) and is left out; all steps of the run are averaged.

Columns: launches and time per step; HBM bytes per step from the counters (FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024,
tools/pmc_traffic.py); the rate those bytes moved at and its share of the 8 TB/s peak; the bytes the kernel's job needs
at least ("useful": each input element read once and each output written once, DESIGN.md section 4 lists the formulas)
and the roofline fraction on THAT figure, which is what bench.py's `roofline.frac` is for the localise kernel.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 8000.0  # GB/s


def short(name):
    s = name.replace("void ", "").split("(")[0].strip()
    return s.replace("bf::", "")


def steps_of(rows, start_key, name_key, grid_key):
    """rows sorted by start; returns (first index of the first step, number of steps)"""
    meta = [(i, int(r[grid_key])) for i, r in enumerate(rows) if "read_meta_kernel" in r[name_key]]
    big = max(g for _, g in meta)
    starts = [i for i, g in meta if g == big]
    return starts[0], len(starts)


def load_trace(d):
    f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    first, n = steps_of(rows, "Start_Timestamp", "Kernel_Name", "Grid_Size_X")
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in rows[first:]:
        a = acc[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return {k: (v[0] / n, v[1] / n) for k, v in acc.items()}, n


def load_pmc(d, counter, scale):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    rows = sorted((r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter), key=lambda r: int(r["Start_Timestamp"]))
    first, n = steps_of(rows, "Start_Timestamp", "Kernel_Name", "Grid_Size")
    acc = collections.defaultdict(float)
    for r in rows[first:]:
        acc[short(r["Kernel_Name"])] += float(r["Counter_Value"]) * scale
    return {k: v / n for k, v in acc.items()}


def useful_bytes(n, m, s, c, L):
    """Least bytes per step each kernel's job needs (n evidence records, m member slots of qualifying clusters, s
    supporting reads, c reported clusters, L bases a read).  One line per kernel, the same formulas as DESIGN.md."""
    lb = (L + 1) // 2
    return {
        "localise_fast_kernel": n * (lb + 2 * ((L + 3) // 4) + 16 + 16 + (L + 7) // 8),   # SURVEY 8d B_A(L)
        "read_meta_kernel": n * (8 + 8 + 8 + 2),             # fragment id bytes + its offset in, hash + length out
        "gather_key_kernel": n * (4 + 8 + 8),                 # permutation in, key in, key out
        "gather_pos_kernel": n * (4 + 8 + 8),                 # permutation in, (lpos, rpos) in, both out
        "zip_pos_kernel": n * 16,
        "group_keys_kernel": n * (4 + 4 + 4 + 4),
        "head_mark_kernel": n * (8 + 4), "covered_kernel": n * (8 + 8 + 4 + 4), "tails_kernel": n * (8 + 4 + 8 + 4),
        "flag_ge_kernel": n * 8,
        "fill_members_kernel": m * (4 + 8 + 8 + 4) + s * 16,  # per member: index, info word, id hash in, slot out; per survivor a read row
        "emit_kernel": s * (4 + 8 + 16) + c * 56,
        "render_packed_kernel<3>": s * (4 + 8 + lb + 32 + 52),   # index, offset, the read's codes, its render record in; ~52 B record out
        "record_bytes_kernel": n * 8,
        # class partition + chunk sort: 20 B per record (comp 8 + {id, rpos, upper key bits} 12) in and out of every step
        "comp_build_kernel": n * (4 + 8 + 4 + 4 + 20),        # id, key, lpos, rpos in; the 20 B out
        "rs_upsweep_kernel<unsigned long, 8>": 2 * n * 8,     # two partition passes: the keys
        "rs_downsweep_kernel<unsigned long, RsV3, 8>": 2 * n * 40,
        "chunk_sort_kernel": n * (20 + 20),                   # partitioned records in; skey 8, lpos 4, rpos 4, perm 4 out
        "group_id_kernel": n * (8 + 4),
    }


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", tag)
    bench = json.load(open(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", tag + "_bench_cfg1.json")))
    n = bench["config"]["reads_per_gpu"]; L = bench["config"]["read_len"]
    s = bench["result"]["supporting_reads"]; c = bench["result"]["clusters"]
    times, n_steps = load_trace(os.path.join(src, "trace1"))
    fetch = load_pmc(os.path.join(src, "pmc_fetch"), "FETCH_SIZE", 1024 * 2)
    write = load_pmc(os.path.join(src, "pmc_write"), "WRITE_SIZE", 1024)
    m = int(s * 1.25)                                     # member slots of qualifying clusters before the dedup (measured once: 4.9M for 3.94M)
    useful = useful_bytes(n, m, s, c, L)
    out = os.path.join(ROOT, "profiles", tag + "_kernel_roofline.md")
    total_us = sum(v[1] for v in times.values())
    with open(out, "w") as o:
        o.write("# Per-kernel roofline of one step (%s): configs[1], %d reads x %d bp, reads as %s\n\n" % (tag, n, L, bench["config"].get("reads_as", "ASCII")))
        o.write("Made by tools/kernel_roofline.py from gpurun_out/%s (rocprofv3 kernel trace of `bench.py --one-stream`, %d steps averaged;\n"
                "PMC passes FETCH_SIZE and WRITE_SIZE).  Every kernel here is HBM-bound work (gather / scan / sort of bytes and integers);\n"
                "peak 8 TB/s, about 6.3 TB/s achievable.  Sum of kernel time per step: %.2f ms (on one stream nothing overlaps; the\n"
                "shipped configuration overlaps the result copy and the print-order sorts with the kernels).\n\n" % (tag, n_steps, total_us / 1e3))
        o.write("| kernel | launches / step | us / step | HBM MB / step (PMC) | moved at GB/s | % of 8 TB/s | useful MB / step | roofline frac on useful bytes |\n|---|---|---|---|---|---|---|---|\n")
        for k, (cnt, us) in sorted(times.items(), key=lambda kv: -kv[1][1]):
            if us < 3.0:
                continue
            hb = fetch.get(k, 0.0) + write.get(k, 0.0)
            rate = hb / (us * 1e-6) / 1e9 if us else 0
            u = useful.get(k, useful.get(k.split("<")[0]) if k.split("<")[0] not in ("rs_upsweep_kernel", "rs_downsweep_kernel") else None)
            o.write("| `%s` | %.1f | %.1f | %s | %s | %s | %s | %s |\n" % (
                k[:70], cnt, us, "%.1f" % (hb / 1e6) if hb else "-", "%.0f" % rate if hb else "-",
                "%.1f" % (100 * rate / PEAK) if hb else "-", "%.1f" % (u / 1e6) if u else "-",
                "%.3f" % (u / (us * 1e-6) / 1e9 / PEAK) if u else "-"))
    print(open(out).read())


if __name__ == "__main__":
    main()
