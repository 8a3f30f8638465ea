#!/usr/bin/env python3
"""Timing of `breakfast-hip filter` and `breakfast-hip annotate` (text_tools.cpp = filter.rs:39-156, annotate.rs:26-93;
SURVEY section 8f rows 3-4) on what `detect` writes for configs[1], beside the pure-Python restatements under oracle/
on a sample of the same rows.  Not collected by pytest (no test_ prefix); lives under tests/ because it runs the oracle.

  python tests/e2e_cli.py --runs 1 --no-oracle          # leaves /tmp/bf_e2e/gpu.sv (603 MB, 108k rows) behind
  python tests/text_tools_bench.py [--dir /tmp/bf_e2e] [--sample-rows 4000]

Host code only.  Prints one JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
EXE = os.path.join(ROOT, "breakfast_amd", "csrc", "breakfast-hip")


def timed(cmd, cwd, stdout_path=None):
    t = time.time()
    with open(stdout_path or os.devnull, "wb") as fo:
        p = subprocess.run(cmd, cwd=cwd, stdout=fo, stderr=subprocess.PIPE, timeout=3000)
    dt = time.time() - t
    if p.returncode != 0:
        raise SystemExit("%s failed: %s" % (" ".join(cmd), p.stderr.decode()[-400:]))
    return dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default="/tmp/bf_e2e")
    ap.add_argument("--sample-rows", type=int, default=4000)
    args = ap.parse_args()
    import annotate_oracle as ao
    import filter_oracle as fo
    d = args.dir
    sv = os.path.join(d, "gpu.sv")
    if not os.path.exists(sv):
        raise SystemExit("run tests/e2e_cli.py first: %s is missing" % sv)
    with open(sv, "rb") as f:
        header = f.readline()
        rows = f.read().split(b"\n")
    if rows and rows[-1] == b"":
        rows.pop()
    n_rows, sv_bytes = len(rows), os.path.getsize(sv)

    # ---- annotate: a BED of 30k features spread over the contigs the table names ------------------------------
    rng = np.random.default_rng(5)
    contigs = sorted({r.split(b"\t", 1)[0] for r in rows[:20000]})
    bed = []
    for i in range(30000):
        c = contigs[int(rng.integers(0, len(contigs)))]
        s = int(rng.integers(0, 240_000_000))
        bed.append(b"%s\t%d\t%d\tGENE%05d" % (c, s, s + int(rng.integers(500, 80000)), i))
    bed_path = os.path.join(d, "genes.bed")
    open(bed_path, "wb").write(b"\n".join(bed) + b"\n")
    ann_s = timed([EXE, "annotate", sv, bed_path], d, os.path.join(d, "annotated.sv"))
    k = min(args.sample_rows, n_rows)
    sample = os.path.join(d, "sample.sv")
    open(sample, "wb").write(header + b"\n".join(rows[:k]) + b"\n")
    t = time.time()
    want = ao.annotate(sample, bed_path)
    ann_oracle_s = time.time() - t
    timed([EXE, "annotate", sample, bed_path], d, os.path.join(d, "sample_annotated.sv"))
    ann_same = open(os.path.join(d, "sample_annotated.sv"), "rb").read() == want

    # ---- filter: the table as a tumour sample beside two controls that share a tenth of its rows ----------------
    cohort = os.path.join(d, "cohort")
    os.makedirs(cohort, exist_ok=True)

    def write_cohort(rows_t, tag):
        open(os.path.join(cohort, "tumor%s.sv" % tag), "wb").write(header + b"\n".join(rows_t) + b"\n")
        for c in range(2):
            open(os.path.join(cohort, "ctrl%d%s.sv" % (c, tag)), "wb").write(header + b"\n".join(rows_t[c::20]) + b"\n")
        sheet = "PATIENT\tSAMPLE\tTYPE\nP1\ttumor%s\ttumor\nP1\tctrl0%s\twbc\nP2\tctrl1%s\tgermline\n" % (tag, tag, tag)
        open(os.path.join(cohort, "sheet%s.tsv" % tag), "w").write(sheet)
    write_cohort(rows, "")
    os.makedirs(os.path.join(cohort, "annotated"), exist_ok=True)
    flt_s = timed([EXE, "filter", "sheet.tsv"], cohort)
    write_cohort(rows[:k], "_s")
    t = time.time()
    ref = fo.filter_cohort(os.path.join(cohort, "sheet_s.tsv"), cwd=cohort)
    flt_oracle_s = time.time() - t
    timed([EXE, "filter", "sheet_s.tsv"], cohort)
    flt_same = all(open(os.path.join(cohort, p), "rb").read() == want_b for p, want_b in ref.items()) and len(ref) == 3
    kept = sum(v.count(b"\n") - 1 for v in ref.values())

    print(json.dumps({
        "input": "%s: %d rows, %d bytes (detect's output for configs[1])" % (sv, n_rows, sv_bytes),
        "annotate": {"features": len(bed), "wall_s": round(ann_s, 3), "rows_per_s": round(n_rows / ann_s, 1),
                     "MB_per_s": round(sv_bytes / ann_s / 1e6, 1),
                     "oracle_python": {"rows": k, "wall_s": round(ann_oracle_s, 3), "rows_per_s": round(k / ann_oracle_s, 1),
                                       "identical_on_sample": bool(ann_same)}},
        "filter": {"samples": 3, "bytes_read": int(sv_bytes * 1.1), "wall_s": round(flt_s, 3),
                   "rows_per_s": round(n_rows * 1.1 / flt_s, 1), "MB_per_s": round(sv_bytes * 1.1 / flt_s / 1e6, 1),
                   "oracle_python": {"rows": int(k * 1.1), "wall_s": round(flt_oracle_s, 3),
                                     "rows_per_s": round(k * 1.1 / flt_oracle_s, 1), "identical_on_sample": bool(flt_same),
                                     "rows_kept_on_sample": kept}},
        "cores": 1, "note": "both tools are single-threaded host code, like the reference's"}))


if __name__ == "__main__":
    main()
