#!/usr/bin/env python3
"""Whole-BAM scan rate of the BAM front-end (`breakfast-hip anchors` = dispatch_reads_to_bowtie,
detect.rs:290-338): the stage the reference's only published number is about ("~40 million reads per
minute per CPU core", README.md:6 -- 0.67 M reads/s/core).  Host code, no GPU involved.

  python tools/bam_scan_bench.py [--records 4000000] [--threads 1,4,8] [--dir /tmp/bf_bam]

Writes a synthetic coordinate-free BAM (150 bp reads; 94 % fully aligned, 3 % aligned with a 30-base soft
clip, 3 % unaligned; binned base qualities so that the BGZF blocks inflate like real ones), scans it with
1..N inflate threads and prints one JSON line.  Nothing from oracle/ is used here.
"""
import argparse
import json
import os
import struct
import subprocess
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "breakfast_amd", "csrc", "breakfast-hip")
L = 150
QN = 24                      # read name bytes incl. NUL


def bgzf_block(data: bytes) -> bytes:
    c = zlib.compressobj(4, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


def records(rng, n, first):
    """n BAM records as one byte string (three record shapes, interleaved at random)."""
    kind = rng.choice(3, size=n, p=[0.94, 0.03, 0.03])              # 0 aligned 150M, 1 aligned 30S120M, 2 unaligned
    n_cig = np.array([1, 2, 0])[kind]
    size = 36 + QN + 4 * n_cig + (L + 1) // 2 + L                    # incl. the 4-byte block_size field
    off = np.zeros(n + 1, np.int64)
    off[1:] = np.cumsum(size)
    out = np.zeros(int(off[-1]), np.uint8)
    names = np.frombuffer(b"".join(b"HWI:7:%08d:%08d\x00" % (11, first + i) for i in range(n)), np.uint8).reshape(n, QN)
    seq = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    nib = np.array([1, 2, 4, 8], np.uint8)[seq]                      # A C G T in BAM's 4-bit code
    packed = (nib[:, 0::2] << 4) | nib[:, 1::2]
    qual = np.array([2, 12, 23, 37], np.uint8)[rng.choice(4, size=(n, L), p=[0.03, 0.07, 0.2, 0.7])]
    for k in range(3):
        sel = np.nonzero(kind == k)[0]
        m = sel.size
        if m == 0:
            continue
        sz = int(size[sel[0]])
        rows = np.zeros((m, sz), np.uint8)
        head = np.zeros(m, np.dtype([("bs", "<i4"), ("ref", "<i4"), ("pos", "<i4"), ("lrn", "u1"), ("mapq", "u1"), ("bin", "<u2"),
                                     ("nc", "<u2"), ("flag", "<u2"), ("lseq", "<i4"), ("nref", "<i4"), ("npos", "<i4"), ("tlen", "<i4")]))
        head["bs"] = sz - 4
        head["ref"] = -1 if k == 2 else 0
        head["pos"] = -1 if k == 2 else rng.integers(0, 1 << 27, size=m)
        head["lrn"] = QN
        head["mapq"] = 0 if k == 2 else 60
        head["bin"] = 4680
        head["nc"] = [1, 2, 0][k]
        head["flag"] = 4 if k == 2 else 0
        head["lseq"] = L
        head["nref"] = -1
        head["npos"] = -1
        rows[:, :36] = head.view(np.uint8).reshape(m, 36)
        rows[:, 36:36 + QN] = names[sel]
        p = 36 + QN
        if k == 0:
            rows[:, p:p + 4] = np.frombuffer(struct.pack("<I", (150 << 4) | 0), np.uint8)
        elif k == 1:
            rows[:, p:p + 8] = np.frombuffer(struct.pack("<II", (30 << 4) | 4, (120 << 4) | 0), np.uint8)
        p += 4 * [1, 2, 0][k]
        rows[:, p:p + (L + 1) // 2] = packed[sel]
        rows[:, p + (L + 1) // 2:] = qual[sel]
        out[(off[sel][:, None] + np.arange(sz)[None, :]).reshape(-1)] = rows.reshape(-1)
    return out.tobytes(), int((kind > 0).sum())


def write_bam(path, n, seed=7):
    rng = np.random.Generator(np.random.PCG64(seed))
    text = b"@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:chr1\tLN:248956422\n"
    head = b"BAM\x01" + struct.pack("<I", len(text)) + text + struct.pack("<I", 1) + struct.pack("<I", 5) + b"chr1\x00" + struct.pack("<I", 248956422)
    sent = 0
    raw_bytes = 0
    with open(path, "wb") as f, ThreadPoolExecutor(8) as ex:
        f.write(bgzf_block(head))
        step = 1 << 18
        for a in range(0, n, step):
            raw, s = records(rng, min(step, n - a), a)
            sent += s
            raw_bytes += len(raw)
            pieces = [raw[i:i + 0xff00] for i in range(0, len(raw), 0xff00)]
            for blk in ex.map(bgzf_block, pieces):
                f.write(blk)
        f.write(bgzf_block(b""))
    return sent, raw_bytes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=4_000_000)
    ap.add_argument("--threads", default="1,2,4,8")
    ap.add_argument("--dir", default="/tmp/bf_bam")
    args = ap.parse_args()
    os.makedirs(args.dir, exist_ok=True)
    path = os.path.join(args.dir, "scan.bam")
    t0 = time.time()
    sent, raw_bytes = write_bam(path, args.records)
    print("[bam] %d records (%d candidates), %.2f GB raw, %.2f GB BGZF in %.1fs" %
          (args.records, sent, raw_bytes / 1e9, os.path.getsize(path) / 1e9, time.time() - t0), file=sys.stderr, flush=True)
    runs = []
    for t in (int(x) for x in args.threads.split(",")):
        best = None
        for _ in range(2):
            with open(os.devnull, "wb") as null:
                t1 = time.time()
                p = subprocess.run([EXE, "anchors", "--anchor-len=30", "--threads=%d" % t, path], stdout=subprocess.PIPE if best is None else null,
                                   stderr=subprocess.PIPE, timeout=1800)
                dt = time.time() - t1
            if p.returncode != 0:
                raise SystemExit(p.stderr.decode())
            if best is None:
                assert p.stdout.count(b">5p:") == sent, "anchor count differs from the generator's"
            best = dt if best is None else min(best, dt)
        runs.append({"inflate_threads": t, "wall_s": round(best, 3), "records_per_s": round(args.records / best, 1),
                     "records_per_s_per_thread": round(args.records / best / (t + 2), 1)})
        print("[bam] threads=%d: %.2fs, %.2f M records/s" % (t, best, args.records / best / 1e6), file=sys.stderr, flush=True)
    print(json.dumps({"records": args.records, "candidates": sent, "bgzf_bytes": os.path.getsize(path), "raw_bytes": raw_bytes,
                      "host_cores": os.cpu_count(), "reference_published": "~40M reads/min/core = 0.67M reads/s/core (README.md:6)",
                      "note": "threads = inflate workers; one reader thread and one record-walking thread run beside them",
                      "runs": runs}))


if __name__ == "__main__":
    main()
