#!/usr/bin/env python3
"""End-to-end run of the `breakfast-hip detect` command line on files, timed stage by stage, with the
.sv bytes compared against the CPU oracle's on the SAME files (test infrastructure: this script is the
only place outside bench.py's cpu_baseline where the oracle binary is executed on full-size inputs).

  python tests/e2e_cli.py [--config 1] [--scale 1.0] [--threads 16] [--dir /tmp/bf_e2e] [--no-oracle]

Writes <dir>/g.fa (the synthetic genome as FASTA) and <dir>/hits.txt (what Bowtie1 would print for the
anchors, detect.rs:61), runs
    breakfast-hip detect --anchor-len=A --anchor-hits=<dir>/hits.txt --timing unused.bam <dir>/g
and prints one JSON line: wall time and reads/s of the whole command, its TIMING stages, the oracle's wall
time on the same files and whether the two outputs are byte-identical.  SURVEY.md H8: the text side of
the boundary, not the GPU, bounds this rate; it is reported next to bench.py's device-pipeline rate and
is never bench.py's `value`.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from breakfast_amd import synth  # noqa: E402

EXE = os.path.join(ROOT, "breakfast_amd", "csrc", "breakfast-hip")
ORACLE = os.path.join(ROOT, "oracle", "detect_oracle")


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--dir", default="/tmp/bf_e2e")
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--runs", type=int, default=2, help="CLI runs (the first one pays cold file cache / first GPU use)")
    args = ap.parse_args()
    os.makedirs(args.dir, exist_ok=True)
    prefix = os.path.join(args.dir, "g")
    hits = os.path.join(args.dir, "hits.txt")

    t0 = time.time()
    contigs, gk, spec = synth.config(args.config, scale=args.scale, genome_scale=args.scale)
    if args.scale >= 0.05:
        spec.margin = 10000 + 2 * spec.read_len + 128
    g = synth.make_genome(contigs, **gk)
    with open(prefix + ".fa", "wb") as f:
        for i in range(len(g.names)):
            f.write(synth.Genome([g.names[i]], [g.seqs[i]]).fasta())
    b = synth.make_reads(g, spec)
    with open(hits, "wb") as f:
        step = 1 << 20
        for a in range(0, b.n, step):
            f.write(synth.bowtie_lines(b.slice(a, min(b.n, a + step)), g.names, first_read_num=1 + a))
    gen_s = time.time() - t0
    sizes = {"fasta_bytes": os.path.getsize(prefix + ".fa"), "hits_bytes": os.path.getsize(hits)}
    print("[e2e] inputs written in %.1fs: %s" % (gen_s, sizes), file=sys.stderr, flush=True)

    cmd = [EXE, "detect", "--anchor-len=%d" % spec.anchor_len, "--anchor-hits=" + hits, "--threads=%d" % args.threads,
           "--timing", "unused.bam", prefix]
    out_path = os.path.join(args.dir, "gpu.sv")
    image = prefix + ".bf2bit"
    if os.path.exists(image):
        os.unlink(image)

    def clocks(label):
        runs = []
        for r in range(args.runs):
            # the command returns when its output is complete; the process that did the work frees its GPU context and
            # pinned memory after that, in the background: let it finish, so that each run is a single invocation
            time.sleep(1.0)
            with open(out_path + ".new", "wb") as fo:
                t = time.time()
                p = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE, timeout=900)
                wall = time.time() - t
            if p.returncode != 0:
                sys.stderr.write(p.stderr.decode(errors="replace"))
                raise SystemExit("breakfast-hip failed with status %d" % p.returncode)
            stages = {}
            for line in p.stderr.decode().splitlines():
                if line.startswith("TIMING\t"):
                    _, name, sec = line.split("\t")
                    stages[name.strip()] = float(sec)
            runs.append({"wall_s": round(wall, 3), "reads_per_s": round(b.n / wall, 1), "stages_s": stages})
            print("[e2e] %s, run %d: %.2fs  %s" % (label, r, wall, stages), file=sys.stderr, flush=True)
        # the second clock: the same command as ONE process (BREAKFAST_HIP_NO_DETACH=1), i.e. until the GPU context, the device
        # memory and the pinned buffers are released as well -- what the next command on that GPU waits for
        held = []
        for r in range(min(args.runs, 3)):
            time.sleep(1.0)
            with open(out_path + ".held", "wb") as fo:
                t = time.time()
                p = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE, timeout=900, env=dict(os.environ, BREAKFAST_HIP_NO_DETACH="1"))
                wall = time.time() - t
            if p.returncode != 0:
                sys.stderr.write(p.stderr.decode(errors="replace"))
                raise SystemExit("breakfast-hip (one process) failed with status %d" % p.returncode)
            held.append(round(wall, 3))
            print("[e2e] %s, one process, run %d: %.2fs" % (label, r, wall), file=sys.stderr, flush=True)
        if sha(out_path + ".held") != sha(out_path + ".new"):
            raise SystemExit("e2e: the one-process run wrote other bytes")
        os.unlink(out_path + ".held")
        return runs, held

    runs, held = clocks("from the FASTA")
    os.replace(out_path + ".new", out_path)
    # once more with the genome as a packed image beside the FASTA (`breakfast-hip index`, made once like Bowtie's index)
    t = time.time()
    p = subprocess.run([EXE, "index", "--threads=%d" % args.threads, prefix], capture_output=True, timeout=900)
    index_s = time.time() - t
    if p.returncode != 0:
        sys.stderr.write(p.stderr.decode(errors="replace"))
        raise SystemExit("breakfast-hip index failed with status %d" % p.returncode)
    print("[e2e] index: %.2fs, %d bytes" % (index_s, os.path.getsize(image)), file=sys.stderr, flush=True)
    iruns, iheld = clocks("from the image")
    if sha(out_path + ".new") != sha(out_path):
        raise SystemExit("e2e: the run from the genome image wrote other bytes")
    os.unlink(out_path + ".new")
    if not all("image_open" in x["stages_s"] for x in iruns):
        raise SystemExit("e2e: the genome image was not used")
    result = {"workload": "configs[%d] x %.4g: %d reads x %d bp, anchor %d, genome %.3f Gb" %
                          (args.config, args.scale, b.n, spec.read_len, spec.anchor_len, sum(l for _, l in contigs) / 1e9),
              "threads": args.threads, **sizes, "runs": runs,
              "clocks": {"by_the_caller_s": min(x["wall_s"] for x in runs[1:] or runs),
                         "until_the_gpu_is_released_s": min(held[1:] or held),
                         "note": "the command returns when its output is complete (a child process did the work and frees its GPU "
                                 "context, 10 GB of device memory and the pinned buffers behind the caller's back); the second "
                                 "figure is the same command as one process"},
              "with_genome_image": {"index_s": round(index_s, 3), "image_bytes": os.path.getsize(image), "runs": iruns,
                                    "clocks": {"by_the_caller_s": min(x["wall_s"] for x in iruns[1:] or iruns),
                                               "until_the_gpu_is_released_s": min(iheld[1:] or iheld)},
                                    "note": "`breakfast-hip index` wrote <genome>.bf2bit once; detect hands it to the GPU instead of reading the FASTA; same .sv bytes"},
              "sv_bytes": os.path.getsize(out_path), "sv_sha256": sha(out_path)}

    if not args.no_oracle:
        ref_path = os.path.join(args.dir, "oracle.sv")
        with open(ref_path, "wb") as fo:
            t = time.time()
            p = subprocess.run([ORACLE, "--anchor-len=%d" % spec.anchor_len, hits, prefix], stdout=fo,
                               stderr=subprocess.PIPE, timeout=3000)
            owall = time.time() - t
        if p.returncode != 0:
            sys.stderr.write(p.stderr.decode(errors="replace"))
            raise SystemExit("oracle failed with status %d" % p.returncode)
        same = sha(ref_path) == result["sv_sha256"] and os.path.getsize(ref_path) == result["sv_bytes"]
        result["oracle"] = {"wall_s": round(owall, 3), "reads_per_s": round(b.n / owall, 1), "cores": 1,
                            "identical_sv": bool(same)}
        print("[e2e] oracle: %.2fs identical=%s" % (owall, same), file=sys.stderr, flush=True)
    print(json.dumps(result), flush=True)
    if not args.no_oracle and not result["oracle"]["identical_sv"]:
        raise SystemExit("e2e: GPU .sv differs from the oracle's")


if __name__ == "__main__":
    main()
