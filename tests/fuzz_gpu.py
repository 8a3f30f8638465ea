#!/usr/bin/env python3
"""Random mid-size cases on the GPU against the C++ restatement (oracle/), for a time budget.

The parity tests hold hand-made edge cases and small random ones; this walks the space between those and the full-size
runs: 2k-400k reads on genomes of 1-60 contigs, plain or built from repeated units (equal windows at many places: ties
in the split scan, signatures shared by far-apart junctions), N runs and soft-masked runs, A / max_frag_len /
min_evidence at odd values, piles on hot junctions, mates, substitutions and N in the reads; every case through the
ways the library can be driven (reads as ASCII, 4-bit, 2-bit; one go and contig groups; the exact path; reads kept on
the host; two or three ranks over the host transport).  The whole .sv text must be the oracle's, byte for byte.

    python3 tests/fuzz_gpu.py --minutes 8 --seed0 0 --log gpurun_out/fuzz/log.txt

A checker run by hand, not collected by pytest (it lives in tests/ because only tests may use the oracle): it prints one line per case and the failing (seed, way) pairs at the end; exit status 1 if any.
"""
import argparse
import os
import shutil
import sys
import tempfile
import threading
import time
import uuid

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from breakfast_amd import engine, synth  # noqa: E402
import oracle_lib as ol  # noqa: E402


def repeat_genome(rng, contigs, seed, unit_len, n_units, mut):
    """Contigs tiled from a few random units with sparse point mutations: the same windows at many places."""
    acgt = np.frombuffer(b"ACGT", np.uint8)
    units = [acgt[rng.integers(0, 4, size=unit_len)] for _ in range(n_units)]
    seqs = []
    for _, length in contigs:
        parts, have = [], 0
        while have < length:
            u = units[int(rng.integers(0, n_units))]
            parts.append(u)
            have += u.size
        s = np.concatenate(parts)[:length].copy()
        hit = rng.random(length) < mut
        s[hit] = acgt[rng.integers(0, 4, size=int(hit.sum()))]
        seqs.append(s)
    return synth.Genome([n.encode() for n, _ in contigs], seqs)


def make_case(seed, dense=False, large=False):
    rng = np.random.default_rng(1_000_003 * seed + 17)
    n_contigs = int(rng.choice([1, 2, 3, 5, 12, 25, 60]))
    order = rng.permutation(200)[:n_contigs]
    total = int(rng.choice([40_000, 300_000, 3_000_000, 20_000_000]))
    if large:
        total = int(rng.choice([20_000_000, 60_000_000, 200_000_000]))
    w = rng.random(n_contigs) + 0.05
    A = int(rng.choice([10, 12, 16, 20, 25, 30]))
    L = int(rng.choice([2 * A, 2 * A + 1, 2 * A + 7, 76, 100, 150, 250]))
    L = max(L, 2 * A)
    contigs = [("chr%d" % order[k] if k % 3 else "scaf_%d" % order[k], max(int(total * w[k] / w.sum()), 3 * L + 400)) for k in range(n_contigs)]
    kind = rng.choice(["plain", "plain", "repeats", "masked"])
    end_n = int(rng.choice([0, 0, 13, 1000])) if kind != "repeats" else 0
    if kind == "repeats":
        g = repeat_genome(rng, contigs, seed, int(rng.choice([37, 200, 1500])), int(rng.choice([1, 3, 20])), float(rng.choice([0.0, 0.002, 0.03])))
    else:
        n_runs, lower_runs = [], []
        for _ in range(int(rng.integers(0, 6))):
            ci = int(rng.integers(0, n_contigs))
            a = int(rng.integers(0, contigs[ci][1] - 50))
            n_runs.append((ci, a, int(rng.integers(1, min(5000, contigs[ci][1] - a)))))
        if kind == "masked":
            for _ in range(int(rng.integers(1, 40))):
                ci = int(rng.integers(0, n_contigs))
                a = int(rng.integers(0, contigs[ci][1] - 50))
                lower_runs.append((ci, a, int(rng.integers(1, min(contigs[ci][1] // 2 + 2, contigs[ci][1] - a)))))
        g = synth.make_genome(contigs, seed=77_000 + seed, end_n=end_n, n_runs=n_runs, lower_runs=lower_runs)
    n_reads = int(rng.choice([2_000, 10_000, 40_000, 120_000, 400_000]))
    if large:
        n_reads = int(rng.choice([1_000_000, 2_000_000, 4_500_000, 9_000_000]))
    hot = float(rng.choice([0.0, 0.0, 0.3, 0.8]))
    hot_events = int(rng.choice([1, 3, 40])) if hot else 0
    n_events = int(rng.choice([1, 5, 50, max(n_reads // 60, 1), max(n_reads // 8, 1)]))
    if large:
        n_events = int(rng.choice([max(n_reads // 300, 1), max(n_reads // 60, 1), max(n_reads // 8, 1)]))
    single = float(rng.choice([0.0, 0.2, 0.5, 0.9]))
    if large and hot:
        hot, hot_events = 0.02, 40
    single = min(single, 1.0 - hot)
    # the reference's sweep is quadratic in the reads of one junction (so is the restatement): n_reads x (reads of a pile) stays below 1e9
    pile = max(hot / max(hot_events, 1), max(1.0 - single - hot, 0.0) / n_events)
    n_reads = max(min(n_reads, int((1e9 / max(pile, 1e-9)) ** 0.5)), 500)
    F = int(rng.choice([0, 1, 30, 300, 5000, 5000, 100_000, 2 ** 31 - 1]))
    E = int(rng.choice([1, 2, 2, 3, 7]))
    if not dense:                                    # ... and in the reads of one class within max_frag_len of each other
        n_reads = max(min(n_reads, int((1e9 / min(1.0, (F + 1.0) * n_contigs / total)) ** 0.5)), 500)
    spec = synth.ReadSpec(n_reads=n_reads, read_len=L, anchor_len=A, n_events=n_events,
                          singleton_frac=single, mate_frac=float(rng.choice([0.0, 0.3, 0.9])),
                          subst_rate=float(rng.choice([0.0, 0.005, 0.05])), n_rate=float(rng.choice([0.0, 0.0, 0.01])),
                          deletion_frac=float(rng.choice([0.0, 0.5, 1.0])), hot_frac=hot, hot_events=hot_events,
                          seed=500_000 + seed, margin=int(rng.choice([0, 0, 5000])))
    b = synth.make_reads(g, spec)
    what = "contigs=%d bases=%d %s end_n=%d reads=%d L=%d A=%d F=%d E=%d events=%d single=%.1f mate=%.1f subst=%.3f n=%.2f del=%.1f hot=%.1f/%d" % (
        n_contigs, sum(c[1] for c in contigs), kind, end_n, b.n, L, A, F, E, spec.n_events, spec.singleton_frac, spec.mate_frac,
        spec.subst_rate, spec.n_rate, spec.deletion_frac, hot, spec.hot_events)
    return g, b, A, F, E, what


def one_gpu(g, b, A, F, E, flags, reads_as):
    eng = engine.Engine(A, F, E, flags=flags)
    try:
        eng.set_genome(g.names, g.seqs)
        batch = b
        if reads_as in ("packed", "packed2") and bool((b.seq_bytes & 0x20).any()):
            reads_as = "pieces"                       # (reads cut from soft-masked bases: BAM's 4-bit codes have no lower case)
        if reads_as in ("packed", "packed2"):
            batch = engine.pack_batch(b)
            if reads_as == "packed2":
                batch = engine.pack2_batch(batch)
        cut = b.n // 3
        if reads_as == "pieces" and cut:
            eng.submit(b.slice(0, cut))
            eng.submit(b.slice(cut, b.n))
        else:
            eng.submit(batch)
        res = eng.finish(reads=False)
        return eng.format_sv(res)                     # (BF_FLAG_READS_ON_HOST: `batch` is still alive here, as the flag asks)
    finally:
        eng.close()


def over_ranks(g, b, A, F, E, world, flags, slot_bytes=8 << 20, cuts=None):
    cuts = cuts or [b.n * r // world for r in range(world + 1)]
    name = "/bf_fuzz_%s" % uuid.uuid4().hex[:12]
    results, errors = [None] * world, []

    def work(rank):
        try:
            eng = engine.Engine(A, F, E, device=0, flags=flags)
            try:
                eng.set_genome(g.names, g.seqs)
                eng.shard_init_host(rank, world, cuts[rank], name, slot_bytes)
                if cuts[rank + 1] > cuts[rank]:
                    eng.submit(b.slice(cuts[rank], cuts[rank + 1]))
                results[rank] = eng.shard_finish()
            finally:
                eng.close()
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=250)
    if errors or any(x is None for x in results):
        raise RuntimeError("ranks: %r" % (errors or "a rank did not finish",))
    return engine.shard_format_sv(g.names, results)


EXE = os.path.join(ROOT, "breakfast_amd", "csrc", "breakfast-hip")


def command_line(g, b, A, F, E, tmp, extra, image=False):
    """`breakfast-hip detect --anchor-hits=...`: from a FASTA (or its image) and bowtie's lines on disk to stdout."""
    import subprocess
    prefix = os.path.join(tmp, "g")
    if not os.path.exists(prefix + ".fa"):
        open(prefix + ".fa", "wb").write(g.fasta())
        open(os.path.join(tmp, "hits.txt"), "wb").write(synth.bowtie_lines(b, g.names))
    if image and not os.path.exists(prefix + ".bf2bit"):
        subprocess.run([EXE, "index", prefix], check=True, capture_output=True, timeout=100)
    env = dict(os.environ)
    if not image:
        env["BREAKFAST_HIP_NO_IMAGE"] = "1"
    if bool((b.seq_bytes & 0x20).any()):
        extra = extra + ["--ascii-reads"]
    p = subprocess.run([EXE, "detect", "--anchor-len=%d" % A, "--max-frag-len=%d" % F, "--min-evidence=%d" % E,
                        "--anchor-hits=" + os.path.join(tmp, "hits.txt")] + extra + ["unused.bam", prefix],
                       capture_output=True, timeout=200, env=env)
    if p.returncode != 0:
        raise RuntimeError("exit %d: %s" % (p.returncode, p.stderr.decode(errors="replace")[-300:]))
    return p.stdout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--seed0", type=int, default=0)
    ap.add_argument("--seeds", type=int, default=10 ** 9)
    ap.add_argument("--log", default=None)
    ap.add_argument("--large", action="store_true", help="1M - 4.5M reads on 20M - 200M bases, four ways a case")
    ap.add_argument("--stress", action="store_true", help="the rarely taken branches: few class buckets, a weak dedup hash, the sorted-run dedup")
    ap.add_argument("--ranks", action="store_true", help="the sharded path only: three random (world, shard cuts, slot size, flags) a case")
    ap.add_argument("--cli", action="store_true", help="also through the command line (FASTA, genome image, two ranks on one GPU)")
    ap.add_argument("--time-dense", type=int, default=None, metavar="SEED",
                    help="no oracle: time the GPU alone on this seed's case WITHOUT the bound on reads per window (the reference is "
                         "quadratic there; this shows what the GPU path does)")
    args = ap.parse_args()
    out = open(args.log, "a") if args.log else None

    def say(*a):
        line = " ".join(str(x) for x in a)
        print(line, flush=True)
        if out:
            out.write(line + "\n")
            out.flush()

    if args.time_dense is not None:
        g, b, A, F, E, what = make_case(args.time_dense, dense=True)
        say("dense case", what)
        for name, flags in (("one go", engine.BF_FLAG_ONE_GROUP), ("groups", engine.BF_FLAG_GROUPS_ALWAYS)):
            t0 = time.time()
            sv = one_gpu(g, b, A, F, E, flags, "ascii")
            say("  %s: %.2f s, %d bytes of .sv" % (name, time.time() - t0, len(sv)))
        return 0
    G, X, H = engine.BF_FLAG_GROUPS_ALWAYS, engine.BF_FLAG_FORCE_EXACT, engine.BF_FLAG_READS_ON_HOST
    t_end = time.time() + args.minutes * 60
    bad, cases, seed = [], 0, args.seed0
    while time.time() < t_end and seed < args.seed0 + args.seeds:
        try:
            g, b, A, F, E, what = make_case(seed, large=args.large)
        except Exception as e:  # noqa: BLE001   (a spec the generator cannot place: not a finding)
            say("seed", seed, "generator:", repr(e)[:200])
            seed += 1
            continue
        say("seed", seed, "...", what)
        t0 = time.time()
        ref = ol.detect_batch(ol.OracleGenome(g.names, g.seqs), g.names, b, A, F, E)
        t_or = time.time() - t0
        ways = [("ascii", lambda: one_gpu(g, b, A, F, E, 0, "ascii")),
                ("packed+groups", lambda: one_gpu(g, b, A, F, E, G, "packed")),
                ("packed2", lambda: one_gpu(g, b, A, F, E, 0, "packed2")),
                ("exact+groups+pieces", lambda: one_gpu(g, b, A, F, E, X | G, "pieces")),
                ("reads_on_host+groups", lambda: one_gpu(g, b, A, F, E, H | G, "packed")),
                ("ranks2", lambda: over_ranks(g, b, A, F, E, 2, 0)),
                ("ranks3+groups", lambda: over_ranks(g, b, A, F, E, 3, G))]
        if args.large:
            ways = [("packed2", ways[2][1]), ("packed, as the library cuts it", lambda: one_gpu(g, b, A, F, E, 0, "packed")),
                    ("reads_on_host+groups", ways[4][1]), ("ranks2", ways[5][1])]
        if args.stress:
            # the rarely taken branches: few buckets (chunks overflow their LDS sort and are set aside), a 3-bit dedup hash (every
            # group decided on the id bytes), the sorted-run dedup for every cluster
            Wk, Sd = engine.BF_FLAG_WEAK_DEDUP_HASH, engine.BF_FLAG_SORTED_DEDUP

            def with_bits(bits, fn):
                os.environ["BREAKFAST_HIP_BUCKET_BITS"] = str(bits)
                try:
                    return fn()
                finally:
                    del os.environ["BREAKFAST_HIP_BUCKET_BITS"]
            ways = [("weak hash", lambda: one_gpu(g, b, A, F, E, Wk, "packed")),
                    ("weak hash+groups", lambda: one_gpu(g, b, A, F, E, Wk | G, "ascii")),
                    ("sorted dedup", lambda: one_gpu(g, b, A, F, E, Sd, "packed2")),
                    ("sorted dedup+weak+groups", lambda: one_gpu(g, b, A, F, E, Sd | Wk | G, "pieces")),
                    ("2 bucket bits", lambda: with_bits(2, lambda: one_gpu(g, b, A, F, E, 0, "packed"))),
                    ("5 bucket bits+groups", lambda: with_bits(5, lambda: one_gpu(g, b, A, F, E, G, "ascii"))),
                    ("1 bucket bit, ranks3, weak", lambda: with_bits(1, lambda: over_ranks(g, b, A, F, E, 3, Wk)))]
        if args.ranks:
            # the sharded path alone: 2 - 8 ranks, uneven and empty shards, slots from 64 KB (many rounds of the host transport) up
            r2 = np.random.default_rng(seed)
            ways = []
            for _ in range(3):
                world = int(r2.choice([2, 3, 4, 5, 8]))
                inner = np.sort(r2.integers(0, b.n + 1, size=world - 1)) if r2.random() < 0.5 else np.array([b.n * r // world for r in range(1, world)])
                cuts = [0] + [int(x) for x in inner] + [b.n]
                slot = int(r2.choice([1 << 16, 1 << 20, 8 << 20]))
                fl = int(r2.choice([0, G, X, G | X]))
                ways.append(("ranks%d slot=%d flags=%d cuts=%s" % (world, slot, fl, cuts),
                             lambda world=world, fl=fl, slot=slot, cuts=cuts: over_ranks(g, b, A, F, E, world, fl, slot, cuts)))
        if args.cli:
            tmp = tempfile.mkdtemp(prefix="bf_fuzz_")
            ways += [("cli", lambda: command_line(g, b, A, F, E, tmp, ["--threads=%d" % (1 + seed % 7), "--block-bytes=%d" % (1000 + 37 * (seed % 5000))])),
                     ("cli image", lambda: command_line(g, b, A, F, E, tmp, [], image=True)),
                     ("cli devices=0,0", lambda: command_line(g, b, A, F, E, tmp, ["--devices=0,0"], image=bool(seed & 1)))]
        t0 = time.time()
        failed = []
        for name, fn in ways:
            try:
                got = fn()
                if got != ref.sv:
                    failed.append(name + ":differs")
            except Exception as e:  # noqa: BLE001
                failed.append(name + ":" + repr(e)[:160])
        say("seed", seed, "ok" if not failed else "FAILED " + "; ".join(failed), "| clusters=%d sv=%d bytes | oracle %.1fs gpu %.1fs |" % (
            ref.n_clusters, len(ref.sv), t_or, time.time() - t0), what)
        if args.cli:
            shutil.rmtree(tmp, ignore_errors=True)
        if failed:
            bad.append((seed, failed))
        cases += 1
        seed += 1
    say("%d cases, %d failed: %s" % (cases, len(bad), [s for s, _ in bad]))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
