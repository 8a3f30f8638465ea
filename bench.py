#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, one JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config I] [--scale S]

metric   candidate reads/s breakpoint-localised + deduped (whole job, all ranks)
workload configs[1]: 10M unaligned 2x150 bp reads with anchor hits on an hg38-sized genome,
         anchor length 20 (per GPU: weak scaling -- with N GPUs every rank holds its own 10M-read
         input range).  Synthetic, seeded (no genome/BAM exists in the image).
step     one pass of the hot path over the resident batch: encode + localise (+ exact path) ->
         [N>1: pack, all-to-all, import] -> sort -> sweep -> dedup -> order -> junction strings ->
         cluster table in host memory.  Inputs (ASCII reads, fragment ids, hit records, packed
         genome) are resident in HBM before the timed region starts: that is `value`, as the round's
         measurement contract words it.
host_to_host  SURVEY.md section 8d's own wording of the metric -- parsed anchor-hit batches in HOST memory in,
         cluster table in host memory out -- measured in the same run (one GPU): every step is bf_reset +
         bf_submit of the whole batch (asynchronous, piece by piece, the per-read kernels queued behind every
         uploaded piece) + bf_finish.  Reported for batches in bf_host_alloc (pinned) memory, which is what a
         parser that writes into the library's buffers holds, and for ordinary pageable arrays (staged through
         pinned slots by host threads).  PCIe bounds both; `value` never includes them.
roofline the dominant kernel, `localise`: algorithmic bytes/read (SURVEY.md section 8d,
         B_A(L) = ceil(L/2) + 2*ceil(L/4) + 16 + 16 + ceil(L/8) = 202 B at L=150) x reads / its average
         duration measured with HIP events on the library's stream, against 8 TB/s HBM3E.
cpu_baseline  the C++ restatement of the reference (oracle/, single thread like detect.rs:76-241)
         timed on a bounded sample of the same batch on this host's cores.
--gpus N  one process per GPU.  Under torchrun (WORLD_SIZE set) this process is one rank; without it the command
         starts its N ranks itself, as fresh processes of a parent that never touches the GPU, and exits non-zero when
         a rank fails -- it never measures fewer GPUs than it was asked for.  Every rank runs the library's sharded
         path (bf_shard_finish over RCCL).  configs[1] is weak scaling (10M reads per rank); --config 2 is BASELINE's
         strong-scaling case: the 100M reads are 16 blocks over one junction table (synth.BLOCKS) and rank r of N holds
         the contiguous blocks [16r/N, 16(r+1)/N), whatever N is ("scaling": "strong").
--check  the GPU's table against the oracle's on a sample (one GPU), and on the sharded path the merged table of all
         ranks (bf_shard_format_sv) against the one-GPU table of the same reads (SHA-256).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from breakfast_amd import engine, launch, synth  # noqa: E402
from breakfast_amd.launch import bind_to_gpu_numa_node, init_from_env  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def algorithmic_bytes_per_read(L: int) -> int:
    """SURVEY.md section 8d, split-scan stage."""
    return math.ceil(L / 2) + 2 * math.ceil(L / 4) + 16 + 16 + math.ceil(L / 8)


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def claim_stdout():
    """Keeps stdout for the ONE JSON line: file descriptor 1 is pointed at stderr for everything else
    (RCCL prints a version banner on stdout when a communicator is created)."""
    sys.stdout.flush()
    keep = os.dup(1)
    os.dup2(2, 1)
    return os.fdopen(keep, "w")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=1, help="index into BASELINE.json configs (default 1)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink reads AND genome (smoke runs only)")
    ap.add_argument("--cpu-sample", type=int, default=6_000_000, help="reads in the CPU baseline sample (0 = skip; the default is about 11 s of one core)")
    ap.add_argument("--variant", choices=["fused", "exact"], default="fused",
                    help="the localise stage: the packed kernel (shipped), or every read through the byte-exact kernel (diagnostic)")
    ap.add_argument("--exchange", action="store_true",
                    help="with one GPU: run the sharded code path anyway (bf_shard_finish over RCCL with one rank: pack -> "
                         "all-to-all with itself -> import -> cluster -> request lists -> source-side records), to time "
                         "what the exchange adds per rank")
    ap.add_argument("--groups", type=int, default=0, help="contig groups of the cluster stage (0 = library default, 1 = one go)")
    ap.add_argument("--soft-mask", type=float, default=0.0,
                    help="fraction of the genome in lowercase (soft-masked repeats, as in UCSC hg38.fa), in runs of 300-30000 "
                         "bases; 0 = the all-uppercase genome of BASELINE's config (diagnostic for the byte-exact path)")
    ap.add_argument("--check", action="store_true", help="also compare the GPU .sv with the oracle's on the sample")
    ap.add_argument("--one-stream", action="store_true", help="profiling: all kernels and copies of the library on one stream "
                                                               "(a kernel trace then shows each kernel's own duration)")
    ap.add_argument("--reads", choices=["packed", "ascii"], default="packed",
                    help="how the reads are handed over and kept resident: BAM 4-bit base codes (bf_submit_packed: what the "
                         "command line's parser produces, the shipped path) or ASCII (bf_submit)")
    ap.add_argument("--roh-groups", type=int, default=-1, help="contig groups of the reads-on-host leg (-1: like --groups)")
    ap.add_argument("--no-h2h", action="store_true", help="skip the host-to-host loops (resident-input value only)")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from the committed PMC passes instead of two live rocprofv3 passes")
    ap.add_argument("--host-threads", type=int, default=0, help="staging threads of bf_submit (0 = library default)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher above us: start the ranks here.  Nothing in this process has touched the GPU (no HIP call, no
        # torch.cuda.is_available()), and it never will: the ranks are children, this process only waits for them.
        raise SystemExit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    json_out = claim_stdout()
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %s ranks (WORLD_SIZE); refusing to measure another "
                         "number of GPUs than asked for" % (args.gpus, os.environ.get("WORLD_SIZE")))
    rank, world, local = init_from_env()
    forced = args.exchange and world == 1
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE); refusing to measure another "
                         "number of GPUs than asked for" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    if local >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d wants GPU %d but this machine has %d" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    node = bind_to_gpu_numa_node(local) if world > 1 else None
    if node is not None:
        print("[bench] rank %d: GPU %d, CPUs of NUMA node %d" % (rank, local, node), file=sys.stderr, flush=True)

    # ---- synthetic workload ----------------------------------------------------------------------
    t0 = time.time()
    contigs, gk, spec = synth.config(args.config, scale=args.scale, genome_scale=args.scale)
    L, A = spec.read_len, spec.anchor_len
    if args.scale >= 0.05:
        spec.margin = 10000 + 2 * L + 128            # junctions clear of the 10 kb N runs at contig ends
    genome = synth.make_genome(contigs, **gk)
    if args.soft_mask > 0:
        mrng = np.random.Generator(np.random.PCG64(spec.seed + 99))
        for sq in genome.seqs:
            n_runs = max(1, int(sq.shape[0] * args.soft_mask / 3000))
            st = mrng.integers(0, max(sq.shape[0] - 30000, 1), size=n_runs)
            ln = (300 * np.exp(mrng.random(n_runs) * np.log(100))).astype(np.int64)      # 300 .. 30000, log-uniform
            mask = np.zeros(sq.shape[0] + 1, np.int32)
            np.add.at(mask, st, 1)
            np.add.at(mask, np.minimum(st + ln, sq.shape[0]), -1)
            sq[np.cumsum(mask[:-1]) > 0] |= 0x20
    log(rank, "genome: %d contigs, %.3f Gb in %.1fs" % (len(contigs), sum(l for _, l in contigs) / 1e9, time.time() - t0))
    # The genome goes into HBM first, as in detect.rs (:48-56 load it before the first read is looked at), and the reads are
    # made after that.  Not only for the order's sake: bf_genome_end releases the 3.1 GB of raw bytes the pack read, the
    # driver clears released device memory in the background, and for the next 0.1-0.2 s device-to-host copies run at 41
    # instead of 54 GB/s (tools/step_series4.py: a pause between the genome and the reads, and every step is a later one).
    eng = engine.Engine(A, 5000, 2, device=local, groups=args.groups, host_threads=args.host_threads, flags=(engine.BF_FLAG_ONE_STREAM if args.one_stream else 0) | {"fused": 0, "exact": engine.BF_FLAG_FORCE_EXACT}[args.variant])
    t2 = time.time()
    eng.set_genome(genome.names, genome.seqs)
    log(rank, "genome packed into HBM in %.1fs" % (time.time() - t2))
    t1 = time.time()
    strong = args.config == 2
    want_packed = args.reads == "packed" and args.variant in ("fused", "exact")
    if not want_packed:
        args.reads = "ascii"
    if strong:
        # configs[2]: the SAME 100M reads for every N, as 16 blocks; this rank holds a contiguous run of them
        my_blocks = list(synth.rank_blocks(rank, world))
        index_base = synth.block_first_read(spec, my_blocks[0])
        total_reads = spec.n_reads
        specs = [synth.block_spec(spec, k) for k in my_blocks]
    else:
        # weak scaling: same junctions on every rank, different reads
        spec.read_seed = spec.seed * 1000 + rank
        spec.frag_base = rank * spec.n_reads
        index_base = rank * spec.n_reads
        total_reads = spec.n_reads * world
        specs = [spec]
    ascii_batch, batches = None, []
    for sp in specs:
        blk = synth.make_reads(genome, sp)
        if args.soft_mask > 0:
            blk.seq_bytes &= np.uint8(0xDF)            # sequencer reads are upper case whatever the reference's case
        if ascii_batch is None:
            ascii_batch = blk                          # CPU baseline sample, --check sample, the ASCII legs
        batches.append(engine.pack_batch(blk) if want_packed else blk)
    n_local = sum(b.n for b in batches)
    batch = batches[0]
    log(rank, "reads: %d x %d bp on this rank (%d block(s)%s) in %.1fs" %
        (n_local, L, len(batches), ", as 4-bit codes: the parser's job on the command line" if want_packed else "", time.time() - t1))
    if n_local > 20_000_000 or len(batches) > 1:
        args.no_h2h = True                             # the host-to-host legs pin a second copy of the batch: configs[1]-sized runs only

    # ---- resident inputs -------------------------------------------------------------------------
    t3 = time.time()
    if world > 1 or forced:
        # the sharded path runs inside the library (bf_shard_*): rank 0 makes the RCCL id, the launcher's process
        # group only carries those 128 bytes
        box = [engine.Engine.shard_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        eng.shard_init_rccl(rank, world, index_base, box[0])
    for b_ in batches:
        eng.submit(b_)
    eng.submit_wait()

    def nbytes(b):
        return sum(getattr(b, f).nbytes for f in b.__dataclass_fields__)
    in_bytes = sum(nbytes(b_) for b_ in batches)

    def wire_bytes(b):
        """what crosses PCIe: for 4-bit batches the library uploads lengths (2 + 2 bytes a read) instead of the two offset arrays"""
        if isinstance(b, engine.PackedBatch):
            return nbytes(b) - b.seq_off.nbytes - b.frag_off.nbytes + 2 * b.n
        return nbytes(b)
    on_wire = sum(wire_bytes(b_) for b_ in batches)
    log(rank, "reads resident in HBM in %.2fs (first hand-over, buffers still growing: %.2f GB/s)" %
        (time.time() - t3, in_bytes / 1e9 / max(time.time() - t3, 1e-9)))

    step = eng.shard_finish_raw if (world > 1 or forced) else eng.finish_raw

    sharded_run = world > 1 or forced

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local])
        torch.cuda.synchronize()
        eng.sync()

    def measure_resident():
        """W untimed steps, then exactly K timed ones between barriers; the maximum over the ranks"""
        for _ in range(args.warmup):
            step()
        eng.kernel_times()                                # drop warm-up timings
        barrier()
        t_start = time.perf_counter()
        last_ = None
        for _ in range(args.steps):
            last_ = step()
        barrier()
        dt_ = time.perf_counter() - t_start
        if world > 1:
            t = torch.tensor([dt_], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t.item())
        return dt_, last_, eng.kernel_times()

    elapsed, last, ktimes = measure_resident()
    sharded = world > 1 or forced
    n_clusters = int(last.n_clusters)
    n_support = int(last.n_reads)
    if world > 1:                                      # totals over all owner ranks
        t = torch.tensor([n_clusters, n_support], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        n_clusters, n_support = (int(v) for v in t.tolist())

    if world > 1:                                      # what the ranks really hold (strong scaling: the sum is the config)
        t = torch.tensor([n_local], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        assert int(t.item()) == total_reads, "ranks hold %d reads, the workload has %d" % (int(t.item()), total_reads)
    # ---- host memory in, host memory out (SURVEY 8d's wording), one GPU -------------------------------
    h2h = None
    if not sharded and not args.no_h2h:
        def timed(source, name):
            def one():
                eng.reset()
                eng.submit(source)
                return eng.finish_raw()
            for _ in range(max(args.warmup, 1)):
                one()
            eng.kernel_times()
            barrier()
            t0_ = time.perf_counter()
            for _ in range(args.steps):
                r_ = one()
            barrier()
            dt = (time.perf_counter() - t0_) / args.steps
            kt = eng.kernel_times()
            assert int(r_.n_clusters) == n_clusters and int(r_.n_reads) == n_support, "host-to-host result differs"
            log(rank, "%s: %.2f ms per step, %.1f M reads/s, h2d %.2f ms (%.1f GB/s)" %
                (name, dt * 1e3, n_local / dt / 1e6, kt.get("h2d", (0, 1))[0] / args.steps,
                 on_wire / 1e9 / max(kt.get("h2d", (0, 1))[0] / args.steps * 1e-3, 1e-9)))
            return {"value": round(n_local / dt, 1), "unit": "reads/s", "ms_per_step": round(dt * 1e3, 3),
                    "stage_ms": {k: round(v[0] / args.steps, 4) for k, v in kt.items()}}
        t4 = time.time()
        pinned = eng.pin_batch(batch)
        log(rank, "batch copied into bf_host_alloc memory in %.1fs" % (time.time() - t4))
        h2h = timed(pinned, "host-to-host, pinned batches, %s reads" % args.reads)
        h2h["input"] = "the whole batch (%s reads) in bf_host_alloc (pinned) memory, one bf_submit per step" % args.reads
        h2h["input_bytes"] = int(in_bytes)
        h2h["wire_bytes"] = int(on_wire)
        h2h["pcie_bound_ms"] = round(on_wire / 55e9 * 1e3, 2)        # ~55 GB/s achievable on PCIe 5.0 x16
        h2h["pageable"] = timed(batch, "host-to-host, pageable batches, %s reads" % args.reads)
        h2h["pageable"]["input"] = "the same batch in ordinary numpy arrays, staged by the library's host threads"
        if args.reads == "packed":
            # the same reads at two bits a base on PCIe (bf_submit_packed2: expanded to the resident 4-bit form on the device,
            # behind every uploaded piece), pinned and pageable
            t5 = time.time()
            two = engine.pack2_batch(batch)
            del pinned
            pinned = eng.pin_batch(two)
            log(rank, "batch packed to 2-bit codes (%d exceptions) and copied into bf_host_alloc memory in %.1fs" % (two.exc_read.size, time.time() - t5))
            keep_in, keep_wire = in_bytes, on_wire
            in_bytes = nbytes(two)
            on_wire = in_bytes - two.seq_off.nbytes - two.frag_off.nbytes + 2 * two.n
            h2h["two_bit_reads"] = timed(pinned, "host-to-host, pinned batches, 2-bit reads")
            h2h["two_bit_reads"].update({"input": "bf_submit_packed2: 2 bits a base + exceptions for the bases that are none of ACGT",
                                         "input_bytes": int(in_bytes), "wire_bytes": int(on_wire), "pcie_bound_ms": round(on_wire / 55e9 * 1e3, 2)})
            h2h["two_bit_reads"]["pageable"] = timed(two, "host-to-host, pageable batches, 2-bit reads")
            in_bytes, on_wire = keep_in, keep_wire
            del two
            # the same with ASCII reads (bf_submit): a caller that does not pack
            in_bytes_ascii = nbytes(ascii_batch)
            del pinned
            pinned = eng.pin_batch(ascii_batch)
            keep_in, keep_wire = in_bytes, on_wire
            in_bytes = on_wire = in_bytes_ascii
            h2h["ascii_reads"] = timed(pinned, "host-to-host, pinned batches, ascii reads")
            h2h["ascii_reads"]["input_bytes"] = int(in_bytes_ascii)
            # ... and the resident-input step on them
            for _ in range(args.warmup):
                eng.finish_raw()
            eng.kernel_times()
            barrier()
            t0_ = time.perf_counter()
            for _ in range(args.steps):
                eng.finish_raw()
            barrier()
            dt = (time.perf_counter() - t0_) / args.steps
            kt = eng.kernel_times()
            h2h["ascii_reads"]["resident_step"] = {"value": round(n_local / dt, 1), "ms_per_step": round(dt * 1e3, 3),
                                                   "localise_ms": round(kt.get("localise", (0, 1))[0] / max(kt.get("localise", (0, 1))[1], 1), 4)}
            log(rank, "ascii reads, resident step: %.3f ms, localise %.3f ms" % (dt * 1e3, h2h["ascii_reads"]["resident_step"]["localise_ms"]))
            in_bytes, on_wire = keep_in, keep_wire

    # ---- the same step when the caller keeps its batches (BF_FLAG_READS_ON_HOST): junction records name the read instead
    #      of carrying its bases back over PCIe.  A second ctx; reported beside `value`, never as `value`. ------------------
    roh = None
    if not sharded and not args.no_h2h and args.variant == "fused":
        try:
            e3 = engine.Engine(A, 5000, 2, device=local, groups=args.groups if args.roh_groups < 0 else args.roh_groups, host_threads=args.host_threads,
                               flags=engine.BF_FLAG_READS_ON_HOST | (engine.BF_FLAG_ONE_STREAM if args.one_stream else 0))
            e3.set_genome(genome.names, genome.seqs)
            time.sleep(0.5)                            # (the driver clears the raw bytes bf_genome_end released: see above)
            e3.submit(batch)
            e3.submit_wait()
            for _ in range(max(args.warmup, 1)):
                r_ = e3.finish_raw()
            e3.kernel_times()
            e3.sync()
            t0_ = time.perf_counter()
            for _ in range(args.steps):
                r_ = e3.finish_raw()
            dt = (time.perf_counter() - t0_) / args.steps
            kt = e3.kernel_times()
            assert int(r_.n_clusters) == n_clusters and int(r_.n_reads) == n_support, "reads-on-host result differs"
            roh = {"value": round(n_local / dt, 1), "unit": "reads/s", "ms_per_step": round(dt * 1e3, 3),
                   "record_bytes": int(r_.junction_size), "record_bytes_self_contained": int(last.junction_size),
                   "stage_ms": {k: round(v[0] / args.steps, 4) for k, v in kt.items()},
                   "note": "BF_FLAG_READS_ON_HOST: the caller keeps the submitted batches; records name the read, bf_format_sv "
                           "reads the bases where the caller left them (same .sv bytes, tests/test_gpu_reads_on_host.py)"}
            log(rank, "reads stay on the host: %.3f ms per step, %.1f M reads/s, %d record bytes (self-contained: %d)" %
                (dt * 1e3, n_local / dt / 1e6, roh["record_bytes"], roh["record_bytes_self_contained"]))
            # ... and host memory in, host memory out under the same flag (the batch in this ctx's pinned memory)
            pinned3 = e3.pin_batch(batch)

            def one3():
                e3.reset()
                e3.submit(pinned3)
                return e3.finish_raw()
            for _ in range(max(args.warmup, 1)):
                one3()
            e3.sync()
            t0_ = time.perf_counter()
            for _ in range(args.steps):
                r_ = one3()
            dt = (time.perf_counter() - t0_) / args.steps
            assert int(r_.n_clusters) == n_clusters and int(r_.n_reads) == n_support, "reads-on-host result differs"
            roh["host_to_host"] = {"value": round(n_local / dt, 1), "unit": "reads/s", "ms_per_step": round(dt * 1e3, 3)}
            log(rank, "reads stay on the host, host to host: %.2f ms per step, %.1f M reads/s" % (dt * 1e3, n_local / dt / 1e6))
            e3.close()
        except Exception as e:                     # an extra leg must not cost the bench line
            log(rank, "reads-on-host leg failed: %r" % (e,))
            roh = {"error": repr(e)}

    # ---- the genome's way into HBM (outside the step: once per genome): the FASTA's bytes, or a packed image ----------------
    gload = None
    if not sharded and not args.no_h2h and rank == 0:
        try:
            import tempfile
            eg = engine.Engine(A, 5000, 2, device=local)
            raw_bytes = int(sum(s_.size for s_ in genome.seqs))

            def best(fn, n=2):
                out = []
                for _ in range(n):
                    t0_ = time.perf_counter(); fn(); eg.sync(); out.append(time.perf_counter() - t0_)
                return min(out)
            t_raw = best(lambda: eg.set_genome(genome.names, genome.seqs))
            resident = eg.stats()[1]
            with tempfile.TemporaryDirectory(dir="/tmp") as td:
                path = os.path.join(td, "genome.bf2bit")
                t0_ = time.perf_counter()
                engine.write_genome_image(path, genome.names, genome.seqs)
                t_pack = time.perf_counter() - t0_
                img = engine.GenomeImage(path)
                t_img = best(lambda: eg.set_genome_image(img))
                file_bytes, up = os.path.getsize(path), img.upload_bytes
                img.close()
            eg.close()
            gload = {"fasta_bytes": {"ms": round(t_raw * 1e3, 2), "wire_bytes": raw_bytes, "pcie_bound_ms": round(raw_bytes / 55e9 * 1e3, 2)},
                     "packed_image": {"ms": round(t_img * 1e3, 2), "wire_bytes": int(up), "pcie_bound_ms": round(up / 55e9 * 1e3, 2),
                                      "file_bytes": int(file_bytes), "written_on_the_host_in_s": round(t_pack, 2)},
                     "resident_bytes": int(resident),
                     "note": "bf_genome_set from pageable contig bytes against bf_genome_set_image from a mapped image file (page cache); "
                             "the raw bytes are released after the pack either way; same resident genome (tests/test_gpu_submit.py)"}
            log(rank, "genome into HBM: %.1f ms from its bytes (%.2f GB), %.1f ms from its image (%.2f GB, written in %.1fs); %.2f GB resident" %
                (t_raw * 1e3, raw_bytes / 1e9, t_img * 1e3, up / 1e9, t_pack, resident / 1e9))
        except Exception as e:                     # an extra leg must not cost the bench line
            log(rank, "genome-load leg failed: %r" % (e,))
            gload = {"error": repr(e)}

    ms_per_step = elapsed / args.steps * 1e3
    value = total_reads * args.steps / elapsed

    # ---- roofline of the dominant kernel -----------------------------------------------------------
    loc_ms, loc_n = ktimes.get("localise", (0.0, 0))
    loc_avg_ms = loc_ms / max(loc_n, 1)
    bpr = algorithmic_bytes_per_read(L)
    achieved = n_local * bpr / (loc_avg_ms * 1e-3) / 1e9 if loc_avg_ms > 0 else 0.0
    traffic, traffic_source = None, None
    if rank == 0 and world == 1 and not sharded and not args.no_live_traffic and not args.no_h2h:
        traffic, traffic_source = live_pmc_traffic(args)
    if traffic is None:
        why = traffic_source
        traffic = pmc_traffic(args, n_local, L)
        traffic_source = None if traffic is None else "committed passes of this workload (profiles/*_pmc_traffic.json, tools/pmc_traffic.py)" + (
            "; live passes: %s" % why if why else "")
    roofline = {"bound": "hbm", "kernel": "localise_fast_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_read": bpr, "reads_per_launch": n_local, "avg_launch_ms": round(loc_avg_ms, 4)}
    # per step (a stage runs once per contig group and step; the roofline kernel above once per step)
    stages = {k: round(v[0] / args.steps, 4) for k, v in ktimes.items()}

    # ---- CPU baseline (rank 0, single GPU run only) -------------------------------------------------
    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        import oracle_lib as ol
        ns = min(args.cpu_sample, ascii_batch.n)
        sample = ascii_batch.slice(0, ns)
        og = ol.OracleGenome(genome.names, genome.seqs)
        tc = time.perf_counter()
        ref = ol.detect_batch(og, genome.names, sample, A, 5000, 2, stage=3)
        dt = time.perf_counter() - tc
        cpu = {"value": round(ns / dt, 1), "unit": "reads/s", "cores": 1, "kind": "port",
               "sample": "first %d reads of the same batch, the GPU step's scope (localise + sort + cluster + dedup -> "
                         "cluster table, no .sv text), C++ restatement of detect.rs:105-287, single thread, %.1fs"
                         % (ns, dt),
               "cpu_model": cpu_model(), "host_cores": os.cpu_count()}
        if args.check and not sharded:
            ref = ol.detect_batch(og, genome.names, sample, A, 5000, 2)
            e2 = engine.Engine(A, 5000, 2, device=local)
            e2.set_genome(genome.names, genome.seqs)
            e2.submit(sample)
            same = e2.format_sv(e2.finish(reads=False)) == ref.sv
            e2.close()
            cpu["gpu_output_identical_on_sample"] = bool(same)
            if not same:
                raise SystemExit("bench --check: GPU .sv differs from the oracle's on the sample")

    shard_check = sharded_check(args, eng, rank, world, dev, genome, spec, strong, A) if (args.check and sharded) else None

    if rank == 0:
        out = {
            "metric": "candidate reads/sec breakpoint-localised+deduped", "value": round(value, 1), "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u8/u64 bit-planes",
            "data": "synthetic (seed %d): random hg38-sized genome, simulated junction reads + the Bowtie1 hits of their anchors" % spec.seed,
            "config": {"workload": ("configs[2]: %d unaligned 2x%d bp reads in %d blocks sharded over %d GPU(s) in contiguous input ranges, "
                                    "anchor hits on a %.2f Gb genome, anchor length %d" % (total_reads, L, synth.BLOCKS, world, sum(l for _, l in contigs) / 1e9, A))
                       if strong else
                       "configs[%d]: %d unaligned 2x%d bp reads per GPU with anchor hits on a %.2f Gb genome, anchor length %d"
                       % (args.config, n_local, L, sum(l for _, l in contigs) / 1e9, A),
                       "reads_per_gpu": n_local, "reads_total": total_reads, "read_len": L, "anchor_len": A, "max_frag_len": 5000, "min_evidence": 2,
                       "reads_as": {"packed": "BAM 4-bit base codes (bf_submit_packed)", "ascii": "ASCII (bf_submit)"}[args.reads],
                       "parallelism": "shard%d+alltoall" % world if sharded else "single"},
            "result": {"clusters": n_clusters, "supporting_reads": n_support},
            "stage_ms": stages, "roofline": roofline, "cpu_baseline": cpu, "host_to_host": h2h, "reads_on_host": roh, "genome_load": gload,
            "sharded_check": shard_check,
            "pipeline": pipeline_headline(n_local, L, ms_per_step),
            "published_reference": "~40M BAM reads/min/core whole-file scan (README.md:6), a different stage",
        }
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    eng.close()
    if world > 1:
        dist.destroy_process_group()


def sharded_check(args, eng, rank, world, dev, genome, spec, strong, A):
    """--check on the sharded path: the merged table of all ranks (bf_shard_format_sv over the ranks' results) against the
    table ONE GPU makes of the same reads through bf_finish (which the parity tests hold against the oracle): SHA-256.
    Rank 0 regenerates the other ranks' reads for the one-GPU run, so this is for --scale'd runs or a roomy host."""
    import hashlib
    mine = eng.shard_finish()
    if world > 1:
        box = [None] * world if rank == 0 else None
        dist.gather_object(mine, box, dst=0)
    else:
        box = [mine]
    if rank != 0:
        return None
    merged = hashlib.sha256(memoryview(eng.shard_format_sv(box, as_array=True))).hexdigest()
    e2 = engine.Engine(A, 5000, 2, device=dev.index)
    try:
        e2.set_genome(genome.names, genome.seqs)
        if strong:
            all_specs = [synth.block_spec(spec, k) for k in range(synth.BLOCKS)]
        else:
            import copy
            all_specs = []
            for r in range(world):
                sp = copy.copy(spec)
                sp.read_seed = spec.seed * 1000 + r
                sp.frag_base = r * spec.n_reads
                all_specs.append(sp)
        n_all = 0
        for sp in all_specs:
            blk = synth.make_reads(genome, sp)
            if args.soft_mask > 0:
                blk.seq_bytes &= np.uint8(0xDF)
            n_all += blk.n
            e2.submit(engine.pack_batch(blk) if args.reads == "packed" else blk)
        single = hashlib.sha256(memoryview(e2.format_sv(e2.finish(copy=False, reads=False), as_array=True))).hexdigest()
    finally:
        e2.close()
    same = merged == single
    log(rank, "sharded --check: merged table of %d rank(s) %s the one-GPU table of the same %d reads (sha256 %s)" %
        (world, "==" if same else "DIFFERS FROM", n_all, merged[:16]))
    if not same:
        raise SystemExit("bench --check: the sharded path's merged .sv differs from the one-GPU .sv")
    return {"ranks": world, "reads": n_all, "sha256": merged, "identical_to_one_gpu": True}


def pipeline_headline(n_local, L, ms_per_step):
    """SURVEY.md section 8d's pipeline headline: N * (B_A + B_S) / t_device_pipeline, B_S = 424 B/read as the survey
    prices an 8-bit-digit LSD sort of the full key (this library sorts 8 passes, not 12; the figure is the survey's)."""
    b_a, b_s = algorithmic_bytes_per_read(L), 424
    gbps = n_local * (b_a + b_s) / (ms_per_step * 1e-3) / 1e9
    return {"algorithmic_bytes_per_read": b_a + b_s, "achieved": round(gbps, 1), "unit": "GB/s",
            "frac": round(gbps / HBM_PEAK_GBPS, 4), "note": "whole device pipeline incl. the PCIe copy of the result"}


def live_pmc_traffic(args):
    """HBM bytes per launch of the localise kernel, measured NOW: two rocprofv3 passes (FETCH_SIZE, WRITE_SIZE: separate, as
    the TCC counter slots demand) of a short run of this very command in a child process each, turned into bytes as
    tools/pmc_traffic.py does (FETCH_SIZE counts 128-byte requests at 64 bytes on gfx950: the read side is doubled).
    -> (bytes, where they came from) or (None, why not)."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself under a profiler"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import pmc_traffic as pt
        t0 = time.time()
        per_launch = {}
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                d = os.path.join(td, counter)
                cmd = ["timeout", "-s", "KILL", "170", exe, "--pmc", counter, "-d", d, "-o", "p", "--output-format", "csv", "--",
                       sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--no-h2h",
                       "--one-stream", "--config", str(args.config), "--scale", str(args.scale), "--reads", args.reads,
                       "--variant", args.variant, "--no-live-traffic"]
                r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", timeout=200)
                if r.returncode != 0:
                    return None, "rocprofv3 --pmc %s ended with status %d" % (counter, r.returncode)
                vals = [v for k, v in pt.per_kernel(d, counter).items() if "localise_fast_kernel" in k]
                if not vals or not vals[0]:
                    return None, "no localise_fast_kernel in the %s pass" % counter
                per_launch[counter] = sum(vals[0]) / len(vals[0])
        total = per_launch["FETCH_SIZE"] * 1024 * 2 + per_launch["WRITE_SIZE"] * 1024
        print("[bench] live PMC passes: %.2f GB per launch of the localise kernel (%.0f s)" % (total / 1e9, time.time() - t0), file=sys.stderr, flush=True)
        return int(round(total)), "two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of this command during this run, per launch over the whole input"
    except Exception as e:                         # an extra leg must not cost the bench line
        return None, repr(e)


def pmc_traffic(args, n_local, L):
    """HBM bytes per launch of the localise kernel from the committed rocprofv3 PMC passes
    (profiles/*_pmc_traffic.json, made by tools/pmc_traffic.py on this very workload and variant);
    None when no pass matches the run's configuration."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
            want = "config=%d reads=%d len=%d variant=%s reads_as=%s" % (args.config, n_local, L, args.variant, args.reads)
            if want in d.get("workload", ""):
                for k, v in d["kernels"].items():
                    if "localise_fast_kernel" in k:
                        return v["hbm_bytes_per_launch"]
        except Exception:
            continue
    return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
