"""Seeded synthetic inputs for BASELINE.json's configs.

Neither a genome, a BAM nor `bowtie` exists in the build or GPU image, so inputs are generated:
a random genome and split reads spanning simulated junctions, together with the anchor alignments
Bowtie1 (`-v0 -m1 -B1`, detect.rs:61) would report for them: exact, unique hits of the first and
last `anchor_len` bases, 1-based leftmost offset, strand.  Reads whose anchors are not error-free
are never emitted (bowtie prints nothing for them).

PRNG: numpy PCG64 seeded with `seed + stream` (seed 20240807 in the BASELINE configs).
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .engine import HIT_DTYPE, Batch

SEED = 20240807
_ACGT = np.frombuffer(b"ACGT", np.uint8)
_COMP = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"AGCTYRWSKMDVHBN", b"TCGARYWSMKHBDVN"):
    _COMP[_a] = _b
    _COMP[_a + 32] = _b + 32

# hg38 primary assembly lengths (chr1..22, X, Y, M) -- only the sizes are used
HG38 = [("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555),
        ("chr5", 181538259), ("chr6", 170805979), ("chr7", 159345973), ("chr8", 145138636),
        ("chr9", 138394717), ("chr10", 133797422), ("chr11", 135086622), ("chr12", 133275309),
        ("chr13", 114364328), ("chr14", 107043718), ("chr15", 101991189), ("chr16", 90338345),
        ("chr17", 83257441), ("chr18", 80373285), ("chr19", 58617616), ("chr20", 64444167),
        ("chr21", 46709983), ("chr22", 50818468), ("chrX", 156040895), ("chrY", 57227415),
        ("chrM", 16569)]


_fast = None


def fast_lib():
    """csrc/libbf_synth.so (synth_fill.c: the generator's two inner loops in C + OpenMP), or None: the numpy code
    below then does the same work, byte for byte, a dozen times slower.  BREAKFAST_SYNTH_NUMPY=1 forces numpy."""
    global _fast
    if _fast is None:
        _fast = False
        so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libbf_synth.so")
        if os.path.exists(so) and os.environ.get("BREAKFAST_SYNTH_NUMPY") != "1":
            try:
                import ctypes as C
                lib = C.CDLL(so)
                lib.bf_synth_fill.restype = None
                lib.bf_synth_fill.argtypes = [C.c_void_p] * 10 + [C.c_int64, C.c_int64, C.c_void_p]
                lib.bf_synth_pack_fixed.restype = C.c_int64
                lib.bf_synth_pack_fixed.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
                if hasattr(lib, "bf_synth_pack2_fixed"):
                    lib.bf_synth_pack2_fixed.restype = C.c_int64
                    lib.bf_synth_pack2_fixed.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
                _fast = lib
            except OSError:
                _fast = False
    return _fast or None


def host_threads() -> int:
    """Worker threads for the generators: the GPU boxes give a job 16 CPUs whatever os.cpu_count() says."""
    env = os.environ.get("BREAKFAST_SYNTH_THREADS")
    if env:
        return max(1, int(env))
    try:
        have = len(os.sched_getaffinity(0))
    except Exception:
        have = os.cpu_count() or 1
    return max(1, min(16, have))


@dataclass
class Genome:
    names: List[bytes]
    seqs: List[np.ndarray]            # uint8 ASCII, case preserved

    def as_dict(self) -> Dict[bytes, bytes]:
        return {n: s.tobytes() for n, s in zip(self.names, self.seqs)}

    def fasta(self, width: int = 60) -> bytes:
        out = []
        for n, s in zip(self.names, self.seqs):
            out.append(b">" + n + b"\n")
            full = (s.shape[0] // width) * width
            if full:
                rows = np.empty((full // width, width + 1), np.uint8)
                rows[:, :width] = s[:full].reshape(-1, width)
                rows[:, width] = 10
                out.append(rows.tobytes())
            if s.shape[0] > full:
                out.append(s[full:].tobytes() + b"\n")
        return b"".join(out)


def make_genome(contigs: Sequence[Tuple[str, int]], seed: int = SEED, end_n: int = 0,
                n_runs: Sequence[Tuple[int, int, int]] = (), lower_runs: Sequence[Tuple[int, int, int]] = ()
                ) -> Genome:
    """Uniform i.i.d. ACGT contigs.  end_n: length of the N run at both contig ends;
    n_runs / lower_runs: (contig index, start, length) runs of 'N' / of lowercase bases."""
    def one(ci: int) -> np.ndarray:
        length = contigs[ci][1]
        rng = np.random.Generator(np.random.PCG64([seed, 3, ci]))        # one stream per contig
        s = _ACGT[rng.integers(0, 4, size=length, dtype=np.uint8)]
        if end_n and length > 4 * end_n:
            s[:end_n] = ord("N")
            s[-end_n:] = ord("N")
        return s

    names = [name.encode() for name, _ in contigs]
    workers = max(1, min(host_threads(), len(contigs)))
    if workers == 1:
        seqs = [one(ci) for ci in range(len(contigs))]
    else:
        with ThreadPoolExecutor(workers) as ex:
            seqs = list(ex.map(one, range(len(contigs))))
    for ci, a, n in n_runs:
        seqs[ci][a:a + n] = ord("N")
    for ci, a, n in lower_runs:
        seqs[ci][a:a + n] |= 0x20
    return Genome(names, seqs)


@dataclass
class ReadSpec:
    n_reads: int
    read_len: int = 150
    anchor_len: int = 20
    n_events: int = 1000           # junctions shared by many reads
    singleton_frac: float = 0.5    # reads from one-off random chimeras (dropped by min_evidence)
    mate_frac: float = 0.3         # fraction of event reads that re-use the previous read's QNAME
    subst_rate: float = 0.005      # per-base substitutions outside the anchors
    n_rate: float = 0.0            # per-base 'N' in the read outside the anchors
    deletion_frac: float = 0.5     # share of events that are same-contig deletions (rest: random pairs)
    hot_frac: float = 0.0          # reads piling onto `hot_events` junctions with distinct QNAMEs
    hot_events: int = 0
    margin: int = 0                # keep junctions this far from contig ends (0 -> read_len + 8)
    seed: int = SEED               # junction tables (shared by all ranks of a sharded run)
    read_seed: Optional[int] = None  # per-read draws (None -> seed); differs per rank
    frag_base: int = 0             # first fragment id number (keeps ids unique across ranks)
    chunk: int = 1 << 18


def _digits(ids: np.ndarray, width: int) -> np.ndarray:
    """ids -> fixed-width ASCII decimal, shape (n, width)."""
    out = np.empty((ids.shape[0], width), np.uint8)
    v = ids.astype(np.uint64).copy()
    for k in range(width - 1, -1, -1):
        out[:, k] = (v % 10).astype(np.uint8) + 48
        v //= 10
    return out


def make_reads(genome: Genome, spec: ReadSpec) -> Batch:
    """Split reads + the bowtie hits of their anchors, in a seeded random input order."""
    erng = np.random.Generator(np.random.PCG64(spec.seed + 1))          # junction tables
    rseed = spec.seed if spec.read_seed is None else spec.read_seed
    rng = np.random.Generator(np.random.PCG64([rseed, 5]))              # per-read draws
    L, A, n = spec.read_len, spec.anchor_len, spec.n_reads
    assert L >= 2 * A and A >= 8
    lens = np.array([s.shape[0] for s in genome.seqs], np.int64)
    starts = np.zeros(len(lens) + 1, np.int64)
    starts[1:] = np.cumsum(lens)
    cat = np.concatenate(genome.seqs) if len(genome.seqs) > 1 else genome.seqs[0]
    margin = spec.margin or (L + 8)
    usable = np.maximum(lens - 2 * margin, 0)
    assert usable.sum() > 0, "contigs too short for this read length"
    cw = usable / usable.sum()

    def random_loci(g, m: int) -> Tuple[np.ndarray, np.ndarray]:
        c = g.choice(len(lens), size=m, p=cw)
        x = margin + (g.random(m) * usable[c]).astype(np.int64)
        return c.astype(np.int64), x

    def make_events(g, m: int):
        c1, x1 = random_loci(g, m)
        c2, y1 = random_loci(g, m)
        s1 = g.random(m) < 0.5
        s2 = g.random(m) < 0.5
        dele = g.random(m) < spec.deletion_frac            # same-contig deletion, both '+'
        gap = (50 + g.random(m) * 1_000_000).astype(np.int64)
        yd = np.minimum(x1 + 1 + gap, margin + usable[c1] - 1)
        c2 = np.where(dele, c1, c2)
        y1 = np.where(dele, yd, y1)
        s1 = np.where(dele, True, s1)
        s2 = np.where(dele, True, s2)
        return c1, x1, s1, c2, y1, s2

    n_sing = int(round(n * spec.singleton_frac))
    n_hot = int(round(n * spec.hot_frac)) if spec.hot_events else 0
    n_ev = n - n_sing - n_hot
    ev = make_events(erng, max(spec.n_events, 1))
    hot = make_events(erng, max(spec.hot_events, 1))
    sing = make_events(rng, max(n_sing, 1))

    # per-read junction parameters, grouped (event reads sorted by event so mates are neighbours)
    eid = np.sort(rng.integers(0, max(spec.n_events, 1), size=n_ev))
    hid = rng.integers(0, max(spec.hot_events, 1), size=n_hot)
    cols = []
    for k in range(6):
        cols.append(np.concatenate([ev[k][eid], hot[k][hid], sing[k][:n_sing]]))
    c1, x1, s1, c2, y1, s2 = cols
    # fragment ids: unique per read, except that a "mate" re-uses the previous read's id when it
    # covers the same event
    fid = spec.frag_base + np.arange(n, dtype=np.int64)
    is_mate = np.zeros(n, bool)
    if n_ev > 1:
        m = (rng.random(n_ev) < spec.mate_frac)
        m[0] = False
        m[1:] &= eid[1:] == eid[:-1]
        m[1:] &= ~m[:-1]                      # pairs, not chains
        is_mate[:n_ev] = m
        fid[is_mate] = fid[np.nonzero(is_mate)[0] - 1]
    b = rng.integers(A, L - A, size=n)          # split offset in [A, L-A-1]
    flip = rng.random(n) < 0.5
    perm = rng.permutation(n)
    c1, x1, s1, c2, y1, s2, fid, b, flip = (a[perm] for a in (c1, x1, s1, c2, y1, s2, fid, b, flip))

    seq = np.empty((n, L), np.uint8)
    hits = np.zeros(n, HIT_DTYPE)
    k = np.arange(L, dtype=np.int64)[None, :]
    # 5' part: '+' reads G1[x1-b+1+k]; '-' reads comp(G1[x1+b-1-k]); it ends on x1 either way.
    # 3' part: '+' reads G2[y1+(k-b)]; '-' reads comp(G2[y1-(k-b)]).  Both are base + dir*k.
    base1 = starts[c1] + np.where(s1, x1 - b + 1, x1 + b - 1)
    base2 = starts[c2] + np.where(s2, y1 - b, y1 + b)
    dir1 = np.where(s1, 1, -1).astype(np.int64)
    dir2 = np.where(s2, 1, -1).astype(np.int64)
    # bowtie hits of the unflipped read (1-based leftmost offset of each anchor); presenting the
    # read as its reverse complement swaps the anchors and flips their strands
    pos5 = np.where(s1, x1 - b + 2, x1 + b - A + 1)
    pos3 = np.where(s2, y1 + L - A - b + 1, y1 - L + b + 2)
    hits["pos5"] = np.where(flip, pos3, pos5)
    hits["pos3"] = np.where(flip, pos5, pos3)
    hits["contig5"] = np.where(flip, c2, c1)
    hits["contig3"] = np.where(flip, c1, c2)
    hits["strands"] = (np.where(flip, ~s2, s1).astype(np.uint8) | (np.where(flip, ~s1, s2).astype(np.uint8) << 1))
    fast = fast_lib()
    if fast is not None:                                    # the whole gather in C (same bytes); errors are added per chunk below
        cc = [np.ascontiguousarray(a, np.int64) for a in (base1, base2, dir1, dir2, b)]
        bb = [np.ascontiguousarray(a, np.uint8) for a in (flip, s1, s2)]
        fast.bf_synth_fill(seq.ctypes.data, cat.ctypes.data, *[a.ctypes.data for a in cc], *[a.ctypes.data for a in bb],
                           n, L, _COMP.ctypes.data)

    def build_chunk(ci: int) -> None:
        lo = ci * spec.chunk
        hi = min(n, lo + spec.chunk)
        sl = slice(lo, hi)
        crng = np.random.Generator(np.random.PCG64([rseed, 7, ci]))   # per-chunk stream: thread-count independent
        if fast is not None:
            r = seq[sl]
        else:
            f = flip[sl, None]
            ko = np.where(f, L - 1 - k, k)                      # position inside the unflipped read
            left = ko < b[sl, None]
            gpos = np.where(left, base1[sl, None] + dir1[sl, None] * ko, base2[sl, None] + dir2[sl, None] * ko)
            r = cat[gpos]
            del gpos
            need_comp = np.where(left, ~s1[sl, None], ~s2[sl, None]) ^ f
            r = np.where(need_comp, _COMP[r], r)
            del need_comp, left, ko
        # sequencing errors outside the anchors (an error inside an anchor = no bowtie hit)
        m = hi - lo
        for rate, kind in ((spec.subst_rate, 0), (spec.n_rate, 1)):
            if rate <= 0 or L - 2 * A <= 0:
                continue
            ne = crng.binomial(m * (L - 2 * A), rate)
            rows = crng.integers(0, m, size=ne)
            colz = crng.integers(A, L - A, size=ne)
            cur = r[rows, colz]
            if kind == 0:
                code = (cur >> 1) & 3                        # A0 C1 T2 G3
                alt = np.frombuffer(b"ACTG", np.uint8)[(code + crng.integers(1, 4, size=ne, dtype=np.uint8)) & 3]
                r[rows, colz] = np.where(np.isin(cur, _ACGT), alt, cur)
            else:
                r[rows, colz] = ord("N")
        if fast is None:
            seq[sl] = r

    n_chunks = (n + spec.chunk - 1) // spec.chunk
    workers = max(1, min(host_threads(), n_chunks))
    if workers == 1:
        for ci in range(n_chunks):
            build_chunk(ci)
    else:
        with ThreadPoolExecutor(workers) as ex:
            list(ex.map(build_chunk, range(n_chunks)))
    width = 11
    frag = np.empty((n, width + 1), np.uint8)
    frag[:, 0] = ord("q")
    frag[:, 1:] = _digits(fid, width)
    return Batch(seq.reshape(-1), np.arange(n + 1, dtype=np.uint64) * np.uint64(L),
                 frag.reshape(-1), np.arange(n + 1, dtype=np.uint64) * np.uint64(width + 1), hits)


# ---- a config as independent blocks --------------------------------------------------------------------------------
# configs[2] (100M reads sharded over 2/4/8 GPUs) is DEFINED as kBlocks blocks over one junction table: block b holds
# n_reads / kBlocks reads drawn with its own read seed, fragment ids numbered from b * block size.  Rank r of N holds
# the contiguous blocks [kBlocks * r / N, kBlocks * (r + 1) / N) -- "contiguous line ranges" of SURVEY.md section 8d --
# so the workload is the same bytes for every N (strong scaling), a rank generates only its own slice, and a one-GPU
# run of the whole config builds it block by block.
BLOCKS = 16


def block_spec(spec: ReadSpec, block: int, n_blocks: int = BLOCKS) -> ReadSpec:
    """ReadSpec of block `block` of `n_blocks`: its share of the reads, its own read seed and fragment id range."""
    import copy
    lo = spec.n_reads * block // n_blocks
    hi = spec.n_reads * (block + 1) // n_blocks
    s = copy.copy(spec)
    s.n_reads = hi - lo
    s.read_seed = spec.seed * 1000 + 500 + block
    s.frag_base = spec.frag_base + lo
    return s


def block_first_read(spec: ReadSpec, block: int, n_blocks: int = BLOCKS) -> int:
    return spec.n_reads * block // n_blocks


def rank_blocks(rank: int, world: int, n_blocks: int = BLOCKS) -> range:
    return range(n_blocks * rank // world, n_blocks * (rank + 1) // world)


def bowtie_lines_slow(batch: Batch, contig_names: Sequence[bytes], first_read_num: int = 1) -> bytes:
    """The batch as bowtie's stdout (detect.rs:61 flags: `--suppress 5,6,7,8`): two lines per read.
    Plain loop; bowtie_lines() is the vectorised form used for large batches."""
    out = []
    for i in range(batch.n):
        h = batch.hits[i]
        num = first_read_num + i
        out.append(b"5p:%d:\t%s\t%s\t%d\n" % (num, b"+" if h["strands"] & 1 else b"-",
                                               contig_names[int(h["contig5"])], int(h["pos5"])))
        out.append(b"3p:%d:%s:%s:\t%s\t%s\t%d\n" % (num, batch.frag(i), batch.seq(i),
                                                     b"+" if h["strands"] & 2 else b"-",
                                                     contig_names[int(h["contig3"])], int(h["pos3"])))
    return b"".join(out)


def _decimal_cols(v: np.ndarray) -> np.ndarray:
    """Non-negative ints -> (n, 20) ASCII digits with the leading zeros replaced by NUL bytes."""
    d = _digits(v, 20)
    lead = np.cumsum(d != 48, axis=1) == 0
    lead[:, -1] = False
    d[lead] = 0
    return d


def bowtie_lines(batch: Batch, contig_names: Sequence[bytes], first_read_num: int = 1, chunk: int = 1 << 17) -> bytes:
    """Vectorised bowtie_lines_slow for batches whose reads (and fragment ids) all have one length:
    every line is laid out in a fixed-width row padded with NUL bytes, and the NULs are dropped."""
    n = batch.n
    if n == 0:
        return b""
    sl = np.diff(batch.seq_off.astype(np.int64))
    fl = np.diff(batch.frag_off.astype(np.int64))
    if sl.min() != sl.max() or fl.min() != fl.max():
        return bowtie_lines_slow(batch, contig_names, first_read_num)
    L, F = int(sl[0]), int(fl[0])
    wn = max(len(c) for c in contig_names)
    names = np.zeros((len(contig_names), wn), np.uint8)
    for i, c in enumerate(contig_names):
        names[i, :len(c)] = np.frombuffer(c, np.uint8)
    seq = batch.seq_bytes.reshape(n, L)
    frag = batch.frag_bytes.reshape(n, F)
    lit = lambda b, m: np.broadcast_to(np.frombuffer(b, np.uint8), (m, len(b)))
    out = []
    for a in range(0, n, chunk):
        e = min(n, a + chunk)
        m = e - a
        h = batch.hits[a:e]
        num = _decimal_cols(np.arange(first_read_num + a, first_read_num + e, dtype=np.int64))
        st5 = np.where(h["strands"] & 1, 43, 45).astype(np.uint8)[:, None]
        st3 = np.where(h["strands"] & 2, 43, 45).astype(np.uint8)[:, None]
        row = np.concatenate([
            lit(b"5p:", m), num, lit(b":\t", m), st5, lit(b"\t", m), names[h["contig5"]], lit(b"\t", m),
            _decimal_cols(h["pos5"].astype(np.int64)), lit(b"\n", m),
            lit(b"3p:", m), num, lit(b":", m), frag[a:e], lit(b":", m), seq[a:e], lit(b":\t", m), st3, lit(b"\t", m),
            names[h["contig3"]], lit(b"\t", m), _decimal_cols(h["pos3"].astype(np.int64)), lit(b"\n", m)], axis=1)
        flat = row.reshape(-1)
        out.append(flat[flat != 0].tobytes())
    return b"".join(out)


# ---- BASELINE.json configs ------------------------------------------------------------------------

def config(index: int, scale: float = 1.0, genome_scale: float = 1.0) -> Tuple[List[Tuple[str, int]], dict, ReadSpec]:
    """(contigs, make_genome kwargs, ReadSpec) for BASELINE.json configs[index].  `scale` shrinks the
    read count, `genome_scale` the contig lengths (tests use small values; bench uses 1.0)."""
    seed = SEED + index
    hg = [(n, max(int(l * genome_scale), 20000)) for n, l in HG38]
    if index == 0:      # CPU-runnable case: 100k reads, 2x100 bp, one chr22-sized contig
        ln = max(int(50818468 * genome_scale), 400000)
        contigs = [("chr22", ln)]
        gk = dict(seed=seed, n_runs=[(0, ln // 4, 10000)], lower_runs=[(0, ln // 2, 100000)])
        spec = ReadSpec(n_reads=max(int(100_000 * scale), 200), read_len=100, anchor_len=30, n_events=20,
                        singleton_frac=0.994, mate_frac=0.3, subst_rate=0.005, seed=seed)
        return contigs, gk, spec
    gk = dict(seed=seed, end_n=10000)
    if index == 1:      # headline: 10M x 150 bp, anchor 20
        spec = ReadSpec(n_reads=int(10_000_000 * scale), n_events=max(int(100_000 * scale), 10), seed=seed)
    elif index == 2:    # 100M over 2/4/8 GPUs
        spec = ReadSpec(n_reads=int(100_000_000 * scale), n_events=max(int(1_000_000 * scale), 10), seed=seed)
    elif index == 3:    # duplicate-heavy: 50M, 60 % duplicates
        spec = ReadSpec(n_reads=int(50_000_000 * scale), n_events=max(int(300_000 * scale), 10),
                        singleton_frac=0.2, mate_frac=0.6, hot_frac=0.3,
                        hot_events=max(int(5000 * scale), 2), seed=seed)
    elif index == 4:    # 250 bp, anchor 30, 5 % mismatches + 1 % N
        spec = ReadSpec(n_reads=int(10_000_000 * scale), read_len=250, anchor_len=30,
                        n_events=max(int(100_000 * scale), 10), subst_rate=0.05, n_rate=0.01, seed=seed)
    else:
        raise ValueError("configs are 0..4")
    return hg, gk, spec
