"""ctypes binding of libbreakfast_hip.so (include/breakfast_hip.h) and the host-side mirror of the
reference's consume-loop interface (/root/reference/src/detect.rs:76-103).

This module is plumbing: every computation happens in the HIP library.  There is no CPU
implementation here and no fallback: if the library is missing, or no gfx950 device is usable,
construction fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BREAKFAST_HIP_LIB") or os.path.join(_HERE, "csrc", "libbreakfast_hip.so")   # override: kernel A/B builds

BF_FLAG_KEEP_EVIDENCE = 1
BF_FLAG_FORCE_EXACT = 2
BF_FLAG_SORTED_DEDUP = 32
BF_FLAG_WEAK_DEDUP_HASH = 64
BF_FLAG_ONE_GROUP = 128
BF_FLAG_GROUPS_ALWAYS = 256
BF_FLAG_ONE_STREAM = 512
BF_FLAG_READS_ON_HOST = 2048
BF_EV_STRAND, BF_EV_MSTRAND, BF_EV_SWAPPED, BF_EV_EXACT, BF_EV_SIGX = 1, 2, 4, 8, 16

HIT_DTYPE = np.dtype([("pos5", "<u4"), ("pos3", "<u4"), ("contig5", "<u2"), ("contig3", "<u2"),
                      ("strands", "u1"), ("pad", "u1", (3,))])
CLUSTER_DTYPE = np.dtype([("lpos", "<i8"), ("rpos", "<i8"), ("seed_index", "<u8"),
                          ("read_begin", "<u8"), ("read_end", "<u8"), ("contig", "<u4"),
                          ("mcontig", "<u4"), ("strand", "u1"), ("mstrand", "u1"), ("pad", "u1", (6,))])
READ_DTYPE = np.dtype([("index", "<u8"), ("junction_off", "<u8")])
assert HIT_DTYPE.itemsize == 16 and CLUSTER_DTYPE.itemsize == 56 and READ_DTYPE.itemsize == 16


class BfConfig(C.Structure):
    _fields_ = [("anchor_len", C.c_uint32), ("max_frag_len", C.c_uint32), ("min_evidence", C.c_uint32),
                ("device", C.c_int32), ("flags", C.c_uint32), ("reserved", C.c_uint32 * 3)]


class BfBatch(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("seq_bytes", C.c_void_p), ("seq_off", C.c_void_p),
                ("frag_bytes", C.c_void_p), ("frag_off", C.c_void_p), ("hits", C.c_void_p)]


class BfPackedBatch(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("seq_nibbles", C.c_void_p), ("seq_off", C.c_void_p), ("seq_len", C.c_void_p),
                ("frag_bytes", C.c_void_p), ("frag_off", C.c_void_p), ("hits", C.c_void_p)]


class BfPacked2Batch(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("seq_2bit", C.c_void_p), ("seq_off", C.c_void_p), ("seq_len", C.c_void_p),
                ("n_exceptions", C.c_uint64), ("exc_read", C.c_void_p), ("exc_pos", C.c_void_p), ("exc_code", C.c_void_p),
                ("frag_bytes", C.c_void_p), ("frag_off", C.c_void_p), ("hits", C.c_void_p)]


class BfResult(C.Structure):
    _fields_ = [("n_evidence", C.c_uint64), ("n_clusters", C.c_uint64), ("n_reads", C.c_uint64),
                ("clusters", C.c_void_p), ("reads", C.c_void_p), ("junction_bytes", C.c_void_p),
                ("junction_size", C.c_uint64), ("cluster_junction_off", C.c_void_p)]


class BfShardResult(C.Structure):
    _fields_ = [("n_local", C.c_uint64), ("n_evidence", C.c_uint64), ("n_clusters", C.c_uint64), ("n_reads", C.c_uint64),
                ("clusters", C.c_void_p), ("src_rank", C.c_void_p), ("n_groups", C.c_uint32), ("reserved", C.c_uint32),
                ("group_read_begin", C.c_void_p), ("n_records", C.c_uint64), ("block_off", C.c_void_p),
                ("record_bytes", C.c_void_p), ("record_size", C.c_uint64), ("places", C.c_void_p), ("record_off", C.c_void_p)]


FILL_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_uint64)     # bf_fill_fn(user, dst, n)

PLACE_DTYPE = np.dtype([("src_rank", "<u4"), ("slot", "<u4")])


@dataclass
class ShardResult:
    """Copy of a bf_shard_result (numpy arrays own their memory; picklable, so ranks can hand it to the merger)."""
    n_local: int
    n_evidence: int
    clusters: np.ndarray       # CLUSTER_DTYPE, this owner's rows
    src_rank: np.ndarray       # uint8 per supporting read: the rank that holds its record
    group_read_begin: np.ndarray   # uint64[n_groups + 1]
    n_records: int
    block_off: np.ndarray      # uint64[n_groups * n_ranks + 1], this SOURCE's records by (round, owner)
    record_bytes: np.ndarray
    places: Optional[np.ndarray] = None       # PLACE_DTYPE per supporting read (shard_finish(places=True))
    record_off: Optional[np.ndarray] = None   # uint64[n_records + 1]


class BfEvidenceView(C.Structure):
    _fields_ = [("n", C.c_uint64), ("mask_words", C.c_uint32), ("class_key", C.c_void_p),
                ("lpos", C.c_void_p), ("rpos", C.c_void_p), ("bp", C.c_void_p), ("flags", C.c_void_p),
                ("mask", C.c_void_p)]


class BfKernelTime(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ms", C.c_float), ("launches", C.c_uint64)]


# every symbol include/breakfast_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "bf_abi_version", "bf_ctx_create", "bf_ctx_destroy", "bf_last_error", "bf_genome_set", "bf_submit",
    "bf_localise", "bf_clusterise", "bf_finish", "bf_reset", "bf_sync", "bf_format_sv",
    "bf_render_junctions", "bf_render_junctions_dev", "bf_render_junctions_dev_static",
    "bf_evidence_fetch", "bf_kernel_times", "bf_stats", "bf_junction_text", "bf_junction_text_of", "bf_result_junction_text",
    "bf_junction_record_bound", "bf_junction_record_bytes", "bf_result_reads",
    "bf_host_alloc", "bf_host_free", "bf_submit_wait", "bf_reserve",
    "bf_genome_begin", "bf_genome_write", "bf_genome_fill", "bf_genome_end", "bf_genome_read", "bf_submit_packed", "bf_submit_packed2",
    "bf_shard_unique_id", "bf_shard_init_rccl", "bf_shard_init_host", "bf_shard_finish", "bf_shard_result_places", "bf_shard_format_sv",
    "bf_shard_format_sv_names",
    "bf_genome_image_write", "bf_genome_image_open", "bf_genome_image_info", "bf_genome_image_close", "bf_genome_begin_image",
    "bf_genome_write_image", "bf_genome_set_image",
]

_lib = None


class BreakfastError(RuntimeError):
    """A C-ABI call returned a negative status.  `.code` is the BF_ERR_* value."""

    def __init__(self, code: int, msg: str):
        super().__init__("breakfast_hip error %d: %s" % (code, msg))
        self.code = code


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Loads the in-tree HIP library.  No fallback: a missing library is an error."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(or `make -C breakfast_amd/csrc`); there is no CPU fallback" % p)
    if os.environ.get("BREAKFAST_NO_TORCH") != "1":
        # torch ships its own libamdhip64.so.7; load it first so that both share ONE HIP runtime
        # (device pointers, streams) whichever of the two is imported first by the caller.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
    vp, u64p = C.c_void_p, C.POINTER(C.c_uint64)
    lib.bf_abi_version.restype = C.c_int
    lib.bf_ctx_create.argtypes = [C.POINTER(BfConfig), C.POINTER(vp)]
    lib.bf_ctx_destroy.argtypes = [vp]
    lib.bf_ctx_destroy.restype = None
    lib.bf_last_error.argtypes = [vp]
    lib.bf_last_error.restype = C.c_char_p
    lib.bf_genome_set.argtypes = [vp, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(vp), u64p]
    lib.bf_genome_begin.argtypes = [vp, C.c_uint32, C.POINTER(C.c_char_p), u64p]
    lib.bf_genome_write.argtypes = [vp, C.c_uint32, C.c_uint64, vp, C.c_uint64]
    lib.bf_genome_end.argtypes = [vp]
    lib.bf_genome_fill.argtypes = [vp, C.c_uint32, C.c_uint64, C.c_uint64, FILL_FN, vp]
    lib.bf_genome_read.argtypes = [vp, C.c_uint32, C.c_uint64, C.c_uint64, vp]
    lib.bf_genome_image_write.argtypes = [C.c_char_p, C.c_uint32, vp, vp, vp, C.c_uint64, C.c_int64, C.c_uint32]
    lib.bf_genome_image_open.argtypes = [C.c_char_p, C.c_uint64, C.c_int64, C.POINTER(vp)]
    lib.bf_genome_image_info.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint64)]
    lib.bf_genome_image_close.argtypes = [vp]
    lib.bf_genome_image_close.restype = None
    lib.bf_genome_begin_image.argtypes = [vp, C.c_uint32, vp, vp]
    lib.bf_genome_set_image.argtypes = [vp, vp]
    lib.bf_genome_write_image.argtypes = [vp, vp]
    lib.bf_submit.argtypes = [vp, C.POINTER(BfBatch)]
    lib.bf_submit_packed.argtypes = [vp, C.POINTER(BfPackedBatch)]
    lib.bf_submit_packed2.argtypes = [vp, C.POINTER(BfPacked2Batch)]
    lib.bf_host_alloc.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
    lib.bf_host_free.argtypes = [vp, vp]
    lib.bf_submit_wait.argtypes = [vp]
    lib.bf_reserve.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64]
    lib.bf_shard_unique_id.argtypes = [vp]
    lib.bf_shard_init_rccl.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64, vp]
    lib.bf_shard_init_host.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_char_p, C.c_uint64]
    lib.bf_shard_finish.argtypes = [vp, C.POINTER(BfShardResult)]
    lib.bf_shard_result_places.argtypes = [vp, C.POINTER(BfShardResult)]
    lib.bf_shard_format_sv.argtypes = [vp, C.c_uint32, C.POINTER(BfShardResult), vp, C.c_uint64]
    lib.bf_shard_format_sv.restype = C.c_int64
    lib.bf_shard_format_sv_names.argtypes = [C.c_uint32, C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(BfShardResult), vp, C.c_uint64]
    lib.bf_shard_format_sv_names.restype = C.c_int64
    lib.bf_localise.argtypes = [vp]
    lib.bf_clusterise.argtypes = [vp, C.POINTER(BfResult)]
    lib.bf_finish.argtypes = [vp, C.POINTER(BfResult)]
    lib.bf_result_reads.argtypes = [vp, C.POINTER(BfResult)]
    lib.bf_reset.argtypes = [vp]
    lib.bf_sync.argtypes = [vp]
    lib.bf_format_sv.argtypes = [vp, C.POINTER(BfResult), vp, C.c_uint64]
    lib.bf_format_sv.restype = C.c_int64
    lib.bf_render_junctions.argtypes = [vp, C.c_uint64, vp, C.POINTER(vp), C.POINTER(vp)]
    lib.bf_render_junctions_dev.argtypes = [vp, C.c_uint64, vp, vp, vp, C.c_uint64, u64p]
    lib.bf_render_junctions_dev_static.argtypes = [vp, C.c_uint64, vp, vp, vp, C.c_uint64, u64p]
    lib.bf_junction_record_bytes.argtypes = [vp, C.c_uint64]
    lib.bf_junction_record_bytes.restype = C.c_int64
    lib.bf_evidence_fetch.argtypes = [vp, C.POINTER(BfEvidenceView)]
    lib.bf_kernel_times.argtypes = [vp, C.POINTER(C.POINTER(BfKernelTime)), C.POINTER(C.c_uint32)]
    lib.bf_stats.argtypes = [vp, u64p, u64p]
    lib.bf_junction_record_bound.argtypes = [vp]
    lib.bf_junction_record_bound.restype = C.c_uint64
    lib.bf_junction_text.argtypes = [vp, C.c_uint64, vp, C.c_uint64]
    lib.bf_junction_text.restype = C.c_int64
    lib.bf_junction_text_of.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64]
    lib.bf_junction_text_of.restype = C.c_int64
    lib.bf_result_junction_text.argtypes = [C.POINTER(BfResult), vp, C.c_uint64, vp]
    lib.bf_result_junction_text.restype = C.c_int64
    if path is None:
        _lib = lib
    return lib


@dataclass
class Batch:
    """Parsed anchor-hit pairs in input order = bf_batch (what detect.rs:89-103 extracts per read)."""
    seq_bytes: np.ndarray    # uint8
    seq_off: np.ndarray      # uint64, n+1
    frag_bytes: np.ndarray   # uint8
    frag_off: np.ndarray     # uint64, n+1
    hits: np.ndarray         # HIT_DTYPE, n

    @property
    def n(self) -> int:
        return int(self.hits.shape[0])

    def slice(self, a: int, b: int) -> "Batch":
        so, fo = self.seq_off, self.frag_off
        return Batch(self.seq_bytes[int(so[a]):int(so[b])], so[a:b + 1] - so[a],
                     self.frag_bytes[int(fo[a]):int(fo[b])], fo[a:b + 1] - fo[a], self.hits[a:b])

    def seq(self, i: int) -> bytes:
        return self.seq_bytes[int(self.seq_off[i]):int(self.seq_off[i + 1])].tobytes()

    def frag(self, i: int) -> bytes:
        return self.frag_bytes[int(self.frag_off[i]):int(self.frag_off[i + 1])].tobytes()


NIBBLE_CODES = b"=ACMGRSVTWYHKDBN"            # SAMv1 4.2


@dataclass
class PackedBatch:
    """A Batch with the reads as BAM 4-bit base codes = bf_packed_batch (two bases per byte, high nibble first; every
    read starts on a byte)."""
    seq_nibbles: np.ndarray  # uint8
    seq_off: np.ndarray      # uint64 byte offsets, n+1
    seq_len: np.ndarray      # uint16, n
    frag_bytes: np.ndarray
    frag_off: np.ndarray
    hits: np.ndarray

    @property
    def n(self) -> int:
        return int(self.hits.shape[0])


def pack_batch(b: Batch, chunk: int = 1 << 20, threads: int = 0) -> PackedBatch:
    """ASCII reads -> 4-bit codes (what a parser that packs while it copies would produce).  Raises ValueError when a
    read holds a byte outside "=ACMGRSVTWYHKDBN" (such batches go through bf_submit as ASCII)."""
    lut = np.full(256, 255, np.uint8)
    lut[np.frombuffer(NIBBLE_CODES, np.uint8)] = np.arange(16, dtype=np.uint8)
    so = np.ascontiguousarray(b.seq_off, np.uint64).astype(np.int64)
    lens = np.diff(so)
    if lens.size and int(lens.max()) > 65535:
        raise ValueError("a read longer than 65535 bases")
    nbytes = (lens + 1) // 2
    boff = np.zeros(b.n + 1, np.int64)
    np.cumsum(nbytes, out=boff[1:])
    out = np.zeros(int(boff[-1]), np.uint8)
    seq = np.ascontiguousarray(b.seq_bytes, np.uint8)
    if lens.size and int(lens.min()) == int(lens.max()) and int(so[0]) == 0:
        from . import synth                                  # reads of one length: the C helper, when it is built
        fast = synth.fast_lib()
        if fast is not None:
            bad = fast.bf_synth_pack_fixed(out.ctypes.data, seq.ctypes.data, b.n, int(lens[0]))
            if bad:
                raise ValueError("a read holds a byte outside BAM's base alphabet %r" % NIBBLE_CODES)
            return PackedBatch(out, boff.astype(np.uint64), lens.astype(np.uint16), np.ascontiguousarray(b.frag_bytes, np.uint8),
                               np.ascontiguousarray(b.frag_off, np.uint64), np.ascontiguousarray(b.hits, HIT_DTYPE))

    def one(a: int) -> None:
        e = min(b.n, a + chunk)
        codes = lut[seq[int(so[a]):int(so[e])]]
        if codes.size and int(codes.max()) == 255:
            raise ValueError("a read holds a byte outside BAM's base alphabet %r" % NIBBLE_CODES)
        ln = lens[a:e]
        if ln.size and int(ln.min()) == int(ln.max()):                 # one length: a plain reshape
            L = int(ln[0])
            m = codes.reshape(e - a, L)
            if L & 1:
                m = np.concatenate([m, np.zeros((e - a, 1), np.uint8)], axis=1)
            out[int(boff[a]):int(boff[e])] = ((m[:, 0::2] << 4) | m[:, 1::2]).reshape(-1)
        else:
            read = np.repeat(np.arange(e - a, dtype=np.int64), ln)
            k = np.arange(codes.size, dtype=np.int64) - np.repeat(so[a:e] - so[a], ln)
            byte = boff[a:e][read] + (k >> 1)
            hi = (k & 1) == 0
            out[byte[hi]] = codes[hi] << 4
            out[byte[~hi]] |= codes[~hi]

    starts = list(range(0, b.n, chunk))
    if threads <= 0:
        try:
            threads = min(16, len(os.sched_getaffinity(0)))
        except Exception:
            threads = min(16, os.cpu_count() or 1)
    if len(starts) > 1 and threads > 1:              # chunks write disjoint byte ranges (every read starts on a byte)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(min(threads, len(starts))) as ex:
            list(ex.map(one, starts))
    else:
        for a in starts:
            one(a)
    return PackedBatch(out, boff.astype(np.uint64), lens.astype(np.uint16), np.ascontiguousarray(b.frag_bytes, np.uint8),
                       np.ascontiguousarray(b.frag_off, np.uint64), np.ascontiguousarray(b.hits, HIT_DTYPE))


@dataclass
class Packed2Batch:
    """A Batch with the reads at two bits a base = bf_packed2_batch: A C G T = 0 1 2 3, four bases a byte (base k in bits
    2(k & 3).. of byte k >> 2), every read on a byte; the other bases as exceptions (read, position, BAM 4-bit code)."""
    seq_2bit: np.ndarray     # uint8
    seq_off: np.ndarray      # uint64 byte offsets, n+1
    seq_len: np.ndarray      # uint16, n
    exc_read: np.ndarray     # uint32, ascending
    exc_pos: np.ndarray      # uint16
    exc_code: np.ndarray     # uint8
    frag_bytes: np.ndarray
    frag_off: np.ndarray
    hits: np.ndarray

    @property
    def n(self) -> int:
        return int(self.hits.shape[0])


def pack2_batch(pb: PackedBatch, chunk: int = 1 << 18) -> Packed2Batch:
    """4-bit codes -> 2-bit codes + exceptions (what a parser that classifies every base anyway would produce)."""
    n = pb.n
    lens = np.ascontiguousarray(pb.seq_len, np.uint16).astype(np.int64)
    so4 = np.ascontiguousarray(pb.seq_off, np.uint64).astype(np.int64)
    nb2 = (lens + 3) // 4
    off2 = np.zeros(n + 1, np.int64)
    np.cumsum(nb2, out=off2[1:])
    out = np.zeros(int(off2[-1]), np.uint8)
    nib = np.ascontiguousarray(pb.seq_nibbles, np.uint8)
    if n and int(lens.min()) == int(lens.max()) and int(so4[0]) == 0:
        from . import synth                                  # reads of one length, all plain ACGT: the C helper, when it is built
        fast = synth.fast_lib()
        if fast is not None and hasattr(fast, "bf_synth_pack2_fixed") and fast.bf_synth_pack2_fixed(out.ctypes.data, nib.ctypes.data, n, int(lens[0])) == 0:
            return Packed2Batch(out, off2.astype(np.uint64), lens.astype(np.uint16), np.zeros(0, np.uint32), np.zeros(0, np.uint16), np.zeros(0, np.uint8),
                                np.ascontiguousarray(pb.frag_bytes, np.uint8), np.ascontiguousarray(pb.frag_off, np.uint64),
                                np.ascontiguousarray(pb.hits, HIT_DTYPE))
        out[:] = 0
    two = np.zeros(16, np.uint8)
    two[[1, 2, 4, 8]] = [0, 1, 2, 3]
    plain = np.zeros(16, bool)
    plain[[1, 2, 4, 8]] = True
    er, ep, ec = [], [], []
    for a in range(0, n, chunk):
        e = min(n, a + chunk)
        ln = lens[a:e]
        tot = int(ln.sum())
        if tot == 0:
            continue
        read = np.repeat(np.arange(e - a, dtype=np.int64), ln)
        first = np.zeros(e - a, np.int64)
        np.cumsum(ln[:-1], out=first[1:])
        k = np.arange(tot, dtype=np.int64) - first[read]
        byte4 = nib[so4[a:e][read] + (k >> 1)]
        code = np.where((k & 1) == 0, byte4 >> 4, byte4 & 15).astype(np.uint8)
        np.add.at(out, off2[a:e][read] + (k >> 2), (two[code] << (2 * (k & 3)).astype(np.uint8)).astype(np.uint8))
        bad = ~plain[code]
        if bad.any():
            er.append((read[bad] + a).astype(np.uint32)); ep.append(k[bad].astype(np.uint16)); ec.append(code[bad])
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
    return Packed2Batch(out, off2.astype(np.uint64), lens.astype(np.uint16), cat(er, np.uint32), cat(ep, np.uint16), cat(ec, np.uint8),
                        np.ascontiguousarray(pb.frag_bytes, np.uint8), np.ascontiguousarray(pb.frag_off, np.uint64),
                        np.ascontiguousarray(pb.hits, HIT_DTYPE))


@dataclass
class Result:
    """Copy of a bf_result (numpy arrays own their memory)."""
    n_evidence: int
    clusters: np.ndarray      # CLUSTER_DTYPE
    reads: np.ndarray         # READ_DTYPE
    junction_bytes: np.ndarray
    n_reads: int = -1         # supporting reads (== reads.shape[0] unless the table is still on the device)
    cluster_junction_off: Optional[np.ndarray] = None   # uint64[n_clusters + 1], one-GPU results (include/breakfast_hip.h)

    def junction(self, k: int) -> bytes:
        """Junction string of supporting read k (detect.rs:155-162), expanded from its record."""
        return junction_text(self.junction_bytes, int(self.reads[k]["junction_off"]))

    def junction_texts(self) -> Tuple[np.ndarray, np.ndarray]:
        """All junction strings, concatenated in read order -> (uint8 text, uint64 offsets[n+1])."""
        lib = load_library()
        rd = np.ascontiguousarray(self.reads, READ_DTYPE)
        jb = np.ascontiguousarray(self.junction_bytes, np.uint8)
        n = rd.shape[0]
        r = BfResult(self.n_evidence, 0, n, None, _ptr(rd), _ptr(jb), jb.size, None)
        if n == 0 and self.n_reads > 0 and self.cluster_junction_off is not None:       # table left on the device
            cl = np.ascontiguousarray(self.clusters, CLUSTER_DTYPE)
            cjo = np.ascontiguousarray(self.cluster_junction_off, np.uint64)
            n = self.n_reads
            r = BfResult(self.n_evidence, cl.shape[0], n, _ptr(cl), None, _ptr(jb), jb.size, _ptr(cjo))
        off = np.zeros(n + 1, np.uint64)
        total = lib.bf_result_junction_text(C.byref(r), None, 0, off.ctypes.data)
        if total < 0:
            raise BreakfastError(int(total), "malformed junction record")
        text = np.zeros(int(total), np.uint8)
        if total:
            lib.bf_result_junction_text(C.byref(r), text.ctypes.data, int(total), off.ctypes.data)
        return text, off


def junction_text(records, off: int = 0) -> bytes:
    """Expands the junction record that starts at byte `off` of `records` (bytes or a uint8 array; see
    include/breakfast_hip.h, "Junction records") with the library's own host decoder."""
    lib = load_library()
    rec = records if isinstance(records, np.ndarray) else np.frombuffer(records, np.uint8)
    avail = max(int(rec.size) - off, 0)
    ptr = rec.ctypes.data + off if avail else None
    n = lib.bf_junction_text(ptr, avail, None, 0)
    if n < 0:
        raise BreakfastError(int(n), "malformed junction record")
    buf = C.create_string_buffer(int(n))
    lib.bf_junction_text(ptr, avail, buf, n)
    return buf.raw[:n]


def junction_header(records, off: int = 0) -> Tuple[int, int, bool, int]:
    """-> (read length, breakpoint offset, reverse-complemented, stored bytes) of the record at `off`."""
    lib = load_library()
    rec = records if isinstance(records, np.ndarray) else np.frombuffer(records, np.uint8)
    size = lib.bf_junction_record_bytes(rec.ctypes.data + off if rec.size > off else None, max(int(rec.size) - off, 0))
    if size < 0:
        raise BreakfastError(int(size), "malformed junction record")
    head = int(np.frombuffer(rec[off:off + 4].tobytes(), "<u4")[0])
    if head & 3 == 0:                                    # text record: bp in the header word, length in the next one
        return int(np.frombuffer(rec[off + 4:off + 8].tobytes(), "<u4")[0]), (head >> 8) & 0xffff, bool(head & 4), int(size)
    return (head >> 3) & 0x1ff, (head >> 12) & 0x1ff, bool(head & 4), int(size)


def _ptr(a: Optional[np.ndarray]) -> Optional[int]:
    return None if a is None or a.size == 0 else a.ctypes.data


def parse_bowtie_lines(lines: Iterable[bytes], contig_id: Dict[bytes, int]) -> Batch:
    """Host mirror of the consume loop's pairing + parsing, detect.rs:76-103: a `5p:` line is
    remembered; a `3p:` line is used only when its read number equals the remembered one.
    Raises KeyError for a contig the genome lacks (the reference panics, detect.rs:115)."""
    seqs: List[bytes] = []
    frags: List[bytes] = []
    hits: List[Tuple[int, int, int, int, int]] = []
    prev = b""
    prev_read_num = 0
    for line in lines:
        if line.endswith(b"\n"):
            line = line[:-1]
        if line.endswith(b"\r"):
            line = line[:-1]
        read_num = int(line.split(b":")[1])
        if line.startswith(b"5p:"):
            prev, prev_read_num = line, read_num
            continue
        elif line.startswith(b"3p:") and read_num != prev_read_num:
            continue
        info = line.split(b":")
        cols, mcols = prev.split(b"\t"), line.split(b"\t")
        strands = (1 if cols[1] == b"+" else 0) | (2 if mcols[1] == b"+" else 0)
        hits.append((int(cols[3]), int(mcols[3]), contig_id[cols[2]], contig_id[mcols[2]], strands))
        frags.append(info[2])
        seqs.append(info[3])
    return make_batch(seqs, frags, hits)


def make_batch(seqs: Sequence[bytes], frags: Sequence[bytes],
               hits: Sequence[Tuple[int, int, int, int, int]]) -> Batch:
    """Builds a Batch from per-read (seq, frag_id, (pos5, pos3, contig5, contig3, strands))."""
    n = len(seqs)
    so = np.zeros(n + 1, np.uint64)
    fo = np.zeros(n + 1, np.uint64)
    if n:
        so[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
        fo[1:] = np.cumsum([len(f) for f in frags], dtype=np.uint64)
    h = np.zeros(n, HIT_DTYPE)
    for i, (p5, p3, c5, c3, st) in enumerate(hits):
        h[i] = (p5, p3, c5, c3, st, (0, 0, 0))
    return Batch(np.frombuffer(b"".join(seqs), np.uint8).copy(), so,
                 np.frombuffer(b"".join(frags), np.uint8).copy(), fo, h)


def write_genome_image(path: str, names: Sequence[bytes], seqs: Sequence[np.ndarray], src_size: int = 0, src_mtime_ns: int = 0,
                       threads: int = 0) -> None:
    """bf_genome_image_write: host only (no GPU is touched)."""
    lib = load_library()
    n = len(names)
    arrs = [np.ascontiguousarray(np.frombuffer(s, np.uint8) if isinstance(s, (bytes, bytearray)) else s, dtype=np.uint8) for s in seqs]
    c_names = (C.c_char_p * n)(*[bytes(x) for x in names])
    c_seqs = (C.c_void_p * n)(*[a.ctypes.data if a.size else None for a in arrs])
    c_lens = (C.c_uint64 * n)(*[a.size for a in arrs])
    rc = lib.bf_genome_image_write(os.fsencode(path), n, c_names, c_seqs, c_lens, src_size, src_mtime_ns, threads)
    if rc < 0:
        raise BreakfastError(rc, (lib.bf_last_error(None) or b"").decode())


class GenomeImage:
    """bf_genome_image_open / _info / _close."""

    def __init__(self, path: str, src_size: int = 0, src_mtime_ns: int = 0):
        self.lib = load_library()
        self.h = C.c_void_p()
        rc = self.lib.bf_genome_image_open(os.fsencode(path), src_size, src_mtime_ns, C.byref(self.h))
        if rc < 0:
            raise BreakfastError(rc, (self.lib.bf_last_error(None) or b"").decode())
        n, names, lens, up = C.c_uint32(), C.POINTER(C.c_char_p)(), C.POINTER(C.c_uint64)(), C.c_uint64()
        self.lib.bf_genome_image_info(self.h, C.byref(n), C.byref(names), C.byref(lens), C.byref(up))
        self.names = [names[i] for i in range(n.value)]
        self.lens = [int(lens[i]) for i in range(n.value)]
        self.upload_bytes = int(up.value)

    def close(self) -> None:
        if self.h:
            self.lib.bf_genome_image_close(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One bf_ctx = one GPU.  Mirrors the stages of detect::main (detect.rs:39-242)."""

    def __init__(self, anchor_len: int = 30, max_frag_len: int = 5000, min_evidence: int = 2,
                 device: int = -1, flags: int = 0, lib_path: Optional[str] = None, groups: int = 0,
                 host_threads: int = 0, piece_reads: int = 0):
        self.lib = load_library(lib_path)
        cfg = BfConfig(anchor_len, max_frag_len, min_evidence, device, flags, (C.c_uint32 * 3)(groups, host_threads, piece_reads))
        h = C.c_void_p()
        rc = self.lib.bf_ctx_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise BreakfastError(rc, (self.lib.bf_last_error(None) or b"").decode())
        self.h = h
        self.flags = flags
        self._kept_reads: list = []
        self.contig_names: List[bytes] = []

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.bf_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int) -> None:
        if rc < 0:
            raise BreakfastError(rc, (self.lib.bf_last_error(self.h) or b"").decode())

    # -- genome (detect.rs:48-56) --
    def set_genome(self, names: Sequence[bytes], seqs: Sequence[np.ndarray]) -> None:
        n = len(names)
        self.contig_names = [bytes(x) for x in names]
        arrs = [np.ascontiguousarray(np.frombuffer(s, np.uint8) if isinstance(s, (bytes, bytearray)) else s,
                                     dtype=np.uint8) for s in seqs]
        c_names = (C.c_char_p * n)(*self.contig_names)
        c_seqs = (C.c_void_p * n)(*[a.ctypes.data if a.size else None for a in arrs])
        c_lens = (C.c_uint64 * n)(*[a.size for a in arrs])
        self._check(self.lib.bf_genome_set(self.h, n, c_names, c_seqs, c_lens))

    def set_genome_image(self, img: "GenomeImage", begin_first: bool = False) -> None:
        """bf_genome_set_image (begin_first: in its three steps, as a caller does that submits reads meanwhile)."""
        self.contig_names = list(img.names)
        if begin_first:
            n = len(img.names)
            self._check(self.lib.bf_genome_begin_image(self.h, n, (C.c_char_p * n)(*img.names), (C.c_uint64 * n)(*img.lens)))
            self._check(self.lib.bf_genome_write_image(self.h, img.h))
            self._check(self.lib.bf_genome_end(self.h))
        else:
            self._check(self.lib.bf_genome_set_image(self.h, img.h))

    def genome_read(self, contig: int, offset: int, n: int) -> np.ndarray:
        """bf_genome_read: the bytes of a piece of a contig, rebuilt from the packed images."""
        out = np.empty(n, np.uint8)
        self._check(self.lib.bf_genome_read(self.h, contig, offset, n, out.ctypes.data if n else None))
        return out

    def set_genome_in_pieces(self, names: Sequence[bytes], seqs: Sequence[np.ndarray], piece: int = 1 << 20,
                             fill: bool = False, order=None) -> None:
        """bf_genome_begin / bf_genome_write (or, fill=True, bf_genome_fill: the bytes are written by a callback straight
        into the library's staging memory) / bf_genome_end; `order` permutes the pieces (any order is allowed)."""
        n = len(names)
        self.contig_names = [bytes(x) for x in names]
        arrs = [np.ascontiguousarray(np.frombuffer(s, np.uint8) if isinstance(s, (bytes, bytearray)) else s, dtype=np.uint8) for s in seqs]
        c_names = (C.c_char_p * n)(*self.contig_names)
        c_lens = (C.c_uint64 * n)(*[a.size for a in arrs])
        self._check(self.lib.bf_genome_begin(self.h, n, c_names, c_lens))
        pieces = [(c, a) for c in range(n) for a in range(0, arrs[c].size, piece)]
        if order is not None:
            pieces = [pieces[i] for i in order(len(pieces))]
        for c, a in pieces:
            m = min(piece, arrs[c].size - a)
            if fill:
                src = arrs[c][a:a + m]

                def cb(user, dst, nbytes, src=src):
                    C.memmove(dst, src.ctypes.data, nbytes)
                rc = self.lib.bf_genome_fill(self.h, c, a, m, FILL_FN(cb), None)
            else:
                rc = self.lib.bf_genome_write(self.h, c, a, arrs[c][a:a + m].ctypes.data, m)
            if rc != 0:
                break                                   # the message is kept for bf_genome_end
        self._check(self.lib.bf_genome_end(self.h))

    @property
    def contig_id(self) -> Dict[bytes, int]:
        return {n: i for i, n in enumerate(self.contig_names)}

    # -- hot path --
    def submit(self, b) -> None:
        if isinstance(b, PackedBatch):
            return self.submit_packed(b)
        if isinstance(b, Packed2Batch):
            return self.submit_packed2(b)
        arrs = [np.ascontiguousarray(b.seq_bytes, np.uint8), np.ascontiguousarray(b.seq_off, np.uint64),
                np.ascontiguousarray(b.frag_bytes, np.uint8), np.ascontiguousarray(b.frag_off, np.uint64),
                np.ascontiguousarray(b.hits, HIT_DTYPE)]
        cb = BfBatch(b.n, *[a.ctypes.data for a in arrs])
        self._check(self.lib.bf_submit(self.h, C.byref(cb)))
        if self.flags & BF_FLAG_READS_ON_HOST:
            self._kept_reads.append(arrs)                 # the library's records point into these until reset()

    def submit_packed(self, b: PackedBatch) -> None:
        arrs = [np.ascontiguousarray(b.seq_nibbles, np.uint8), np.ascontiguousarray(b.seq_off, np.uint64),
                np.ascontiguousarray(b.seq_len, np.uint16), np.ascontiguousarray(b.frag_bytes, np.uint8),
                np.ascontiguousarray(b.frag_off, np.uint64), np.ascontiguousarray(b.hits, HIT_DTYPE)]
        cb = BfPackedBatch(b.n, *[a.ctypes.data for a in arrs])
        self._check(self.lib.bf_submit_packed(self.h, C.byref(cb)))
        if self.flags & BF_FLAG_READS_ON_HOST:
            self._kept_reads.append(arrs)

    def submit_packed2(self, b: Packed2Batch) -> None:
        arrs = [np.ascontiguousarray(b.seq_2bit, np.uint8), np.ascontiguousarray(b.seq_off, np.uint64),
                np.ascontiguousarray(b.seq_len, np.uint16), np.ascontiguousarray(b.exc_read, np.uint32),
                np.ascontiguousarray(b.exc_pos, np.uint16), np.ascontiguousarray(b.exc_code, np.uint8),
                np.ascontiguousarray(b.frag_bytes, np.uint8), np.ascontiguousarray(b.frag_off, np.uint64),
                np.ascontiguousarray(b.hits, HIT_DTYPE)]
        p = [a.ctypes.data for a in arrs]
        cb = BfPacked2Batch(b.n, p[0], p[1], p[2], int(arrs[3].shape[0]), p[3], p[4], p[5], p[6], p[7], p[8])
        self._check(self.lib.bf_submit_packed2(self.h, C.byref(cb)))

    def host_alloc(self, nbytes: int, dtype=np.uint8) -> np.ndarray:
        """A numpy array over bf_host_alloc memory (pinned; owned by the ctx, valid until close())."""
        dt = np.dtype(dtype)
        p = C.c_void_p()
        self._check(self.lib.bf_host_alloc(self.h, max(int(nbytes), 1), C.byref(p)))
        if nbytes == 0:
            return np.zeros(0, dt)
        return np.frombuffer((C.c_uint8 * int(nbytes)).from_address(p.value), dt)

    def pin_batch(self, b: Batch) -> Batch:
        """Copy of the batch in bf_host_alloc memory: what a parser that writes into pinned buffers holds."""
        def pin(a, dt):
            a = np.ascontiguousarray(a, dt)
            out = self.host_alloc(a.nbytes, dt)
            out[...] = a
            return out
        if isinstance(b, PackedBatch):
            return PackedBatch(pin(b.seq_nibbles, np.uint8), pin(b.seq_off, np.uint64), pin(b.seq_len, np.uint16),
                               pin(b.frag_bytes, np.uint8), pin(b.frag_off, np.uint64), pin(b.hits, HIT_DTYPE))
        if isinstance(b, Packed2Batch):                    # (the exception arrays are copied by the call: they stay where they are)
            return Packed2Batch(pin(b.seq_2bit, np.uint8), pin(b.seq_off, np.uint64), pin(b.seq_len, np.uint16), b.exc_read, b.exc_pos,
                                b.exc_code, pin(b.frag_bytes, np.uint8), pin(b.frag_off, np.uint64), pin(b.hits, HIT_DTYPE))
        return Batch(pin(b.seq_bytes, np.uint8), pin(b.seq_off, np.uint64), pin(b.frag_bytes, np.uint8),
                     pin(b.frag_off, np.uint64), pin(b.hits, HIT_DTYPE))

    def submit_wait(self) -> None:
        self._check(self.lib.bf_submit_wait(self.h))

    def reserve(self, n_reads: int, seq_bytes: int, frag_bytes: int) -> None:
        self._check(self.lib.bf_reserve(self.h, n_reads, seq_bytes, frag_bytes))

    def localise(self) -> None:
        self._check(self.lib.bf_localise(self.h))

    def _wrap(self, r: BfResult, copy: bool = True) -> Result:
        def arr(ptr, n, dt):
            if not ptr or n == 0:
                return np.zeros(0, dt)
            a = np.frombuffer((C.c_uint8 * (n * dt.itemsize)).from_address(ptr), dt)
            return a.copy() if copy else a
        cjo = arr(r.cluster_junction_off, r.n_clusters + 1, np.dtype("<u8")) if r.cluster_junction_off else None
        return Result(int(r.n_evidence), arr(r.clusters, r.n_clusters, CLUSTER_DTYPE),
                      arr(r.reads, r.n_reads, READ_DTYPE),
                      arr(r.junction_bytes, r.junction_size, np.dtype("u1")), n_reads=int(r.n_reads),
                      cluster_junction_off=cjo)

    def clusterise(self, copy: bool = True) -> Result:
        r = BfResult()
        self._check(self.lib.bf_clusterise(self.h, C.byref(r)))
        if not r.reads and r.cluster_junction_off:          # a one-GPU result: the table is still on the device
            self._check(self.lib.bf_result_reads(self.h, C.byref(r)))
        self._last = r
        return self._wrap(r, copy)

    def finish(self, copy: bool = True, reads: bool = True) -> Result:
        """bf_finish; reads=False leaves the read table on the device (Result.reads is empty, the records are
        found through Result.cluster_junction_off), as the C call does until bf_result_reads."""
        r = BfResult()
        self._check(self.lib.bf_finish(self.h, C.byref(r)))
        if reads:
            self._check(self.lib.bf_result_reads(self.h, C.byref(r)))
        self._last = r
        return self._wrap(r, copy)

    def finish_raw(self) -> BfResult:
        """bf_finish without copying the result out (timing loops)."""
        r = BfResult()
        self._check(self.lib.bf_finish(self.h, C.byref(r)))
        self._last = r
        return r

    def reset(self) -> None:
        self._check(self.lib.bf_reset(self.h))
        self._kept_reads = []

    def junction_text_of(self, records, off: int = 0) -> bytes:
        """junction_text() for a record of THIS ctx: also expands the reference records of BF_FLAG_READS_ON_HOST."""
        rec = records if isinstance(records, np.ndarray) else np.frombuffer(records, np.uint8)
        avail = max(int(rec.size) - off, 0)
        ptr = rec.ctypes.data + off if avail else None
        n = self.lib.bf_junction_text_of(self.h, ptr, avail, None, 0)
        self._check(n)
        buf = C.create_string_buffer(int(n))
        self.lib.bf_junction_text_of(self.h, ptr, avail, buf, n)
        return buf.raw[:n]

    def sync(self) -> None:
        self._check(self.lib.bf_sync(self.h))

    def format_sv(self, res: Result, as_array: bool = False):
        """stdout bytes of `breakfast detect` for this result (detect.rs:201,233-240); as_array: a uint8 array instead
        of bytes (no extra copies: the table of 100M reads is 6 GB)."""
        cl = np.ascontiguousarray(res.clusters, CLUSTER_DTYPE)
        rd = np.ascontiguousarray(res.reads, READ_DTYPE)
        jb = np.ascontiguousarray(res.junction_bytes, np.uint8)
        n_reads = rd.shape[0]
        cjo = None
        if n_reads == 0 and res.n_reads > 0 and res.cluster_junction_off is not None:   # table left on the device
            cjo = np.ascontiguousarray(res.cluster_junction_off, np.uint64)
            n_reads = res.n_reads
        r = BfResult(res.n_evidence, cl.shape[0], n_reads, _ptr(cl), _ptr(rd), _ptr(jb), jb.size, _ptr(cjo))
        n = self.lib.bf_format_sv(self.h, C.byref(r), None, 0)
        self._check(n)
        buf = np.empty(int(n) + 1, np.uint8)
        n2 = self.lib.bf_format_sv(self.h, C.byref(r), buf.ctypes.data, n)
        self._check(n2)
        return buf[:n] if as_array else buf[:n].tobytes()

    # -- the sharded path in one call (bf_shard_*) --
    @staticmethod
    def shard_unique_id() -> bytes:
        lib = load_library()
        buf = (C.c_uint8 * 128)()
        rc = lib.bf_shard_unique_id(buf)
        if rc != 0:
            raise BreakfastError(rc, (lib.bf_last_error(None) or b"").decode())
        return bytes(buf)

    def shard_init_rccl(self, rank: int, n_ranks: int, index_base: int, unique_id: bytes) -> None:
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self.lib.bf_shard_init_rccl(self.h, rank, n_ranks, index_base, buf))
        self.n_ranks = n_ranks

    def shard_init_host(self, rank: int, n_ranks: int, index_base: int, name: str, slot_bytes: int = 64 << 20) -> None:
        self._check(self.lib.bf_shard_init_host(self.h, rank, n_ranks, index_base, name.encode(), slot_bytes))
        self.n_ranks = n_ranks

    def shard_finish_raw(self) -> BfShardResult:
        r = BfShardResult()
        self._check(self.lib.bf_shard_finish(self.h, C.byref(r)))
        return r

    def shard_finish(self, places: bool = False) -> ShardResult:
        r = self.shard_finish_raw()
        if places:
            self._check(self.lib.bf_shard_result_places(self.h, C.byref(r)))

        def arr(ptr, n, dt):
            dt = np.dtype(dt)
            if not ptr or n == 0:
                return np.zeros(0, dt)
            return np.frombuffer((C.c_uint8 * (n * dt.itemsize)).from_address(ptr), dt).copy()
        return ShardResult(int(r.n_local), int(r.n_evidence), arr(r.clusters, r.n_clusters, CLUSTER_DTYPE),
                           arr(r.src_rank, r.n_reads, "u1"), arr(r.group_read_begin, r.n_groups + 1, "<u8"), int(r.n_records),
                           arr(r.block_off, r.n_groups * self.n_ranks + 1, "<u8"), arr(r.record_bytes, r.record_size, "u1"),
                           arr(r.places, r.n_reads, PLACE_DTYPE) if places else None,
                           arr(r.record_off, r.n_records + 1 if r.n_records else 0, "<u8") if places else None)

    def shard_format_sv(self, results: Sequence[ShardResult], as_array: bool = False):
        """detect.rs:201,233-240 for the results of ALL ranks (rank order); needs only this ctx's contig names."""
        arr, keep = _shard_result_array(results)
        size = self.lib.bf_shard_format_sv(self.h, len(results), arr, None, 0)
        self._check(size)
        buf = np.empty(int(size) + 1, np.uint8)
        self._check(self.lib.bf_shard_format_sv(self.h, len(results), arr, buf.ctypes.data, size))
        del keep
        return buf[:size] if as_array else buf[:size].tobytes()

    # -- junction strings on request --
    def render_junctions(self, local_index: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        idx = np.ascontiguousarray(local_index, np.uint32)
        pb, po = C.c_void_p(), C.c_void_p()
        self._check(self.lib.bf_render_junctions(self.h, idx.size, _ptr(idx), C.byref(pb), C.byref(po)))
        off = np.frombuffer((C.c_uint64 * (idx.size + 1)).from_address(po.value), np.uint64).copy()
        total = int(off[-1])
        by = (np.frombuffer((C.c_uint8 * total).from_address(pb.value), np.uint8).copy() if total
              else np.zeros(0, np.uint8))
        return by, off

    # device index lists: arguments are raw device pointers (e.g. torch tensors' data_ptr())
    def render_junctions_dev(self, n: int, dev_index: int, dev_off: int, dev_bytes: int = 0, cap: int = 0) -> int:
        """-> total bytes; with dev_bytes == 0 only the n+1 offsets are written (size query)."""
        total = C.c_uint64()
        self._check(self.lib.bf_render_junctions_dev(self.h, n, C.c_void_p(dev_index or None), C.c_void_p(dev_off),
                                                     C.c_void_p(dev_bytes or None), cap, C.byref(total)))
        return int(total.value)

    def render_junctions_dev_static(self, n: int, dev_index: int, dev_off: int, dev_bytes: int = 0, cap: int = 0) -> int:
        """bf_render_junctions_dev_static: the same call on the library's static scratch array (allocates nothing)."""
        total = C.c_uint64()
        self._check(self.lib.bf_render_junctions_dev_static(self.h, n, C.c_void_p(dev_index or None), C.c_void_p(dev_off),
                                                            C.c_void_p(dev_bytes or None), cap, C.byref(total)))
        return int(total.value)

    def junction_record_bound(self) -> int:
        return int(self.lib.bf_junction_record_bound(self.h))

    # -- introspection --
    def evidence(self) -> Dict[str, np.ndarray]:
        v = BfEvidenceView()
        self._check(self.lib.bf_evidence_fetch(self.h, C.byref(v)))
        n, w = int(v.n), int(v.mask_words)

        def arr(ptr, cnt, dt):
            dt = np.dtype(dt)
            if cnt == 0:
                return np.zeros(0, dt)
            return np.frombuffer((C.c_uint8 * (cnt * dt.itemsize)).from_address(ptr), dt).copy()
        return {"class_key": arr(v.class_key, n, "<u8"), "lpos": arr(v.lpos, n, "<i4"), "rpos": arr(v.rpos, n, "<i4"),
                "bp": arr(v.bp, n, "<u2"), "flags": arr(v.flags, n, "<u2"),
                "mask": arr(v.mask, n * w, "<u8").reshape(n, w)}

    def kernel_times(self) -> Dict[str, Tuple[float, int]]:
        """name -> (total ms, launches) since the previous call (HIP events on the ctx stream)."""
        p, n = C.POINTER(BfKernelTime)(), C.c_uint32()
        self._check(self.lib.bf_kernel_times(self.h, C.byref(p), C.byref(n)))
        return {p[i].name.decode(): (float(p[i].ms), int(p[i].launches)) for i in range(n.value)}

    def stats(self) -> Tuple[int, int]:
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self.lib.bf_stats(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)


def _shard_result_array(results: Sequence[ShardResult]):
    n = len(results)
    keep, arr = [], (BfShardResult * n)()
    for i, x in enumerate(results):
        cl = np.ascontiguousarray(x.clusters, CLUSTER_DTYPE); sr = np.ascontiguousarray(x.src_rank, np.uint8)
        gb = np.ascontiguousarray(x.group_read_begin, np.uint64); bo = np.ascontiguousarray(x.block_off, np.uint64)
        rb = np.ascontiguousarray(x.record_bytes, np.uint8)
        keep += [cl, sr, gb, bo, rb]
        arr[i] = BfShardResult(x.n_local, x.n_evidence, cl.shape[0], sr.shape[0], _ptr(cl), _ptr(sr), gb.shape[0] - 1, 0,
                               _ptr(gb), x.n_records, _ptr(bo), _ptr(rb), rb.size, None, None)
    return arr, keep


def shard_format_sv(contig_names: Sequence[bytes], results: Sequence[ShardResult]) -> bytes:
    """bf_shard_format_sv_names: the merge of all ranks' results into the reference's stdout (detect.rs:201,233-240) by a
    process that holds no GPU and no ctx -- pure host code of the library."""
    lib = load_library()
    arr, keep = _shard_result_array(results)
    names = (C.c_char_p * len(contig_names))(*[bytes(n) for n in contig_names])
    size = lib.bf_shard_format_sv_names(len(contig_names), names, len(results), arr, None, 0)
    if size < 0:
        raise BreakfastError(int(size), "bf_shard_format_sv_names: malformed shard results")
    buf = np.empty(int(size) + 1, np.uint8)
    if lib.bf_shard_format_sv_names(len(contig_names), names, len(results), arr, buf.ctypes.data, size) != size:
        raise BreakfastError(-1, "bf_shard_format_sv_names: size changed between calls")
    del keep
    return buf[:size].tobytes()


def detect(fasta: Dict[bytes, bytes], bowtie_lines: Iterable[bytes], anchor_len: int = 30,
           max_frag_len: int = 5000, min_evidence: int = 2, flags: int = 0, packed: bool = False) -> bytes:
    """`breakfast detect` on one GPU: genome dict + bowtie's stdout lines -> .sv bytes (packed: the reads handed over as
    4-bit base codes, bf_submit_packed)."""
    eng = Engine(anchor_len, max_frag_len, min_evidence, flags=flags)
    try:
        names = list(fasta.keys())
        eng.set_genome(names, [np.frombuffer(fasta[n], np.uint8) for n in names])
        b = parse_bowtie_lines(bowtie_lines, eng.contig_id)
        eng.submit(pack_batch(b) if packed else b)
        return eng.format_sv(eng.finish())
    finally:
        eng.close()
