"""ctypes access to oracle/liboracle.so (the C++ CPU restatement) -- test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional, Sequence

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_lib = None


def build() -> None:
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        vp, u64 = C.c_void_p, C.c_uint64
        L.oracle_genome_from_fasta.restype = vp
        L.oracle_genome_from_fasta.argtypes = [C.c_char_p, u64]
        L.oracle_genome_new.restype = vp
        L.oracle_genome_add.argtypes = [vp, C.c_char_p, vp, u64]
        L.oracle_genome_free.argtypes = [vp]
        L.oracle_detect_text.restype = vp
        L.oracle_detect_text.argtypes = [vp, C.c_char_p, u64, u64, u64, u64, C.c_int]
        L.oracle_detect_batch.restype = vp
        L.oracle_detect_batch.argtypes = [vp, C.POINTER(C.c_char_p), u64, vp, vp, vp, vp, vp, u64, u64, u64, C.c_int, C.c_int]
        L.oracle_sv.restype = vp
        L.oracle_sv.argtypes = [vp, C.POINTER(u64)]
        for f in ("oracle_n_evidence", "oracle_n_clusters", "oracle_n_support"):
            getattr(L, f).restype = u64
            getattr(L, f).argtypes = [vp]
        for f in ("oracle_ev_bp", "oracle_ev_len", "oracle_ev_flags", "oracle_ev_lpos", "oracle_ev_rpos",
                  "oracle_ev_index", "oracle_ev_signature", "oracle_ev_junction", "oracle_ev_junction_off"):
            getattr(L, f).restype = vp
            getattr(L, f).argtypes = [vp]
        L.oracle_run_free.argtypes = [vp]
        _lib = L
    return _lib


class OracleGenome:
    def __init__(self, names: Sequence[bytes], seqs: Sequence[np.ndarray]):
        self.L = lib()
        self.h = self.L.oracle_genome_new()
        self.names = [bytes(n) for n in names]
        for n, s in zip(self.names, seqs):
            a = np.ascontiguousarray(np.frombuffer(s, np.uint8) if isinstance(s, (bytes, bytearray)) else s, np.uint8)
            self.L.oracle_genome_add(self.h, n, a.ctypes.data if a.size else None, a.size)

    @classmethod
    def from_fasta(cls, text: bytes) -> "OracleGenome":
        self = cls.__new__(cls)
        self.L = lib()
        self.h = self.L.oracle_genome_from_fasta(text, len(text))
        self.names = []
        return self

    def __del__(self):
        try:
            self.L.oracle_genome_free(self.h)
        except Exception:
            pass


class OracleRun:
    """Result of one oracle run: .sv bytes and (optionally) per-read evidence in consume order."""

    def __init__(self, L, h, keep: bool):
        n = C.c_uint64()
        p = L.oracle_sv(h, C.byref(n))
        self.sv = C.string_at(p, n.value) if n.value else b""
        self.n_evidence = int(L.oracle_n_evidence(h))
        self.n_clusters = int(L.oracle_n_clusters(h))
        self.n_support = int(L.oracle_n_support(h))
        if keep:
            m = self.n_evidence

            def arr(fn, cnt, dt):
                dt = np.dtype(dt)
                if cnt == 0:
                    return np.zeros(0, dt)
                return np.frombuffer((C.c_uint8 * (cnt * dt.itemsize)).from_address(getattr(L, fn)(h)), dt).copy()
            self.bp = arr("oracle_ev_bp", m, "<u4")
            self.len = arr("oracle_ev_len", m, "<u4")
            self.flags = arr("oracle_ev_flags", m, "u1")
            self.lpos = arr("oracle_ev_lpos", m, "<u8")
            self.rpos = arr("oracle_ev_rpos", m, "<u8")
            self.index = arr("oracle_ev_index", m, "<u8")
            self.signature = arr("oracle_ev_signature", m * 17, "u1").reshape(m, 17)
            self.junction_off = arr("oracle_ev_junction_off", m + 1, "<u8")
            self.junction = arr("oracle_ev_junction", int(self.junction_off[-1]) if m else 0, "u1")
        L.oracle_run_free(h)

    def junction_of(self, i: int) -> bytes:
        return self.junction[int(self.junction_off[i]):int(self.junction_off[i + 1])].tobytes()


def detect_text(genome: OracleGenome, lines: bytes, anchor_len=30, max_frag_len=5000, min_evidence=2,
                keep: bool = False) -> OracleRun:
    L = lib()
    h = L.oracle_detect_text(genome.h, lines, len(lines), anchor_len, max_frag_len, min_evidence, int(keep))
    return OracleRun(L, h, keep)


def detect_batch(genome: OracleGenome, contig_names: Sequence[bytes], batch, anchor_len=30, max_frag_len=5000,
                 min_evidence=2, keep: bool = False, stage: int = 2) -> OracleRun:
    L = lib()
    names = (C.c_char_p * len(contig_names))(*[bytes(n) for n in contig_names])
    arrs = [np.ascontiguousarray(batch.seq_bytes, np.uint8), np.ascontiguousarray(batch.seq_off, np.uint64),
            np.ascontiguousarray(batch.frag_bytes, np.uint8), np.ascontiguousarray(batch.frag_off, np.uint64),
            np.ascontiguousarray(batch.hits)]
    h = L.oracle_detect_batch(genome.h, names, batch.n, *[a.ctypes.data for a in arrs], anchor_len, max_frag_len,
                              min_evidence, int(keep), stage)
    return OracleRun(L, h, keep)
