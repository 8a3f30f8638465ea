"""Multi-GPU `breakfast detect`: one process per GPU, one bf_ctx per process.

The path shards by contiguous input ranges: rank r holds reads [index_base, index_base + n).
Per-read work (localise) needs no communication.  Clustering needs every read of one breakpoint
class on one rank, so there is exactly one real exchange step:

    localise -> pack records by owner = hash(class) % world  -> ALL-TO-ALL (RCCL over xGMI with the
    "nccl" backend; CPU staging with "gloo")                 -> import -> sort/sweep/dedup on the owner
    -> the owner asks each source rank for the junction strings of its surviving reads (a second,
       small all-to-all pair) because sequences and mismatch masks never leave their source rank.

Only fixed-size records travel in the big exchange (40 B header + padded fragment id).  The engine
object is duck-typed (`localise`, `exchange_local`, `exchange_pack_tensor`, `exchange_import_tensor`,
`clusterise`, `render_junctions`) so the orchestration is testable on CPU with gloo.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from .engine import CLUSTER_DTYPE, READ_DTYPE, Engine, Result


def init_from_env(backend: Optional[str] = None):
    """Initialises torch.distributed from torchrun's environment.  -> (rank, world, local_rank)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def bind_to_gpu_numa_node(local_rank: int) -> Optional[int]:
    """Pins this process (and the pinned result buffers it allocates afterwards) to the CPUs of the NUMA
    node its GPU hangs off, so that eight ranks copying 0.7 GB of results per step to host memory do not
    all write through one socket.  Returns the node, or None when the machine does not say (then nothing
    changes).  One process per GPU is what makes this possible at all."""
    try:
        p = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        allowed = cpus & os.sched_getaffinity(0)
        if not allowed:
            return None
        os.sched_setaffinity(0, allowed)
        return node
    except Exception:
        return None


class HipStageEngine:
    """Adapts breakfast_amd.Engine (device pointers) to torch tensors for the exchange.

    The library works on its own HIP stream and every entry point returns after its GPU work has
    completed; tensors produced by torch (collectives included) are made visible to it by
    synchronising torch's current stream before the call (`_torch_done`)."""

    def __init__(self, eng: Engine, device: torch.device):
        self.eng, self.device = eng, device
        self._send = None

    def _torch_done(self):
        torch.cuda.current_stream(self.device).synchronize()

    def localise(self):
        self.eng.localise()

    def n_local(self) -> int:
        return self.eng.stats()[0]

    def exchange_local(self):
        return self.eng.exchange_local()

    def exchange_pack_tensor(self, max_frag: int, has_sigx: int, world: int):
        rec = Engine.wire_record_bytes(max_frag, has_sigx)
        n = self.n_local()
        send = torch.empty(max(n * rec, 1), dtype=torch.uint8, device=self.device)
        self._torch_done()
        counts, _, rec2 = self.eng.exchange_pack(max_frag, has_sigx, world, send.data_ptr(), send.numel())
        assert rec2 == rec
        return send[:n * rec], counts.astype(np.int64), rec

    def exchange_import_tensor(self, recv: torch.Tensor, src_counts: np.ndarray, bases: np.ndarray):
        if recv.device != self.device:
            recv = recv.to(self.device)
        self._torch_done()     # the all-to-all that filled `recv` ran on torch's stream, not the library's
        self.eng.exchange_import(recv.data_ptr() if recv.numel() else 0, src_counts.astype(np.uint64),
                                 bases.astype(np.uint64))

    def clusterise(self) -> Result:
        return self.eng.clusterise(copy=False)

    def render_junctions(self, local_index: np.ndarray):
        return self.eng.render_junctions(local_index)

    # ---- device-resident junction routing (no PCIe round trips) ----
    def result_sources_tensors(self, res: Result):
        """-> (source rank, source-local index, junction record bytes) of the supporting reads, int32 on the device"""
        nr = res.n_reads if res.n_reads >= 0 else int(res.reads.shape[0])
        sr = torch.empty(nr, dtype=torch.int32, device=self.device)
        si = torch.empty(nr, dtype=torch.int32, device=self.device)
        jl = torch.empty(nr, dtype=torch.int32, device=self.device)
        if nr:
            self._torch_done()
            self.eng.result_sources_dev(sr.data_ptr(), si.data_ptr())
            self.eng.result_junction_sizes_dev(jl.data_ptr())
        return sr, si, jl

    def render_junctions_tensor(self, idx: torch.Tensor):
        """idx: int32 device tensor of local read indices -> (uint8 junction records, int64 offsets[n+1]) on the
        device.  One library call: the buffer is sized by the per-record bound, not by a size query."""
        idx = idx.to(self.device, torch.int32).contiguous()
        n = int(idx.numel())
        off = torch.empty(n + 1, dtype=torch.int64, device=self.device)
        cap = n * self.eng.junction_record_bound()
        by = torch.empty(max(cap, 4), dtype=torch.uint8, device=self.device)
        self._torch_done()
        total = self.eng.render_junctions_dev(n, idx.data_ptr() if n else 0, off.data_ptr(), by.data_ptr(), cap)
        return by[:total], off

    def attach_junctions(self, res: Result, joff: torch.Tensor, text: torch.Tensor) -> Result:
        joff = joff.to(self.device, torch.int64).contiguous()
        text = text.to(self.device).contiguous()
        self._torch_done()
        out = self.eng.attach_junctions_dev(joff.data_ptr() if joff.numel() else 0,
                                            text.data_ptr() if text.numel() else 0, int(text.numel()))
        out.src_rank, out.src_index = res.src_rank, res.src_index
        return out

    # pipelined attach in result order: layout first, then the records piece by piece (the library moves each
    # record to its place and copies the piece to the host on its copy stream)
    fetch_reads = True      # False: leave the read table on the device (bf_result_reads was not asked for)

    def result_layout(self, res: Result, n_pieces: int):
        self._torch_done()
        return self.eng.result_layout_dev(n_pieces)

    def result_requests(self, res: Result, n_pieces: int, n_ranks: int):
        """One library call for the owner's side of the routing -> (request tensor int32[n_reads] ordered by
        (piece, source rank), counts[piece, rank], bytes[piece, rank], piece_read_begin, piece_byte_off)."""
        nr = res.n_reads if res.n_reads >= 0 else int(res.reads.shape[0])
        req = torch.empty(nr, dtype=torch.int32, device=self.device)
        self._torch_done()
        cnt, by, rb, pb = self.eng.result_requests_dev(n_pieces, n_ranks, req.data_ptr() if nr else 0)
        return req, cnt, by, rb, pb

    def attach_records(self, res: Result, piece: int, records: torch.Tensor, read_pos: Optional[torch.Tensor]):
        """read_pos None: the records come in the order result_requests chose."""
        self._torch_done()      # the all-to-all that filled `records` ran on torch's stream
        n = int(records.numel())
        pos = read_pos.data_ptr() if read_pos is not None and read_pos.numel() else 0
        self.eng.attach_records_dev(piece, records.data_ptr() if n else 0, n, pos)

    def attach_finish(self, res: Result) -> Result:
        out = self.eng.attach_finish(reads=self.fetch_reads)
        out.src_rank, out.src_index = res.src_rank, res.src_index
        return out


_TRACE = os.environ.get("BREAKFAST_TRACE", "0") not in ("", "0")


def _trace(what: str, name: str):
    """BREAKFAST_TRACE=1: the same enter/leave lines the library writes, for the collectives (a rank that hangs in a
    library wait makes the OTHER ranks time out inside a collective: the last lines of every rank tell which is which)."""
    if _TRACE:
        import sys
        import time
        r = dist.get_rank() if dist.is_initialized() else 0
        w = dist.get_world_size() if dist.is_initialized() else 1
        print("[bf-trace pid=%d rank=%d/%d t=%.6f] %s %s" % (os.getpid(), r, w, time.monotonic(), what, name),
              file=sys.stderr, flush=True)


def _a2a(out: torch.Tensor, inp: torch.Tensor, out_splits: Sequence[int], in_splits: Sequence[int], group=None):
    """all_to_all_single with CPU staging when the backend cannot take the tensors' device."""
    _trace("enter", "all_to_all %d->%d bytes" % (inp.numel() * inp.element_size(), out.numel() * out.element_size()))
    try:
        _a2a_impl(out, inp, out_splits, in_splits, group)
    finally:
        _trace("leave", "all_to_all")


def _a2a_impl(out: torch.Tensor, inp: torch.Tensor, out_splits: Sequence[int], in_splits: Sequence[int], group=None):
    backend = dist.get_backend(group)
    if backend == "gloo" and inp.is_cuda:
        o = torch.empty(out.shape, dtype=out.dtype)
        _a2a_impl(o, inp.cpu(), out_splits, in_splits, group)
        out.copy_(o)
        return
    if backend == "gloo":
        # gloo has no alltoallv everywhere: use point-to-point pairs (deadlock-free: all isend first)
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        ins = list(torch.split(inp, list(in_splits)))
        outs = list(torch.split(out, list(out_splits)))
        reqs = []
        for r in range(world):
            if r == rank:
                outs[r].copy_(ins[r])
            else:
                if ins[r].numel():
                    reqs.append(dist.isend(ins[r].contiguous(), r, group=group))
        for r in range(world):
            if r != rank and outs[r].numel():
                buf = torch.empty_like(outs[r])
                dist.recv(buf, r, group=group)
                outs[r].copy_(buf)
        for q in reqs:
            q.wait()
        return
    dist.all_to_all_single(out, inp, list(out_splits), list(in_splits), group=group)


@dataclass
class StepStats:
    n_local: int = 0
    n_owned: int = 0
    record_bytes: int = 0
    sent_bytes: int = 0


class DistributedDetect:
    """Runs the sharded hot path.  `step()` is the timed unit; `gather_sv()` is output formatting."""

    def __init__(self, stage, rank: int, world: int, index_base: int, n_local: int, device: torch.device, group=None):
        self.stage, self.rank, self.world, self.group = stage, rank, world, group
        self.device = device
        self.index_base, self.n_local = index_base, n_local
        t = torch.tensor([index_base], dtype=torch.int64)
        if world > 1:
            ts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
            if dist.get_backend(group) == "nccl":
                tg = t.to(device)
                tsg = [x.to(device) for x in ts]
                dist.all_gather(tsg, tg, group=group)
                ts = [x.cpu() for x in tsg]
            else:
                dist.all_gather(ts, t, group=group)
            self.bases = np.array([int(x.item()) for x in ts], np.uint64)
        else:
            self.bases = np.array([index_base], np.uint64)
        self.stats = StepStats()
        self.junction_chunks = 4        # pieces the junction records come in (each piece's host copy hides the next one's
                                        # exchange); the same on every rank: each piece is a round of collectives
        self.chunk_threshold = 0        # (unused: kept for callers that set it)
        self.profile = False            # True: synchronise after every phase and accumulate wall ms in phase_ms
        self.phase_ms = {}
        self._t = None

    def _phase(self, name: Optional[str]):
        """Closes the running phase (if profiling) and opens `name`."""
        if not self.profile:
            return
        import time
        if torch.cuda.is_available() and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        now = time.perf_counter()
        if self._t is not None:
            k, t0 = self._t
            self.phase_ms[k] = self.phase_ms.get(k, 0.0) + (now - t0) * 1e3
        self._t = (name, now) if name else None

    def _ctl(self, values: List[int]) -> torch.Tensor:
        dev = self.device if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        return torch.tensor(values, dtype=torch.int64, device=dev)

    def _exchange_counts(self, counts: np.ndarray) -> np.ndarray:
        inp = self._ctl([int(c) for c in counts])
        out = torch.empty_like(inp)
        ones = [1] * self.world
        _a2a(out, inp, ones, ones, self.group)
        return out.cpu().numpy().astype(np.int64)

    def step(self) -> Result:
        st, W = self.stage, self.world
        self._phase("localise")
        st.localise()
        self._phase("agree")
        mf, sx = st.exchange_local()
        agreed = self._ctl([mf, sx])
        _trace("enter", "all_reduce(max) of the wire parameters")
        dist.all_reduce(agreed, op=dist.ReduceOp.MAX, group=self.group)
        _trace("leave", "all_reduce(max)")
        mf, sx = (int(v) for v in agreed.cpu().tolist())
        self._phase("pack")
        send, counts, rec = st.exchange_pack_tensor(mf, sx, W)
        self._phase("alltoall_records")
        recv_counts = self._exchange_counts(counts)
        recv = torch.empty(int(recv_counts.sum()) * rec, dtype=torch.uint8, device=send.device)
        _a2a(recv, send, [int(c) * rec for c in recv_counts], [int(c) * rec for c in counts], self.group)
        self._phase("import")
        st.exchange_import_tensor(recv, recv_counts, self.bases)
        del recv, send
        self._phase("clusterise")
        res = st.clusterise()
        self.stats = StepStats(int(counts.sum()), int(recv_counts.sum()), rec, int(counts.sum()) * rec)
        out = self._route_junctions(res)
        self._phase(None)
        return out

    # Stage hooks with host fall-backs, so that stand-in engines (tests) only need the numpy calls.
    def _sources(self, res: Result, dev):
        if hasattr(self.stage, "result_sources_tensors"):
            return self.stage.result_sources_tensors(res)
        nr = res.reads.shape[0]
        sr = res.src_rank if nr else np.zeros(0, np.uint32)
        si = res.src_index if nr else np.zeros(0, np.uint32)
        jl = res.junction_sizes if nr else np.zeros(0, np.uint32)
        return (torch.from_numpy(sr.astype(np.int32)).to(dev), torch.from_numpy(si.astype(np.int32)).to(dev),
                torch.from_numpy(jl.astype(np.int32)).to(dev))

    def _render(self, got: torch.Tensor, dev):
        if hasattr(self.stage, "render_junctions_tensor"):
            return self.stage.render_junctions_tensor(got)
        by, off = self.stage.render_junctions(got.cpu().numpy().astype(np.uint32))
        return torch.from_numpy(by).to(dev), torch.from_numpy(off.astype(np.int64)).to(dev)

    def _layout(self, res: Result, K: int):
        """-> (first read, first junction byte) of each of the K result-order pieces, K + 1 entries each."""
        if hasattr(self.stage, "result_layout"):
            return self.stage.result_layout(res, K)
        nr = res.reads.shape[0]
        sizes = res.junction_sizes.astype(np.int64) if nr else np.zeros(0, np.int64)
        joff = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        rb = np.array([nr * c // K for c in range(K + 1)], np.int64)
        res.reads = res.reads.copy()
        res.reads["junction_off"] = joff[:-1].astype(np.uint64)
        res.junction_bytes = np.zeros(int(joff[-1]), np.uint8)
        self._host_layout = (joff, sizes)
        return rb, joff[rb]

    def _attach_records(self, res: Result, piece: int, records: torch.Tensor, read_pos: torch.Tensor):
        if hasattr(self.stage, "attach_records"):
            return self.stage.attach_records(res, piece, records, read_pos)
        joff, sizes = self._host_layout
        m, a = records.cpu().numpy(), 0
        for j in read_pos.cpu().numpy():
            n = int(sizes[j])
            res.junction_bytes[int(joff[j]):int(joff[j]) + n] = m[a:a + n]
            a += n
        assert a == m.shape[0]

    def _attach_finish(self, res: Result) -> Result:
        if hasattr(self.stage, "attach_finish"):
            return self.stage.attach_finish(res)
        return res

    def _exchange_matrix(self, m: np.ndarray) -> np.ndarray:
        """m[r] = K numbers for rank r -> what every rank sent me, same shape."""
        K = m.shape[1]
        inp = self._ctl([int(x) for x in m.ravel()])
        out = torch.empty_like(inp)
        _a2a(out, inp, [K] * self.world, [K] * self.world, self.group)
        return out.cpu().numpy().astype(np.int64).reshape(self.world, K)

    def _route_junctions(self, res: Result) -> Result:
        """Owners request the junction records of their surviving reads from the source ranks (sequences and
        mismatch masks never leave their source rank).  All bulk data (index lists, records) stays in device
        tensors; only per-rank counts and piece boundaries visit the host.

        The owner knows every record's size (it travelled in the exchange record), so the result is laid out as
        on one GPU before any record has arrived: records in printed order.  They come in K pieces -- piece c =
        the c-th K-th of the supporting reads in result order, requested from whichever ranks hold them -- and
        each piece is put in place and copied to host memory behind the rendering and the all-to-all of the
        next one.  Every rank runs the same K rounds, whatever its own share of the result."""
        W = self.world
        nccl = dist.get_backend(self.group) == "nccl"
        self._phase("route_requests")
        K = max(1, min(int(self.junction_chunks), 64))
        if hasattr(self.stage, "result_requests"):
            # the library orders, counts and lays out in one call (one synchronisation)
            req, cnt, gbytes, rb, pb = self.stage.result_requests(res, K, W)
            dev, order = req.device, None
        else:
            src_rank, src_index, jlen = self._sources(res, self.device if nccl else torch.device("cpu"))
            dev = src_rank.device
            nr = int(src_rank.numel())
            rb, pb = self._layout(res, K)
            # requests ordered by (piece, source rank), result order inside: one radix pass when the keys fit a byte
            if K > 1 and nr:
                piece_of = torch.bucketize(torch.arange(nr, device=dev), torch.from_numpy(rb[1:K].copy()).to(dev), right=True)
                key = piece_of * W + src_rank.to(torch.int64)
            else:
                key = src_rank.to(torch.int64)
            order = torch.argsort(key.to(torch.uint8) if K * W <= 256 else key, stable=True)
            cnt = torch.bincount(key, minlength=K * W).cpu().numpy().astype(np.int64).reshape(K, W)   # [piece, source]
            csum = torch.zeros(nr + 1, dtype=torch.int64, device=dev)
            torch.cumsum(jlen[order].to(torch.int64), 0, out=csum[1:])
            cuts = np.concatenate([[0], np.cumsum(cnt.ravel())]).astype(np.int64)
            gbytes = np.diff(csum[torch.from_numpy(cuts).to(dev)].cpu().numpy()).reshape(K, W)   # record bytes per group
            req = src_index[order].contiguous()
        got_cnt = self._exchange_matrix(np.ascontiguousarray(cnt.T))                              # [asking rank, piece]
        cut = np.concatenate([[0], np.cumsum(cnt.ravel())]).astype(np.int64)
        keep = []
        for c in range(K):
            self._phase("render")
            a, b = int(cut[c * W]), int(cut[(c + 1) * W])
            assert (a, b) == (int(rb[c]), int(rb[c + 1])) and int(gbytes[c].sum()) == int(pb[c + 1] - pb[c])
            asked = [int(x) for x in got_cnt[:, c]]
            got = torch.empty(sum(asked), dtype=torch.int32, device=dev)
            _a2a(got, req[a:b], asked, [int(x) for x in cnt[c]], self.group)
            by, off = self._render(got, dev)
            self._phase("alltoall_junctions")
            bounds = np.concatenate([[0], np.cumsum(asked)]).astype(np.int64)
            ob = off[torch.from_numpy(bounds).to(off.device)].cpu().numpy()
            out_bytes = [int(ob[r + 1] - ob[r]) for r in range(W)]
            in_bytes = [int(x) for x in gbytes[c]]
            mine = torch.empty(sum(in_bytes), dtype=torch.uint8, device=by.device)
            _a2a(mine, by[:int(ob[-1])], in_bytes, out_bytes, self.group)
            pos = order[a:b].contiguous() if order is not None else None
            self._attach_records(res, c, mine, pos)
            keep.append((mine, by, pos))
        self._phase("attach_to_host")
        out = self._attach_finish(res)
        del keep
        return out

    # ---- output formatting (not part of the timed path) ----
    def gather_result(self, res: Result, contig_names: Sequence[bytes]) -> Optional[Result]:
        """Rank 0 merges the per-owner cluster lists by the seed's rank in the reference's sort:
        (contig by byte-wise name order, left position as wrapped usize, input index)."""
        payload = (res.n_evidence, res.clusters, res.reads, res.junction_bytes)
        if self.world > 1:
            gathered = [None] * self.world if self.rank == 0 else None
            dist.gather_object(payload, gathered, dst=0, group=self.group)
        else:
            gathered = [payload]
        if self.rank != 0:
            return None
        name_rank = {i: r for r, i in enumerate(sorted(range(len(contig_names)), key=lambda i: contig_names[i]))}
        clusters, reads, jb = [], [], []
        read_base = junc_base = 0
        n_ev = 0
        for (ne, cl, rd, by) in gathered:
            n_ev += ne
            cl, rd = cl.copy(), rd.copy()
            cl["read_begin"] += read_base
            cl["read_end"] += read_base
            rd["junction_off"] += junc_base
            read_base += rd.shape[0]
            junc_base += by.shape[0]
            clusters.append(cl); reads.append(rd); jb.append(by)
        cl = np.concatenate(clusters) if clusters else np.zeros(0, CLUSTER_DTYPE)
        rd = np.concatenate(reads) if reads else np.zeros(0, READ_DTYPE)
        by = np.concatenate(jb) if jb else np.zeros(0, np.uint8)
        key_rank = np.array([name_rank[int(c)] for c in cl["contig"]], np.int64)
        order = np.lexsort((cl["seed_index"], cl["lpos"].astype(np.uint64), key_rank))
        cl = cl[order]
        # re-pack reads in cluster order
        begin = cl["read_begin"].astype(np.int64)
        cnt = cl["read_end"].astype(np.int64) - begin
        new_begin = np.cumsum(cnt) - cnt
        idx = np.repeat(begin - new_begin, cnt) + np.arange(int(cnt.sum()), dtype=np.int64)
        rd = rd[idx] if idx.size else np.zeros(0, READ_DTYPE)
        cl["read_begin"] = new_begin.astype(np.uint64)
        cl["read_end"] = (new_begin + cnt).astype(np.uint64)
        return Result(n_ev, cl, rd, by)


SV_HEADER = (b"CHROM\tSTRAND\tPOSITION\tNEARBY FEATURES\tCHROM\tSTRAND\tPOSITION\tNEARBY FEATURES\t"
             b"SUPPORTING READS\tSIGNATURE\tNOTES\n")


def format_sv(res: Result, contig_names: Sequence[bytes]) -> bytes:
    """The print statements detect.rs:201,233-240 for a merged multi-rank result (pure formatting;
    single-GPU runs use the library's bf_format_sv)."""
    out = [SV_HEADER]
    from .engine import junction_text
    jb = np.ascontiguousarray(res.junction_bytes, np.uint8)
    for c in res.clusters:
        out.append(b"%s\t%s\t%d\t\t%s\t%s\t%d\t\t" % (
            contig_names[int(c["contig"])], b"+" if c["strand"] else b"-", int(c["lpos"]) & ((1 << 64) - 1),
            contig_names[int(c["mcontig"])], b"+" if c["mstrand"] else b"-", int(c["rpos"]) & ((1 << 64) - 1)))
        rows = res.reads[int(c["read_begin"]):int(c["read_end"])]
        out.append(b";".join(junction_text(jb, int(r["junction_off"])) for r in rows))
        out.append(b"\t\t\n")
    return b"".join(out)
