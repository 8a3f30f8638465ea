"""Process plumbing of multi-GPU runs: one process per GPU.  The sharded path itself is in the
library (bf_shard_*, csrc/shard.hip); what a launcher has to do is start the ranks, hand every rank the 128-byte RCCL
id rank 0 made, and -- for a benchmark -- agree on a clock."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import List, Optional, Sequence


def init_from_env(backend: Optional[str] = None):
    """Initialises torch.distributed from torchrun's environment.  -> (rank, world, local_rank)"""
    import torch
    import torch.distributed as dist
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
    node its GPU hangs off, so that eight ranks copying their results to host memory do not all write
    through one socket.  Returns the node, or None when the machine does not say (then nothing changes)."""
    try:
        import torch
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


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(script: str, argv: Sequence[str], n: int, timeout: Optional[float] = None) -> int:
    """Starts `n` fresh rank processes of `script argv...` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set the way
    torchrun sets them, rendezvous on 127.0.0.1) and waits for them.  The caller must not have touched the GPU: the
    children are new processes of a parent that never does.  Rank 0 inherits stdout (its one JSON line), every rank
    stderr.  When a rank fails the others are ended and its status is returned; 0 when all succeed."""
    port = free_port()
    procs: List[subprocess.Popen] = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, script, *argv], env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    import time
    t0 = time.time()
    status = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 128 - rc
                print("[launch] rank %d ended with status %d; ending the other ranks" % (r, rc), file=sys.stderr, flush=True)
                for q in live:
                    procs[q].terminate()
        if timeout is not None and time.time() - t0 > timeout and live:
            print("[launch] time limit of %.0f s reached; ending %d ranks" % (timeout, len(live)), file=sys.stderr, flush=True)
            for q in live:
                procs[q].kill()
            status = status or 124
        time.sleep(0.05)
    return status
