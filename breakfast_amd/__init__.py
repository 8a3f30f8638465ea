"""breakfast_amd: MI355X-native `breakfast detect` hot path (HIP kernels behind a C ABI).

Only what the path needs lives here: `csrc/` (HIP kernels, C ABI, CLI driver), `engine` (ctypes
binding + host mirror of the reference's consume-loop interface), `distributed` (one process per
GPU, exchange over torch.distributed/RCCL) and `synth` (seeded synthetic inputs for BASELINE.json's
configs; no real genome or BAM exists in the build or GPU image).
"""
from .engine import (Batch, BreakfastError, Engine, Result, detect, load_library, make_batch,  # noqa: F401
                     parse_bowtie_lines)
