"""breakfast_amd: MI355X-native `breakfast detect` hot path (HIP kernels behind a C ABI).

Only what the path needs lives here: `csrc/` (HIP kernels, C ABI, CLI driver), `engine` (ctypes
binding + host mirror of the reference's consume-loop interface), `launch` (starting one process per
GPU; the sharded path itself is in the library, csrc/shard.hip) and `synth` (seeded synthetic inputs for
BASELINE.json's configs; no real genome or BAM exists in the build or GPU image).
"""
from .engine import (Batch, BreakfastError, Engine, Result, detect, load_library, make_batch,  # noqa: F401
                     parse_bowtie_lines)
