#!/bin/bash
# Runs on the GPU box (through gpurun): the GPU tests without the full-size file, then two bench lines (one GPU; the sharded
# code path on one GPU) into gpurun_out/<tag>/.   bash tools/quick_gpu.sh <tag> [extra pytest args]
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread --ignore tests/test_gpu_full_size.py "$@" > $out/tests.log 2>&1
rc=$?; tail -4 $out/tests.log
if [ $rc -ne 0 ]; then echo "tests failed ($rc)"; exit $rc; fi
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-h2h > $out/bench.json 2> $out/bench.log || exit 1
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-h2h --exchange > $out/bench_x.json 2> $out/bench_x.log || exit 1
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-h2h --one-stream > $out/bench_1s.json 2> $out/bench_1s.log || exit 1
python - <<PY
import json
for f in ("bench", "bench_x", "bench_1s"):
    d = json.load(open("$out/%s.json" % f)); print(f, d["ms_per_step"], d["stage_ms"])
PY
