#!/bin/bash
# Runs on the GPU box (through gpurun): bench lines, rocprofv3 kernel trace and the two PMC passes of the headline
# workload into gpurun_out/<tag>/.  tools/collect_profiles_post.sh <tag> then turns them into profiles/<tag>_*.
#   bash tools/collect_profiles.sh r01k [full]      ("full" adds configs 2-4 and the sharded code path)
set -u
tag=$1; full=${2:-}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() { echo "== $*" >&2; "$@"; rc=$?; if [ $rc -ne 0 ]; then echo "step failed ($rc): $*" >&2; exit $rc; fi; }
step timeout -k 10 400 python bench.py --steps 20 --warmup 12 --cpu-sample 1000000 --check > $out/bench_cfg1.json 2> $out/bench_cfg1.log
# the profiler changes how the copy stream's blit kernels and the compute kernels share the GPU (its own bench line, kept
# in trace.log, shows it): the kernel trace of the SHIPPED configuration is taken as it is, and a second one with every
# kernel and copy on one stream (--one-stream) gives each kernel's own duration for the per-kernel roofline table
step timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-h2h > $out/trace.log 2>&1
step timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace1 -o t --output-format csv -- python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-h2h --one-stream > $out/trace1.log 2>&1
step timeout -s KILL 280 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-h2h --one-stream > $out/pmc_fetch.log 2>&1
step timeout -s KILL 280 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-h2h --one-stream > $out/pmc_write.log 2>&1
if [ "$full" = full ]; then
  step timeout -k 10 300 python bench.py --steps 10 --warmup 12 --cpu-sample 0 --no-h2h --exchange > $out/bench_cfg1_sharded_path_1gpu.json 2> $out/x.log
  step timeout -k 10 300 python bench.py --steps 10 --warmup 12 --cpu-sample 200000 --check --config 4 > $out/bench_cfg4.json 2> $out/bench_cfg4.log
  step timeout -k 10 500 python bench.py --steps 5 --warmup 10 --cpu-sample 200000 --check --config 3 > $out/bench_cfg3.json 2> $out/bench_cfg3.log
  step timeout -k 10 700 python bench.py --steps 5 --warmup 10 --cpu-sample 200000 --check --config 2 > $out/bench_cfg2.json 2> $out/bench_cfg2.log
fi
echo collected >&2
