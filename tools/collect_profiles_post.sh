#!/bin/bash
# gpurun_out/<tag>/ (made by tools/collect_profiles.sh) -> profiles/<tag>_*
set -eu
tag=$1
src=gpurun_out/$tag
for f in $src/bench_*.json; do cp $f profiles/${tag}_$(basename $f); done
python tools/kernel_stats_md.py $src/trace profiles/${tag}_kernel_stats_cfg1_10M.md "rocprofv3 --kernel-trace --stats ($tag), shipped configuration" \
  "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-h2h"
python tools/kernel_stats_md.py $src/trace1 profiles/${tag}_kernel_stats_cfg1_10M_one_stream.md "rocprofv3 --kernel-trace --stats ($tag), every kernel and copy on one stream" \
  "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-h2h --one-stream"
grep -h '^{"metric"' $src/trace.log > profiles/${tag}_bench_cfg1_under_rocprofv3.json || true
grep -h '^{"metric"' $src/trace1.log > profiles/${tag}_bench_cfg1_under_rocprofv3_one_stream.json || true
python tools/pmc_traffic.py $src/pmc_fetch $src/pmc_write profiles/${tag}_pmc_traffic.json \
  "config=1 reads=10000000 len=150 variant=fused reads_as=packed (bench.py --steps 2 --warmup 1), $tag"
