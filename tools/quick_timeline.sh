#!/bin/bash
# Runs on the GPU box: kernel trace of the SHIPPED configuration (streams as they are) and the timeline of one step.
#   bash tools/quick_timeline.sh <tag> [bench args, e.g. --exchange]
set -u
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 bench.py --steps 4 --warmup 2 --cpu-sample 0 --no-h2h "$@" > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
python3 tools/step_timeline.py $out/trace 2 > $out/timeline.txt
tail -1 $out/timeline.txt
