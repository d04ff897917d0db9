#!/bin/bash
# rocprofv3 marker + kernel trace of the end-to-end bunny registration with the engine's roctx ranges switched on.
# usage (on the GPU box): tools/marker_trace.sh <outdir under gpurun_out/>
set -e
out=$1
export TMPDIR=/tmp
export GOICP_ROCTX=1
mkdir -p "$out"
rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d "$out" -- python3 bench.py --no-cpu --no-probe --no-icp --steps 2 --warmup 1 --prewarm 0 > "$out/bench.json" 2> "$out/bench.err"
find "$out" -name "*stats*.csv" -o -name "*marker*.csv" | head
