#!/bin/bash
# One round's measurement artifacts on the GPU box: the default bench line, the kernel trace of the same command, and the PMC
# passes of the three kernels the bench's rooflines quote.  usage: tools/profile_round.sh rNN   (writes gpurun_out/prof_rNN/)
set -e
r=$1; out=gpurun_out/prof_$r
export TMPDIR=/tmp
mkdir -p $out
export GOICP_GIT_HEAD=${GOICP_GIT_HEAD:-unknown}
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/${r}_bench.json 2> $out/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/${r}_bench_under_rocprof.json 2> $out/trace.err
cp $(ls $out/trace/*/*kernel_stats.csv | head -1) $out/${r}_bench_kernel_stats.csv
echo "kernel trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_b -- python3 bench.py --steps 20 --warmup 5 --sustain-s 0 --s2-steps 0 --no-cpu --no-e2e --no-icp --no-probe > $out/bounds_only.json 2> $out/trace_b.err
cp $(ls $out/trace_b/*/*kernel_stats.csv | head -1) $out/${r}_bench_bounds_only_kernel_stats.csv
echo "bounds-only trace done"
bash tools/pmc_collect.sh $out/pmc_bunny bounds_kernel --steps 3 --warmup 1 --sustain-s 0 --s2-steps 0 --no-cpu --no-e2e --no-icp --no-probe
python3 tools/pmc_to_profile.py bounds $out/pmc_bunny "bench.py default (bunny N=30379, DT 300^3, 65536 cubes per launch)" > $out/${r}_pmc_bounds_bunny.json
bash tools/pmc_collect.sh $out/pmc_s2 bounds_kernel --workload s2 --steps 2 --warmup 1 --prewarm 2 --sustain-s 0 --s2-steps 0 --no-cpu --no-e2e --no-icp --no-probe
python3 tools/pmc_to_profile.py bounds $out/pmc_s2 "bench.py --workload s2 (synthetic N=M=1e6, DT 512^3 = 537 MB, 65536 cubes per launch)" > $out/${r}_pmc_bounds_s2.json
bash tools/pmc_collect.sh $out/pmc_icp icp_ --steps 1 --warmup 0 --prewarm 0 --sustain-s 0 --s2-steps 0 --no-cpu --no-e2e --no-probe
python3 tools/pmc_to_profile.py icp $out/pmc_icp "bench.py ICP leg (bunny N=30379, M=35947)" > $out/${r}_pmc_icp.json
rm -rf $out/trace $out/trace_b $out/pmc_*/pass*/   # raw rocprofv3 output stays on the box: only the summaries travel back
ls -la $out
