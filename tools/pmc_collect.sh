#!/bin/bash
# Collect the PMC passes of one bench.py command on the GPU box (separate passes: FETCH_SIZE needs 3 of the 4 TCC
# slots; counters never combined with the trace domains gpurun refuses).  usage:
#   tools/pmc_collect.sh <outdir under gpurun_out/> <kernel substring> <bench.py args...>
# Writes <outdir>/pass{1..5}/ (rocprofv3 csv) and <outdir>/summary.json (per-dispatch averages, tools/pmc_summary.py).
set -e
out=$1; kern=$2; shift 2
export TMPDIR=/tmp
mkdir -p "$out"
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
            "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum" \
            "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
	i=$((i+1))
	rocprofv3 --pmc $ctrs --output-format csv -d "$out/pass$i" -- python3 bench.py "$@" > "$out/pass$i.log" 2>&1 || echo "pass $i failed" >> "$out/errors.log"
	echo "pmc pass $i done"
done
python3 tools/pmc_summary.py "$out" "$kern" > "$out/summary.json"
