#!/bin/bash
# Counters of the LDS-tile kernel against the gathering kernel on the same deep batch (tools/tile_probe.py, depth 8), and the kernel
# trace of a prove-the-optimum registration (share of bounds_tile_kernel).  usage: tools/tile_pmc.sh <outdir under gpurun_out/>
set -e
out=$1
export TMPDIR=/tmp
mkdir -p $out
i=0
for ctrs in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum"; do
	i=$((i+1))
	rocprofv3 --pmc $ctrs --output-format csv -d $out/pass$i -- python3 tools/tile_probe.py bunny 8 > $out/pass$i.log 2>&1 || echo "pass $i failed" >> $out/errors.log
	echo "pmc pass $i done"
done
python3 tools/pmc_summary.py $out bounds_ > $out/tile_vs_direct_pmc.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/tile_deep.py 3e-5 10,8 > $out/deep_trace.log 2>&1
cp $(ls $out/trace/*/*kernel_stats.csv | head -1) $out/deep_kernel_stats.csv
rm -rf $out/pass*/ $out/trace
ls -la $out
