#!/bin/bash
# PMC counters for one GEMM shape / tile config (separate rocprofv3 passes; --pmc with --kernel-trace only).
# usage (on the GPU box, repo root):  bash tools/pmc_gemm2.sh "L0 lin 320->320" 4 gpurun_out/pmc_xxx
FILTER="${1:-L0 lin 320->320}"
CFG="${2:-4}"
OUT="${3:-gpurun_out/pmc_gemm2}"
R=$PWD
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$OUT/pass$i -o g -- python $R/tools/gemm_bench.py --cfgs $CFG --iters 2 --filter "$FILTER" > $R/$OUT/pass$i.out 2>&1
done
cd $R
python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
grep -l -i "error\|invalid" $OUT/pass*.out | head
find $OUT -name "*.csv" -size +2000k -delete
