#!/bin/bash
# PMC counters for one GEMM shape (separate rocprofv3 passes; --pmc with --kernel-trace only).
# usage (on the GPU box, repo root):  bash tools/pmc_gemm.sh "L1 conv3x3 1280->640" gpurun_out/pmc_xxx
FILTER="${1:-L1 conv3x3 1280->640}"
OUT="${2:-gpurun_out/pmc_r1}"
R=$PWD
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$OUT/pass$i -o g -- python $R/tools/gemm_bench.py --cfgs 0 --iters 2 --filter "$FILTER" > $R/$OUT/pass$i.out 2>&1
done
cd $R
python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +2000k -delete
