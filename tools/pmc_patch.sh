#!/bin/bash
# PMC counters of the halo-patch conv on the level-0 320 -> 320 shape (separate rocprofv3 passes; --pmc with --kernel-trace only)
OUT="${1:-gpurun_out/pmc_patch}"
R=$PWD
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$OUT/pass$i -o a -- python $R/tools/pmc_patch_conv_workload.py > $R/$OUT/pass$i.out 2>&1
done
cd $R
python tools/pmc_summary.py $OUT conv3x3_patch > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +2000k -delete
