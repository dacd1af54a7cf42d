#!/bin/bash
# PMC counters (separate rocprofv3 passes; --pmc with --kernel-trace only) of the kernels matching PATTERN in an arbitrary command.
# usage (GPU box, repo root):  bash tools/pmc_cmd.sh gpurun_out/pmc_xxx <kernel-name pattern> python tools/ff_fused_bench.py
OUT="$1"; PAT="$2"; shift 2
R=$PWD
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS"; do
  i=$((i+1))
  (cd $R && timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$OUT/pass$i -o a -- "$@" > $R/$OUT/pass$i.out 2>&1)
done
cd $R
python tools/pmc_summary.py $OUT "$PAT" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +2000k -delete
