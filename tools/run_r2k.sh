#!/bin/bash
# round 2, call K: row-resident linear kernel -- tests, microbench (events + rocprofv3 kernel durations)
mkdir -p gpurun_out
R=$PWD
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "row_linear or test_linear" 2>&1 | tail -8 > gpurun_out/r2k_pytest.txt; cat gpurun_out/r2k_pytest.txt
timeout 300 python tools/row_linear_ab.py 2>/dev/null | tee gpurun_out/r2k_row_linear_ab.jsonl
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r2k_prof
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2k_prof -o trace -- python $R/tools/row_linear_ab.py > /dev/null 2> $R/gpurun_out/r2k_rocprof.err
cd $R
DB=$(find gpurun_out/r2k_prof -name "*.db" | head -1)
python tools/rocprof_summary.py $DB gpurun_out/r2k_row_linear_kernel_durations.md
head -14 gpurun_out/r2k_row_linear_kernel_durations.md | cut -c1-200
rm -rf gpurun_out/r2k_prof
