#!/bin/bash
OUT="${1:-gpurun_out/insitu}"
R=$PWD
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT -o t -- python $R/tools/insitu_gemm.py --dump $R/$OUT/gemm_trace.json > $R/$OUT/run.out 2>&1
cd $R
python tools/insitu_gemm.py --join $OUT > $OUT/insitu_gemm.md 2>&1
cat $OUT/insitu_gemm.md
find $OUT -name "*.csv" -size +3000k -delete
