#!/bin/bash
# kernel-trace profile of the default bench command; summary -> gpurun_out/<tag>_kernel_trace_summary.md
TAG="${1:-prof}"
R=$PWD
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG -o trace -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-geometry-secondary --no-parity --no-latency --no-flops --no-live-traffic --no-power > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_rocprof.err
cd $R
DB=$(find gpurun_out/$TAG -name "*.db" | head -1)
python tools/rocprof_summary.py $DB gpurun_out/${TAG}_kernel_trace_summary.md
head -32 gpurun_out/${TAG}_kernel_trace_summary.md
cat gpurun_out/${TAG}_bench_under_rocprof.json
find gpurun_out/$TAG -name "*.db" -size +20000k -delete
