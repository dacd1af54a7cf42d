#!/bin/bash
# rocprofv3 --kernel-trace of an arbitrary command on the GPU box: bash tools/prof_cmd.sh TAG cmd...  -> gpurun_out/TAG_kernel_trace_summary.md
R="$(cd "$(dirname "$0")/.." && pwd)"
tag="$1"; shift
mkdir -p $R/gpurun_out/$tag
(cd /tmp && export TMPDIR=/tmp && cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag -o trace -- "$@" > $R/gpurun_out/${tag}_cmd.out 2> $R/gpurun_out/${tag}_rocprof.err)
DB=$(find $R/gpurun_out/$tag -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $DB $R/gpurun_out/${tag}_kernel_trace_summary.md
head -${HEAD:-30} $R/gpurun_out/${tag}_kernel_trace_summary.md
find $R/gpurun_out/$tag -name "*.db" -size +20000k -delete
