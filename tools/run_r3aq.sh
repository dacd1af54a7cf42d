#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "fused_out_projection" 2>&1 | tail -3)
python tools/fused_proj_bench.py 2>/dev/null | tee gpurun_out/r3ar_fused_proj_microbench.txt
