#!/bin/bash
# one gpurun call: parity of the halo-patch conv, then its A/B against the gather GEMM
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "halo_patch or fused_groupnorm or conv3x3" 2>&1 | tail -15
timeout 300 python tools/patch_bench.py > gpurun_out/patch_bench.jsonl 2> gpurun_out/patch_bench.err
cat gpurun_out/patch_bench.jsonl; tail -5 gpurun_out/patch_bench.err
