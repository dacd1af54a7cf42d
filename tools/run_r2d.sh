#!/bin/bash
# round 2, call D: LDS-DMA staging variant of the d=40 kernel: parity + A/B
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_processors_gpu.py -q -m gpu -k "attention or hybrid or cache or diffusers" 2>&1 | tail -15 > gpurun_out/r2d_pytest.txt
cat gpurun_out/r2d_pytest.txt
timeout 300 python tools/attn_bench.py --variants 7,9 --iters 40 > gpurun_out/r2d_attn_ab.jsonl 2>&1
cat gpurun_out/r2d_attn_ab.jsonl
timeout 300 python tools/attn_bench.py --variants 7,9 --iters 30 --dtype fp16 > gpurun_out/r2d_attn_ab_f16.jsonl 2>&1
cat gpurun_out/r2d_attn_ab_f16.jsonl
