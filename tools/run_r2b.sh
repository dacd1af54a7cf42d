#!/bin/bash
# round 2, call B: new d=40 attention kernel: parity (kernel tests + full-size oracle tests) and variants A/B
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_processors_gpu.py -q -m gpu -k "attention or hybrid or cache or diffusers" 2>&1 | tail -25 > gpurun_out/r2b_pytest.txt
cat gpurun_out/r2b_pytest.txt
timeout 300 python tools/attn_bench.py --variants 2,5,6,7,8 --iters 30 > gpurun_out/r2b_attn_ab.jsonl 2>&1
cat gpurun_out/r2b_attn_ab.jsonl
timeout 300 python tools/attn_bench.py --variants 5,7 --iters 30 --dtype fp16 > gpurun_out/r2b_attn_ab_f16.jsonl 2>&1
cat gpurun_out/r2b_attn_ab_f16.jsonl
