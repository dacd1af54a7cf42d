#!/bin/bash
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
timeout 300 python tools/attn_bench.py --variants 7,9,16,17,18,19 --iters 30 > gpurun_out/r2e_attn_ablate.jsonl 2>&1
cat gpurun_out/r2e_attn_ablate.jsonl
timeout 300 python tools/attn_bench.py --variants 7,9 --iters 30 --zero > gpurun_out/r2e_attn_zero.jsonl 2>&1
cat gpurun_out/r2e_attn_zero.jsonl
