#!/bin/bash
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
rm -f gpurun_out/fp8_attention_error_budget.jsonl
timeout 900 python -m pytest tests/test_attention_fp8_gpu.py -q -m gpu 2>&1 | tail -5
cat gpurun_out/fp8_attention_error_budget.jsonl
timeout 300 python tools/attn_bench.py --variants 9 --N 6912 --fp8 --iters 20 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/attn_bench.py --variants 9 --N 4096 --fp8 --iters 20 2>&1 | grep -v amdgpu.ids
timeout 1200 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -k "inpaint_768" 2>&1 | tail -8
grep inpaint_768 gpurun_out/parity_stats.jsonl
for f in "" "--fp8-attention"; do timeout 200 python tools/configs.py --config 5 $f 2>/dev/null; done
