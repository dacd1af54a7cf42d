#!/bin/bash
# round 2, call A: full GPU parity suite (new full-size oracle tests), attention variants A/B, default bench
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 > gpurun_out/r2a_pytest.txt
cat gpurun_out/r2a_pytest.txt
timeout 300 python tools/attn_bench.py --variants 2,5 --iters 30 > gpurun_out/r2a_attn_ab.jsonl 2>&1
cat gpurun_out/r2a_attn_ab.jsonl
timeout 300 python tools/attn_bench.py --variants 2,5 --iters 30 --dtype fp16 >> gpurun_out/r2a_attn_ab.jsonl 2>&1
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err
cat gpurun_out/r2a_bench.json; tail -5 gpurun_out/r2a_bench.err
