#!/bin/bash
# round 2, call C: ablations + PMC of the new d=40 attention kernel; spike golden re-check
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_processors_gpu.py -q -m gpu -k "benchmarked or diffusers" 2>&1 | tail -8 > gpurun_out/r2c_pytest.txt
cat gpurun_out/r2c_pytest.txt
timeout 300 python tools/attn_bench.py --variants 7,10,11,12,13,14,15 --iters 30 > gpurun_out/r2c_attn_ablate.jsonl 2>&1
cat gpurun_out/r2c_attn_ablate.jsonl
bash tools/pmc_attn.sh gpurun_out/pmc_attn_r2c attn40 2>&1 | tail -30
