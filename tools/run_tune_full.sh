#!/bin/bash
# one gpurun call: full per-shape tuning with the current library, bench before/after on the same box
mkdir -p gpurun_out
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_before.json 2> gpurun_out/bench_before.err
cut -c1-200 gpurun_out/bench_before.json
cp imagdressing_amd/gemm_tuning.json gpurun_out/gemm_tuning_old.json
timeout 1200 python tools/gemm_tune.py --out gpurun_out/gemm_tuning.json > gpurun_out/gemm_tune.log 2>&1
tail -1 gpurun_out/gemm_tune.log
cp gpurun_out/gemm_tuning.json imagdressing_amd/gemm_tuning.json
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_after.json 2> gpurun_out/bench_after.err
cut -c1-200 gpurun_out/bench_after.json
cp gpurun_out/gemm_tuning_old.json imagdressing_amd/gemm_tuning.json
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-200
