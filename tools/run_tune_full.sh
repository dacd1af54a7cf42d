#!/bin/bash
# one gpurun call: GEMM parity for every tile config, full per-shape tuning, bench before/after on the same box
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "linear or conv" 2>&1 | tail -5
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_before.json 2> gpurun_out/bench_before.err
cut -c1-200 gpurun_out/bench_before.json
timeout 1200 python tools/gemm_tune.py --out gpurun_out/gemm_tuning.json > gpurun_out/gemm_tune.log 2>&1
tail -2 gpurun_out/gemm_tune.log
cp gpurun_out/gemm_tuning.json imagdressing_amd/gemm_tuning.json
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_after.json 2> gpurun_out/bench_after.err
cut -c1-200 gpurun_out/bench_after.json
