#!/bin/bash
# one gpurun call: bench with the shipped table, re-tune the 3x3 convs with the halo-patch kernel as a candidate,
# bench again with the merged table on the same box
mkdir -p gpurun_out
timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_before.json 2> gpurun_out/bench_before.err
cat gpurun_out/bench_before.json
timeout 600 python tools/gemm_tune.py --only-conv3x3 --out gpurun_out/gemm_tuning.json > gpurun_out/gemm_tune.log 2>&1
tail -3 gpurun_out/gemm_tune.log
cp gpurun_out/gemm_tuning.json imagdressing_amd/gemm_tuning.json
timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_after.json 2> gpurun_out/bench_after.err
cat gpurun_out/bench_after.json
