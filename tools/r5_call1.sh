#!/bin/bash
# round-5 lease 1: probes, the new kernels' parity tests, trajectory parity, fp16 attention A/B, in-situ A/B of the 256-row GEMM, quick bench of both dtypes
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
tools/probes/mfma_subnormal_probe > gpurun_out/r5a_mfma_subnormal_probe.jsonl 2>&1; tail -1 gpurun_out/r5a_mfma_subnormal_probe.jsonl
tools/probes/fp8_step_probe > gpurun_out/r5a_fp8_step_probe.jsonl 2>&1; tail -1 gpurun_out/r5a_fp8_step_probe.jsonl
(timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_dma or split_k or d40 or per_call" 2>&1 | tail -6) | tee gpurun_out/r5a_pytest_kernels.txt
(timeout 400 python -m pytest tests/test_processors_gpu.py -q 2>&1 | tail -6) | tee gpurun_out/r5a_pytest_processors.txt
(timeout 600 python -m pytest tests/test_trajectory_gpu.py -q 2>&1 | tail -12) | tee gpurun_out/r5a_pytest_trajectory.txt
timeout 200 python tools/attn_bench.py --only-l0 --variants 12,13 --dtype fp16 --iters 50 > gpurun_out/r5a_attn_fp16_v12_v13.jsonl 2>&1; tail -3 gpurun_out/r5a_attn_fp16_v12_v13.jsonl
timeout 200 python tools/attn_bench.py --only-l0 --variants 12,13 --dtype bf16 --iters 50 > gpurun_out/r5a_attn_bf16_v12_v13.jsonl 2>&1; tail -3 gpurun_out/r5a_attn_bf16_v12_v13.jsonl
for f in "" "8192,5120,640,1,1,0=30 2048,10240,1280,1,1,0=30 8192,640,2560,1,1,0=30:2" "8192,5120,640,1,1,0=32 2048,10240,1280,1,1,0=32 8192,640,2560,1,1,0=30:1" "8192,5120,640,1,1,0=31 2048,10240,1280,1,1,0=31 8192,640,2560,1,1,0=32:2"; do
  echo "force: $f"
  timeout 300 python tools/insitu_conv.py --steps 6 --taps 1 --top 60 --force $f 2>&1 | grep -E "total_ms|8192,5120,640|2048,10240,1280|8192,640,2560"
done | tee gpurun_out/r5a_insitu_gemm256.txt
bash tools/gpu.sh quick r5a_bf16
bash tools/gpu.sh quick r5a_fp16 --dtype fp16
