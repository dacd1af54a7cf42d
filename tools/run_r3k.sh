#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "halo_patch or groupnorm" 2>&1 | tail -6) > gpurun_out/r3k_pytest.txt; cat gpurun_out/r3k_pytest.txt
timeout 600 python tools/patch_ab.py 2>/dev/null > gpurun_out/r3k_patch_dma_ab.jsonl; cat gpurun_out/r3k_patch_dma_ab.jsonl
B="--no-secondary --no-geometry-secondary --no-parity --no-latency --no-flops --no-live-traffic --no-cpu-baseline --no-power"
for fl in 23 535 23 535; do timeout 300 python bench.py --steps 3 --warmup 1 $B --gemm-flags $fl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gemm-flags $fl', d['ms_per_step'], d['value'])"; done | tee gpurun_out/r3k_e2e_ab.txt
