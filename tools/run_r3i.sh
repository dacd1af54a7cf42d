#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "groupnorm or halo_patch" 2>&1 | tail -6) > gpurun_out/r3i_pytest.txt; cat gpurun_out/r3i_pytest.txt
(timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_fullsize_gpu.py -q -x -k "unet_forward or pipeline_small_20 or deterministic or batched_equals" 2>&1 | tail -6) > gpurun_out/r3i_pytest_e2e.txt; cat gpurun_out/r3i_pytest_e2e.txt
B="--no-secondary --no-geometry-secondary --no-parity --no-latency --no-flops --no-live-traffic --no-cpu-baseline --no-power"
cat > /tmp/ab.py <<'P'
import sys, json, subprocess
P
for rep in 1 2; do
  for on in 1 0; do
    IMD_FUSED_GN_STATS=$on timeout 300 python bench.py --steps 3 --warmup 1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused_gn_stats=$on', d['ms_per_step'], d['value'])"
  done
done | tee gpurun_out/r3i_gn_stats_e2e_ab.txt
