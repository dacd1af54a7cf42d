#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -q -x -k "concat or small_20 or ipa_controlnet_small or inpaint_small" 2>&1 | tail -2)
B="--no-secondary --no-geometry-secondary --no-parity --no-latency --no-flops --no-live-traffic --no-cpu-baseline --no-power"
run() { timeout 300 python bench.py --steps 3 --warmup 1 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])"; }
echo "== one launch per skip concat"; run; run
