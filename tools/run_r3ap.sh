#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
O=gpurun_out
(timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "fused_out_projection" 2>&1 | tail -3) | tee $O/r3ap_pytest.txt
B="--no-secondary --no-geometry-secondary --no-parity --no-latency --no-flops --no-live-traffic --no-cpu-baseline --no-power"
run() { timeout 300 python bench.py --steps 3 --warmup 1 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"; }
for f in 0 1 0 1; do
  export IMD_FUSED_OUT_PROJ=$f
  echo "== IMD_FUSED_OUT_PROJ=$f"; run
done 2>&1 | tee $O/r3ap_fused_out_proj_ab2.txt
