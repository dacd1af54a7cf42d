#!/bin/bash
# out-projection fused into the level-0 attention launches (ABI v7): parity tests, then end-to-end A/B on one box
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
O=gpurun_out
(timeout 1200 python -m pytest tests/test_processors_gpu.py tests/test_e2e_gpu.py -q -x 2>&1 | tail -4) | tee $O/r3ao_pytest.txt
B="--no-secondary --no-geometry-secondary --no-parity --no-latency --no-flops --no-live-traffic --no-cpu-baseline --no-power"
run() { timeout 300 python bench.py --steps 3 --warmup 1 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"; }
for f in 0 1 0 1; do
  export IMD_FUSED_OUT_PROJ=$f
  echo "== IMD_FUSED_OUT_PROJ=$f"; run; run --width 512 --height 640; run --batch 1
done 2>&1 | tee $O/r3ao_fused_out_proj_ab.txt
