#!/bin/bash
# CFG-pair de-duplication (conv_in + first resnet once for the two identical halves of the CFG batch): tests, then same-box A/B
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py tests/test_fullsize_gpu.py -q -x -k "concat or pipeline or step_graph or full" 2>&1 | tail -3) | tee gpurun_out/r3ba_pytest.txt
B="--no-secondary --no-geometry-secondary --no-parity --no-latency --no-flops --no-live-traffic --no-cpu-baseline --no-power"
run() { timeout 300 python bench.py --steps 3 --warmup 1 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])"; }
for f in 0 1 0 1; do export IMD_CFG_PAIR_DEDUP=$f; echo "== IMD_CFG_PAIR_DEDUP=$f"; run; done 2>&1 | tee gpurun_out/r3ba_cfg_pair_dedup_ab.txt
