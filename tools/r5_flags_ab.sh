#!/bin/bash
# same-box A/B of two values of tuning knob 2 (bench.py --gemm-flags), interleaved twice:  bash tools/r5_flags_ab.sh TAG V1 V2 [bench args]
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
tag=$1; v1=$2; v2=$3; shift 3
QUICK="--no-secondary --no-geometry-secondary --no-parity --no-latency --no-flops --no-live-traffic --no-cpu-baseline --no-power --no-configs"
for rep in 1 2; do for v in $v1 $v2; do
  timeout 400 python bench.py --steps 3 --warmup 1 $QUICK --gemm-flags $v "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('gemm_flags=$v', d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/${tag}_flags_ab.txt
