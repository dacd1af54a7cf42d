#!/bin/bash
# round 2, call V: + fused norm1 -> q/k/v integrated: tests touching the UNet + same-box A/B
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r2v_pytest.txt; cat gpurun_out/r2v_pytest.txt
for v in "--no-row-linear" "" "--no-row-linear" ""; do
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary $v 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(dict(variant='$v' or 'row kernels (projections, qkv)', ms_per_step=d['ms_per_step'], value=d['value'], attn=d['roofline']['achieved'])))" | tee -a gpurun_out/r2v_e2e_ab.jsonl
done
