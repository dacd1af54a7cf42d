#!/bin/bash
# round 2, call L: row-resident linear kernel integrated (tuning table cfg 12 + fused LayerNorm -> attn2.to_q): GPU suite + same-box A/B
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r2l_pytest.txt; cat gpurun_out/r2l_pytest.txt
for v in "--no-row-linear" "" "--no-row-linear" ""; do
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary $v 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(dict(variant='$v' or 'row-linear', ms_per_step=d['ms_per_step'], value=d['value'], attn=d['roofline']['achieved'])))" | tee -a gpurun_out/r2l_e2e_ab.jsonl
done
