#!/bin/bash
mkdir -p gpurun_out
for v in 3 2 3 2; do
  timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --attn-qw $v 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('attn variant $v', d['ms_per_step'], d['value'], d['roofline']['achieved'])"
done
