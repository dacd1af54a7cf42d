#!/bin/bash
A="${1:-3}"; B="${2:-7}"
for v in $A $B $A $B; do
  timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --gemm-flags $v 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gemm flags $v', d['ms_per_step'], d['value'], d['roofline']['achieved'])"
done
