#!/bin/bash
# same-box A/B of two tuning tables (tools/_table_old.json vs the shipped one)
cp imagdressing_amd/gemm_tuning.json /tmp/table_new.json
for rep in 1 2; do
  for t in old new; do
    if [ $t = old ]; then cp tools/_table_old.json imagdressing_amd/gemm_tuning.json; else cp /tmp/table_new.json imagdressing_amd/gemm_tuning.json; fi
    timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('table $t', d['ms_per_step'], d['value'])"
  done
done
cp /tmp/table_new.json imagdressing_amd/gemm_tuning.json
