#!/bin/bash
# same-box A/B of several builds of the library: imagdressing_amd/libimagdressing_hip_<tag>.so ("cur" = the current one)
R=$PWD
TAGS="${@:-prev cur}"
for rep in 1 2; do
  for lib in $TAGS; do
    if [ $lib = cur ]; then unset IMD_LIB_PATH; else export IMD_LIB_PATH=$R/imagdressing_amd/libimagdressing_hip_$lib.so; fi
    timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib $lib', d['ms_per_step'], d['value'], d['roofline']['achieved'])"
  done
done
