#!/bin/bash
# GroupNorm chunking A/B: default (1024 blocks target) vs A (2048) vs B (512, ppc<=128); same box
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
O=gpurun_out
B="--no-secondary --no-geometry-secondary --no-parity --no-latency --no-flops --no-live-traffic --no-cpu-baseline --no-power"
run() { timeout 300 python bench.py --steps 3 --warmup 1 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])"; }
for lib in default gnB gnC gnD gnE default gnB gnC gnD gnE; do
  if [ $lib = default ]; then unset IMD_LIB_PATH; else export IMD_LIB_PATH=$PWD/imagdressing_amd/libimd_$lib.so; fi
  echo "== $lib"; python tools/norm_bench.py 2>/dev/null | cut -c60-200 | tr "\n" " "; run
done 2>&1 | tee $O/r3z_gn_ab.txt
