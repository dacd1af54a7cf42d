#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
O=gpurun_out
for lib in default nst4 nst5; do
  if [ $lib = default ]; then unset IMD_LIB_PATH; else export IMD_LIB_PATH=$PWD/imagdressing_amd/libimd_$lib.so; fi
  echo "== $lib"; python tools/l3_conv_bench.py --cfgs 0,18 --splits 1 --shapes LIN:8192:5120:640,LIN:32768:1280:1280,LIN:8192:640:2560,LIN:16384:4096:4096 --iters 100 2>/dev/null
done | tee $O/r3ab_ring_depth_big_linears.txt
