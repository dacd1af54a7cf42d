#!/bin/bash
# conv_patch2 with a weight ring of four taps (80 KB, still two workgroups per CU): tests, per-shape probe, 512 x 640 end to end
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
B="--no-secondary --no-geometry-secondary --no-parity --no-latency --no-flops --no-live-traffic --no-cpu-baseline --no-power"
run() { timeout 300 python bench.py --steps 3 --warmup 1 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])"; }
for lib in default ring4 default ring4; do
  if [ $lib = default ]; then unset IMD_LIB_PATH; else export IMD_LIB_PATH=$PWD/imagdressing_amd/libimd_$lib.so; fi
  echo "== $lib"
  [ $lib = ring4 ] && (timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "256_pixel" 2>&1 | tail -1)
  PATCH_PROBE_CFGS=21 PATCH_PROBE_SPLITS=1,3 python tools/patch_probe.py 2>/dev/null | tr ']' '\n' | grep -v "^}" | tr '\n' ' '; echo
  run --width 512 --height 640
done 2>&1 | tee gpurun_out/r3av_patch2_ring4.txt
