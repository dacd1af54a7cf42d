#!/bin/bash
# halo-patch conv: activation fragments read one tap ahead (under the previous tap's MFMAs); same-box A/B against the shipped library
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
B="--no-secondary --no-geometry-secondary --no-parity --no-latency --no-flops --no-live-traffic --no-cpu-baseline --no-power"
run() { timeout 300 python bench.py --steps 3 --warmup 1 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])"; }
for lib in default prio default prio; do
  if [ $lib = default ]; then unset IMD_LIB_PATH; else export IMD_LIB_PATH=$PWD/imagdressing_amd/libimd_$lib.so; fi
  echo "== $lib"
  [ $lib = prio ] && (timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "halo_patch and not 256" 2>&1 | tail -1)
  PATCH_PROBE_CFGS=5 python tools/patch_probe.py 2>/dev/null | tr ']' '\n' | grep -v "^}" | tr '\n' ' '; echo
  run
done 2>&1 | tee gpurun_out/r3az_patch_setprio.txt
