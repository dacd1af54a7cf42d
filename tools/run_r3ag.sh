#!/bin/bash
# narrow last channel tile of the halo-patch conv (4 x 1 waves): tests, per-shape probe and end-to-end A/B against the previous library
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
O=gpurun_out
(timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "halo_patch or epilogue_groupnorm or conv3x3" 2>&1 | tail -4) | tee $O/r3ag_pytest.txt
B="--no-secondary --no-geometry-secondary --no-parity --no-latency --no-flops --no-live-traffic --no-cpu-baseline --no-power"
run() { timeout 300 python bench.py --steps 3 --warmup 1 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])"; }
for lib in old new old new; do
  if [ $lib = new ]; then unset IMD_LIB_PATH; else export IMD_LIB_PATH=$PWD/imagdressing_amd/libimd_old.so; fi
  echo "== $lib"; python tools/patch_probe.py 2>/dev/null; run; run --width 512 --height 640
done 2>&1 | tee $O/r3ag_narrow_tile_ab.txt
