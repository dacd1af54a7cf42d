#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
for lib in default tNOLOOP; do
  if [ $lib = default ]; then unset IMD_LIB_PATH; else export IMD_LIB_PATH=$PWD/imagdressing_amd/libimd_$lib.so; fi
  echo "== $lib"; python tools/patch_probe.py 2>/dev/null
done | tee gpurun_out/r3ah_patch_probe3.txt
