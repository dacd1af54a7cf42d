#!/bin/bash
# Build libimagdressing_hip.so for gfx950 (cross-compiles without a GPU).  In-tree output so the
# library travels with the repo snapshot to the GPU box.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libimagdressing_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${IMD_EXTRA_FLAGS:-}"
objs=()
for f in conv_gemm.hip conv_patch.hip attention.hip norm.hip elementwise.hip; do
  o="build/${f%.hip}.o"; mkdir -p build
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ common.h -nt "$o" ] || [ imd_kernels.h -nt "$o" ] || [ gemm_common.h -nt "$o" ] || [ ../../include/imagdressing_hip.h -nt "$o" ]; then
    hipcc $FLAGS -c "$f" -o "$o" &
  fi
  objs+=("$o")
done
o=build/capi.o
if [ ! -f "$o" ] || [ capi.cpp -nt "$o" ] || [ imd_kernels.h -nt "$o" ] || [ gemm_common.h -nt "$o" ] || [ ../../include/imagdressing_hip.h -nt "$o" ]; then
  hipcc $FLAGS -x hip -c capi.cpp -o "$o" &
fi
objs+=("$o")
wait
hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$OUT"
echo "built $(realpath $OUT)"
