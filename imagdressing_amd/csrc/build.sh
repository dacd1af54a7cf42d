#!/bin/bash
# Build libimagdressing_hip.so for gfx950 (cross-compiles without a GPU).  In-tree output so the
# library travels with the repo snapshot to the GPU box.
set -euo pipefail
cd "$(dirname "$0")"
OUT="${IMD_OUT:-../libimagdressing_hip.so}"          # IMD_OUT / IMD_BUILD_DIR: a second build beside the product (e.g. -DIMD_ABLATIONS)
BD="${IMD_BUILD_DIR:-build}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${IMD_EXTRA_FLAGS:-}"
objs=()
pids=()
for f in conv_gemm.hip conv_patch.hip conv_patch2.hip conv_patch3.hip conv_img.hip row_linear.hip row_linear_k640.hip row_linear_k1280.hip row_qkv.hip gemm_dma.hip gemm_dma256.hip ff_fused.hip attention.hip attention_d40.hip attention_d40_fp8.hip norm.hip elementwise.hip; do
  o="$BD/${f%.hip}.o"; mkdir -p "$BD"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ common.h -nt "$o" ] || [ imd_kernels.h -nt "$o" ] || [ gemm_common.h -nt "$o" ] || [ lds_dma.h -nt "$o" ] || [ ../../include/imagdressing_hip.h -nt "$o" ]; then
    rm -f "$o"                      # a failed compile must not leave a stale object for the link step
    hipcc $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
  objs+=("$o")
done
o=$BD/capi.o
if [ ! -f "$o" ] || [ capi.cpp -nt "$o" ] || [ imd_kernels.h -nt "$o" ] || [ gemm_common.h -nt "$o" ] || [ ../../include/imagdressing_hip.h -nt "$o" ]; then
  rm -f "$o"
  hipcc $FLAGS -x hip -c capi.cpp -o "$o" &
  pids+=($!)
fi
objs+=("$o")
for pid in ${pids[@]+"${pids[@]}"}; do wait "$pid" || { echo "build.sh: a compile job failed" >&2; exit 1; }; done
hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$OUT"
echo "built $(realpath $OUT)"
