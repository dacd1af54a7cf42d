#!/bin/bash
# same-box A/B of two tuning tables over the tuned workloads: bench.py quick (512x512 and 512x640) and tools/configs.py (configs[2], configs[4]), interleaved twice
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
A=${1:-profiles/r5_tables/gemm_tuning_before_r5.json}; B=${2:-imagdressing_amd/gemm_tuning.json}
QUICK="--no-secondary --no-geometry-secondary --no-parity --no-latency --no-flops --no-live-traffic --no-cpu-baseline --no-power --no-configs"
for rep in 1 2; do for T in $A $B; do
  export IMD_GEMM_TUNING=$T
  a=$(timeout 300 python bench.py --steps 3 --warmup 1 $QUICK 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().splitlines()[-1])['ms_per_step'])")
  b=$(timeout 300 python bench.py --steps 3 --warmup 1 $QUICK --width 512 --height 640 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().splitlines()[-1])['ms_per_step'])")
  c=$(timeout 300 python tools/configs.py --config 3 --steps 10 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().splitlines()[-1])['ms_per_step'])")
  d=$(timeout 300 python tools/configs.py --config 5 --steps 10 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().splitlines()[-1])['ms_per_step'])")
  echo "table=$T  512x512_ms_per_bench_step=$a  512x640=$b  configs2_ms_per_ddim_step=$c  configs4_ms_per_ddim_step=$d"
done; done | tee gpurun_out/r5i_table_ab.txt
