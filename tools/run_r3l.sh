#!/bin/bash
# re-time the 3x3 stride-1 convolutions (gather tiles vs halo-patch kernel) now that the halo-patch kernel stages by LDS-DMA
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
O=gpurun_out
cp imagdressing_amd/gemm_tuning.json $O/gemm_tuning_before_r3l.json
for spec in "--config 1" "--config 1 --width 512 --height 640" "--config 1 --batch 1" "--config 1 --batch 1 --width 512 --height 640" "--config 3" "--config 5"; do
  timeout 600 python tools/gemm_tune.py $spec --only-conv3x3 --out $O/gemm_tuning_r3l.json > $O/r3l_tune.log 2>&1; tail -1 $O/r3l_tune.log
  python - <<'P'
import json
new = json.load(open("gpurun_out/gemm_tuning_r3l.json")); cur = json.load(open("imagdressing_amd/gemm_tuning.json"))
ch = sum(1 for k, v in new["shapes"].items() if cur["shapes"].get(k) != v)
cur["shapes"].update(new["shapes"]); json.dump(cur, open("imagdressing_amd/gemm_tuning.json", "w"), indent=1)
print("entries changed:", ch, "of", len(new["shapes"]))
P
done
cp imagdressing_amd/gemm_tuning.json $O/gemm_tuning_after_r3l.json
B="--no-secondary --no-geometry-secondary --no-parity --no-latency --no-flops --no-live-traffic --no-cpu-baseline --no-power"
for a in "" "--width 512 --height 640" "--batch 1"; do timeout 300 python bench.py --steps 3 --warmup 1 $B $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('after $a', d['ms_per_step'], d['value'])"; done | tee $O/r3l_after.txt
