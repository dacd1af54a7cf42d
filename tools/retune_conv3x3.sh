#!/bin/bash
# re-time the 3x3 stride-1 convolutions of every tuned workload against the halo-patch tilings (tile configs 5 / 21 / 22) and merge the winners
# into imagdressing_amd/gemm_tuning.json (3 % hysteresis); the merged table is also left in gpurun_out/gemm_tuning.json
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
T=imagdressing_amd/gemm_tuning.json
cp $T gpurun_out/gemm_tuning_before.json
run() { timeout 600 python tools/gemm_tune.py --only-conv3x3 "$@" --out gpurun_out/gemm_tuning.json > gpurun_out/retune_$1_$2_$3.log 2>&1 && cp gpurun_out/gemm_tuning.json $T; tail -1 gpurun_out/retune_$1_$2_$3.log; }
run --config 1
run --config 1 --width 512 --height 640
run --config 1 --batch 1
run --config 1 --batch 1 --width 512 --height 640
run --config 3
run --config 5
python - <<'PY'
import json
a = json.load(open("gpurun_out/gemm_tuning_before.json"))["shapes"]; b = json.load(open("imagdressing_amd/gemm_tuning.json"))["shapes"]
ch = {k: (a.get(k), b[k]) for k in b if a.get(k) != b[k]}
print(len(ch), "entries changed")
for k, (x, y) in sorted(ch.items()): print(k, x, "->", y)
PY
