#!/bin/bash
# re-time every plain linear layer (K % 64 == 0) of the five tuned workloads against the 256-row producer / consumer LDS-DMA kernel (tile config 30,
# gemm_dma256.hip) and merge the winners (3 % hysteresis) into imagdressing_amd/gemm_tuning.json; the table before the run is kept beside it
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
T=imagdressing_amd/gemm_tuning.json
cp $T gpurun_out/gemm_tuning_before_r5.json
run() { timeout 500 python tools/gemm_tune.py --only-linear --cands 30 --min-k 64 "$@" --out gpurun_out/gemm_tuning.json > gpurun_out/retune_r5_$2_$3_$4.log 2>&1 && cp gpurun_out/gemm_tuning.json $T; grep -c changed gpurun_out/retune_r5_$2_$3_$4.log; }
run --config 1
run --config 1 --width 512 --height 640
run --config 1 --batch 1
run --config 3
run --config 5
python - <<'PY'
import json
a = json.load(open("gpurun_out/gemm_tuning_before_r5.json"))["shapes"]; b = json.load(open("imagdressing_amd/gemm_tuning.json"))["shapes"]
ch = {k: (a.get(k), b[k]) for k in b if a.get(k) != b[k]}
print(len(ch), "entries changed")
for k, (x, y) in sorted(ch.items()): print(k, x, "->", y)
PY
cp imagdressing_amd/gemm_tuning.json gpurun_out/gemm_tuning_after_r5.json
cat gpurun_out/retune_r5_*.log | grep '^{' > gpurun_out/retune_linear_r5_log.jsonl
