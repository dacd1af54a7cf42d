#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
O=gpurun_out
(timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "split_k" 2>&1 | tail -6) > $O/r3at_pytest.txt; cat $O/r3at_pytest.txt
B="--no-secondary --no-geometry-secondary --no-parity --no-latency --no-flops --no-live-traffic --no-cpu-baseline --no-power"
run() { timeout 300 python bench.py --steps 3 --warmup 1 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])"; }
echo "== before"; (run; run --width 512 --height 640; run --batch 1) | tee $O/r3at_before.txt
for spec in "--config 1" "--config 1 --width 512 --height 640" "--config 1 --batch 1" "--config 1 --batch 1 --width 512 --height 640"; do
  timeout 900 python tools/gemm_tune.py $spec --merge --keep-margin 0.04 --out $O/gemm_tuning_r3at.json > $O/r3at_tune.log 2>&1; tail -1 $O/r3at_tune.log
  python - <<'P'
import json
new = json.load(open("gpurun_out/gemm_tuning_r3at.json")); cur = json.load(open("imagdressing_amd/gemm_tuning.json"))
ch = sum(1 for k, v in new["shapes"].items() if cur["shapes"].get(k) != v)
cur["shapes"] = new["shapes"]; json.dump(cur, open("imagdressing_amd/gemm_tuning.json", "w"), indent=1)
print("entries changed:", ch, "of", len(new["shapes"]))
P
done
cp imagdressing_amd/gemm_tuning.json $O/gemm_tuning_after_r3at.json
echo "== after"; (run; run --width 512 --height 640; run --batch 1; run) | tee $O/r3at_after.txt
