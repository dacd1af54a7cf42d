#!/bin/bash
# round 3: tuning-table entries for the shapes users hit first -- the reference scripts' default geometry (512 x 640) at batch 4
# and batch 1, and 512 x 512 at batch 1 (the reference's literal usage) -- merged into the shipped table; before / after timing
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
O=gpurun_out
B="--no-secondary --no-geometry-secondary --no-parity --no-latency --no-flops --no-live-traffic --no-cpu-baseline"
before() { timeout 300 python bench.py --steps 2 --warmup 1 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])"; }
echo "== before"; before --width 512 --height 640 > $O/r3g_before.txt; before --batch 1 >> $O/r3g_before.txt; before --batch 1 --width 512 --height 640 >> $O/r3g_before.txt; cat $O/r3g_before.txt
cp imagdressing_amd/gemm_tuning.json $O/gemm_tuning_r2.json
timeout 900 python tools/gemm_tune.py --config 1 --width 512 --height 640 --merge --skip-known --out $O/gemm_tuning_g640.json > $O/r3g_tune_g640.log 2>&1; tail -1 $O/r3g_tune_g640.log
cp $O/gemm_tuning_g640.json imagdressing_amd/gemm_tuning.json
timeout 900 python tools/gemm_tune.py --config 1 --batch 1 --merge --skip-known --out $O/gemm_tuning_b1.json > $O/r3g_tune_b1.log 2>&1; tail -1 $O/r3g_tune_b1.log
cp $O/gemm_tuning_b1.json imagdressing_amd/gemm_tuning.json
timeout 900 python tools/gemm_tune.py --config 1 --batch 1 --width 512 --height 640 --merge --skip-known --out $O/gemm_tuning_b1g640.json > $O/r3g_tune_b1g640.log 2>&1; tail -1 $O/r3g_tune_b1g640.log
cp $O/gemm_tuning_b1g640.json imagdressing_amd/gemm_tuning.json
cp imagdressing_amd/gemm_tuning.json $O/gemm_tuning_r3.json
echo "== after"; before --width 512 --height 640 > $O/r3g_after.txt; before --batch 1 >> $O/r3g_after.txt; before --batch 1 --width 512 --height 640 >> $O/r3g_after.txt; before >> $O/r3g_after.txt; cat $O/r3g_after.txt
