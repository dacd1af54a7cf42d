#!/bin/bash
# round 2, call F: ragged halo-patch conv + new full-size parity tests (configs[2], configs[4]) + UniPC; tuning of configs 3 / 5 shapes
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_e2e_gpu.py -q -m gpu -k "halo_patch or ipa_lora or inpaint_768 or unipc" 2>&1 | tail -15 > gpurun_out/r2f_pytest.txt
cat gpurun_out/r2f_pytest.txt
for c in 3 5; do
  timeout 200 python tools/configs.py --config $c > gpurun_out/r2f_config${c}_before.json 2>gpurun_out/r2f_config${c}.err; cat gpurun_out/r2f_config${c}_before.json
done
timeout 900 python tools/gemm_tune.py --config 5 --merge --skip-known --quick --iters 6 --out gpurun_out/gemm_tuning_c5.json > gpurun_out/r2f_tune_c5.log 2>&1; tail -2 gpurun_out/r2f_tune_c5.log
cp gpurun_out/gemm_tuning_c5.json imagdressing_amd/gemm_tuning.json
timeout 900 python tools/gemm_tune.py --config 3 --merge --skip-known --quick --iters 6 --out gpurun_out/gemm_tuning_c35.json > gpurun_out/r2f_tune_c3.log 2>&1; tail -2 gpurun_out/r2f_tune_c3.log
cp gpurun_out/gemm_tuning_c35.json imagdressing_amd/gemm_tuning.json
for c in 3 5; do
  timeout 200 python tools/configs.py --config $c > gpurun_out/r2f_config${c}_after.json 2>>gpurun_out/r2f_config${c}.err; cat gpurun_out/r2f_config${c}_after.json
done
