#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "groupnorm" 2>&1 | tail -4) > gpurun_out/r3j_pytest.txt; cat gpurun_out/r3j_pytest.txt
bash tools/run_profile.sh r3j 2>&1 | tail -40 | cut -c1-220
bash tools/pmc_attn.sh gpurun_out/pmc_attn_r3j attn40 2>&1 | tail -25
