#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "256_pixel" 2>&1 | tail -5) | tee gpurun_out/r3am_pytest.txt
PATCH_PROBE_CFGS=5,21 PATCH_PROBE_SPLITS=1 python tools/patch_probe.py 2>/dev/null | tr ']' '\n' | tee gpurun_out/r3am_patch2_probe.txt
