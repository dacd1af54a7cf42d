#!/bin/bash
# round 3: quick A/B of attention kernel variants (parity test of every variant first)
TAG="${1:-r3c}"; VARS="${2:-10,12,13,14}"
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "d40_kernel_variants" 2>&1 | tail -4) > gpurun_out/${TAG}_pytest.txt; cat gpurun_out/${TAG}_pytest.txt
timeout 300 python tools/attn_bench.py --variants $VARS --iters 30 2>/dev/null > gpurun_out/${TAG}_attn_ab.jsonl
timeout 300 python tools/attn_bench.py --variants $VARS --iters 30 --dtype fp16 2>/dev/null > gpurun_out/${TAG}_attn_ab_fp16.jsonl
python - <<P
import json
for f in ("gpurun_out/${TAG}_attn_ab.jsonl", "gpurun_out/${TAG}_attn_ab_fp16.jsonl"):
    print(f)
    for l in open(f):
        if l.startswith("{"):
            r = json.loads(l); print(" ", r["qw"], r["us"], r["tflops"])
P
