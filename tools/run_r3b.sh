#!/bin/bash
# round 3, lease 2: where does the time of the default d = 40 attention kernel (16x16x32 tail + LDS-DMA) go -- timing ablations
# from the -DIMD_ABLATIONS build (WRONG results by construction; product library untouched)
TAG="${1:-r3b}"
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export IMD_LIB_PATH=$PWD/imagdressing_amd/libimagdressing_hip_abl.so
timeout 600 python tools/attn_bench.py --variants 10,40,41,42,43,44,45,46,47,48,9 --iters 20 2>/dev/null > gpurun_out/${TAG}_attn_ablations.jsonl
python - <<'P'
import json
rows=[json.loads(l) for l in open("gpurun_out/r3b_attn_ablations.jsonl") if l.startswith("{")]
for r in rows: print(r["qw"], r["us"], r["tflops"])
P
