#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
timeout 600 python tools/power_sampler.py gpurun_out/r3f_power_during_bench.jsonl -- python bench.py --steps 6 --warmup 1 --no-secondary --no-geometry-secondary --no-parity --no-latency --no-flops --no-live-traffic --no-cpu-baseline > gpurun_out/r3f_bench.json 2>/dev/null
python - <<'P'
import json
rows=[json.loads(l) for l in open("gpurun_out/r3f_power_during_bench.jsonl")]
print(len(rows), "samples")
for r in rows[::3]: print(r)
P
cut -c1-400 gpurun_out/r3f_bench.json
