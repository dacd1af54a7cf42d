#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
for args in "--cfg 5" "--cfg 5 --zero" "--cfg 21" "--cfg 21 --zero" "--cfg 0"; do
  python tools/power_probe.py --what conv $args --seconds 3 2>/dev/null | cut -c1-330
done | tee gpurun_out/r3an_conv_power_probe.jsonl
