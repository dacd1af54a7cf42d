#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -30 > gpurun_out/r3e_smi_idle.txt
for args in "--variant 10" "--variant 10 --zero" "--variant 9" "--what idle"; do
  timeout 120 python tools/power_probe.py $args --seconds 3 2>/dev/null >> gpurun_out/r3e_power_probe.jsonl
done
cat gpurun_out/r3e_smi_idle.txt; cut -c1-1500 gpurun_out/r3e_power_probe.jsonl
