#!/bin/bash
# round 3, lease 1: the 16x16x32 P.V tail of the d = 40 attention kernel (parity + A/B), the new golden / fixture tests, the
# step-graph replay, and one default bench line with its new measured fields
TAG="${1:-r3a}"
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out
(timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | tail -5) > $O/${TAG}_pytest_attn_kernels.txt; cat $O/${TAG}_pytest_attn_kernels.txt
(timeout 600 python -m pytest tests/test_processors_gpu.py -q 2>&1 | tail -8) > $O/${TAG}_pytest_processors.txt; cat $O/${TAG}_pytest_processors.txt
(timeout 600 python -m pytest tests/test_fullsize_gpu.py -q -k "attention" 2>&1 | tail -5) > $O/${TAG}_pytest_fullsize_attn.txt; cat $O/${TAG}_pytest_fullsize_attn.txt
for dt in bf16 fp16; do
  timeout 300 python tools/attn_bench.py --variants 9,10 --iters 30 --dtype $dt 2>/dev/null > $O/${TAG}_attn_ab_${dt}.jsonl; cut -c1-120 $O/${TAG}_attn_ab_${dt}.jsonl
done
timeout 300 python tools/attn_bench.py --variants 9,10 --iters 20 --N 5120 2>/dev/null > $O/${TAG}_attn_ab_n5120.jsonl; cut -c1-120 $O/${TAG}_attn_ab_n5120.jsonl
timeout 300 python tools/attn_bench.py --variants 9,10 --iters 20 --zero 2>/dev/null > $O/${TAG}_attn_ab_zero.jsonl; cut -c1-120 $O/${TAG}_attn_ab_zero.jsonl
(timeout 900 python -m pytest tests/test_e2e_gpu.py -q -k "teacher_forced or step_graph or pipeline_small_20" 2>&1 | tail -15) > $O/${TAG}_pytest_e2e_new.txt; cat $O/${TAG}_pytest_e2e_new.txt
timeout 900 python bench.py --steps 2 --warmup 1 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; cat $O/${TAG}_bench.json; tail -5 $O/${TAG}_bench.err
