#!/bin/bash
# round 2 evidence run (one gpurun call): GPU suite, rocprofv3 kernel trace of the bench command, PMC of the level-0 attention
# launch, same-box A/B of the attention kernels end to end, 2-rank rehearsal on one GPU, default bench line
TAG="${1:-r2z}"
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
R=$PWD
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/${TAG}_pytest.txt; cat gpurun_out/${TAG}_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
bash tools/run_profile.sh ${TAG} 2>&1 | tail -14
bash tools/pmc_attn.sh gpurun_out/pmc_attn_${TAG} attn40 2>&1 | tail -20
bash tools/pmc_cmd.sh gpurun_out/pmc_ff_${TAG} ff_geglu320 python tools/ff_fused_bench.py 2>&1 | tail -18
FF_M=8192 RL_K=640 bash tools/pmc_cmd.sh gpurun_out/pmc_rowk640_${TAG} row_linear_k640 python tools/row_linear_ab.py 2>&1 | tail -18
for q in 2 9; do timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --attn-qw $q 2>/dev/null > gpurun_out/${TAG}_bench_attn${q}.json; cut -c1-330 gpurun_out/${TAG}_bench_attn${q}.json; done
# same-box A/B of the round-2 fusions (row-resident projections / qkv, fused LayerNorm, fused feed-forward) against the tiled kernels
for v in "--no-row-linear --no-fused-ff" ""; do timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(dict(variant='$v' or 'default (all round-2 kernels)', ms_per_step=d['ms_per_step'], value=d['value'], attn=d['roofline']['achieved'])))" | tee -a gpurun_out/${TAG}_fusions_ab.jsonl; done
bash tools/run_rehearse_2ranks.sh > gpurun_out/${TAG}_rehearse_2ranks.txt 2>&1; cat gpurun_out/${TAG}_rehearse_2ranks.txt | cut -c1-400
timeout 900 python bench.py --steps 4 --warmup 1 > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; cat gpurun_out/${TAG}_bench_default.json
