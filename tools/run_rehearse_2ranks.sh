#!/bin/bash
# rehearsal of bench.py at N = 2 on ONE GPU: two ranks share device 0 and talk over gloo (the real run uses RCCL, one GPU per rank)
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 1 --warmup 1 --ddim-steps 6 --no-secondary --backend gloo --device 0 2>&1 | grep -v "amdgpu.ids\|^W0\|^\*\*\*" | tail -5
timeout 300 python bench.py --gpus 1 --steps 1 --warmup 1 --ddim-steps 6 --no-cpu-baseline --no-secondary 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
