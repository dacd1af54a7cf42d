python -m pytest tests/test_kernels_gpu.py -x -q -k "test_linear or conv3x3" 2>&1 | tail -2
python tools/gemm_bench.py --cfgs 4,6,11 --flags 23 --iters 30 --filter "L0" 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['shape'].ljust(28), ' '.join(f'{k[3:-7]}={r[k]}' for k in r if k.endswith('_f23_us')))"
