"""Template-parameter sweep of the generic hybrid-attention kernel at head dims 80 / 160 (UNet levels 1 / 2 and the mid block of the 512x512 CFG batch).
Variant 0 = the shipped choice (SCHED = 2 since round 6), 1 = the choice before it, 2.. = the others (attention.hip).  Needs a library built with -DIMD_ATTN_SWEEP (IMD_LIB_PATH=imagdressing_amd/libimagdressing_hip_sweep.so): imd_set_tuning(3 / 4, variant)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd import ops
from tools.attn_bench import run

lib = ops.L.load()
for D, N, knob, nv in ((80, 1024, 3, 8), (160, 256, 4, 4), (160, 64, 4, 4)):
    for rep in range(2):
        for v in range(nv):
            ops.L.check(lib.imd_set_tuning(knob, v))
            r = run(D, N, N, 4, 50, torch.bfloat16)
            print(json.dumps(dict(D=D, N=N, variant=v, us=r["us"], tflops=r["tflops"])), flush=True)
