"""Does running the two CFG halves as two independent streams hide the per-launch fill / drain of the small kernels?
A dependent chain of (LayerNorm -> K = N linear + residual) x 12 on one stream at M rows, against the same chain at M / 2
rows on each of two streams (separate buffers)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd import ops

dt = torch.bfloat16
dev = torch.device("cuda", 0)

def make(M, C):
    return dict(x=torch.randn(M, C, device=dev).to(dt), y=torch.empty(M, C, dtype=dt, device=dev), n=torch.empty(M, C, dtype=dt, device=dev),
                w=(torch.randn(C, C, device=dev) * C ** -0.5).to(dt), b=torch.randn(C, device=dev) * 0.1,
                g=torch.ones(C, device=dev), be=torch.zeros(C, device=dev))

def chain(s, reps=12):
    x, y = s["x"], s["y"]
    for _ in range(reps):
        ops.layer_norm(x, s["g"], s["be"], out=s["n"])
        ops.linear(s["n"], s["w"], s["b"], res=x, out=y)
        x, y = y, x

for M, C in ((32768, 320), (8192, 640), (2048, 1280)):
    one = make(M, C)
    halves = [make(M // 2, C), make(M // 2, C)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    def run_one():
        chain(one)
    def run_two():
        cur = torch.cuda.current_stream()
        for st, h in zip(streams, halves):
            st.wait_stream(cur)
        # interleave launches so neither stream runs ahead on the host
        xs = [[h["x"], h["y"]] for h in halves]
        for _ in range(12):
            for st, h, xy in zip(streams, halves, xs):
                with torch.cuda.stream(st):
                    ops.layer_norm(xy[0], h["g"], h["be"], out=h["n"])
                    ops.linear(h["n"], h["w"], h["b"], res=xy[0], out=xy[1])
                xy.reverse()
        for st in streams:
            cur.wait_stream(st)
    row = dict(M=M, C=C)
    for name, fn in (("one_stream", run_one), ("two_streams", run_two)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        row[name + "_us_per_pair"] = round(e0.elapsed_time(e1) * 1e3 / 10 / 12, 1)
        row[name + "_host_us_per_pair"] = round((time.perf_counter() - t0) * 1e6 / 10 / 12, 1)
    print(json.dumps(row), flush=True)
