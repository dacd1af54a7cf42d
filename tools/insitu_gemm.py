"""In-situ time of every GEMM / conv launch of the denoising loop: run under
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python tools/insitu_gemm.py --dump DIR/gemm_trace.json
then   python tools/insitu_gemm.py --join DIR   matches the i-th traced launch with the i-th conv_gemm / conv3x3_patch
dispatch of the traced (last) pipeline run and prints per-shape in-situ durations next to the table's microbench time."""
import argparse, collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def dump(path, steps):
    import torch
    import bench
    from imagdressing_amd import ops
    dev = torch.device("cuda", 0)
    pipe = bench.build_pipeline(dev, torch.bfloat16, 0)
    inp = bench.synthetic_inputs(512, 512, 4, dev, torch.bfloat16, 0, 1)

    def run():
        return pipe(prompt=None, null_prompt=None, negative_prompt=None, ref_image=None, width=512, height=512,
                    num_inference_steps=steps, guidance_scale=7.5, num_images_per_prompt=4, output_type="latent", **inp).images
    run(); torch.cuda.synchronize()
    ops.GEMM_TRACE = []
    run(); torch.cuda.synchronize()
    tr, ops.GEMM_TRACE = ops.GEMM_TRACE, None
    json.dump(dict(steps=steps, trace=tr), open(path, "w"))


def join(d):
    tr = json.load(open(os.path.join(d, "gemm_trace.json")))
    steps, trace = tr["steps"], tr["trace"]
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace*.csv", recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                n = r["Kernel_Name"]
                if "conv_gemm_kernel" in n or "conv3x3_patch_kernel" in n:
                    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), n))
    rows.sort()
    rows = rows[-len(trace):]
    assert len(rows) == len(trace), (len(rows), len(trace))
    table = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "imagdressing_amd", "gemm_tuning.json")))
    micro = {l["key"]: l for l in table.get("log", [])}
    agg = collections.defaultdict(lambda: [0, 0.0, ""])
    for t, (_, dur, name) in zip(trace, rows):
        key = f"{t['M']},{t['N']},{t['K']},{t['taps']},{t['stride']},{t['ups']}"
        a = agg[key]; a[0] += 1; a[1] += dur / 1e3
        a[2] = "patch" if "patch" in name else name.split("conv_gemm_kernel<")[1].split(">")[0].replace("false, ", "")
    tot = sum(a[1] for a in agg.values())
    print(f"traced run: {steps} DDIM steps + garment pass, {len(trace)} GEMM launches, {tot / 1e3:.2f} ms in GEMM kernels (excluding split-K finish)")
    print("| shape M,N,K,taps,stride,ups | launches | in-situ avg us | TFLOP/s | microbench us | kernel | total ms |")
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        M, N, K = [int(v) for v in key.split(",")[:3]]
        avg = a[1] / a[0]
        mb = micro.get(key, {}).get("best_us", float("nan"))
        print(f"| {key} | {a[0]} | {avg:.1f} | {2.0 * M * N * K / avg / 1e6:.0f} | {mb} | {a[2]} | {a[1] / 1e3:.2f} |")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dump", default="")
    ap.add_argument("--join", default="")
    ap.add_argument("--steps", type=int, default=4)
    a = ap.parse_args()
    if a.dump:
        dump(a.dump, a.steps)
    if a.join:
        join(a.join)
