import torch, json
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for mb in (21, 42, 84, 336, 1344):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device="cuda"); b = torch.randn(n, device="cuda").to(torch.bfloat16)
    us_fill = t(lambda: a.fill_(1.0)); us_copy = t(lambda: a.copy_(b))
    print(json.dumps(dict(mb=mb, fill_us=round(us_fill, 1), fill_tbs=round(mb * 1.048576 / us_fill, 2), copy_us=round(us_copy, 1), copy_rw_tbs=round(2 * mb * 1.048576 / us_copy, 2))))
