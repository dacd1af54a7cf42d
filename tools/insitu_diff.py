"""Per-shape in-situ A/B of two tuning tables: python tools/insitu_diff.py A.txt B.txt  (outputs of tools/insitu_conv.py --top 400 run under
IMD_GEMM_TUNING=<table A> / <table B> on the same box) -> shapes whose tile config differs, with launches, in-situ us in both, and the total ms moved."""
import json, sys
def load(p):
    d = {}
    for l in open(p):
        if l.startswith('{"key"'):
            r = json.loads(l); d[r["key"]] = r
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
rows = []
for k in a:
    if k in b and (a[k]["cfg"], a[k]["split"]) != (b[k]["cfg"], b[k]["split"]):
        rows.append((b[k]["total_ms"] - a[k]["total_ms"], k, a[k], b[k]))
tot = 0.0
for d, k, x, y in sorted(rows):
    tot += d
    print(f"{k:40s} x{x['launches']:3d}  cfg {x['cfg']:2d}:{x['split']} {x['avg_us']:7.1f} us -> cfg {y['cfg']:2d}:{y['split']} {y['avg_us']:7.1f} us   {d:+.3f} ms")
print(f"sum over changed shapes: {tot:+.3f} ms per traced run")
