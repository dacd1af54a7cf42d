"""Per-launch view of a rocprofv3 kernel trace (rocpd sqlite): the launches of ONE UNet forward in issue order with duration and the gap to the
previous kernel's end -- where does a kernel's in-situ time differ from its microbench, and after what.
Usage: python tools/rocprof_sequence.py <results.db> [pattern] [forward index] [rows of the per-kernel table, default 14; 0 = all]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else "conv3x3_patch_kernel"
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    start = "start" if "start" in cols else "start_timestamp"
    end = "end" if "end" in cols else "end_timestamp"
    rows = list(c.execute(f"select name, grid_x, grid_y, {start}, {end}, duration from kernels order by {start}"))
    # the timed forwards: find ddim step launches as separators
    seps = [i for i, r in enumerate(rows) if "ddim_cfg_step" in r[0]]
    k = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] else len(seps) // 2
    lo, hi = seps[k - 1] + 1, seps[k]
    print(f"{len(rows)} launches, {len(seps)} DDIM steps; forward {k}: launches {lo}..{hi} ({hi - lo}), {(rows[hi][3] - rows[lo][3]) / 1e3:.1f} us wall, "
          f"{sum(r[5] for r in rows[lo:hi]) / 1e3:.1f} us of kernel time")
    short = lambda n: n.split("::")[-1].split("(")[0][:44]
    agg = {}
    for i in range(lo, hi):
        n, gx, gy, s, e, d = rows[i]
        gap = (s - rows[i - 1][4]) / 1e3
        if pat in n:
            print(f"{i - lo:4d} {short(n):44s} grid {gx}x{gy:<3d} {d / 1e3:7.1f} us  gap {gap:5.1f} us  after {short(rows[i - 1][0])} ({rows[i - 1][5] / 1e3:.1f} us)")
        a = agg.setdefault(short(n), [0, 0.0, 0.0])
        a[0] += 1; a[1] += d / 1e3; a[2] += max(gap, 0.0)
    tot_gap = sum(a[2] for a in agg.values())
    print(f"sum of gaps in front of kernels: {tot_gap:.1f} us")
    top = int(sys.argv[4]) if len(sys.argv) > 4 else 14
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top or None]:
        print(f"  {n:44s} x{a[0]:3d}  {a[1]:8.1f} us  gaps in front {a[2]:6.1f} us")


if __name__ == "__main__":
    main()
