"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average duration.
Usage: python tools/rocprof_summary.py <results.db> [out.md]   (rocprofv3 --kernel-trace --stats -d DIR -o NAME)"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = list(c.execute(
        "select name, grid_x, grid_y, grid_z, count(*), sum(duration), avg(duration), min(duration), max(duration), lds_size, "
        "vgpr_count, accum_vgpr_count from kernels group by name, grid_x, grid_y, grid_z order by sum(duration) desc"))
    tot = sum(r[5] for r in rows)
    byname = {}
    for r in rows:
        a = byname.setdefault(r[0], [0, 0])
        a[0] += r[4]; a[1] += r[5]
    out = []
    out.append(f"total kernel time {tot / 1e6:.2f} ms over {sum(r[4] for r in rows)} launches\n")
    out.append("## by kernel\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
    for n, (calls, d) in sorted(byname.items(), key=lambda kv: -kv[1][1])[:25]:
        out.append(f"| `{n[:100]}` | {calls} | {d / 1e6:.2f} | {d / calls / 1e3:.1f} | {100 * d / tot:.1f} |")
    out.append("\n## by kernel and grid (top 40)\n\n| kernel | grid | calls | total ms | avg us | min us | max us | LDS B | VGPR+AGPR | % |\n|---|---|---|---|---|---|---|---|---|---|")
    for r in rows[:40]:
        out.append(f"| `{r[0][:90]}` | {r[1]}x{r[2]}x{r[3]} | {r[4]} | {r[5] / 1e6:.2f} | {r[6] / 1e3:.1f} | {r[7] / 1e3:.1f} | {r[8] / 1e3:.1f} | {r[9]} | "
                   f"{r[10]}+{r[11]} | {100 * r[5] / tot:.1f} |")
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        print(text)


if __name__ == "__main__":
    main()
