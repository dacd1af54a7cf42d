"""Per-kernel HBM-side traffic from a tools/pmc_traffic_all.sh directory: FETCH_SIZE (KB, x2 on gfx950 for wide streams),
WRITE_SIZE (KB) summed per kernel name + grid, with the kernel durations of the same pass."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
dur = collections.defaultdict(float)
for f in glob.glob(root + "/**/*counter_collection*.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            key = (row["Kernel_Name"][:70], row.get("Grid_Size", ""))
            acc[key][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Counter_Name"] == "FETCH_SIZE":
                cnt[key] += 1
for f in glob.glob(root + "/pass1/**/*kernel_trace*.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            key = (row["Kernel_Name"][:70], row.get("Grid_Size", row.get("Grid_Size_X", "")))
            dur[key] += (float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
tot_f = sum(v.get("FETCH_SIZE", 0) for v in acc.values()); tot_w = sum(v.get("WRITE_SIZE", 0) for v in acc.values())
print(f"total FETCH_SIZE {tot_f/1e6:.2f} GB (x2 = {2*tot_f/1e6:.2f} GB)  WRITE_SIZE {tot_w/1e6:.2f} GB over {sum(cnt.values())} dispatches")
print("| kernel | grid | calls | fetch MB/call (x2) | write MB/call | total GB (2F+W) |")
for key, v in sorted(acc.items(), key=lambda kv: -(2 * kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0)))[:45]:
    n = max(cnt[key], 1)
    print(f"| {key[0]} | {key[1]} | {n} | {2*v.get('FETCH_SIZE',0)/n/1e3:.1f} | {v.get('WRITE_SIZE',0)/n/1e3:.1f} | {(2*v.get('FETCH_SIZE',0)+v.get('WRITE_SIZE',0))/1e6:.2f} |")
