"""Average the PMC counters of the conv_gemm dispatches found under a tools/gpu.sh pmc output directory."""
import csv, glob, sys, collections
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "conv_gemm_kernel"
acc = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection*.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if pat in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
if "SQ_WAVE_CYCLES" in acc and "SQ_VALU_MFMA_BUSY_CYCLES" in acc:
    print("note: SQ_WAVE_CYCLES counts quad-cycles per wave; SQ_VALU_MFMA_BUSY_CYCLES counts cycles")
