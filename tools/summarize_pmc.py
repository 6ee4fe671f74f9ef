"""Aggregate rocprofv3 counter_collection CSVs per kernel name (mean per dispatch)."""
import csv, glob, os, sys, collections
raw, out = sys.argv[1], sys.argv[2]
for tag in ("fetch", "write", "mfma"):
    files = glob.glob(os.path.join(raw, tag, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("no counter csv for", tag); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    with open(files[0]) as f:
        rd = csv.DictReader(f)
        for r in rd:
            k = r.get("Kernel_Name", "?")
            k = k.split("(")[0].replace(",", ";")
            c = r.get("Counter_Name"); v = float(r.get("Counter_Value", 0) or 0)
            a = acc[k][c]; a[0] += v; a[1] += 1
    with open(os.path.join(out, f"pmc_{tag}.csv"), "w") as f:
        f.write("kernel,counter,dispatches,mean_per_dispatch,total\n")
        for k in sorted(acc):
            for c, (s, n) in acc[k].items():
                f.write(f"{k},{c},{n},{s/n:.6g},{s:.6g}\n")
    print("wrote", tag, len(acc), "kernels")
