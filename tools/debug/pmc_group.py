"""rocprofv3 counter_collection.csv -> mean counter value per (kernel, grid) for kernels matching argv[2]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
g = collections.defaultdict(list)
for r in rows:
    if pat in r["Kernel_Name"]:
        g[(r["Kernel_Name"][:44], r.get("Grid_Size", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(g.items()):
    print(f"{k[0]:44s} grid={k[1]:>8s} {k[2]:32s} n={len(v):4d} mean={sum(v)/len(v):.4g}")
