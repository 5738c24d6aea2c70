"""group a rocprofv3 kernel_trace.csv by (kernel, grid) -> calls, mean/min duration"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
g = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    if len(sys.argv) > 2 and not any(k in name for k in sys.argv[2:]):
        continue
    key = (name[:64], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
    g[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{k[0]:64s} grid=({k[1]},{k[2]}) wg={k[3]} calls={len(v):5d} mean={sum(v)/len(v)/1e3:8.2f}us med={v[len(v)//2]/1e3:8.2f} min={v[0]/1e3:8.2f}")
