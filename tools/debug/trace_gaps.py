"""rocprofv3 kernel_trace.csv -> idle time between consecutive kernels on the GPU (where does a decode step wait?)"""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: r[0])
gaps = {}
tot_gap = tot_busy = 0
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = s1 - e0
    tot_busy += e0 - s0
    if g > 0:
        tot_gap += g
        k = (n0, n1)
        a = gaps.setdefault(k, [0, 0])
        a[0] += 1; a[1] += g
print(f"kernels {len(rows)}  busy {tot_busy/1e6:.2f} ms  idle between kernels {tot_gap/1e6:.2f} ms")
for k, (n, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{k[0]:40s} -> {k[1]:40s} n={n:6d} mean gap {g/n/1e3:7.2f} us total {g/1e6:7.2f} ms")
