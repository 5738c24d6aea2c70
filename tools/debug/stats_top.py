"""top kernels of a rocprofv3 kernel_stats.csv; argv[2] = divide the call counts by this many forward passes"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
div = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    print(f'{r["Name"][:84]:84s} calls={int(r["Calls"]) / div:7.1f} avg={float(r["AverageNs"]) / 1e3:8.1f}us {100 * int(r["TotalDurationNs"]) / tot:5.1f}%  per-fwd={int(r["TotalDurationNs"]) / div / 1e3:8.1f}us')
print(f"total per forward: {tot / div / 1e6:.2f} ms")
