#!/usr/bin/env python
"""Per-(kernel, grid) averages from a rocprofv3 --kernel-trace CSV: kernels that share a template instance (the decode GEMVs of
qkv / o_proj) are told apart by their grid.  usage: prof_by_grid.py <kernel_trace.csv> [top_n]"""
import csv
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: [0, 0])
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name") or r.get("Name")
            grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
            wg = r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or "?"
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            k = (name.split("(")[0][:70], grid, wg)
            acc[k][0] += 1
            acc[k][1] += d
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
    rows = sorted(acc.items(), key=lambda kv: -kv[1][1])[:top]
    tot = sum(v[1] for v in acc.values())
    for (n, g, w), (c, t) in rows:
        print(f"{t / c / 1e3:9.2f} us x {c:6d}  {100.0 * t / tot:5.1f}%  grid={g:>8s} wg={w:>4s}  {n}")


if __name__ == "__main__":
    main()
