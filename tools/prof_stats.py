#!/usr/bin/env python
"""Per-kernel totals of a rocprofv3 run (its rocpd sqlite output), printed as the --stats CSV used to be: name, calls, total ns,
average ns, share.  usage: prof_stats.py <results.db> [top_n]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for n, c, s, a, mn, mx in rows[:top]:
        print(f'"{n}",{c},{s},{a:.1f},{100.0 * s / tot:.2f},{mn},{mx}')


if __name__ == "__main__":
    main()
