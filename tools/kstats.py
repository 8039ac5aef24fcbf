#!/usr/bin/env python3
"""Per-kernel summary (calls, avg / min / max us) of a rocprofv3 --kernel-trace run: reads the rocpd SQLite file rocprofv3 7.x writes
(<dir>/<name>_results.db) or every such file below a directory.  usage: tools/kstats.py gpurun_out/prof_dir [name-filter]"""
import glob
import os
import sqlite3
import sys


def main():
    root = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else "vdb::"
    files = [root] if os.path.isfile(root) else sorted(glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True))
    for f in files:
        db = sqlite3.connect(f)
        rows = db.execute("select name, count(*), avg(duration), min(duration), max(duration), sum(duration) from kernels "
                          "group by name order by sum(duration) desc").fetchall()
        print(f"# {f}")
        print('"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs"')
        for name, n, avg, mn, mx, tot in rows:
            if flt and flt not in name:
                continue
            print(f'"{name}",{n},{int(tot)},{avg:.1f},{int(mn)},{int(mx)}')


if __name__ == "__main__":
    main()
