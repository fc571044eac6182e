#!/usr/bin/env python3
"""Per-kernel summary (calls, average us, total ms) from a rocprofv3 `*_results.db` (rocpd sqlite output)."""
import sqlite3
import sys

for path in sys.argv[1:]:
    c = sqlite3.connect(path)
    print("==", path)
    q = ("select name, count(*), avg(end-start)/1000.0, sum(end-start)/1e6, max(grid_x/workgroup_x), max(vgpr_count), max(lds_size) "
         "from kernels group by name order by 4 desc limit 16")
    for r in c.execute(q):
        print(f"{r[0][:86]:86s} n={r[1]:6d} avg={r[2]:8.2f}us tot={r[3]:8.2f}ms wg={r[4]} vgpr={r[5]} lds={r[6]}")
