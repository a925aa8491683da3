"""Per-dispatch kernel durations (in launch order) out of a rocprofv3 result database.  usage: dispatch_times.py x_results.db [name filter]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else "lz4amd"
try:
    rows = list(c.execute("select name, start, end from kernels order by start"))
except sqlite3.Error as e:
    print("tables:", [r[0] for r in c.execute("select name from sqlite_master")]); raise
t0 = rows[0][1] if rows else 0
for name, s, e in rows:
    if flt in name:
        print("%10.3f ms  +%9.3f ms  %s" % ((s - t0) / 1e6, (e - s) / 1e6, name[:50]))
