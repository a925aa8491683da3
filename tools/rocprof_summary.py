#!/usr/bin/env python3
"""Turn a rocprofv3 result database (rocpd sqlite, the default output of ROCm 7.2's
`rocprofv3 --kernel-trace --stats`) into the small text summary committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_x/x_results.db > profiles/rNN_x.txt
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("# %-60s %6s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in c.execute(
            "select name, total_calls, total_duration, average, percentage from top_kernels"):
        short = name if len(name) <= 60 else name[:57] + "..."
        print("  %-60s %6d %14.1f %12.2f %7.2f" % (short, calls, total, avg, pct))
    try:
        rows = list(c.execute(
            "select k.name, e.counter_name, avg(e.value), count(*) from pmc_events e "
            "join kernels k on k.dispatch_id = e.dispatch_id group by k.name, e.counter_name"))
    except sqlite3.Error:
        try:
            rows = list(c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                  "group by kernel_name, counter_name"))
        except sqlite3.Error:
            rows = []
    if rows:
        print("# PMC counters (average per dispatch)")
        for name, ctr, v, n in rows:
            if "lz4amd" in name:
                print("  %-45s %-24s %18.1f  (n=%d)" % (name[:45], ctr, v, n))


if __name__ == "__main__":
    main(sys.argv[1])
