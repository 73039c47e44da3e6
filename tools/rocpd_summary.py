#!/usr/bin/env python3
"""Turns a rocprofv3 (rocpd sqlite) result into the text summary kept under profiles/."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print("# rocprofv3 --kernel-trace --stats  (durations in nanoseconds as stored by rocpd -> shown in us)")
print("%-100s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, total, avg, pct in rows:
    print("%-100s %8d %14.1f %12.3f %7.2f" % (name[:100], calls, total, avg, pct))
