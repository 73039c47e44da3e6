#!/usr/bin/env python3
"""counter / known bytes per access pattern: tools/pmc_calib.py known.txt fetch.db write.db (see tools/pmc_calib.hip)"""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    acc = {}
    for name, value in cur.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        short = name.split("(")[0].replace("void ", "").strip()
        acc[short] = acc.get(short, 0.0) + value * 1024.0          # FETCH_SIZE / WRITE_SIZE are reported in KiB
    return acc


known = {}
for line in open(sys.argv[1]):
    f = line.split()
    if len(f) >= 5 and f[0] in ("read", "write"):
        known[f[1]] = (f[0], int(f[2]), int(f[4]))
fetch, write = per_kernel(sys.argv[2], "FETCH_SIZE"), per_kernel(sys.argv[3], "WRITE_SIZE")
print("# rocprofv3 HBM counters vs known byte counts (1.2 GB working set, every byte touched once)")
print("%-24s %6s %14s %14s %10s %12s %12s" % ("kernel", "kind", "known MB", "idx MB", "counter MB", "raw ratio", "factor"))
out = {}
for k, (kind, nbytes, idx) in known.items():
    c = fetch.get(k, 0.0) if kind == "read" else write.get(k, 0.0)
    if kind == "write":
        c_other = fetch.get(k, 0.0)
    ratio = c / nbytes if nbytes else 0.0
    # the index array is a coalesced 4-byte stream: counted like the streaming read (x 1/2 raw), so it is subtracted at that rate
    c_data = c - (0.5 * idx if kind == "read" else 0.0)
    factor = nbytes / c_data if c_data > 0 else float("nan")
    out[k] = {"kind": kind, "known_bytes": nbytes, "counter_bytes": c, "raw_ratio": ratio, "correction_factor": factor}
    print("%-24s %6s %14.1f %14.1f %10.1f %12.3f %12.3f" % (k, kind, nbytes / 1e6, idx / 1e6, c / 1e6, ratio, factor))
print("# factor = known bytes / counter bytes (index stream subtracted): multiply a kernel's raw counter by the factor of its access pattern")
print(json.dumps(out))
