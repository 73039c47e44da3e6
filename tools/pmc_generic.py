#!/usr/bin/env python3
"""Per-kernel averages of whatever counters a rocprofv3 --pmc pass collected (rocpd sqlite): python tools/pmc_generic.py <db>"""
import sqlite3, sys
from collections import defaultdict
cur = sqlite3.connect(sys.argv[1]).cursor()
acc, cnt = defaultdict(float), defaultdict(int)
for name, cname, value in cur.execute("select kernel_name, counter_name, value from counters_collection"):
    short = name.split("(")[0].replace("void ", "").replace("fgo::", "")
    acc[(short, cname)] += value; cnt[(short, cname)] += 1
keep = ("k_chol_acc", "k_panel_tri", "k_panel_rows", "k_linearize", "k_chol_leaf", "k_bwd")
for (k, c) in sorted(acc):
    if k.startswith(keep):
        print("%-24s %-32s launches %5d  avg per launch %16.1f" % (k[:24], c, cnt[(k, c)], acc[(k, c)] / cnt[(k, c)]))
