#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE in one, WRITE_SIZE in the other), as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: separate --pmc passes with --kernel-trace only; on gfx950
FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled; WRITE_SIZE is used as reported (uncalibrated).

  python tools/pmc_traffic.py <fetch.db> <write.db>

Prints bytes per kernel name, per launch, and per factor sweep (= per level-0 k_chol_leaf / k_chol_fact launch)."""
import sqlite3
import sys
from collections import defaultdict


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    acc, cnt = defaultdict(float), defaultdict(int)
    for name, value in cur.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("fgo::", "")
        acc[short] += value * 1024.0
        cnt[short] += 1
    return acc, cnt


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
sweeps = max(1, nf.get("k_chol_leaf<4>", 0) or nf.get("k_chol_fact<4, 3>", 0))     # one level-0 launch per sweep
print("# HBM traffic from PMC counters; FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE as reported")
print("# factor sweeps in the run: %d" % sweeps)
print("%-28s %8s %16s %16s %18s" % ("kernel", "launches", "read MB/launch", "write MB/launch", "MB per sweep (r+w)"))
tot = 0.0
factor_kernels = ("k_chol_fact", "k_chol_acc", "k_panel_tri", "k_panel_rows", "k_solve_fwd")
tot_factor = 0.0
for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0))):
    r = 2.0 * fetch.get(k, 0.0)
    w = write.get(k, 0.0)
    n = max(nf.get(k, 0), nw.get(k, 0), 1)
    print("%-28s %8d %16.3f %16.3f %18.1f" % (k[:28], n, r / n / 1e6, w / n / 1e6, (r + w) / sweeps / 1e6))
    tot += r + w
    if k.startswith(factor_kernels):
        tot_factor += r + w
print("# factor sweep incl. fused forward solve (k_chol_*, k_panel_*, k_solve_fwd): %.1f MB per sweep" % (tot_factor / sweeps / 1e6))
