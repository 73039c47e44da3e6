#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE in one, WRITE_SIZE in the other), as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: separate --pmc passes with --kernel-trace only; on gfx950
FETCH_SIZE tallies 128-B requests at 64 B (x 2 for coalesced streams); gather patterns carry their own measured factor.

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


# FETCH_SIZE correction per ACCESS PATTERN, measured on known byte counts (tools/pmc_calib.hip, profiles/r04_b_pmc_calibration.txt):
#   16 B per lane, coalesced stream            x 2.00   (the guide's gfx950 correction: 128-B requests tallied at 64 B)
#   48-byte rows of consecutive 288-B blocks   x 1.98
#   48-byte rows of blocks in random order     x 1.49   (the gather-form accumulate, the leaf kernel's loads, BA's W gathers)
#   8-byte elements of random blocks           x 1.35   (k_panel_rows' U gather, k_panel_tri1's column rows are 48-byte rows)
# WRITE_SIZE is exact for streaming and row stores (x 1.00) and 7.6 % high for scattered 8-byte stores (x 0.93).
# Round 3 applied x 2 to every kernel, which put the gather kernels above the achievable HBM rate.
READ_FACTOR = [("k_chol_acc", 1.49), ("k_chol_leaf", 1.49), ("k_chol_fact", 1.49), ("k_panel_rows", 1.35), ("k_panel_tri", 1.49), ("k_solve", 1.49),
               ("k_bwd", 1.49), ("k_fwd", 1.49), ("k_ba_schur", 1.49), ("k_ba_back", 1.49), ("k_ba_cameras", 1.98), ("k_ba_linearize", 1.49),
               ("k_linearize", 1.98), ("k_imu", 1.98)]
WRITE_FACTOR = [("k_panel_rows", 0.93), ("k_panel_tri1", 0.93)]


def factor(table, k, dflt):
    for prefix, f in table:
        if k.startswith(prefix):
            return f
    return dflt


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
sweeps = max(1, nf.get("k_chol_leaf<4>", 0) or nf.get("k_chol_fact<4, 3>", 0))     # one level-0 launch per sweep
print("# HBM traffic from PMC counters, per launch: raw FETCH_SIZE, FETCH_SIZE x 2 (the uniform round-3 figure), FETCH_SIZE x the factor")
print("# calibrated for the kernel's access pattern (tools/pmc_calib.hip); WRITE_SIZE x its factor.  factor sweeps in the run: %d" % sweeps)
print("%-28s %8s %10s %10s %12s %8s %12s %18s" % ("kernel", "launches", "raw rd MB", "x2 rd MB", "calib rd MB", "(factor)", "write MB", "calib MB/sweep r+w"))
tot_cal = tot_x2 = 0.0
factor_kernels = ("k_chol_fact", "k_chol_acc", "k_chol_leaf", "k_panel_tri", "k_panel_rows", "k_solve_fwd", "k_fwd_combine")
fac_cal = fac_x2 = 0.0
for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0))):
    raw = fetch.get(k, 0.0)
    fr, fw = factor(READ_FACTOR, k, 2.0), factor(WRITE_FACTOR, k, 1.0)
    r2, rc = 2.0 * raw, fr * raw
    w = fw * write.get(k, 0.0)
    n = max(nf.get(k, 0), nw.get(k, 0), 1)
    print("%-28s %8d %10.3f %10.3f %12.3f %8.2f %12.3f %18.1f" % (k[:28], n, raw / n / 1e6, r2 / n / 1e6, rc / n / 1e6, fr, w / n / 1e6, (rc + w) / sweeps / 1e6))
    tot_cal += rc + w; tot_x2 += r2 + write.get(k, 0.0)
    if k.startswith(factor_kernels):
        fac_cal += rc + w; fac_x2 += r2 + write.get(k, 0.0)
print("# factor sweep incl. fused forward solve (k_chol_*, k_panel_*, k_solve_fwd): calibrated %.1f MB per sweep (uniform x 2: %.1f MB)" % (fac_cal / sweeps / 1e6, fac_x2 / sweeps / 1e6))
