#!/usr/bin/env python3
"""Per-level duration breakdown of one factor/solve sequence from a rocprofv3 rocpd database."""
import sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select name, start, end, duration, grid_x, workgroup_x from kernels order by start"))
def seq(tag):
    return [(r[3], r[4] // r[5]) for r in rows if tag in r[0]]
fact, acc, fwd, bwd = seq('k_chol_fact'), seq('k_chol_acc'), seq('k_solve_fwd'), seq('k_solve_bwd')
# one sequence = from the largest-grid launch to the next one
def one(arr, rev=False):
    a = np.array(arr)
    big = a[:, 1].max()
    idx = np.where(a[:, 1] == big)[0]
    if len(idx) < 3: return a
    return a[idx[1]:idx[2]] if not rev else a[idx[1] + 1:idx[2] + 1]
for name, arr in (('fact', one(fact)), ('acc', one(acc)), ('fwd', one(fwd)), ('bwd', one(bwd, True))):
    d = arr[:, 0] / 1e3
    print("%s: n=%d sum=%.2f ms first10(us)=%s grids=%s  rest: mean %.1f us median %.1f max %.1f" % (
        name, len(d), d.sum() / 1e3, np.round(d[:10], 1).tolist(), arr[:10, 1].tolist(), d[10:].mean(), np.median(d[10:]), d[10:].max()))
ks = [r for r in rows if 'k_chol' in r[0] or 'k_solve' in r[0]]
gaps = np.array([(ks[i + 1][1] - ks[i][2]) / 1e3 for i in range(len(ks) - 1)])
print("inter-kernel gap us: mean %.2f median %.2f" % (gaps.mean(), np.median(gaps)))
