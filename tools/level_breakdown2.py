import sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select name, start, end, duration, grid_x, workgroup_x from kernels order by start"))
# find one factor sweep: between two k_chol_leaf launches
idx = [i for i, r in enumerate(rows) if 'k_chol_leaf' in r[0]]
a, b = (idx[5], idx[6]) if len(idx) > 6 else (idx[-2], idx[-1])
seq = rows[a:b]
lvl = 0; out = []
cur_l = {'acc': 0, 'tri': 0, 'rows': 0, 'ntri': 0, 'nacc': 0}
for r in seq:
    n = r[0]; d = r[3] / 1e3; grid = r[4] // r[5]
    if 'k_chol_acc' in n or 'k_acc_tile' in n: cur_l = {'acc': d, 'tri': 0, 'rows': 0, 'ntri': 0, 'nacc': grid}; out.append(cur_l)
    elif 'k_panel_tri' in n: cur_l['tri'] = d; cur_l['ntri'] = grid
    elif 'k_panel_rows' in n: cur_l['rows'] = d; cur_l['nrows'] = grid
for i, l in enumerate(out):
    if i < 30 or i % 8 == 0: print(i + 1, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in l.items()})
print("sum acc %.2f tri %.2f rows %.2f ms" % (sum(l['acc'] for l in out) / 1e3, sum(l['tri'] for l in out) / 1e3, sum(l['rows'] for l in out) / 1e3))
# backward sweep of the same trial: kernels after the sweep's last row launch up to the next linearisation, in launch order (top level first)
bw = []
for r in rows[a:b]:
    n = r[0]
    if 'k_bwd_chain' in n or 'k_bwd_fused' in n or 'k_bwd_ext' in n or 'k_bwd_tri' in n or 'k_solve_bwd' in n:
        bw.append((n.split('(')[0].split('::')[-1], round(r[3] / 1e3, 1), r[4] // r[5]))
print("backward sweep (kernel, us, workgroups), top level first:")
print(" ".join("%s:%s/%d" % (n.replace('k_bwd_', '').replace('k_solve_', 's_'), d, g) for n, d, g in bw))
