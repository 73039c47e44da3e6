#!/bin/bash
# kernel trace of single-pose ISAM2 updates on a settled graph (tools/isam_build_breakdown.py): where an update's device time goes
N=${1:-100000}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/isam_trace; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/t" -- python "$ROOT/tools/isam_build_breakdown.py" $N > "$OUT/log.txt" 2>&1
cd "$ROOT"
python tools/rocpd_summary.py "$(ls -t "$OUT"/t/*/*.db | head -1)" > "$OUT/kernel_stats.txt"
python - "$(ls -t "$OUT"/t/*/*.db | head -1)" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sy = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {sy} s on d.kernel_id = s.id order by d.start").fetchall()
# the last update: from the last k_isam2_relin to the end
last = max(i for i, r in enumerate(rows) if "k_isam2_relin" in r[0])
seq = rows[last:]
t0 = seq[0][1]
print("last update: %d kernels, %.1f us from first start to last end; busy %.1f us" % (len(seq), (seq[-1][2] - t0) / 1e3, sum(r[2] - r[1] for r in seq) / 1e3))
agg = {}
prev_end = t0
gaps = 0
for name, a, b in seq:
    short = name.split("(")[0].replace("void ", "").replace("fgo::", "")
    x = agg.setdefault(short, [0, 0.0]); x[0] += 1; x[1] += (b - a) / 1e3
    gaps += max(0, a - prev_end); prev_end = max(prev_end, b)
print("gaps between kernels: %.1f us" % (gaps / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]: print("%-50s %4d launches %9.1f us  (%.1f us each)" % (k[:50], v[0], v[1], v[1] / v[0]))
PY
