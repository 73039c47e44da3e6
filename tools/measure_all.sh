#!/bin/bash
# Reproduces the measurements quoted in DESIGN.md on a 1xMI355X box (run from the repo root, e.g. through gpurun):
#   headline bench + rocprofv3 kernel stats + the two PMC passes for the HBM traffic, BASELINE configs 3 / 4 / 5.
# Outputs land in gpurun_out/ (scratch); copy what should be kept into profiles/.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/measure
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python "$ROOT/bench.py" --steps 10 --warmup 2 --repeats 1 --other-configs 0 > "$OUT/bench.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -- python "$ROOT/bench.py" --steps 3 --warmup 1 --cpu-iters 0 --phase-reps 1 --repeats 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -- python "$ROOT/bench.py" --steps 3 --warmup 1 --cpu-iters 0 --phase-reps 1 --repeats 1 > /dev/null 2>&1
cd "$ROOT"
python tools/rocpd_summary.py "$(ls -t "$OUT"/trace/*/*.db | head -1)" > "$OUT/kernel_stats.txt"
python tools/pmc_traffic.py "$(ls -t "$OUT"/pmc_fetch/*/*.db | head -1)" "$(ls -t "$OUT"/pmc_write/*/*.db | head -1)" > "$OUT/pmc_traffic.txt"
# the line the driver will see (five timed regions, CPU legs, other_configs), outside rocprof
timeout 900 python "$ROOT/bench.py" > "$OUT/bench_full.log" 2>&1
grep '^{' "$OUT/bench_full.log" | tail -1 > "$OUT/bench.json"
timeout 600 python tools/run_scenarios.py ba --kf 10000 --pts 500000 --iters 5 2>/dev/null | tail -1 > "$OUT/cfg3_ba.json"
timeout 600 python tools/run_scenarios.py vio --kf 50000 --iters 5 2>/dev/null | tail -1 > "$OUT/cfg4_vio.json"
timeout 900 python bench.py --poses 1000000 --steps 3 --warmup 1 --cpu-iters 0 2>/dev/null | tail -1 > "$OUT/cfg5_1m.json"
head -12 "$OUT/kernel_stats.txt"; tail -1 "$OUT/pmc_traffic.txt"; cut -c1-160 "$OUT/bench.json"
