#!/bin/bash
# A/B sweep of the rider capacity model (symbolic.cpp "riders") on cfg 2: prints it/s per setting
mkdir -p gpurun_out/ride_sweep
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --cpu-iters 0 --repeats 3 --phase-reps 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.1f it/s' % d['value'], ' '.join('%.3f' % v for v in d['roofline']['phases_ms'].values()))" | tee -a gpurun_out/ride_sweep/out.txt; }
run base FGO_RIDE=1
run off FGO_RIDE=0
for o in 200000 450000 600000; do run ops$o FGO_RIDE_OPS=$o FGO_RIDE_WIN=40; done
for m in 60 200; do run min2_$m FGO_RIDE_MIN2=$m; done
for m in 20 80; do run min_$m FGO_RIDE_MIN=$m; done
for w in 16 30; do run win$w FGO_RIDE_WIN=$w; done
run t0_2 FGO_RIDE_T0=2.0
run t0_4 FGO_RIDE_T0=4.0
