#!/bin/bash
# A/B sweeps on cfg 2: prints it/s and the phase times (linearise / factor sweep / backward) per setting
mkdir -p gpurun_out/ride_sweep
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --cpu-iters 0 --repeats 3 --phase-reps 3 $BENCH_ARGS 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.1f it/s' % d['value'], ' '.join('%.3f' % v for v in d['roofline']['phases_ms'].values()))" | tee -a gpurun_out/ride_sweep/out.txt; }
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_panels.py -x -q -m gpu 2>&1 | tail -2
run base FGO_RIDE=1
run w4off FGO_TRI_WIDE4=1000000
run w4_256 FGO_TRI_WIDE4=256
run w4_800 FGO_TRI_WIDE4=800
run w4_1200 FGO_TRI_WIDE4=1200
BENCH_ARGS="--poses 1000000 --steps 3 --warmup 1 --repeats 1"
run cfg5_base FGO_RIDE=1
run cfg5_w4off FGO_TRI_WIDE4=100000000
run cfg5_rideoff FGO_RIDE=0
