#!/bin/bash
# A/B sweeps on cfg 2: prints it/s and the phase times (linearise / factor sweep / backward) per setting
mkdir -p gpurun_out/ride_sweep
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --cpu-iters 0 --repeats 3 --phase-reps 3 $BENCH_ARGS 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.1f it/s' % d['value'], ' '.join('%.3f' % v for v in d['roofline']['phases_ms'].values()))" | tee -a gpurun_out/ride_sweep/out.txt; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_panels.py tests/test_golden_fixtures.py tests/test_gpu_gtsam.py -x -q -m gpu 2>&1 | tail -2
run base FGO_RIDE=1
python tools/tri_prof.py 2>&1 | tail -22
bash tools/level_trace.sh ride_sweep/lv
tail -1 gpurun_out/ride_sweep/lv/levels.txt
BENCH_ARGS="--poses 1000000 --steps 3 --warmup 1 --repeats 1"
run cfg5_base FGO_RIDE=1
