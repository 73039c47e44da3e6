mkdir -p gpurun_out/ride1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_panels.py tests/test_gpu_isam2.py tests/test_golden_fixtures.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/ride1/tests.txt
for r in 0 1; do FGO_RIDE=$r timeout 300 python bench.py --cpu-iters 0 --repeats 3 2>/dev/null | tail -1 > gpurun_out/ride1/bench_ride$r.json; done
FGO_RIDE=1 bash tools/level_trace.sh ride1/lv1
FGO_RIDE=0 bash tools/level_trace.sh ride1/lv0
cat gpurun_out/ride1/tests.txt; cut -c1-200 gpurun_out/ride1/bench_ride0.json; cut -c1-200 gpurun_out/ride1/bench_ride1.json; tail -3 gpurun_out/ride1/lv1/levels.txt
