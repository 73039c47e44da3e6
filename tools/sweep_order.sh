# sweep of ordering / task parameters, cfg 2 (env knobs of ordering.cpp / symbolic.cpp)
run() { r=$(env "$@" python bench.py --cpu-iters 0 --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), [round(v,3) for v in d['roofline']['phases_ms'].values()], d['structure']['levels'], d['structure']['nnz_L_blocks'], round(d['t_symbolic_s'],3))"); echo "$* -> $r"; }
run FGO_ND_LEAF=64
run FGO_ND_LEAF=48
run FGO_ND_LEAF=96
run FGO_ND_LEAF=128
run FGO_TASK_WORK=2500
run FGO_TASK_WORK=10000
run FGO_ND_BAL_W=4
run FGO_ND_BAL_W=12
run FGO_ND_BAL_T=0.25
run FGO_ND_BAL_T=0.45
run FGO_LEAF_BLOCKS=160
