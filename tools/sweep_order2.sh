run() { for seed in 42 7 99 1234; do r=$(env "$@" python bench.py --cpu-iters 0 --steps 10 --seed $seed 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['structure']['levels'], d['config']['lm_trials_in_timed_region'], round(d['config']['ms_per_trial_device'],3))"); echo "$* seed=$seed -> $r"; done; }
run FGO_ND_BAL_W=8
run FGO_ND_BAL_W=4
run FGO_ND_BAL_W=5
run FGO_ND_BAL_W=6
run FGO_ND_BAL_W=4 FGO_ND_BAL_T=0.45
run FGO_ND_BAL_W=6 FGO_ND_BAL_T=0.45
