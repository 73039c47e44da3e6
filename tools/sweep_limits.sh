for tw in 20000; do for cw in 20000 60000 200000 1000000; do
FGO_TASK_WORK=$tw FGO_CHAIN_WORK=$cw timeout 120 python bench.py --steps 5 --warmup 1 --cpu-iters 0 --phase-reps 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tw=$tw cw=$cw', 'it/s %.1f'%d['value'], 'levels', d['structure']['levels'], {k:round(v,2) for k,v in d['roofline']['phases_ms'].items()})"
done; done
for tw in 5000 60000; do cw=200000
FGO_TASK_WORK=$tw FGO_CHAIN_WORK=$cw timeout 120 python bench.py --steps 5 --warmup 1 --cpu-iters 0 --phase-reps 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tw=$tw cw=$cw', 'it/s %.1f'%d['value'], 'levels', d['structure']['levels'], {k:round(v,2) for k,v in d['roofline']['phases_ms'].items()})"
done
