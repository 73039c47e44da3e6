#!/bin/bash
# LM iterations/s and device time per trial on small pose graphs (the sizes the reference's drivers produce)
for n in ${@:-1000 3000 10000 30000}; do
  python bench.py --cpu-iters 0 --poses $n --steps 20 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print($n, 'poses', round(d['value'],1), 'it/s', round(d['config']['ms_per_trial_device'],3), 'ms/trial', d['structure']['levels'], 'levels', {k[:12]: round(v,3) for k,v in d['roofline']['phases_ms'].items()})"
done
