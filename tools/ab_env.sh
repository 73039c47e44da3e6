#!/bin/bash
# A/B of one environment knob on the headline bench: tools/ab_env.sh VAR valueA valueB [reps] [extra bench args]
VAR=$1; A=$2; B=$3; REPS=${4:-2}; shift 4 2>/dev/null
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],2), 'it/s', round(d['ms_per_step'],3), 'ms/step')"; }
for i in $(seq $REPS); do
  for v in $A $B; do env $VAR=$v python bench.py --cpu-iters 0 "$@" 2>/dev/null | tail -1 | show "$VAR=$v"; done
done
