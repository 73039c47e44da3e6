# sweep of the gather-accumulate launch selection (env knobs of launch_factor / symbolic.cpp), cfg 2
for cfg in "4000 15000 64" "1500 15000 64" "600 15000 64" "0 15000 64" "4000 15000 128" "4000 15000 256" "1500 15000 128" "600 6000 128" "0 6000 256" "1500 15000 32"; do
  set -- $cfg
  r=$(FGO_ACC_NARROW=$1 FGO_ACC_MID2=$2 FGO_ACC_LONG=$3 python bench.py --cpu-iters 0 --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), [round(v,3) for v in d['roofline']['phases_ms'].values()])")
  echo "narrow=$1 mid2=$2 long=$3 -> $r"
done
