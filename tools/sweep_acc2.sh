for cfg in "0 400 6000" "2000 400 6000" "8000 400 6000" "20000 400 6000" "0 1500 6000" "0 400 20000" "0 1500 20000" "0 4000 40000"; do
  set -- $cfg
  r=$(FGO_ACC2_MIN=$1 FGO_ACC2_NARROW=$2 FGO_ACC2_MID=$3 python bench.py --cpu-iters 0 --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), [round(v,3) for v in d['roofline']['phases_ms'].values()])")
  echo "min=$1 narrow=$2 mid=$3 -> $r"
done
