for cfg in "0.25 4" "0.5 4" "1.0 2" "1.0 1" "2.0 1"; do set -- $cfg; echo "share $1 trees $2"; for n in 100000 1000000; do FGO_DIST_MAX_SHARE=$1 FGO_DIST_MIN_TREES=$2 python tools/dist_rank_timing.py $n 8 0 3 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   ', d['poses'], 'rank', d['rank'], 'dev ms/trial %.2f' % d['device_ms_per_trial'], 'MB/trial %.1f' % (d['collective_bytes_per_trial']/1e6), 'levels', d['levels'])"; done; done
