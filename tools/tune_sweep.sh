#!/bin/bash
# A/B or sweep of any tuning constant through the ONE override variable (csrc/fgo_internal.hpp tune()):
#   tools/tune_sweep.sh out_tag "ride_win=18" "ride_win=22" "ride_win=26,ride_ops=400000" ...
# runs bench.py once per setting (cfg 2, 20 timed iterations, median of 3 regions) and prints iterations/s and the phase times;
# BENCH_ARGS passes extra arguments (e.g. "--poses 1000000").  The keys are the lower-case names next to the constants:
#   ordering   nd_leaf nd_bal_t nd_bal_w nd_starts nd_min_side dense_factor hub_deg
#   schedule   leaf_blocks merge_multi chain_work acc_long acc2_min acc_v1 fwd_split dist_min_trees dist_max_share
#   riders     ride ride_t0 ride_tb ride_win ride_ops ride_min ride_max ride_min2 ride_hub ride_hub_force ride_cus ride_xcd ride_win2 ...
#   launches   tri_wide tri1 tri1_min tri_lpt (0 task order, 1 by width: narrowest first for tri1, widest first for tri<8>; 2 / 3 force) leaf_lpt (leaf sub-trees by descending work) acc_narrow acc_mid2 acc_wide2 acc_wide_split acc2_narrow acc2_mid bwd_fused bwd_chain bwd_chain_max bwd_chain_mode ba_small
#   ordering   nd_try (candidates ranked by the cost model; fgo_config.order_candidates)
#   isam2      isam_lookahead isam_masked isam_ranges (0: the plain forms, for A/B runs and the equivalence test)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for cfg in "$@"; do
  FGO_TUNE="$cfg" timeout 600 python bench.py --steps 20 --warmup 5 --cpu-iters 0 --repeats 3 ${BENCH_ARGS:-} > gpurun_out/$TAG/b.json 2> gpurun_out/$TAG/b.err
  python - "$cfg" gpurun_out/$TAG/b.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-40s %8.2f it/s   linearise / sweep / backward ms: %s   levels %d  nnzL %d  ops %d  t_sym %.3f" % (sys.argv[1] or "(defaults)", d["value"], " / ".join("%.3f" % v for v in d["roofline"]["phases_ms"].values()), d["structure"]["levels"], d["structure"]["nnz_L_blocks"], d["structure"]["update_ops"], d["t_symbolic_s"]))
PY
done | tee gpurun_out/$TAG/sweep.txt
