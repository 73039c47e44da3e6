"""Per-level statistics of the accumulate lists on the host (FGO_TUNE=acc_stats=1 makes build_symbolic print them; no GPU needed:
fgo_debug_partition runs ordering + symbolic phase only).  usage: acc_stats.py [poses]"""
import os, sys
os.environ["FGO_TUNE"] = "acc_stats=1"
sys.path.insert(0, ".")
import graph_slam_amd as G
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
g = G.synth_manhattan3d(n, 5, 4, seed=42 if n != 1000000 else 45)
G.debug_partition(n, g["ei"], g["ej"], 1)
