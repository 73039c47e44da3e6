"""oracle OpenMP leg: threads x sub-tree cap sweep on the cfg-2 graph (host only)"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
import graph_slam_amd as G
from tests import orc_binding as orc
n = 100000
g = G.synth_manhattan3d(n, 5, 4, seed=42)
fixed = np.zeros(n, np.uint8); fixed[0] = 1
for th in [int(x) for x in sys.argv[1:]] or [1, 16, 64]:
    orc.set_threads(th)
    po = orc.Problem(g["poses"], fixed, g["ei"].astype(np.int32), g["ej"].astype(np.int32), g["meas"], g["info"])
    t = time.time(); rc, st = po.optimize(1)
    print("threads", th, "cap", os.environ.get("ORC_MT_CAP", "8"), "factor %.2f s linearize %.2f s total %.2f s" % (st.t_factor, st.t_linearize, time.time() - t - st.t_symbolic), flush=True)
orc.set_threads(1)
