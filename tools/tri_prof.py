"""Shader-clock stamps of the pivot wave of k_panel_tri (FGO_TRI_PROF=1) on the LAST single-panel, full-width level of a cfg-2 sweep."""
import os, sys, ctypes as C
os.environ["FGO_TRI_PROF"] = "1"; os.environ["FGO_GRAPH"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graph_slam_amd as G
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
g = G.synth_manhattan3d(n, 5, 4, 42)
fixed = np.zeros(n, np.uint8); fixed[0] = 1
gr = G.Graph(); gr.add_poses(g["poses"], fixed); gr.add_edges(g["ei"], g["ej"], g["meas"], g["info"])
gr.bench_phase(1, 2)
out = np.zeros(256)
G.lib.fgo_debug_read_scratch.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int64]
assert G.lib.fgo_debug_read_scratch(gr._h, out.ctypes.data_as(C.POINTER(C.c_double)), 256) == 0
st = out.view(np.int64)
M = 16      # columns of the stamped panel
print("kernel: begin->prologue done %d, ->pivot loop done %d, ->L stored+tiles+inverse %d cycles (shader clock)" % (st[1] - st[0], st[2] - st[1], st[3] - st[2]))
ph = st[8:8 + 5 * M].reshape(M, 5)
print("col  update  chol6   trsm  barrier   total")
for k in range(M):
    p = ph[k]
    nxt = ph[k + 1][0] if k < M - 1 else st[2]
    print("%3d %7d %6d %6d %8d %7d" % (k, p[1] - p[0], p[2] - p[1], p[3] - p[2], p[4] - p[3], nxt - p[0]))
print("mean per column: update %.0f chol6 %.0f trsm %.0f barrier %.0f" % tuple(np.mean(ph[:, i + 1] - ph[:, i]) for i in range(4)))
