#!/usr/bin/env python3
"""Developer tool: write the edge list of a scenario graph for tools/symstats (FGO_EDGES=<file>): int64 [n, E, ei[E], ej[E]].
   usage: dump_edges.py hubs|torus <size> <out>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from graph_slam_amd import scenarios as S

which, size, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
g = S.hub_graph(n=size) if which == "hubs" else S.torus_graph(nu=size, nv=size)
ei, ej = g["ei"].astype(np.int64), g["ej"].astype(np.int64)
with open(out, "wb") as f:
    np.array([len(g["poses"]), len(ei)], np.int64).tofile(f); ei.tofile(f); ej.tofile(f)
print(len(g["poses"]), len(ei))
