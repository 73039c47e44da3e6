#!/usr/bin/env python3
"""Developer tool (round 6, VERDICT r5 next #5: speculate the reject branch).  Before building it: what does a factor sweep running on a SECOND
stream cost the latency-bound tail of a trial (backward sweep + update + linearise) that it would overlap?  Two contexts on the same
100k-pose graph, one loops factor sweeps from a host thread while the other times its backward sweep / linearisation / factor sweep."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import graph_slam_amd as G

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
g = G.synth_manhattan3d(n, 5, 4, seed=42)
fixed = np.zeros(n, np.uint8); fixed[0] = 1


def make():
    gr = G.Graph()
    gr.add_poses(g["poses"], fixed)
    gr.add_edges(g["ei"], g["ej"], g["meas"], g["info"])
    gr.chi2()
    return gr


os.environ.pop("FGO_DEBUG_CU_MASK", None)
if len(sys.argv) > 3 and sys.argv[3] != "-":
    os.environ["FGO_DEBUG_STREAM_PRIO"] = sys.argv[3]          # priority of the TIMED context's stream (high)
a = make()
os.environ.pop("FGO_DEBUG_STREAM_PRIO", None)
if len(sys.argv) > 2 and sys.argv[2] != "-":
    os.environ["FGO_DEBUG_CU_MASK"] = sys.argv[2]              # compute units the LOADING context's stream may use, e.g. 3/4
b = make()
os.environ.pop("FGO_DEBUG_CU_MASK", None)
print("timed context: stream priority %s; loading context: CU mask %s" % (sys.argv[3] if len(sys.argv) > 3 else "default", sys.argv[2] if len(sys.argv) > 2 else "none"))
for gr in (a, b):
    for p in (0, 1, 2):
        gr.bench_phase(p, 2)
alone = {p: a.bench_phase(p, 20) for p in (0, 1, 2)}
stop = False


def load():
    while not stop:
        b.bench_phase(1, 4)


th = threading.Thread(target=load)
th.start()
time.sleep(0.2)
busy = {p: a.bench_phase(p, 20) for p in (0, 1, 2)}
stop = True
th.join()
print("loading context's own factor sweep (alone, its stream): %.3f ms" % b.bench_phase(1, 10))
names = {0: "linearise", 1: "factor sweep", 2: "backward sweep"}
for p in (0, 1, 2):
    print("%-15s alone %.3f ms, beside another context's factor sweeps %.3f ms (x %.2f)" % (names[p], alone[p], busy[p], busy[p] / alone[p]))
