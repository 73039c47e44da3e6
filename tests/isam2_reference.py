"""An INDEPENDENT ISAM2 restatement (test infrastructure) for Pose3 graphs with BetweenFactor<Pose3> / PriorFactor<Pose3>:
what CGraphGT::optimizeGraphIncremental reaches through isam2->update + calculateEstimate (gtsam/gtsam_graph.cpp:1768-1776,
ISAM2Params :93-99: relinearizeThreshold, relinearizeSkip = 1; ISAM2's default wildfireThreshold = 1e-3).

Unlike oracle/orc_gtsam.c::orc_isam2_step (which restates the product's own shortcut: re-linearise everything at theta, solve
the whole system) this one works the way ISAM2 does:
  * every factor keeps its OWN cached linearisation (whitened Jacobians and residual at the linearisation point it was last
    linearised at); an update re-linearises only the factors that touch a variable whose theta moved, plus new factors,
  * variables are marked for relinearisation one by one (max |delta_v| >= relinearizeThreshold),
  * the linear system is re-assembled from the cache, eliminated in variable order (square-root information matrix R), and
    the back-substitution follows ISAM2's wildfire rule: a variable is re-solved if it was re-eliminated or if a variable
    it is conditioned on changed by at least wildfireThreshold; below the threshold the change does not propagate.
Factor arithmetic comes from the oracle's factor functions (tests/orc_binding.py: orc.between / orc.prior / orc.retract)."""
import numpy as np

from tests import orc_binding as orc


def _sqrt_info(info21):
    W = np.zeros((6, 6)); W[np.triu_indices(6)] = info21; W = W + W.T - np.diag(np.diag(W))
    return np.linalg.cholesky(W).T                      # R with R^T R = W


class Isam2Reference:
    def __init__(self, relin_threshold=0.1, wildfire=1e-3):
        self.thr, self.wild = relin_threshold, wildfire
        self.theta = []                                  # linearisation point per variable (7-vectors)
        self.delta = []                                  # linear solution per variable (6-vectors)
        self.factors = []                                # dicts: kind, vars, z, R, cache (A blocks, b) or None
        self.n_relinearised_factors = 0

    def add_pose(self, x):
        self.theta.append(np.array(x, float)); self.delta.append(np.zeros(6))

    def add_between(self, i, j, z, info21):
        self.factors.append(dict(kind="between", vars=(i, j), z=np.array(z, float), R=_sqrt_info(info21), cache=None))

    def add_prior(self, i, mean, info21):
        self.factors.append(dict(kind="prior", vars=(i,), z=np.array(mean, float), R=_sqrt_info(info21), cache=None))

    def _linearise(self, f):
        if f["kind"] == "between":
            i, j = f["vars"]
            e, Ji, Jj = orc.between(self.theta[i], self.theta[j], f["z"])
            f["cache"] = ([f["R"] @ Ji, f["R"] @ Jj], -(f["R"] @ e))
        else:
            (i,) = f["vars"]
            e, J = orc.prior(self.theta[i], f["z"])
            f["cache"] = ([f["R"] @ J], -(f["R"] @ e))
        self.n_relinearised_factors += 1

    def update(self):
        """one ISAM2::update() + calculateEstimate(); returns (estimate[n, 7], variables relinearised)"""
        n = len(self.theta)
        moved = [v for v in range(n) if np.abs(self.delta[v]).max() >= self.thr]
        for v in moved:                                  # theta_v <- theta_v (+) delta_v, delta_v <- 0
            self.theta[v] = orc.retract(self.theta[v], self.delta[v]); self.delta[v] = np.zeros(6)
        mv = set(moved)
        touched = set(moved)                             # variables whose conditionals change
        for f in self.factors:
            if f["cache"] is None or mv.intersection(f["vars"]):
                if f["cache"] is None: touched.update(f["vars"])
                self._linearise(f); touched.update(f["vars"])
        # re-assemble from the CACHED linearisations, eliminate in variable order
        H = np.zeros((6 * n, 6 * n)); g = np.zeros(6 * n)
        for f in self.factors:
            A, b = f["cache"]
            for a, u in enumerate(f["vars"]):
                g[6 * u:6 * u + 6] += A[a].T @ b
                for c, w in enumerate(f["vars"]):
                    H[6 * u:6 * u + 6, 6 * w:6 * w + 6] += A[a].T @ A[c]
        L = np.linalg.cholesky(H)                        # H = L L^T, R = L^T
        y = np.linalg.solve(L, g)
        # re-eliminated variables: the touched ones and everything eliminated after them that they reach (ancestors in the
        # elimination tree = the path to the root of the Bayes tree)
        parent = [-1] * n
        for k in range(n):
            below = np.nonzero(np.abs(L[6 * k + 6:, 6 * k:6 * k + 6]).reshape(-1, 6, 6).max(axis=(1, 2)) > 0)[0] if k + 1 < n else []
            if len(below): parent[k] = k + 1 + int(below[0])
        redone = set()
        for v in touched:
            k = v
            while k >= 0 and k not in redone: redone.add(k); k = parent[k]
        # back-substitution from the root with the wildfire rule
        changed = [False] * n
        for k in range(n - 1, -1, -1):
            rows = [i for i in range(k + 1, n) if np.abs(L[6 * i:6 * i + 6, 6 * k:6 * k + 6]).max() > 0]
            if not (k in redone or any(changed[i] for i in rows)):
                continue                                 # the conditional and everything it is conditioned on are unchanged
            s = y[6 * k:6 * k + 6].copy()
            for i in rows: s -= L[6 * i:6 * i + 6, 6 * k:6 * k + 6].T @ self.delta[i]
            new = np.linalg.solve(L[6 * k:6 * k + 6, 6 * k:6 * k + 6].T, s)
            changed[k] = np.abs(new - self.delta[k]).max() >= self.wild
            self.delta[k] = new
        est = np.array([orc.retract(self.theta[v], self.delta[v]) for v in range(n)])
        return est, len(moved)
