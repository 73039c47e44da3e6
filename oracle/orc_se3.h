/* ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into, imported by, or called from the
 * product path (graph_slam_amd/, include/).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may use anything under oracle/.
 *
 * PARITY UNPINNED for the g2o pose-graph path: the arithmetic restated here lives in g2o, which
 * is NOT vendored under /root/reference (g2o/CMakeLists.txt:15-18 points at an un-pinned
 * install, pre-Oct-2017 raw-pointer API, see g2o/g2o_graph.cpp:72-74).  The reference holds no
 * golden vectors for this path.  What is restated is g2o's published
 * VertexSE3 / EdgeSE3 semantics as pinned by the reference's call sites:
 *   - vertex type + oplus:        g2o/g2o_graph.cpp:88,115-119  (VertexSE3, estimate = v1*T)
 *   - edge type, meas, info:      g2o/g2o_graph.cpp:125-132     (EdgeSE3, setInformation(6x6))
 *   - chi2 read-back (no 1/2):    g2o/g2o_graph.cpp:254-258
 * and validated by central differences / dense numpy solves in tests/.
 *
 * Conventions: pose p[7] = tx ty tz qx qy qz qw (unit quaternion, Eigen coeff order);
 * X = (R(q), t) maps local -> world.  Tangent increment d[6] = [dt(3); dq(3)] (g2o order).
 */
#ifndef ORC_SE3_H
#define ORC_SE3_H
#include <math.h>
#include <string.h>

/* quaternion product r = a (x) b, storage (x,y,z,w) */
static inline void orc_qmul(const double a[4], const double b[4], double r[4]) {
  const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
  const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
  r[0] = aw * bx + bw * ax + (ay * bz - az * by);
  r[1] = aw * by + bw * ay + (az * bx - ax * bz);
  r[2] = aw * bz + bw * az + (ax * by - ay * bx);
  r[3] = aw * bw - (ax * bx + ay * by + az * bz);
}
static inline void orc_qconj(const double a[4], double r[4]) {
  r[0] = -a[0]; r[1] = -a[1]; r[2] = -a[2]; r[3] = a[3];
}
/* rotation matrix (row-major 3x3) of a unit quaternion */
static inline void orc_qmat(const double q[4], double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}
static inline void orc_m3v(const double R[9], const double v[3], double r[3]) {
  r[0] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  r[1] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  r[2] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
}
static inline void orc_qrot(const double q[4], const double v[3], double r[3]) {
  double R[9];
  orc_qmat(q, R);
  orc_m3v(R, v, r);
}
static inline void orc_qnormalize(double q[4]) {
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
/* c = a * b for poses */
static inline void orc_pose_mul(const double a[7], const double b[7], double c[7]) {
  double rt[3], q[4];
  orc_qrot(a + 3, b, rt);
  orc_qmul(a + 3, b + 3, q);
  c[0] = a[0] + rt[0]; c[1] = a[1] + rt[1]; c[2] = a[2] + rt[2];
  c[3] = q[0]; c[4] = q[1]; c[5] = q[2]; c[6] = q[3];
}
static inline void orc_pose_inv(const double a[7], double c[7]) {
  double qc[4], r[3];
  orc_qconj(a + 3, qc);
  orc_qrot(qc, a, r);
  c[0] = -r[0]; c[1] = -r[1]; c[2] = -r[2];
  c[3] = qc[0]; c[4] = qc[1]; c[5] = qc[2]; c[6] = qc[3];
}
/* g2o internal::fromVectorMQT: d = [dt; dq]  ->  (R(sqrt(1-|dq|^2), dq), dt); identity rotation
 * when |dq|^2 > 1.  [UPSTREAM g2o types/slam3d/isometry3d_mappings] */
static inline void orc_from_vector_mqt(const double d[6], double inc[7]) {
  double w = 1.0 - (d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
  inc[0] = d[0]; inc[1] = d[1]; inc[2] = d[2];
  if (w < 0) { inc[3] = inc[4] = inc[5] = 0; inc[6] = 1; }
  else { inc[3] = d[3]; inc[4] = d[4]; inc[5] = d[5]; inc[6] = sqrt(w); }
}
/* VertexSE3::oplusImpl: X <- X * fromVectorMQT(d); the quaternion is re-normalised (g2o keeps a
 * rotation matrix and re-orthogonalises every 1000 calls; equivalent to rounding). */
static inline void orc_pose_oplus(const double x[7], const double d[6], double out[7]) {
  double inc[7];
  orc_from_vector_mqt(d, inc);
  orc_pose_mul(x, inc, out);
  orc_qnormalize(out + 3);
}

/* EdgeSE3::computeError + linearizeOplus (no sensor offsets):
 *   Delta = Z^-1 * Xi^-1 * Xj ; e = toVectorMQT(Delta) = [t(Delta); vec(q(Delta)) with w >= 0]
 * Ji, Jj row-major 6x6 = d e / d (oplus increment of Xi / Xj) at 0.  Either may be NULL. */
static inline void orc_edge_se3(const double xi[7], const double xj[7], const double z[7],
                                double e[6], double Ji[36], double Jj[36]) {
  double a[7], qic[4], qb[4], d[3], tb[3], qe[4], Ra[9], te[3];
  orc_pose_inv(z, a);                       /* A = Z^-1 */
  orc_qconj(xi + 3, qic);
  orc_qmul(qic, xj + 3, qb);                /* B = Xi^-1 Xj */
  d[0] = xj[0] - xi[0]; d[1] = xj[1] - xi[1]; d[2] = xj[2] - xi[2];
  orc_qrot(qic, d, tb);
  orc_qmul(a + 3, qb, qe);                  /* E = A B */
  orc_qmat(a + 3, Ra);
  orc_m3v(Ra, tb, te);
  te[0] += a[0]; te[1] += a[1]; te[2] += a[2];
  const double s = (qe[3] < 0) ? -1.0 : 1.0;
  e[0] = te[0]; e[1] = te[1]; e[2] = te[2];
  e[3] = s * qe[0]; e[4] = s * qe[1]; e[5] = s * qe[2];
  if (Jj) {
    double Re[9];
    const double w = s * qe[3], vx = s * qe[0], vy = s * qe[1], vz = s * qe[2];
    orc_qmat(qe, Re);
    memset(Jj, 0, 36 * sizeof(double));
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) Jj[r * 6 + c] = Re[r * 3 + c];
    /* w I + [v]x */
    Jj[3 * 6 + 3] = w;   Jj[3 * 6 + 4] = -vz; Jj[3 * 6 + 5] = vy;
    Jj[4 * 6 + 3] = vz;  Jj[4 * 6 + 4] = w;   Jj[4 * 6 + 5] = -vx;
    Jj[5 * 6 + 3] = -vy; Jj[5 * 6 + 4] = vx;  Jj[5 * 6 + 5] = w;
  }
  if (Ji) {
    memset(Ji, 0, 36 * sizeof(double));
    /* d te / d dt_i = -Ra ; d te / d dq_i = 2 Ra [tb]x */
    const double S[9] = {0, -tb[2], tb[1], tb[2], 0, -tb[0], -tb[1], tb[0], 0};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        Ji[r * 6 + c] = -Ra[r * 3 + c];
        double acc = 0;
        for (int k = 0; k < 3; ++k) acc += Ra[r * 3 + k] * S[k * 3 + c];
        Ji[r * 6 + 3 + c] = 2 * acc;
      }
    /* d eq / d dq_i = -s * M,  M = (wb I - [vb]x)(wa I + [va]x) - vb va^T
     * (vector part of qa (x) (0,d) (x) qb as a linear map of d) */
    const double wa = a[6], ax = a[3], ay = a[4], az = a[5];
    const double wb = qb[3], bx = qb[0], by = qb[1], bz = qb[2];
    const double P[9] = {wb, bz, -by, -bz, wb, bx, by, -bx, wb};      /* wb I - [vb]x */
    const double Q[9] = {wa, -az, ay, az, wa, -ax, -ay, ax, wa};      /* wa I + [va]x */
    const double vb[3] = {bx, by, bz}, va[3] = {ax, ay, az};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        double acc = 0;
        for (int k = 0; k < 3; ++k) acc += P[r * 3 + k] * Q[k * 3 + c];
        acc -= vb[r] * va[c];
        Ji[(3 + r) * 6 + 3 + c] = -s * acc;
      }
  }
}

/* expand 21 upper-triangular row-major entries to a full symmetric row-major 6x6 */
static inline void orc_info_full(const double ut[21], double W[36]) {
  int k = 0;
  for (int r = 0; r < 6; ++r)
    for (int c = r; c < 6; ++c) { W[r * 6 + c] = ut[k]; W[c * 6 + r] = ut[k]; ++k; }
}
#endif
