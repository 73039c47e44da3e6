/* ORACLE — TEST INFRASTRUCTURE ONLY.  OrientedPlane3 / OrientedPlane3Factor as the reference uses them
 * (gtsam/gtsam_graph.cpp:1198-1202 landmark insertion, :1265 factor with Gaussian::Covariance(S 3x3)).
 *
 * PINNED by the GTSAM unit tests the reference carries (gtsam/test/testOrientedPlane3.cpp:61-70 transform,
 * :111-140 retract/local round trip, :143-149 errorVector regression; gtsam/test/testOrientedPlane3Factor.cpp:
 * 37-81, 84-126 two-measurement fusion) — see tests/test_plane_golden.py, which holds those vectors.
 * The fusion tests take ONE linear step (isam2.update) and expect the geodesic midpoint to 1e-9, which fixes the
 * factor to the GTSAM 4.0 form restated here:
 *     predicted = plane.transform(pose):  n' = R^T n,  d' = n . t + d
 *     r = [ -localCoordinates_{n'}(n_z) ; d' - d_z ]          (3)
 *     J = Jacobians of transform() in the local coordinates of the predicted plane (d r / d predicted ~ I)
 * Unit3: basis() = [b1 b2], b1 = normalise(n x axis) with axis the coordinate axis of smallest |n_i|
 * (ties: x before y before z as in GTSAM), b2 = n x b1; retract = exponential map on the sphere.
 * Plane storage: p[4] = nx ny nz d (unit normal).  Tangent: [dn(2); dd].
 */
#ifndef ORC_PLANE_H
#define ORC_PLANE_H
#include "orc_pose3.h"

static inline void orc_unit3_basis(const double n[3], double B[6] /* 3x2 row-major: rows xyz, cols b1 b2 */) {
  const double mx = fabs(n[0]), my = fabs(n[1]), mz = fabs(n[2]);
  double ax[3] = {0, 0, 1};
  if (mx <= my && mx <= mz) { ax[0] = 1; ax[2] = 0; }
  else if (my <= mx && my <= mz) { ax[1] = 1; ax[2] = 0; }
  double b1[3] = {n[1] * ax[2] - n[2] * ax[1], n[2] * ax[0] - n[0] * ax[2], n[0] * ax[1] - n[1] * ax[0]};
  const double nb = sqrt(b1[0] * b1[0] + b1[1] * b1[1] + b1[2] * b1[2]);
  b1[0] /= nb; b1[1] /= nb; b1[2] /= nb;
  const double b2[3] = {n[1] * b1[2] - n[2] * b1[1], n[2] * b1[0] - n[0] * b1[2], n[0] * b1[1] - n[1] * b1[0]};
  for (int k = 0; k < 3; ++k) { B[k * 2] = b1[k]; B[k * 2 + 1] = b2[k]; }
}
/* Unit3::retract: exponential map on S^2 */
static inline void orc_unit3_retract(const double n[3], const double v[2], double out[3]) {
  double B[6], xi[3];
  orc_unit3_basis(n, B);
  for (int k = 0; k < 3; ++k) xi[k] = B[k * 2] * v[0] + B[k * 2 + 1] * v[1];
  const double th = sqrt(xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2]);
  const double s = th < 1e-300 ? 1.0 : sin(th) / th, c = cos(th);
  for (int k = 0; k < 3; ++k) out[k] = c * n[k] + s * xi[k];
  const double nn = sqrt(out[0] * out[0] + out[1] * out[1] + out[2] * out[2]);
  for (int k = 0; k < 3; ++k) out[k] /= nn;
}
/* Unit3::localCoordinates */
static inline void orc_unit3_local(const double n[3], const double y[3], double v[2]) {
  const double x = n[0] * y[0] + n[1] * y[1] + n[2] * y[2];
  if (x > 1.0 - 1e-16) { v[0] = v[1] = 0; return; }
  if (x < -1.0 + 1e-16) { v[0] = M_PI; v[1] = 0; return; }
  const double th = acos(x), k = th / sin(th);
  double B[6];
  orc_unit3_basis(n, B);
  double h[3];
  for (int i = 0; i < 3; ++i) h[i] = k * (y[i] - x * n[i]);
  v[0] = B[0] * h[0] + B[2] * h[1] + B[4] * h[2];
  v[1] = B[1] * h[0] + B[3] * h[1] + B[5] * h[2];
}
/* Unit3::errorVector = B(this)^T q */
static inline void orc_unit3_error_vector(const double n[3], const double q[3], double e[2]) {
  double B[6];
  orc_unit3_basis(n, B);
  e[0] = B[0] * q[0] + B[2] * q[1] + B[4] * q[2];
  e[1] = B[1] * q[0] + B[3] * q[1] + B[5] * q[2];
}
static inline void orc_plane_normalize(const double in[4], double p[4]) {
  const double nn = sqrt(in[0] * in[0] + in[1] * in[1] + in[2] * in[2]);
  p[0] = in[0] / nn; p[1] = in[1] / nn; p[2] = in[2] / nn; p[3] = in[3];   /* OrientedPlane3(a,b,c,d): Unit3(a,b,c), d */
}
/* OrientedPlane3::errorVector(other) = [n.errorVector(other.n); d - other.d]  (testOrientedPlane3.cpp:143-149) */
static inline void orc_plane_error_vector(const double p[4], const double o[4], double e[3]) {
  orc_unit3_error_vector(p, o, e);
  e[2] = p[3] - o[3];
}
static inline void orc_plane_retract(const double p[4], const double v[3], double out[4]) {
  orc_unit3_retract(p, v, out);
  out[3] = p[3] + v[2];
}
static inline void orc_plane_local(const double p[4], const double q[4], double v[3]) {
  orc_unit3_local(p, q, v);
  v[2] = q[3] - p[3];
}
/* OrientedPlane3::transform(pose) with Jacobians in local coordinates of the result.
 * Hpose: 3x6 row-major ([omega; v] right perturbation), Hplane: 3x3 row-major.  Either may be NULL. */
static inline void orc_plane_transform(const double p[4], const double x[7], double out[4], double *Hpose, double *Hplane) {
  double R[9], qc[4];
  orc_qmat(x + 3, R);
  orc_qconj(x + 3, qc);
  orc_qrot(qc, p, out);                                     /* n' = R^T n */
  out[3] = p[0] * x[0] + p[1] * x[1] + p[2] * x[2] + p[3];
  if (!Hpose && !Hplane) return;
  double Bp[6], B[6];
  orc_unit3_basis(out, Bp);
  orc_unit3_basis(p, B);
  if (Hpose) {
    memset(Hpose, 0, 18 * sizeof(double));
    double S[9];
    orc_skew(out, S);                                        /* [n']x */
    for (int a = 0; a < 2; ++a)
      for (int c = 0; c < 3; ++c) Hpose[a * 6 + c] = Bp[0 * 2 + a] * S[0 * 3 + c] + Bp[1 * 2 + a] * S[1 * 3 + c] + Bp[2 * 2 + a] * S[2 * 3 + c];
    for (int c = 0; c < 3; ++c) Hpose[2 * 6 + 3 + c] = out[c];
  }
  if (Hplane) {
    memset(Hplane, 0, 9 * sizeof(double));
    /* B'^T R^T B */
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b) {
        double s = 0;
        for (int i = 0; i < 3; ++i)          /* (R^T B)_{i b} = sum_k R[k][i] B[k][b] */
          s += Bp[i * 2 + a] * (R[0 * 3 + i] * B[0 * 2 + b] + R[1 * 3 + i] * B[1 * 2 + b] + R[2 * 3 + i] * B[2 * 2 + b]);
        Hplane[a * 3 + b] = s;
      }
    for (int b = 0; b < 2; ++b) Hplane[2 * 3 + b] = B[0 * 2 + b] * x[0] + B[1 * 2 + b] * x[1] + B[2 * 2 + b] * x[2];
    Hplane[8] = 1;
  }
}
/* OrientedPlane3Factor: r = [-local_{n'}(n_z); d' - d_z], Jacobians = those of transform() */
static inline void orc_plane_factor(const double x[7], const double plane[4], const double z[4], double r[3], double *Hpose,
                                    double *Hplane) {
  double pred[4], l[2];
  orc_plane_transform(plane, x, pred, Hpose, Hplane);
  orc_unit3_local(pred, z, l);
  r[0] = -l[0]; r[1] = -l[1]; r[2] = pred[3] - z[3];
}
#endif
