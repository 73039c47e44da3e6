// Stand-in for the un-vendored `plane` package: CPlane with the members CGraphGT reads
// (gtsam/gtsam_graph.cpp:750-763, 904-925, 1118-1298), a PCL-like cloud, and the covariance hygiene helpers the wrapper
// calls (MatrixCheck / DominateCheck / TriangleMatrix, :1167, :1244-1248).  Plane segmentation itself is front-end work
// and is not reproduced; extractPlanes of the stand-in reports no planes unless the synthetic world provides them.
#pragma once
#include <cmath>
#include <memory>
#include <vector>
#include <Eigen/Core>
#include "opencv2/opencv.hpp"
#include "cam_model.h"

typedef enum { RED = 0, GREEN, BLUE, PURPLE, WHITE, YELLOW, DARK } COLOR;

struct Point { float x = 0, y = 0, z = 0; };
struct Cloud { std::vector<Point> points; };
typedef std::shared_ptr<Cloud> CloudPtr;

class CPlane {
 public:
  CPlane() : nx_(0), ny_(0), nz_(1), d1_(0), m_E_Sdi(1e-4) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) m_CP[r][c] = r == c ? 1e-4 : 0.0; }
  double nx_, ny_, nz_, d1_;          // unit normal and distance: n . p + d = 0
  double m_CP[4][4];                  // covariance of (nx, ny, nz, d)
  double m_E_Sdi;                     // estimated variance of d
  double dis2plane(double px, double py, double pz) const { return nx_ * px + ny_ * py + nz_ * pz + d1_; }
  template <class M> void getNVCov(M &S) const { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) S(r, c) = m_CP[r][c]; }
  double getTraceSVN() const { return m_CP[0][0] + m_CP[1][1] + m_CP[2][2]; }
  void regularizeCOV() { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) m_CP[r][c] = r == c ? (m_CP[r][r] > 1e-8 ? m_CP[r][r] : 1e-8) : 0.0; }
  // the reference re-estimates a propagated plane from its points (gtsam/gtsam_graph.cpp:1031): total least squares --
  // centroid + the eigenvector of the smallest eigenvalue of the scatter matrix (cyclic Jacobi on the 3x3), oriented like
  // the predicted normal; the covariance keeps the stand-in's fixed values
  void computeCOVSparse(const cv::Mat &, const CloudPtr &pts, const std::vector<int> &, int step) {
    if (!pts || pts->points.size() < 50) return;
    if (step < 1) step = 1;
    double c[3] = {0, 0, 0}; size_t cnt = 0;
    for (size_t i = 0; i < pts->points.size(); i += step) { const Point &p = pts->points[i]; c[0] += p.x; c[1] += p.y; c[2] += p.z; ++cnt; }
    for (int k = 0; k < 3; ++k) c[k] /= (double)cnt;
    double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (size_t i = 0; i < pts->points.size(); i += step) {
      const Point &p = pts->points[i];
      const double q[3] = {p.x - c[0], p.y - c[1], p.z - c[2]};
      for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) A[r][s] += q[r] * q[s];
    }
    for (int sweep = 0; sweep < 30; ++sweep)
      for (int p = 0; p < 2; ++p)
        for (int q = p + 1; q < 3; ++q) {
          if (std::fabs(A[p][q]) < 1e-300) continue;
          const double th = 0.5 * std::atan2(2 * A[p][q], A[q][q] - A[p][p]), cs = std::cos(th), sn = std::sin(th);
          for (int k = 0; k < 3; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = cs * akp - sn * akq; A[k][q] = sn * akp + cs * akq; }
          for (int k = 0; k < 3; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = cs * apk - sn * aqk; A[q][k] = sn * apk + cs * aqk; }
          for (int k = 0; k < 3; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = cs * vkp - sn * vkq; V[k][q] = sn * vkp + cs * vkq; }
        }
    int m = 0;
    for (int k = 1; k < 3; ++k) if (A[k][k] < A[m][m]) m = k;
    double n[3] = {V[0][m], V[1][m], V[2][m]};
    if (n[0] * nx_ + n[1] * ny_ + n[2] * nz_ < 0) for (int k = 0; k < 3; ++k) n[k] = -n[k];
    nx_ = n[0]; ny_ = n[1]; nz_ = n[2];
    d1_ = -(n[0] * c[0] + n[1] * c[1] + n[2] * c[2]);
  }
};

// a covariance is usable if it is finite with a positive diagonal
template <class M> inline bool MatrixCheck(const M &S) {
  for (int r = 0; r < S.rows(); ++r) { for (int c = 0; c < S.cols(); ++c) if (!std::isfinite(S(r, c))) return false; if (!(S(r, r) > 0)) return false; }
  return true;
}
// diagonal dominance: |S_ii| >= sum_{j != i} |S_ij|
template <class M> inline bool DominateCheck(const M &S) {
  for (int r = 0; r < S.rows(); ++r) { double off = 0; for (int c = 0; c < S.cols(); ++c) if (c != r) off += std::fabs(S(r, c)); if (std::fabs(S(r, r)) < off) return false; }
  return true;
}
// make it diagonally dominant by dropping the off-diagonal part
template <class M> inline void TriangleMatrix(M &S) { for (int r = 0; r < S.rows(); ++r) for (int c = 0; c < S.cols(); ++c) if (r != c) S(r, c) = 0; }

inline void markColor(cv::Mat &, const std::vector<int> &, COLOR) {}
