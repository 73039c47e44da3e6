// Stand-in for boost::math::chi_squared + quantile as used by the reference's gtsam/chi2.h:17-26 (chi-square gate on VRO
// edges): quantile by bisection on the regularised lower incomplete gamma function P(k/2, x/2).
#pragma once
#include <cmath>
namespace boost { namespace math {
class chi_squared {
 public:
  explicit chi_squared(double dof) : k_(dof) {}
  double degrees_of_freedom() const { return k_; }
 private:
  double k_;
};
namespace fgo_detail {
inline double gamma_p(double a, double x) {          // series for x < a + 1, continued fraction otherwise
  if (x <= 0) return 0;
  const double gln = std::lgamma(a);
  if (x < a + 1) {
    double ap = a, sum = 1 / a, del = sum;
    for (int n = 0; n < 500; ++n) { ap += 1; del *= x / ap; sum += del; if (std::fabs(del) < std::fabs(sum) * 1e-15) break; }
    return sum * std::exp(-x + a * std::log(x) - gln);
  }
  double b = x + 1 - a, c = 1e300, d = 1 / b, h = d;
  for (int i = 1; i < 500; ++i) {
    const double an = -i * (i - a);
    b += 2; d = an * d + b; if (std::fabs(d) < 1e-300) d = 1e-300; c = b + an / c; if (std::fabs(c) < 1e-300) c = 1e-300;
    d = 1 / d; const double del = d * c; h *= del; if (std::fabs(del - 1) < 1e-15) break;
  }
  return 1 - std::exp(-x + a * std::log(x) - gln) * h;
}
}  // namespace fgo_detail
inline double cdf(const chi_squared &d, double x) { return fgo_detail::gamma_p(0.5 * d.degrees_of_freedom(), 0.5 * x); }
inline double quantile(const chi_squared &d, double p) {
  double lo = 0, hi = 10 + 10 * d.degrees_of_freedom();
  while (cdf(d, hi) < p) hi *= 2;
  for (int i = 0; i < 200; ++i) { const double mid = 0.5 * (lo + hi); if (cdf(d, mid) < p) lo = mid; else hi = mid; }
  return 0.5 * (lo + hi);
}
}}  // namespace boost::math
