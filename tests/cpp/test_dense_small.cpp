// Host-only check of csrc/dense_small.hpp (the projected dense problems of GCRO-DR): reads matrices from stdin,
// prints eigenvalues / residuals.  Driven by tests/test_host_dense.py against numpy.
#include <cstdio>
#include <vector>

#include "dense_small.hpp"

using namespace isph::dense;

int main() {
  int n = 0;
  if (std::scanf("%d", &n) != 1) return 2;
  std::vector<double> A((size_t)n * n);
  for (auto &v : A) if (std::scanf("%lf", &v) != 1) return 2;
  std::vector<cplx> lam, X;
  if (!eig_general(n, A, lam, X)) { std::printf("FAIL\n"); return 1; }
  // residual ||A x - lam x|| per eigenpair
  double worst = 0.0;
  for (int k = 0; k < n; ++k) {
    double r = 0.0;
    for (int i = 0; i < n; ++i) {
      cplx s = 0.0;
      for (int j = 0; j < n; ++j) s += A[(size_t)i * n + j] * X[(size_t)j * n + k];
      r += std::norm(s - lam[(size_t)k] * X[(size_t)i * n + k]);
    }
    worst = std::max(worst, std::sqrt(r));
  }
  std::printf("%.17g\n", worst);
  for (int k = 0; k < n; ++k) std::printf("%.17g %.17g\n", lam[(size_t)k].real(), lam[(size_t)k].imag());
  // QR + least squares on the first n-1 columns
  const int c = n - 1;
  std::vector<double> T((size_t)n * c), rhs((size_t)n), y, Q, R;
  for (int i = 0; i < n; ++i) { rhs[(size_t)i] = 1.0 + i; for (int j = 0; j < c; ++j) T[(size_t)i * c + j] = A[(size_t)i * n + j]; }
  qr_thin(n, c, T, Q, R);
  double qerr = 0.0;
  for (int a = 0; a < c; ++a)
    for (int b = 0; b < c; ++b) {
      double s = 0.0;
      for (int i = 0; i < n; ++i) s += Q[(size_t)i * c + a] * Q[(size_t)i * c + b];
      qerr = std::max(qerr, std::fabs(s - (a == b ? 1.0 : 0.0)));
    }
  double rerr = 0.0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < c; ++j) {
      double s = 0.0;
      for (int l = 0; l < c; ++l) s += Q[(size_t)i * c + l] * R[(size_t)l * c + j];
      rerr = std::max(rerr, std::fabs(s - T[(size_t)i * c + j]));
    }
  std::printf("%.17g %.17g\n", qerr, rerr);
  least_squares(n, c, T, rhs, y);
  for (int j = 0; j < c; ++j) std::printf("%.17g\n", y[(size_t)j]);
  return 0;
}
