// dense_small.hpp -- host-side dense kernels for the (<= 64 x 64) projected problems of the Krylov solvers:
// LU solve, Householder QR, least squares, and the eigen-decomposition of a general real matrix (complex
// Hessenberg reduction + single-shift QR iteration + back substitution).  What Belos gets from LAPACK
// (GEQRF / GESV / GEEV / GGEVX) for GCRO-DR's harmonic Ritz problems.  Row-major storage throughout.
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <vector>

namespace isph {
namespace dense {

typedef std::complex<double> cplx;

// solve A X = B in place (A n x n, B n x nrhs), partial pivoting; returns false on a zero pivot
inline bool lu_solve(int n, std::vector<double> A, int nrhs, std::vector<double> &B) {
  for (int k = 0; k < n; ++k) {
    int p = k;
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(A[(size_t)i * n + k]) > std::fabs(A[(size_t)p * n + k])) p = i;
    if (A[(size_t)p * n + k] == 0.0) return false;
    if (p != k) {
      for (int j = 0; j < n; ++j) std::swap(A[(size_t)k * n + j], A[(size_t)p * n + j]);
      for (int j = 0; j < nrhs; ++j) std::swap(B[(size_t)k * nrhs + j], B[(size_t)p * nrhs + j]);
    }
    const double d = A[(size_t)k * n + k];
    for (int i = k + 1; i < n; ++i) {
      const double f = A[(size_t)i * n + k] / d;
      if (f == 0.0) continue;
      for (int j = k + 1; j < n; ++j) A[(size_t)i * n + j] -= f * A[(size_t)k * n + j];
      for (int j = 0; j < nrhs; ++j) B[(size_t)i * nrhs + j] -= f * B[(size_t)k * nrhs + j];
    }
  }
  for (int i = n - 1; i >= 0; --i)
    for (int j = 0; j < nrhs; ++j) {
      double s = B[(size_t)i * nrhs + j];
      for (int l = i + 1; l < n; ++l) s -= A[(size_t)i * n + l] * B[(size_t)l * nrhs + j];
      B[(size_t)i * nrhs + j] = s / A[(size_t)i * n + i];
    }
  return true;
}

// thin QR of A (r x c, r >= c): Q (r x c, orthonormal columns), R (c x c upper triangular), Householder
inline void qr_thin(int r, int c, const std::vector<double> &A, std::vector<double> &Q, std::vector<double> &R) {
  std::vector<double> W(A), vs((size_t)r * c, 0.0);
  for (int k = 0; k < c; ++k) {
    double nrm = 0.0;
    for (int i = k; i < r; ++i) nrm += W[(size_t)i * c + k] * W[(size_t)i * c + k];
    nrm = std::sqrt(nrm);
    const double alpha = W[(size_t)k * c + k] > 0.0 ? -nrm : nrm;
    double vn = 0.0;
    for (int i = k; i < r; ++i) {
      const double v = W[(size_t)i * c + k] - (i == k ? alpha : 0.0);
      vs[(size_t)i * c + k] = v;
      vn += v * v;
    }
    if (vn > 0.0) {
      for (int j = k; j < c; ++j) {
        double s = 0.0;
        for (int i = k; i < r; ++i) s += vs[(size_t)i * c + k] * W[(size_t)i * c + j];
        s *= 2.0 / vn;
        for (int i = k; i < r; ++i) W[(size_t)i * c + j] -= s * vs[(size_t)i * c + k];
      }
    }
  }
  R.assign((size_t)c * c, 0.0);
  for (int i = 0; i < c; ++i)
    for (int j = i; j < c; ++j) R[(size_t)i * c + j] = W[(size_t)i * c + j];
  Q.assign((size_t)r * c, 0.0);
  for (int j = 0; j < c; ++j) Q[(size_t)j * c + j] = 1.0;
  for (int k = c - 1; k >= 0; --k) {
    double vn = 0.0;
    for (int i = k; i < r; ++i) vn += vs[(size_t)i * c + k] * vs[(size_t)i * c + k];
    if (vn == 0.0) continue;
    for (int j = 0; j < c; ++j) {
      double s = 0.0;
      for (int i = k; i < r; ++i) s += vs[(size_t)i * c + k] * Q[(size_t)i * c + j];
      s *= 2.0 / vn;
      for (int i = k; i < r; ++i) Q[(size_t)i * c + j] -= s * vs[(size_t)i * c + k];
    }
  }
}

// y = argmin ||rhs - A y||, A r x c full column rank
inline void least_squares(int r, int c, const std::vector<double> &A, const std::vector<double> &rhs, std::vector<double> &y) {
  std::vector<double> Q, R;
  qr_thin(r, c, A, Q, R);
  y.assign((size_t)c, 0.0);
  for (int j = 0; j < c; ++j) {
    double s = 0.0;
    for (int i = 0; i < r; ++i) s += Q[(size_t)i * c + j] * rhs[(size_t)i];
    y[(size_t)j] = s;
  }
  for (int i = c - 1; i >= 0; --i) {
    double s = y[(size_t)i];
    for (int l = i + 1; l < c; ++l) s -= R[(size_t)i * c + l] * y[(size_t)l];
    y[(size_t)i] = s / R[(size_t)i * c + i];
  }
}

// eigenvalues lam[n] and eigenvectors X (n x n, column k belongs to lam[k]) of a general real matrix
inline bool eig_general(int n, const std::vector<double> &A, std::vector<cplx> &lam, std::vector<cplx> &X) {
  std::vector<cplx> H((size_t)n * n), Z((size_t)n * n, cplx(0.0));
  for (size_t i = 0; i < (size_t)n * n; ++i) H[i] = A[i];
  for (int i = 0; i < n; ++i) Z[(size_t)i * n + i] = 1.0;
  auto h = [&](int i, int j) -> cplx & { return H[(size_t)i * n + j]; };
  auto z = [&](int i, int j) -> cplx & { return Z[(size_t)i * n + j]; };
  // Hessenberg reduction (Householder)
  std::vector<cplx> v((size_t)n);
  for (int k = 0; k + 2 < n; ++k) {
    double nrm = 0.0;
    for (int i = k + 1; i < n; ++i) nrm += std::norm(h(i, k));
    nrm = std::sqrt(nrm);
    if (nrm == 0.0) continue;
    const cplx x0 = h(k + 1, k);
    const cplx ph = std::abs(x0) > 0.0 ? x0 / std::abs(x0) : cplx(1.0);
    const cplx alpha = -ph * nrm;
    double vn = 0.0;
    for (int i = k + 1; i < n; ++i) {
      v[(size_t)i] = h(i, k) - (i == k + 1 ? alpha : cplx(0.0));
      vn += std::norm(v[(size_t)i]);
    }
    if (vn == 0.0) continue;
    for (int j = 0; j < n; ++j) {  // H <- (I - 2 v v^H / vn) H
      cplx s = 0.0;
      for (int i = k + 1; i < n; ++i) s += std::conj(v[(size_t)i]) * h(i, j);
      s *= 2.0 / vn;
      for (int i = k + 1; i < n; ++i) h(i, j) -= s * v[(size_t)i];
    }
    for (int i = 0; i < n; ++i) {  // H <- H (I - 2 v v^H / vn), Z likewise
      cplx s = 0.0, sz = 0.0;
      for (int j = k + 1; j < n; ++j) { s += h(i, j) * v[(size_t)j]; sz += z(i, j) * v[(size_t)j]; }
      s *= 2.0 / vn;
      sz *= 2.0 / vn;
      for (int j = k + 1; j < n; ++j) { h(i, j) -= s * std::conj(v[(size_t)j]); z(i, j) -= sz * std::conj(v[(size_t)j]); }
    }
  }
  // shifted QR iteration on the Hessenberg matrix -> upper triangular T = Z^H A Z
  const double eps = 2.2e-16;
  int hi = n - 1, iter = 0, total = 0;
  std::vector<cplx> gc((size_t)n), gs((size_t)n);
  while (hi > 0) {
    int l = hi;
    while (l > 0) {
      const double sub = std::abs(h(l, l - 1));
      double ref = std::abs(h(l, l)) + std::abs(h(l - 1, l - 1));
      if (ref == 0.0) ref = 1.0;
      if (sub <= eps * ref) { h(l, l - 1) = 0.0; break; }
      --l;
    }
    if (l == hi) { --hi; iter = 0; continue; }
    if (++total > 100 * n) return false;
    // Wilkinson shift of the trailing 2 x 2 block (an exceptional shift every 10 iterations)
    const cplx a = h(hi - 1, hi - 1), b = h(hi - 1, hi), c = h(hi, hi - 1), d = h(hi, hi);
    const cplx tr = a + d, det = a * d - b * c, disc = std::sqrt(tr * tr * 0.25 - det);
    const cplx m1 = tr * 0.5 + disc, m2 = tr * 0.5 - disc;
    cplx mu = std::abs(m1 - d) < std::abs(m2 - d) ? m1 : m2;
    if (++iter % 10 == 0) mu = d + std::abs(c) * cplx(0.75, 0.3);
    for (int i = l; i <= hi; ++i) h(i, i) -= mu;
    for (int k = l; k < hi; ++k) {  // QR by Givens rotations, rows k, k+1
      const cplx p = h(k, k), q = h(k + 1, k);
      const double r = std::sqrt(std::norm(p) + std::norm(q));
      cplx cc = 1.0, ss = 0.0;
      if (r > 0.0) { cc = p / r; ss = q / r; }
      gc[(size_t)k] = cc;
      gs[(size_t)k] = ss;
      for (int j = k; j < n; ++j) {
        const cplx t1 = h(k, j), t2 = h(k + 1, j);
        h(k, j) = std::conj(cc) * t1 + std::conj(ss) * t2;
        h(k + 1, j) = -ss * t1 + cc * t2;
      }
    }
    for (int k = l; k < hi; ++k) {  // R Q: columns k, k+1 (rows 0..min(k+2, hi)), accumulate in Z
      const cplx cc = gc[(size_t)k], ss = gs[(size_t)k];
      const int top = std::min(k + 2, hi);
      for (int i = 0; i <= top; ++i) {
        const cplx t1 = h(i, k), t2 = h(i, k + 1);
        h(i, k) = t1 * cc + t2 * ss;
        h(i, k + 1) = -t1 * std::conj(ss) + t2 * std::conj(cc);
      }
      for (int i = 0; i < n; ++i) {
        const cplx t1 = z(i, k), t2 = z(i, k + 1);
        z(i, k) = t1 * cc + t2 * ss;
        z(i, k + 1) = -t1 * std::conj(ss) + t2 * std::conj(cc);
      }
    }
    for (int i = l; i <= hi; ++i) h(i, i) += mu;
  }
  lam.resize((size_t)n);
  for (int i = 0; i < n; ++i) lam[(size_t)i] = h(i, i);
  // eigenvectors of the triangular factor by back substitution, X = Z Y
  double tnorm = 0.0;
  for (int i = 0; i < n; ++i)
    for (int j = i; j < n; ++j) tnorm = std::max(tnorm, std::abs(h(i, j)));
  const double tiny = std::max(tnorm, 1e-300) * eps;
  X.assign((size_t)n * n, cplx(0.0));
  std::vector<cplx> y((size_t)n);
  for (int k = 0; k < n; ++k) {
    std::fill(y.begin(), y.end(), cplx(0.0));
    y[(size_t)k] = 1.0;
    for (int i = k - 1; i >= 0; --i) {
      cplx s = 0.0;
      for (int j = i + 1; j <= k; ++j) s += h(i, j) * y[(size_t)j];
      cplx dd = h(i, i) - h(k, k);
      if (std::abs(dd) < tiny) dd = tiny;
      y[(size_t)i] = -s / dd;
    }
    double nrm = 0.0;
    for (int i = 0; i < n; ++i) {
      cplx s = 0.0;
      for (int j = 0; j <= k; ++j) s += z(i, j) * y[(size_t)j];
      X[(size_t)i * n + k] = s;
      nrm += std::norm(s);
    }
    nrm = std::sqrt(nrm);
    if (nrm > 0.0)
      for (int i = 0; i < n; ++i) X[(size_t)i * n + k] /= nrm;
  }
  return true;
}

// real basis (n x kout, row-major, unit columns) of the invariant subspace of the k eigenvalues of smallest modulus
// (largest when `largest`); a complex pair enters as Re x, Im x and is never split: kout = k or k + 1
inline int select_real_basis(int n, const std::vector<cplx> &lam, const std::vector<cplx> &X, int k, bool largest,
                             std::vector<double> &P) {
  std::vector<int> order((size_t)n);
  for (int i = 0; i < n; ++i) order[(size_t)i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    return largest ? std::abs(lam[(size_t)a]) > std::abs(lam[(size_t)b]) : std::abs(lam[(size_t)a]) < std::abs(lam[(size_t)b]);
  });
  std::vector<char> used((size_t)n, 0);
  std::vector<std::vector<double>> cols;
  for (int oi = 0; oi < n && (int)cols.size() < k; ++oi) {
    const int idx = order[(size_t)oi];
    if (used[(size_t)idx]) continue;
    used[(size_t)idx] = 1;
    const cplx l = lam[(size_t)idx];
    std::vector<double> re((size_t)n), im((size_t)n);
    if (std::fabs(l.imag()) <= 1e-12 * std::max(std::abs(l), 1e-300)) {
      int jm = 0;
      for (int i = 1; i < n; ++i)
        if (std::abs(X[(size_t)i * n + idx]) > std::abs(X[(size_t)jm * n + idx])) jm = i;
      const cplx piv = X[(size_t)jm * n + idx];
      for (int i = 0; i < n; ++i) re[(size_t)i] = (X[(size_t)i * n + idx] / piv).real();
      cols.push_back(re);
    } else {
      int partner = -1;
      double best = 0.0;
      for (int c = 0; c < n; ++c) {
        if (used[(size_t)c]) continue;
        const double dist = std::abs(lam[(size_t)c] - std::conj(l));
        if (partner < 0 || dist < best) { partner = c; best = dist; }
      }
      if (partner >= 0) used[(size_t)partner] = 1;
      for (int i = 0; i < n; ++i) { re[(size_t)i] = X[(size_t)i * n + idx].real(); im[(size_t)i] = X[(size_t)i * n + idx].imag(); }
      cols.push_back(re);
      cols.push_back(im);
    }
  }
  const int kout = (int)cols.size();
  P.assign((size_t)n * kout, 0.0);
  for (int c = 0; c < kout; ++c) {
    double nrm = 0.0;
    for (int i = 0; i < n; ++i) nrm += cols[(size_t)c][(size_t)i] * cols[(size_t)c][(size_t)i];
    nrm = std::sqrt(nrm);
    for (int i = 0; i < n; ++i) P[(size_t)i * kout + c] = nrm > 0.0 ? cols[(size_t)c][(size_t)i] / nrm : 0.0;
  }
  return kout;
}

}  // namespace dense
}  // namespace isph
