// gcrodr.hpp -- "Recycling GMRES": GCRO-DR(m, k) in place of Belos::GCRODRSolMgr
// (ref: solver_lin_belos.h:178-179, keys "Num Blocks" = m and "Num Recycled Blocks" = k, :224-264).
//
// The reference builds a new solver manager for every solve, so no subspace survives from one solve to the next;
// what the option means there is GCRO-DR inside a single solve (Parks, de Sturler, Mackey, Johnson, Maiti, SIAM J.
// Sci. Comput. 28, 2006): one GMRES(m) cycle, then cycles of m - k Arnoldi steps of (I - C C^T) A M^-1 that carry the k
// harmonic Ritz vectors of smallest magnitude as U, C = A M^-1 U (C^T C = I).  Right preconditioning with a fixed
// preconditioner (the recycled space lives in the preconditioned variable t, x = x0 + M^-1 t), the operator is the
// same projected one as in solveProblem.  One iteration = one Arnoldi step; convergence = implicit residual / |r0|.
// oracle/gcrodr.py restates the same algorithm with numpy/LAPACK; the projected dense problems here use
// dense_small.hpp.  Vector work reuses the Krylov kernels (multi-dot, multi-axpy); this solver type is not on the
// bench path and trades a few extra sweeps per cycle for simplicity.
#pragma once
#include "dense_small.hpp"
#include "solver.hpp"

namespace isph {

// c[0..nk) = Basis^T w (all-reduced), to the host
inline int multi_dot_host(isph_ctx *ctx, int n, int nk, const double *Basis, long long ld, const double *w, double *out) {
  int g = stream_grid(n);
  if (g > 1024) g = 1024;
  ISPH_CHECK(ctx->partial.reserve((size_t)(nk + 2) * kMaxRedBlocks > (size_t)kMaxRedBlocks * 66 ? (size_t)(nk + 2) * kMaxRedBlocks : (size_t)kMaxRedBlocks * 66));
  hipLaunchKernelGGL((k_multi_dot<1>), dim3(g), dim3(kBlock), 0, ctx->stream, n, nk, Basis, ld, w, ctx->partial.p);
  hipLaunchKernelGGL(k_reduce_partials, dim3(nk + 1), dim3(kBlock), 0, ctx->stream, nk + 1, g, ctx->partial.p,
                     ctx->dscal.p + SC_DOT);
  ISPH_CHECK(allreduce_inplace(ctx, ctx->dscal.p + SC_DOT, nk + 1));
  ISPH_CHECK(fetch_scalars(ctx, SC_DOT, nk + 1));
  for (int i = 0; i < nk; ++i) out[i] = ctx->hscal[SC_DOT + i];
  return ISPH_SUCCESS;
}

// out (+)= sum_i coef[i] Basis_i, coefficients from the host
inline int combine(isph_ctx *ctx, int n, int nin, const double *Basis, long long ld, const double *coef, double *out,
                   bool accumulate, DevBuf<double> &cbuf) {
  const int sg = stream_grid(n);
  if (!accumulate) hipLaunchKernelGGL(k_fill, dim3(sg), dim3(kBlock), 0, ctx->stream, n, out, 0.0);
  if (nin <= 0) return ISPH_SUCCESS;
  ISPH_CHECK(cbuf.reserve(128));
  ISPH_CHECK_HIP(hipMemcpyAsync(cbuf.p, coef, sizeof(double) * (size_t)nin, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_multi_axpy, dim3(sg), dim3(kBlock), 0, ctx->stream, n, nin, Basis, ld, (const double *)cbuf.p, out);
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));  // the pageable source must outlive the copy
  return ISPH_SUCCESS;
}

inline int gcrodr(const LinOp &op, const double *b, double *x, const isph_solver_params *prm, isph_solve_info *info) {
  isph_ctx *ctx = op.ctx;
  const int n = op.n, m = prm->num_blocks, k = prm->num_recycled;
  ISPH_REQUIRE(m >= 2 && m <= 62, "Num Blocks must be in [2,62]");
  ISPH_REQUIRE(k >= 1 && k < m, "Recycling GMRES needs 0 < Num Recycled Blocks < Num Blocks (Belos::GCRODRSolMgr rejects the "
                                "reference's default list 50/50 as well)");
  const long long ld = ((long long)n + 63) / 64 * 64;
  const int kc = k + 1;  // a complex pair that straddles the k-th slot is kept whole
  DevTmp<double> Vb, Cb, Ub, Cn, Un, tb, rb, wb, zb, cbuf;
  ISPH_CHECK(Vb.reserve((size_t)ld * (size_t)(m + 1)));
  ISPH_CHECK(Cb.reserve((size_t)ld * (size_t)kc));
  ISPH_CHECK(Ub.reserve((size_t)ld * (size_t)kc));
  ISPH_CHECK(Cn.reserve((size_t)ld * (size_t)kc));
  ISPH_CHECK(Un.reserve((size_t)ld * (size_t)kc));
  ISPH_CHECK(tb.reserve((size_t)ld));
  ISPH_CHECK(rb.reserve((size_t)ld));
  ISPH_CHECK(wb.reserve((size_t)ld));
  ISPH_CHECK(zb.reserve((size_t)ld));
  double *V = Vb.p, *C = Cb.p, *U = Ub.p, *t = tb.p, *r = rb.p, *w = wb.p, *z = zb.p;
  const int sg = stream_grid(n);
  hipStream_t st = ctx->stream;
  auto release_all = [&]() {
    Vb.release(); Cb.release(); Ub.release(); Cn.release(); Un.release(); tb.release(); rb.release(); wb.release();
    zb.release(); cbuf.release();
  };
  auto norm_of = [&](const double *v, double *out) -> int {
    ISPH_CHECK(dot_dev(ctx, n, v, v, nullptr, nullptr, SC_MISC + 4));
    ISPH_CHECK(fetch_scalars(ctx, SC_MISC + 4, 1));
    *out = std::sqrt(ctx->hscal[SC_MISC + 4]);
    return ISPH_SUCCESS;
  };

  hipLaunchKernelGGL(k_fill, dim3(sg), dim3(kBlock), 0, st, n, t, 0.0);
  ISPH_CHECK(op.apply(x, r));
  hipLaunchKernelGGL(k_residual, dim3(sg), dim3(kBlock), 0, st, n, b, r);
  double beta = 0.0;
  ISPH_CHECK(norm_of(r, &beta));
  const double scale = beta == 0.0 ? 1.0 : beta;
  info->iters = 0; info->restarts = 0; info->converged = 0;
  info->rel_res_implicit = beta / scale;
  if (beta / scale <= prm->tol) info->converged = 1;
  int kk = 0;  // current size of the recycle space
  int rc = ISPH_SUCCESS;

  while (!info->converged && info->iters < prm->max_iters && rc == ISPH_SUCCESS) {
    const int steps = kk == 0 ? m : m - kk;
    // ---- Arnoldi with (I - C C^T) Op
    std::vector<double> H((size_t)(steps + 1) * steps, 0.0), B((size_t)(kk > 0 ? kk : 1) * steps, 0.0);  // row-major
    std::vector<double> cs((size_t)steps), sn((size_t)steps), g((size_t)steps + 1, 0.0);
    rc = norm_of(r, &beta);
    if (rc != ISPH_SUCCESS) break;
    if (beta == 0.0) { info->converged = 1; break; }
    hipLaunchKernelGGL(k_scale_copy, dim3(sg), dim3(kBlock), 0, st, n, (const double *)r, V, 1.0 / beta, (const double *)nullptr, 0);
    g[0] = beta;
    int j = 0;
    bool conv = false;
    while (j < steps && rc == ISPH_SUCCESS) {
      rc = op.prec(V + (long long)j * ld, z);
      if (rc == ISPH_SUCCESS) rc = op.apply(z, w);
      if (rc != ISPH_SUCCESS) break;
      if (kk > 0) {
        rc = cgs_pass(ctx, n, kk, C, ld, w);
        if (rc != ISPH_SUCCESS) break;
        for (int i = 0; i < kk; ++i) B[(size_t)i * steps + j] = ctx->hscal[SC_DOT + i];
      }
      std::vector<double> hcol((size_t)j + 2, 0.0);
      double ww = 0.0;
      for (int pass = 0; pass < 2 && rc == ISPH_SUCCESS; ++pass) {  // two classical Gram-Schmidt passes
        rc = cgs_pass(ctx, n, j + 1, V, ld, w);
        for (int i = 0; i <= j; ++i) hcol[(size_t)i] += ctx->hscal[SC_DOT + i];
        ww = ctx->hscal[SC_DOT + j + 2];
      }
      if (rc != ISPH_SUCCESS) break;
      const double hn = std::sqrt(ww > 0.0 ? ww : 0.0);
      hcol[(size_t)j + 1] = hn;
      for (int i = 0; i <= j + 1; ++i) H[(size_t)i * steps + j] = hcol[(size_t)i];
      if (hn != 0.0)
        hipLaunchKernelGGL(k_scale_copy, dim3(sg), dim3(kBlock), 0, st, n, (const double *)w, V + (long long)(j + 1) * ld, 1.0 / hn,
                           (const double *)nullptr, 0);
      // implicit residual by Givens rotations on a copy of the column
      for (int i = 0; i < j; ++i) {
        const double a = cs[(size_t)i] * hcol[(size_t)i] + sn[(size_t)i] * hcol[(size_t)i + 1];
        hcol[(size_t)i + 1] = -sn[(size_t)i] * hcol[(size_t)i] + cs[(size_t)i] * hcol[(size_t)i + 1];
        hcol[(size_t)i] = a;
      }
      const double a = hcol[(size_t)j], bb = hcol[(size_t)j + 1], rr = std::hypot(a, bb);
      cs[(size_t)j] = rr == 0.0 ? 1.0 : a / rr;
      sn[(size_t)j] = rr == 0.0 ? 0.0 : bb / rr;
      g[(size_t)j + 1] = -sn[(size_t)j] * g[(size_t)j];
      g[(size_t)j] = cs[(size_t)j] * g[(size_t)j];
      ++j;
      ++info->iters;
      info->rel_res_implicit = std::fabs(g[(size_t)j]) / scale;
      if (info->rel_res_implicit <= prm->tol) { conv = true; break; }
      if (info->iters >= prm->max_iters) break;
    }
    if (rc != ISPH_SUCCESS) break;
    // ---- projected least-squares problem and the updates of t and r
    const int rows = kk + j + 1, cols = kk + j;
    std::vector<double> G((size_t)rows * cols, 0.0), dsc((size_t)(kk > 0 ? kk : 1), 1.0), rhs((size_t)rows, 0.0), y;
    if (kk > 0) {
      for (int i = 0; i < kk && rc == ISPH_SUCCESS; ++i) {
        double un = 0.0;
        rc = norm_of(U + (long long)i * ld, &un);
        dsc[(size_t)i] = un > 0.0 ? 1.0 / un : 1.0;
        G[(size_t)i * cols + i] = dsc[(size_t)i];
      }
      if (rc != ISPH_SUCCESS) break;
      for (int i = 0; i < kk; ++i)
        for (int c = 0; c < j; ++c) G[(size_t)i * cols + kk + c] = B[(size_t)i * steps + c];
      rc = multi_dot_host(ctx, n, kk, C, ld, r, rhs.data());  // C^T r (round-off level)
      if (rc != ISPH_SUCCESS) break;
    }
    for (int i = 0; i <= j; ++i)
      for (int c = 0; c < j; ++c) G[(size_t)(kk + i) * cols + kk + c] = H[(size_t)i * steps + c];
    rhs[(size_t)kk] = beta;
    dense::least_squares(rows, cols, G, rhs, y);
    std::vector<double> coef((size_t)std::max(rows, 1));
    for (int i = 0; i < kk; ++i) coef[(size_t)i] = dsc[(size_t)i] * y[(size_t)i];  // Ut = U D
    if (kk > 0) rc = combine(ctx, n, kk, U, ld, coef.data(), t, true, cbuf);
    if (rc == ISPH_SUCCESS) rc = combine(ctx, n, j, V, ld, y.data() + kk, t, true, cbuf);
    std::vector<double> gy((size_t)rows, 0.0);
    for (int i = 0; i < rows; ++i) {
      double s = 0.0;
      for (int c = 0; c < cols; ++c) s += G[(size_t)i * cols + c] * y[(size_t)c];
      gy[(size_t)i] = -s;
    }
    if (rc == ISPH_SUCCESS && kk > 0) rc = combine(ctx, n, kk, C, ld, gy.data(), r, true, cbuf);
    if (rc == ISPH_SUCCESS) rc = combine(ctx, n, j + 1, V, ld, gy.data() + kk, r, true, cbuf);
    if (rc != ISPH_SUCCESS) break;
    if (conv) { info->converged = 1; break; }
    if (info->iters >= prm->max_iters || info->restarts >= prm->max_restarts) break;
    // ---- harmonic Ritz vectors -> new recycle space
    std::vector<double> P;
    int knew = 0;
    std::vector<dense::cplx> lam, X;
    if (kk == 0) {
      // (H_m + h_{m+1,m}^2 H_m^{-T} e_m e_m^T) z = theta z
      std::vector<double> HmT((size_t)j * j), f((size_t)j, 0.0), Mh((size_t)j * j);
      for (int a2 = 0; a2 < j; ++a2)
        for (int c = 0; c < j; ++c) { HmT[(size_t)a2 * j + c] = H[(size_t)c * steps + a2]; Mh[(size_t)a2 * j + c] = H[(size_t)a2 * steps + c]; }
      f[(size_t)j - 1] = 1.0;
      if (!dense::lu_solve(j, HmT, 1, f)) { rc = fail("GCRO-DR: singular Hessenberg matrix", __FILE__, __LINE__); break; }
      const double h2 = H[(size_t)j * steps + (j - 1)] * H[(size_t)j * steps + (j - 1)];
      for (int a2 = 0; a2 < j; ++a2) Mh[(size_t)a2 * j + (j - 1)] += h2 * f[(size_t)a2];
      if (!dense::eig_general(j, Mh, lam, X)) { rc = fail("GCRO-DR: eigenvalue iteration did not converge", __FILE__, __LINE__); break; }
      knew = dense::select_real_basis(j, lam, X, k, /*largest=*/false, P);
    } else {
      // G^T G z = theta G^T W^T Vh z  <=>  (G^T G)^-1 (G^T W^T Vh) z = (1/theta) z : largest |1/theta|
      std::vector<double> WtV((size_t)rows * cols, 0.0), col((size_t)std::max(kk, j + 1));
      for (int i = 0; i < kk && rc == ISPH_SUCCESS; ++i) {
        rc = multi_dot_host(ctx, n, kk, C, ld, U + (long long)i * ld, col.data());
        for (int a2 = 0; a2 < kk; ++a2) WtV[(size_t)a2 * cols + i] = col[(size_t)a2] * dsc[(size_t)i];
        if (rc == ISPH_SUCCESS) rc = multi_dot_host(ctx, n, j + 1, V, ld, U + (long long)i * ld, col.data());
        for (int a2 = 0; a2 <= j; ++a2) WtV[(size_t)(kk + a2) * cols + i] = col[(size_t)a2] * dsc[(size_t)i];
      }
      if (rc != ISPH_SUCCESS) break;
      for (int c = 0; c < j; ++c) WtV[(size_t)(kk + c) * cols + kk + c] = 1.0;
      std::vector<double> GtG((size_t)cols * cols, 0.0), GtW((size_t)cols * cols, 0.0);
      for (int a2 = 0; a2 < cols; ++a2)
        for (int c = 0; c < cols; ++c) {
          double s1 = 0.0, s2 = 0.0;
          for (int i = 0; i < rows; ++i) { s1 += G[(size_t)i * cols + a2] * G[(size_t)i * cols + c]; s2 += G[(size_t)i * cols + a2] * WtV[(size_t)i * cols + c]; }
          GtG[(size_t)a2 * cols + c] = s1;
          GtW[(size_t)a2 * cols + c] = s2;
        }
      if (!dense::lu_solve(cols, GtG, cols, GtW)) { rc = fail("GCRO-DR: singular projected matrix", __FILE__, __LINE__); break; }
      if (!dense::eig_general(cols, GtW, lam, X)) { rc = fail("GCRO-DR: eigenvalue iteration did not converge", __FILE__, __LINE__); break; }
      knew = dense::select_real_basis(cols, lam, X, k, /*largest=*/true, P);
    }
    // [Q,R] = qr(G P)  (first cycle: G = Hbar), C = W Q, U = Vh P R^-1
    const int pr = kk == 0 ? j : cols;  // rows of P
    std::vector<double> GP((size_t)rows * knew, 0.0), Q, R;
    for (int i = 0; i < rows; ++i)
      for (int c = 0; c < knew; ++c) {
        double s = 0.0;
        for (int a2 = 0; a2 < pr; ++a2) s += G[(size_t)i * cols + a2] * P[(size_t)a2 * knew + c];
        GP[(size_t)i * knew + c] = s;
      }
    dense::qr_thin(rows, knew, GP, Q, R);
    // T = P R^-1 (column by column: back substitution with the upper triangular R)
    std::vector<double> T((size_t)pr * knew, 0.0);
    for (int a2 = 0; a2 < pr; ++a2)
      for (int c = 0; c < knew; ++c) {
        double s = P[(size_t)a2 * knew + c];
        for (int l = 0; l < c; ++l) s -= T[(size_t)a2 * knew + l] * R[(size_t)l * knew + c];
        T[(size_t)a2 * knew + c] = s / R[(size_t)c * knew + c];
      }
    for (int c = 0; c < knew && rc == ISPH_SUCCESS; ++c) {
      double *cn = Cn.p + (long long)c * ld, *un = Un.p + (long long)c * ld;
      for (int i = 0; i < rows; ++i) coef[(size_t)i] = Q[(size_t)i * knew + c];
      if (kk > 0) rc = combine(ctx, n, kk, C, ld, coef.data(), cn, false, cbuf);
      if (rc == ISPH_SUCCESS) rc = combine(ctx, n, j + 1, V, ld, coef.data() + kk, cn, kk > 0, cbuf);
      std::vector<double> cu((size_t)std::max(pr, 1));
      for (int i = 0; i < kk; ++i) cu[(size_t)i] = dsc[(size_t)i] * T[(size_t)i * knew + c];
      for (int i = kk; i < pr; ++i) cu[(size_t)i] = T[(size_t)i * knew + c];
      if (rc == ISPH_SUCCESS && kk > 0) rc = combine(ctx, n, kk, U, ld, cu.data(), un, false, cbuf);
      if (rc == ISPH_SUCCESS) rc = combine(ctx, n, pr - kk, V, ld, cu.data() + kk, un, kk > 0, cbuf);
    }
    if (rc != ISPH_SUCCESS) break;
    std::swap(static_cast<DevBuf<double> &>(Cb), static_cast<DevBuf<double> &>(Cn));  // both stay owned by their scope guards
    std::swap(static_cast<DevBuf<double> &>(Ub), static_cast<DevBuf<double> &>(Un));
    C = Cb.p;
    U = Ub.p;
    kk = knew;
    ++info->restarts;
  }
  if (rc == ISPH_SUCCESS) {  // x = x0 + M^-1 t
    rc = op.prec(t, z);
    if (rc == ISPH_SUCCESS) hipLaunchKernelGGL(k_axpy_dev, dim3(sg), dim3(kBlock), 0, st, n, 1.0, (const double *)nullptr, (const double *)z, x);
    if (rc == ISPH_SUCCESS && hipStreamSynchronize(st) != hipSuccess) rc = fail("GCRO-DR: stream error", __FILE__, __LINE__);
  }
  release_all();
  if (rc == ISPH_SUCCESS) ISPH_CHECK_HIP(hipGetLastError());
  return rc;
}

}  // namespace isph
