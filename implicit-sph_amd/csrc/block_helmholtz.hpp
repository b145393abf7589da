// block_helmholtz.hpp -- the dim x dim block Helmholtz system.
//
// Replaces FunctorOuterIncompNavierStokesBlockHelmholtz (ref: functor_incomp_navier_stokes_block_helmholtz.h:57-187)
// together with the block branch of Corrected::FunctorOuterLaplacianMatrix (functor_laplacian_matrix.h:269-314) and
// FunctorOuterBoundaryNavierSlip (functor_boundary_navier_slip.h:53-184), AS WRITTEN in the reference:
//   pass ib = 0   Laplacian(dt, nu), filter (Fluid,Solid): a fluid row whose own wall normal is set (|n_i|^2 > 0.5)
//                 goes to blocks (ib*, jb) with weights n^_jb n^_ib*, n^ = normalised sum of the normals of all row
//                 entries (neighbours in the cut + self), ib* = first component with n^_ib^2 >= 1/dim; other fluid
//                 rows go to block (0,0)
//   pass ib >= 1  Navier slip, -beta dt: robin_i on the diagonal entry of EVERY block, weight delta - n^ n^T,
//                 once per pass (dim-1 times); G_i = Gc[i] for both operator families
//   every ib      Laplacian(dt, nu), filter (Fluid,Fluid) into block (ib,ib)
//   w_ib = (1-theta) sum_jb A(ib,jb) v_ib  (the same component ib of v for every jb, as written),
//   A *= -theta, diag(ib,ib) = 1 + A_ii (fluid) | 1 (solid), b += w + dt (f/rho + g) - dt/rho grad p.
// One lane per row; the two Laplacian passes keep separate accumulators (grad m, c_i, diagonal) and share the
// neighbour sweeps; every block is written on the scalar pattern (same slice offsets / columns).
#pragma once
#include "assemble.hpp"

namespace isph {

inline bool sell_cols16(isph_ctx *ctx, const Sell &S);  // solver.hpp: the SpMV's 16-bit window columns


struct BlockHelmholtzArgs {
  HelmholtzArgs h;
  double beta;
  const double *normal;  // [nall][3] or NULL
  int *bcol[9];          // per block SELL column arrays (same content)
  double *bval[9];       // per block SELL value arrays; NULL: block not stored (off-diagonal blocks without normals)
};

__global__ __launch_bounds__(kBlock) void k_asm_block_helmholtz(AsmTables T, BlockHelmholtzArgs A,
                                                                const long long *__restrict__ slice_off,
                                                                double *__restrict__ b) {
  const HelmholtzArgs &a = A.h;
  const int i = xcd_block() * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int dim = T.dim, d2 = dim * dim;
  if (i >= a.nlocal) {
    const int nslices = (a.nlocal + kSlice - 1) / kSlice;
    const int slice = i >> 6;
    if (slice < nslices) {
      const long long off = slice_off[slice];
      const int w = (int)((slice_off[slice + 1] - off) >> 6);
      for (int k = 0; k < w; ++k) {
        const long long p = sell_pos(off, lane, k);
        for (int q = 0; q < d2; ++q)
          if (A.bval[q]) { A.bcol[q][p] = 0; A.bval[q][p] = 0.0; }
      }
    }
    return;
  }
  const int nt1 = T.ntypes + 1, dL = dim * (dim + 1) / 2;
  const int it = a.type[i], ikind = T.kind[it];
  const long long off = slice_off[i >> 6];
  const int w = (int)((slice_off[(i >> 6) + 1] - off) >> 6);
  const double alpha = a.dt;
  const double mi = a.nu[i];
  const int jb0 = 0, je = T.nlen[i];
  const int ci_own = a.colmap[i];
  int cnt = 0, pdiag = -1;

  if (!(ikind & KIND_FLUID)) {  // solid row: zero row, unit diagonal in the diagonal blocks, b = v
    for (int jj = jb0; jj < je; ++jj) {
      const int j = neigh_at(T, i, jj - jb0);
      double rij[3];
      if (pair_rsq(dim, a.x, i, j, rij) < T.cutsq[it * nt1 + a.type[j]]) {
        const int cj = a.colmap[j];
        if (T.sorted && pdiag < 0 && cj > ci_own) pdiag = cnt++;
        const long long p = sell_pos(off, lane, cnt++);
        for (int q = 0; q < d2; ++q)
          if (A.bval[q]) { A.bcol[q][p] = cj; A.bval[q][p] = 0.0; }
      }
    }
    if (pdiag < 0) pdiag = cnt++;
    const long long pd = sell_pos(off, lane, pdiag);
    for (int ib = 0; ib < dim; ++ib)
      for (int jb = 0; jb < dim; ++jb) {
        const int q = ib * dim + jb;
        if (A.bval[q]) { A.bcol[q][pd] = ci_own; A.bval[q][pd] = ib == jb ? 1.0 : 0.0; }
      }
    for (int k = cnt; k < w; ++k) {
      const long long p = sell_pos(off, lane, k);
      for (int q = 0; q < d2; ++q)
        if (A.bval[q]) { A.bcol[q][p] = ci_own; A.bval[q][p] = 0.0; }
    }
    for (int k = 0; k < dim; ++k) b[(size_t)k * a.lda + i] = a.v[3 * (size_t)i + k];
    return;
  }

  double G[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, L[6] = {1, 0, 1, 0, 0, 1};
  if (dim == 2) { G[0] = 1; G[1] = 0; G[2] = 0; G[3] = 1; L[0] = 1; L[1] = 0; L[2] = 1; }
  if (!a.antisym) {
    for (int k = 0; k < d2; ++k) G[k] = a.Gc[(size_t)i * d2 + k];
    for (int k = 0; k < dL; ++k) L[k] = a.Lc[(size_t)i * dL + k];
  }
  const double vi = a.vfrac[i];
  // accumulators of the two Laplacian passes: index 0 = (Fluid,Fluid), 1 = (Fluid,Solid)
  double grad_m[2][3] = {{0, 0, 0}, {0, 0, 0}}, cacc[3] = {0, 0, 0};
  double diag1[2] = {0, 0}, gp[3] = {0, 0, 0};
  double nsum[3] = {0, 0, 0}, robin = 0.0;
  const bool have_normal = A.normal != nullptr;
  double ni[3] = {0, 0, 0};
  if (have_normal)
    for (int k = 0; k < dim; ++k) ni[k] = A.normal[3 * (size_t)i + k];
  double GcI[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (have_normal && a.Gc)
    for (int k = 0; k < d2; ++k) GcI[k] = a.Gc[(size_t)i * d2 + k];

  for (int jj = jb0; jj < je; ++jj) {  // sweep 1
    const int j = neigh_at(T, i, jj - jb0);
    const int jt = a.type[j], jkind = T.kind[jt];
    double rij[3];
    const double rsq = pair_rsq(dim, a.x, i, j, rij);
    if (!(rsq < T.cutsq[it * nt1 + jt])) continue;
    const double mj = a.nu[j];
    const bool jsolid = (jkind & KIND_SOLID) != 0;
    // filter coefficients of the two passes (FilterBinary::yes(ikind,ikind) / yes(ikind,jkind))
    double coeff[2];
    coeff[0] = jsolid ? 0.0 : 1.0;   // (Fluid,Fluid): fluid-fluid pairs
    coeff[1] = jsolid ? 1.0 : 0.0;   // (Fluid,Solid): fluid-solid pairs
    if (jsolid && a.morris)
      coeff[1] = mirror_coeff(a.pnd, a.vfrac, a.safe, T.h[it * nt1 + jt], i, j, sqrt(T.cutsq[it * nt1 + jt]));
    const double r = sqrt(rsq) + kEps;
    const double rinv = 1.0 / r;
    const double dwdr = kernel_dval(T.kernel, r, T.hinv[it * nt1 + jt], T.kdnorm[it * nt1 + jt]);
    double e[3] = {0, 0, 0};
    for (int k = 0; k < dim; ++k) e[k] = rij[k] * rinv;
    const double vfrac = a.antisym ? sqrt(vi * a.vfrac[j]) : a.vfrac[j];
    const double vjtmp = dwdr * vfrac;
    if (ikind & jkind)  // same test in both passes (functor_laplacian_matrix.h:164)
      for (int k2 = 0; k2 < dim; ++k2) {
        double gitmp = 0.0;
        for (int k1 = 0; k1 < dim; ++k1) gitmp += G[k2 * dim + k1] * e[k1];
        const double t = gitmp * vjtmp * (a.antisym ? (mi + mj) : (mj - mi));
        grad_m[0][k2] += t;
        grad_m[1][k2] += t;
      }
    double aij = 0.0;
    for (int k2 = 0, op = 0; k2 < dim; ++k2)
      for (int k1 = 0; k1 < k2 + 1; ++k1, ++op) aij += L[op] * e[k1] * e[k2] * (k1 == k2 ? 1.0 : 2.0);
    aij *= 2.0 * dwdr * vfrac;
    if (!a.antisym)
      for (int k = 0; k < dim; ++k) cacc[k] += aij * e[k];  // c_i takes every neighbour in the cut, in both passes
    diag1[0] += aij * mi * coeff[0] * rinv;
    diag1[1] += aij * mi * coeff[1] * rinv;
    if (a.incremental && !jsolid) {  // grad p, filter (Fluid,Fluid)
      const double vd = dwdr * rinv * vfrac;
      for (int k2 = 0; k2 < dim; ++k2) {
        double gitmp = 0.0;
        for (int k1 = 0; k1 < dim; ++k1) gitmp += G[k2 * dim + k1] * rij[k1];
        gp[k2] += gitmp * vd * (a.antisym ? (a.p[i] + a.p[j]) : (a.p[j] - a.p[i]));
      }
    }
    if (have_normal) {
      for (int k = 0; k < dim; ++k) nsum[k] += A.normal[3 * (size_t)j + k];
      if (jkind == KIND_SOLID && dim > 1) {  // Navier slip term
        double tmp = 0.0;
        for (int k2 = 0; k2 < dim; ++k2) {
          double ak = 0.0;
          for (int k1 = 0; k1 < dim; ++k1) ak += GcI[k2 * dim + k1] * rij[k1];
          tmp += (ni[k2] + A.normal[3 * (size_t)j + k2]) * ak;
        }
        robin -= (-A.beta * a.dt) * dwdr / r * a.vfrac[j] / a.rho[i] * tmp;
      }
    }
  }
  // wall-normal distribution of the (Fluid,Solid) row
  double nn = 0.0;
  for (int k = 0; k < dim; ++k) nn += ni[k] * ni[k];
  const bool distribute = have_normal && nn > 0.5;
  double nh[3] = {0, 0, 0};
  int ibp = 0;
  if (distribute) {
    double norm = 0.0;
    for (int k = 0; k < dim; ++k) { nh[k] = nsum[k] + ni[k]; norm += nh[k] * nh[k]; }
    norm = sqrt(norm);
    for (int k = 0; k < dim; ++k) nh[k] /= norm;
    for (; ibp < dim - 1 && (nh[ibp] * nh[ibp] < 1.0 / dim); ++ibp);
  }
  // weight of the (Fluid,Solid) value in block (ib,jb)
  auto wfs = [&](int ib, int jb) -> double {
    if (!distribute) return (ib == 0 && jb == 0) ? 1.0 : 0.0;
    return ib == ibp ? nh[jb] * nh[ibp] : 0.0;
  };
  double wv[3] = {0, 0, 0}, diag2[2] = {0, 0};
  for (int jj = jb0; jj < je; ++jj) {  // sweep 2
    const int j = neigh_at(T, i, jj - jb0);
    const int jt = a.type[j], jkind = T.kind[jt];
    double rij[3];
    const double rsq = pair_rsq(dim, a.x, i, j, rij);
    if (!(rsq < T.cutsq[it * nt1 + jt])) continue;
    const bool jsolid = (jkind & KIND_SOLID) != 0;
    const double r = sqrt(rsq) + kEps;
    const double rinv = 1.0 / r;
    const double dwdr = kernel_dval(T.kernel, r, T.hinv[it * nt1 + jt], T.kdnorm[it * nt1 + jt]);
    double e[3] = {0, 0, 0};
    for (int k = 0; k < dim; ++k) e[k] = rij[k] * rinv;
    const double vfrac = a.antisym ? sqrt(vi * a.vfrac[j]) : a.vfrac[j];
    const double vjtmp = dwdr * vfrac;
    double a0 = 0.0;
    for (int k2 = 0, op = 0; k2 < dim; ++k2)
      for (int k1 = 0; k1 < k2 + 1; ++k1, ++op) a0 += L[op] * e[k1] * e[k2] * (k1 == k2 ? 1.0 : 2.0);
    a0 *= 2.0 * dwdr * vfrac * mi * rinv;
    double bc = 0.0, bg[2] = {0, 0};
    for (int k2 = 0; k2 < dim; ++k2) {
      double bij = 0.0;
      for (int k1 = 0; k1 < dim; ++k1) bij += G[k2 * dim + k1] * e[k1];
      bc += bij * cacc[k2];
      bg[0] += bij * grad_m[0][k2];
      bg[1] += bij * grad_m[1][k2];
    }
    double val[2];
    for (int s = 0; s < 2; ++s) {
      const double coeff = (s == 0) ? (jsolid ? 0.0 : 1.0) : (jsolid ? 1.0 : 0.0);   // plain filter coefficient
      double coeff_a = coeff;
      if (s == 1 && jsolid && a.morris)
        coeff_a = mirror_coeff(a.pnd, a.vfrac, a.safe, T.h[it * nt1 + jt], i, j, sqrt(T.cutsq[it * nt1 + jt]));
      const double tmp = coeff * (mi * bc * vjtmp - bg[s] * vjtmp);
      double v = -(a0 * coeff_a);
      v -= tmp;
      diag2[s] += tmp;
      val[s] = v * alpha;
    }
    const int cj = a.colmap[j];
    if (T.sorted && pdiag < 0 && cj > ci_own) pdiag = cnt++;
    const long long p = sell_pos(off, lane, cnt++);
    for (int ib = 0; ib < dim; ++ib)
      for (int jb = 0; jb < dim; ++jb) {
        const int q = ib * dim + jb;
        const double v = (ib == jb ? val[0] : 0.0) + val[1] * wfs(ib, jb);
        wv[ib] += v * a.v[3 * (size_t)j + ib];
        if (A.bval[q]) { A.bcol[q][p] = cj; A.bval[q][p] = v * (-a.theta); }
      }
  }
  // diagonal entry
  double nhs[3] = {0, 0, 0};
  if (have_normal) {
    const double norm = sqrt(nn);
    if (norm != 0.0)
      for (int k = 0; k < dim; ++k) nhs[k] = ni[k] / norm;
  }
  const double dff = (diag1[0] + diag2[0]) * alpha, dfs = (diag1[1] + diag2[1]) * alpha;
  {
    if (pdiag < 0) pdiag = cnt++;
    const long long p = sell_pos(off, lane, pdiag);
    for (int ib = 0; ib < dim; ++ib)
      for (int jb = 0; jb < dim; ++jb) {
        const int q = ib * dim + jb;
        double v = (ib == jb ? dff : 0.0) + dfs * wfs(ib, jb);
        if (have_normal)
          for (int pass = 1; pass < dim; ++pass) v += robin * ((ib == jb ? 1.0 : 0.0) - nhs[jb] * nhs[ib]);
        wv[ib] += v * a.v[3 * (size_t)i + ib];
        v *= -a.theta;
        if (ib == jb) v += 1.0;
        if (A.bval[q]) { A.bcol[q][p] = ci_own; A.bval[q][p] = v; }
      }
  }
  for (int k = cnt; k < w; ++k) {
    const long long p = sell_pos(off, lane, k);
    for (int q = 0; q < d2; ++q)
      if (A.bval[q]) { A.bcol[q][p] = ci_own; A.bval[q][p] = 0.0; }
  }
  for (int k = 0; k < dim; ++k) {
    double bk = a.v[3 * (size_t)i + k];
    bk += wv[k] * (1.0 - a.theta);
    bk += a.dt * (a.f[3 * (size_t)i + k] / a.rho[i] + a.g[k]);
    if (a.incremental) bk += a.dt * (-1.0 / a.rho[i] * gp[k]);
    b[(size_t)k * a.lda + i] = bk;
  }
}

// host side: blocks_out[ib*dim+jb]; without normals the off-diagonal blocks are identically zero and returned NULL
inline int assemble_block_helmholtz(isph_ctx *ctx, const isph_particles *P, int ncol, int antisym, double dt, double theta,
                                    double beta, const double *nu, const double *rho, const double *pres,
                                    const double *force, const double *gvec, int incremental, const double *vel,
                                    const double *normal, int lda, isph_mat **blocks_out, double *b_out, int on_device) {
  ISPH_REQUIRE(P->dim == 2 || P->dim == 3, "dim must be 2 or 3");
  ISPH_REQUIRE(P->x && P->type && (P->neigh_ptr || P->neigh_ptr64) && P->neigh_idx && P->colmap, "particle arrays missing");
  ISPH_REQUIRE(antisym || (P->Gc && P->Lc), "Symmetric family needs Gc and Lc");
  ISPH_REQUIRE(!normal || P->Gc, "the Navier-slip wall terms need Gc");
  ISPH_REQUIRE(P->vfrac, "vfrac is required (isph_compute_volumes + forward comm first)");
  ISPH_REQUIRE(ncol >= P->nlocal && lda >= P->nlocal, "need ncol >= nlocal and lda >= nlocal");
  const int n = P->nlocal, dim = P->dim, d2 = dim * dim, dL = dim * (dim + 1) / 2;
  StagedParticles S;
  DevTmp<double> snu, sp, sf, sv, sn;
  AsmTables T;
  BlockHelmholtzArgs A;
  memset(&A, 0, sizeof(A));
  HelmholtzArgs &a = A.h;
  isph_mat *blk[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  DevTmp<double> bdev;
  DevTmp<int> newlen;
  int rc = stage_tables(ctx, P, S, T);
  long long nnb = 0;
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->x, (size_t)P->nall * 3, on_device, S.x, &a.x);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->type, (size_t)P->nall, on_device, S.type, &a.type);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->colmap, (size_t)P->nall, on_device, S.colmap, &a.colmap);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->vfrac, (size_t)P->nall, on_device, S.vfrac, &a.vfrac);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->Gc, (size_t)P->nlocal * d2, on_device, S.Gc, &a.Gc);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->Lc, (size_t)P->nlocal * dL, on_device, S.Lc, &a.Lc);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, rho, (size_t)P->nall, on_device, S.rho, &a.rho);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, nu, (size_t)P->nall, on_device, snu, &a.nu);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, pres, (size_t)P->nall, on_device, sp, &a.p);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, force, (size_t)P->nall * 3, on_device, sf, &a.f);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, vel, (size_t)P->nall * 3, on_device, sv, &a.v);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, normal, (size_t)P->nall * 3, on_device, sn, &A.normal);
  if (rc == ISPH_SUCCESS && P->morris_holmes) {
    if (!P->pnd) rc = fail("MorrisHolmes needs pnd", __FILE__, __LINE__);
    else rc = stage(ctx, P->pnd, (size_t)P->nall, on_device, S.pnd, &a.pnd);
  }
  NeighPtr np;
  if (rc == ISPH_SUCCESS) rc = stage_neigh_ptr(ctx, P, n, on_device, S.nptr, S.nptr64, np, &nnb);
  a.nptr = np.p32;
  if (rc == ISPH_SUCCESS) {
    if (!on_device) {
      for (long long k = 0; k < nnb && rc == ISPH_SUCCESS; ++k)
        if (P->neigh_idx[k] < 0 || P->neigh_idx[k] >= P->nall) rc = fail("neighbour index out of range", __FILE__, __LINE__);
      for (int j = 0; j < P->nall && rc == ISPH_SUCCESS; ++j)
        if (P->colmap[j] < 0 || P->colmap[j] >= ncol) rc = fail("colmap entry out of range", __FILE__, __LINE__);
    }
  }
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->neigh_idx, (size_t)nnb, on_device, S.nidx, &a.nidx);
  NeighEll E;
  if (rc == ISPH_SUCCESS) rc = build_neigh_ell(ctx, n, np, a.nidx, E, T, a.colmap);
  if (rc == ISPH_SUCCESS)
    for (int t = 1; t <= P->ntypes; ++t)
      if (P->kind[t] != KIND_FLUID && P->kind[t] != KIND_SOLID) rc = fail("only fluid/solid particle kinds are supported", __FILE__, __LINE__);
  // structure of the first block; the others copy it
  isph_mat *A0 = nullptr;
  if (rc == ISPH_SUCCESS) {
    A0 = new isph_mat();
    blk[0] = A0;
    Sell &M = A0->S;
    M.nrow = n; M.ncol = ncol; M.nslices = (n + kSlice - 1) / kSlice;
    rc = M.rowlen.reserve((size_t)(n > 0 ? n : 1));
    if (rc == ISPH_SUCCESS) rc = M.slice_off.reserve((size_t)M.nslices + 1);
  }
  double *db = b_out;
  if (rc == ISPH_SUCCESS && !on_device) { rc = bdev.reserve((size_t)lda * dim); db = bdev.p; }
  if (rc == ISPH_SUCCESS && n > 0) {
    Sell &M = A0->S;
    const int grid = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_asm_count, dim3(xcd_grid(grid)), dim3(kBlock), 0, ctx->stream, T, n, a.x, a.type, a.nptr, a.nidx, M.rowlen.p);
    hipLaunchKernelGGL(k_slicew_from_rowlen, dim3(grid), dim3(kBlock), 0, ctx->stream, n, M.rowlen.p, M.slice_off.p);
    rc = sell_finalize_offsets(ctx, M);
    for (int q = 1; q < d2 && rc == ISPH_SUCCESS; ++q) {
      const bool diagblk = (q / dim) == (q % dim);
      if (!normal && !diagblk) continue;
      blk[q] = new isph_mat();
      Sell &B = blk[q]->S;
      B.nrow = n; B.ncol = ncol; B.nslices = M.nslices; B.stored = M.stored;
      rc = B.rowlen.reserve((size_t)n);
      if (rc == ISPH_SUCCESS) rc = B.slice_off.reserve((size_t)M.nslices + 1);
      if (rc == ISPH_SUCCESS) rc = B.col.reserve((size_t)(M.stored > 0 ? M.stored : 1));
      if (rc == ISPH_SUCCESS) rc = B.val.reserve((size_t)(M.stored > 0 ? M.stored : 1));
      if (rc == ISPH_SUCCESS &&
          (hipMemcpyAsync(B.rowlen.p, M.rowlen.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
           hipMemcpyAsync(B.slice_off.p, M.slice_off.p, sizeof(long long) * ((size_t)M.nslices + 1), hipMemcpyDeviceToDevice,
                          ctx->stream) != hipSuccess))
        rc = fail("copy failed", __FILE__, __LINE__);
    }
    if (rc == ISPH_SUCCESS) {
      for (int q = 0; q < d2; ++q)
        if (blk[q]) { A.bcol[q] = blk[q]->S.col.p; A.bval[q] = blk[q]->S.val.p; }
      a.nlocal = n; a.antisym = antisym; a.incremental = incremental; a.lda = lda; a.dt = dt; a.theta = theta;
      a.morris = P->morris_holmes ? 1 : 0; a.safe = P->morris_safe_coeff;
      A.beta = beta;
      for (int k = 0; k < 3; ++k) a.g[k] = gvec ? gvec[k] : 0.0;
      const int gridp = M.nslices * kSlice / kBlock + ((M.nslices * kSlice) % kBlock ? 1 : 0);
      hipLaunchKernelGGL(k_asm_block_helmholtz, dim3(xcd_grid(gridp)), dim3(kBlock), 0, ctx->stream, T, A, M.slice_off.p, db);
      for (int q = 0; q < d2 && rc == ISPH_SUCCESS; ++q) {
        if (!blk[q]) continue;
        Sell &B = blk[q]->S;
        if (n <= 32768) {
          rc = newlen.reserve((size_t)n);
          if (rc == ISPH_SUCCESS) {
            hipLaunchKernelGGL(k_sell_merge_duplicates, dim3(grid), dim3(kBlock), 0, ctx->stream, n, B.rowlen.p, B.slice_off.p,
                               B.col.p, B.val.p, newlen.p);
            if (hipMemcpyAsync(B.rowlen.p, newlen.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
              rc = fail("copy failed", __FILE__, __LINE__);
          }
        }
        if (rc == ISPH_SUCCESS) rc = (T.sorted && n > 32768) ? sell_set_wmax(ctx, B) : sell_sort_rows(ctx, B);
      }
      // Above the merge threshold every block keeps the scalar pattern as the row kernel wrote it: the 16-bit window
      // columns of the SpMV are built once, for block (0,0), and copied (0.2 ms per block instead of a 1 ms rebuild on
      // each block's first SpMV; a 2 M-particle cavity step has nine of them)
      if (rc == ISPH_SUCCESS && T.sorted && n > 32768 && sell_cols16(ctx, A0->S)) {
        const Sell &S0 = A0->S;
        for (int q = 1; q < d2 && rc == ISPH_SUCCESS; ++q) {
          if (!blk[q]) continue;
          Sell &B = blk[q]->S;
          rc = B.col16.reserve((size_t)S0.stored);
          if (rc == ISPH_SUCCESS) rc = B.wtab.reserve((size_t)S0.nslices * 64);
          if (rc == ISPH_SUCCESS &&
              (hipMemcpyAsync(B.col16.p, S0.col16.p, sizeof(unsigned short) * (size_t)S0.stored, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
               hipMemcpyAsync(B.wtab.p, S0.wtab.p, sizeof(int) * (size_t)S0.nslices * 64, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess))
            rc = fail("copy failed", __FILE__, __LINE__);
          if (rc == ISPH_SUCCESS) B.c16_state = 1;
        }
      }
      if (rc == ISPH_SUCCESS && !on_device &&
          hipMemcpyAsync(b_out, db, sizeof(double) * (size_t)lda * dim, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        rc = fail("copy failed", __FILE__, __LINE__);
      if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess)
        rc = fail("block assembly kernel failed", __FILE__, __LINE__);
    }
    if (rc == ISPH_SUCCESS) {
      long long tot = 0;
      rc = sell_sum_rowlen(ctx, A0->S, &tot);
      for (int q = 0; q < d2; ++q)
        if (blk[q]) blk[q]->S.nnz = tot;
    }
  }
  S.release(); E.release(); snu.release(); sp.release(); sf.release(); sv.release(); sn.release();
  bdev.release();
  newlen.release();
  if (rc != ISPH_SUCCESS) {
    for (int q = 0; q < 9; ++q)
      if (blk[q]) { blk[q]->S.release(); delete blk[q]; }
    return rc;
  }
  for (int q = 0; q < d2; ++q) blocks_out[q] = blk[q];
  return ISPH_SUCCESS;
}

}  // namespace isph
