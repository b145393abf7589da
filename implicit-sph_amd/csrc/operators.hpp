// operators.hpp -- the streaming SPH operators either side of the solve
// (SURVEY §8(f).2): corrected gradient / divergence and the velocity / pressure
// corrections and particle advance built on them, so a whole ISPH step can stay
// on the device between neighbour rebuilds.
//
// Replaces
//   Corrected::FunctorOuterGradient<.,AS>      (ref: functor_gradient.h:78-170)
//   Corrected::FunctorOuterDivergence<.,AS>    (ref: functor_divergence.h:54-124)
//   FunctorOuterCorrectVelocity                (ref: functor_correct_velocity.h:42-78)
//   FunctorOuterCorrectPressure                (ref: functor_correct_pressure.h:30-45)
//   FunctorOuterAdvanceTimeBegin / End         (ref: functor_advance_time_begin.h:40-75, functor_advance_time_end.h:45-72)
// One lane per particle, neighbour list read through the lane-interleaved copy
// (assemble.hpp), MirrorNothing coefficients.
#pragma once
#include "assemble.hpp"

namespace isph {

struct OpArgs {
  int nlocal, antisym, use_filter, filt_i, filt_j;
  double alpha;
  const double *x, *vfrac, *Gc;
  const int *type, *nptr;
};

// grad_i = alpha sum_j (G^T r_ij) (f_i (+|-) f_j) W'/r V    (scalar field f[nall])
__global__ __launch_bounds__(kBlock) void k_gradient(AsmTables T, OpArgs a, const double *__restrict__ f,
                                                     double *__restrict__ grad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.nlocal) return;
  const int dim = T.dim, nt1 = T.ntypes + 1, it = a.type[i], ikind = T.kind[it];
  double g[3] = {0, 0, 0};
  if (!a.use_filter || (ikind & a.filt_i)) {
    double G[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (dim == 2) { G[0] = 1; G[1] = 0; G[2] = 0; G[3] = 1; }
    if (!a.antisym)
      for (int k = 0; k < dim * dim; ++k) G[k] = a.Gc[(size_t)i * dim * dim + k];
    const double vi = a.vfrac[i], fi = f[i];
    const int jb = a.nptr[i], je = a.nptr[i + 1];
    for (int jj = jb; jj < je; ++jj) {
      const int j = neigh_at(T, i, jj - jb);
      const int jt = a.type[j], jkind = T.kind[jt];
      if (a.use_filter && !((ikind & a.filt_i) && (jkind & a.filt_j))) continue;
      double rij[3];
      const double rsq = pair_rsq(dim, a.x, i, j, rij);
      if (!(rsq < T.cutsq[it * nt1 + jt])) continue;
      const double r = sqrt(rsq) + kEps;
      const double dwdr = kernel_dval(T.kernel, r, T.hinv[it * nt1 + jt], T.kdnorm[it * nt1 + jt]);
      const double vfrac = a.antisym ? sqrt(vi * a.vfrac[j]) : a.vfrac[j];
      const double vjtmp = dwdr / r * vfrac;
      const double df = a.antisym ? (fi + f[j]) : (f[j] - fi);
      for (int k2 = 0; k2 < dim; ++k2) {
        double gitmp = 0.0;
        for (int k1 = 0; k1 < dim; ++k1) gitmp += G[k2 * dim + k1] * rij[k1];
        g[k2] += gitmp * vjtmp * df;
      }
    }
  }
  for (int k = 0; k < 3; ++k) grad[3 * (size_t)i + k] = k < dim ? g[k] * a.alpha : 0.0;
}

// div_i = alpha sum_j (G^T r_ij).(f_i (+|-) f_j) W'/r V     (vector field f[nall][3])
__global__ __launch_bounds__(kBlock) void k_divergence(AsmTables T, OpArgs a, const double *__restrict__ f,
                                                       double *__restrict__ div) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.nlocal) return;
  const int dim = T.dim, nt1 = T.ntypes + 1, it = a.type[i], ikind = T.kind[it];
  double d = 0.0;
  if (!a.use_filter || (ikind & a.filt_i)) {
    double G[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (dim == 2) { G[0] = 1; G[1] = 0; G[2] = 0; G[3] = 1; }
    if (!a.antisym)
      for (int k = 0; k < dim * dim; ++k) G[k] = a.Gc[(size_t)i * dim * dim + k];
    const double vi = a.vfrac[i];
    const int jb = a.nptr[i], je = a.nptr[i + 1];
    for (int jj = jb; jj < je; ++jj) {
      const int j = neigh_at(T, i, jj - jb);
      const int jt = a.type[j], jkind = T.kind[jt];
      if (a.use_filter && !((ikind & a.filt_i) && (jkind & a.filt_j))) continue;
      double rij[3];
      const double rsq = pair_rsq(dim, a.x, i, j, rij);
      if (!(rsq < T.cutsq[it * nt1 + jt])) continue;
      const double r = sqrt(rsq) + kEps;
      const double dwdr = kernel_dval(T.kernel, r, T.hinv[it * nt1 + jt], T.kdnorm[it * nt1 + jt]);
      const double vfrac = a.antisym ? sqrt(vi * a.vfrac[j]) : a.vfrac[j];
      const double vjtmp = dwdr / r * vfrac;
      for (int k2 = 0; k2 < dim; ++k2) {
        double gitmp = 0.0;
        for (int k1 = 0; k1 < dim; ++k1) gitmp += G[k2 * dim + k1] * rij[k1];
        const double fi = f[3 * (size_t)i + k2], fj = f[3 * (size_t)j + k2];
        d += gitmp * (a.antisym ? (fi + fj) : (fj - fi)) * vjtmp;
      }
    }
  }
  div[i] = d * a.alpha;
}

// vstar_i -= dt/rho_i grad_i   for fluid particles (functor_correct_velocity.h:58-70)
__global__ void k_correct_velocity(int nlocal, int dim, double dt, const int *__restrict__ type,
                                   const int *__restrict__ kind, const double *__restrict__ rho,
                                   const double *__restrict__ grad, double *__restrict__ vstar) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nlocal) return;
  if (!(kind[type[i]] & KIND_FLUID)) return;
  for (int k = 0; k < dim; ++k) vstar[3 * (size_t)i + k] -= dt / rho[i] * grad[3 * (size_t)i + k];
}

// p += dp (incremental) or p = dp, over nlocal+nghost (functor_correct_pressure.h:30-45)
__global__ void k_correct_pressure(int n, int incremental, const double *__restrict__ dp, double *__restrict__ p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  p[i] = incremental ? p[i] + dp[i] : dp[i];
}

// AdvanceTimeBegin: dp_i = grad p_i . dx_i, dx = dt/2 (vnp1 + v)  (fluid only)
__global__ void k_advance_begin(int nlocal, int dim, double dt, const int *__restrict__ type,
                                const int *__restrict__ kind, const double *__restrict__ gradp,
                                const double *__restrict__ v, const double *__restrict__ vnp1, double *__restrict__ dp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nlocal) return;
  double s = 0.0;
  if (kind[type[i]] & KIND_FLUID)
    for (int k = 0; k < dim; ++k) s += gradp[3 * (size_t)i + k] * (0.5 * dt * (vnp1[3 * (size_t)i + k] + v[3 * (size_t)i + k]));
  dp[i] = s;
}

// AdvanceTimeEnd over nlocal+nghost: p += dp; x += dt/2 (vnp1+v); v = vnp1
__global__ void k_advance_end(int n, int dim, double dt, const double *__restrict__ dp, const double *__restrict__ vnp1,
                              double *__restrict__ p, double *__restrict__ x, double *__restrict__ v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  p[i] += dp[i];
  for (int k = 0; k < dim; ++k) {
    const double delta = 0.5 * dt * (vnp1[3 * (size_t)i + k] + v[3 * (size_t)i + k]);
    x[3 * (size_t)i + k] += delta;
    v[3 * (size_t)i + k] = vnp1[3 * (size_t)i + k];
  }
}

// common staging for the neighbour-sweep operators
struct OpStage {
  StagedParticles S;
  NeighEll E;
  AsmTables T;
  OpArgs a;
  DevBuf<double> fin, out;
  void release() { S.release(); E.release(); fin.release(); out.release(); }
};

inline int op_stage(isph_ctx *ctx, const isph_particles *P, int antisym, int on_device, OpStage &st) {
  ISPH_REQUIRE(P->dim == 2 || P->dim == 3, "dim must be 2 or 3");
  ISPH_REQUIRE(P->x && P->type && P->neigh_ptr && P->neigh_idx && P->vfrac, "particle arrays missing");
  ISPH_REQUIRE(antisym || P->Gc, "Symmetric family needs Gc");
  const int n = P->nlocal, dim = P->dim;
  memset(&st.a, 0, sizeof(st.a));
  ISPH_CHECK(stage_tables(ctx, P, st.S, st.T));
  const int *di = nullptr;
  long long nnb = 0;
  ISPH_CHECK(stage(ctx, P->x, (size_t)P->nall * 3, on_device, st.S.x, &st.a.x));
  ISPH_CHECK(stage(ctx, P->type, (size_t)P->nall, on_device, st.S.type, &st.a.type));
  ISPH_CHECK(stage(ctx, P->vfrac, (size_t)P->nall, on_device, st.S.vfrac, &st.a.vfrac));
  ISPH_CHECK(stage(ctx, P->Gc, (size_t)P->nall * dim * dim, on_device, st.S.Gc, &st.a.Gc));
  ISPH_CHECK(stage(ctx, P->neigh_ptr, (size_t)n + 1, on_device, st.S.nptr, &st.a.nptr));
  if (on_device) {
    int last = 0;
    ISPH_CHECK_HIP(hipMemcpyAsync(&last, P->neigh_ptr + n, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    nnb = last;
  } else {
    nnb = P->neigh_ptr[n];
    for (long long k = 0; k < nnb; ++k)
      ISPH_REQUIRE(P->neigh_idx[k] >= 0 && P->neigh_idx[k] < P->nall, "neighbour index out of range");
  }
  ISPH_CHECK(stage(ctx, P->neigh_idx, (size_t)nnb, on_device, st.S.nidx, &di));
  ISPH_CHECK(build_neigh_ell(ctx, n, st.a.nptr, di, st.E, st.T));
  st.a.nlocal = n;
  st.a.antisym = antisym;
  return ISPH_SUCCESS;
}

// mode 0: gradient of a scalar [nall] -> [nlocal][3]; mode 1: divergence of a vector [nall][3] -> [nlocal]
inline int op_apply(isph_ctx *ctx, const isph_particles *P, int mode, int antisym, const double *f, double alpha,
                    int use_filter, int filt_i, int filt_j, double *out, int on_device) {
  OpStage st;
  int rc = op_stage(ctx, P, antisym, on_device, st);
  const int n = P->nlocal;
  const size_t nin = (size_t)P->nall * (mode == 0 ? 1 : 3), nout = (size_t)n * (mode == 0 ? 3 : 1);
  const double *df = nullptr;
  double *dout = out;
  if (rc == ISPH_SUCCESS) rc = stage(ctx, f, nin, on_device, st.fin, &df);
  if (rc == ISPH_SUCCESS && !on_device) { rc = st.out.reserve(nout > 0 ? nout : 1); dout = st.out.p; }
  if (rc == ISPH_SUCCESS && n > 0) {
    st.a.alpha = alpha; st.a.use_filter = use_filter; st.a.filt_i = filt_i; st.a.filt_j = filt_j;
    const int grid = (n + kBlock - 1) / kBlock;
    if (mode == 0) hipLaunchKernelGGL(k_gradient, dim3(grid), dim3(kBlock), 0, ctx->stream, st.T, st.a, df, dout);
    else hipLaunchKernelGGL(k_divergence, dim3(grid), dim3(kBlock), 0, ctx->stream, st.T, st.a, df, dout);
    if (!on_device && hipMemcpyAsync(out, dout, sizeof(double) * nout, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
      rc = fail("copy failed", __FILE__, __LINE__);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess)
      rc = fail("operator kernel failed", __FILE__, __LINE__);
  }
  st.release();
  return rc;
}

// in/out staging for operands that are updated in place
struct InOut {
  DevBuf<double> buf;
  double *host = nullptr, *dev = nullptr;
  size_t n = 0;
  int open(isph_ctx *ctx, double *p, size_t count, int on_device) {
    n = count;
    if (on_device) { dev = p; return ISPH_SUCCESS; }
    host = p;
    ISPH_CHECK(buf.reserve(count > 0 ? count : 1));
    ISPH_CHECK_HIP(hipMemcpyAsync(buf.p, p, sizeof(double) * count, hipMemcpyHostToDevice, ctx->stream));
    dev = buf.p;
    return ISPH_SUCCESS;
  }
  int close(isph_ctx *ctx) {
    if (host) ISPH_CHECK_HIP(hipMemcpyAsync(host, dev, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    return ISPH_SUCCESS;
  }
};

// PairISPH_Corrected::correctVelocity + correctPressure (ref: pair_isph_corrected.cpp:1020-1050):
// vstar[nlocal rows of nall][3] -= dt/rho grad(dp), filter (Fluid,Fluid); p[nall] (+)= dp[nall]
inline int correct_velocity_pressure(isph_ctx *ctx, const isph_particles *P, int antisym, double dt, const double *rho,
                                     const double *dp, double *vstar, double *p, int incremental, int on_device) {
  OpStage st;
  int rc = op_stage(ctx, P, antisym, on_device, st);
  const int n = P->nlocal;
  DevBuf<double> grad, srho, sdp;
  InOut iv, ip;
  const double *drho = nullptr, *ddp = nullptr;
  if (rc == ISPH_SUCCESS) rc = grad.reserve((size_t)(n > 0 ? n : 1) * 3);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, rho, (size_t)P->nall, on_device, srho, &drho);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, dp, (size_t)P->nall, on_device, sdp, &ddp);
  if (rc == ISPH_SUCCESS) rc = iv.open(ctx, vstar, (size_t)P->nall * 3, on_device);
  if (rc == ISPH_SUCCESS) rc = ip.open(ctx, p, (size_t)P->nall, on_device);
  if (rc == ISPH_SUCCESS && n > 0) {
    st.a.alpha = 1.0; st.a.use_filter = 1; st.a.filt_i = KIND_FLUID; st.a.filt_j = KIND_FLUID;
    const int grid = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_gradient, dim3(grid), dim3(kBlock), 0, ctx->stream, st.T, st.a, ddp, grad.p);
    hipLaunchKernelGGL(k_correct_velocity, dim3(grid), dim3(kBlock), 0, ctx->stream, n, P->dim, dt, st.a.type, st.T.kind,
                       drho, (const double *)grad.p, iv.dev);
    hipLaunchKernelGGL(k_correct_pressure, dim3((P->nall + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, P->nall,
                       incremental, ddp, ip.dev);
    if (iv.close(ctx) != ISPH_SUCCESS || ip.close(ctx) != ISPH_SUCCESS) rc = ISPH_FAILURE;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess)
      rc = fail("correction kernels failed", __FILE__, __LINE__);
  }
  grad.release(); srho.release(); sdp.release(); iv.buf.release(); ip.buf.release();
  st.release();
  return rc;
}

// FunctorOuterAdvanceTimeBegin (ref: functor_advance_time_begin.h:40-75): dp_out[nlocal] = grad p . dt/2 (vnp1 + v)
inline int advance_begin(isph_ctx *ctx, const isph_particles *P, int antisym, double dt, const double *p, const double *v,
                         const double *vnp1, double *dp_out, int on_device) {
  OpStage st;
  int rc = op_stage(ctx, P, antisym, on_device, st);
  const int n = P->nlocal;
  DevBuf<double> grad, sp, sv, svn, sout;
  const double *dpp = nullptr, *dv = nullptr, *dvn = nullptr;
  double *dout = dp_out;
  if (rc == ISPH_SUCCESS) rc = grad.reserve((size_t)(n > 0 ? n : 1) * 3);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, p, (size_t)P->nall, on_device, sp, &dpp);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, v, (size_t)P->nall * 3, on_device, sv, &dv);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, vnp1, (size_t)P->nall * 3, on_device, svn, &dvn);
  if (rc == ISPH_SUCCESS && !on_device) { rc = sout.reserve((size_t)(n > 0 ? n : 1)); dout = sout.p; }
  if (rc == ISPH_SUCCESS && n > 0) {
    st.a.alpha = 1.0; st.a.use_filter = 1; st.a.filt_i = KIND_FLUID; st.a.filt_j = KIND_FLUID;
    const int grid = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_gradient, dim3(grid), dim3(kBlock), 0, ctx->stream, st.T, st.a, dpp, grad.p);
    hipLaunchKernelGGL(k_advance_begin, dim3(grid), dim3(kBlock), 0, ctx->stream, n, P->dim, dt, st.a.type, st.T.kind,
                       (const double *)grad.p, dv, dvn, dout);
    if (!on_device && hipMemcpyAsync(dp_out, dout, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
      rc = fail("copy failed", __FILE__, __LINE__);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess)
      rc = fail("advance kernels failed", __FILE__, __LINE__);
  }
  grad.release(); sp.release(); sv.release(); svn.release(); sout.release();
  st.release();
  return rc;
}

// FunctorOuterAdvanceTimeEnd over the first `count` atoms (nlocal + nghost in the reference,
// functor_advance_time_end.h:36-72): p += dp; x += dt/2 (vnp1 + v); v = vnp1
inline int advance_end(isph_ctx *ctx, int count, int dim, double dt, const double *dp, const double *vnp1, double *p,
                       double *x, double *v, int on_device) {
  ISPH_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
  DevBuf<double> sdp, svn;
  InOut ip, ix, iv;
  const double *ddp = nullptr, *dvn = nullptr;
  int rc = stage(ctx, dp, (size_t)count, on_device, sdp, &ddp);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, vnp1, (size_t)count * 3, on_device, svn, &dvn);
  if (rc == ISPH_SUCCESS) rc = ip.open(ctx, p, (size_t)count, on_device);
  if (rc == ISPH_SUCCESS) rc = ix.open(ctx, x, (size_t)count * 3, on_device);
  if (rc == ISPH_SUCCESS) rc = iv.open(ctx, v, (size_t)count * 3, on_device);
  if (rc == ISPH_SUCCESS && count > 0) {
    hipLaunchKernelGGL(k_advance_end, dim3((count + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, count, dim, dt, ddp,
                       dvn, ip.dev, ix.dev, iv.dev);
    if (ip.close(ctx) != ISPH_SUCCESS || ix.close(ctx) != ISPH_SUCCESS || iv.close(ctx) != ISPH_SUCCESS) rc = ISPH_FAILURE;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess)
      rc = fail("advance kernel failed", __FILE__, __LINE__);
  }
  sdp.release(); svn.release(); ip.buf.release(); ix.buf.release(); iv.buf.release();
  return rc;
}

}  // namespace isph
