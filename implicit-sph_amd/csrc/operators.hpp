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
//   FunctorOuterComputeShift / ApplyShift      (ref: functor_compute_shift.h:48-113, functor_apply_shift.h:76-108)
//   PairISPH_Corrected::shiftParticles         (ref: pair_isph_corrected.cpp:1203-1262)
// One lane per particle, neighbour list read through the lane-interleaved copy
// (assemble.hpp), MirrorNothing coefficients.
#pragma once
#include "assemble.hpp"
#include "comm.hpp"

namespace isph {

struct OpArgs {
  int nlocal, antisym, use_filter, filt_i, filt_j;
  double alpha;
  const double *x, *vfrac, *Gc;
  const int *type, *nptr;
};

// grad_i = alpha sum_j (G^T r_ij) (f_i (+|-) f_j) W'/r V    (scalar field f[nall])
template <int DIMT, int FAM>  // 0 / -1: dimension and family read at run time; 3 / 1: 3-D AntiSymmetric (G = I) folded
__global__ __launch_bounds__(kBlock) void k_gradient(AsmTables T, OpArgs a, const double *__restrict__ f,
                                                     double *__restrict__ grad) {
  const int i = xcd_block() * blockDim.x + threadIdx.x;
  if (i >= a.nlocal) return;
  const int dim = DIMT ? DIMT : T.dim, nt1 = T.ntypes + 1, it = a.type[i], ikind = T.kind[it];
  const bool antisym = FAM < 0 ? (a.antisym != 0) : (FAM != 0);
  double g[3] = {0, 0, 0};
  if (!a.use_filter || (ikind & a.filt_i)) {
    double G[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (dim == 2) { G[0] = 1; G[1] = 0; G[2] = 0; G[3] = 1; }
    if (!antisym)
      for (int k = 0; k < dim * dim; ++k) G[k] = a.Gc[(size_t)i * dim * dim + k];
    const double vi = a.vfrac[i], fi = f[i];
    const int jb = 0, je = T.nlen[i];
    for (int jj = jb; jj < je; ++jj) {
      const int j = neigh_at(T, i, jj - jb);
      const int jt = a.type[j], jkind = T.kind[jt];
      if (a.use_filter && !((ikind & a.filt_i) && (jkind & a.filt_j))) continue;
      double rij[3];
      const double rsq = pair_rsq(dim, a.x, i, j, rij);
      if (!(rsq < T.cutsq[it * nt1 + jt])) continue;
      const double r = sqrt(rsq) + kEps;
      const double dwdr = kernel_dval(T.kernel, r, T.hinv[it * nt1 + jt], T.kdnorm[it * nt1 + jt]);
      const double vfrac = antisym ? sqrt(vi * a.vfrac[j]) : a.vfrac[j];
      const double vjtmp = dwdr / r * vfrac;
      const double df = antisym ? (fi + f[j]) : (f[j] - fi);
      for (int k2 = 0; k2 < dim; ++k2) {
        double gitmp = 0.0;
        if (antisym) gitmp = rij[k2];  // G = I: the sum below gives exactly this
        else for (int k1 = 0; k1 < dim; ++k1) gitmp += G[k2 * dim + k1] * rij[k1];
        g[k2] += gitmp * vjtmp * df;
      }
    }
  }
  for (int k = 0; k < 3; ++k) grad[3 * (size_t)i + k] = k < dim ? g[k] * a.alpha : 0.0;
}

// div_i = alpha sum_j (G^T r_ij).(f_i (+|-) f_j) W'/r V     (vector field f[nall][3])
template <int DIMT, int FAM>  // 0 / -1: dimension and family read at run time; 3 / 1: 3-D AntiSymmetric (G = I) folded
__global__ __launch_bounds__(kBlock) void k_divergence(AsmTables T, OpArgs a, const double *__restrict__ f,
                                                       double *__restrict__ div) {
  const int i = xcd_block() * blockDim.x + threadIdx.x;
  if (i >= a.nlocal) return;
  const int dim = DIMT ? DIMT : T.dim, nt1 = T.ntypes + 1, it = a.type[i], ikind = T.kind[it];
  const bool antisym = FAM < 0 ? (a.antisym != 0) : (FAM != 0);
  double d = 0.0;
  if (!a.use_filter || (ikind & a.filt_i)) {
    double G[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (dim == 2) { G[0] = 1; G[1] = 0; G[2] = 0; G[3] = 1; }
    if (!antisym)
      for (int k = 0; k < dim * dim; ++k) G[k] = a.Gc[(size_t)i * dim * dim + k];
    const double vi = a.vfrac[i];
    const int jb = 0, je = T.nlen[i];
    for (int jj = jb; jj < je; ++jj) {
      const int j = neigh_at(T, i, jj - jb);
      const int jt = a.type[j], jkind = T.kind[jt];
      if (a.use_filter && !((ikind & a.filt_i) && (jkind & a.filt_j))) continue;
      double rij[3];
      const double rsq = pair_rsq(dim, a.x, i, j, rij);
      if (!(rsq < T.cutsq[it * nt1 + jt])) continue;
      const double r = sqrt(rsq) + kEps;
      const double dwdr = kernel_dval(T.kernel, r, T.hinv[it * nt1 + jt], T.kdnorm[it * nt1 + jt]);
      const double vfrac = antisym ? sqrt(vi * a.vfrac[j]) : a.vfrac[j];
      const double vjtmp = dwdr / r * vfrac;
      for (int k2 = 0; k2 < dim; ++k2) {
        double gitmp = 0.0;
        if (antisym) gitmp = rij[k2];  // G = I: the sum below gives exactly this
        else for (int k1 = 0; k1 < dim; ++k1) gitmp += G[k2 * dim + k1] * rij[k1];
        const double fi = f[3 * (size_t)i + k2], fj = f[3 * (size_t)j + k2];
        d += gitmp * (antisym ? (fi + fj) : (fj - fi)) * vjtmp;
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

// max over fluid particles of |v| (pair_isph_corrected.cpp:1216-1236); non-negative doubles order like their bit patterns
__global__ void k_max_fluid_speed(int nlocal, int dim, const int *__restrict__ type, const int *__restrict__ kind,
                                  const double *__restrict__ v, unsigned long long *__restrict__ vmax_bits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double mag = 0.0;
  if (i < nlocal && (kind[type[i]] & KIND_FLUID)) {
    double s = 0.0;
    for (int k = 0; k < dim; ++k) { const double a = fabs(v[3 * (size_t)i + k]); s += a * a; }
    mag = sqrt(s);
  }
  for (int o = 32; o > 0; o >>= 1) mag = fmax(mag, __shfl_xor(mag, o, 64));
  if ((threadIdx.x & 63) == 0 && mag > 0.0) atomicMax(vmax_bits, (unsigned long long)__double_as_longlong(mag));
}

// dr_i = sum_j beta_ij r_ij, beta = alpha/r (ri/r)^2 (1 + [j not fluid] w (ri/r)^2), ri = mean neighbour distance;
// fluid rows, filter (Fluid, All), pairs inside min(cutsq, shiftcut^2); alpha = alpha0 * (*scale) when scale != NULL
__global__ __launch_bounds__(kBlock) void k_compute_shift(AsmTables T, OpArgs a, double alpha0,
                                                          const double *__restrict__ scale, double shiftcutsq,
                                                          double nonfluidweight, double *__restrict__ dr) {
  const int i = xcd_block() * blockDim.x + threadIdx.x;
  if (i >= a.nlocal) return;
  const int dim = T.dim, nt1 = T.ntypes + 1, it = a.type[i], ikind = T.kind[it];
  double d[3] = {0, 0, 0};
  if (ikind & KIND_FLUID) {
    const double alpha = scale ? alpha0 * scale[0] : alpha0;
    const int jb = 0, je = T.nlen[i];
    int cnt = 0;
    double ri = 0.0;
    for (int jj = jb; jj < je; ++jj) {
      const int j = neigh_at(T, i, jj - jb);
      const int jt = a.type[j];
      if (!(T.kind[jt] & KIND_ALL)) continue;
      double rij[3];
      const double rsq = pair_rsq(dim, a.x, i, j, rij);
      if (rsq < fmin(T.cutsq[it * nt1 + jt], shiftcutsq)) { ++cnt; ri += sqrt(rsq); }
    }
    if (cnt) ri /= (double)cnt;
    for (int jj = jb; jj < je; ++jj) {
      const int j = neigh_at(T, i, jj - jb);
      const int jt = a.type[j], jkind = T.kind[jt];
      if (!(jkind & KIND_ALL)) continue;
      double rij[3];
      const double rsq = pair_rsq(dim, a.x, i, j, rij);
      if (rsq < fmin(T.cutsq[it * nt1 + jt], shiftcutsq)) {
        const double r = sqrt(rsq) + kEps, q = ri / r, rir2 = q * q;
        const double beta = alpha / r * rir2 * (1.0 + ((jkind & KIND_FLUID) ? 0.0 : nonfluidweight) * rir2);
        for (int k = 0; k < dim; ++k) d[k] += beta * rij[k];
      }
    }
  }
  for (int k = 0; k < 3; ++k) dr[3 * (size_t)i + k] = d[k];
}

// ApplyShift: p_i += grad p . dr_i, v_i^k += grad v^k . dr_i, x_i += dr_i (gradients with filter (Fluid, All)).
// Every row reads the PRE-shift x / p / v and writes to xn / vn / pn: the reference's serial loop updates
// them in place, so its row i sees rows < i already shifted -- an ordering artefact of O(|dr|^2) that no
// parallel execution can (or should) reproduce; the oracle restates both and the tests bound the gap.
__global__ __launch_bounds__(kBlock) void k_apply_shift(AsmTables T, OpArgs a, const int *__restrict__ fixed,
                                                        const double *__restrict__ dr, const double *__restrict__ xin,
                                                        const double *__restrict__ v, const double *__restrict__ p,
                                                        double *__restrict__ xn, double *__restrict__ vn,
                                                        double *__restrict__ pn) {
  const int i = xcd_block() * blockDim.x + threadIdx.x;
  if (i >= a.nlocal) return;
  const int dim = T.dim, nt1 = T.ntypes + 1, it = a.type[i], ikind = T.kind[it];
  double gp[3] = {0, 0, 0}, gv[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  const double pi = p[i];
  double vi3[3], xi3[3], di[3];
  for (int k = 0; k < 3; ++k) { vi3[k] = v[3 * (size_t)i + k]; xi3[k] = xin[3 * (size_t)i + k]; di[k] = dr[3 * (size_t)i + k]; }
  const bool moves = !(fixed && fixed[it]);
  if (moves && (ikind & KIND_FLUID)) {
    double G[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (dim == 2) { G[0] = 1; G[1] = 0; G[2] = 0; G[3] = 1; }
    if (!a.antisym)
      for (int k = 0; k < dim * dim; ++k) G[k] = a.Gc[(size_t)i * dim * dim + k];
    const double vfi = a.vfrac[i];
    const int jb = 0, je = T.nlen[i];
    for (int jj = jb; jj < je; ++jj) {
      const int j = neigh_at(T, i, jj - jb);
      const int jt = a.type[j];
      if (!(T.kind[jt] & KIND_ALL)) continue;
      double rij[3];
      const double rsq = pair_rsq(dim, a.x, i, j, rij);
      if (!(rsq < T.cutsq[it * nt1 + jt])) continue;
      const double r = sqrt(rsq) + kEps;
      const double dwdr = kernel_dval(T.kernel, r, T.hinv[it * nt1 + jt], T.kdnorm[it * nt1 + jt]);
      const double vfrac = a.antisym ? sqrt(vfi * a.vfrac[j]) : a.vfrac[j];
      const double vjtmp = dwdr / r * vfrac;
      const double dfp = a.antisym ? (pi + p[j]) : (p[j] - pi);
      double dfv[3];
      for (int k = 0; k < dim; ++k) {
        const double vj = v[3 * (size_t)j + k];
        dfv[k] = a.antisym ? (vi3[k] + vj) : (vj - vi3[k]);
      }
      for (int k2 = 0; k2 < dim; ++k2) {
        double gitmp = 0.0;
        for (int k1 = 0; k1 < dim; ++k1) gitmp += G[k2 * dim + k1] * rij[k1];
        const double ijtmp = gitmp * vjtmp;
        gp[k2] += ijtmp * dfp;
        for (int k1 = 0; k1 < dim; ++k1) gv[k1][k2] += ijtmp * dfv[k1];
      }
    }
  }
  double pnew = pi;
  if (moves) {
    double s = 0.0;
    for (int k = 0; k < dim; ++k) s += gp[k] * di[k];
    pnew += s;
    for (int k = 0; k < dim; ++k) {
      double t = 0.0;
      for (int q = 0; q < dim; ++q) t += gv[k][q] * di[q];
      vi3[k] += t;
      xi3[k] += di[k];
    }
  }
  pn[i] = pnew;
  for (int k = 0; k < 3; ++k) { vn[3 * (size_t)i + k] = vi3[k]; xn[3 * (size_t)i + k] = xi3[k]; }
}

__global__ void k_shift_commit(int nlocal, const double *__restrict__ xn, const double *__restrict__ vn,
                               const double *__restrict__ pn, double *__restrict__ x, double *__restrict__ v,
                               double *__restrict__ p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nlocal) return;
  p[i] = pn[i];
  for (int k = 0; k < 3; ++k) { x[3 * (size_t)i + k] = xn[3 * (size_t)i + k]; v[3 * (size_t)i + k] = vn[3 * (size_t)i + k]; }
}

// common staging for the neighbour-sweep operators
struct OpStage {
  StagedParticles S;
  NeighEll E;
  AsmTables T;
  OpArgs a;
  DevTmp<double> fin, out;
  void release() { S.release(); E.release(); fin.release(); out.release(); }
};

inline int op_stage(isph_ctx *ctx, const isph_particles *P, int antisym, int on_device, OpStage &st) {
  ISPH_REQUIRE(P->dim == 2 || P->dim == 3, "dim must be 2 or 3");
  ISPH_REQUIRE(P->x && P->type && (P->neigh_ptr || P->neigh_ptr64) && P->neigh_idx && P->vfrac, "particle arrays missing");
  ISPH_REQUIRE(antisym || P->Gc, "Symmetric family needs Gc");
  const int n = P->nlocal, dim = P->dim;
  memset(&st.a, 0, sizeof(st.a));
  ISPH_CHECK(stage_tables(ctx, P, st.S, st.T));
  const int *di = nullptr;
  long long nnb = 0;
  ISPH_CHECK(stage(ctx, P->x, (size_t)P->nall * 3, on_device, st.S.x, &st.a.x));
  ISPH_CHECK(stage(ctx, P->type, (size_t)P->nall, on_device, st.S.type, &st.a.type));
  ISPH_CHECK(stage(ctx, P->vfrac, (size_t)P->nall, on_device, st.S.vfrac, &st.a.vfrac));
  ISPH_CHECK(stage(ctx, P->Gc, (size_t)P->nlocal * dim * dim, on_device, st.S.Gc, &st.a.Gc));
  NeighPtr np;
  ISPH_CHECK(stage_neigh_ptr(ctx, P, n, on_device, st.S.nptr, st.S.nptr64, np, &nnb));
  st.a.nptr = np.p32;
  if (!on_device) {
    for (long long k = 0; k < nnb; ++k)
      ISPH_REQUIRE(P->neigh_idx[k] >= 0 && P->neigh_idx[k] < P->nall, "neighbour index out of range");
  }
  ISPH_CHECK(stage(ctx, P->neigh_idx, (size_t)nnb, on_device, st.S.nidx, &di));
  ISPH_CHECK(build_neigh_ell(ctx, n, np, di, st.E, st.T));
  st.a.nlocal = n;
  st.a.antisym = antisym;
  return ISPH_SUCCESS;
}

// mode 0: gradient of a scalar [nall] -> [nlocal][3]; mode 1: divergence of a vector [nall][3] -> [nlocal]

// family / dimension dispatch of the two row operators: 3-D with the family known at compile time (the correction-tensor
// loops fold away or unroll), everything else through the run-time variant
template <class ARGS>
inline void launch_gradient(isph_ctx *ctx, const AsmTables &T, const ARGS &a, int grid, const double *f, double *out) {
  if (T.dim == 3 && a.antisym) hipLaunchKernelGGL((k_gradient<3, 1>), dim3(xcd_grid(grid)), dim3(kBlock), 0, ctx->stream, T, a, f, out);
  else if (T.dim == 3) hipLaunchKernelGGL((k_gradient<3, 0>), dim3(xcd_grid(grid)), dim3(kBlock), 0, ctx->stream, T, a, f, out);
  else hipLaunchKernelGGL((k_gradient<0, -1>), dim3(xcd_grid(grid)), dim3(kBlock), 0, ctx->stream, T, a, f, out);
}
template <class ARGS>
inline void launch_divergence(isph_ctx *ctx, const AsmTables &T, const ARGS &a, int grid, const double *f, double *out) {
  if (T.dim == 3 && a.antisym) hipLaunchKernelGGL((k_divergence<3, 1>), dim3(xcd_grid(grid)), dim3(kBlock), 0, ctx->stream, T, a, f, out);
  else if (T.dim == 3) hipLaunchKernelGGL((k_divergence<3, 0>), dim3(xcd_grid(grid)), dim3(kBlock), 0, ctx->stream, T, a, f, out);
  else hipLaunchKernelGGL((k_divergence<0, -1>), dim3(xcd_grid(grid)), dim3(kBlock), 0, ctx->stream, T, a, f, out);
}

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
    if (mode == 0) launch_gradient(ctx, st.T, st.a, grid, df, dout);
    else launch_divergence(ctx, st.T, st.a, grid, df, dout);
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
  DevTmp<double> buf;
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
  DevTmp<double> grad, srho, sdp;
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
    launch_gradient(ctx, st.T, st.a, grid, ddp, grad.p);
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
  DevTmp<double> grad, sp, sv, svn, sout;
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
    launch_gradient(ctx, st.T, st.a, grid, dpp, grad.p);
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
  DevTmp<double> sdp, svn;
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

// FunctorOuterComputeShift (ref: functor_compute_shift.h:48-113): dr[nlocal][3]
inline int compute_shift(isph_ctx *ctx, const isph_particles *P, double alpha, double shiftcut, double nonfluidweight,
                         double *dr, int on_device) {
  OpStage st;
  int rc = op_stage(ctx, P, 1, on_device, st);
  const int n = P->nlocal;
  double *dout = dr;
  if (rc == ISPH_SUCCESS && !on_device) { rc = st.out.reserve((size_t)(n > 0 ? n : 1) * 3); dout = st.out.p; }
  if (rc == ISPH_SUCCESS && n > 0) {
    hipLaunchKernelGGL(k_compute_shift, dim3(xcd_grid((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, st.T, st.a, alpha,
                       (const double *)nullptr, shiftcut * shiftcut, nonfluidweight, dout);
    if (!on_device && hipMemcpyAsync(dr, dout, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
      rc = fail("copy failed", __FILE__, __LINE__);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess)
      rc = fail("shift kernel failed", __FILE__, __LINE__);
  }
  st.release();
  return rc;
}

// shift step on a staged particle set: optional compute (dr == NULL: shift distance from shift*dt*vmax, the
// whole PairISPH_Corrected::shiftParticles, pair_isph_corrected.cpp:1203-1262) then ApplyShift.
// x, v [nall][3], p [nall] are updated on their nlocal rows.
inline int shift_apply(isph_ctx *ctx, const isph_particles *P, int antisym, const int *fixed, const double *dr_in,
                       double shift, double shiftcut, double nonfluidweight, double dt, double *x, double *v, double *p,
                       double *vmax_out, int on_device) {
  OpStage st;
  int rc = op_stage(ctx, P, antisym, on_device, st);
  const int n = P->nlocal, n1 = n > 0 ? n : 1;
  DevTmp<double> sdr, xn, vn, pn, scal;
  DevTmp<int> sfix;
  InOut ix, iv, ip;
  const double *ddr = nullptr;
  const int *dfix = nullptr;
  if (rc == ISPH_SUCCESS) rc = stage(ctx, fixed, (size_t)P->ntypes + 1, 0, sfix, &dfix);
  if (rc == ISPH_SUCCESS) rc = xn.reserve((size_t)n1 * 3);
  if (rc == ISPH_SUCCESS) rc = vn.reserve((size_t)n1 * 3);
  if (rc == ISPH_SUCCESS) rc = pn.reserve((size_t)n1);
  if (rc == ISPH_SUCCESS) rc = scal.reserve(2);
  if (rc == ISPH_SUCCESS) rc = ix.open(ctx, x, (size_t)P->nall * 3, on_device);
  if (rc == ISPH_SUCCESS) rc = iv.open(ctx, v, (size_t)P->nall * 3, on_device);
  if (rc == ISPH_SUCCESS) rc = ip.open(ctx, p, (size_t)P->nall, on_device);
  double vmax = 0.0;
  if (rc == ISPH_SUCCESS && dr_in) rc = stage(ctx, dr_in, (size_t)n * 3, on_device, sdr, &ddr);
  const int grid = (n1 + kBlock - 1) / kBlock;
  if (rc == ISPH_SUCCESS && !dr_in) {
    // vshift = max fluid speed over all ranks (MPI_Allreduce MAX in the reference)
    rc = sdr.reserve((size_t)n1 * 3);
    if (rc == ISPH_SUCCESS && hipMemsetAsync(scal.p, 0, 2 * sizeof(double), ctx->stream) != hipSuccess)
      rc = fail("memset failed", __FILE__, __LINE__);
    if (rc == ISPH_SUCCESS) {
      if (n > 0)
        hipLaunchKernelGGL(k_max_fluid_speed, dim3(grid), dim3(kBlock), 0, ctx->stream, n, P->dim, st.a.type, st.T.kind,
                           (const double *)iv.dev, reinterpret_cast<unsigned long long *>(scal.p));
      rc = comm_allreduce(ctx, scal.p, 1, /*max*/ 1, ctx->stream);
      if (rc == ISPH_SUCCESS && n > 0)
        hipLaunchKernelGGL(k_compute_shift, dim3(xcd_grid(grid)), dim3(kBlock), 0, ctx->stream, st.T, st.a, shift * dt,
                           (const double *)scal.p, shiftcut * shiftcut, nonfluidweight, sdr.p);
      ddr = sdr.p;
      if (vmax_out && hipMemcpyAsync(&vmax, scal.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        rc = fail("copy failed", __FILE__, __LINE__);
    }
  }
  if (rc == ISPH_SUCCESS && n > 0) {
    hipLaunchKernelGGL(k_apply_shift, dim3(xcd_grid(grid)), dim3(kBlock), 0, ctx->stream, st.T, st.a, dfix, ddr,
                       (const double *)ix.dev, (const double *)iv.dev, (const double *)ip.dev, xn.p, vn.p, pn.p);
    hipLaunchKernelGGL(k_shift_commit, dim3(grid), dim3(kBlock), 0, ctx->stream, n, (const double *)xn.p,
                       (const double *)vn.p, (const double *)pn.p, ix.dev, iv.dev, ip.dev);
    if (ix.close(ctx) != ISPH_SUCCESS || iv.close(ctx) != ISPH_SUCCESS || ip.close(ctx) != ISPH_SUCCESS) rc = ISPH_FAILURE;
  }
  if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess)
    rc = rc == ISPH_SUCCESS ? fail("shift kernels failed", __FILE__, __LINE__) : rc;
  if (vmax_out) *vmax_out = vmax;
  sdr.release(); xn.release(); vn.release(); pn.release(); scal.release(); sfix.release();
  ix.buf.release(); iv.buf.release(); ip.buf.release();
  st.release();
  return rc;
}

}  // namespace isph
