// solver.hpp -- host-side Krylov drivers issuing HIP kernels on one stream.
//
// Restates SolverLin_Belos::solveProblem (ref: solver_lin_belos.h:130-222):
// Belos BlockGmresSolMgr (flexible, block size 1, DGKS) and BlockCGSolMgr,
// right preconditioning, PoissonProjection for the singular pressure system
// (ref: solver_lin.h:131-140).  Hessenberg / Givens live on the host (m<=50);
// every vector operation is a kernel from krylov.hpp / sell.hpp.
#pragma once
#include <cmath>

#include "comm.hpp"
#include "core.hpp"
#include "krylov.hpp"

namespace isph {

int prec_apply_dev(isph_ctx *ctx, const isph_prec *M, const double *r, double *z);  // isph_capi.hip
int prec_apply_multi_dev(isph_ctx *ctx, const isph_prec *M, int K, const double *const *rs, double *const *zs);  // isph_capi.hip

inline int allreduce_inplace(isph_ctx *ctx, double *d, int count) {
  return comm_allreduce(ctx, d, count, /*sum*/ 0, ctx->stream);
}

// device scalar mailbox layout
enum { SC_DOT = 0 /* SC_DOT..SC_DOT+63: multi-dot results */, SC_Y = 64 /* 64 ys */, SC_MISC = 128, SC_COUNT = 160 };

constexpr int kMaxLockstep = 4;  // right-hand sides a lockstep solve advances together (one mailbox each)

inline int ensure_scalars(isph_ctx *ctx) {
  ISPH_CHECK(ctx->dscal.reserve(SC_COUNT * kMaxLockstep));
  ISPH_CHECK(ctx->partial.reserve((size_t)kMaxRedBlocks * 66));
  if (ctx->hscal_cap < (size_t)SC_COUNT * kMaxLockstep) {
    if (ctx->hscal) (void)hipHostFree(ctx->hscal);
    ISPH_CHECK_HIP(hipHostMalloc((void **)&ctx->hscal, (size_t)SC_COUNT * kMaxLockstep * sizeof(double)));
    ctx->hscal_cap = (size_t)SC_COUNT * kMaxLockstep;
  }
  return ISPH_SUCCESS;
}

// read `count` device scalars starting at slot `first` into hscal[first..]
inline int fetch_scalars(isph_ctx *ctx, int first, int count) {
  ISPH_CHECK_HIP(hipMemcpyAsync(ctx->hscal + first, ctx->dscal.p + first, sizeof(double) * (size_t)count,
                                hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return ISPH_SUCCESS;
}

// ---- halo exchange + SpMV -------------------------------------------------
__global__ void k_gather(int n, const int *__restrict__ idx, const double *__restrict__ x, double *__restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = x[idx[i]];
}

// Starts the halo exchange of x for a matrix with ghost columns: the boundary values are packed on the compute stream,
// the point-to-point exchange (comm.hpp: grouped ncclSend/ncclRecv, or the host-staged transport) runs on
// ctx->comm_stream and lands in ctx->xghost; ctx->ev_halo marks its end.  The
// caller launches the interior slices on the compute stream meanwhile and makes the compute stream wait on ev_halo
// before the boundary slices.  The communicator is only ever used by one operation at a time: the exchange starts
// after everything queued before it (ev_pack) and every later collective is queued behind the boundary kernel, which
// itself waits for ev_halo.  (Ifpack/Epetra do this Import inside Epetra_CrsMatrix::Apply, solver_lin.h:133.)
// hev (profile mode, the caller's operator only): three timing events of this product -- [0] values packed (compute
// stream), [1] ghost values landed (halo stream), [2] interior slices done (compute stream, recorded by spmv_dev)
inline int halo_begin(isph_ctx *ctx, const isph_mat *A, const double *x, hipEvent_t *hev = nullptr) {
  ISPH_REQUIRE(comm_active(ctx) && ctx->comm_stream,
               "a matrix with a halo needs a context made by isph_ctx_create_dist or isph_ctx_create_hostcomm");
  const Sell &S = A->S;
  const isph_halo &H = A->halo;
  ISPH_REQUIRE(H.nrecv == S.ncol - S.nrow, "matrix has ghost columns but no matching halo plan");
  ISPH_REQUIRE(H.n_int + H.n_bnd == S.nslices, "halo plan without the interior/boundary slice lists");
  ISPH_CHECK(ctx->xghost.reserve((size_t)(H.nrecv > 0 ? H.nrecv : 1)));
  ISPH_CHECK(ctx->sendbuf.reserve((size_t)(H.nsend > 0 ? H.nsend : 1)));
  if (H.nsend > 0)
    hipLaunchKernelGGL(k_gather, dim3(stream_grid(H.nsend)), dim3(kBlock), 0, ctx->stream, H.nsend, H.send_idx.p, x,
                       ctx->sendbuf.p);
  ISPH_CHECK_HIP(hipEventRecord(ctx->ev_pack, ctx->stream));
  if (hev) ISPH_CHECK_HIP(hipEventRecord(hev[0], ctx->stream));
  ISPH_CHECK_HIP(hipStreamWaitEvent(ctx->comm_stream, ctx->ev_pack, 0));
  ISPH_CHECK(comm_exchange(ctx, H, ctx->sendbuf.p, ctx->xghost.p, 1, false, ctx->comm_stream));
  ISPH_CHECK_HIP(hipEventRecord(ctx->ev_halo, ctx->comm_stream));
  if (hev) ISPH_CHECK_HIP(hipEventRecord(hev[1], ctx->comm_stream));
  return ISPH_SUCCESS;
}

// builds the 16-bit column copy of a matrix on first use (see k_sell_compress_cols); returns whether it is usable
inline bool sell_cols16(isph_ctx *ctx, const Sell &S) {
  if (S.c16_state != 0) return S.c16_state > 0;
  S.c16_state = -1;
  if (S.nslices == 0 || S.stored == 0) return false;
  DevTmp<int> flag;
  if (flag.reserve(1) != ISPH_SUCCESS || S.col16.reserve((size_t)S.stored) != ISPH_SUCCESS ||
      S.wtab.reserve((size_t)S.nslices * 64) != ISPH_SUCCESS) { flag.release(); return false; }
  int h = 0;
  if (hipMemsetAsync(flag.p, 0, sizeof(int), ctx->stream) == hipSuccess) {
    hipLaunchKernelGGL(k_sell_compress_cols, dim3((S.nslices + 3) / 4), dim3(kBlock), 0, ctx->stream, 0, S.nslices,
                       (const long long *)S.slice_off.p, (const int *)S.col.p, S.col16.p, S.wtab.p, flag.p);
    if (hipMemcpyAsync(&h, flag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
        hipStreamSynchronize(ctx->stream) == hipSuccess && hipGetLastError() == hipSuccess && h == 0)
      S.c16_state = 1;
  }
  flag.release();
  if (S.c16_state < 0) { S.col16.release(); S.wtab.release(); }
  return S.c16_state > 0;
}

template <bool DOT, bool LIST, bool GHOST>
inline void spmv_launch(isph_ctx *ctx, const Sell &S, bool c16, int nsl, const int *list, const double *x, const double *xg,
                        double *y, const double *nvec, const double *badd = nullptr, double alpha = 1.0) {
  if (nsl <= 0) return;
  int nbp = 0;
  const int grid = spmv_grid(nsl, &nbp);
  double *part = DOT ? ctx->partial.p : nullptr;
  if (c16)
    hipLaunchKernelGGL((k_sell_spmv16<8, DOT, LIST, GHOST>), dim3(grid), dim3(kBlock), 0, ctx->stream, S.nrow, nsl, nbp,
                       S.slice_off.p, S.col16.p, S.wtab.p, S.val.p, x, y, nvec, part, list, xg, badd, alpha);
  else
    hipLaunchKernelGGL((k_sell_spmv<8, DOT, true, LIST, GHOST>), dim3(grid), dim3(kBlock), 0, ctx->stream, S.nrow, nsl, nbp,
                       S.slice_off.p, S.col.p, S.val.p, x, y, nvec, part, list, xg, badd, alpha);
}

// y = A x ; if nvec: also SC_MISC+0 = y.nvec (all-reduced).  badd / alpha (without nvec): y = badd + alpha (A x), badd may be y
inline int spmv_dev(isph_ctx *ctx, const isph_mat *A, const double *x, double *y, const double *nvec, const double *badd = nullptr,
                    double alpha = 1.0) {
  const Sell &S = A->S;
  // (a rank whose own rows reference no ghost column still sends what its neighbours' rows reference)
  const bool halo = !A->local && (S.ncol != S.nrow || A->halo.nsend > 0);
  // AMG transfer / coarse operators are small or have very long rows: the window tables do not pay there
  const bool c16 = !A->local && !A->aux && sell_cols16(ctx, S);
  ISPH_REQUIRE(nvec == nullptr || (badd == nullptr && alpha == 1.0), "spmv: the dot epilogue takes the plain product");
  ProfScope prof((A->local || A->aux) ? nullptr : ctx, PROF_SPMV);  // the SpMV statistics are those of the caller's operator only
  if (nvec)
    ISPH_CHECK(ctx->partial.reserve((size_t)S.nslices > (size_t)kMaxRedBlocks * 66 ? (size_t)S.nslices : (size_t)kMaxRedBlocks * 66));
  if (!halo) {
    if (nvec) spmv_launch<true, false, false>(ctx, S, c16, S.nslices, nullptr, x, nullptr, y, nvec);
    else spmv_launch<false, false, false>(ctx, S, c16, S.nslices, nullptr, x, nullptr, y, nvec, badd, alpha);
  } else {
    const isph_halo &H = A->halo;
    hipEvent_t *hev = nullptr;
    if (ctx->profile && !A->aux && !A->local) {  // exchange vs interior time of this product (isph_ctx_halo_profile_read)
      if (ctx->hev_used + 3 > ctx->hev.size())
        for (int k = 0; k < 3; ++k) { hipEvent_t e = nullptr; ISPH_CHECK_HIP(hipEventCreate(&e)); ctx->hev.push_back(e); }
      hev = ctx->hev.data() + ctx->hev_used;
      ctx->hev_used += 3;
    }
    ISPH_CHECK(halo_begin(ctx, A, x, hev));
    // interior slices (no ghost column) while the exchange is in flight
    if (nvec) spmv_launch<true, true, false>(ctx, S, c16, H.n_int, H.list_int.p, x, nullptr, y, nvec);
    else spmv_launch<false, true, false>(ctx, S, c16, H.n_int, H.list_int.p, x, nullptr, y, nvec, badd, alpha);
    if (hev) ISPH_CHECK_HIP(hipEventRecord(hev[2], ctx->stream));
    ISPH_CHECK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_halo, 0));
    if (nvec) spmv_launch<true, true, true>(ctx, S, c16, H.n_bnd, H.list_bnd.p, x, ctx->xghost.p, y, nvec);
    else spmv_launch<false, true, true>(ctx, S, c16, H.n_bnd, H.list_bnd.p, x, ctx->xghost.p, y, nvec, badd, alpha);
  }
  prof.end();
  if (nvec) {
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(kBlock), 0, ctx->stream, 1, S.nslices, ctx->partial.p,
                       ctx->dscal.p + SC_MISC);
    ISPH_CHECK(allreduce_inplace(ctx, ctx->dscal.p + SC_MISC, 1));
  }
  ISPH_CHECK_HIP(hipGetLastError());
  return ISPH_SUCCESS;
}

// y_k = A x_k, k < K: one sweep of the matrix for all vectors when the matrix has no ghost columns and its 16-bit
// columns exist (k_sell_spmm16), otherwise one product after the other.  Same bits either way.
inline int spmm_dev(isph_ctx *ctx, const isph_mat *A, int K, const double *const *xs, double *const *ys) {
  const Sell &S = A->S;
  const bool halo = !A->local && (S.ncol != S.nrow || A->halo.nsend > 0);
  if (K < 2 || K > 4 || halo || A->local || !sell_cols16(ctx, S)) {
    for (int k = 0; k < K; ++k) ISPH_CHECK(spmv_dev(ctx, A, xs[k], ys[k], nullptr));
    return ISPH_SUCCESS;
  }
  ProfScope prof(ctx, PROF_SPMV);
  SpmmVecs V;
  for (int k = 0; k < 4; ++k) { V.x[k] = xs[k < K ? k : 0]; V.y[k] = ys[k < K ? k : 0]; }
  int nbp = 0;
  const int grid = spmv_grid(S.nslices, &nbp);
  if (S.nslices > 0) {
    if (K == 2)
      hipLaunchKernelGGL((k_sell_spmm16<2>), dim3(grid), dim3(kBlock), 0, ctx->stream, S.nrow, S.nslices, nbp, S.slice_off.p,
                         S.col16.p, S.wtab.p, S.val.p, V);
    else if (K == 3)
      hipLaunchKernelGGL((k_sell_spmm16<3>), dim3(grid), dim3(kBlock), 0, ctx->stream, S.nrow, S.nslices, nbp, S.slice_off.p,
                         S.col16.p, S.wtab.p, S.val.p, V);
    else
      hipLaunchKernelGGL((k_sell_spmm16<4>), dim3(grid), dim3(kBlock), 0, ctx->stream, S.nrow, S.nslices, nbp, S.slice_off.p,
                         S.col16.p, S.wtab.p, S.val.p, V);
  }
  prof.end();
  ISPH_CHECK_HIP(hipGetLastError());
  return ISPH_SUCCESS;
}

struct LinOp {
  isph_ctx *ctx;
  const isph_mat *A;
  const isph_prec *M;
  const double *nvec;  // device, NULL unless singular
  int n;               // length of the Krylov vectors (dim * nloc for a blocked operator)
  // blocked operator (SolverLin::setBlock / Thyra::DefaultBlockedLinearOp, ref: solver_lin.cpp:127-138):
  // dim x dim sparse blocks over product vectors [x_0; ..; x_{dim-1}], NULL blocks are zero; the
  // preconditioner is block-diagonal with the same operator on every component (precond_ml.h:138-155)
  int dim = 1, nloc = 0;
  const isph_mat *const *blk = nullptr;
  double *tmp = nullptr;  // [nloc] scratch of the blocked product
  // PoissonProjection::Apply: y = A x; y -= (y.n) n
  int apply(const double *x, double *y) const {
    if (blk) {
      for (int i = 0; i < dim; ++i) {
        double *yi = y + (size_t)i * nloc;
        bool first = true;
        for (int j = 0; j < dim; ++j) {
          const isph_mat *B = blk[i * dim + j];
          if (!B) continue;
          ISPH_CHECK(spmv_dev(ctx, B, x + (size_t)j * nloc, first ? yi : tmp, nullptr));
          if (!first)
            hipLaunchKernelGGL(k_axpy_dev, dim3(stream_grid(nloc)), dim3(kBlock), 0, ctx->stream, nloc, 1.0,
                               (const double *)nullptr, (const double *)tmp, yi);
          first = false;
        }
        if (first) ISPH_CHECK_HIP(hipMemsetAsync(yi, 0, sizeof(double) * (size_t)nloc, ctx->stream));
      }
      return ISPH_SUCCESS;
    }
    ISPH_CHECK(spmv_dev(ctx, A, x, y, nvec));
    if (nvec)
      hipLaunchKernelGGL(k_axpy_dev, dim3(stream_grid(n)), dim3(kBlock), 0, ctx->stream, n, -1.0,
                         ctx->dscal.p + SC_MISC, nvec, y);
    return ISPH_SUCCESS;
  }
  int prec(const double *r, double *z) const {
    if (!M) {  // no preconditioner object: identity
      ISPH_CHECK_HIP(hipMemcpyAsync(z, r, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
      return ISPH_SUCCESS;
    }
    if (blk) {  // block-diagonal with the same operator on every component: one sweep for all of them where that exists
      const double *rs[4];
      double *zs[4];
      for (int i = 0; i < dim && i < 4; ++i) { rs[i] = r + (size_t)i * nloc; zs[i] = z + (size_t)i * nloc; }
      return prec_apply_multi_dev(ctx, M, dim, rs, zs);
    }
    return prec_apply_dev(ctx, M, r, z);
  }
};

// dot(a,b) -> dscal[slot] (all-reduced); optionally dot(c,d) -> dscal[slot+1]
inline int dot_dev(isph_ctx *ctx, int n, const double *a, const double *b, const double *c, const double *d, int slot) {
  const int g = stream_grid(n);
  hipLaunchKernelGGL(k_dot2, dim3(g), dim3(kBlock), 0, ctx->stream, n, a, b, c, d, ctx->partial.p);
  hipLaunchKernelGGL(k_reduce_partials, dim3(2), dim3(kBlock), 0, ctx->stream, 2, g, ctx->partial.p, ctx->dscal.p + slot);
  return allreduce_inplace(ctx, ctx->dscal.p + slot, 2);
}

// project v -= (v.n) n
inline int project_dev(isph_ctx *ctx, int n, const double *nvec, double *v) {
  ISPH_CHECK(dot_dev(ctx, n, v, nvec, nullptr, nullptr, SC_MISC + 2));
  hipLaunchKernelGGL(k_axpy_dev, dim3(stream_grid(n)), dim3(kBlock), 0, ctx->stream, n, -1.0, ctx->dscal.p + SC_MISC + 2,
                     nvec, v);
  return ISPH_SUCCESS;
}

// one classical Gram-Schmidt pass of w against V[0..nk): c = V^T w (one fused
// multi-dot + one all-reduce), w -= V c, and the norms before/after.
// Host gets c[0..nk), ww_old, ww_new in hscal[SC_DOT ..].
inline int cgs_pass(isph_ctx *ctx, int n, int nk, const double *V, long long ld, double *w) {
  int g = stream_grid(n);
  if (g > 1024) g = 1024;  // 4 workgroups per CU: more did not raise the achieved bandwidth (measured)
  ISPH_CHECK(ctx->partial.reserve((size_t)(nk + 2) * kMaxRedBlocks > (size_t)kMaxRedBlocks * 66 ? (size_t)(nk + 2) * kMaxRedBlocks : (size_t)kMaxRedBlocks * 66));
  hipLaunchKernelGGL((k_multi_dot<1>), dim3(g), dim3(kBlock), 0, ctx->stream, n, nk, V, ld, w, ctx->partial.p);
  hipLaunchKernelGGL(k_reduce_partials, dim3(nk + 1), dim3(kBlock), 0, ctx->stream, nk + 1, g, ctx->partial.p,
                     ctx->dscal.p + SC_DOT);
  ISPH_CHECK(allreduce_inplace(ctx, ctx->dscal.p + SC_DOT, nk + 1));
  const int g2 = stream_grid(n);
  hipLaunchKernelGGL(k_multi_axpy_norm, dim3(g2), dim3(kBlock), 0, ctx->stream, n, nk, V, ld, ctx->dscal.p + SC_DOT, w,
                     ctx->partial.p);
  hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(kBlock), 0, ctx->stream, 1, g2, ctx->partial.p,
                     ctx->dscal.p + SC_DOT + nk + 1);
  ISPH_CHECK(allreduce_inplace(ctx, ctx->dscal.p + SC_DOT + nk + 1, 1));
  ISPH_CHECK_HIP(hipGetLastError());
  return fetch_scalars(ctx, SC_DOT, nk + 2);
}

// DGKS / ICGS step without any host round trip.  Pass 1: c = V^T w; then ONE kernel applies w -= V c and
// already accumulates the second pass's projection c2 = V^T w_new and |w_new|^2 while the rows of V are in
// registers; the DGKS test (dep_tol = 1/sqrt(2); ICGS: always) is taken on the device, inside the second update's kernel, and that update
// w -= V c2 (+ its norm) skips itself when it is not due.  Same arithmetic as two separate passes, one read of
// V less.  When vnext != NULL the next basis vector w/|w| is formed on the device as well, so the caller can
// queue the next preconditioner/operator application before it looks at the scalars.
// Mailbox: c at SC_DOT.., |w|^2 at SC_DOT+nk; c2 at SC_Y.., |w_new|^2 at SC_Y+nk; flag, |w_final|^2 at SC_ORTHO..
// (|w_final|^2 = |w_new|^2 - |c2|^2 when the second pass runs: see k_multi_axpy_norm)
enum { SC_ORTHO = SC_MISC + 20 };
inline int ortho_enqueue(isph_ctx *ctx, int n, int nk, const double *V, long long ld, double *w, int ortho, double *vnext,
                         bool deflate = false, int mb = 0) {
  int g = stream_grid(n);
  // 2 workgroups per CU with two rows per thread in flight: multi-dot 53.6 -> 44.4 us, the 64-wide fused update
  // 92.8 -> 79.5 us against 4 workgroups per CU, one row (rocprofv3, bench matrix)
  if (g > 1024) g = 1024;
  ISPH_CHECK(ctx->partial.reserve((size_t)kMaxRedBlocks * 66));
  hipStream_t st = ctx->stream;
  double *dh1 = ctx->dscal.p + mb + SC_DOT, *dh2 = ctx->dscal.p + mb + SC_Y, *dor = ctx->dscal.p + mb + SC_ORTHO;  // mb: mailbox of this right-hand side
  constexpr int dot_grid = 512;   // (256 ... 1024 workgroups: the same 36 us for the multi-dot, the fused update slower above 512)
  if (g > dot_grid) g = dot_grid;
  {
    ProfScope prof(ctx, PROF_MULTI_DOT);
    if ((reinterpret_cast<uintptr_t>(w) & 15) == 0 && (reinterpret_cast<uintptr_t>(V) & 15) == 0)
      hipLaunchKernelGGL(k_multi_dot_v2, dim3(g), dim3(kBlock), 0, st, n, nk, V, ld, w, ctx->partial.p);
    else
      hipLaunchKernelGGL((k_multi_dot<2>), dim3(g), dim3(kBlock), 0, st, n, nk, V, ld, w, ctx->partial.p);
  }
  hipLaunchKernelGGL(k_reduce_partials, dim3(nk + 1), dim3(kBlock), 0, st, nk + 1, g, ctx->partial.p, dh1,
                     (const double *)nullptr);
  ISPH_CHECK(allreduce_inplace(ctx, dh1, nk + 1));
  {
    ProfScope prof(ctx, PROF_MULTI_AXPY_DOT);
    if (nk <= 16)
      hipLaunchKernelGGL((k_multi_axpy_dot<16>), dim3(g), dim3(kBlock), 0, st, n, nk, V, ld, dh1, w, ctx->partial.p);
    else if (nk <= 32)
      hipLaunchKernelGGL((k_multi_axpy_dot<32>), dim3(g), dim3(kBlock), 0, st, n, nk, V, ld, dh1, w, ctx->partial.p);
    else if (nk <= 48)  // between the two: 96 + 96 registers keep two waves per SIMD where the 64-wide instance has one
      hipLaunchKernelGGL((k_multi_axpy_dot<48>), dim3(g), dim3(kBlock), 0, st, n, nk, V, ld, dh1, w, ctx->partial.p);
    else
      hipLaunchKernelGGL((k_multi_axpy_dot<64>), dim3(g), dim3(kBlock), 0, st, n, nk, V, ld, dh1, w, ctx->partial.p);
  }
  hipLaunchKernelGGL(k_reduce_partials, dim3(nk + 1), dim3(kBlock), 0, st, nk + 1, g, ctx->partial.p, dh2,
                     (const double *)nullptr);
  ISPH_CHECK(allreduce_inplace(ctx, dh2, nk + 1));
  const int g2 = stream_grid(n);
  // the DGKS decision (taken by the kernel itself from the reduced scalars), the second update and the normalised next
  // basis vector (vnext = w / |w|) leave in one sweep
  {
    ProfScope prof(ctx, PROF_MULTI_AXPY_NORM);
    hipLaunchKernelGGL(k_multi_axpy_norm, dim3(g2), dim3(kBlock), 0, st, n, nk, V, ld, (const double *)dh2, w, ctx->partial.p,
                       (const double *)(dh1 + nk), deflate ? (const double *)dh1 : (const double *)nullptr, ortho == 1 ? 1 : 0,
                       dor, vnext);
  }
  ISPH_CHECK_HIP(hipGetLastError());
  return ISPH_SUCCESS;
}

// host side of ortho_enqueue, once the mailbox is in hscal
inline void ortho_collect(isph_ctx *ctx, int nk, double *h, double *wnorm, int mb = 0) {
  const double *hs = ctx->hscal + mb;
  const bool second = hs[SC_ORTHO] != 0.0;
  if (second) ++ctx->stat_reorth;
  for (int k = 0; k < nk; ++k) h[k] = hs[SC_DOT + k] + (second ? hs[SC_Y + k] : 0.0);
  *wnorm = std::sqrt(hs[SC_ORTHO + 1]);
}

// Belos DGKS / ICGS / IMGS for block size 1. h[0..j] coefficients, returns ||w||.
inline int orthogonalize(isph_ctx *ctx, int n, int j, const double *V, long long ld, double *w, double *h, int ortho,
                         double *wnorm) {
  const int nk = j + 1;
  for (int k = 0; k < nk; ++k) h[k] = 0.0;
  if (ortho == 2) {  // IMGS: two modified Gram-Schmidt sweeps
    double nn = 0.0;
    for (int pass = 0; pass < 2; ++pass)
      for (int k = 0; k < nk; ++k) {
        ISPH_CHECK(cgs_pass(ctx, n, 1, V + (long long)k * ld, ld, w));
        h[k] += ctx->hscal[SC_DOT];
        nn = ctx->hscal[SC_DOT + 2];
      }
    *wnorm = std::sqrt(nn);
    return ISPH_SUCCESS;
  }
  // DGKS / ICGS: device-only part, then wait for the scalars
  ISPH_CHECK(ortho_enqueue(ctx, n, nk, V, ld, w, ortho, nullptr));
  ISPH_CHECK(fetch_scalars(ctx, 0, SC_COUNT));
  ortho_collect(ctx, nk, h, wnorm);
  return ISPH_SUCCESS;
}

inline int gmres(const LinOp &op, const double *b, double *x, const isph_solver_params *prm, isph_solve_info *info) {
  isph_ctx *ctx = op.ctx;
  const int n = op.n, m = prm->num_blocks;
  ISPH_REQUIRE(m >= 1 && m <= 62, "Num Blocks must be in [1,62]");
  const long long ld = ((long long)n + 63) / 64 * 64;
  // Singular systems (null vector n): n is kept as column 0 of the basis array.  Every Krylov vector is orthogonal
  // to n, so V^T (w - (w.n) n) = V^T w: the projection of PoissonProjection::Apply is folded into the Gram-Schmidt
  // step (n rides along as one more vector of the multi-dot / update), which saves one reduction, one all-reduce
  // and one vector pass per iteration.  IMGS keeps the explicit projection.
  const bool deflate = op.nvec != nullptr && !op.blk && prm->ortho != 2;
  ISPH_CHECK(ctx->V.reserve((size_t)ld * (size_t)(m + 2)));
  if (prm->flexible) ISPH_CHECK(ctx->Z.reserve((size_t)ld * (size_t)m));
  ISPH_CHECK(ctx->wv.reserve((size_t)ld));
  ISPH_CHECK(ctx->tv.reserve((size_t)ld));
  double *Vall = ctx->V.p, *V = ctx->V.p + (deflate ? ld : 0), *Z = ctx->Z.p, *w = ctx->wv.p, *t = ctx->tv.p;
  if (deflate)
    ISPH_CHECK_HIP(hipMemcpyAsync(Vall, op.nvec, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
  std::vector<double> H((size_t)(m + 1) * (size_t)m, 0.0), cs((size_t)m), sn((size_t)m), g((size_t)m + 1), y((size_t)m);
  const int sg = stream_grid(n);
  hipStream_t st = ctx->stream;

  // r0 = b - Op(x)
  ISPH_CHECK(op.apply(x, w));
  hipLaunchKernelGGL(k_residual, dim3(sg), dim3(kBlock), 0, st, n, b, w);
  ISPH_CHECK(dot_dev(ctx, n, w, w, nullptr, nullptr, SC_MISC + 4));
  ISPH_CHECK(fetch_scalars(ctx, SC_MISC + 4, 1));
  double beta = std::sqrt(ctx->hscal[SC_MISC + 4]);
  const double scale = beta == 0.0 ? 1.0 : beta;  // Belos: zero scale -> 1
  info->iters = 0;
  info->restarts = 0;
  info->converged = 0;
  info->rel_res_implicit = beta / scale;
  if (beta / scale <= prm->tol) info->converged = 1;

  while (!info->converged && info->iters < prm->max_iters) {
    hipLaunchKernelGGL(k_scale_copy, dim3(sg), dim3(kBlock), 0, st, n, w, V, 1.0 / beta, (const double *)nullptr, 0);
    std::fill(g.begin(), g.end(), 0.0);
    g[0] = beta;
    int j = 0;
    // queue z_j = M^-1 v_j, w = Op z_j
    auto enqueue_op = [&](int col) -> int {
      const double *zj;
      double *vj = V + (long long)col * ld;
      if (prm->flexible) {
        ISPH_CHECK(op.prec(vj, Z + (long long)col * ld));
        zj = Z + (long long)col * ld;
      } else {
        ISPH_CHECK(op.prec(vj, t));
        zj = t;
      }
      // with deflation the raw product goes to the Gram-Schmidt step, which removes the n component itself
      return deflate ? spmv_dev(ctx, op.A, zj, w, nullptr) : op.apply(zj, w);
    };
    bool op_queued = false;  // column j's operator application is already in the stream
    while (j < m) {
      if (!op_queued) ISPH_CHECK(enqueue_op(j));
      op_queued = false;
      double *h = &H[(size_t)j * (size_t)(m + 1)];
      double wn = 0.0;
      if (prm->ortho == 2) {
        ISPH_CHECK(orthogonalize(ctx, n, j, V, ld, w, h, prm->ortho, &wn));
        if (wn != 0.0)
          hipLaunchKernelGGL(k_scale_copy, dim3(sg), dim3(kBlock), 0, st, n, w, V + (long long)(j + 1) * ld, 1.0 / wn,
                             (const double *)nullptr, 0);
      } else {
        // the whole Gram-Schmidt step and v_{j+1} = w/|w| stay on the device; the host only waits for the copy of
        // the scalar mailbox, and while it does the Givens update the GPU already works on the next column
        const int nkt = j + 1 + (deflate ? 1 : 0);  // basis vectors (+ n) the step projects against
        ISPH_CHECK(ortho_enqueue(ctx, n, nkt, Vall, ld, w, prm->ortho, V + (long long)(j + 1) * ld, deflate));
        ISPH_CHECK_HIP(hipMemcpyAsync(ctx->hscal, ctx->dscal.p, sizeof(double) * SC_COUNT, hipMemcpyDeviceToHost, st));
        ISPH_CHECK_HIP(hipEventRecord(ctx->ev_fetch, st));
        if (j + 1 < m && info->iters + 1 < prm->max_iters) {  // speculative: discarded if this column converges
          ISPH_CHECK(enqueue_op(j + 1));
          op_queued = true;
        }
        ISPH_CHECK_HIP(hipEventSynchronize(ctx->ev_fetch));
        if (deflate) {
          double hh[66];
          ortho_collect(ctx, nkt, hh, &wn);
          for (int k = 0; k <= j; ++k) h[k] = hh[k + 1];
        } else {
          ortho_collect(ctx, nkt, h, &wn);
        }
      }
      h[j + 1] = wn;
      for (int k = 0; k < j; ++k) {
        const double a = cs[k] * h[k] + sn[k] * h[k + 1];
        h[k + 1] = -sn[k] * h[k] + cs[k] * h[k + 1];
        h[k] = a;
      }
      {
        const double a = h[j], bb = h[j + 1], rr = std::hypot(a, bb);
        cs[j] = rr == 0.0 ? 1.0 : a / rr;
        sn[j] = rr == 0.0 ? 0.0 : bb / rr;
        h[j] = rr;
        h[j + 1] = 0.0;
        g[j + 1] = -sn[j] * g[j];
        g[j] = cs[j] * g[j];
      }
      ++j;
      ++info->iters;
      info->rel_res_implicit = std::fabs(g[j]) / scale;
      if (prm->verbose && ctx->rank == 0 && info->iters % 10 == 0)
        printf(">> isph::gmres iter %d  rel res %.3e\n", info->iters, info->rel_res_implicit);
      if (info->rel_res_implicit <= prm->tol) { info->converged = 1; break; }
      if (info->iters >= prm->max_iters) break;
    }
    for (int k = j - 1; k >= 0; --k) {
      double s = g[k];
      for (int l = k + 1; l < j; ++l) s -= H[(size_t)l * (size_t)(m + 1) + k] * y[l];
      y[k] = s / H[(size_t)k * (size_t)(m + 1) + k];
    }
    // x += Z y  (flexible)  or  x += M^-1 (V y)
    for (int k = 0; k < j; ++k) ctx->hscal[SC_Y + k] = y[k];
    ISPH_CHECK_HIP(hipMemcpyAsync(ctx->dscal.p + SC_Y, ctx->hscal + SC_Y, sizeof(double) * (size_t)j, hipMemcpyHostToDevice, st));
    if (prm->flexible) {
      hipLaunchKernelGGL(k_multi_axpy, dim3(sg), dim3(kBlock), 0, st, n, j, Z, ld, ctx->dscal.p + SC_Y, x);
    } else {
      hipLaunchKernelGGL(k_fill, dim3(sg), dim3(kBlock), 0, st, n, w, 0.0);
      hipLaunchKernelGGL(k_multi_axpy, dim3(sg), dim3(kBlock), 0, st, n, j, V, ld, ctx->dscal.p + SC_Y, w);
      ISPH_CHECK(op.prec(w, t));
      hipLaunchKernelGGL(k_axpy_dev, dim3(sg), dim3(kBlock), 0, st, n, 1.0, (const double *)nullptr, t, x);
    }
    // the H2D source (hscal) must not be rewritten before the copy has run
    ISPH_CHECK_HIP(hipStreamSynchronize(st));
    if (info->converged || info->iters >= prm->max_iters) break;
    if (info->restarts >= prm->max_restarts) break;
    ++info->restarts;
    ISPH_CHECK(op.apply(x, w));
    hipLaunchKernelGGL(k_residual, dim3(sg), dim3(kBlock), 0, st, n, b, w);
    ISPH_CHECK(dot_dev(ctx, n, w, w, nullptr, nullptr, SC_MISC + 4));
    ISPH_CHECK(fetch_scalars(ctx, SC_MISC + 4, 1));
    beta = std::sqrt(ctx->hscal[SC_MISC + 4]);
    if (beta == 0.0) { info->converged = 1; break; }
  }
  ISPH_CHECK_HIP(hipGetLastError());
  return ISPH_SUCCESS;
}

// K right-hand sides of one matrix (the Helmholtz system: one per velocity component, pair_isph.cpp:925-966) advanced
// together: Belos solves them one after the other (block size 1), and so does every Krylov space here -- own basis,
// own Hessenberg, own convergence test and restarts, the arithmetic of gmres() above vector by vector --, but the K
// operator applications of an iteration share one sweep of the matrix (spmm_dev), and while the host waits for one
// system's Gram-Schmidt scalars the device already works on the next system's.  Flexible GMRES, DGKS / ICGS,
// non-singular, un-blocked operators; everything else goes through gmres() per right-hand side.
inline bool gmres_lockstep_ok(const LinOp &op, const isph_solver_params *prm, int K) {
  return K >= 2 && K <= kMaxLockstep && !op.nvec && !op.blk && prm->flexible && prm->ortho != 2 && prm->solver_type == 0;
}

inline int gmres_lockstep(const LinOp &op, int K, const double *const *bs, double *const *xs, const isph_solver_params *prm,
                          isph_solve_info *infos) {
  isph_ctx *ctx = op.ctx;
  const int n = op.n, m = prm->num_blocks;
  ISPH_REQUIRE(m >= 1 && m <= 62, "Num Blocks must be in [1,62]");
  const long long ld = ((long long)n + 63) / 64 * 64;
  ISPH_CHECK(ctx->V.reserve((size_t)ld * (size_t)(m + 2) * (size_t)K));
  ISPH_CHECK(ctx->Z.reserve((size_t)ld * (size_t)m * (size_t)K));
  ISPH_CHECK(ctx->wv.reserve((size_t)ld * (size_t)K));
  while ((int)ctx->ev_ls.size() < K) {
    hipEvent_t e;
    ISPH_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->ev_ls.push_back(e);
  }
  struct Sys {
    double *V, *Z, *w;
    std::vector<double> H, cs, sn, g, y;
    double beta = 0.0, scale = 1.0;
    int j = 0;
    bool in_cycle = false, done = false;
  };
  std::vector<Sys> sys((size_t)K);
  const int sg = stream_grid(n);
  hipStream_t st = ctx->stream;
  const double *xin[kMaxLockstep];
  double *yout[kMaxLockstep];
  for (int k = 0; k < K; ++k) {
    Sys &s = sys[(size_t)k];
    s.V = ctx->V.p + (size_t)k * ld * (size_t)(m + 2);
    s.Z = ctx->Z.p + (size_t)k * ld * (size_t)m;
    s.w = ctx->wv.p + (size_t)k * ld;
    s.H.assign((size_t)(m + 1) * (size_t)m, 0.0);
    s.cs.assign((size_t)m, 0.0); s.sn.assign((size_t)m, 0.0); s.g.assign((size_t)m + 1, 0.0); s.y.assign((size_t)m, 0.0);
    infos[k].iters = 0; infos[k].restarts = 0; infos[k].converged = 0;
    xin[k] = xs[k]; yout[k] = s.w;
  }
  // r0 = b - A x for every system
  ISPH_CHECK(spmm_dev(ctx, op.A, K, xin, yout));
  for (int k = 0; k < K; ++k) {
    Sys &s = sys[(size_t)k];
    hipLaunchKernelGGL(k_residual, dim3(sg), dim3(kBlock), 0, st, n, bs[k], s.w);
    ISPH_CHECK(dot_dev(ctx, n, s.w, s.w, nullptr, nullptr, SC_MISC + 4));
    ISPH_CHECK(fetch_scalars(ctx, SC_MISC + 4, 1));
    s.beta = std::sqrt(ctx->hscal[SC_MISC + 4]);
    s.scale = s.beta == 0.0 ? 1.0 : s.beta;
    infos[k].rel_res_implicit = s.beta / s.scale;
    if (s.beta / s.scale <= prm->tol) { infos[k].converged = 1; s.done = true; }
  }
  auto start_cycle = [&](int k) {
    Sys &s = sys[(size_t)k];
    hipLaunchKernelGGL(k_scale_copy, dim3(sg), dim3(kBlock), 0, st, n, s.w, s.V, 1.0 / s.beta, (const double *)nullptr, 0);
    std::fill(s.g.begin(), s.g.end(), 0.0);
    s.g[0] = s.beta;
    s.j = 0;
    s.in_cycle = true;
  };
  // end of a cycle of system k: x += Z y, then either done or the residual of the restart
  auto end_cycle = [&](int k) -> int {
    Sys &s = sys[(size_t)k];
    isph_solve_info &inf = infos[k];
    const int j = s.j, mb = k * SC_COUNT;
    for (int q = j - 1; q >= 0; --q) {
      double acc = s.g[(size_t)q];
      for (int l = q + 1; l < j; ++l) acc -= s.H[(size_t)l * (size_t)(m + 1) + q] * s.y[(size_t)l];
      s.y[(size_t)q] = acc / s.H[(size_t)q * (size_t)(m + 1) + q];
    }
    for (int q = 0; q < j; ++q) ctx->hscal[mb + SC_Y + q] = s.y[(size_t)q];
    ISPH_CHECK_HIP(hipMemcpyAsync(ctx->dscal.p + mb + SC_Y, ctx->hscal + mb + SC_Y, sizeof(double) * (size_t)j, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_multi_axpy, dim3(sg), dim3(kBlock), 0, st, n, j, s.Z, ld, ctx->dscal.p + mb + SC_Y, xs[k]);
    ISPH_CHECK_HIP(hipStreamSynchronize(st));  // the H2D source must not be rewritten before the copy has run
    s.in_cycle = false;
    if (inf.converged || inf.iters >= prm->max_iters || inf.restarts >= prm->max_restarts) { s.done = true; return ISPH_SUCCESS; }
    ++inf.restarts;
    ISPH_CHECK(op.apply(xs[k], s.w));
    hipLaunchKernelGGL(k_residual, dim3(sg), dim3(kBlock), 0, st, n, bs[k], s.w);
    ISPH_CHECK(dot_dev(ctx, n, s.w, s.w, nullptr, nullptr, SC_MISC + 4));
    ISPH_CHECK(fetch_scalars(ctx, SC_MISC + 4, 1));
    s.beta = std::sqrt(ctx->hscal[SC_MISC + 4]);
    if (s.beta == 0.0) { inf.converged = 1; s.done = true; return ISPH_SUCCESS; }
    start_cycle(k);
    return ISPH_SUCCESS;
  };
  for (int k = 0; k < K; ++k)
    if (!sys[(size_t)k].done) {
      if (infos[k].iters < prm->max_iters) start_cycle(k);
      else sys[(size_t)k].done = true;
    }
  int act[kMaxLockstep];
  while (true) {
    int na = 0;
    for (int k = 0; k < K; ++k)
      if (!sys[(size_t)k].done && sys[(size_t)k].in_cycle) act[na++] = k;
    if (na == 0) break;
    // z = M^-1 v_j and w = A z for all active systems: one sweep of the factor stream, one of the matrix
    {
      const double *vin[kMaxLockstep];
      double *zout[kMaxLockstep];
      for (int a = 0; a < na; ++a) {
        Sys &s = sys[(size_t)act[a]];
        vin[a] = s.V + (long long)s.j * ld;
        zout[a] = s.Z + (long long)s.j * ld;
        xin[a] = zout[a];
        yout[a] = s.w;
      }
      if (op.M) ISPH_CHECK(prec_apply_multi_dev(ctx, op.M, na, vin, zout));
      else
        for (int a = 0; a < na; ++a) ISPH_CHECK(op.prec(vin[a], zout[a]));
    }
    ISPH_CHECK(spmm_dev(ctx, op.A, na, xin, yout));
    // Gram-Schmidt of every system on the device, its scalars on their way to the system's own mailbox
    for (int a = 0; a < na; ++a) {
      const int k = act[a], mb = k * SC_COUNT;
      Sys &s = sys[(size_t)k];
      ISPH_CHECK(ortho_enqueue(ctx, n, s.j + 1, s.V, ld, s.w, prm->ortho, s.V + (long long)(s.j + 1) * ld, false, mb));
      ISPH_CHECK_HIP(hipMemcpyAsync(ctx->hscal + mb, ctx->dscal.p + mb, sizeof(double) * SC_COUNT, hipMemcpyDeviceToHost, st));
      ISPH_CHECK_HIP(hipEventRecord(ctx->ev_ls[(size_t)k], st));
    }
    for (int a = 0; a < na; ++a) {
      const int k = act[a], mb = k * SC_COUNT;
      Sys &s = sys[(size_t)k];
      isph_solve_info &inf = infos[k];
      ISPH_CHECK_HIP(hipEventSynchronize(ctx->ev_ls[(size_t)k]));
      const int j = s.j;
      double *h = &s.H[(size_t)j * (size_t)(m + 1)];
      double wn = 0.0;
      for (int q = 0; q <= j; ++q) h[q] = 0.0;
      ortho_collect(ctx, j + 1, h, &wn, mb);
      h[j + 1] = wn;
      for (int q = 0; q < j; ++q) {
        const double t = s.cs[(size_t)q] * h[q] + s.sn[(size_t)q] * h[q + 1];
        h[q + 1] = -s.sn[(size_t)q] * h[q] + s.cs[(size_t)q] * h[q + 1];
        h[q] = t;
      }
      {
        const double t = h[j], bb = h[j + 1], rr = std::hypot(t, bb);
        s.cs[(size_t)j] = rr == 0.0 ? 1.0 : t / rr;
        s.sn[(size_t)j] = rr == 0.0 ? 0.0 : bb / rr;
        h[j] = rr;
        h[j + 1] = 0.0;
        s.g[(size_t)j + 1] = -s.sn[(size_t)j] * s.g[(size_t)j];
        s.g[(size_t)j] = s.cs[(size_t)j] * s.g[(size_t)j];
      }
      ++s.j;
      ++inf.iters;
      inf.rel_res_implicit = std::fabs(s.g[(size_t)s.j]) / s.scale;
      if (prm->verbose && ctx->rank == 0 && inf.iters % 10 == 0)
        printf(">> isph::gmres[%d] iter %d  rel res %.3e\n", k, inf.iters, inf.rel_res_implicit);
      if (inf.rel_res_implicit <= prm->tol) inf.converged = 1;
      if (inf.converged || inf.iters >= prm->max_iters || s.j >= m) ISPH_CHECK(end_cycle(k));
    }
  }
  ISPH_CHECK_HIP(hipGetLastError());
  return ISPH_SUCCESS;
}

// Belos BlockCGSolMgr, block size 1: the right-preconditioner slot is the
// preconditioner (ref: USER-REAXC-T/solver_lin_belos.h:236-245).
inline int pcg(const LinOp &op, const double *b, double *x, const isph_solver_params *prm, isph_solve_info *info) {
  isph_ctx *ctx = op.ctx;
  const int n = op.n;
  const long long ld = ((long long)n + 63) / 64 * 64;
  ISPH_CHECK(ctx->wv.reserve((size_t)ld));
  ISPH_CHECK(ctx->tv.reserve((size_t)ld));
  ISPH_CHECK(ctx->rv.reserve((size_t)ld));
  ISPH_CHECK(ctx->pv.reserve((size_t)ld));
  double *r = ctx->rv.p, *z = ctx->tv.p, *p = ctx->pv.p, *ap = ctx->wv.p;
  const int sg = stream_grid(n);
  hipStream_t st = ctx->stream;
  double *ds = ctx->dscal.p;
  enum { RZ = SC_MISC + 8, PAP = SC_MISC + 10, RR = SC_MISC + 12, RZN = SC_MISC + 14 };
  ISPH_CHECK(op.apply(x, r));
  hipLaunchKernelGGL(k_residual, dim3(sg), dim3(kBlock), 0, st, n, b, r);
  ISPH_CHECK(op.prec(r, z));
  ISPH_CHECK_HIP(hipMemcpyAsync(p, z, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
  ISPH_CHECK(dot_dev(ctx, n, r, z, r, r, RZ));  // RZ = r.z, RZ+1 = r.r
  ISPH_CHECK(fetch_scalars(ctx, RZ, 2));
  const double r0 = std::sqrt(ctx->hscal[RZ + 1]);
  const double scale = r0 == 0.0 ? 1.0 : r0;
  info->iters = 0;
  info->restarts = 0;
  info->rel_res_implicit = r0 / scale;
  info->converged = (r0 / scale <= prm->tol);
  int rz = RZ, rzn = RZN;
  // One host wait per iteration (the residual norm), and it is taken AFTER the next iteration's preconditioner
  // application, direction update and operator application have been enqueued: they only write z, p, A p and scalars,
  // so when the norm says "converged" they are wasted but harmless, and otherwise the GPU never idles while the host
  // launches (small systems are launch-latency bound: BASELINE configs[0] is 16 k rows).
  bool have_ap = false;
  while (!info->converged && info->iters < prm->max_iters) {
    if (!have_ap) {
      ISPH_CHECK(op.apply(p, ap));
      ISPH_CHECK(dot_dev(ctx, n, p, ap, nullptr, nullptr, PAP));
    }
    const int g = stream_grid(n);
    hipLaunchKernelGGL(k_cg_update_xr, dim3(g), dim3(kBlock), 0, st, n, p, ap, x, r, ds + rz, ds + PAP, ctx->partial.p);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(kBlock), 0, st, 1, g, ctx->partial.p, ds + RR);
    ISPH_CHECK(allreduce_inplace(ctx, ds + RR, 1));
    ISPH_CHECK_HIP(hipMemcpyAsync(ctx->hscal + RR, ds + RR, sizeof(double), hipMemcpyDeviceToHost, st));
    ISPH_CHECK_HIP(hipEventRecord(ctx->ev_fetch, st));
    ++info->iters;
    have_ap = false;
    if (info->iters < prm->max_iters) {  // next iteration's first half, enqueued behind the copy
      ISPH_CHECK(op.prec(r, z));
      ISPH_CHECK(dot_dev(ctx, n, r, z, nullptr, nullptr, rzn));
      hipLaunchKernelGGL(k_cg_update_p, dim3(sg), dim3(kBlock), 0, st, n, z, p, ds + rzn, ds + rz);
      const int tsw = rz; rz = rzn; rzn = tsw;
      ISPH_CHECK(op.apply(p, ap));
      ISPH_CHECK(dot_dev(ctx, n, p, ap, nullptr, nullptr, PAP));
      have_ap = true;
    }
    ISPH_CHECK_HIP(hipEventSynchronize(ctx->ev_fetch));
    info->rel_res_implicit = std::sqrt(ctx->hscal[RR]) / scale;
    if (prm->verbose && ctx->rank == 0 && info->iters % 10 == 0)
      printf(">> isph::cg iter %d  rel res %.3e\n", info->iters, info->rel_res_implicit);
    if (info->rel_res_implicit <= prm->tol) { info->converged = 1; break; }
  }
  ISPH_CHECK_HIP(hipGetLastError());
  return ISPH_SUCCESS;
}

}  // namespace isph
