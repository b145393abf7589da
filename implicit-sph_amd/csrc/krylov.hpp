// krylov.hpp -- fused Krylov primitives (what Epetra_MultiVector Dot/Norm2/
// Update/Scale and Belos' orthogonalisation supply in the reference:
// solver_lin.h:136-137, solver_lin_belos.h:142-143,217-218).
//
// All reductions are two-stage and deterministic: per-block partials via DPP
// wave sums (+ LDS across the 4 waves), then one small kernel that adds the
// partials in a fixed order.  Scalars stay on the device between kernels; the
// host reads one small batch per Gram-Schmidt pass.
#pragma once
#include "common.hpp"

namespace isph {

constexpr int kDotBatch = 16;  // basis vectors whose partial sums a thread keeps in registers

// partial[(k)*nblk + blockIdx] = sum over this block's rows of V_k[i]*w[i],
// k in [0,nk); slot nk holds w.w.  V_k = V + k*ld.
// Basis vectors are processed in batches of kDotBatch: every thread keeps one running sum per vector of the
// batch in registers while it grid-strides over its rows, so the wave/block reduction (DPP + LDS) happens once
// per batch instead of once per row chunk; w is re-read once per batch (+8N bytes per 16 vectors).
// The Gram-Schmidt form: two ADJACENT rows per thread and 16-byte load (ld is a multiple of 64 and the vectors come from the
// pool: 16-byte aligned; the caller checks), one trip in flight -- 43.5 -> 36 us at nk ~ 27 against k_multi_dot<2> (two
// strided rows, 8-byte loads).  The same change in the two update kernels below measured no gain (39.3 -> 40.1, 36.6 -> 37.6 us).
__global__ __launch_bounds__(kBlock) void k_multi_dot_v2(int n, int nk, const double *__restrict__ V, long long ld,
                                                         const double *__restrict__ w, double *__restrict__ partial) {
  __shared__ double sred[kDotBatch + 1][4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long stride = (long long)gridDim.x * kBlock, np = n >> 1;
  const long long p0 = (long long)blockIdx.x * kBlock + threadIdx.x;
  const double2 *__restrict__ w2 = reinterpret_cast<const double2 *>(w);
  for (int k0 = 0; k0 < nk || k0 == 0; k0 += kDotBatch) {
    const int nb = min(kDotBatch, nk - k0);
    const bool last = k0 + kDotBatch >= nk;
    double acc[kDotBatch];
#pragma unroll
    for (int u = 0; u < kDotBatch; ++u) acc[u] = 0.0;
    double ww = 0.0;
    const double *__restrict__ vb = V + (long long)k0 * ld;
    for (long long p = p0; p < np; p += stride) {
      const double2 wi = w2[p];
      if (last) ww = fma(wi.y, wi.y, fma(wi.x, wi.x, ww));
      double2 v[kDotBatch];
#pragma unroll
      for (int u = 0; u < kDotBatch; ++u)
        if (u < nb) v[u] = reinterpret_cast<const double2 *>(vb + (long long)u * ld)[p];
#pragma unroll
      for (int u = 0; u < kDotBatch; ++u)
        if (u < nb) acc[u] = fma(v[u].y, wi.y, fma(v[u].x, wi.x, acc[u]));
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {   // the odd last row
      const double wi = w[n - 1];
      if (last) ww = fma(wi, wi, ww);
      for (int u = 0; u < nb; ++u) acc[u] = fma(vb[(long long)u * ld + n - 1], wi, acc[u]);
    }
#pragma unroll
    for (int u = 0; u < kDotBatch; ++u) {
      if (u < nb) {
        const double s = wave_sum(acc[u]);
        if (lane == 0) sred[u][wave] = s;
      }
    }
    if (last) {
      ww = wave_sum(ww);
      if (lane == 0) sred[kDotBatch][wave] = ww;
    }
    __syncthreads();
    if (threadIdx.x < nb)
      partial[(long long)(k0 + threadIdx.x) * gridDim.x + blockIdx.x] =
          (sred[threadIdx.x][0] + sred[threadIdx.x][1]) + (sred[threadIdx.x][2] + sred[threadIdx.x][3]);
    if (last && threadIdx.x == 0)
      partial[(long long)nk * gridDim.x + blockIdx.x] =
          (sred[kDotBatch][0] + sred[kDotBatch][1]) + (sred[kDotBatch][2] + sred[kDotBatch][3]);
    __syncthreads();
    if (last) break;
  }
}

template <int ROWS>
__global__ __launch_bounds__(kBlock) void k_multi_dot(int n, int nk, const double *__restrict__ V, long long ld,
                                                      const double *__restrict__ w,
                                                      double *__restrict__ partial) {
  __shared__ double sred[kDotBatch + 1][4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long stride = (long long)gridDim.x * kBlock;
  const long long i0 = (long long)blockIdx.x * kBlock + threadIdx.x;
  for (int k0 = 0; k0 < nk || k0 == 0; k0 += kDotBatch) {
    const int nb = min(kDotBatch, nk - k0);
    const bool last = k0 + kDotBatch >= nk;  // the last batch also accumulates w.w
    double acc[kDotBatch];
#pragma unroll
    for (int u = 0; u < kDotBatch; ++u) acc[u] = 0.0;
    double ww = 0.0;
    const double *__restrict__ vb = V + (long long)k0 * ld;
    long long i = i0;
    if (ROWS == 2) {  // two rows per trip: twice the loads in flight per thread
      for (; i + stride < n; i += 2 * stride) {
        const double wa = w[i], wb = w[i + stride];
        if (last) ww = fma(wb, wb, fma(wa, wa, ww));
        double va[kDotBatch], vc[kDotBatch];
#pragma unroll
        for (int u = 0; u < kDotBatch; ++u)
          if (u < nb) { va[u] = vb[(long long)u * ld + i]; vc[u] = vb[(long long)u * ld + i + stride]; }
#pragma unroll
        for (int u = 0; u < kDotBatch; ++u)
          if (u < nb) acc[u] = fma(vc[u], wb, fma(va[u], wa, acc[u]));
      }
    }
    for (; i < n; i += stride) {
      const double wi = w[i];
      if (last) ww = fma(wi, wi, ww);
#pragma unroll
      for (int u = 0; u < kDotBatch; ++u)
        if (u < nb) acc[u] = fma(vb[(long long)u * ld + i], wi, acc[u]);
    }
#pragma unroll
    for (int u = 0; u < kDotBatch; ++u) {
      if (u < nb) {  // wave-uniform
        const double s = wave_sum(acc[u]);
        if (lane == 0) sred[u][wave] = s;
      }
    }
    if (last) {
      ww = wave_sum(ww);
      if (lane == 0) sred[kDotBatch][wave] = ww;
    }
    __syncthreads();
    if (threadIdx.x < nb)
      partial[(long long)(k0 + threadIdx.x) * gridDim.x + blockIdx.x] =
          (sred[threadIdx.x][0] + sred[threadIdx.x][1]) + (sred[threadIdx.x][2] + sred[threadIdx.x][3]);
    if (last && threadIdx.x == 0)
      partial[(long long)nk * gridDim.x + blockIdx.x] =
          (sred[kDotBatch][0] + sred[kDotBatch][1]) + (sred[kDotBatch][2] + sred[kDotBatch][3]);
    __syncthreads();
    if (last) break;
  }
}

// out[k] = sum_b partial[k*nblk + b]: one 256-thread block per k, fixed
// (launch-independent) summation order -> bitwise reproducible
__global__ __launch_bounds__(kBlock) void k_reduce_partials(int nk, int nblk, const double *__restrict__ partial,
                                                            double *__restrict__ out,
                                                            const double *__restrict__ flag = nullptr) {
  __shared__ double sw[4];
  const int k = blockIdx.x;
  if (k >= nk) return;
  if (flag && *flag == 0.0) {  // the producing kernel was skipped: nothing to sum
    if (threadIdx.x == 0) out[k] = 0.0;
    return;
  }
  const double *__restrict__ p = partial + (long long)k * nblk;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int b = threadIdx.x;
  for (; b + 3 * kBlock < nblk; b += 4 * kBlock) {
    s0 += p[b];
    s1 += p[b + kBlock];
    s2 += p[b + 2 * kBlock];
    s3 += p[b + 3 * kBlock];
  }
  for (; b < nblk; b += kBlock) s0 += p[b];
  double s = wave_sum((s0 + s1) + (s2 + s3));
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[k] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}

// w -= sum_k c[k] V_k ; partial[blockIdx] = sum of the new w.w  (c on device)
// dec_old != NULL: the DGKS test is taken here, by every workgroup for itself, from the reduced scalars of the two
// passes -- dec_old = |w_old|^2 (minus dec_dn[0]^2, the null-vector component, with deflation), c[0..nk) = c2,
// c[nk] = |w_new|^2: second pass when force (ICGS) or |w_new| < |w_old| / sqrt(2); |w_final|^2 = |w_new|^2 - |c2|^2
// then, |w_new|^2 otherwise (Pythagoras: c2 is the projection of w_new on span V).  Workgroup 0 leaves the flag and
// |w_final|^2 in dec_out[0..1] for the host.  Without a second pass w stays.
// vnext != NULL: the normalised vector vnext = w_final / |w_final| is written in the same sweep, second pass or not.
__global__ __launch_bounds__(kBlock) void k_multi_axpy_norm(int n, int nk, const double *__restrict__ V, long long ld,
                                                            const double *__restrict__ c, double *__restrict__ w,
                                                            double *__restrict__ partial,
                                                            const double *__restrict__ dec_old = nullptr,
                                                            const double *__restrict__ dec_dn = nullptr, int dec_force = 0,
                                                            double *__restrict__ dec_out = nullptr,
                                                            double *__restrict__ vnext = nullptr) {
  __shared__ double sw[4];
  __shared__ double sdec[2];
  bool second = true;
  double wfinal2 = 1.0;
  if (dec_old) {
    if (threadIdx.x == 0) {
      double old2 = *dec_old;
      if (dec_dn) old2 = fmax(old2 - dec_dn[0] * dec_dn[0], 0.0);
      const double ww_new = c[nk];
      const bool sec = dec_force || sqrt(ww_new) < M_SQRT1_2 * sqrt(old2);
      double s2 = 0.0;
      for (int k = 0; k < nk; ++k) s2 += c[k] * c[k];
      sdec[0] = sec ? 1.0 : 0.0;
      sdec[1] = sec ? fmax(ww_new - s2, 0.0) : ww_new;
      if (blockIdx.x == 0) { dec_out[0] = sdec[0]; dec_out[1] = sdec[1]; }
    }
    __syncthreads();
    second = sdec[0] != 0.0;
    wfinal2 = sdec[1];
  }
  if (!second && !vnext) return;
  double a = 1.0;
  if (vnext) a *= 1.0 / sqrt(wfinal2);
  double ww = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    double s = w[i];
    if (second) {
      for (int k = 0; k < nk; ++k) s = fma(-c[k], V[(long long)k * ld + i], s);
      w[i] = s;
    }
    if (vnext) vnext[i] = a * s;
    ww = fma(s, s, ww);
  }
  ww = wave_sum(ww);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = ww;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}

// Fused first-pass update + second-pass projection of a DGKS / ICGS step:
//   w -= sum_k c[k] V_k ;  partial[k*nblk+b] = sum_rows V_k[i] * w_new[i] ;  slot nk = w_new . w_new
// The second Gram-Schmidt pass needs V^T w_new; w_new[i] only depends on row i, so the projection is
// accumulated while the row of V is still in registers: V is read once instead of twice per pass pair.
template <int NK>
__global__ __launch_bounds__(kBlock) void k_multi_axpy_dot(int n, int nk, const double *__restrict__ V, long long ld,
                                                           const double *__restrict__ c, double *__restrict__ w,
                                                           double *__restrict__ partial) {
  __shared__ double sred[NK + 1][4];
  __shared__ double sc[NK];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x < NK) sc[threadIdx.x] = threadIdx.x < nk ? c[threadIdx.x] : 0.0;
  __syncthreads();
  double acc[NK];
#pragma unroll
  for (int u = 0; u < NK; ++u) acc[u] = 0.0;
  double ww = 0.0;
  const long long stride = (long long)gridDim.x * kBlock;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    double v[NK];
#pragma unroll
    for (int u = 0; u < NK; ++u) v[u] = u < nk ? V[(long long)u * ld + i] : 0.0;
    double s = w[i];
#pragma unroll
    for (int u = 0; u < NK; ++u) s = fma(-sc[u], v[u], s);
    w[i] = s;
    ww = fma(s, s, ww);
#pragma unroll
    for (int u = 0; u < NK; ++u) acc[u] = fma(v[u], s, acc[u]);
  }
#pragma unroll
  for (int u = 0; u < NK; ++u) {
    if (u < nk) {  // wave-uniform
      const double r = wave_sum(acc[u]);
      if (lane == 0) sred[u][wave] = r;
    }
  }
  ww = wave_sum(ww);
  if (lane == 0) sred[NK][wave] = ww;
  __syncthreads();
  if (threadIdx.x < nk)
    partial[(long long)threadIdx.x * gridDim.x + blockIdx.x] =
        (sred[threadIdx.x][0] + sred[threadIdx.x][1]) + (sred[threadIdx.x][2] + sred[threadIdx.x][3]);
  if (threadIdx.x == 0)
    partial[(long long)nk * gridDim.x + blockIdx.x] = (sred[NK][0] + sred[NK][1]) + (sred[NK][2] + sred[NK][3]);
}

// x += sum_k c[k] Z_k   (solution update, c on device)
__global__ void k_multi_axpy(int n, int nk, const double *__restrict__ Z, long long ld, const double *__restrict__ c,
                             double *__restrict__ x) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    double s = x[i];
    for (int k = 0; k < nk; ++k) s = fma(c[k], Z[(long long)k * ld + i], s);
    x[i] = s;
  }
}

// dot partials of two vectors (plus optional second pair): partial[0*nblk+b]=a.b
__global__ __launch_bounds__(kBlock) void k_dot2(int n, const double *__restrict__ a, const double *__restrict__ b,
                                                 const double *__restrict__ c, const double *__restrict__ d,
                                                 double *__restrict__ partial) {
  __shared__ double s0[4], s1[4];
  double p = 0.0, q = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    p = fma(a[i], b[i], p);
    if (c) q = fma(c[i], d[i], q);
  }
  p = wave_sum(p);
  q = wave_sum(q);
  if ((threadIdx.x & 63) == 0) { s0[threadIdx.x >> 6] = p; s1[threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = (s0[0] + s0[1]) + (s0[2] + s0[3]);
    partial[gridDim.x + blockIdx.x] = (s1[0] + s1[1]) + (s1[2] + s1[3]);
  }
}

// y = a*x  with a = alpha_host * (inv_sqrt ? 1/sqrt(*s) : (s ? *s : 1))
__global__ void k_scale_copy(int n, const double *__restrict__ x, double *__restrict__ y, double alpha,
                             const double *__restrict__ s, int inv_sqrt) {
  double a = alpha;
  if (s) a *= inv_sqrt ? 1.0 / sqrt(*s) : *s;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = a * x[i];
}

// y += (alpha * (*s)) * x       (s on device, may be NULL)
__global__ void k_axpy_dev(int n, double alpha, const double *__restrict__ s, const double *__restrict__ x,
                           double *__restrict__ y) {
  const double a = s ? alpha * (*s) : alpha;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = fma(a, x[i], y[i]);
}

// r = b - r
__global__ void k_residual(int n, const double *__restrict__ b, double *__restrict__ r) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    r[i] = b[i] - r[i];
}

// p = z + beta p   with beta = num/den on device
__global__ void k_cg_update_p(int n, const double *__restrict__ z, double *__restrict__ p,
                              const double *__restrict__ num, const double *__restrict__ den) {
  const double beta = *num / *den;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    p[i] = fma(beta, p[i], z[i]);
}

// x += a p ; r -= a ap ; a = num/den on device; partial = new r.r
__global__ __launch_bounds__(kBlock) void k_cg_update_xr(int n, const double *__restrict__ p,
                                                         const double *__restrict__ ap, double *__restrict__ x,
                                                         double *__restrict__ r, const double *__restrict__ num,
                                                         const double *__restrict__ den, double *__restrict__ partial) {
  __shared__ double sw[4];
  const double a = *num / *den;
  double rr = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    x[i] = fma(a, p[i], x[i]);
    const double t = fma(-a, ap[i], r[i]);
    r[i] = t;
    rr = fma(t, t, rr);
  }
  rr = wave_sum(rr);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = rr;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}

__global__ void k_mul_elem(int n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ y) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = a[i] * b[i];
}

__global__ void k_fill(int n, double *__restrict__ y, double v) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = v;
}

__global__ void k_mask_to_double(int n, const int *__restrict__ m, double *__restrict__ y) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = (double)m[i];
}

inline int stream_grid(long long n) {
  long long g = (n + kBlock - 1) / kBlock;
  if (g > kMaxRedBlocks) g = kMaxRedBlocks;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace isph
