// schwarz.hpp -- Ifpack_AdditiveSchwarz<Ifpack_ILU> with the reference's own semantics on the GPU.
//
// Replaces PrecondWrapper_Ifpack::create() (ref: precond_ifpack.h:52-75) for the settings the block-Jacobi
// stream of ilu.hpp cannot express:
//   * one subdomain = the WHOLE local matrix (what Ifpack factors on one MPI rank; "ilu<k>"),
//   * subdomains of any size (no 1024-row limit), extended by "Overlap Level" layers of rows
//     (precond_ifpack.h:43, Ifpack_OverlappingRowMatrix) and combined with "schwarz: combine mode"
//     Add (the reference, :37) or Zero (restricted additive Schwarz),
//   * "fact: level-of-fill" = k for such subdomains.
//
// Division of labour.  The integer work that is sequential by definition -- the level-of-fill pattern of
// Ifpack_IlukGraph (row i merges the FINAL patterns of the rows it eliminates with) and the dependency levels of
// the two triangular solves -- runs on the host, once per create(), over the matrix pattern (threads over
// subdomains); the level-1 pattern depends on A's pattern only, and for a whole-matrix factor it is built on the device
// (k_gilu_symbolic1).  The floating-point work runs on the device, as ONE persistent launch per phase whose rows wait for
// the very words they depend on (round 3; the round-2 form, one launch per dependency level, is kept behind
// isph_schwarz_params::level_launches as the bit-for-bit cross-check):
//   k_gilu_factor_sf   IKJ numeric factorisation, one wave per row, rows dequeued in level order; the row image
//                      (columns + values) lives in LDS; a finished row's values are their own ready flags
//   k_gilu_solve_run   the triangular sweeps, 16 lanes per row, runs of 64 positions per workgroup; a result is its
//                      own ready flag, handed on through LDS inside a run and through global memory between runs
//   k_gilu_gather / k_gilu_combine   import on the extended rows / export with the combine mode (fixed
//                      summation order: bitwise reproducible)
// What bounds it: the dependency chain.  A whole-matrix factor of the periodic 100^3 bench system has 67 084 levels
// per direction; measured per level: 1.35 us for a hand-off between workgroups through global
// memory, 0.40 us inside a workgroup through LDS, 0.61 us on average over a sweep; 6.6 us in the factorisation.  It exists for fidelity with the reference's configuration and for the systems the reference
// itself runs on one rank (BASELINE configs[0]); the production path for large systems stays the block stream of
// ilu.hpp, whose blocks break the chain.
#pragma once
#include <algorithm>
#include <chrono>
#include <climits>
#include <queue>
#include <thread>
#include <type_traits>

#include "core.hpp"
#include "sell.hpp"

struct isph_schwarz {
  int n = 0, nsub = 0, nloc = 0, fill = 0, overlap = 0, combine = 0, maxrow = 0;
  long long nnz = 0;
  int nlev_l = 0, nlev_u = 0;
  isph::DevBuf<long long> rp;     // [nloc+1]
  isph::DevBuf<int> ci, dg;       // [nnz] local columns (ascending), [nloc] position of the diagonal
  isph::DevBuf<double> val, w;    // [nnz] factor (strict L, D, strict U), [nloc] work vector
  isph::DevBuf<int> rows;         // [nloc] global row of every local row
  isph::DevBuf<int> lord, uord;   // [nloc] local rows by L- / U-level
  isph::DevBuf<long long> rev_ptr;  // [n+1]  global row -> contributing local rows (combine)
  isph::DevBuf<int> rev_idx;
  std::vector<int> lptr, uptr;    // host: first entry of every level in lord / uord
  std::vector<int> loc_ptr;       // host: [nsub+1]
  isph::DevBuf<int> err;
  // synchronisation-free path (one persistent launch per sweep, see k_gilu_solve_run): the level orders padded so that
  // every level starts at a multiple of 4 positions (-1 = padding), the two result vectors that double as ready flags,
  // and the work counters [0] L sweep, [1] U sweep, [2] factorisation, [3] spin time-out, [4] [5] runs completed (L, U)
  isph::DevBuf<int> lord4, uord4, ctr;
  isph::DevBuf<int> lpos4, upos4;  // [nloc] position of every local row in lord4 / uord4
  isph::DevBuf<int> lrun, urun;    // first position of every run of the LDS hand-off sweeps (+ end)
  int nrun_l = 0, nrun_u = 0;
  isph::DevBuf<unsigned long long> ybits, zbits;
  int n4l = 0, n4u = 0;
  bool long_rows = false;   // many rows have more entries on one side of the diagonal than a chunk of the sweeps holds
  bool syncfree = true;
  // many small subdomains (k_gilu_solve_sub): ONE launch per application, one workgroup per subdomain with its part of
  // the vector in LDS and a barrier per dependency level.  Per subdomain ONE list of levels -- the L levels with work
  // (1 ..), then the U levels (0 ..) -- with a 16-byte descriptor per row in that order (sub_desc: first entry on the
  // sweep's side of the diagonal, their number, the local row), level starts sub_lptr, per-subdomain offsets into both
  // and the number of L levels in the list
  bool subsweep = false, syncfree_requested = true;
  int sub_maxrows = 0;
  isph::DevBuf<long long> sub_desc;              // [2 x 8 bytes per listed row]: start | n, i  (layout: k_gilu_sub_levels)
  isph::DevBuf<int> sub_lptr, sub_nlist, loc_ptr_dev;
  int *h_tmo = nullptr;   // pinned: the time-out word of the last application (read by prec_health with the stream drained)
  // persistent workgroups (1024 threads) of the L / U sweep, and how many earlier runs may still be open when a run
  // starts polling (k_gilu_solve_run): both follow the width of the levels
  int sweep_blocks[2] = {0, 0}, runs_near[2] = {2, 2};
  // wall time of create(), ms: [0] matrix to the host [1] subdomains + local matrices [2] level-of-fill pattern
  // [3] dependency levels, orders, runs, combine lists [4] upload [5] numeric factorisation (to the final synchronise)
  double t_ms[6] = {0, 0, 0, 0, 0, 0};
};

namespace isph {

constexpr int kGiluWideLevel = 4096;  // rows per dependency level from which one launch per level is the faster form
constexpr int kGiluMaxRow = 5000;  // row image in LDS: 12 B per entry, below the 64 KiB default dynamic-LDS limit

// rl = r[rows]
__global__ void k_gilu_gather(int nloc, const int *__restrict__ rows, const double *__restrict__ r,
                              double *__restrict__ w) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < nloc) w[q] = r[rows[q]];
}

// z[g] = sum of the local rows that map to g (Add: all of them, Zero: the owner's only -- the lists are built so)
__global__ void k_gilu_combine(int n, const long long *__restrict__ rev_ptr, const int *__restrict__ rev_idx,
                               const double *__restrict__ w, double *__restrict__ z) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  double s = 0.0;
  for (long long p = rev_ptr[g]; p < rev_ptr[g + 1]; ++p) s += w[rev_idx[p]];
  z[g] = s;
}

// scatter the values of the local matrix into the factor pattern happened on the host; this kernel factors the rows
// of one dependency level.  One wave (= one workgroup) per row.
__global__ __launch_bounds__(64) void k_gilu_factor(int count, const int *__restrict__ order,
                                                    const long long *__restrict__ rp, const int *__restrict__ ci,
                                                    const int *__restrict__ dg, double *__restrict__ val,
                                                    int *__restrict__ err) {
  extern __shared__ double gilu_lds[];
  const int i = order[blockIdx.x];
  const long long b = rp[i];
  const int len = (int)(rp[i + 1] - b), nlow = dg[i];
  double *w = gilu_lds;
  int *cols = reinterpret_cast<int *>(gilu_lds + len);
  const int lane = threadIdx.x;
  for (int t = lane; t < len; t += 64) {
    w[t] = val[b + t];
    cols[t] = ci[b + t];
  }
  __syncthreads();
  for (int t = 0; t < nlow; ++t) {
    const int k = cols[t];
    const long long kb = rp[k], ke = rp[k + 1];
    const int kd = dg[k];
    const double lik = w[t] / val[kb + kd];
    __syncthreads();
    if (lane == 0) w[t] = lik;
    for (long long q = kb + kd + 1 + lane; q < ke; q += 64) {
      const int j = ci[q];
      // binary search for column j among this row's columns right of position t
      int lo = t + 1, hi = len - 1, pos = -1;
      while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const int c = cols[mid];
        if (c == j) { pos = mid; break; }
        if (c < j) lo = mid + 1; else hi = mid - 1;
      }
      if (pos >= 0) w[pos] -= lik * val[q];
    }
    __syncthreads();
  }
  if (len > nlow && !(fabs(w[nlow]) > 0.0) && lane == 0) atomicOr(err, 2);  // zero pivot
  for (int t = lane; t < len; t += 64) val[b + t] = w[t];
}

// y_i = r_i - sum_{p < diag} l_ip y_p for the rows of one level; 16 lanes per row
__global__ __launch_bounds__(256) void k_gilu_lower(int count, const int *__restrict__ order,
                                                    const long long *__restrict__ rp, const int *__restrict__ ci,
                                                    const int *__restrict__ dg, const double *__restrict__ val,
                                                    double *__restrict__ w) {
  const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, sub = threadIdx.x & 15;
  const bool live = g < count;
  const int i = live ? order[g] : 0;
  double s = 0.0;
  if (live) {
    const long long b = rp[i];
    const int nlow = dg[i];
    for (int t = sub; t < nlow; t += 16) s += val[b + t] * w[ci[b + t]];
  }
  s = group16_sum(s);
  if (live && sub == 0) w[i] -= s;
}

// z_i = (y_i - sum_{p > diag} u_ip z_p) / d_i
__global__ __launch_bounds__(256) void k_gilu_upper(int count, const int *__restrict__ order,
                                                    const long long *__restrict__ rp, const int *__restrict__ ci,
                                                    const int *__restrict__ dg, const double *__restrict__ val,
                                                    double *__restrict__ w) {
  const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, sub = threadIdx.x & 15;
  const bool live = g < count;
  const int i = live ? order[g] : 0;
  double s = 0.0, d = 1.0;
  if (live) {
    const long long b = rp[i], e = rp[i + 1];
    const int kd = dg[i];
    d = val[b + kd];
    for (long long q = b + kd + 1 + sub; q < e; q += 16) s += val[q] * w[ci[q]];
  }
  s = group16_sum(s);
  if (live && sub == 0) w[i] = (w[i] - s) / d;
}

// ---- many small subdomains: both sweeps of one subdomain in one workgroup ----------------------------------------------
// Additive Schwarz on hundreds of subdomains of a few thousand rows (512-row blocks + one overlap layer at 10^6 rows: 1954
// subdomains of ~2200 rows, 294 levels each): a launch per global level is 588 launches of 15 000 short rows per
// application, the persistent form pays a hand-off through memory per level.  Here a workgroup owns a subdomain: its part
// of the vector lives in LDS (the gather of r is fused in), a level's rows take 16 lanes each, one barrier per level, L
// sweep then U sweep, result to w for k_gilu_combine.  Rows of a subdomain only depend on rows of the same subdomain.
constexpr int kSubSweepMaxRows = 4096;   // 32 KB of LDS per workgroup
constexpr int kSubSweepMinSubs = 32;     // fewer subdomains do not fill the chip this way
constexpr int kSubFactorMaxRow = 1024;   // longest factor row of the per-subdomain factorisation (row image: 12 B per entry and wave;
                                         // eight waves per subdomain up to 512 entries, four beyond: 48 KB of LDS either way)
// one row of a level as a group of 16 lanes holds it: up to kSubK entries per lane in registers (rows of up to 16 kSubK
// entries on the side in question; longer ones finish from memory).  Nothing of it depends on the vector, so it is fetched
// ahead: a level is ~8 rows and ~0.3 us of work, a load from HBM ~2 us, so the entries of the group's row travel FOUR levels
// ahead (four row images in registers, the loop unrolled by four so that they keep their registers) and its 16-byte
// descriptor five -- the dependent chain descriptor -> entries never sits between two barriers.
constexpr int kSubK = 4;
struct SubDesc { long long start; int n, i; };   // i < 0: the group has no row in the level
struct SubRow {
  SubDesc d;
  double piv;
  double v[kSubK];
  int c[kSubK];
};
__device__ __forceinline__ SubDesc sub_desc_at(const long long *__restrict__ D, long long q) {
  SubDesc d;
  d.start = D[2 * q];
  const long long w = D[2 * q + 1];
  d.n = (int)(w & 0xffffffffLL);
  d.i = (int)(w >> 32);
  return d;
}
__device__ __forceinline__ SubDesc sub_desc_get(const long long *__restrict__ D, const int *__restrict__ lp, int l, int nlev, int grp) {
  SubDesc d{0, 0, -1};
  if (l < nlev) {
    const int q = lp[l] + grp;
    if (q < lp[l + 1]) d = sub_desc_at(D, q);
  }
  return d;
}
__device__ __forceinline__ void sub_row_load(SubRow &R, const SubDesc &d, bool upper, int base, int sub,
                                             const int *__restrict__ ci, const double *__restrict__ val) {
  R.d = d;
  R.piv = (upper && d.i >= 0) ? val[d.start - 1] : 1.0;   // the pivot sits in front of the upper part
#pragma unroll
  for (int k = 0; k < kSubK; ++k) {
    const int t = sub + 16 * k;
    const bool in = t < d.n;
    R.v[k] = in ? val[d.start + t] : 0.0;
    R.c[k] = in ? ci[d.start + t] - base : 0;
  }
}
__device__ __forceinline__ void sub_row_apply(const SubRow &R, bool upper, int base, int sub, const int *__restrict__ ci,
                                              const double *__restrict__ val, double *sub_y) {
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < kSubK; ++k) acc = fma(R.v[k], sub_y[R.c[k]], acc);   // absent entries: 0 * y[0]
  for (int t = sub + 16 * kSubK; t < R.d.n; t += 16) acc = fma(val[R.d.start + t], sub_y[ci[R.d.start + t] - base], acc);
  acc = group16_sum(acc);
  if (R.d.i >= 0 && sub == 0) sub_y[R.d.i - base] = upper ? (sub_y[R.d.i - base] - acc) / R.piv : sub_y[R.d.i - base] - acc;
}

__global__ __launch_bounds__(256) void k_gilu_solve_sub(const int *__restrict__ loc_ptr, const int *__restrict__ rows,
                                                        const long long *__restrict__ desc, const int *__restrict__ lptr,
                                                        const int *__restrict__ nlist, const int *__restrict__ ci,
                                                        const double *__restrict__ val, const double *__restrict__ r,
                                                        double *__restrict__ w) {
  extern __shared__ double sub_y[];
  const int s = blockIdx.x, base = loc_ptr[s], m = loc_ptr[s + 1] - base;
  const int tid = threadIdx.x, grp = tid >> 4, sub = tid & 15, ngrp = blockDim.x >> 4;
  const long long *D = desc + 2 * (2 * (long long)base);   // the layout of k_gilu_sub_levels
  const int *lp = lptr + 2 * base + 2 * s;
  const int nL = nlist[2 * s], nlev = nlist[2 * s + 1];
  SubRow r0, r1, r2, r3;
  sub_row_load(r0, sub_desc_get(D, lp, 0, nlev, grp), 0 >= nL, base, sub, ci, val);
  sub_row_load(r1, sub_desc_get(D, lp, 1, nlev, grp), 1 >= nL, base, sub, ci, val);
  sub_row_load(r2, sub_desc_get(D, lp, 2, nlev, grp), 2 >= nL, base, sub, ci, val);
  sub_row_load(r3, sub_desc_get(D, lp, 3, nlev, grp), 3 >= nL, base, sub, ci, val);
  SubDesc dpend = sub_desc_get(D, lp, 4, nlev, grp);   // descriptor of the row four levels ahead of the level at hand
  for (int t = tid; t < m; t += blockDim.x) sub_y[t] = r[rows[base + t]];
  __syncthreads();
  // level `lvl` with the row image R, then R is refilled for level lvl + 4 (its descriptor arrived a level ago)
#define ISPH_SUB_STEP(R, lvl)                                                                                   \
  {                                                                                                             \
    const int l_ = (lvl);                                                                                       \
    const bool upper_ = l_ >= nL;                                                                               \
    const SubDesc dn_ = sub_desc_get(D, lp, l_ + 5, nlev, grp);                                                 \
    sub_row_apply(R, upper_, base, sub, ci, val, sub_y);                                                        \
    for (int q = lp[l_] + grp + ngrp; q < lp[l_ + 1]; q += ngrp) { /* more rows than groups: the rest from memory */ \
      SubRow extra_;                                                                                            \
      sub_row_load(extra_, sub_desc_at(D, q), upper_, base, sub, ci, val);                                      \
      sub_row_apply(extra_, upper_, base, sub, ci, val, sub_y);                                                 \
    }                                                                                                           \
    sub_row_load(R, dpend, l_ + 4 >= nL, base, sub, ci, val);                                                   \
    dpend = dn_;                                                                                                \
    __syncthreads();                                                                                            \
  }
  for (int l = 0; l < nlev; l += 4) {   // the conditions are uniform over the workgroup
    ISPH_SUB_STEP(r0, l)
    if (l + 1 < nlev) ISPH_SUB_STEP(r1, l + 1)
    if (l + 2 < nlev) ISPH_SUB_STEP(r2, l + 2)
    if (l + 3 < nlev) ISPH_SUB_STEP(r3, l + 3)
  }
#undef ISPH_SUB_STEP
  for (int t = tid; t < m; t += blockDim.x) w[base + t] = sub_y[t];
}

// ---- many small subdomains: the row lists and the combine lists on the device -------------------------------------------------
// Subdomain s owns the consecutive rows s B .. s B + B; one overlap layer adds the columns its rows reference outside that
// range (Ifpack_OverlappingRowMatrix on a row partition), ascending.  One workgroup per subdomain: the referenced columns
// go through an LDS hash set (the owned rows are one contiguous piece of the CSR image: a flat, coalesced walk), the set is
// compacted and sorted in LDS (bitonic) and parked at a fixed stride; after the prefix sum over the sizes
// k_gilu_sub_rows_fill writes the lists where they belong.  A subdomain that outgrows the one-workgroup form
// (kSubSweepMaxRows) raises flag bit 0: the caller takes the host path then.
constexpr int kSubHash = 8192;
__global__ __launch_bounds__(256) void k_gilu_sub_rows(int nsub, int B, int n, int overlap, const long long *__restrict__ arp,
                                                       const int *__restrict__ aci, int stride, int *__restrict__ ext,
                                                       int *__restrict__ msize, int *__restrict__ flag) {
  __shared__ int tab[kSubHash];
  __shared__ int lst[kSubSweepMaxRows];
  __shared__ int s_cnt, s_k, s_bad;
  const int s = blockIdx.x, tid = threadIdx.x;
  const int lo = s * B, hi = min(n, lo + B), no = hi - lo;
  for (int t = tid; t < kSubHash; t += 256) tab[t] = -1;
  if (tid == 0) { s_cnt = 0; s_k = 0; s_bad = no > stride ? 1 : 0; }
  __syncthreads();
  const int cap = stride - no;   // overlap rows the one-workgroup form has room for
  if (overlap > 0 && nsub > 1 && no <= stride) {
    const long long b = arp[lo], e = arp[hi];
    for (long long p = b + tid; p < e; p += 256) {
      if (*(volatile int *)&s_bad) break;
      const int c = aci[p];
      if (c >= n || (c >= lo && c < hi)) continue;
      unsigned h = ((unsigned)c * 2654435761u) >> 19;   // 13 bits
      for (int probe = 0; probe < kSubHash; ++probe) {
        const int old = atomicCAS(&tab[h], -1, c);
        if (old == -1) { if (atomicAdd(&s_cnt, 1) >= cap) s_bad = 1; break; }
        if (old == c) break;
        h = (h + 1) & (kSubHash - 1);
      }
    }
  }
  __syncthreads();
  const int cnt = s_cnt;
  if (s_bad) {   // uniform
    if (tid == 0) { atomicOr(flag, 1); msize[s] = 0; }
    return;
  }
  for (int t = tid; t < kSubHash; t += 256) {
    const int c = tab[t];
    if (c >= 0) lst[atomicAdd(&s_k, 1)] = c;
  }
  int np = 1;
  while (np < cnt) np <<= 1;
  __syncthreads();
  for (int t = cnt + tid; t < np; t += 256) lst[t] = INT_MAX;
  __syncthreads();
  for (int k = 2; k <= np; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < np; t += 256) {
        const int u = t ^ j;
        if (u > t) {
          const int a = lst[t], c = lst[u];
          if ((a > c) == ((t & k) == 0)) { lst[t] = c; lst[u] = a; }
        }
      }
      __syncthreads();
    }
  for (int t = tid; t < cnt; t += 256) ext[(long long)s * stride + t] = lst[t];
  if (tid == 0) msize[s] = no + cnt;
}

__global__ __launch_bounds__(256) void k_gilu_sub_rows_fill(int nsub, int B, int n, int stride, const long long *__restrict__ lp64,
                                                            const int *__restrict__ ext, int *__restrict__ loc_ptr,
                                                            int *__restrict__ rows) {
  const int s = blockIdx.x, tid = threadIdx.x;
  const int base = (int)lp64[s], m = (int)lp64[s + 1] - base, lo = s * B, no = min(n, lo + B) - lo;
  if (tid == 0) {
    loc_ptr[s] = base;
    if (s == nsub - 1) loc_ptr[nsub] = (int)lp64[nsub];
  }
  for (int q = tid; q < m; q += 256) rows[base + q] = q < no ? lo + q : ext[(long long)s * stride + (q - no)];
}

// combine lists (global row -> local rows).  Zero: the owned copy only, a closed form.  Add: every copy, in ascending local
// row (= subdomain order, the host's order: the sum in k_gilu_combine is bitwise the same) -- counted and placed with
// atomics, then each row's short list sorted.
__global__ void k_gilu_rev_zero(int n, int B, const int *__restrict__ loc_ptr, long long *__restrict__ rev_ptr, int *__restrict__ rev_idx) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g > n) return;
  rev_ptr[g] = g;
  if (g < n) rev_idx[g] = loc_ptr[g / B] + g % B;
}
__global__ void k_gilu_rev_count(int nloc, const int *__restrict__ rows, int *__restrict__ cnt) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < nloc) atomicAdd(&cnt[rows[q]], 1);
}
__global__ void k_gilu_rev_fill(int nloc, const int *__restrict__ rows, const long long *__restrict__ rev_ptr, int *__restrict__ cur,
                                int *__restrict__ rev_idx) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nloc) return;
  const int g = rows[q];
  rev_idx[rev_ptr[g] + atomicAdd(&cur[g], 1)] = q;
}
__global__ void k_gilu_rev_sort(int n, const long long *__restrict__ rev_ptr, int *__restrict__ rev_idx) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const long long b = rev_ptr[g], e = rev_ptr[g + 1];
  for (long long p = b + 1; p < e; ++p) {
    const int v = rev_idx[p];
    long long t = p;
    for (; t > b && rev_idx[t - 1] > v; --t) rev_idx[t] = rev_idx[t - 1];
    rev_idx[t] = v;
  }
}

// ---- many small subdomains: the local matrices on the device (level of fill 0, at most one overlap layer) ------------------
// Ifpack_LocalFilter per subdomain: local row q = global row rows[q]; its entries whose column is a row of the same
// subdomain are kept and renumbered -- owned columns (a consecutive global range) first, then the columns of the overlap
// layer (ascending global row, found by bisection in the subdomain's row list), both in their original order, which is the
// ascending local order.  One wave per local row; pass 1 counts (kept entries, owned among them), pass 2 writes columns,
// values and the position of the diagonal.  Replaces 0.14 s of host threads + 0.07 s of upload at 10^6 rows.
template <bool FILL>
__global__ __launch_bounds__(256) void k_gilu_sub_local(int nloc, int nsub, int n, const int *__restrict__ loc_ptr,
                                                        const int *__restrict__ nown, const int *__restrict__ rows,
                                                        const long long *__restrict__ arp, const int *__restrict__ aci,
                                                        const double *__restrict__ aval, int *__restrict__ cnt,
                                                        int *__restrict__ cown, const long long *__restrict__ lrp,
                                                        int *__restrict__ lci, double *__restrict__ lval, int *__restrict__ dg,
                                                        int *__restrict__ meta /* [0] longest row, [1] bit 0: missing diagonal */) {
  const int q = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (q >= nloc) return;
  int lo = 0, hi = nsub;   // subdomain of local row q: last s with loc_ptr[s] <= q
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (loc_ptr[mid] <= q) lo = mid; else hi = mid; }
  const int s = lo, base = loc_ptr[s], m = loc_ptr[s + 1] - base, no = nown[s];
  const int g = rows[q], g0 = rows[base];           // owned rows are the consecutive global range g0 .. g0 + no
  const int *ext = rows + base + no;                // overlap layer, ascending
  const int next = m - no;
  const long long b = arp[g], e = arp[g + 1];
  const unsigned long long below = (1ull << lane) - 1ull;
  int nown_seen = 0, next_seen = 0, dpos = -1;
  const int own_total = FILL ? cown[q] : 0;
  const long long out0 = FILL ? lrp[q] : 0;
  for (long long p0 = b; p0 < e; p0 += 64) {
    const long long p = p0 + lane;
    int li = -1;       // local column, -1: dropped
    bool owned = false;
    if (p < e) {
      const int c = aci[p];
      if (c >= g0 && c < g0 + no) { li = base + (c - g0); owned = true; }
      else if (c < n && next > 0) {
        int l2 = 0, h2 = next - 1;
        while (l2 <= h2) {
          const int mid = (l2 + h2) >> 1;
          const int r = ext[mid];
          if (r == c) { li = base + no + mid; break; }
          if (r < c) l2 = mid + 1; else h2 = mid - 1;
        }
      }
    }
    const unsigned long long mo = __ballot(li >= 0 && owned), me = __ballot(li >= 0 && !owned);
    if (FILL && li >= 0) {
      const int pos = owned ? nown_seen + __popcll(mo & below) : own_total + next_seen + __popcll(me & below);
      lci[out0 + pos] = li;
      lval[out0 + pos] = aval[p];
      if (li == q) dpos = pos;
    }
    nown_seen += __popcll(mo);
    next_seen += __popcll(me);
  }
  if (!FILL) {
    if (lane == 0) {
      cnt[q] = nown_seen + next_seen; cown[q] = nown_seen;
      // (an atomic of every wave on the one word took 41 of the pass' 49 ms at 4.3 M rows)
      if (nown_seen + next_seen > *(volatile int *)&meta[0]) atomicMax(&meta[0], nown_seen + next_seen);
    }
  } else {
    dpos = wave_max_i32(dpos);
    if (lane == 0) { dg[q] = dpos; if (dpos < 0) atomicOr(&meta[1], 1); }
  }
}

// ---- many small subdomains: level analysis and numeric factorisation on the device ------------------------------------
// k_gilu_sub_levels: one workgroup (two waves) per subdomain.  Wave 0 walks the rows upwards for the L levels, wave 1
// downwards for the U levels (lev = 1 + max over the dependencies, which all lie in the same subdomain), lanes over a
// row's dependencies, the columns of the next eight rows already requested; then both directions are sorted by level
// (ascending row inside a level) straight into the descriptor list of k_gilu_solve_sub: L levels 1 .., then U levels 0 ..
// Replaces the host recurrences + orders of schwarz_create for this form.
// Layout (upper bounds, no prefix sums): subdomain s with m rows at base owns descriptor slots 2 base .. 2 base + 2 m and
// level starts 2 base + 2 s .. + 2 m + 2; nlist[2 s] = listed L levels, nlist[2 s + 1] = all listed levels;
// maxlev[0 / 1] = deepest L / U level count over all subdomains.
__global__ __launch_bounds__(128) void k_gilu_sub_levels(const int *__restrict__ loc_ptr, const long long *__restrict__ rp,
                                                         const int *__restrict__ ci, const int *__restrict__ dg,
                                                         long long *__restrict__ desc, int *__restrict__ lptr,
                                                         int *__restrict__ nlist, int *__restrict__ maxlev) {
  extern __shared__ int sub_lds[];
  const int s = blockIdx.x, base = loc_ptr[s], m = loc_ptr[s + 1] - base;
  int *levL = sub_lds, *levU = levL + m, *cnt = levU + m;   // cnt: [m + 2]
  __shared__ int s_n[2];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid < 2) s_n[tid] = 0;
  {
    constexpr int kAhead = 8;
    int *lv = wave == 0 ? levL : levU;
    const int step = wave == 0 ? 1 : -1, r0 = wave == 0 ? 0 : m - 1;
    int cq[kAhead], nq[kAhead];
    long long dq[kAhead];
    auto request = [&](int k, int &c, int &nd, long long &d0) {
      c = -1; nd = 0; d0 = 0;
      if (k < m) {
        const int q = base + r0 + k * step;
        const long long b = rp[q];
        const int kd = dg[q];
        d0 = wave == 0 ? b : b + kd + 1;
        nd = wave == 0 ? kd : (int)(rp[q + 1] - d0);
        if (lane < nd) c = ci[d0 + lane];
      }
    };
#pragma unroll
    for (int u = 0; u < kAhead; ++u) request(u, cq[u], nq[u], dq[u]);
    int deepest = 0;
    for (int k0 = 0; k0 < m; k0 += kAhead) {
#pragma unroll
      for (int u = 0; u < kAhead; ++u) {
        const int k = k0 + u;
        if (k < m) {
          int nl = cq[u] >= 0 ? lv[cq[u] - base] + 1 : 0;
          for (int e = lane + 64; e < nq[u]; e += 64) nl = max(nl, lv[ci[dq[u] + e] - base] + 1);
          nl = wave_max_i32(nl);
          if (lane == 0) lv[r0 + k * step] = nl;
          deepest = max(deepest, nl + 1);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          request(k + kAhead, cq[u], nq[u], dq[u]);
        }
      }
    }
    if (lane == 0) s_n[wave] = deepest;
  }
  __syncthreads();
  const int nl = s_n[0], nu = s_n[1];
  if (tid == 0) { atomicMax(&maxlev[0], nl); atomicMax(&maxlev[1], nu); }
  long long *D = desc + 2 * (2 * (long long)base);
  int *lp = lptr + 2 * base + 2 * s;
  int pos_base = 0, lev_base = 0;
  for (int dir = 0; dir < 2; ++dir) {
    const int *lev = dir == 0 ? levL : levU;
    const int nlv = dir == 0 ? nl : nu, first = dir == 0 ? 1 : 0;   // L level 0 has nothing to subtract: not listed
    for (int t = tid; t <= nlv + 1; t += blockDim.x) cnt[t] = 0;
    __syncthreads();
    for (int t = tid; t < m; t += blockDim.x) atomicAdd(&cnt[lev[t] + 1], 1);
    __syncthreads();
    if (tid == 0)
      for (int l = 0; l < nlv; ++l) cnt[l + 1] += cnt[l];   // cnt[l] = rows of levels < l
    __syncthreads();
    const int skip = first ? cnt[1] : 0;   // rows of L level 0
    for (int l = first + tid; l < nlv; l += blockDim.x) {
      int pos = pos_base + cnt[l] - skip;
      lp[lev_base + l - first] = pos;
      for (int r = 0; r < m; ++r)
        if (lev[r] == l) {
          const int i = base + r;
          const long long b = rp[i];
          const int kd = dg[i];
          const long long start = dir == 0 ? b : b + kd + 1;
          const int nn = dir == 0 ? kd : (int)(rp[i + 1] - start);
          D[2 * (long long)pos] = start;
          D[2 * (long long)pos + 1] = ((long long)i << 32) | (unsigned int)nn;
          ++pos;
        }
    }
    __syncthreads();
    pos_base += m - skip;
    lev_base += nlv - first > 0 ? nlv - first : 0;
    if (dir == 0 && tid == 0) nlist[2 * s] = lev_base;
    __syncthreads();
  }
  if (tid == 0) { lp[lev_base] = pos_base; nlist[2 * s + 1] = lev_base; }
}

// k_gilu_sub_factor: IKJ ILU(k) numeric factorisation of one subdomain per workgroup, level by level over the L list of
// k_gilu_sub_levels (level 0 rows have no lower part), a row per wave, one barrier per level.  Arithmetic of a row = that
// of k_gilu_factor, operation for operation (same bits); what differs is how the operands arrive: the column -> slot
// look-up is a table in LDS (the subdomain has at most 4096 columns) instead of a binary search, what a step needs to know
// about its pivot row is looked up once per row, one pivot per lane, and handed over by v_readlane, and the pivot rows'
// upper parts are requested four steps ahead (cf. k_ilu_factor, ilu.hpp).
// LDS per wave: row image (values + columns, 12 B per entry) + the table (2 B per row of the subdomain).
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_gilu_sub_factor(const int *__restrict__ loc_ptr, const long long *__restrict__ desc,
                                                                const int *__restrict__ lptr, const int *__restrict__ nlist,
                                                                const long long *__restrict__ rp, const int *__restrict__ ci,
                                                                const int *__restrict__ dg, double *__restrict__ val,
                                                                int maxrow, int msub, int *__restrict__ err) {
  extern __shared__ double subf_lds[];
  const int s = blockIdx.x, base = loc_ptr[s], m = loc_ptr[s + 1] - base;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double *w = subf_lds + (size_t)wave * maxrow;
  int *cols = reinterpret_cast<int *>(subf_lds + (size_t)WAVES * maxrow) + (size_t)wave * maxrow;
  unsigned short *mp = reinterpret_cast<unsigned short *>(reinterpret_cast<int *>(subf_lds + (size_t)WAVES * maxrow) + (size_t)WAVES * maxrow) +
                       (size_t)wave * msub;   // slot + 1 of a column of the subdomain in the row at hand (0: not in its pattern)
  for (int t = lane; t < msub; t += 64) mp[t] = 0;
  const long long *D = desc + 2 * (2 * (long long)base);
  const int *lp = lptr + 2 * base + 2 * s;
  const int nL = nlist[2 * s];
  __syncthreads();
  constexpr int kPF = 4;   // 8 measured the same (58.9 / 58.2 ms at 100^3): the launch is bound by L2 misses, see below
  for (int l = 0; l < nL; ++l) {
    for (int q = lp[l] + wave; q < lp[l + 1]; q += WAVES) {
      const int i = (int)(D[2 * (long long)q + 1] >> 32);
      const long long b = rp[i];
      const int len = (int)(rp[i + 1] - b), nlow = dg[i];
      for (int t = lane; t < len; t += 64) {
        const int c = ci[b + t];
        w[t] = val[b + t];
        cols[t] = c;
        mp[c - base] = (unsigned short)(t + 1);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // per pivot (one per lane, the first 64 of the row): start and length of the pivot row's upper part, its pivot
      long long pstart = 0;
      int pcount = 0;
      double ppiv = 1.0;
      if (lane < nlow) {
        const int k = cols[lane];
        const long long kb = rp[k];
        const int kd = dg[k];
        pstart = kb + kd + 1;
        pcount = (int)(rp[k + 1] - pstart);
        ppiv = val[kb + kd];
      }
      int pcq[kPF], pnq[kPF];
      double pvq[kPF], pdq[kPF];
      long long psq[kPF];
      auto request = [&](int t, int &pc, double &pv, double &pd, int &pn, long long &ps0) {
        pc = -1; pv = 0.0; pd = 1.0; pn = 0; ps0 = 0;
        if (t < nlow) {
          if (t < 64) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)pstart, t);
            const int hi = __builtin_amdgcn_readlane((int)(pstart >> 32), t);
            ps0 = (long long)(((unsigned long long)(unsigned)hi << 32) | lo);
            pn = __builtin_amdgcn_readlane(pcount, t);
            pd = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ppiv), t), __builtin_amdgcn_readlane(__double2loint(ppiv), t));
          } else {
            const int k = cols[t];
            const long long kb = rp[k];
            const int kd = dg[k];
            ps0 = kb + kd + 1;
            pn = (int)(rp[k + 1] - ps0);
            pd = val[kb + kd];
          }
          if (lane < pn) { pc = ci[ps0 + lane]; pv = val[ps0 + lane]; }
        }
      };
#pragma unroll
      for (int u = 0; u < kPF; ++u) request(u, pcq[u], pvq[u], pdq[u], pnq[u], psq[u]);
      for (int t0 = 0; t0 < nlow; t0 += kPF) {
#pragma unroll
        for (int u = 0; u < kPF; ++u) {
          const int t = t0 + u;
          if (t < nlow) {
            const int ps = pcq[u] >= 0 ? mp[pcq[u] - base] : 0;
            const double lik = w[t] / pdq[u];
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) w[t] = lik;
            if (ps) w[ps - 1] -= lik * pvq[u];
            for (int e = lane + 64; e < pnq[u]; e += 64) {   // pivot rows with more than 64 upper entries
              const int p2 = mp[ci[psq[u] + e] - base];
              if (p2) w[p2 - 1] -= lik * val[psq[u] + e];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            request(t + kPF, pcq[u], pvq[u], pdq[u], pnq[u], psq[u]);
          }
        }
      }
      for (int t = lane; t < len; t += 64) {
        val[b + t] = w[t];
        mp[cols[t] - base] = 0;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();   // level l is in memory before level l + 1 reads its rows
  }
  for (int t = threadIdx.x; t < m; t += blockDim.x) {   // zero pivots, level 0 rows included
    const int q = base + t;
    if (!(fabs(val[rp[q] + dg[q]]) > 0.0)) atomicOr(err, 2);
  }
}

// ---- synchronisation-free sweeps ----------------------------------------------------------------------------------
// One launch per triangular sweep instead of one per dependency level (the whole-matrix factor of the 100^3 system has
// 67 084 L levels: 134 168 launches per application before).  Persistent workgroups take RUNS of the level order from
// one device-wide counter (a position's dependencies sit at earlier positions, and a position is only ever held by a
// running workgroup, so the sweep cannot deadlock whatever the residency of the grid), 16 lanes per row as before.  A
// result IS its own ready flag: the output vector starts as a signalling-NaN pattern no arithmetic produces, every
// result is ONE 8-byte write-through (sc1) store, and a consumer polls the very word it needs with sc1 loads -- no
// flag, no fence (MI355X_MICROARCH.md, hand-off price list, "granule").  Every access to the shared words is an
// agent-scope atomic on a global-address-space pointer; nothing else touches them in the launch.
// A spin that sees no progress for kSpinLimit polls sets ctr[3] and stores 0 -- every wave drains, the host fails the
// application loudly.
typedef __attribute__((address_space(1))) unsigned long long gu64_t;
typedef __attribute__((address_space(1))) int gi32_t;
constexpr unsigned long long kGiluSentinel = 0xFFF4A5A5DEADBEEFull;  // signalling NaN, payload of our own
constexpr int kSpinLimit = 1 << 20;  // polls of one word (~0.2 us each) before a wait gives up

__device__ __forceinline__ unsigned long long sf_load(const unsigned long long *p) {
  return __hip_atomic_load((const gu64_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sf_store(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store((gu64_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int sf_flag(const int *p) {
  return __hip_atomic_load((const gi32_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// [n] of both: a word that is always there (0.0), see k_gilu_solve_run
__global__ void k_gilu_fill_bits(int n, unsigned long long *__restrict__ a, unsigned long long *__restrict__ b) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x) {
    a[i] = i < n ? kGiluSentinel : 0ull;
    b[i] = i < n ? kGiluSentinel : 0ull;
  }
}

// A lane's share of a row is handled in chunks of kSfChunk entries (16 lanes x 6 = 96 entries: a whole ILU(0) row of
// the SPH operators).  A workgroup of 16 waves takes a RUN of up to 64 consecutive positions of the padded level order,
// whole levels (three or four of the SPH factors), four positions = rows of one level per wave; a dependency
// whose position lies inside the run is read from the run's LDS image of the results, the others from global memory
// with sc1 loads.  Only the first level of a run waits for a global hand-off (0.5 us for the bare exchange between two
// workgroups, scripts/lds_handoff_probe.hip); inside the run a hand-off is an LDS write and read (0.08 us).
// At those latencies the instructions between "last dependency seen" and "result stored" are what is left, so the row
// does everything it can before it waits: the products with the dependencies outside the run are summed as soon as
// those are there (they belong to earlier levels), over all chunks of the row; the ones inside the run are kept as a
// list of kNearCap (value, slot) pairs per lane -- a run holds 63 earlier positions, a lane a sixteenth of the row:
// more is rare, and goes through a slower loop -- and what remains after the last word arrives is kNearCap
// multiply-adds, the 16-lane sum and the store.  (With the wait for the outside words inside the chunk loop, as first
// written, every chunk of a long row put a round trip to memory on the chain: 3.07 us per level on the ILU(1) factor
// of the 100^3 system, whose rows have two to four chunks; 0.73 us now.)  Summation order per lane: outside-the-run terms by
// entry, then inside-the-run terms by entry; fixed by the factor's pattern and the run table, so the result is
// reproducible bit for bit from application to application, and differs from the level-launch kernels' by rounding
// (theirs divide by the pivot, the U sweep here multiplies with its reciprocal, taken before the row waits).
// Deadlock-free for the same reason as above: runs are dequeued in order, a row waits for earlier positions only,
// and all waves of a workgroup are resident together.
constexpr int kSfChunk = 6;
constexpr int kRun = 64;
constexpr int kNearCap = 6;   // (value, slot) pairs a lane keeps for the dependencies inside the run
constexpr int kFarCap = 6;    // (value, row) pairs a lane keeps for the recent dependencies outside it
constexpr int kLongWindow = 4;       // throttle window of the factors with long rows: their rows need longer to get ready
constexpr int kRecent = 2 * kRun;    // positions before the run that count as recent (the two runs before it; what the wider
                                     // window lets be open beyond that is so far back that it is there when asked for:
                                     // 256 positions and a list of 10 measured slower)
template <bool UPPER, bool SPLIT>
__global__ __launch_bounds__(1024) void k_gilu_solve_run(int nruns, const int *__restrict__ runstart,
                                                         const int *__restrict__ order4,
                                                         const int *__restrict__ rowpos4,
                                                         const long long *__restrict__ rp, const int *__restrict__ ci,
                                                         const int *__restrict__ dg, const double *__restrict__ val,
                                                         const double *__restrict__ rhs, unsigned long long *out,
                                                         int ready, int *ctr, int knear) {
  // [kRun] of an image is a word that is always there (0.0): the target of a lane's unused list entries
  __shared__ unsigned long long s_out[2][kRun + 1];
  __shared__ int s_base[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane & 15, grp = lane >> 4;
  int *head = ctr + (UPPER ? 1 : 0), *tmo = ctr + 3, *done = ctr + (UPPER ? 5 : 4);
  bool had_run = false;
  for (int par = 0;; par ^= 1) {
    __builtin_amdgcn_s_setprio(0);
    if (had_run) {
      __syncthreads();   // every row of the previous run is stored
      if (tid == 0) __hip_atomic_fetch_add((gi32_t *)done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) s_base[par] = __hip_atomic_fetch_add((gi32_t *)head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid <= kRun) s_out[par][tid] = tid < kRun ? kGiluSentinel : 0ull;
    __syncthreads();   // the other image is free again once every wave has passed this barrier
    const int run = s_base[par];
    if (run >= nruns) break;
    had_run = true;
    const int base = runstart[run], len = runstart[run + 1] - base;
    const int pos = base + 4 * wave + grp;
    const int i = pos - base < len ? order4[pos] : -1;
    double s = 0.0, d = 1.0, ri = 0.0;
    bool timed_out = sf_flag(tmo) != 0;
    if (i >= 0) {
      const long long b = rp[i];
      const int kd = dg[i];
      const long long first = UPPER ? b + kd + 1 : b;   // the chunk loop is uniform over the row's 16 lanes
      const long long last = UPPER ? rp[i + 1] : b + kd;
      ri = rhs[i];
      if (UPPER) d = 1.0 / val[b + kd];   // the division happens before the row waits, not after
      // phase 0: far from the front of the sweep nothing this run waits for can be there; sleep until all but a few
      // of the runs before it are complete instead of asking for every missing word again and again
      if (knear > 0) {
        int sp = 0;
        while (sf_flag(done) < run - knear && !timed_out) {
          __builtin_amdgcn_s_sleep(32);
          if (++sp >= kSpinLimit || sf_flag(tmo)) timed_out = true;
        }
      }
      // phase 1, over all chunks of the row (one for ILU(0), up to ~six for ILU(2)).  Three kinds of dependency:
      //  - inside the run: on the lane's list of (value, slot) pairs, waited for in LDS (phase 2);
      //  - outside the run but within the last kRecent positions before it (the two runs before this one, which the
      //    throttle above lets be open): on a second list of (value, row) pairs, waited for in global memory AFTER the
      //    chunk loop -- a long row has several chunks, and with the wait inside the loop every chunk after the one that
      //    held the late word put its loads (three dependent round trips) between that word and the row's store: 4.1 us
      //    per hand-off between runs on the ILU(1) factor against 1.4 us for ILU(0)'s one-chunk rows;
      //  - older: there when the throttle opens (checked all the same), summed inside the loop.
      double nw[kNearCap], rw[kFarCap];
      int nslot[kNearCap], rc[kFarCap], nn = 0, nr = 0;
      // (SPLIT = false, factors whose rows all fit one chunk: nothing stands behind a row's one wait, its recent words
      // stay in the chunk's round of requests and the second list is not used)
      constexpr int lo_recent = SPLIT ? -kRecent : 0;
#pragma unroll
      for (int j = 0; j < kNearCap; ++j) { nw[j] = 0.0; nslot[j] = kRun; }
#pragma unroll
      for (int j = 0; j < kFarCap; ++j) { rw[j] = 0.0; rc[j] = ready; }
      for (long long q0 = first; q0 < last; q0 += 16 * kSfChunk) {
        double vf[kSfChunk];
        int c[kSfChunk];
        unsigned long long x[kSfChunk];
#pragma unroll
        for (int k = 0; k < kSfChunk; ++k) {
          const long long q = q0 + sub + 16 * k;
          const bool ok = q < last;
          vf[k] = ok ? val[q] : 0.0;
          c[k] = ok ? ci[q] : -1;
        }
#pragma unroll
        for (int k = 0; k < kSfChunk; ++k) {
          const int sl = c[k] >= 0 ? rowpos4[c[k]] - base : INT_MIN;   // >= 0: inside this run
          if (sl >= 0) {
#pragma unroll
            for (int j = 0; j < kNearCap; ++j)
              if (nn == j) { nslot[j] = sl; nw[j] = vf[k]; }
            ++nn;   // past kNearCap: phase 2 walks the row again for it
          } else if (sl >= lo_recent) {
#pragma unroll
            for (int j = 0; j < kFarCap; ++j)
              if (nr == j) { rc[j] = c[k]; rw[j] = vf[k]; }
            ++nr;   // past kFarCap: walked again below
          }
          if (sl >= lo_recent || c[k] < 0) { c[k] = ready; vf[k] = 0.0; }   // listed / no entry: the word that is always there, times 0
        }
        int spins = 0;
        for (;;) {   // all six words of the lane per round
          bool pending = false;
#pragma unroll
          for (int k = 0; k < kSfChunk; ++k) x[k] = sf_load(out + c[k]);
#pragma unroll
          for (int k = 0; k < kSfChunk; ++k) pending = pending || x[k] == kGiluSentinel;
          if (!pending || timed_out) break;
          __builtin_amdgcn_s_sleep(1);
          if (++spins >= kSpinLimit || ((spins & 255) == 0 && sf_flag(tmo))) timed_out = true;
        }
#pragma unroll
        for (int k = 0; k < kSfChunk; ++k) s += vf[k] * (timed_out ? 0.0 : __longlong_as_double((long long)x[k]));
      }
      if (SPLIT) {   // the recent words outside the run: all of the lane's list per round
        unsigned long long y[kFarCap];
        int spins = 0;
        for (;;) {
          bool pending = false;
#pragma unroll
          for (int j = 0; j < kFarCap; ++j) y[j] = sf_load(out + rc[j]);
#pragma unroll
          for (int j = 0; j < kFarCap; ++j) pending = pending || y[j] == kGiluSentinel;
          if (!pending || timed_out) break;
          __builtin_amdgcn_s_sleep(1);
          if (++spins >= kSpinLimit || ((spins & 255) == 0 && sf_flag(tmo))) timed_out = true;
        }
#pragma unroll
        for (int j = 0; j < kFarCap; ++j) s += rw[j] * (timed_out ? 0.0 : __longlong_as_double((long long)y[j]));
      }
      if (SPLIT && nr > kFarCap) {   // more recent words than the list holds: one after the other
        int seen = 0;
        for (long long q = first + sub; q < last; q += 16) {
          const int cq = ci[q];
          const int sl = rowpos4[cq] - base;
          if (sl >= 0 || sl < lo_recent || seen++ < kFarCap) continue;
          unsigned long long y;
          int spins = 0;
          while ((y = sf_load(out + cq)) == kGiluSentinel && !timed_out) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins >= kSpinLimit || ((spins & 255) == 0 && sf_flag(tmo))) timed_out = true;
          }
          s += val[q] * (timed_out ? 0.0 : __longlong_as_double((long long)y));
        }
      }
      // phase 2: the dependencies inside the run, from its LDS image: the first two of the list (all there is, for most
      // lanes of an ILU(0) row), then the rest of it.  A wave of the run's first level has no such word at all.
      if (__ballot(nn > 0) != 0ull) {
        unsigned long long y[kNearCap];
        int spins = 0;
        for (;;) {
          y[0] = __hip_atomic_load(&s_out[par][nslot[0]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          y[1] = __hip_atomic_load(&s_out[par][nslot[1]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if ((y[0] != kGiluSentinel && y[1] != kGiluSentinel) || timed_out) break;
          if ((++spins & 4095) == 0 && (sf_flag(tmo) || spins >= 64 * kSpinLimit)) timed_out = true;
        }
        s += nw[0] * (timed_out ? 0.0 : __longlong_as_double((long long)y[0]));
        s += nw[1] * (timed_out ? 0.0 : __longlong_as_double((long long)y[1]));
        if (nn > 2) {
          spins = 0;
          for (;;) {
            bool pending = false;
#pragma unroll
            for (int j = 2; j < kNearCap; ++j) y[j] = __hip_atomic_load(&s_out[par][nslot[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int j = 2; j < kNearCap; ++j) pending = pending || y[j] == kGiluSentinel;
            if (!pending || timed_out) break;
            if ((++spins & 4095) == 0 && (sf_flag(tmo) || spins >= 64 * kSpinLimit)) timed_out = true;
          }
#pragma unroll
          for (int j = 2; j < kNearCap; ++j) s += nw[j] * (timed_out ? 0.0 : __longlong_as_double((long long)y[j]));
        }
      }
      if (nn > kNearCap) {   // more dependencies of this lane inside the run than its list holds: one after the other
        int seen = 0;
        for (long long q = first + sub; q < last; q += 16) {
          const int sl = rowpos4[ci[q]] - base;
          if (sl < 0 || seen++ < kNearCap) continue;
          unsigned long long y;
          int spins = 0;
          while ((y = __hip_atomic_load(&s_out[par][sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == kGiluSentinel && !timed_out)
            if ((++spins & 4095) == 0 && (sf_flag(tmo) || spins >= 64 * kSpinLimit)) timed_out = true;
          s += val[q] * (timed_out ? 0.0 : __longlong_as_double((long long)y));
        }
      }
    }
    __builtin_amdgcn_s_setprio(3);   // the wave that holds the chain goes first on its SIMD
    s = group16_sum(s);
    if (__ballot(timed_out) != 0ull) {
      if (lane == 0) __hip_atomic_store((gi32_t *)tmo, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s = 0.0;
    }
    if (i >= 0 && sub == 0) {
      const double r = UPPER ? (ri - s) * d : ri - s;
      unsigned long long bits = (unsigned long long)__double_as_longlong(r);
      if (bits == kGiluSentinel) bits ^= 1ull;
      __hip_atomic_store(&s_out[par][pos - base], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      sf_store(out + i, bits);
    }
  }
}

// numeric factorisation, one wave per row, rows taken in level order from ctr[2].  The factor is written to a second
// array that starts as the signalling-NaN pattern: as in the sweeps a value is its own ready flag -- a finished row is
// published by plain write-through stores, a consumer polls the very words of the pivot row it needs (the pivot and the
// entries of its upper part that meet this row's pattern) with sc1 loads.  No flag per row, no wait for the stores to be
// acknowledged before a flag may follow them; the input values are read-only in the launch and are read with ordinary
// loads.  Everything about a pivot step that depends on the PATTERN only -- the pivot row's pointers, its columns, the
// slot of each of them in this row -- is done before the step asks for values, so that between "pivot row visible" and
// "this row stored" stand one division, one multiply-subtract per lane and two barriers.  (The first version of this
// launch polled a flag per row, then loaded the pivot row's pointers, its pivot, its columns and values one after the
// other and waited for its own stores before raising its flag: five round trips to memory per level, 7.2 us measured
// on the 100^3 factor.)  WAVES = 4 for factors whose pivot rows have more than 64 entries right of the
// diagonal on average (ILU(k > 0)): a thread per entry, so that a step stays one round of requests (ILU(1) of the 100^3
// system: 12.8 -> 5.5 us per level).  Same operations in the same
// order as k_gilu_factor: same bits.
__global__ void k_gilu_fill_sentinel(long long n, unsigned long long *__restrict__ a) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) a[i] = kGiluSentinel;
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_gilu_factor_sf(int nloc, const int *__restrict__ order,
                                                               const long long *__restrict__ rp, const int *__restrict__ ci,
                                                               const int *__restrict__ dg, const double *__restrict__ ain,
                                                               unsigned long long *fout, int *ctr, int *__restrict__ err, int hbits) {
  extern __shared__ double gilu_lds[];
  __shared__ int s_pos, s_dead;
  constexpr int T = 64 * WAVES;   // threads per row: one per entry of a pivot row's upper part and round
  const int lane = threadIdx.x;
  int *tmo = ctr + 3;
  for (;;) {
    __syncthreads();  // the previous row's image and position are no longer read
    if (lane == 0) s_pos = __hip_atomic_fetch_add((gi32_t *)(ctr + 2), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // through LDS: `if (lane == 0) p = add(); p = readfirstlane(p)` was compiled into a loop nest in which lane 0 had
    // left the active set at the readfirstlane, which then returned another lane's 0
    __syncthreads();
    const int pos = s_pos;
    if (pos >= nloc) break;
    const int i = order[pos];
    const long long b = rp[i];
    const int len = (int)(rp[i + 1] - b), nlow = dg[i];
    double *w = gilu_lds;
    int *cols = reinterpret_cast<int *>(gilu_lds + len);
    // column -> position table of this row (open addressing, at most half full): a look-up is one or two LDS reads
    // instead of the seven dependent ones of a binary search.  hbits = 0: the table does not fit, binary search.
    int *hkey = cols + len + (len & 1);
    unsigned short *hpos = reinterpret_cast<unsigned short *>(hkey + (hbits ? (1 << hbits) : 0));
    const int hmask = (1 << hbits) - 1;
    for (int t = lane; t < len; t += T) {
      w[t] = ain[b + t];
      cols[t] = ci[b + t];
    }
    if (hbits)
      for (int t = lane; t <= hmask; t += T) hkey[t] = -1;
    if (lane == 0) s_dead = sf_flag(tmo);  // the launch has given up: drain without waiting
    __syncthreads();
    if (hbits) {
      for (int t = lane; t < len; t += T) {
        const int key = cols[t];
        int h = (int)(((unsigned)key * 2654435761u) >> (32 - hbits));
        while (atomicCAS(&hkey[h], -1, key) != -1) h = (h + 1) & hmask;
        hpos[h] = (unsigned short)t;
      }
      __syncthreads();
    }
    bool dead = s_dead != 0;
    // A pivot step needs: the pivot row's pointers, then its columns and its values (their addresses do not depend on
    // each other), then the slot look-up in LDS for the column (seven dependent LDS reads, 0.6 us), then the update.
    // All the loads go beyond the L2 (0.4 + 0.8 GB of factor per 10^6 rows against 4 MB per XCD).  A row's pivots lie
    // on consecutive levels (34 distinct levels for the 51 pivots of an ILU(0) row), so they arrive one level apart and
    // the row is never ahead of them: the time of ONE step is what a level costs.  Per-row time stamps (32^3 factor)
    // gave 0.7 us to see a value that is there + 1.2 us to get from there to the next step's wait, most of it the
    // look-up.  So the steps are software-pipelined: pointers three steps ahead, columns two, values and pivot one, and
    // the look-up of step t + 1 runs while step t's values are being asked for again.
    auto slot_of = [&](int j, int t) {   // position of column j among this row's columns right of position t, or -1
      if (hbits) {   // (a column of a pivot row's upper part that this row has lies right of the pivot's position)
        int h = (int)(((unsigned)j * 2654435761u) >> (32 - hbits));
        for (;;) {
          const int key = hkey[h];
          if (key == j) return (int)hpos[h];
          if (key == -1) return -1;
          h = (h + 1) & hmask;
        }
      }
      int lo = t + 1, hi = len - 1;
      while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const int c = cols[mid];
        if (c == j) return mid;
        if (c < j) lo = mid + 1; else hi = mid - 1;
      }
      return -1;
    };
    long long kb = 0, ke = 0, kb1 = 0, ke1 = 0, kb2 = 0, ke2 = 0;
    int kd = 0, kd1 = 0, kd2 = 0, j1 = -1, p = -1;
    unsigned long long xd = 0ull, xu = 0ull;
    if (nlow > 0) { const int k = cols[0]; kb = rp[k]; ke = rp[k + 1]; kd = dg[k]; }
    if (nlow > 1) { const int k = cols[1]; kb1 = rp[k]; ke1 = rp[k + 1]; kd1 = dg[k]; }
    if (nlow > 2) { const int k = cols[2]; kb2 = rp[k]; ke2 = rp[k + 1]; kd2 = dg[k]; }
    if (nlow > 0) {
      const long long q = kb + kd + 1 + lane;
      int j0 = -1;
      if (q < ke) { j0 = ci[q]; xu = sf_load(fout + q); }
      xd = sf_load(fout + kb + kd);
      if (nlow > 1) { const long long q1 = kb1 + kd1 + 1 + lane; if (q1 < ke1) j1 = ci[q1]; }
      if (j0 >= 0) p = slot_of(j0, 0);
      if (p < 0) xu = 0ull;
    }
    for (int t = 0; t < nlow && !dead; ++t) {
      // requests for the steps to come
      long long kb3 = 0, ke3 = 0;
      int kd3 = 0, j2 = -1, p1 = -1;
      unsigned long long xd1 = 0ull, xu1 = 0ull;
      if (t + 3 < nlow) { const int k = cols[t + 3]; kb3 = rp[k]; ke3 = rp[k + 1]; kd3 = dg[k]; }
      if (t + 2 < nlow) { const long long q2 = kb2 + kd2 + 1 + lane; if (q2 < ke2) j2 = ci[q2]; }
      const long long q1 = kb1 + kd1 + 1 + lane;
      if (t + 1 < nlow) {
        if (q1 < ke1) xu1 = sf_load(fout + q1);
        xd1 = sf_load(fout + kb1 + kd1);
      }
      // this step: ask again for what was not there a step ago, and look the next step's column up meanwhile
      const long long ub = kb + kd + 1;   // the pivot row's upper part
      long long q = ub + lane;
      if (xd == kGiluSentinel) xd = sf_load(fout + kb + kd);
      if (xu == kGiluSentinel) xu = sf_load(fout + q);
      if (j1 >= 0) p1 = slot_of(j1, t + 1);
      int spins = 0;
      bool gave_up = false;
      while ((xd == kGiluSentinel || xu == kGiluSentinel) && !gave_up) {   // the pivot and this lane's entry
        __builtin_amdgcn_s_sleep(1);
        if (xd == kGiluSentinel) xd = sf_load(fout + kb + kd);
        if (xu == kGiluSentinel) xu = sf_load(fout + q);
        if (++spins >= kSpinLimit || ((spins & 255) == 0 && sf_flag(tmo))) gave_up = true;
      }
      const double lik = w[t] / __longlong_as_double((long long)xd);
      if (gave_up) s_dead = 1;
      __syncthreads();
      if (lane == 0) w[t] = lik;
      if (p >= 0) w[p] -= lik * __longlong_as_double((long long)xu);
      for (q += T; q < ke && !gave_up; q += T) {   // upper parts longer than the workgroup
        const int pl = slot_of(ci[q], t);
        if (pl < 0) continue;
        unsigned long long x1;
        spins = 0;
        while ((x1 = sf_load(fout + q)) == kGiluSentinel) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins >= kSpinLimit || ((spins & 255) == 0 && sf_flag(tmo))) { gave_up = true; break; }
        }
        if (!gave_up) w[pl] -= lik * __longlong_as_double((long long)x1);
      }
      if (gave_up) s_dead = 1;
      __syncthreads();
      dead = s_dead != 0;
      kb = kb1; ke = ke1; kd = kd1; kb1 = kb2; ke1 = ke2; kd1 = kd2; kb2 = kb3; ke2 = ke3; kd2 = kd3;
      p = p1; j1 = j2; xd = xd1; xu = p1 >= 0 ? xu1 : 0ull;
    }
    if (dead && lane == 0) __hip_atomic_store((gi32_t *)tmo, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (len > nlow && !(fabs(w[nlow]) > 0.0) && lane == 0) atomicOr(err, 2);  // zero pivot (every row, level 0 included)
    for (int t = lane; t < len; t += T) {
      unsigned long long bits = dead ? 0ull : (unsigned long long)__double_as_longlong(w[t]);
      if (bits == kGiluSentinel) bits ^= 1ull;  // cannot come out of arithmetic; never leave a dependant waiting
      sf_store(fout + b + t, bits);
    }
  }
}

inline void schwarz_destroy(isph_schwarz *S) {
  if (!S) return;
  S->rp.release(); S->ci.release(); S->dg.release(); S->val.release(); S->w.release(); S->rows.release();
  S->lord.release(); S->uord.release(); S->rev_ptr.release(); S->rev_idx.release(); S->err.release();
  S->sub_desc.release(); S->sub_lptr.release(); S->sub_nlist.release(); S->loc_ptr_dev.release();
  S->lord4.release(); S->uord4.release(); S->lpos4.release(); S->upos4.release(); S->lrun.release(); S->urun.release(); S->ctr.release(); S->ybits.release(); S->zbits.release();
  if (S->h_tmo) (void)hipHostFree(S->h_tmo);
  delete S;
}

// host copy of A as CSR with ascending columns; columns >= nrow (ghosts owned by other ranks) are dropped: they
// are outside every local subdomain (Ifpack_LocalFilter)
// host array that is either a std::vector taken over from the caller or a block whose elements are NOT value-initialised:
// the 4.4 GB of factor arrays of a 10^6-row ILU(1) pattern are written once, by 16 threads -- a std::vector::resize would
// first zero them on one
// Process-wide cache of the large HOST work arrays of schwarz_create.  The preconditioner is rebuilt every solve
// (solver_lin_belos.h:153,190); at 10^6 rows its set-up touches 4-6 GB of host arrays, and with malloc / free per call the
// first-touch page faults and the munmap of those pages cost more than the work done on them (0.6 s of a 1.5 s create).
// Blocks are reused when the request fits within 50 % slack; at most kCapBytes stay cached; isph_pool_trim() frees them.
struct HostPool {
  static constexpr size_t kCapBytes = (size_t)24 << 30;
  std::multimap<size_t, void *> blocks;
  size_t cached = 0;
  std::mutex mu;
  static HostPool &get() { static HostPool p; return p; }
  void *alloc(size_t bytes, size_t *got) {
    {
      std::lock_guard<std::mutex> lk(mu);
      auto it = blocks.lower_bound(bytes);
      if (it != blocks.end() && it->first <= bytes + bytes / 2 + 4096) {
        void *p = it->second;
        *got = it->first;
        cached -= it->first;
        blocks.erase(it);
        return p;
      }
    }
    *got = bytes;
    return malloc(bytes > 0 ? bytes : 1);
  }
  void release(void *p, size_t bytes) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(mu);
    if (bytes >= ((size_t)1 << 20) && cached + bytes <= kCapBytes) { blocks.emplace(bytes, p); cached += bytes; }
    else free(p);
  }
  void trim() {
    std::lock_guard<std::mutex> lk(mu);
    for (auto &kv : blocks) free(kv.second);
    blocks.clear();
    cached = 0;
  }
};

template <class T>
struct HostArr {
  using value_type = T;
  std::vector<T> v;
  T *raw = nullptr;
  size_t n = 0, bytes = 0;
  HostArr() = default;
  HostArr(const HostArr &) = delete;
  HostArr &operator=(const HostArr &) = delete;
  ~HostArr() { HostPool::get().release(raw, bytes); }
  bool alloc(size_t k) {
    HostPool::get().release(raw, bytes);
    raw = static_cast<T *>(HostPool::get().alloc((k > 0 ? k : 1) * sizeof(T), &bytes));
    n = k;
    return raw != nullptr;
  }
  T *data() { return raw ? raw : v.data(); }
  const T *data() const { return raw ? raw : v.data(); }
  size_t size() const { return raw ? n : v.size(); }
  bool empty() const { return size() == 0; }
  T &operator[](size_t i) { return data()[i]; }
  const T &operator[](size_t i) const { return data()[i]; }
  void swap(HostArr &o) { v.swap(o.v); std::swap(raw, o.raw); std::swap(n, o.n); std::swap(bytes, o.bytes); }
};

// keep != nullptr: the CSR image is written into keep's factor arrays and stays there; only the pattern comes to the
// host (the one-subdomain ILU(0) case, where the matrix IS the factor pattern: no 0.8 GB of values down and up again)
inline int schwarz_host_csr(isph_ctx *ctx, const isph_mat *A, std::vector<long long> &rp, HostArr<int> &ci,
                            HostArr<double> &v, isph_schwarz *keep = nullptr, bool pattern_to_host = true) {
  const Sell &S = A->S;
  const int n = S.nrow;
  std::vector<int> len((size_t)n);
  ISPH_CHECK_HIP(hipMemcpyAsync(len.data(), S.rowlen.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  rp.assign((size_t)n + 1, 0);
  for (int i = 0; i < n; ++i) rp[(size_t)i + 1] = rp[(size_t)i] + len[(size_t)i];
  const long long nnz = rp[(size_t)n];
  if (keep) {
    ISPH_CHECK(keep->rp.reserve((size_t)n + 1));
    ISPH_CHECK(keep->ci.reserve((size_t)(nnz > 0 ? nnz : 1)));
    ISPH_CHECK(keep->val.reserve((size_t)(nnz > 0 ? nnz : 1)));
    ISPH_CHECK_HIP(hipMemcpyAsync(keep->rp.p, rp.data(), sizeof(long long) * ((size_t)n + 1), hipMemcpyHostToDevice, ctx->stream));
    if (n > 0)
      hipLaunchKernelGGL(k_sell_to_csr, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, n, S.rowlen.p,
                         S.slice_off.p, S.col.p, S.val.p, keep->rp.p, keep->ci.p, keep->val.p);
    if (!pattern_to_host) return ISPH_SUCCESS;   // the caller works on the device image alone
    if (!ci.alloc((size_t)nnz)) return fail("host allocation failed", __FILE__, __LINE__);
    ISPH_CHECK_HIP(hipMemcpyAsync(ci.data(), keep->ci.p, sizeof(int) * (size_t)nnz, hipMemcpyDeviceToHost, ctx->stream));
    ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return ISPH_SUCCESS;
  }
  DevTmp<long long> drp;
  DevTmp<int> dci;
  DevTmp<double> dv;
  ISPH_CHECK(drp.reserve((size_t)n + 1));
  ISPH_CHECK(dci.reserve((size_t)(nnz > 0 ? nnz : 1)));
  ISPH_CHECK(dv.reserve((size_t)(nnz > 0 ? nnz : 1)));
  ISPH_CHECK_HIP(hipMemcpyAsync(drp.p, rp.data(), sizeof(long long) * ((size_t)n + 1), hipMemcpyHostToDevice, ctx->stream));
  if (n > 0)
    hipLaunchKernelGGL(k_sell_to_csr, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, n, S.rowlen.p,
                       S.slice_off.p, S.col.p, S.val.p, drp.p, dci.p, dv.p);
  if (!ci.alloc((size_t)nnz) || !v.alloc((size_t)nnz)) return fail("host allocation failed", __FILE__, __LINE__);
  ISPH_CHECK_HIP(hipMemcpyAsync(ci.data(), dci.p, sizeof(int) * (size_t)nnz, hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipMemcpyAsync(v.data(), dv.p, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  drp.release(); dci.release(); dv.release();
  return ISPH_SUCCESS;  // SELL rows are kept column-sorted by the assembly / ingress (sell.hpp)
}

struct GiluRow {  // level-of-fill pattern of one local row
  std::vector<int> col;
  std::vector<unsigned char> lev;
  int diag = -1;
};

// ILU(K) pattern of one subdomain's local matrix (rows base..base+m of the block-diagonal local CSR), in place:
// out[r] gets the sorted columns (local to the subdomain) with their levels.
inline void gilu_symbolic_sub(int m, const long long *lrp, const int *lci, int base, int K, std::vector<GiluRow> &out) {
  std::vector<int> levmap((size_t)m, -1), touched;
  std::priority_queue<int, std::vector<int>, std::greater<int>> pivots;
  for (int r = 0; r < m; ++r) {
    touched.clear();
    for (long long p = lrp[base + r]; p < lrp[base + r + 1]; ++p) {
      const int c = lci[p] - base;
      levmap[(size_t)c] = 0;
      touched.push_back(c);
      if (c < r) pivots.push(c);
    }
    if (levmap[(size_t)r] < 0) {  // structurally missing diagonal: Ifpack inserts it
      levmap[(size_t)r] = 0;
      touched.push_back(r);
    }
    while (K > 0 && !pivots.empty()) {
      const int k = pivots.top();
      pivots.pop();
      const int lik = levmap[(size_t)k];
      const GiluRow &rk = out[(size_t)k];
      for (size_t q = (size_t)rk.diag + 1; q < rk.col.size(); ++q) {
        const int j = rk.col[q];
        const int nl = lik + (int)rk.lev[q] + 1;
        if (nl > K) continue;
        int &cur = levmap[(size_t)j];
        if (cur < 0) {
          cur = nl;
          touched.push_back(j);
          if (j < r) pivots.push(j);
        } else if (nl < cur) {
          cur = nl;
        }
      }
    }
    while (!pivots.empty()) pivots.pop();
    std::sort(touched.begin(), touched.end());
    GiluRow &row = out[(size_t)r];
    row.col = touched;
    row.lev.resize(touched.size());
    for (size_t q = 0; q < touched.size(); ++q) {
      row.lev[q] = (unsigned char)levmap[(size_t)touched[q]];
      if (touched[q] == r) row.diag = (int)q;
      levmap[(size_t)touched[q]] = -1;
    }
  }
}

// ILU(1): a level-1 entry (i,j) is the sum of two level-0 entries (i,k), (k,j), k < min(i,j), and generates nothing
// further (0 + 1 + 1 > 1, 1 + 0 + 1 > 1) -- the pattern of a row depends on A's pattern only, not on the final patterns
// of earlier rows, so the rows of a subdomain are independent: rows [r0, r1) of one subdomain, any thread.  Same result
// as gilu_symbolic_sub(K = 1), which walks the rows in order (minutes for one subdomain of 10^6 rows).
inline void gilu_symbolic1_rows(int m, const long long *lrp, const int *lci, int base, int r0, int r1, std::vector<GiluRow> &out) {
  std::vector<int> levmap((size_t)m, -1), touched;
  for (int r = r0; r < r1; ++r) {
    touched.clear();
    for (long long p = lrp[base + r]; p < lrp[base + r + 1]; ++p) {
      const int c = lci[p] - base;
      levmap[(size_t)c] = 0;
      touched.push_back(c);
    }
    if (levmap[(size_t)r] < 0) {  // structurally missing diagonal: Ifpack inserts it
      levmap[(size_t)r] = 0;
      touched.push_back(r);
    }
    for (long long p = lrp[base + r]; p < lrp[base + r + 1]; ++p) {
      const int k = lci[p] - base;
      if (k >= r) continue;
      for (long long q = lrp[base + k]; q < lrp[base + k + 1]; ++q) {
        const int j = lci[q] - base;
        if (j > k && levmap[(size_t)j] < 0) {
          levmap[(size_t)j] = 1;
          touched.push_back(j);
        }
      }
    }
    std::sort(touched.begin(), touched.end());
    GiluRow &row = out[(size_t)r];
    row.col = touched;
    row.lev.resize(touched.size());
    for (size_t q = 0; q < touched.size(); ++q) {
      row.lev[q] = (unsigned char)levmap[(size_t)touched[q]];
      if (touched[q] == r) row.diag = (int)q;
      levmap[(size_t)touched[q]] = -1;
    }
  }
}


// ---- ILU(1) pattern of one subdomain = the whole matrix, on the device -----------------------------------------------
// The level-1 pattern depends on A's pattern only: row i gets A's row i (level 0) and, for every lower neighbour k of
// i, the columns right of the diagonal of A's row k (level 1) -- Ifpack_IlukGraph with LevelFill 1.  One wave per row
// collects the set in an LDS table (open addressing; the value of a slot is the entry's index in A for level 0, -1
// for fill), a first launch counts, a second one sorts the set by rank and writes columns, values (A's, 0 for fill)
// and the diagonal's position.  Host version: gilu_symbolic1_rows (1.6 s on 16 threads at 10^6 rows; here ~0.1 s plus
// bringing the pattern to the host for the level analysis).  Rows longer than kSymCap / 2 raise err bit 4 and create()
// falls back to the host version.
constexpr int kSymBits = 11, kSymCap = 1 << kSymBits;   // slots per row: rows up to 1024 entries
__global__ void k_csr_upper_start(int n, const long long *__restrict__ rp, const int *__restrict__ ci, long long *__restrict__ ustart) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  long long lo = rp[i], hi = rp[i + 1];
  while (lo < hi) {   // first entry with column > i
    const long long mid = (lo + hi) >> 1;
    if (ci[mid] <= i) lo = mid + 1; else hi = mid;
  }
  ustart[i] = lo;
}

template <bool FILL>
__global__ __launch_bounds__(128) void k_gilu_symbolic1(int n, const long long *__restrict__ rp, const int *__restrict__ ci,
                                                        const double *__restrict__ av, const long long *__restrict__ ustart,
                                                        int *__restrict__ rowlen, const long long *__restrict__ frp,
                                                        int *__restrict__ fci, double *__restrict__ fv, int *__restrict__ fdg,
                                                        int *__restrict__ err) {
  __shared__ int s_key[2][kSymCap], s_val[2][kSymCap], s_list[2][kSymCap / 2], s_cnt[2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int *key = s_key[wave], *vv = s_val[wave], *list = s_list[wave];
  const int row0 = (blockIdx.x * 2 + wave);
  const int stride = gridDim.x * 2;
  for (int i = row0; i < n; i += stride) {   // wave-uniform
    for (int t = lane; t < kSymCap; t += 64) key[t] = -1;
    if (lane == 0) s_cnt[wave] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    auto insert = [&](int j, int v) {
      int h = (int)(((unsigned)j * 2654435761u) >> (32 - kSymBits));
      for (;;) {
        const int old = atomicCAS(&key[h], -1, j);
        if (old == -1) { vv[h] = v; atomicAdd(&s_cnt[wave], 1); return; }
        if (old == j) return;   // level 0 entries go in first and stay
        h = (h + 1) & (kSymCap - 1);
      }
    };
    const long long b = rp[i], e = rp[i + 1];
    if (e - b >= kSymCap / 2) {   // A's own row does not fit the table: the host builds the pattern
      if (lane == 0) { atomicOr(err, 4); if (!FILL) rowlen[i] = 0; }
      continue;
    }
    bool has_diag = false;
    for (long long p = b + lane; p < e; p += 64) { const int c = ci[p]; insert(c, (int)(p - b)); has_diag = has_diag || c == i; }
    if (__ballot(has_diag) == 0ull && lane == 0) insert(i, -2);   // structurally missing diagonal: Ifpack inserts it (value 0)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (long long p = b; p < e; ++p) {
      const int k = ci[p];   // the same address in every lane
      if (k >= i) break;
      const long long ue = rp[k + 1];
      for (long long q = ustart[k] + lane; q < ue; q += 64) {
        if (s_cnt[wave] >= kSymCap / 2) break;   // overflow: reported below
        insert(ci[q], -1);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int cnt = s_cnt[wave];
    if (cnt >= kSymCap / 2) { if (lane == 0) atomicOr(err, 4); if (!FILL && lane == 0) rowlen[i] = 0; continue; }
    if (!FILL) {
      if (lane == 0) rowlen[i] = cnt;
      continue;
    }
    // compact the occupied slots, rank them, write the row
    if (lane == 0) s_cnt[wave] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int t = lane; t < kSymCap; t += 64)
      if (key[t] != -1) list[atomicAdd(&s_cnt[wave], 1)] = t;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const long long w0 = frp[i];
    for (int a = lane; a < cnt; a += 64) {
      const int slot = list[a], ka = key[slot];
      int rank = 0;
      for (int c = 0; c < cnt; ++c) rank += key[list[c]] < ka;
      fci[w0 + rank] = ka;
      const int v = vv[slot];
      fv[w0 + rank] = v >= 0 ? av[b + v] : 0.0;
      if (ka == i) fdg[i] = rank;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

inline int schwarz_create(isph_ctx *ctx, const isph_mat *A, int fill, int block_size, int overlap, int combine,
                          isph_schwarz **out, bool syncfree = true) {
  ISPH_REQUIRE(fill >= 0 && fill <= 8 && overlap >= 0 && (combine == 0 || combine == 1), "bad Schwarz parameters");
  const int n = A->S.nrow;
  std::vector<long long> rp;
  HostArr<int> ci;
  HostArr<double> av;
  auto clk = [] { return std::chrono::steady_clock::now(); };
  auto ms_since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(clk() - t0).count(); };
  auto t0 = clk();
  isph_schwarz *S = new isph_schwarz();
  // one subdomain = the whole matrix, no ghost columns to filter, no fill: the device image of the matrix is the factor
  bool resident = fill == 0 && (block_size <= 0 || block_size >= n) && n > 0 && A->S.ncol == n;
  // the same with fill 1: the level-1 pattern is built on the device from the device image of the matrix
  // (k_gilu_symbolic1), the factor arrays never exist on the host; `Adev` holds the matrix image meanwhile
  bool resident1 = fill == 1 && (block_size <= 0 || block_size >= n) && n > 0 && A->S.ncol == n;
  bool devlocal = false;
  isph_schwarz Adev;
  struct AdevGuard {
    isph_schwarz &a;
    void drop() { a.rp.release(); a.ci.release(); a.val.release(); }
    ~AdevGuard() { drop(); }
  } adev_guard{Adev};
  {
    // many small subdomains with no fill and at most one overlap layer: the local matrices are built on the device from the
    // matrix image (k_gilu_sub_local); only the pattern comes to the host, for the subdomain row lists
    devlocal = syncfree && fill == 0 && overlap <= 1 && !resident && n > 0 && block_size > 0 && A->S.wmax <= kSubFactorMaxRow &&
               (n + block_size - 1) / block_size >= kSubSweepMinSubs;
    const int rc0 = schwarz_host_csr(ctx, A, rp, ci, av, resident ? S : (resident1 || devlocal) ? &Adev : nullptr, !devlocal);
    if (rc0 != ISPH_SUCCESS) { schwarz_destroy(S); return rc0; }
  }
  S->t_ms[0] = ms_since(t0); t0 = clk();
  S->n = n; S->fill = fill; S->overlap = overlap; S->combine = combine; S->syncfree = syncfree;
  S->syncfree_requested = syncfree;
  // ---- subdomains: consecutive owned ranges, extended by `overlap` layers (ascending global row per layer)
  const int B = block_size > 0 ? block_size : (n > 0 ? n : 1);
  const int nsub = n > 0 ? (n + B - 1) / B : 0;
  S->nsub = nsub;
  std::vector<std::vector<int>> srows((size_t)nsub);
  std::vector<int> nown((size_t)nsub, 0);
  if (devlocal) {
    // the row lists on the device: hash set + sort per subdomain, prefix sum, fill (k_gilu_sub_rows / _fill)
    const int stride = kSubSweepMaxRows;
    DevTmp<int> ext, msize, dflag;
    DevTmp<long long> lp64;
    DevBuf<char> scan_tmp;
    int rcd = msize.reserve((size_t)nsub + 1);
    if (rcd == ISPH_SUCCESS) rcd = dflag.reserve(1);
    if (rcd == ISPH_SUCCESS) rcd = lp64.reserve((size_t)nsub + 1);
    if (rcd == ISPH_SUCCESS) rcd = ext.reserve(overlap > 0 ? (size_t)nsub * (size_t)stride : 1);
    if (rcd == ISPH_SUCCESS) rcd = S->loc_ptr_dev.reserve((size_t)nsub + 1);
    hipError_t ed = hipSuccess;
    std::vector<long long> hlp((size_t)nsub + 1, 0);
    int hflag = 0;
    if (rcd == ISPH_SUCCESS) {
      ed = hipMemsetAsync(msize.p + nsub, 0, sizeof(int), ctx->stream);
      if (ed == hipSuccess) ed = hipMemsetAsync(dflag.p, 0, sizeof(int), ctx->stream);
      hipLaunchKernelGGL(k_gilu_sub_rows, dim3(nsub), dim3(256), 0, ctx->stream, nsub, B, n, overlap, (const long long *)Adev.rp.p,
                         (const int *)Adev.ci.p, stride, ext.p, msize.p, dflag.p);
      rcd = amg_scan(ctx, (const int *)msize.p, lp64.p, nsub + 1, scan_tmp);
    }
    if (rcd == ISPH_SUCCESS && ed == hipSuccess) {
      ed = hipMemcpyAsync(hlp.data(), lp64.p, sizeof(long long) * ((size_t)nsub + 1), hipMemcpyDeviceToHost, ctx->stream);
      if (ed == hipSuccess) ed = hipMemcpyAsync(&hflag, dflag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
      if (ed == hipSuccess) ed = hipStreamSynchronize(ctx->stream);
      if (ed == hipSuccess) ed = hipGetLastError();
    }
    scan_tmp.release();
    if (rcd != ISPH_SUCCESS) { schwarz_destroy(S); return rcd; }
    if (ed != hipSuccess) { schwarz_destroy(S); return fail(hipGetErrorString(ed), __FILE__, __LINE__); }
    if (hflag || hlp[(size_t)nsub] > (long long)INT_MAX) {
      // a subdomain too large for the one-workgroup form: the host path, which needs the matrix on the host after all
      devlocal = false;
      adev_guard.drop();
      const int rc0 = schwarz_host_csr(ctx, A, rp, ci, av, nullptr);
      if (rc0 != ISPH_SUCCESS) { schwarz_destroy(S); return rc0; }
    } else {
      S->loc_ptr.resize((size_t)nsub + 1);
      for (int sd = 0; sd <= nsub; ++sd) S->loc_ptr[(size_t)sd] = (int)hlp[(size_t)sd];
      for (int sd = 0; sd < nsub; ++sd) nown[(size_t)sd] = std::min(n, sd * B + B) - sd * B;
      const int nl0 = S->loc_ptr[(size_t)nsub];
      rcd = S->rows.reserve((size_t)(nl0 > 0 ? nl0 : 1));
      if (rcd != ISPH_SUCCESS) { schwarz_destroy(S); return rcd; }
      hipLaunchKernelGGL(k_gilu_sub_rows_fill, dim3(nsub), dim3(256), 0, ctx->stream, nsub, B, n, stride, (const long long *)lp64.p,
                         (const int *)ext.p, S->loc_ptr_dev.p, S->rows.p);
      ed = hipStreamSynchronize(ctx->stream);   // ext, lp64 go back to the pool when this scope ends
      if (ed != hipSuccess) { schwarz_destroy(S); return fail(hipGetErrorString(ed), __FILE__, __LINE__); }
    }
  }
  if (!devlocal) {
    const int nth = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    auto work = [&](int t) {
      std::vector<int> mark((size_t)n, -1);
      for (int s = t; s < nsub; s += nth) {
        const int lo = s * B, hi = std::min(n, lo + B);
        std::vector<int> &rows = srows[(size_t)s];
        for (int i = lo; i < hi; ++i) { rows.push_back(i); mark[(size_t)i] = s; }
        nown[(size_t)s] = hi - lo;
        size_t layer_lo = 0;
        for (int l = 0; l < overlap && nsub > 1; ++l) {  // one subdomain: the overlap is a no-op (Ifpack on one rank)
          const size_t layer_hi = rows.size();
          std::vector<int> cand;
          for (size_t q = layer_lo; q < layer_hi; ++q) {
            const int i = rows[q];
            for (long long p = rp[(size_t)i]; p < rp[(size_t)i + 1]; ++p) {
              const int c = ci[(size_t)p];
              if (c < n && mark[(size_t)c] != s) { mark[(size_t)c] = s; cand.push_back(c); }
            }
          }
          std::sort(cand.begin(), cand.end());
          rows.insert(rows.end(), cand.begin(), cand.end());
          layer_lo = layer_hi;
        }
      }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nth; ++t) th.emplace_back(work, t);
    for (auto &x : th) x.join();
  }
  if (!devlocal) {
    S->loc_ptr.assign((size_t)nsub + 1, 0);
    for (int s = 0; s < nsub; ++s) S->loc_ptr[(size_t)s + 1] = S->loc_ptr[(size_t)s] + (int)srows[(size_t)s].size();
  }
  const int nloc = S->loc_ptr[(size_t)nsub];
  S->nloc = nloc;
  std::vector<int> hrows(devlocal ? (size_t)0 : (size_t)nloc);
  if (!devlocal)
    for (int s = 0; s < nsub; ++s)
      std::copy(srows[(size_t)s].begin(), srows[(size_t)s].end(), hrows.begin() + S->loc_ptr[(size_t)s]);
  // ---- local block-diagonal matrix (Ifpack_LocalFilter), columns in local numbering, ascending
  std::vector<long long> lrp(devlocal ? (size_t)1 : (size_t)nloc + 1, 0);
  HostArr<int> lci;
  HostArr<double> lv;
  int dev_maxrow = 0;
  if (devlocal) {
    DevTmp<int> dnown, dcnt, dcown, dmeta;
    DevBuf<char> scan_tmp;
    int rcd = dnown.reserve((size_t)nsub);
    if (rcd == ISPH_SUCCESS) rcd = dcnt.reserve((size_t)nloc + 1);
    if (rcd == ISPH_SUCCESS) rcd = dcown.reserve((size_t)nloc);
    if (rcd == ISPH_SUCCESS) rcd = dmeta.reserve(2);
    if (rcd == ISPH_SUCCESS) rcd = S->rp.reserve((size_t)nloc + 1);
    if (rcd == ISPH_SUCCESS) rcd = S->dg.reserve((size_t)nloc);
    hipError_t ed = hipSuccess;
    if (rcd == ISPH_SUCCESS) {
      ed = hipMemcpyAsync(dnown.p, nown.data(), sizeof(int) * (size_t)nsub, hipMemcpyHostToDevice, ctx->stream);
      if (ed == hipSuccess) ed = hipMemsetAsync(dmeta.p, 0, 2 * sizeof(int), ctx->stream);
      if (ed == hipSuccess) ed = hipMemsetAsync(dcnt.p + nloc, 0, sizeof(int), ctx->stream);
    }
    const int wgrid = (int)(((long long)nloc * 64 + 255) / 256);
    if (rcd == ISPH_SUCCESS && ed == hipSuccess) {
      hipLaunchKernelGGL((k_gilu_sub_local<false>), dim3(wgrid), dim3(256), 0, ctx->stream, nloc, nsub, n, (const int *)S->loc_ptr_dev.p,
                         (const int *)dnown.p, (const int *)S->rows.p, (const long long *)Adev.rp.p, (const int *)Adev.ci.p,
                         (const double *)Adev.val.p, dcnt.p, dcown.p, (const long long *)nullptr, (int *)nullptr, (double *)nullptr,
                         (int *)nullptr, dmeta.p);
      rcd = amg_scan(ctx, (const int *)dcnt.p, S->rp.p, nloc + 1, scan_tmp);
    }
    long long tot = 0;
    int hmeta[2] = {0, 0};
    if (rcd == ISPH_SUCCESS && ed == hipSuccess) {
      ed = hipMemcpyAsync(&tot, S->rp.p + nloc, sizeof(long long), hipMemcpyDeviceToHost, ctx->stream);
      if (ed == hipSuccess) ed = hipMemcpyAsync(hmeta, dmeta.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
      if (ed == hipSuccess) ed = hipStreamSynchronize(ctx->stream);
    }
    if (rcd == ISPH_SUCCESS && ed == hipSuccess) {
      S->nnz = tot;
      dev_maxrow = hmeta[0];
      rcd = S->ci.reserve((size_t)(tot > 0 ? tot : 1));
      if (rcd == ISPH_SUCCESS) rcd = S->val.reserve((size_t)(tot > 0 ? tot : 1));
    }
    if (rcd == ISPH_SUCCESS && ed == hipSuccess) {
      hipLaunchKernelGGL((k_gilu_sub_local<true>), dim3(wgrid), dim3(256), 0, ctx->stream, nloc, nsub, n, (const int *)S->loc_ptr_dev.p,
                         (const int *)dnown.p, (const int *)S->rows.p, (const long long *)Adev.rp.p, (const int *)Adev.ci.p,
                         (const double *)Adev.val.p, dcnt.p, dcown.p, (const long long *)S->rp.p, S->ci.p, S->val.p, S->dg.p, dmeta.p);
      ed = hipMemcpyAsync(hmeta, dmeta.p, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
      if (ed == hipSuccess) ed = hipStreamSynchronize(ctx->stream);
      if (ed == hipSuccess) ed = hipGetLastError();
    }
    scan_tmp.release();
    adev_guard.drop();   // the matrix image is no longer needed
    if (rcd != ISPH_SUCCESS) { schwarz_destroy(S); return rcd; }
    if (ed != hipSuccess) { schwarz_destroy(S); return fail(hipGetErrorString(ed), __FILE__, __LINE__); }
    if (hmeta[1] & 1) { schwarz_destroy(S); return fail("structurally missing diagonal in a Schwarz subdomain", __FILE__, __LINE__); }
    resident = true;     // the factor arrays are where they belong: nothing to upload
  } else if (nsub == 1 && nloc == n && A->S.ncol == n) {
    // one subdomain = the whole matrix and no ghost columns to filter (the reference on one rank; the extended matrix of
    // isph_prec_create_overlap): the local matrix is the host CSR itself, rows already column-sorted
    lrp.swap(rp);
    lci.swap(ci);
    lv.swap(av);
  } else {
    const int nth = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    // pass 1: counts
    auto count = [&](int t) {
      std::vector<int> loc((size_t)n, -1);
      for (int s = t; s < nsub; s += nth) {
        const std::vector<int> &rows = srows[(size_t)s];
        for (size_t q = 0; q < rows.size(); ++q) loc[(size_t)rows[q]] = (int)q;
        for (size_t q = 0; q < rows.size(); ++q) {
          const int i = rows[q];
          long long c = 0;
          for (long long p = rp[(size_t)i]; p < rp[(size_t)i + 1]; ++p)
            if (ci[(size_t)p] < n && loc[(size_t)ci[(size_t)p]] >= 0) ++c;
          lrp[(size_t)S->loc_ptr[(size_t)s] + q + 1] = c;
        }
        for (size_t q = 0; q < rows.size(); ++q) loc[(size_t)rows[q]] = -1;
      }
    };
    {
      std::vector<std::thread> th;
      for (int t = 0; t < nth; ++t) th.emplace_back(count, t);
      for (auto &x : th) x.join();
    }
    for (int q = 0; q < nloc; ++q) lrp[(size_t)q + 1] += lrp[(size_t)q];
    // written once by the threads below: no value-initialisation of 4 GB on one thread first
    if (!lci.alloc((size_t)lrp[(size_t)nloc]) || !lv.alloc((size_t)lrp[(size_t)nloc])) { schwarz_destroy(S); return fail("host allocation failed", __FILE__, __LINE__); }
    auto fillm = [&](int t) {
      std::vector<int> loc((size_t)n, -1);
      std::vector<std::pair<int, double>> tmp;
      for (int s = t; s < nsub; s += nth) {
        const std::vector<int> &rows = srows[(size_t)s];
        const int base = S->loc_ptr[(size_t)s];
        for (size_t q = 0; q < rows.size(); ++q) loc[(size_t)rows[q]] = (int)q;
        // local columns ascending: a row's global columns are ascending, the owned rows of the subdomain are a
        // contiguous ascending range and sit in front of the overlap rows, and every overlap layer is ascending -- so
        // the owned columns come out in order, and the others after them are in order too unless they mix layers
        const int nown_s = nown[(size_t)s];
        for (size_t q = 0; q < rows.size(); ++q) {
          const int i = rows[q];
          tmp.clear();
          long long wq = lrp[(size_t)base + q];
          for (long long p = rp[(size_t)i]; p < rp[(size_t)i + 1]; ++p) {
            const int c = ci[(size_t)p];
            if (c >= n) continue;
            const int l = loc[(size_t)c];
            if (l < 0) continue;
            if (l < nown_s) { lci[(size_t)wq] = base + l; lv[(size_t)wq] = av[(size_t)p]; ++wq; }
            else tmp.emplace_back(base + l, av[(size_t)p]);
          }
          auto less = [](const std::pair<int, double> &a, const std::pair<int, double> &b) { return a.first < b.first; };
          if (!std::is_sorted(tmp.begin(), tmp.end(), less)) std::sort(tmp.begin(), tmp.end(), less);
          for (auto &e : tmp) { lci[(size_t)wq] = e.first; lv[(size_t)wq] = e.second; ++wq; }
        }
        for (size_t q = 0; q < rows.size(); ++q) loc[(size_t)rows[q]] = -1;
      }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nth; ++t) th.emplace_back(fillm, t);
    for (auto &x : th) x.join();
  }
  S->t_ms[1] = ms_since(t0); t0 = clk();
  // ---- level-of-fill pattern (k > 0) and the factor arrays
  std::vector<long long> frp(devlocal ? (size_t)1 : (size_t)nloc + 1, 0);
  HostArr<int> fci;
  HostArr<double> fv;
  std::vector<int> fdg(devlocal ? (size_t)0 : (size_t)nloc, -1);
  bool missing_diag = false;
  if (resident1) {
    // count, prefix on the host (4 MB down, 8 MB up), fill; then the pattern comes to the host for the level analysis
    DevTmp<long long> ustart;
    DevTmp<int> rowlen;
    int rcd = ustart.reserve((size_t)n);
    if (rcd == ISPH_SUCCESS) rcd = rowlen.reserve((size_t)n);
    if (rcd == ISPH_SUCCESS) rcd = S->err.reserve(1);
    if (rcd != ISPH_SUCCESS) { schwarz_destroy(S); return rcd; }
    hipError_t e1 = hipMemsetAsync(S->err.p, 0, sizeof(int), ctx->stream);
    hipLaunchKernelGGL(k_csr_upper_start, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, n, (const long long *)Adev.rp.p,
                       (const int *)Adev.ci.p, ustart.p);
    int ncu1 = 256;
    (void)hipDeviceGetAttribute(&ncu1, hipDeviceAttributeMultiprocessorCount, ctx->device);
    const int sgrid = std::min((n + 1) / 2, ncu1 * 8);
    hipLaunchKernelGGL((k_gilu_symbolic1<false>), dim3(sgrid), dim3(128), 0, ctx->stream, n, (const long long *)Adev.rp.p, (const int *)Adev.ci.p,
                       (const double *)Adev.val.p, (const long long *)ustart.p, rowlen.p, (const long long *)nullptr, (int *)nullptr,
                       (double *)nullptr, (int *)nullptr, S->err.p);
    std::vector<int> hlen((size_t)n);
    int herr1 = 0;
    if (e1 == hipSuccess) e1 = hipMemcpyAsync(hlen.data(), rowlen.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
    if (e1 == hipSuccess) e1 = hipMemcpyAsync(&herr1, S->err.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e1 == hipSuccess) e1 = hipStreamSynchronize(ctx->stream);
    if (e1 != hipSuccess) { schwarz_destroy(S); return fail(hipGetErrorString(e1), __FILE__, __LINE__); }
    if (herr1 & 4) {
      // a row too long for the device table: the host builds the pattern (needs the values on the host after all)
      resident1 = false;
      adev_guard.drop();
      lrp.swap(rp); lci.swap(ci); lv.swap(av);   // back to where schwarz_host_csr puts them
      const int rc0 = schwarz_host_csr(ctx, A, rp, ci, av, nullptr);
      if (rc0 != ISPH_SUCCESS) { schwarz_destroy(S); return rc0; }
      lrp.swap(rp); lci.swap(ci); lv.swap(av);
    } else {
      for (int q = 0; q < n; ++q) frp[(size_t)q + 1] = frp[(size_t)q] + hlen[(size_t)q];
      const long long fnz = frp[(size_t)n];
      int rcf = S->rp.reserve((size_t)n + 1);
      if (rcf == ISPH_SUCCESS) rcf = S->ci.reserve((size_t)(fnz > 0 ? fnz : 1));
      if (rcf == ISPH_SUCCESS) rcf = S->val.reserve((size_t)(fnz > 0 ? fnz : 1));
      if (rcf == ISPH_SUCCESS) rcf = S->dg.reserve((size_t)n);
      if (rcf == ISPH_SUCCESS && !fci.alloc((size_t)fnz)) rcf = fail("host allocation failed", __FILE__, __LINE__);
      if (rcf != ISPH_SUCCESS) { schwarz_destroy(S); return rcf; }
      e1 = hipMemcpyAsync(S->rp.p, frp.data(), sizeof(long long) * ((size_t)n + 1), hipMemcpyHostToDevice, ctx->stream);
      hipLaunchKernelGGL((k_gilu_symbolic1<true>), dim3(sgrid), dim3(128), 0, ctx->stream, n, (const long long *)Adev.rp.p, (const int *)Adev.ci.p,
                         (const double *)Adev.val.p, (const long long *)ustart.p, rowlen.p, (const long long *)S->rp.p, S->ci.p, S->val.p,
                         S->dg.p, S->err.p);
      if (e1 == hipSuccess) e1 = hipMemcpyAsync(fci.data(), S->ci.p, sizeof(int) * (size_t)fnz, hipMemcpyDeviceToHost, ctx->stream);
      if (e1 == hipSuccess) e1 = hipMemcpyAsync(fdg.data(), S->dg.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
      if (e1 == hipSuccess) e1 = hipStreamSynchronize(ctx->stream);
      if (e1 == hipSuccess) e1 = hipGetLastError();
      if (e1 != hipSuccess) { schwarz_destroy(S); return fail(hipGetErrorString(e1), __FILE__, __LINE__); }
      adev_guard.drop();   // the matrix image is no longer needed
      resident = true;           // the factor arrays are where they belong: nothing to upload
    }
  }
  if (resident1 || devlocal) {
    // done above
  } else if (fill == 0) {
    frp.swap(lrp);  // the local matrix IS the factor pattern: no second copy of 1.2 GB at 10^6 rows
    fci.swap(lci);
    fv.swap(lv);
    {
      const int nth = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
      std::vector<char> miss((size_t)nth, 0);
      auto find_diag = [&](int t) {
        const int q0 = (int)((long long)nloc * t / nth), q1 = (int)((long long)nloc * (t + 1) / nth);
        for (int q = q0; q < q1; ++q) {
          const int *b = fci.data() + frp[(size_t)q], *e = fci.data() + frp[(size_t)q + 1];
          const auto it = std::lower_bound(b, e, q);
          if (it == e || *it != q) miss[(size_t)t] = 1; else fdg[(size_t)q] = (int)(it - b);
        }
      };
      std::vector<std::thread> th;
      for (int t = 0; t < nth; ++t) th.emplace_back(find_diag, t);
      for (auto &x : th) x.join();
      for (char c : miss) missing_diag = missing_diag || c;
    }
    if (missing_diag) { schwarz_destroy(S); return fail("structurally missing diagonal in a Schwarz subdomain", __FILE__, __LINE__); }
  } else {
    std::vector<std::vector<GiluRow>> pat((size_t)nsub);
    const int nth = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (fill == 1) {  // rows are independent: threads over row ranges of every subdomain (one subdomain = the reference's case)
      for (int s = 0; s < nsub; ++s) pat[(size_t)s].resize((size_t)(S->loc_ptr[(size_t)s + 1] - S->loc_ptr[(size_t)s]));
      auto sym1 = [&](int t) {
        for (int s = 0; s < nsub; ++s) {
          const int base = S->loc_ptr[(size_t)s], m = S->loc_ptr[(size_t)s + 1] - base;
          const int r0 = (int)((long long)m * t / nth), r1 = (int)((long long)m * (t + 1) / nth);
          if (r1 > r0) gilu_symbolic1_rows(m, lrp.data(), lci.data(), base, r0, r1, pat[(size_t)s]);
        }
      };
      std::vector<std::thread> th;
      for (int t = 0; t < nth; ++t) th.emplace_back(sym1, t);
      for (auto &x : th) x.join();
    } else {
      auto sym = [&](int t) {
        for (int s = t; s < nsub; s += nth) {
          const int base = S->loc_ptr[(size_t)s], m = S->loc_ptr[(size_t)s + 1] - base;
          pat[(size_t)s].resize((size_t)m);
          gilu_symbolic_sub(m, lrp.data(), lci.data(), base, fill, pat[(size_t)s]);
        }
      };
      std::vector<std::thread> th;
      for (int t = 0; t < nth; ++t) th.emplace_back(sym, t);
      for (auto &x : th) x.join();
    }
    for (int s = 0; s < nsub; ++s) {
      const int base = S->loc_ptr[(size_t)s];
      for (size_t r = 0; r < pat[(size_t)s].size(); ++r) frp[(size_t)base + r + 1] = (long long)pat[(size_t)s][r].col.size();
    }
    for (int q = 0; q < nloc; ++q) frp[(size_t)q + 1] += frp[(size_t)q];
    if (!fci.alloc((size_t)frp[(size_t)nloc]) || !fv.alloc((size_t)frp[(size_t)nloc])) { schwarz_destroy(S); return fail("host allocation failed", __FILE__, __LINE__); }
    // the factor arrays: pattern columns, A's values scattered into them (fill entries 0); threads over row ranges
    // (370 M entries for the ILU(1) pattern of the 100^3 system: 1.6 s on one thread)
    auto scatter = [&](int t) {
      for (int s = 0; s < nsub; ++s) {
        const int base = S->loc_ptr[(size_t)s], m = (int)pat[(size_t)s].size();
        const int r0 = (int)((long long)m * t / nth), r1 = (int)((long long)m * (t + 1) / nth);
        for (int r = r0; r < r1; ++r) {
          const GiluRow &row = pat[(size_t)s][(size_t)r];
          long long w = frp[(size_t)base + r];
          long long p = lrp[(size_t)base + r];
          const long long pe = lrp[(size_t)base + r + 1];
          fdg[(size_t)base + r] = row.diag;
          for (size_t q = 0; q < row.col.size(); ++q, ++w) {
            const int c = base + row.col[q];
            fci[(size_t)w] = c;
            while (p < pe && lci[(size_t)p] < c) ++p;
            fv[(size_t)w] = p < pe && lci[(size_t)p] == c ? lv[(size_t)p] : 0.0;
          }
          std::vector<int>().swap(pat[(size_t)s][(size_t)r].col);
          std::vector<unsigned char>().swap(pat[(size_t)s][(size_t)r].lev);
        }
      }
    };
    {
      std::vector<std::thread> th;
      for (int t = 0; t < nth; ++t) th.emplace_back(scatter, t);
      for (auto &x : th) x.join();
    }
    for (int s = 0; s < nsub; ++s) std::vector<GiluRow>().swap(pat[(size_t)s]);
  }
  if (!devlocal) S->nnz = frp[(size_t)nloc];
  int maxrow = dev_maxrow;
  if (!devlocal)
    for (int q = 0; q < nloc; ++q) maxrow = std::max(maxrow, (int)(frp[(size_t)q + 1] - frp[(size_t)q]));
  S->maxrow = maxrow;
  if (!devlocal) {   // (more than one row in twenty: the periodic SPH ILU(0) factors have a few rows with all 104 neighbours on one side)
    long long nlong = 0;
    for (int q = 0; q < nloc; ++q)
      nlong += fdg[(size_t)q] > 16 * kSfChunk || frp[(size_t)q + 1] - frp[(size_t)q] - fdg[(size_t)q] - 1 > 16 * kSfChunk;
    S->long_rows = nlong * 20 > nloc;
  }
  if (maxrow > kGiluMaxRow) { schwarz_destroy(S); return fail("ILU(k) row exceeds the LDS row image (lower the level of fill)", __FILE__, __LINE__); }
  S->t_ms[2] = ms_since(t0); t0 = clk();
  // the factor arrays start on their way to the device now, on a helper thread (copies from pageable memory keep the
  // calling thread busy): the level analysis below needs the pattern on the host only -- 4.4 GB and 0.3 s for the ILU(1)
  // pattern of the 100^3 system, next to 0.45 s of level recurrences
  int rc_up = ISPH_SUCCESS;
  std::thread uploader;
  if (!resident) {
    rc_up = S->rp.reserve((size_t)nloc + 1);
    if (rc_up == ISPH_SUCCESS) rc_up = S->ci.reserve((size_t)(S->nnz > 0 ? S->nnz : 1));
    if (rc_up == ISPH_SUCCESS) rc_up = S->val.reserve((size_t)(S->nnz > 0 ? S->nnz : 1));
    if (rc_up != ISPH_SUCCESS) { schwarz_destroy(S); return rc_up; }
    uploader = std::thread([&, dev = ctx->device, st = ctx->stream] {
      hipError_t e = hipSetDevice(dev);
      if (e == hipSuccess) e = hipMemcpyAsync(S->rp.p, frp.data(), sizeof(long long) * ((size_t)nloc + 1), hipMemcpyHostToDevice, st);
      if (e == hipSuccess && S->nnz > 0) e = hipMemcpyAsync(S->ci.p, fci.data(), sizeof(int) * (size_t)S->nnz, hipMemcpyHostToDevice, st);
      if (e == hipSuccess && S->nnz > 0) e = hipMemcpyAsync(S->val.p, fv.data(), sizeof(double) * (size_t)S->nnz, hipMemcpyHostToDevice, st);
      if (e != hipSuccess) rc_up = fail(hipGetErrorString(e), __FILE__, __LINE__);
    });
  }
  struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{uploader};   // on every way out
  // many small subdomains: one workgroup per subdomain does the level analysis, the factorisation and both sweeps on the
  // device (k_gilu_sub_levels, k_gilu_sub_factor, k_gilu_solve_sub); `syncfree == false` on entry
  // (isph_schwarz_params::level_launches) keeps the host analysis and the launch per level as the cross-check
  int maxsub = 0;
  for (int sd = 0; sd < nsub; ++sd) maxsub = std::max(maxsub, S->loc_ptr[(size_t)sd + 1] - S->loc_ptr[(size_t)sd]);
  S->sub_maxrows = maxsub;
  S->subsweep = S->syncfree_requested && nsub >= kSubSweepMinSubs && maxsub <= kSubSweepMaxRows && maxrow <= kSubFactorMaxRow;
  // ---- dependency levels of the two solves (the factorisation follows the L levels)
  std::vector<int> llev, ulev;
  if (!S->subsweep) { llev.assign((size_t)nloc, 0); ulev.assign((size_t)nloc, 0); }
  int nl = 0, nu = 0;
  if (S->subsweep) {
    // on the device, below
  } else if (nsub >= 4) {
    // a row only depends on rows of its own subdomain: the recurrences of different subdomains run on different threads
    const int nth = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<int> tnl((size_t)nth, 0), tnu((size_t)nth, 0);
    auto rec = [&](int t) {
      for (int sd = t; sd < nsub; sd += nth) {
        const int q0 = S->loc_ptr[(size_t)sd], q1 = S->loc_ptr[(size_t)sd + 1];
        for (int q = q0; q < q1; ++q) {
          int l = 0;
          const long long b = frp[(size_t)q];
          for (int k = 0; k < fdg[(size_t)q]; ++k) l = std::max(l, llev[(size_t)fci[(size_t)(b + k)]] + 1);
          llev[(size_t)q] = l;
          tnl[(size_t)t] = std::max(tnl[(size_t)t], l + 1);
        }
        for (int q = q1 - 1; q >= q0; --q) {
          int l = 0;
          for (long long p = frp[(size_t)q] + fdg[(size_t)q] + 1; p < frp[(size_t)q + 1]; ++p) l = std::max(l, ulev[(size_t)fci[(size_t)p]] + 1);
          ulev[(size_t)q] = l;
          tnu[(size_t)t] = std::max(tnu[(size_t)t], l + 1);
        }
      }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nth; ++t) th.emplace_back(rec, t);
    for (auto &x : th) x.join();
    for (int t = 0; t < nth; ++t) { nl = std::max(nl, tnl[(size_t)t]); nu = std::max(nu, tnu[(size_t)t]); }
  } else {   // the two recurrences are sequential in themselves and independent of each other: one thread each
    std::thread tl([&] {
      for (int q = 0; q < nloc; ++q) {
        int l = 0;
        const long long b = frp[(size_t)q];
        for (int t = 0; t < fdg[(size_t)q]; ++t) l = std::max(l, llev[(size_t)fci[(size_t)(b + t)]] + 1);
        llev[(size_t)q] = l;
        nl = std::max(nl, l + 1);
      }
    });
    for (int q = nloc - 1; q >= 0; --q) {
      int l = 0;
      for (long long p = frp[(size_t)q] + fdg[(size_t)q] + 1; p < frp[(size_t)q + 1]; ++p) l = std::max(l, ulev[(size_t)fci[(size_t)p]] + 1);
      ulev[(size_t)q] = l;
      nu = std::max(nu, l + 1);
    }
    tl.join();
  }
  if (nloc == 0) nl = nu = 0;
  S->nlev_l = nl; S->nlev_u = nu;
  // Which form runs: the persistent launches win where the dependency chain is long and narrow (whole-matrix factors:
  // 15 rows per level, 0.6 us per level against 6.6 us for a launch); with thousands of rows per level a launch per
  // level is amortised and streams the rows at full width -- 512-row blocks + one overlap layer at 10^6 rows (294
  // levels of 15 000 rows): application 5.2 ms with level launches, 11.6 ms persistent; factorisation 106 / 234 ms.
  if (syncfree && nloc > 0 && nloc / std::max(1, std::min(nl, nu)) >= kGiluWideLevel) syncfree = false;
  if (S->subsweep) syncfree = false;
  S->syncfree = syncfree;
  auto bucket = [&](const std::vector<int> &lev, int nlev, std::vector<int> &ptr, std::vector<int> &ord) {
    ptr.assign((size_t)nlev + 1, 0);
    for (int q = 0; q < nloc; ++q) ++ptr[(size_t)lev[(size_t)q] + 1];
    for (int l = 0; l < nlev; ++l) ptr[(size_t)l + 1] += ptr[(size_t)l];
    ord.resize((size_t)nloc);
    std::vector<int> cur(ptr.begin(), ptr.end() - 1);
    for (int q = 0; q < nloc; ++q) ord[(size_t)cur[(size_t)lev[(size_t)q]]++] = q;
  };
  std::vector<int> lord, uord;
  if (!S->subsweep) {
    bucket(llev, nl, S->lptr, lord);
    bucket(ulev, nu, S->uptr, uord);
  }
  // the same orders with every level padded to a multiple of four positions: a wave of the synchronisation-free sweeps
  // takes four consecutive positions, and rows of one wave must not wait for each other
  auto pad4 = [&](const std::vector<int> &ptr, const std::vector<int> &ord, std::vector<int> &out4) {
    out4.clear();
    out4.reserve(ord.size() + 4 * (ptr.size() > 0 ? ptr.size() - 1 : 0));
    for (size_t l = 0; l + 1 < ptr.size(); ++l) {
      for (int q = ptr[l]; q < ptr[l + 1]; ++q) out4.push_back(ord[(size_t)q]);
      while (out4.size() % 4) out4.push_back(-1);
    }
  };
  std::vector<int> lord4, uord4;
  if (syncfree) {   // only the persistent sweeps read these
    pad4(S->lptr, lord, lord4);
    pad4(S->uptr, uord, uord4);
  }
  S->n4l = (int)lord4.size(); S->n4u = (int)uord4.size();
  // runs of the LDS hand-off sweeps: whole levels, at most kRun positions (a level longer than that is cut)
  auto runs_of = [&](const std::vector<int> &ptr, std::vector<int> &rs) {
    rs.assign(1, 0);
    int pos = 0, cur = 0;   // cur: positions in the open run
    for (size_t l = 0; l + 1 < ptr.size(); ++l) {
      int sz = (ptr[l + 1] - ptr[l] + 3) / 4 * 4;
      if (cur > 0 && cur + sz > kRun) { rs.push_back(pos); cur = 0; }
      while (sz > 0) {
        const int take = std::min(sz, kRun - cur);
        pos += take; cur += take; sz -= take;
        if (cur == kRun) { rs.push_back(pos); cur = 0; }
      }
    }
    if (cur > 0) rs.push_back(pos);
  };
  std::vector<int> lrun, urun, lpos4, upos4;
  if (syncfree) {
    runs_of(S->lptr, lrun);
    runs_of(S->uptr, urun);
    S->nrun_l = (int)lrun.size() - 1; S->nrun_u = (int)urun.size() - 1;
    lpos4.assign((size_t)nloc, 0); upos4.assign((size_t)nloc, 0);
    for (size_t q = 0; q < lord4.size(); ++q) if (lord4[q] >= 0) lpos4[(size_t)lord4[q]] = (int)q;
    for (size_t q = 0; q < uord4.size(); ++q) if (uord4[q] >= 0) upos4[(size_t)uord4[q]] = (int)q;
  }
  // ---- combine lists: global row -> local rows (Add: every copy, Zero: the owned copy), subdomain order
  std::vector<long long> rev_ptr(devlocal ? (size_t)0 : (size_t)n + 1, 0);
  std::vector<int> rev_idx;
  if (!devlocal) {
    for (int s = 0; s < nsub; ++s) {
      const int base = S->loc_ptr[(size_t)s];
      const int cnt = combine == 0 ? S->loc_ptr[(size_t)s + 1] - base : nown[(size_t)s];
      for (int q = 0; q < cnt; ++q) ++rev_ptr[(size_t)hrows[(size_t)base + q] + 1];
    }
    for (int g = 0; g < n; ++g) rev_ptr[(size_t)g + 1] += rev_ptr[(size_t)g];
    rev_idx.resize((size_t)rev_ptr[(size_t)n]);
    std::vector<long long> cur(rev_ptr.begin(), rev_ptr.end() - 1);
    for (int s = 0; s < nsub; ++s) {
      const int base = S->loc_ptr[(size_t)s];
      const int cnt = combine == 0 ? S->loc_ptr[(size_t)s + 1] - base : nown[(size_t)s];
      for (int q = 0; q < cnt; ++q) rev_idx[(size_t)cur[(size_t)hrows[(size_t)base + q]]++] = base + q;
    }
  }
  S->t_ms[3] = ms_since(t0); t0 = clk();
  // ---- upload
  auto up = [&](auto &buf, const auto &vec) -> int {
    using T = typename std::remove_reference<decltype(vec)>::type::value_type;
    ISPH_CHECK(buf.reserve(vec.size() > 0 ? vec.size() : 1));
    if (!vec.empty())
      ISPH_CHECK_HIP(hipMemcpyAsync(buf.p, vec.data(), sizeof(T) * vec.size(), hipMemcpyHostToDevice, ctx->stream));
    return ISPH_SUCCESS;
  };
  if (uploader.joinable()) uploader.join();
  int rc = rc_up;
  if (rc == ISPH_SUCCESS && !devlocal) rc = up(S->dg, fdg);
  if (rc == ISPH_SUCCESS && !devlocal) rc = up(S->rows, hrows);
  if (rc == ISPH_SUCCESS) rc = up(S->lord, lord);
  if (rc == ISPH_SUCCESS) rc = up(S->uord, uord);
  if (rc == ISPH_SUCCESS && !devlocal) rc = up(S->rev_ptr, rev_ptr);
  if (rc == ISPH_SUCCESS && !devlocal) rc = up(S->rev_idx, rev_idx);
  if (rc == ISPH_SUCCESS && devlocal) {   // the combine lists on the device (k_gilu_rev_*)
    const bool add = combine == 0;
    rc = S->rev_ptr.reserve((size_t)n + 1);
    if (rc == ISPH_SUCCESS) rc = S->rev_idx.reserve((size_t)std::max(add ? nloc : n, 1));
    if (rc == ISPH_SUCCESS && !add) {
      hipLaunchKernelGGL(k_gilu_rev_zero, dim3((n + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, n, B, (const int *)S->loc_ptr_dev.p,
                         S->rev_ptr.p, S->rev_idx.p);
    } else if (rc == ISPH_SUCCESS) {
      DevTmp<int> rcnt;
      DevBuf<char> scan_tmp;
      rc = rcnt.reserve((size_t)n + 1);
      if (rc == ISPH_SUCCESS && hipMemsetAsync(rcnt.p, 0, sizeof(int) * ((size_t)n + 1), ctx->stream) != hipSuccess) rc = fail("memset failed", __FILE__, __LINE__);
      if (rc == ISPH_SUCCESS) {
        hipLaunchKernelGGL(k_gilu_rev_count, dim3((nloc + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, nloc, (const int *)S->rows.p, rcnt.p);
        rc = amg_scan(ctx, (const int *)rcnt.p, S->rev_ptr.p, n + 1, scan_tmp);
      }
      if (rc == ISPH_SUCCESS && hipMemsetAsync(rcnt.p, 0, sizeof(int) * ((size_t)n + 1), ctx->stream) != hipSuccess) rc = fail("memset failed", __FILE__, __LINE__);
      if (rc == ISPH_SUCCESS) {
        hipLaunchKernelGGL(k_gilu_rev_fill, dim3((nloc + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, nloc, (const int *)S->rows.p,
                           (const long long *)S->rev_ptr.p, rcnt.p, S->rev_idx.p);
        hipLaunchKernelGGL(k_gilu_rev_sort, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, n, (const long long *)S->rev_ptr.p, S->rev_idx.p);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail("combine lists failed", __FILE__, __LINE__);   // rcnt, scan_tmp leave scope
      }
      scan_tmp.release();
    }
  }
  if (rc == ISPH_SUCCESS) rc = S->w.reserve((size_t)(nloc > 0 ? nloc : 1));
  if (rc == ISPH_SUCCESS) rc = S->err.reserve(1);
  DevTmp<int> maxlev;
  if (rc == ISPH_SUCCESS && S->subsweep) {
    rc = S->sub_desc.reserve((size_t)4 * (size_t)(nloc > 0 ? nloc : 1));
    if (rc == ISPH_SUCCESS) rc = S->sub_lptr.reserve((size_t)2 * nloc + (size_t)2 * nsub + 2);
    if (rc == ISPH_SUCCESS) rc = S->sub_nlist.reserve((size_t)2 * nsub);
    if (rc == ISPH_SUCCESS) rc = up(S->loc_ptr_dev, S->loc_ptr);
    if (rc == ISPH_SUCCESS) rc = maxlev.reserve(2);
    if (rc == ISPH_SUCCESS && hipMemsetAsync(maxlev.p, 0, 2 * sizeof(int), ctx->stream) != hipSuccess) rc = fail("memset failed", __FILE__, __LINE__);
  }
  if (rc == ISPH_SUCCESS && syncfree) {
    rc = up(S->lord4, lord4);
    if (rc == ISPH_SUCCESS) rc = up(S->uord4, uord4);
    if (rc == ISPH_SUCCESS) rc = up(S->lpos4, lpos4);
    if (rc == ISPH_SUCCESS) rc = up(S->upos4, upos4);
    if (rc == ISPH_SUCCESS) rc = up(S->lrun, lrun);
    if (rc == ISPH_SUCCESS) rc = up(S->urun, urun);
    if (rc == ISPH_SUCCESS) rc = S->ctr.reserve(8);
    if (rc == ISPH_SUCCESS) rc = S->ybits.reserve((size_t)nloc + 1);
    if (rc == ISPH_SUCCESS) rc = S->zbits.reserve((size_t)nloc + 1);
    if (rc == ISPH_SUCCESS && hipHostMalloc((void **)&S->h_tmo, sizeof(int), hipHostMallocDefault) != hipSuccess)
      rc = fail("pinned allocation failed", __FILE__, __LINE__);
    if (rc == ISPH_SUCCESS) *S->h_tmo = 0;
  }
  if (rc != ISPH_SUCCESS) { schwarz_destroy(S); return rc; }
  int ncu = 256;
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device);
  // persistent workgroups of a sweep (16 waves x 4 rows each: one run) and the throttle window.  Narrow levels (the
  // whole-matrix factors: 15 rows per level): the sweep is bound by the hand-off latency of the dependency chain, not by
  // the rows in flight -- the workgroups ahead of the front only have to have their operands loaded by the time it
  // reaches them (about 3 us per run): 16, 32, 64 workgroups gave 9.99, 10.09, 10.18 ms on a 7503-level factor, and a
  // run that starts polling before all but two of the earlier ones are complete only adds traffic
  // (profiles/r03_schwarz_syncfree.txt).  Wide levels (many subdomains: 15 000 rows per level for 512-row blocks + one
  // overlap layer at 10^6 rows): a level is hundreds of runs that do not wait for each other -- all of them, and the
  // next level's, must be open at once, on as many workgroups as the chip holds (one of 1024 threads and 80 registers per CU).
  for (int d = 0; d < 2; ++d) {
    const int nlev = d == 0 ? nl : nu, nrun = d == 0 ? S->nrun_l : S->nrun_u;
    const int per_level = nlev > 0 ? (nrun + nlev - 1) / nlev : 1;   // runs per level
    S->runs_near[d] = std::max(S->long_rows ? kLongWindow : 2, 2 * per_level);
    S->sweep_blocks[d] = std::min(ncu, std::max(std::max(16, ncu / 8), S->runs_near[d] + 16));
  }
  hipError_t e = hipMemsetAsync(S->err.p, 0, sizeof(int), ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // the uploads are done: the stage times below are what they say
  S->t_ms[4] = ms_since(t0); t0 = clk();
  // ---- numeric factorisation
  // LDS of a factorisation workgroup: the row image (values + columns) and, when it fits the 64 KiB every launch may
  // ask for, the column -> position table of k_gilu_factor_sf (a power of two >= twice the longest row, 6 B per slot)
  int hbits = 0;
  while ((1 << hbits) < 2 * std::max(maxrow, 1)) ++hbits;
  const size_t lds_row = (size_t)maxrow * 12 + 16, lds_tab = ((size_t)6 << hbits) + 8;
  if (!syncfree || maxrow >= 65535 || lds_row + lds_tab > 64 * 1024) hbits = 0;
  const size_t lds = lds_row + (hbits ? lds_tab : 0);
  int htmo = 0;
  int hmax[2] = {0, 0};
  if (S->subsweep && nloc > 0) {
    // level analysis (descriptor list of the sweeps) and factorisation, one workgroup per subdomain each
    const size_t lds_lv = sizeof(int) * ((size_t)3 * maxsub + 2);
    hipLaunchKernelGGL(k_gilu_sub_levels, dim3(nsub), dim3(128), lds_lv, ctx->stream, (const int *)S->loc_ptr_dev.p,
                       (const long long *)S->rp.p, (const int *)S->ci.p, (const int *)S->dg.p, S->sub_desc.p, S->sub_lptr.p,
                       S->sub_nlist.p, maxlev.p);
    const int mr = std::max(maxrow, 1), ms = (maxsub + 63) / 64 * 64;
    // (PMC at 100^3, 512 rows + one layer: FETCH_SIZE 76 GB against 8 GB of factor, L2 hit 16 % -- every pivot row's upper
    //  part is re-read ~40 times and 1500 resident subdomains x 0.4 MB of live rows do not fit the 32 MB of L2; fewer resident
    //  subdomains (2 per CU instead of 4-6) were slower, 88 against 59 ms: the chain per subdomain bounds it then)
    const size_t lds_f = (size_t)4 * ((size_t)mr * 12 + (size_t)ms * 2);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_gilu_sub_factor<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f);
    hipLaunchKernelGGL((k_gilu_sub_factor<4>), dim3(nsub), dim3(64 * 4), lds_f, ctx->stream, (const int *)S->loc_ptr_dev.p,
                       (const long long *)S->sub_desc.p, (const int *)S->sub_lptr.p, (const int *)S->sub_nlist.p, (const long long *)S->rp.p,
                       (const int *)S->ci.p, (const int *)S->dg.p, S->val.p, mr, ms, S->err.p);
    if (e == hipSuccess) e = hipMemcpyAsync(hmax, maxlev.p, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
  } else if (syncfree && nloc > 0) {
    // one persistent launch: rows in level order, a row waits for the values of the rows it eliminates with
    // (k_gilu_factor_sf); the factor goes to a second array that starts as the "not there yet" pattern and takes the
    // place of the input afterwards
    DevBuf<double> fout;
    { const int rcf = fout.reserve((size_t)(S->nnz > 0 ? S->nnz : 1)); if (rcf != ISPH_SUCCESS) { schwarz_destroy(S); return rcf; } }
    hipLaunchKernelGGL(k_gilu_fill_sentinel, dim3(stream_grid((int)std::min<long long>(S->nnz, 1 << 30))), dim3(kBlock), 0, ctx->stream,
                       S->nnz, reinterpret_cast<unsigned long long *>(fout.p));
    if (e == hipSuccess) e = hipMemsetAsync(S->ctr.p, 0, 4 * sizeof(int), ctx->stream);
    // threads per row: one per entry of an average pivot row's upper part, up to 256 (512 measured no better on ILU(2))
    const long long upper = nloc > 0 ? (S->nnz - nloc) / 2 / nloc : 0;
    const int fw = upper > 64 ? 4 : 1;
    const void *fk = fw == 4 ? reinterpret_cast<const void *>(k_gilu_factor_sf<4>) : reinterpret_cast<const void *>(k_gilu_factor_sf<1>);
    if (e == hipSuccess) e = hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) {
      const int waves = std::min(nloc, ncu * 16);
      const dim3 grid((waves + fw - 1) / fw), block(64 * fw);
      unsigned long long *fo = reinterpret_cast<unsigned long long *>(fout.p);
      const double *ain = S->val.p;
      if (fw == 4)
        hipLaunchKernelGGL(k_gilu_factor_sf<4>, grid, block, lds, ctx->stream, nloc, S->lord.p, S->rp.p, S->ci.p, S->dg.p, ain, fo, S->ctr.p, S->err.p, hbits);
      else
        hipLaunchKernelGGL(k_gilu_factor_sf<1>, grid, block, lds, ctx->stream, nloc, S->lord.p, S->rp.p, S->ci.p, S->dg.p, ain, fo, S->ctr.p, S->err.p, hbits);
      e = hipMemcpyAsync(&htmo, S->ctr.p + 3, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    }
    std::swap(S->val, fout);   // the stream is synchronised below before anything reads the factor
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    fout.release();
  } else {
    // level by level (level 0 rows have no lower part; their pivots are checked by the launch of level 0 all the same)
    for (int l = 0; l < nl && e == hipSuccess; ++l) {
      const int cnt = S->lptr[(size_t)l + 1] - S->lptr[(size_t)l];
      if (cnt == 0) continue;
      hipLaunchKernelGGL(k_gilu_factor, dim3(cnt), dim3(64), lds, ctx->stream, cnt, S->lord.p + S->lptr[(size_t)l], S->rp.p,
                         S->ci.p, S->dg.p, S->val.p, S->err.p);
    }
  }
  int herr = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&herr, S->err.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // the host vectors above are read by the async copies
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) { schwarz_destroy(S); return fail(hipGetErrorString(e), __FILE__, __LINE__); }
  if (htmo) { schwarz_destroy(S); return fail("Schwarz ILU factorisation: a row waited for its pivot row beyond the spin limit", __FILE__, __LINE__); }
  if (herr) { schwarz_destroy(S); return fail("zero pivot in the Schwarz ILU factorisation", __FILE__, __LINE__); }
  if (S->subsweep) { S->nlev_l = hmax[0]; S->nlev_u = hmax[1]; }
  S->t_ms[5] = ms_since(t0);
  *out = S;
  return ISPH_SUCCESS;
}

inline int schwarz_apply(isph_ctx *ctx, const isph_schwarz *S, const double *r, double *z) {
  if (S->n == 0) return ISPH_SUCCESS;
  const int nloc = S->nloc;
  if (S->subsweep) {
    const size_t lds = sizeof(double) * (size_t)S->sub_maxrows;
    hipLaunchKernelGGL(k_gilu_solve_sub, dim3(S->nsub), dim3(256), lds, ctx->stream, (const int *)S->loc_ptr_dev.p, (const int *)S->rows.p,
                       (const long long *)S->sub_desc.p, (const int *)S->sub_lptr.p, (const int *)S->sub_nlist.p, (const int *)S->ci.p,
                       (const double *)S->val.p, r, S->w.p);
    hipLaunchKernelGGL(k_gilu_combine, dim3((S->n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, S->n, S->rev_ptr.p,
                       S->rev_idx.p, (const double *)S->w.p, z);
    ISPH_CHECK_HIP(hipGetLastError());
    return ISPH_SUCCESS;
  }
  hipLaunchKernelGGL(k_gilu_gather, dim3((nloc + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, nloc, S->rows.p, r, S->w.p);
  if (S->syncfree) {
    // two persistent launches: L sweep (rhs = gathered r, results -> ybits), U sweep (rhs = y, results -> zbits)
    // a time-out of an earlier application is reported by prec_health() at the end of the solve, behind a stream
    // synchronisation and on all ranks together; here the host has not waited for the copy of the word
    hipLaunchKernelGGL(k_gilu_fill_bits, dim3(stream_grid(nloc)), dim3(kBlock), 0, ctx->stream, nloc, S->ybits.p, S->zbits.p);
    ISPH_CHECK_HIP(hipMemsetAsync(S->ctr.p, 0, 2 * sizeof(int), ctx->stream));
    ISPH_CHECK_HIP(hipMemsetAsync(S->ctr.p + 4, 0, 2 * sizeof(int), ctx->stream));
    auto sweep = [&](auto kernel, int d, int nrun, const int *run, const int *ord, const int *pos, const double *rhs, unsigned long long *res) {
      hipLaunchKernelGGL(kernel, dim3(S->sweep_blocks[d]), dim3(16 * kRun), 0, ctx->stream, nrun, run, ord, pos, (const long long *)S->rp.p,
                         (const int *)S->ci.p, (const int *)S->dg.p, (const double *)S->val.p, rhs, res, nloc, S->ctr.p, S->runs_near[d]);
    };
    const double *yv = reinterpret_cast<const double *>(S->ybits.p);
    if (S->long_rows) {
      sweep(k_gilu_solve_run<false, true>, 0, S->nrun_l, S->lrun.p, S->lord4.p, S->lpos4.p, S->w.p, S->ybits.p);
      sweep(k_gilu_solve_run<true, true>, 1, S->nrun_u, S->urun.p, S->uord4.p, S->upos4.p, yv, S->zbits.p);
    } else {
      sweep(k_gilu_solve_run<false, false>, 0, S->nrun_l, S->lrun.p, S->lord4.p, S->lpos4.p, S->w.p, S->ybits.p);
      sweep(k_gilu_solve_run<true, false>, 1, S->nrun_u, S->urun.p, S->uord4.p, S->upos4.p, yv, S->zbits.p);
    }
    ISPH_CHECK_HIP(hipMemcpyAsync(S->h_tmo, S->ctr.p + 3, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    hipLaunchKernelGGL(k_gilu_combine, dim3((S->n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, S->n, S->rev_ptr.p,
                       S->rev_idx.p, reinterpret_cast<const double *>(S->zbits.p), z);
    ISPH_CHECK_HIP(hipGetLastError());
    return ISPH_SUCCESS;
  }
  for (int l = 1; l < S->nlev_l; ++l) {
    const int cnt = S->lptr[(size_t)l + 1] - S->lptr[(size_t)l];
    if (cnt == 0) continue;
    hipLaunchKernelGGL(k_gilu_lower, dim3((cnt * 16 + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, cnt,
                       S->lord.p + S->lptr[(size_t)l], S->rp.p, S->ci.p, S->dg.p, S->val.p, S->w.p);
  }
  for (int l = 0; l < S->nlev_u; ++l) {
    const int cnt = S->uptr[(size_t)l + 1] - S->uptr[(size_t)l];
    if (cnt == 0) continue;
    hipLaunchKernelGGL(k_gilu_upper, dim3((cnt * 16 + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, cnt,
                       S->uord.p + S->uptr[(size_t)l], S->rp.p, S->ci.p, S->dg.p, S->val.p, S->w.p);
  }
  hipLaunchKernelGGL(k_gilu_combine, dim3((S->n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, S->n, S->rev_ptr.p,
                     S->rev_idx.p, S->w.p, z);
  ISPH_CHECK_HIP(hipGetLastError());
  return ISPH_SUCCESS;
}

// test/diagnostic export: rows[nloc], loc_ptr[nsub+1], factor CSR in local numbering
inline int schwarz_export(isph_ctx *ctx, const isph_schwarz *S, int *rows, int *loc_ptr, long long *rowptr, int *colidx,
                          double *val) {
  ISPH_CHECK_HIP(hipMemcpyAsync(rows, S->rows.p, sizeof(int) * (size_t)S->nloc, hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipMemcpyAsync(rowptr, S->rp.p, sizeof(long long) * ((size_t)S->nloc + 1), hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipMemcpyAsync(colidx, S->ci.p, sizeof(int) * (size_t)S->nnz, hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipMemcpyAsync(val, S->val.p, sizeof(double) * (size_t)S->nnz, hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  std::copy(S->loc_ptr.begin(), S->loc_ptr.end(), loc_ptr);
  return ISPH_SUCCESS;
}

}  // namespace isph
