// assemble.hpp -- GPU assembly of the pressure-Poisson system straight into
// the sliced-ELL matrix.
//
// Replaces, for one rank's brick of particles:
//   FunctorOuterGraph                       (ref: functor_graph.h:38-99)
//   FunctorOuterLaplacianMatrix<.,AS>       (ref: functor_laplacian_matrix.h:73-316)
//   FunctorOuterDivergence<.,AS>            (ref: functor_divergence.h:54-124)
//   FunctorOuterIncompNavierStokesPoisson   (ref: functor_incomp_navier_stokes_poisson.h:52-181)
//   PairISPH::modifySingularMatrix          (ref: pair_isph.cpp:493-520)
//   FunctorOuterVolume                      (ref: functor_volume.h:40-80)
// One lane per matrix row (lane == row of a 64-row slice): the row's pattern
// is its in-cut neighbour list in list order followed by the diagonal, so no
// per-entry column search (Epetra SumIntoGlobalValues) is needed.  The two
// neighbour sweeps of the reference stay (the second needs grad m_i and c_i
// complete), but the second one recomputes a_ij instead of re-reading the row.
// MorrisHolmes mirroring (mirror_morris_holmes.h) is applied where the reference's
// *_MorrisHolmes functor combinations apply it; Solid rows with wall normals get the
// homogeneous-Neumann operator rows of functor_gradient_dot_operator_matrix.h.
#pragma once
#include "core.hpp"
#include "sell.hpp"

namespace isph {

int sell_finalize_offsets(isph_ctx *ctx, Sell &S);  // isph_capi.hip
int sell_sort_rows(isph_ctx *ctx, Sell &S);         // isph_capi.hip
int sell_set_wmax(isph_ctx *ctx, Sell &S);          // isph_capi.hip

constexpr double kEps = 1.0e-24;  // ISPH_EPSILON, ref: macrodef.h:6
enum { KIND_FLUID = 99, KIND_SOLID = 12, KIND_BUFFER_DIRICHLET = 32, KIND_BUFFER_NEUMANN = 64, KIND_ALL = 127 };  // pair_isph.h:113-124

struct AsmTables {  // small per-type tables, device resident
  const int *kind;      // [ntypes+1]
  const double *h;      // [(ntypes+1)^2]
  const double *cutsq;  // [(ntypes+1)^2]
  int ntypes, kernel, dim;
  // per type pair, precomputed on the host: 1/h, C(h) and C(h)/h (the reference recomputes the
  // normalisation with pow() on every call, kernel_wendland.h:34-46; the value is the same)
  const double *hinv, *knorm, *kdnorm;
  // neighbour list re-laid out per 64-row slice, lane == row (see k_neigh_transpose):
  // neighbour k of row i sits at nt[noff[i>>6] + k*64 + (i&63)], so the one-lane-per-row
  // kernels read it with coalesced 256-B wave loads instead of 64 scattered lines
  const long long *noff;
  const int *nt;
  const int *nlen;  // [nlocal] neighbours of row i (numneigh); the row kernels never see the CSR offsets, so the
                    // flattened list may be indexed with 32- or 64-bit offsets (neigh_ptr / neigh_ptr64)
  // 1: every row's neighbours are ordered by matrix column (k_neigh_sort), so the row kernels can emit column-sorted
  // rows directly (the diagonal takes its slot on the way) and the SELL row sort is skipped
  int sorted;
};

__device__ __forceinline__ int neigh_at(const AsmTables &T, int i, int k) {
  return T.nt[T.noff[i >> 6] + (long long)k * 64 + (i & 63)];
}

// numneigh per row (for the slice widths of the transposed list)
// rowsrc (may be NULL): row i of the layout is the list of the caller's particle rowsrc[i] (order.hpp: the library's own
// row numbering); idmap (may be NULL): the particle index a list entry j stands for in that numbering
template <class OFF>
__global__ void k_numneigh(int n, const OFF *__restrict__ nptr, int *__restrict__ len, const int *__restrict__ rowsrc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int s = rowsrc ? rowsrc[i] : i;
    len[i] = (int)(nptr[s + 1] - nptr[s]);
  }
}

// Orders every row's neighbour ids by their matrix column (stable: equal columns -- periodic images -- keep their
// list order).  One wave per row.  The keys (column << 10 | list position: unique, so ranks need no tie rule) go to
// LDS; a row that is already in order -- most rows of a list built cell by cell -- is copied through, a short row is
// ranked by counting (two keys per 16-B LDS broadcast read), a long one goes through a bitonic network.
constexpr int kNeighSortCap = 1024;
template <class OFF>
__global__ __launch_bounds__(kBlock) void k_neigh_sort(int n, const OFF *__restrict__ nptr, const int *__restrict__ nidx_in,
                                                       const int *__restrict__ colmap, int *__restrict__ out,
                                                       const int *__restrict__ rowsrc, const int *__restrict__ idmap,
                                                       int stride) {
  // keys of kBlock / 64 rows, `stride` (even, >= longest row + 2) per row: sized by the launch for the lists at hand -- a
  // fixed buffer for the longest row the kernel can take (1024 neighbours) left five workgroups per CU in flight for
  // rows of ~100 neighbours, whose time is the latency of their gathers
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys_dyn[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * (kBlock / 64) + wave;
  if (row >= n) return;
  const int srow = rowsrc ? rowsrc[row] : row;
  const OFF jb = nptr[srow];
  const int len = (int)(nptr[srow + 1] - jb);
  unsigned long long *kw = keys_dyn + (size_t)wave * stride;
  // the list entries in the numbering of the layout (the output keeps the offsets of the source row); with idmap the
  // column map is indexed by the CALLER's particle (one gather per key, the renamed entry is only needed for the store)
  struct Ids {
    const int *p, *m;
    __device__ __forceinline__ int operator[](long long k) const { const int j = p[k]; return m ? m[j] : j; }
  } nidx{nidx_in, idmap};
  for (int k = lane; k < len; k += 64) kw[k] = ((unsigned long long)(unsigned)colmap[nidx_in[jb + k]] << 10) | (unsigned)k;
  if (lane < 2) kw[len + lane] = ~0ull;  // pad: the pair reads below may run one key past the end
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  bool disorder = false;
  for (int k = lane; k + 1 < len; k += 64) disorder |= kw[k + 1] < kw[k];
  if (__ballot(disorder) == 0) {
    for (int k = lane; k < len; k += 64) out[jb + k] = nidx[jb + k];
    return;
  }
  if (len > 128) {
    // long rows (Quintic on a bcc lattice: 748 neighbours): bitonic network over the next power of two, in LDS
    int P = 256;
    while (P < len) P <<= 1;
    for (int k = len + lane; k < P; k += 64) kw[k] = ~0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int k2 = 2; k2 <= P; k2 <<= 1)
      for (int j = k2 >> 1; j > 0; j >>= 1) {
        for (int t = lane; t < (P >> 1); t += 64) {
          const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
          const unsigned long long a = kw[i], b = kw[p];
          if ((a > b) == ((i & k2) == 0)) { kw[i] = b; kw[p] = a; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    for (int k = lane; k < len; k += 64) out[jb + k] = nidx[jb + (int)(kw[k] & 1023ull)];
    return;
  }
  const ulonglong2 *kw2 = reinterpret_cast<const ulonglong2 *>(kw);
  const int npair = (len + 1) >> 1;
  for (int k0 = 0; k0 < len; k0 += 128) {  // two own keys per lane and sweep: the LDS reads are shared
    const int ka = k0 + lane, kb = k0 + 64 + lane;
    const unsigned long long ca = ka < len ? kw[ka] : 0ull, cb = kb < len ? kw[kb] : 0ull;
    int ra = 0, rb = 0;
    for (int q = 0; q < npair; ++q) {
      const ulonglong2 t = kw2[q];
      ra += (t.x < ca) + (t.y < ca);
      rb += (t.x < cb) + (t.y < cb);
    }
    if (ka < len) out[jb + ra] = nidx[jb + ka];
    if (kb < len) out[jb + rb] = nidx[jb + kb];
  }
}

// CSR neighbour list -> lane-interleaved slices.  One wave per slice; 64 rows x 16 ids are staged
// through LDS: the CSR side is read in 64-B row segments, the ELL side written as 256-B wave stores.
template <class OFF>
__global__ __launch_bounds__(kBlock) void k_neigh_transpose(int n, const OFF *__restrict__ nptr,
                                                            const int *__restrict__ nidx,
                                                            const long long *__restrict__ noff, int *__restrict__ nt,
                                                            const int *__restrict__ rowsrc, const int *__restrict__ idmap) {
  __shared__ int lds[kBlock / 64][64 * 17];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slice = blockIdx.x * (kBlock / 64) + wave;
  const int nslices = (n + 63) / 64;
  if (slice >= nslices) return;
  int *wl = lds[wave];
  const int row = slice * 64 + lane;
  const int srow = row < n ? (rowsrc ? rowsrc[row] : row) : 0;
  const long long jb = row < n ? (long long)nptr[srow] : 0, je = row < n ? (long long)nptr[srow + 1] : 0;
  const long long off = noff[slice];
  const int w = (int)((noff[slice + 1] - off) >> 6);
  for (int c0 = 0; c0 < w; c0 += 16) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int r = s * 4 + (lane >> 4), t = lane & 15;
      const long long rb = __shfl(jb, r, 64), re = __shfl(je, r, 64);
      const long long p = rb + c0 + t;
      int j = p < re ? nidx[p] : 0;
      if (idmap && p < re) j = idmap[j];
      wl[r * 17 + t] = j;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int t = 0; t < 16 && c0 + t < w; ++t) nt[off + (long long)(c0 + t) * 64 + lane] = wl[lane * 17 + t];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// kernel normalisation C(h) (ref: kernel_wendland.h:34-46, kernel_quintic.h:34-46, kernel_cubic.h:33-44)
__host__ __device__ inline double kernel_norm(int kernel, int dim, double h) {
  const double pi = 3.14159265358979323846;
  switch (kernel) {
  case 0: return dim == 3 ? 21.0 / (16 * pi * (h * h * h)) : 7.0 / (4 * pi * (h * h));
  case 1: return dim == 3 ? 14.0 / ((h * h * h) * 1745.0 * pi) : 7.0 / ((h * h) * 478.0 * pi);
  default: return dim == 3 ? 1.0 / ((h * h * h) * pi) : 10.0 / ((h * h) * 7.0 * pi);
  }
}

__device__ __forceinline__ double pow3(double a) { return a * a * a; }
__device__ __forceinline__ double pow4(double a) { const double b = a * a; return b * b; }
__device__ __forceinline__ double pow5(double a) { const double b = a * a; return b * b * a; }

// W(r,h)  (ref: kernel_wendland.h:50-58, kernel_quintic.h:48-66, kernel_cubic.h:45-56)
// hinv = 1/h and C = normalisation come from the per-type-pair tables.
__device__ inline double kernel_val(int kernel, double r, double hinv, double C) {
  const double s = fabs(r * hinv);
  double v = 0.0;
  if (kernel == 0) {
    v = s < 2.0 ? pow4(1.0 - 0.5 * s) * (2.0 * s + 1.0) : 0.0;
  } else if (kernel == 1) {
    const int fs = (int)floor(s);
    if (fs <= 0) v += 15.0 * pow5(1.0 - s);
    if (fs <= 1) v -= 6.0 * pow5(2.0 - s);
    if (fs <= 2) v += pow5(3.0 - s);
  } else {
    const int fs = (int)floor(s);
    if (fs == 0) v = 1.0 - 0.75 * (2.0 - s) * s * s;
    else if (fs == 1) v = 0.25 * pow3(2.0 - s);
  }
  return v * C;
}

// dW/dr(r,h)  (ref: kernel_wendland.h:60-68, kernel_quintic.h:68-82, kernel_cubic.h:58-70); Ch = C/h
__device__ inline double kernel_dval(int kernel, double r, double hinv, double Ch) {
  const double s = fabs(r * hinv);
  double v = 0.0;
  if (kernel == 0) {
    v = s < 2.0 ? -5.0 * s * pow3(1.0 - 0.5 * s) : 0.0;
  } else if (kernel == 1) {
    const int fs = (int)floor(s);
    if (fs <= 0) v -= 75.0 * pow4(1.0 - s);
    if (fs <= 1) v += 30.0 * pow4(2.0 - s);
    if (fs <= 2) v -= 5.0 * pow4(3.0 - s);
  } else {
    const int fs = (int)floor(s);
    if (fs == 0) v = (2.25 * s - 3.0) * s;
    else if (fs == 1) { const double a = 2.0 - s; v = -0.75 * a * a; }
  }
  return v * Ch;
}

// r_ij and |r_ij|^2 with the reference's operation order and NO fma
// contraction, so the strict `rsq < cutsq` test (functor_graph.h:84,
// functor_laplacian_matrix.h:142) selects the same pairs as the CPU.
// NOTE: ROCm's __dsub_rn/__dmul_rn/__dadd_rn are plain operators (clang/__clang_hip_math.h), so under hipcc's default
// -ffp-contract=fast the compiler may fuse x*x + s into an fma in one kernel and not in another: the counting pass
// and the fill pass then disagree on pairs that sit EXACTLY on the cut radius (exact lattices: |(3,0,0)| = |(2,2,1)|
// = cut), the row gets padding inside its counted length, the padding carries the row's own column, and everything
// that walks rows (ILU extraction) sees the diagonal several times.  Contraction is therefore switched off inside the
// one function every pass calls.
__device__ __forceinline__ double rsq_nofma(int dim, const double xi[3], const double xj[3], double rij[3]) {
#pragma clang fp contract(off)
  double rsq = 0.0;
  rij[0] = rij[1] = rij[2] = 0.0;
  for (int k = 0; k < dim; ++k) {
    rij[k] = xi[k] - xj[k];
    const double sq = rij[k] * rij[k];
    rsq = rsq + sq;
  }
  return rsq;
}
__device__ __forceinline__ double pair_rsq(int dim, const double *__restrict__ x, int i, int j, double rij[3]) {
  const double xi[3] = {x[3 * (size_t)i], x[3 * (size_t)i + 1], x[3 * (size_t)i + 2]};
  const double xj[3] = {x[3 * (size_t)j], x[3 * (size_t)j + 1], x[3 * (size_t)j + 2]};
  return rsq_nofma(dim, xi, xj, rij);
}

// MirrorMorrisHolmes::computeMirrorCoefficient (ref: mirror_morris_holmes.h:39-52)
__device__ __forceinline__ double mirror_coeff(const double *__restrict__ pnd, const double *__restrict__ vfrac,
                                               double safe, double hij, int i, int j, double cut) {
  double di = 2.0 * cut * (pnd[i] * vfrac[i] - 0.5) + kEps;
  const double dj = 2.0 * cut * (pnd[j] * vfrac[j] - 0.5) + kEps;
  const double dmin = safe * hij;
  if (di < dmin) di = dmin;
  return 1.0 + dj / di;
}

// FunctorOuterVolume: V_i = 1/(W(0) + sum_j W(r_ij))
__global__ void k_volumes(AsmTables T, int nlocal, const double *__restrict__ x, const int *__restrict__ type,
                          const int *__restrict__ nptr, const int *__restrict__ nidx, double *__restrict__ vfrac) {
  const int i = xcd_block() * blockDim.x + threadIdx.x;
  if (i >= nlocal) return;
  const int it = type[i], nt1 = T.ntypes + 1;
  double w = kernel_val(T.kernel, 0.0, T.hinv[it * nt1 + it], T.knorm[it * nt1 + it]);
  for (int jj = 0, nn_i = T.nlen[i]; jj < nn_i; ++jj) {
    const int j = neigh_at(T, i, jj);
    const int jt = type[j];
    double rij[3];
    const double rsq = pair_rsq(T.dim, x, i, j, rij);
    if (rsq < T.cutsq[it * nt1 + jt]) w += kernel_val(T.kernel, sqrt(rsq), T.hinv[it * nt1 + jt], T.knorm[it * nt1 + jt]);
  }
  vfrac[i] = 1.0 / w;
}

// Particle number density of the MorrisHolmes mirror (FunctorOuterNormal's pnd, functor_normal.h:57-133, as called by
// PairISPH_Corrected::computeNormals with the filters (Fluid,Solid) and (Solid,Fluid), pair_isph_corrected.cpp:396-419):
// pnd_i = sum_j W(r_ij) over the neighbours that are NOT of the opposite phase + W(0), so that pnd_i V_i is the
// fraction of the kernel support filled by i's own phase (1 in the bulk, 1/2 at a flat wall).
__global__ void k_pnd(AsmTables T, int nlocal, const double *__restrict__ x, const int *__restrict__ type,
                      double *__restrict__ pnd) {
  const int i = xcd_block() * blockDim.x + threadIdx.x;
  if (i >= nlocal) return;
  const int it = type[i], nt1 = T.ntypes + 1, ikind = T.kind[it];
  if (!(ikind & (KIND_FLUID | KIND_SOLID))) { pnd[i] = 0.0; return; }
  const int opposite = (ikind & KIND_SOLID) ? KIND_FLUID : KIND_SOLID;
  double w = 0.0;
  for (int jj = 0, nn_i = T.nlen[i]; jj < nn_i; ++jj) {
    const int j = neigh_at(T, i, jj);
    const int jt = type[j];
    double rij[3];
    const double rsq = pair_rsq(T.dim, x, i, j, rij);
    if (rsq < T.cutsq[it * nt1 + jt] && !(T.kind[jt] & opposite))
      w += kernel_val(T.kernel, sqrt(rsq) + kEps, T.hinv[it * nt1 + jt], T.knorm[it * nt1 + jt]);
  }
  pnd[i] = w + kernel_val(T.kernel, 0.0, T.hinv[it * nt1 + it], T.knorm[it * nt1 + it]);
}


// ---------------------------------------------------------------------------
// computePre tensors of the Symmetric (consistent) family, one lane per particle.
// G_i = ( - sum_j r_ij (x) r_ij  W'/r  V_j )^-1          (ref: functor_gradient_correction.h:23-71)
// closed-form adjugate inverse like UtilsReference::invertDenseMatrix (utils_reference.cpp:251-326)
__global__ void k_gradient_correction(AsmTables T, int nlocal, const double *__restrict__ x,
                                      const int *__restrict__ type, const int *__restrict__ nptr,
                                      const int *__restrict__ nidx, const double *__restrict__ vfrac,
                                      double *__restrict__ Gc) {
  const int i = xcd_block() * blockDim.x + threadIdx.x;
  if (i >= nlocal) return;
  const int dim = T.dim, nt1 = T.ntypes + 1, it = type[i];
  double G[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int jj = 0, nn_i = T.nlen[i]; jj < nn_i; ++jj) {
    const int j = neigh_at(T, i, jj);
    const int jt = type[j];
    double rij[3];
    const double rsq = pair_rsq(dim, x, i, j, rij);
    if (rsq < T.cutsq[it * nt1 + jt]) {
      const double r = sqrt(rsq) + kEps;
      const double rinv = 1.0 / r;
      const double dwdr = kernel_dval(T.kernel, r, T.hinv[it * nt1 + jt], T.kdnorm[it * nt1 + jt]);
      for (int k2 = 0; k2 < dim; ++k2)
        for (int k1 = 0; k1 < dim; ++k1) G[k2 * dim + k1] -= rij[k1] * rij[k2] * dwdr * rinv * vfrac[j];
    }
  }
  double *B = &Gc[(size_t)i * dim * dim];
#define A_(r, c) G[(c) * dim + (r)]
#define B_(r, c) B[(c) * dim + (r)]
  if (dim == 2) {
    const double det = A_(0, 0) * A_(1, 1) - A_(0, 1) * A_(1, 0);
    B_(0, 0) = A_(1, 1) / det;
    B_(1, 1) = A_(0, 0) / det;
    B_(1, 0) = -A_(1, 0) / det;
    B_(0, 1) = -A_(0, 1) / det;
  } else {
    const double c00 = A_(1, 1) * A_(2, 2) - A_(2, 1) * A_(1, 2);
    const double c01 = -A_(1, 0) * A_(2, 2) + A_(2, 0) * A_(1, 2);
    const double c02 = A_(1, 0) * A_(2, 1) - A_(2, 0) * A_(1, 1);
    const double det = A_(0, 0) * c00 + A_(0, 1) * c01 + A_(0, 2) * c02;
    B_(0, 0) = c00 / det;
    B_(1, 0) = c01 / det;
    B_(2, 0) = c02 / det;
    B_(0, 1) = (-A_(0, 1) * A_(2, 2) + A_(2, 1) * A_(0, 2)) / det;
    B_(1, 1) = (A_(0, 0) * A_(2, 2) - A_(2, 0) * A_(0, 2)) / det;
    B_(2, 1) = (-A_(0, 0) * A_(2, 1) + A_(2, 0) * A_(0, 1)) / det;
    B_(0, 2) = (A_(0, 1) * A_(1, 2) - A_(1, 1) * A_(0, 2)) / det;
    B_(1, 2) = (-A_(0, 0) * A_(1, 2) + A_(1, 0) * A_(0, 2)) / det;
    B_(2, 2) = (A_(0, 0) * A_(1, 1) - A_(1, 0) * A_(0, 1)) / det;
  }
#undef A_
#undef B_
}

// L_i from the dimL x dimL system of functor_laplacian_correction.h:24-153, solved by
// LU with partial pivoting (what LAPACK dgesv does for the reference, utils_reference.cpp:398-407).
// DIM is a template parameter: with the dimension known every index below is a compile-time constant, the tensors
// (A 27, L 36, C 9 doubles) stay in registers and the 6 x 6 elimination is straight-line code with predicated row swaps.
// (With a run-time dimension the same code indexed its local arrays dynamically: 736 B of scratch per lane, 12x slower.)
template <int DIM>
__global__ __launch_bounds__(kBlock) void k_laplacian_correction(AsmTables T, int nlocal, const double *__restrict__ x,
                                                                 const int *__restrict__ type, const int *__restrict__ nptr,
                                                                 const int *__restrict__ nidx, const double *__restrict__ vfrac,
                                                                 const double *__restrict__ Gc, double *__restrict__ Lc,
                                                                 int *__restrict__ nfail) {
  const int i = xcd_block() * blockDim.x + threadIdx.x;
  if (i >= nlocal) return;
  constexpr int dim = DIM, d2 = DIM * DIM, dL = DIM * (DIM + 1) / 2;
  const int nt1 = T.ntypes + 1, it = type[i];
  double A[DIM * DIM * DIM], L[dL * dL], G[d2];
#pragma unroll
  for (int k = 0; k < DIM * DIM * DIM; ++k) A[k] = 0.0;
#pragma unroll
  for (int k = 0; k < dL * dL; ++k) L[k] = 0.0;
#pragma unroll
  for (int k = 0; k < d2; ++k) G[k] = Gc[(size_t)i * d2 + k];
  for (int jj = 0, nn_i = T.nlen[i]; jj < nn_i; ++jj) {  // third-order tensor A^{kmn}
    const int j = neigh_at(T, i, jj);
    const int jt = type[j];
    double rij[3];
    const double rsq = pair_rsq(dim, x, i, j, rij);
    if (rsq < T.cutsq[it * nt1 + jt]) {
      const double r = sqrt(rsq) + kEps;
      const double rinv = 1.0 / r;
      const double dwdr = kernel_dval(T.kernel, r, T.hinv[it * nt1 + jt], T.kdnorm[it * nt1 + jt]);
      double aij[DIM];
#pragma unroll
      for (int k2 = 0; k2 < dim; ++k2) {
        aij[k2] = 0.0;
#pragma unroll
        for (int k1 = 0; k1 < dim; ++k1) aij[k2] += G[k2 * dim + k1] * rij[k1];
        aij[k2] *= dwdr * rinv * vfrac[j];
      }
#pragma unroll
      for (int k3 = 0; k3 < dim; ++k3)
#pragma unroll
        for (int k2 = 0; k2 < dim; ++k2)
#pragma unroll
          for (int k1 = 0; k1 < k2 + 1; ++k1) A[k3 * d2 + k2 * dim + k1] += aij[k3] * rij[k1] * rij[k2];
    }
  }
  for (int jj = 0, nn_i = T.nlen[i]; jj < nn_i; ++jj) {  // linear system
    const int j = neigh_at(T, i, jj);
    const int jt = type[j];
    double rij[3];
    const double rsq = pair_rsq(dim, x, i, j, rij);
    if (rsq < T.cutsq[it * nt1 + jt]) {
      const double r = sqrt(rsq) + kEps;
      const double rinv = 1.0 / r;
      const double dwdr = kernel_dval(T.kernel, r, T.hinv[it * nt1 + jt], T.kdnorm[it * nt1 + jt]);
      double e[DIM];
#pragma unroll
      for (int k = 0; k < dim; ++k) e[k] = rij[k] * rinv;
      double C[d2];
#pragma unroll
      for (int k = 0; k < d2; ++k) C[k] = 0.0;
#pragma unroll
      for (int k3 = 0; k3 < dim; ++k3)
#pragma unroll
        for (int k2 = 0; k2 < dim; ++k2)
#pragma unroll
          for (int k1 = 0; k1 < k2 + 1; ++k1) C[k2 * dim + k1] += A[k3 * d2 + k2 * dim + k1] * e[k3];
#pragma unroll
      for (int k2 = 0; k2 < dim; ++k2)
#pragma unroll
        for (int k1 = 0; k1 < k2 + 1; ++k1) {
          C[k2 * dim + k1] += rij[k1] * e[k2];
          C[k2 * dim + k1] *= dwdr * vfrac[j];
        }
#pragma unroll
      for (int k4 = 0; k4 < dim; ++k4)
#pragma unroll
        for (int k3 = 0; k3 < k4 + 1; ++k3) {
          const int op = k4 * (k4 + 1) / 2 + k3;
#pragma unroll
          for (int k2 = 0; k2 < dim; ++k2)
#pragma unroll
            for (int k1 = 0; k1 < k2 + 1; ++k1) {
              const int mn = k2 * (k2 + 1) / 2 + k1;
              L[op * dL + mn] += C[k2 * dim + k1] * e[k3] * e[k4] * (k3 == k4 ? 1.0 : 2.0);
            }
        }
    }
  }
  double rhs[dL];
#pragma unroll
  for (int k2 = 0; k2 < dim; ++k2)
#pragma unroll
    for (int k1 = 0; k1 < k2 + 1; ++k1) rhs[k2 * (k2 + 1) / 2 + k1] = -(double)(k1 == k2);
  // LU with partial pivoting (first largest entry of the column), column-major L[col*dL + row]
  bool singular = false;
#pragma unroll
  for (int k = 0; k < dL; ++k) {
    int pv = k;
    double amax = fabs(L[k * dL + k]);
#pragma unroll
    for (int r2 = k + 1; r2 < dL; ++r2) {
      const double a = fabs(L[k * dL + r2]);
      if (a > amax) { amax = a; pv = r2; }
    }
    if (amax == 0.0) singular = true;
#pragma unroll
    for (int r2 = k + 1; r2 < dL; ++r2)
      if (pv == r2) {  // swap rows k and r2
#pragma unroll
        for (int c = 0; c < dL; ++c) { const double t = L[c * dL + k]; L[c * dL + k] = L[c * dL + r2]; L[c * dL + r2] = t; }
        const double t = rhs[k]; rhs[k] = rhs[r2]; rhs[r2] = t;
      }
    const double piv = 1.0 / L[k * dL + k];
#pragma unroll
    for (int r2 = k + 1; r2 < dL; ++r2) {
      const double l = L[k * dL + r2] * piv;
      L[k * dL + r2] = l;
      if (l != 0.0) {
#pragma unroll
        for (int c = k + 1; c < dL; ++c) L[c * dL + r2] -= l * L[c * dL + k];
        rhs[r2] -= l * rhs[k];
      }
    }
  }
  if (singular) {  // a zero pivot column: the entries computed past it are meaningless and dropped
    atomicAdd(nfail, 1);
#pragma unroll
    for (int k = 0; k < dL; ++k) Lc[(size_t)i * dL + k] = 0.0;
    return;
  }
#pragma unroll
  for (int r2 = dL - 1; r2 >= 0; --r2) {
    double sacc = rhs[r2];
#pragma unroll
    for (int c = r2 + 1; c < dL; ++c) sacc -= L[c * dL + r2] * rhs[c];
    rhs[r2] = sacc / L[r2 * dL + r2];
  }
#pragma unroll
  for (int k = 0; k < dL; ++k) Lc[(size_t)i * dL + k] = rhs[k];
}

// material = 1/rho over nlocal+nghost (functor_incomp_navier_stokes_poisson.h:88-91)
__global__ void k_reciprocal(int n, const double *__restrict__ a, double *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = 1.0 / a[i];
}

// FunctorOuterGraph row lengths: in-cut neighbours + self
__global__ void k_asm_count(AsmTables T, int nlocal, const double *__restrict__ x, const int *__restrict__ type,
                            const int *__restrict__ nptr, const int *__restrict__ nidx, int *__restrict__ rowlen) {
  const int i = xcd_block() * blockDim.x + threadIdx.x;
  if (i >= nlocal) return;
  const int it = type[i], nt1 = T.ntypes + 1;
  int cnt = 1;
  for (int jj = 0, nn_i = T.nlen[i]; jj < nn_i; ++jj) {
    const int j = neigh_at(T, i, jj);
    double rij[3];
    const double rsq = pair_rsq(T.dim, x, i, j, rij);
    if (rsq < T.cutsq[it * nt1 + type[j]]) ++cnt;
  }
  rowlen[i] = cnt;
}

// the row modifySingularMatrix touches is the first fluid particle in the CALLER's atom order (pair_isph.cpp:493-520):
// with the library's own row numbering (rowsrc = internal row -> caller's row) the minimum is taken over the caller's
// indices and translated back afterwards
__global__ void k_first_fluid(int nlocal, const int *__restrict__ type, const int *__restrict__ kind, int *first,
                              const int *__restrict__ rowsrc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nlocal && kind[type[i]] != KIND_SOLID) atomicMin(first, rowsrc ? rowsrc[i] : i);
}
__global__ void k_first_to_internal(int nlocal, const int *__restrict__ iperm, int *first) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && *first >= 0 && *first < nlocal) *first = iperm[*first];
}

struct PoissonArgs {
  int nlocal, antisym, singular_mode, pin_enabled, morris;
  double dt, safe, solid_normal_diag;
  const double *x, *vfrac, *Gc, *Lc, *rho, *invrho, *vstar, *pnd, *normal;
  const int *type, *nptr, *nidx, *colmap;
  const int *first_fluid;
  // per-particle records for the neighbour gathers of the fluid rows: one 32-B load instead of 3-4 scattered ones
  const double4 *r1;  // x, y, z, vfrac
  const double4 *r2;  // 1/rho, vstar
  const int2 *r3;     // type, matrix column
};

__global__ void k_pack_particles(int nall, const double *__restrict__ x, const double *__restrict__ vfrac,
                                 const double *__restrict__ invrho, const double *__restrict__ vstar,
                                 const int *__restrict__ type, const int *__restrict__ colmap, double4 *__restrict__ r1,
                                 double4 *__restrict__ r2, int2 *__restrict__ r3, int root_of_volume) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nall) return;
  // root_of_volume (the 3-D AntiSymmetric instantiation): the pair volume sqrt(V_i V_j) is formed as sqrt(V_i) sqrt(V_j)
  // from per-particle roots -- one square root per particle instead of one per pair and sweep (2 x 10^8 at 100^3); the
  // product differs from the reference's sqrt(V_i V_j) (functor_laplacian_matrix.h:156) in the last bit at most
  r1[j] = make_double4(x[3 * (size_t)j], x[3 * (size_t)j + 1], x[3 * (size_t)j + 2], root_of_volume ? sqrt(vfrac[j]) : vfrac[j]);
  r2[j] = make_double4(invrho[j], vstar[3 * (size_t)j], vstar[3 * (size_t)j + 1], vstar[3 * (size_t)j + 2]);
  r3[j] = make_int2(type[j], colmap[j]);
}

// One lane per row.  filt = (Fluid, filt_j) per the singular mode
// (functor_incomp_navier_stokes_poisson.h:70-86); alpha = -dt; material = 1/rho.
// DIMT / FAM: 0 / -1 = read dimension and operator family at run time; 3 / 1 = the production case (3-D,
// AntiSymmetric family: G = L = I) compiled with both known, so the correction-tensor loops fold away.
template <int DIMT, int FAM>
__global__ __launch_bounds__(kBlock) void k_asm_poisson(AsmTables T, PoissonArgs a,
                                                        const long long *__restrict__ slice_off,
                                                        int *__restrict__ scol, double *__restrict__ sval,
                                                        double *__restrict__ b) {
  const int i = xcd_block() * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  if (i >= a.nlocal) {
    // pad the tail slice: rows >= nlocal of the last slice keep zeros
    const int nslices = (a.nlocal + kSlice - 1) / kSlice;
    const int slice = i >> 6;
    if (slice < nslices) {
      const long long off = slice_off[slice];
      const int w = (int)((slice_off[slice + 1] - off) >> 6);
      for (int k = 0; k < w; ++k) { const long long p = sell_pos(off, lane, k); scol[p] = 0; sval[p] = 0.0; }
    }
    return;
  }
  const int dim = DIMT ? DIMT : T.dim, nt1 = T.ntypes + 1, dL = dim * (dim + 1) / 2;
  const bool antisym = FAM < 0 ? (a.antisym != 0) : (FAM != 0);
  const int it = a.type[i], ikind = T.kind[it];
  const long long off = slice_off[i >> 6];
  const int w = (int)((slice_off[(i >> 6) + 1] - off) >> 6);
  const int filt_i = KIND_FLUID;
  const int filt_j = a.singular_mode == 0 ? KIND_ALL : KIND_FLUID;
  const double alpha = -a.dt;
  const double mi = a.invrho[i];
  const int jb = 0, je = T.nlen[i];
  int cnt = 0, pdiag = -1;  // pdiag: slot of the diagonal (sorted lists: where the row's own column belongs)
  const int ci_own = a.colmap[i];
  double diag_final;
  double bi = 0.0;

  bool wall_normal = false;
  if (!(ikind & filt_i)) {
    // Laplacian not computed for this row (functor_laplacian_matrix.h:88-96).  Solid rows get the
    // homogeneous-Neumann operator -dt n.grad when a wall normal is present and the Poisson problem is
    // treated as singular (functor_incomp_navier_stokes_poisson.h:98-107,
    // functor_gradient_dot_operator_matrix.h:39-79, functor_gradient_operator.h:74-169: G_i, V_j, coeff 1).
    double nrm[3] = {0, 0, 0}, G[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const bool neumann = a.singular_mode != 0 && a.normal != nullptr && (ikind & KIND_SOLID);
    if (neumann) {
      for (int k = 0; k < dim; ++k) nrm[k] = a.normal[3 * (size_t)i + k];
      for (int k = 0; k < dim * dim; ++k) G[k] = a.Gc[(size_t)i * dim * dim + k];
      double nn = 0.0;
      for (int k = 0; k < dim; ++k) nn += nrm[k] * nrm[k];
      wall_normal = !(nn < 0.5);
    }
    for (int jj = jb; jj < je; ++jj) {
      const int j = neigh_at(T, i, jj - jb);
      double rij[3];
      const int jt = a.type[j];
      const double rsq = pair_rsq(dim, a.x, i, j, rij);
      if (rsq < T.cutsq[it * nt1 + jt]) {
        double v = 0.0;
        if (neumann) {
          const double r = sqrt(rsq) + kEps;
          const double dwdr = kernel_dval(T.kernel, r, T.hinv[it * nt1 + jt], T.kdnorm[it * nt1 + jt]);
          const double vjtmp = dwdr / r * a.vfrac[j];
          for (int k2 = 0; k2 < dim; ++k2) {
            double gitmp = 0.0;
            for (int k1 = 0; k1 < dim; ++k1) gitmp += G[k2 * dim + k1] * rij[k1];
            v += (gitmp * vjtmp) * alpha * nrm[k2];
          }
        }
        const int cj = a.colmap[j];
        if (T.sorted && pdiag < 0 && cj > ci_own) pdiag = cnt++;
        const long long p = sell_pos(off, lane, cnt++);
        scol[p] = cj;
        sval[p] = v;
      }
    }
    diag_final = 0.0;
  } else {
    double G[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, L[6] = {1, 0, 1, 0, 0, 1};
    if (dim == 2) { G[0] = 1; G[1] = 0; G[2] = 0; G[3] = 1; L[0] = 1; L[1] = 0; L[2] = 1; }
    if (!antisym) {
      for (int k = 0; k < dim * dim; ++k) G[k] = a.Gc[(size_t)i * dim * dim + k];
      for (int k = 0; k < dL; ++k) L[k] = a.Lc[(size_t)i * dL + k];
    }
    const double vi = a.vfrac[i];
    const double svi = sqrt(vi);
    double grad_m[3] = {0, 0, 0}, ci[3] = {0, 0, 0};
    double diag1 = 0.0, div = 0.0;
    // ---- sweep 1: grad m_i, c_i, diag, divergence (:127-201, functor_divergence.h:79-117)
    const double xi3[3] = {a.x[3 * (size_t)i], a.x[3 * (size_t)i + 1], a.x[3 * (size_t)i + 2]};
    const double vsi[3] = {a.vstar[3 * (size_t)i], a.vstar[3 * (size_t)i + 1], a.vstar[3 * (size_t)i + 2]};
    for (int jj = jb; jj < je; ++jj) {
      const int j = neigh_at(T, i, jj - jb);
      const double4 q1 = a.r1[j];
      const int2 q3 = a.r3[j];
      const int jt = q3.x, jkind = T.kind[jt];
      double rij[3];
      const double xj3[3] = {q1.x, q1.y, q1.z};
      const double rsq = rsq_nofma(dim, xi3, xj3, rij);  // the same arithmetic as pair_rsq / k_asm_count
      if (!(rsq < T.cutsq[it * nt1 + jt])) continue;
      const double4 q2 = a.r2[j];
      const double vsj[3] = {q2.y, q2.z, q2.w};
      const double mj = q2.x;
      double coeff = ((ikind & filt_i) && (ikind & filt_j)) ? 1.0 : 0.0;
      if (!(ikind & KIND_SOLID) && (jkind & KIND_SOLID)) coeff = ((ikind & filt_i) && (jkind & filt_j)) ? 1.0 : 0.0;
      const double r = sqrt(rsq) + kEps;
      const double rinv = 1.0 / r;
      const double dwdr = kernel_dval(T.kernel, r, T.hinv[it * nt1 + jt], T.kdnorm[it * nt1 + jt]);
      double e[3] = {0, 0, 0};
      for (int k = 0; k < dim; ++k) e[k] = rij[k] * rinv;
      const double vfrac = FAM == 1 ? svi * q1.w : (antisym ? sqrt(vi * q1.w) : q1.w);  // FAM == 1: r1.w holds sqrt(V_j)
      const double vjtmp = dwdr * vfrac;
      for (int k2 = 0; k2 < dim; ++k2) {
        double gitmp = 0.0;
        if (antisym) gitmp = e[k2];  // G = I: the sum below gives exactly this
        else for (int k1 = 0; k1 < dim; ++k1) gitmp += G[k2 * dim + k1] * e[k1];
        if (ikind & jkind) grad_m[k2] += gitmp * vjtmp * (antisym ? (mi + mj) : (mj - mi));
      }
      double aij = 0.0;
      if (antisym) {  // L = I: only the squares survive, summed in the same order
        for (int k2 = 0; k2 < dim; ++k2) aij += e[k2] * e[k2];
      } else {
        for (int k2 = 0, op = 0; k2 < dim; ++k2)
          for (int k1 = 0; k1 < k2 + 1; ++k1, ++op) aij += L[op] * e[k1] * e[k2] * (k1 == k2 ? 1.0 : 2.0);
      }
      aij *= 2.0 * dwdr * vfrac;
      if (!antisym)
        for (int k = 0; k < dim; ++k) ci[k] += aij * e[k];
      aij *= mi * coeff * rinv;
      diag1 += aij;
      // divergence of vstar, filter (Fluid, All); coeff = mirror for fluid-solid pairs
      {
        double dcoeff = 1.0;
        if (a.morris && !(ikind & KIND_SOLID) && (jkind & KIND_SOLID))
          dcoeff = mirror_coeff(a.pnd, a.vfrac, a.safe, T.h[it * nt1 + jt], i, j, sqrt(T.cutsq[it * nt1 + jt]));
        const double vd = dwdr * rinv * vfrac * dcoeff;
        for (int k2 = 0; k2 < dim; ++k2) {
          double gitmp = 0.0;
          if (antisym) gitmp = rij[k2];
          else for (int k1 = 0; k1 < dim; ++k1) gitmp += G[k2 * dim + k1] * rij[k1];
          const double fi = vsi[k2], fj = vsj[k2];
          div += gitmp * (antisym ? (fi + fj) : (fj - fi)) * vd;
        }
      }
    }
    // ---- sweep 2: final off-diagonal values (:204-264), written once
    double diag2 = 0.0;
    for (int jj = jb; jj < je; ++jj) {
      const int j = neigh_at(T, i, jj - jb);
      const double4 q1 = a.r1[j];
      const int2 q3 = a.r3[j];
      const int jt = q3.x, jkind = T.kind[jt];
      double rij[3];
      const double xj3[3] = {q1.x, q1.y, q1.z};
      const double rsq = rsq_nofma(dim, xi3, xj3, rij);  // the same arithmetic as pair_rsq / k_asm_count
      if (!(rsq < T.cutsq[it * nt1 + jt])) continue;
      double coeff = ((ikind & filt_i) && (ikind & filt_j)) ? 1.0 : 0.0;
      if (!(ikind & KIND_SOLID) && (jkind & KIND_SOLID)) coeff = ((ikind & filt_i) && (jkind & filt_j)) ? 1.0 : 0.0;
      const double r = sqrt(rsq) + kEps;
      const double rinv = 1.0 / r;
      const double dwdr = kernel_dval(T.kernel, r, T.hinv[it * nt1 + jt], T.kdnorm[it * nt1 + jt]);
      double e[3] = {0, 0, 0};
      for (int k = 0; k < dim; ++k) e[k] = rij[k] * rinv;
      const double vfrac = FAM == 1 ? svi * q1.w : (antisym ? sqrt(vi * q1.w) : q1.w);  // FAM == 1: r1.w holds sqrt(V_j)
      const double vjtmp = dwdr * vfrac;
      double aij = 0.0;
      if (antisym) {  // L = I: only the squares survive, summed in the same order
        for (int k2 = 0; k2 < dim; ++k2) aij += e[k2] * e[k2];
      } else {
        for (int k2 = 0, op = 0; k2 < dim; ++k2)
          for (int k1 = 0; k1 < k2 + 1; ++k1, ++op) aij += L[op] * e[k1] * e[k2] * (k1 == k2 ? 1.0 : 2.0);
      }
      aij *= 2.0 * dwdr * vfrac;
      aij *= mi * coeff * rinv;
      double bc = 0.0, bg = 0.0;
      for (int k2 = 0; k2 < dim; ++k2) {
        double bij = 0.0;
        if (antisym) bij = e[k2];
        else for (int k1 = 0; k1 < dim; ++k1) bij += G[k2 * dim + k1] * e[k1];
        bc += bij * ci[k2];
        bg += bij * grad_m[k2];
      }
      const double tmp = coeff * (mi * bc * vjtmp - bg * vjtmp);
      double v = -aij;
      v -= tmp;
      diag2 += tmp;
      const int cj = q3.y;
      if (T.sorted && pdiag < 0 && cj > ci_own) pdiag = cnt++;
      const long long p = sell_pos(off, lane, cnt++);
      scol[p] = cj;
      sval[p] = v * alpha;
    }
    diag_final = (diag1 + diag2) * alpha;
    bi = -div;  // b_i = -div(v*)_i  (functor_incomp_navier_stokes_poisson.h:153-156)
  }
  // ---- diagonal / RHS fix-ups (functor_incomp_navier_stokes_poisson.h:126-173)
  if (ikind == KIND_SOLID) {
    // the functor assigns the diagonal only when there is no wall normal (:137-147)
    diag_final = (a.singular_mode != 0 && wall_normal) ? a.solid_normal_diag : 1.0;
    bi = 0.0;
  } else if (a.pin_enabled && *a.first_fluid == i) {  // modifySingularMatrix, once, rank 0
    if (a.singular_mode == 2) {
      for (int k = 0; k < cnt; ++k)
        if (k != pdiag) sval[sell_pos(off, lane, k)] = 0.0;
      diag_final = -1.0;
      bi = 0.0;
    } else if (a.singular_mode == 3) {
      diag_final *= 1.5;
    }
  }
  {
    if (pdiag < 0) pdiag = cnt++;
    const long long p = sell_pos(off, lane, pdiag);
    scol[p] = ci_own;
    sval[p] = diag_final;
  }
  for (int k = cnt; k < w; ++k) {
    const long long p = sell_pos(off, lane, k);
    scol[p] = a.colmap[i];
    sval[p] = 0.0;
  }
  b[i] = bi;
}


// ---------------------------------------------------------------------------
// Helmholtz rows (ref: functor_incomp_navier_stokes_helmholtz.h:52-159):
//   A = diag(1/rho) Laplacian(dt, mu = nu rho), filter (Fluid, All)
//   w = (1-theta) A v ;  A <- -theta A ;  diag = 1 + A_ii (fluid) | 1 (solid)
//   b_ik = v_ik + w_ik + dt (f_ik/rho_i + g_k) - dt/rho_i (grad p)_k
// One lane per row like k_asm_poisson; w is accumulated while the row is built
// (the reference forms it with Epetra's Multiply on the assembled matrix).
// The same rows serve the two scalar callers of the path (MODE template parameter of k_asm_helmholtz):
//   MODE 1, solute transport (ref: functor_solute_transport.h:47-138): Laplacian(dt dcoeff), FilterMatchBinary
//     (Fluid, Fluid - BufferNeumann) -- rows of kind == Fluid only --, w = (1-theta) A c, A <- -theta A,
//     diag = 1 + A_ii on Fluid rows and 1 on Solid / Buffer rows, b = c (+ w on Fluid rows);
//   MODE 2, applied electric potential (ref: functor_applied_electric_potential.h:36-98): Laplacian(-1, sigma),
//     FilterMatchBinary (Fluid, Fluid), no theta scaling, diag = 1 on Solid / Buffer rows, b = phi on Buffer rows, else 0.
// The scalar field (c or phi) travels in r2.y, the material (1 or sigma) in r2.x; dt holds the Laplacian's alpha.
struct HelmholtzArgs {
  int nlocal, antisym, incremental, lda, morris;
  int filt_i, filt_j;  // MODE != 0: row kind to match exactly, neighbour-kind mask
  double dt, theta, g[3], safe;
  const double *x, *vfrac, *Gc, *Lc, *rho, *nu, *p, *f, *v, *pnd;
  const int *type, *nptr, *nidx, *colmap;
  // per-particle records for the neighbour gathers of the fluid rows (as in PoissonArgs)
  const double4 *r1;  // x, y, z, vfrac
  const double4 *r2;  // mu = nu rho, v
  const int2 *r3;     // type, matrix column
};

__global__ void k_pack_particles_helmholtz(int nall, const double *__restrict__ x, const double *__restrict__ vfrac,
                                           const double *__restrict__ nu, const double *__restrict__ rho,
                                           const double *__restrict__ v, const int *__restrict__ type,
                                           const int *__restrict__ colmap, double4 *__restrict__ r1,
                                           double4 *__restrict__ r2, int2 *__restrict__ r3) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nall) return;
  r1[j] = make_double4(x[3 * (size_t)j], x[3 * (size_t)j + 1], x[3 * (size_t)j + 2], vfrac[j]);
  r2[j] = make_double4(nu[j] * rho[j], v[3 * (size_t)j], v[3 * (size_t)j + 1], v[3 * (size_t)j + 2]);
  r3[j] = make_int2(type[j], colmap[j]);
}

__global__ void k_pack_particles_scalar(int nall, const double *__restrict__ x, const double *__restrict__ vfrac,
                                        const double *__restrict__ material, const double *__restrict__ field,
                                        const int *__restrict__ type, const int *__restrict__ colmap,
                                        double4 *__restrict__ r1, double4 *__restrict__ r2, int2 *__restrict__ r3) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nall) return;
  r1[j] = make_double4(x[3 * (size_t)j], x[3 * (size_t)j + 1], x[3 * (size_t)j + 2], vfrac[j]);
  r2[j] = make_double4(material ? material[j] : 1.0, field[j], 0.0, 0.0);
  r3[j] = make_int2(type[j], colmap[j]);
}

// DIMT / FAM as in k_asm_poisson; MODE: 0 velocity Helmholtz, 1 solute transport, 2 applied electric potential
template <int DIMT, int FAM, int MODE = 0>
__global__ __launch_bounds__(kBlock) void k_asm_helmholtz(AsmTables T, HelmholtzArgs a,
                                                          const long long *__restrict__ slice_off,
                                                          int *__restrict__ scol, double *__restrict__ sval,
                                                          double *__restrict__ b) {
  const int i = xcd_block() * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  if (i >= a.nlocal) {
    const int nslices = (a.nlocal + kSlice - 1) / kSlice;
    const int slice = i >> 6;
    if (sval && slice < nslices) {
      const long long off = slice_off[slice];
      const int w = (int)((slice_off[slice + 1] - off) >> 6);
      if (sval)
        for (int k = 0; k < w; ++k) { const long long p = sell_pos(off, lane, k); scol[p] = 0; sval[p] = 0.0; }
    }
    return;
  }
  const int dim = DIMT ? DIMT : T.dim, nt1 = T.ntypes + 1, dL = dim * (dim + 1) / 2;
  const bool antisym = FAM < 0 ? (a.antisym != 0) : (FAM != 0);
  const int it = a.type[i], ikind = T.kind[it];
  const long long off = sval ? slice_off[i >> 6] : 0;
  const int w = sval ? (int)((slice_off[(i >> 6) + 1] - off) >> 6) : 0;
  const int filt_i = MODE ? a.filt_i : KIND_FLUID, filt_j = MODE ? a.filt_j : KIND_ALL;
  const bool row_ok = MODE ? (ikind == filt_i) : ((ikind & filt_i) != 0);   // FilterMatchBinary | FilterBinary
  const int nf = MODE ? 1 : dim;                                            // right-hand-side columns
  const double alpha = a.dt;
  const double invrho = MODE ? 1.0 : 1.0 / a.rho[i];
  const double mi = MODE ? a.r2[i].x : a.nu[i] * a.rho[i];
  const double vscale = MODE == 2 ? 1.0 : -a.theta;                         // what multiplies the Laplacian in A
  const int jb = 0, je = T.nlen[i];
  int cnt = 0, pdiag = -1;
  const int ci_own = a.colmap[i];
  double diag_final;
  double wv[3] = {0, 0, 0}, gp[3] = {0, 0, 0};

  if (!row_ok) {
    for (int jj = jb; jj < je; ++jj) {
      const int j = neigh_at(T, i, jj - jb);
      double rij[3];
      if (pair_rsq(dim, a.x, i, j, rij) < T.cutsq[it * nt1 + a.type[j]]) {
        const int cj = a.colmap[j];
        if (T.sorted && pdiag < 0 && cj > ci_own) pdiag = cnt++;
        const long long p = sell_pos(off, lane, cnt++);
        if (sval) { scol[p] = cj; sval[p] = 0.0; }
      }
    }
    diag_final = 1.0;  // solid rows: unit diagonal, b unchanged (:114-117)
  } else {
    double G[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, L[6] = {1, 0, 1, 0, 0, 1};
    if (dim == 2) { G[0] = 1; G[1] = 0; G[2] = 0; G[3] = 1; L[0] = 1; L[1] = 0; L[2] = 1; }
    if (!antisym) {
      for (int k = 0; k < dim * dim; ++k) G[k] = a.Gc[(size_t)i * dim * dim + k];
      for (int k = 0; k < dL; ++k) L[k] = a.Lc[(size_t)i * dL + k];
    }
    const double vi = a.vfrac[i];
    double grad_m[3] = {0, 0, 0}, ci[3] = {0, 0, 0};
    double diag1 = 0.0;
    const double xi3[3] = {a.x[3 * (size_t)i], a.x[3 * (size_t)i + 1], a.x[3 * (size_t)i + 2]};
    const double pi = a.incremental ? a.p[i] : 0.0;
    for (int jj = jb; jj < je; ++jj) {  // sweep 1
      const int j = neigh_at(T, i, jj - jb);
      const double4 q1 = a.r1[j];
      const int2 q3 = a.r3[j];
      const int jt = q3.x, jkind = T.kind[jt];
      double rij[3];
      const double xj3[3] = {q1.x, q1.y, q1.z};
      const double rsq = rsq_nofma(dim, xi3, xj3, rij);  // the same arithmetic as pair_rsq / k_asm_count
      if (!(rsq < T.cutsq[it * nt1 + jt])) continue;
      const double mj = a.r2[j].x;
      double coeff = (row_ok && (ikind & filt_j)) ? 1.0 : 0.0;
      if (!(ikind & KIND_SOLID) && (jkind & KIND_SOLID)) {
        coeff = (row_ok && (jkind & filt_j)) ? 1.0 : 0.0;
        if (a.morris && coeff != 0.0)  // FunctorOuterLaplacianMatrix_MorrisHolmes (functor_boundary_morris_holmes.h:49-64)
          coeff = mirror_coeff(a.pnd, a.vfrac, a.safe, T.h[it * nt1 + jt], i, j, sqrt(T.cutsq[it * nt1 + jt]));
      }
      const double r = sqrt(rsq) + kEps;
      const double rinv = 1.0 / r;
      const double dwdr = kernel_dval(T.kernel, r, T.hinv[it * nt1 + jt], T.kdnorm[it * nt1 + jt]);
      double e[3] = {0, 0, 0};
      for (int k = 0; k < dim; ++k) e[k] = rij[k] * rinv;
      const double vfrac = antisym ? sqrt(vi * q1.w) : q1.w;
      const double vjtmp = dwdr * vfrac;
      for (int k2 = 0; k2 < dim; ++k2) {
        double gitmp = 0.0;
        if (antisym) gitmp = e[k2];  // G = I: the sum below gives exactly this
        else for (int k1 = 0; k1 < dim; ++k1) gitmp += G[k2 * dim + k1] * e[k1];
        if (ikind & jkind) grad_m[k2] += gitmp * vjtmp * (antisym ? (mi + mj) : (mj - mi));
      }
      double aij = 0.0;
      if (antisym) {  // L = I: only the squares survive, summed in the same order
        for (int k2 = 0; k2 < dim; ++k2) aij += e[k2] * e[k2];
      } else {
        for (int k2 = 0, op = 0; k2 < dim; ++k2)
          for (int k1 = 0; k1 < k2 + 1; ++k1, ++op) aij += L[op] * e[k1] * e[k2] * (k1 == k2 ? 1.0 : 2.0);
      }
      aij *= 2.0 * dwdr * vfrac;
      if (!antisym)
        for (int k = 0; k < dim; ++k) ci[k] += aij * e[k];
      aij *= mi * coeff * rinv;
      diag1 += aij;
      // gradient of p, filter (Fluid, Fluid) (functor_gradient.h:120-150)
      if (MODE == 0 && a.incremental && (ikind & KIND_FLUID) && (jkind & KIND_FLUID)) {
        const double vd = dwdr * rinv * vfrac;
        for (int k2 = 0; k2 < dim; ++k2) {
          double gitmp = 0.0;
          if (antisym) gitmp = rij[k2];
          else for (int k1 = 0; k1 < dim; ++k1) gitmp += G[k2 * dim + k1] * rij[k1];
          gp[k2] += gitmp * vd * (antisym ? (pi + a.p[j]) : (a.p[j] - pi));
        }
      }
    }
    double diag2 = 0.0;
    for (int jj = jb; jj < je; ++jj) {  // sweep 2
      const int j = neigh_at(T, i, jj - jb);
      const double4 q1 = a.r1[j];
      const int2 q3 = a.r3[j];
      const int jt = q3.x, jkind = T.kind[jt];
      double rij[3];
      const double xj3[3] = {q1.x, q1.y, q1.z};
      const double rsq = rsq_nofma(dim, xi3, xj3, rij);  // the same arithmetic as pair_rsq / k_asm_count
      if (!(rsq < T.cutsq[it * nt1 + jt])) continue;
      // the correction term uses the plain filter coefficient (:225-227); a_ij keeps the
      // mirror-weighted coefficient of the first sweep (:144-146)
      double coeff = (row_ok && (ikind & filt_j)) ? 1.0 : 0.0;
      if (!(ikind & KIND_SOLID) && (jkind & KIND_SOLID)) coeff = (row_ok && (jkind & filt_j)) ? 1.0 : 0.0;
      double coeff_a = coeff;
      if (a.morris && coeff != 0.0 && !(ikind & KIND_SOLID) && (jkind & KIND_SOLID))
        coeff_a = mirror_coeff(a.pnd, a.vfrac, a.safe, T.h[it * nt1 + jt], i, j, sqrt(T.cutsq[it * nt1 + jt]));
      const double r = sqrt(rsq) + kEps;
      const double rinv = 1.0 / r;
      const double dwdr = kernel_dval(T.kernel, r, T.hinv[it * nt1 + jt], T.kdnorm[it * nt1 + jt]);
      double e[3] = {0, 0, 0};
      for (int k = 0; k < dim; ++k) e[k] = rij[k] * rinv;
      const double vfrac = antisym ? sqrt(vi * q1.w) : q1.w;
      const double vjtmp = dwdr * vfrac;
      double aij = 0.0;
      if (antisym) {  // L = I: only the squares survive, summed in the same order
        for (int k2 = 0; k2 < dim; ++k2) aij += e[k2] * e[k2];
      } else {
        for (int k2 = 0, op = 0; k2 < dim; ++k2)
          for (int k1 = 0; k1 < k2 + 1; ++k1, ++op) aij += L[op] * e[k1] * e[k2] * (k1 == k2 ? 1.0 : 2.0);
      }
      aij *= 2.0 * dwdr * vfrac;
      aij *= mi * coeff_a * rinv;
      double bc = 0.0, bg = 0.0;
      for (int k2 = 0; k2 < dim; ++k2) {
        double bij = 0.0;
        if (antisym) bij = e[k2];
        else for (int k1 = 0; k1 < dim; ++k1) bij += G[k2 * dim + k1] * e[k1];
        bc += bij * ci[k2];
        bg += bij * grad_m[k2];
      }
      const double tmp = coeff * (mi * bc * vjtmp - bg * vjtmp);
      double v = -aij;
      v -= tmp;
      diag2 += tmp;
      const double aval = (v * alpha) * invrho;  // SumInto(alpha) then LeftScale(1/rho)
      {
        const double4 q2 = a.r2[j];
        const double vj3[3] = {q2.y, q2.z, q2.w};
        for (int k = 0; k < nf; ++k) wv[k] += aval * vj3[k];
      }
      const int cj = q3.y;
      if (T.sorted && pdiag < 0 && cj > ci_own) pdiag = cnt++;
      const long long p = sell_pos(off, lane, cnt++);
      if (sval) { scol[p] = cj; sval[p] = aval * vscale; }
    }
    const double dval = ((diag1 + diag2) * alpha) * invrho;
    if (MODE) wv[0] += dval * a.r2[i].y;
    else for (int k = 0; k < dim; ++k) wv[k] += dval * a.v[3 * (size_t)i + k];
    diag_final = MODE == 2 ? dval : 1.0 + dval * (-a.theta);
  }
  {
    if (pdiag < 0) pdiag = cnt++;
    const long long p = sell_pos(off, lane, pdiag);
    if (sval) { scol[p] = ci_own; sval[p] = diag_final; }
  }
  for (int k = cnt; k < w; ++k) {
    const long long p = sell_pos(off, lane, k);
    if (sval) { scol[p] = a.colmap[i]; sval[p] = 0.0; }
  }
  if (MODE == 1) {         // b = c (+ w on Fluid rows), functor_solute_transport.h:111-121
    double bk = a.r2[i].y;
    if (row_ok) bk += wv[0] * (1.0 - a.theta);
    b[i] = bk;
  } else if (MODE == 2) {  // b = phi on the buffers, 0 elsewhere, functor_applied_electric_potential.h:76-90
    b[i] = (ikind == KIND_BUFFER_DIRICHLET || ikind == KIND_BUFFER_NEUMANN) ? a.r2[i].y : 0.0;
  } else {
    for (int k = 0; k < dim; ++k) {
      double bk = a.v[3 * (size_t)i + k];
      if (ikind & filt_i) {
        bk += wv[k] * (1.0 - a.theta);
        bk += a.dt * (a.f[3 * (size_t)i + k] / a.rho[i] + a.g[k]);
        if (a.incremental) bk += a.dt * (-1.0 / a.rho[i] * gp[k]);
      }
      b[(size_t)k * a.lda + i] = bk;
    }
  }
}

// merge duplicate columns inside a row (periodic images sharing a tag in a
// box narrower than 2*cut; what FillComplete + SumIntoGlobalValues do):
// later duplicates are added into the first occurrence and turned into
// explicit zeros on the row's own column.  Only run for small problems.
__global__ void k_sell_merge_duplicates(int nrow, const int *__restrict__ rowlen, const long long *__restrict__ slice_off,
                                        int *__restrict__ scol, double *__restrict__ sval, int *__restrict__ newlen) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= nrow) return;
  const long long off = slice_off[row >> 6];
  const int lane = row & 63, len = rowlen[row];
  int out = 0;
  for (int k = 0; k < len; ++k) {
    const long long pk = sell_pos(off, lane, k);
    const int c = scol[pk];
    const double v = sval[pk];
    int hit = -1;
    for (int q = 0; q < out; ++q)
      if (scol[sell_pos(off, lane, q)] == c) { hit = q; break; }
    if (hit >= 0) {
      sval[sell_pos(off, lane, hit)] += v;
    } else {
      const long long po = sell_pos(off, lane, out++);
      scol[po] = c;
      sval[po] = v;
    }
  }
  for (int k = out; k < len; ++k) {
    const long long p = sell_pos(off, lane, k);
    scol[p] = row;
    sval[p] = 0.0;
  }
  newlen[row] = out;
}

// nnz = sum of the row lengths, reduced on the device (the 4 MB copy of a million row lengths and the host loop over
// them cost 0.4 ms per assembly)
__global__ __launch_bounds__(kBlock) void k_sum_rowlen(int n, const int *__restrict__ len, unsigned long long *__restrict__ out) {
  unsigned long long s = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) s += (unsigned long long)len[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}
inline int sell_sum_rowlen(isph_ctx *ctx, const Sell &S, long long *nnz) {
  DevTmp<unsigned long long> acc;
  ISPH_CHECK(acc.reserve(1));
  ISPH_CHECK_HIP(hipMemsetAsync(acc.p, 0, sizeof(unsigned long long), ctx->stream));
  if (S.nrow > 0)
    hipLaunchKernelGGL(k_sum_rowlen, dim3(std::min(256, (S.nrow + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, S.nrow,
                       (const int *)S.rowlen.p, acc.p);
  unsigned long long h = 0;
  ISPH_CHECK_HIP(hipMemcpyAsync(&h, acc.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  *nnz = (long long)h;
  return ISPH_SUCCESS;
}

struct StagedParticles {
  DevBuf<double> x, vfrac, Gc, Lc, h, cutsq, rho, vstar, pnd, hinv, knorm, kdnorm, invrho, normal;
  DevBuf<int> type, kind, nptr, nidx, colmap, first;
  DevBuf<long long> nptr64;
  void release() {
    nptr64.release();
    x.release(); vfrac.release(); Gc.release(); Lc.release(); h.release(); cutsq.release(); rho.release();
    vstar.release(); pnd.release(); hinv.release(); knorm.release(); kdnorm.release(); invrho.release(); normal.release(); type.release(); kind.release(); nptr.release(); nidx.release(); colmap.release(); first.release();
  }
};


struct NeighEll {
  DevBuf<long long> off;
  DevBuf<int> idx, len, sorted;
  void release() { off.release(); idx.release(); len.release(); sorted.release(); }
};

// the flattened neighbour list's offsets as the caller gave them: 32-bit (LAMMPS-like numneigh/firstneigh, up to 2^31
// list entries) or 64-bit (isph_particles::neigh_ptr64: BASELINE configs[4] has 4 M x 748 = 3e9 entries)
struct NeighPtr {
  const int *p32 = nullptr;
  const long long *p64 = nullptr;
};

// builds the lane-interleaved neighbour list and hooks it into T
template <class OFF>
inline int build_neigh_ell_t(isph_ctx *ctx, int n, const OFF *dnptr, const int *dnidx, NeighEll &E, AsmTables &T,
                             const int *dcolmap) {
  const int nslices = (n + kSlice - 1) / kSlice;
  // a held list (isph_ctx_hold_neighbours): the layout of an earlier call with the same arrays is still good
  // the library's own row numbering (order.hpp; set by the ordered assembly entry points around this call): rows are
  // read through rowsrc, list entries through idmap, dcolmap is the column map IN that numbering -- a temporary, so the
  // held layout is keyed by the caller's column map and the order object instead
  const int *rowsrc = ctx->nmap.rowsrc, *idmap = ctx->nmap.idmap;
  const int *const colmap_key = ctx->nmap.order ? (dcolmap ? ctx->nmap.colmap_key : nullptr) : dcolmap;
  isph_neigh_layout *slot = (ctx->neigh_hold && n > 0) ? &ctx->neigh_cache[dcolmap ? 1 : 0] : nullptr;
  if (slot && slot->n == n && slot->nptr == (const void *)dnptr && slot->nidx == dnidx && slot->colmap == colmap_key &&
      slot->order == ctx->nmap.order) {
    T.nlen = slot->len.p; T.noff = slot->off.p; T.nt = slot->idx.p; T.sorted = slot->is_sorted;
    return ISPH_SUCCESS;
  }
  const int *const nidx_key = dnidx;
  ISPH_CHECK(E.len.reserve((size_t)(n > 0 ? n : 1)));
  ISPH_CHECK(E.off.reserve((size_t)nslices + 1));
  T.nlen = E.len.p;
  if (n == 0) { T.noff = E.off.p; T.nt = nullptr; return ISPH_SUCCESS; }
  const int grid = (n + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(k_numneigh<OFF>, dim3(grid), dim3(kBlock), 0, ctx->stream, n, dnptr, E.len.p, rowsrc);
  hipLaunchKernelGGL(k_slicew_from_rowlen, dim3(grid), dim3(kBlock), 0, ctx->stream, n, E.len.p, E.off.p);
  hipLaunchKernelGGL(k_exclusive_scan_ll, dim3(1), dim3(1024), 0, ctx->stream, nslices, E.off.p, E.off.p);
  // ONE host round trip for the sizes: the slice offsets (their last entry is the total) when the lists will be ordered
  // by column -- the widest slice decides whether the sort's LDS buffer fits --, the total alone otherwise
  long long total = 0;
  std::vector<long long> so;
  if (dcolmap) {
    so.resize((size_t)nslices + 1);
    ISPH_CHECK_HIP(hipMemcpyAsync(so.data(), E.off.p, sizeof(long long) * so.size(), hipMemcpyDeviceToHost, ctx->stream));
  } else {
    ISPH_CHECK_HIP(hipMemcpyAsync(&total, E.off.p + nslices, sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
  }
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (dcolmap) total = so[(size_t)nslices];
  ISPH_CHECK(E.idx.reserve((size_t)(total > 0 ? total : 1)));
  T.sorted = 0;
  if (dcolmap) {  // order the lists by matrix column when every row fits the sort's LDS buffer
    long long wmax = 0;
    for (int s = 0; s < nslices; ++s) wmax = std::max(wmax, (so[(size_t)s + 1] - so[(size_t)s]) >> 6);
    if (wmax <= kNeighSortCap) {
      ISPH_CHECK(E.sorted.reserve((size_t)(total > 0 ? total : 1)));
      // with the library's own numbering the keys come from the column map indexed by the caller's particle
      const int *keymap = ctx->nmap.order ? ctx->nmap.colkey : dcolmap;
      long long need = wmax + 2;  // two pad keys behind the longest row (pair reads of the rank loop)
      if (wmax > 128) {           // long rows go through a bitonic network over the next power of two (>= 256)
        long long p2 = 256;
        while (p2 < wmax) p2 <<= 1;
        need = std::max(need, p2);
      }
      const int stride = (int)((need + 1) & ~1LL);
      const size_t lds = sizeof(unsigned long long) * (size_t)stride * (kBlock / 64);
      ISPH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_neigh_sort<OFF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k_neigh_sort<OFF>, dim3((n + 3) / 4), dim3(kBlock), lds, ctx->stream, n, dnptr, dnidx, keymap, E.sorted.p,
                         rowsrc, idmap, stride);
      dnidx = E.sorted.p;
      idmap = nullptr;  // the sorted copy holds mapped entries already
      T.sorted = 1;
    }
  }
  hipLaunchKernelGGL(k_neigh_transpose<OFF>, dim3((nslices + 3) / 4), dim3(kBlock), 0, ctx->stream, n, dnptr, dnidx, E.off.p,
                     E.idx.p, rowsrc, idmap);
  ISPH_CHECK_HIP(hipGetLastError());
  T.noff = E.off.p;
  T.nt = E.idx.p;
  if (slot) {   // the buffers move into the context (T keeps pointing at them); E takes the slot's old ones and frees them
    std::swap(slot->off, E.off); std::swap(slot->idx, E.idx); std::swap(slot->len, E.len); std::swap(slot->sorted, E.sorted);
    slot->n = n; slot->nptr = (const void *)dnptr; slot->nidx = nidx_key; slot->colmap = colmap_key; slot->is_sorted = T.sorted;
    slot->order = ctx->nmap.order;
  }
  return ISPH_SUCCESS;
}
inline int build_neigh_ell(isph_ctx *ctx, int n, const NeighPtr &np, const int *dnidx, NeighEll &E, AsmTables &T,
                           const int *dcolmap = nullptr) {
  if (np.p64) return build_neigh_ell_t<long long>(ctx, n, np.p64, dnidx, E, T, dcolmap);
  return build_neigh_ell_t<int>(ctx, n, np.p32, dnidx, E, T, dcolmap);
}

template <class T>
inline int stage(isph_ctx *ctx, const T *src, size_t n, int on_device, DevBuf<T> &tmp, const T **out) {
  if (!src) { *out = nullptr; return ISPH_SUCCESS; }
  if (on_device) { *out = src; return ISPH_SUCCESS; }
  ISPH_REQUIRE(!is_device_pointer(src), "device pointer passed with on_device = 0");
  ISPH_CHECK(tmp.reserve(n > 0 ? n : 1));
  ISPH_CHECK_HIP(hipMemcpyAsync(tmp.p, src, sizeof(T) * n, hipMemcpyHostToDevice, ctx->stream));
  *out = tmp.p;
  return ISPH_SUCCESS;
}

// stages the neighbour-list offsets (neigh_ptr64 wins over neigh_ptr) and returns the number of list entries of a HOST list
struct StagedParticles;
inline int stage_neigh_ptr(isph_ctx *ctx, const isph_particles *P, int n, int on_device, DevBuf<int> &b32,
                           DevBuf<long long> &b64, NeighPtr &np, long long *nnb) {
  ISPH_REQUIRE(P->neigh_ptr || P->neigh_ptr64, "neighbour list offsets missing");
  if (P->neigh_ptr64) {
    ISPH_CHECK(stage(ctx, P->neigh_ptr64, (size_t)n + 1, on_device, b64, &np.p64));
    // (device lists: the count is only needed to stage a host list, and the layout builder reads its own sizes --
    // no host round trip here)
    *nnb = on_device ? 0 : P->neigh_ptr64[n];
  } else {
    ISPH_CHECK(stage(ctx, P->neigh_ptr, (size_t)n + 1, on_device, b32, &np.p32));
    *nnb = on_device ? 0 : P->neigh_ptr[n];
  }
  ISPH_REQUIRE(*nnb >= 0, "negative neighbour count");
  return ISPH_SUCCESS;
}

inline int stage_tables(isph_ctx *ctx, const isph_particles *P, StagedParticles &S, AsmTables &T) {
  (void)S;
  const size_t nt1 = (size_t)P->ntypes + 1;
  ISPH_REQUIRE(P->kind && P->h && P->cutsq, "kind/h/cutsq tables are required (host pointers)");
  T.ntypes = P->ntypes; T.kernel = P->kernel; T.dim = P->dim;
  T.noff = nullptr; T.nt = nullptr; T.sorted = 0;
  isph_table_cache &C = ctx->tables;
  const bool same = C.ntypes == P->ntypes && C.kernel == P->kernel && C.dim == P->dim && C.kind.size() == nt1 &&
                    memcmp(C.kind.data(), P->kind, sizeof(int) * nt1) == 0 &&
                    memcmp(C.h.data(), P->h, sizeof(double) * nt1 * nt1) == 0 &&
                    memcmp(C.cutsq.data(), P->cutsq, sizeof(double) * nt1 * nt1) == 0;
  if (!same) {   // the tables of the context (isph_table_cache): staged when they change, which a run does not do
    C.ntypes = -1;
    const int *dk; const double *dh, *dc, *d1, *d2, *d3;
    ISPH_CHECK(stage(ctx, P->kind, nt1, 0, C.dkind, &dk));
    ISPH_CHECK(stage(ctx, P->h, nt1 * nt1, 0, C.dh, &dh));
    ISPH_CHECK(stage(ctx, P->cutsq, nt1 * nt1, 0, C.dcutsq, &dc));
    std::vector<double> hi(nt1 * nt1), kn(nt1 * nt1), kd(nt1 * nt1);
    for (size_t k = 0; k < nt1 * nt1; ++k) {
      const double hh = P->h[k];
      hi[k] = hh != 0.0 ? 1.0 / hh : 0.0;
      kn[k] = hh != 0.0 ? kernel_norm(P->kernel, P->dim, hh) : 0.0;
      kd[k] = hh != 0.0 ? kn[k] / hh : 0.0;
    }
    ISPH_CHECK(stage(ctx, hi.data(), nt1 * nt1, 0, C.dhinv, &d1));
    ISPH_CHECK(stage(ctx, kn.data(), nt1 * nt1, 0, C.dknorm, &d2));
    ISPH_CHECK(stage(ctx, kd.data(), nt1 * nt1, 0, C.dkdnorm, &d3));
    ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));  // hi/kn/kd are stack-lived
    C.kind.assign(P->kind, P->kind + nt1);
    C.h.assign(P->h, P->h + nt1 * nt1);
    C.cutsq.assign(P->cutsq, P->cutsq + nt1 * nt1);
    C.ntypes = P->ntypes; C.kernel = P->kernel; C.dim = P->dim;
  }
  T.kind = C.dkind.p; T.h = C.dh.p; T.cutsq = C.dcutsq.p;
  T.hinv = C.dhinv.p; T.knorm = C.dknorm.p; T.kdnorm = C.dkdnorm.p;
  return ISPH_SUCCESS;
}

inline int compute_volumes(isph_ctx *ctx, const isph_particles *P, double *vfrac_out, int on_device, bool pnd = false) {
  ISPH_REQUIRE(P->dim == 2 || P->dim == 3, "dim must be 2 or 3");
  ISPH_REQUIRE(P->x && P->type && (P->neigh_ptr || P->neigh_ptr64) && P->neigh_idx, "particle arrays missing");
  StagedParticles S;
  AsmTables T;
  int rc = stage_tables(ctx, P, S, T);
  const double *dx = nullptr; const int *dt = nullptr, *dp = nullptr, *di = nullptr;
  long long nnb = 0;
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->x, (size_t)P->nall * 3, on_device, S.x, &dx);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->type, (size_t)P->nall, on_device, S.type, &dt);
  NeighPtr np;
  if (rc == ISPH_SUCCESS) rc = stage_neigh_ptr(ctx, P, P->nlocal, on_device, S.nptr, S.nptr64, np, &nnb);
  dp = np.p32;
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->neigh_idx, (size_t)nnb, on_device, S.nidx, &di);
  NeighEll E;
  if (rc == ISPH_SUCCESS) rc = build_neigh_ell(ctx, P->nlocal, np, di, E, T);
  DevTmp<double> out;
  double *dout = vfrac_out;
  if (rc == ISPH_SUCCESS && !on_device) { rc = out.reserve((size_t)(P->nlocal > 0 ? P->nlocal : 1)); dout = out.p; }
  if (rc == ISPH_SUCCESS && P->nlocal > 0) {
    if (pnd)
      hipLaunchKernelGGL(k_pnd, dim3(xcd_grid((P->nlocal + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, T, P->nlocal, dx,
                         dt, dout);
    else
      hipLaunchKernelGGL(k_volumes, dim3(xcd_grid((P->nlocal + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, T, P->nlocal, dx,
                         dt, dp, di, dout);
    if (!on_device &&
        hipMemcpyAsync(vfrac_out, dout, sizeof(double) * (size_t)P->nlocal, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
      rc = fail("copy failed", __FILE__, __LINE__);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess)
      rc = fail("volume kernel failed", __FILE__, __LINE__);
  }
  S.release();
  E.release();
  out.release();
  return rc;
}

inline int assemble_poisson(isph_ctx *ctx, const isph_particles *P, int antisym, double dt, const double *rho,
                            const double *vstar, int singular_mode, int is_rank0, int ncol, isph_mat **A_out,
                            double *b_out, int on_device) {
  ISPH_REQUIRE(P->dim == 2 || P->dim == 3, "dim must be 2 or 3");
  ISPH_REQUIRE(P->x && P->type && (P->neigh_ptr || P->neigh_ptr64) && P->neigh_idx && P->colmap, "particle arrays missing");
  ISPH_REQUIRE(antisym || (P->Gc && P->Lc), "Symmetric family needs Gc and Lc");
  ISPH_REQUIRE(P->vfrac, "vfrac is required (isph_compute_volumes + forward comm first)");
  ISPH_REQUIRE(singular_mode >= 0 && singular_mode <= 3, "bad singular mode");
  ISPH_REQUIRE(ncol >= P->nlocal, "ncol < nlocal");
  const int n = P->nlocal, dim = P->dim, dL = dim * (dim + 1) / 2;
  StagedParticles S;
  AsmTables T;
  PoissonArgs a;
  memset(&a, 0, sizeof(a));
  isph_mat *A = new isph_mat();
  DevTmp<double> bdev;
  DevTmp<int> newlen;
  int rc = stage_tables(ctx, P, S, T);
  long long nnb = 0;
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->x, (size_t)P->nall * 3, on_device, S.x, &a.x);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->type, (size_t)P->nall, on_device, S.type, &a.type);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->colmap, (size_t)P->nall, on_device, S.colmap, &a.colmap);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->vfrac, (size_t)P->nall, on_device, S.vfrac, &a.vfrac);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->Gc, (size_t)P->nlocal * dim * dim, on_device, S.Gc, &a.Gc);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->Lc, (size_t)P->nlocal * dL, on_device, S.Lc, &a.Lc);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, rho, (size_t)P->nall, on_device, S.rho, &a.rho);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, vstar, (size_t)P->nall * 3, on_device, S.vstar, &a.vstar);
  if (rc == ISPH_SUCCESS && P->morris_holmes) {
    if (!P->pnd) rc = fail("MorrisHolmes needs pnd", __FILE__, __LINE__);
    else rc = stage(ctx, P->pnd, (size_t)P->nall, on_device, S.pnd, &a.pnd);
  }
  if (rc == ISPH_SUCCESS && P->normal && singular_mode != 0) {
    if (!P->Gc) rc = fail("wall normals need Gc (gradient-operator rows use G_i)", __FILE__, __LINE__);
    else rc = stage(ctx, P->normal, (size_t)P->nall * 3, on_device, S.normal, &a.normal);
    if (rc == ISPH_SUCCESS && !a.Gc) rc = stage(ctx, P->Gc, (size_t)P->nlocal * dim * dim, on_device, S.Gc, &a.Gc);
  }
  NeighPtr np;
  if (rc == ISPH_SUCCESS) rc = stage_neigh_ptr(ctx, P, n, on_device, S.nptr, S.nptr64, np, &nnb);
  a.nptr = np.p32;
  if (rc == ISPH_SUCCESS) {
    if (!on_device) {
      // host-side shape check before any kernel indexes with these
      for (long long k = 0; k < nnb && rc == ISPH_SUCCESS; ++k)
        if (P->neigh_idx[k] < 0 || P->neigh_idx[k] >= P->nall) rc = fail("neighbour index out of range", __FILE__, __LINE__);
      for (int j = 0; j < P->nall && rc == ISPH_SUCCESS; ++j)
        if (P->colmap[j] < 0 || P->colmap[j] >= ncol) rc = fail("colmap entry out of range", __FILE__, __LINE__);
    }
  }
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->neigh_idx, (size_t)nnb, on_device, S.nidx, &a.nidx);
  NeighEll E;
  if (rc == ISPH_SUCCESS) rc = build_neigh_ell(ctx, n, np, a.nidx, E, T, a.colmap);
  // kinds present: refuse what this build does not restate
  if (rc == ISPH_SUCCESS)
    for (int t = 1; t <= P->ntypes; ++t)
      if (P->kind[t] != KIND_FLUID && P->kind[t] != KIND_SOLID) rc = fail("only fluid/solid particle kinds are supported", __FILE__, __LINE__);

  Sell &M = A->S;
  M.nrow = n; M.ncol = ncol; M.nslices = (n + kSlice - 1) / kSlice;
  if (rc == ISPH_SUCCESS) rc = M.rowlen.reserve((size_t)(n > 0 ? n : 1));
  if (rc == ISPH_SUCCESS) rc = M.slice_off.reserve((size_t)M.nslices + 1);
  double *db = b_out;
  if (rc == ISPH_SUCCESS && !on_device) { rc = bdev.reserve((size_t)(n > 0 ? n : 1)); db = bdev.p; }
  if (rc == ISPH_SUCCESS && n > 0) {
    const int grid = (n + kBlock - 1) / kBlock;
    DevBuf<double4> pk1, pk2;
    DevBuf<int2> pk3;
    struct PackRelease { DevBuf<double4> &a, &b; DevBuf<int2> &c; ~PackRelease() { a.release(); b.release(); c.release(); } } pack_release{pk1, pk2, pk3};
    a.nlocal = n; a.antisym = antisym; a.singular_mode = singular_mode; a.dt = dt;
    a.morris = P->morris_holmes ? 1 : 0; a.safe = P->morris_safe_coeff;
    a.solid_normal_diag = P->solid_normal_diag;
    rc = S.invrho.reserve((size_t)P->nall);
    if (rc == ISPH_SUCCESS) rc = pk1.reserve((size_t)P->nall);
    if (rc == ISPH_SUCCESS) rc = pk2.reserve((size_t)P->nall);
    if (rc == ISPH_SUCCESS) rc = pk3.reserve((size_t)P->nall);
    if (rc == ISPH_SUCCESS) {
      hipLaunchKernelGGL(k_reciprocal, dim3((P->nall + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, P->nall, a.rho,
                         S.invrho.p);
      a.invrho = S.invrho.p;
      hipLaunchKernelGGL(k_pack_particles, dim3((P->nall + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, P->nall, a.x,
                         a.vfrac, a.invrho, a.vstar, a.type, a.colmap, pk1.p, pk2.p, pk3.p, (T.dim == 3 && a.antisym) ? 1 : 0);
      a.r1 = pk1.p; a.r2 = pk2.p; a.r3 = pk3.p;
      // (the count reads x and type directly: 24 + 4 bytes per neighbour; from the 32-byte records it ran 13 % slower)
      hipLaunchKernelGGL(k_asm_count, dim3(xcd_grid(grid)), dim3(kBlock), 0, ctx->stream, T, n, a.x, a.type, a.nptr, a.nidx, M.rowlen.p);
      hipLaunchKernelGGL(k_slicew_from_rowlen, dim3(grid), dim3(kBlock), 0, ctx->stream, n, M.rowlen.p, M.slice_off.p);
      rc = sell_finalize_offsets(ctx, M);
    }
    if (rc == ISPH_SUCCESS) {
      a.pin_enabled = (is_rank0 && singular_mode >= 2) ? 1 : 0;
      rc = S.first.reserve(1);
      if (rc == ISPH_SUCCESS) {
        const int big = 0x7fffffff;
        if (hipMemcpyAsync(S.first.p, &big, sizeof(int), hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
          rc = fail("copy failed", __FILE__, __LINE__);
        if (a.pin_enabled) {
          const RowOrder *ord = static_cast<const RowOrder *>(ctx->nmap.order);
          hipLaunchKernelGGL(k_first_fluid, dim3(grid), dim3(kBlock), 0, ctx->stream, n, a.type, T.kind, S.first.p,
                             ord ? (const int *)ord->perm.p : (const int *)nullptr);
          if (ord) hipLaunchKernelGGL(k_first_to_internal, dim3(1), dim3(64), 0, ctx->stream, n, (const int *)ord->iperm.p, S.first.p);
        }
        a.first_fluid = S.first.p;
      }
    }
    if (rc == ISPH_SUCCESS) {
      const int gridp = M.nslices * kSlice / kBlock + ((M.nslices * kSlice) % kBlock ? 1 : 0);
      if (T.dim == 3 && a.antisym)
        hipLaunchKernelGGL((k_asm_poisson<3, 1>), dim3(xcd_grid(gridp)), dim3(kBlock), 0, ctx->stream, T, a, M.slice_off.p, M.col.p, M.val.p, db);
      else if (T.dim == 3)  // 3-D, Symmetric (corrected) family: G_i / L_i loops with compile-time bounds
        hipLaunchKernelGGL((k_asm_poisson<3, 0>), dim3(xcd_grid(gridp)), dim3(kBlock), 0, ctx->stream, T, a, M.slice_off.p, M.col.p, M.val.p, db);
      else
        hipLaunchKernelGGL((k_asm_poisson<0, -1>), dim3(xcd_grid(gridp)), dim3(kBlock), 0, ctx->stream, T, a, M.slice_off.p, M.col.p, M.val.p, db);
      if (n <= 32768) {  // tiny boxes may see the same tag twice in a row
        rc = newlen.reserve((size_t)n);
        if (rc == ISPH_SUCCESS) {
          hipLaunchKernelGGL(k_sell_merge_duplicates, dim3(grid), dim3(kBlock), 0, ctx->stream, n, M.rowlen.p, M.slice_off.p,
                             M.col.p, M.val.p, newlen.p);
          if (hipMemcpyAsync(M.rowlen.p, newlen.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
            rc = fail("copy failed", __FILE__, __LINE__);
        }
      }
      // rows come out column-sorted when the neighbour lists were ordered; merged duplicates break that order
      // (sorted lists, no merge: the widest slice is known since sell_finalize_offsets)
      if (rc == ISPH_SUCCESS && !(T.sorted && n > 32768)) rc = sell_sort_rows(ctx, M);  // columns ascending, like Epetra after FillComplete
      if (rc == ISPH_SUCCESS && !on_device &&
          hipMemcpyAsync(b_out, db, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        rc = fail("copy failed", __FILE__, __LINE__);
      // nnz = sum of the row lengths (after the merge), read back with the synchronisation that ends the assembly
      DevTmp<unsigned long long> acc;
      unsigned long long hacc = 0;
      if (rc == ISPH_SUCCESS) rc = acc.reserve(1);
      if (rc == ISPH_SUCCESS) {
        if (hipMemsetAsync(acc.p, 0, sizeof(unsigned long long), ctx->stream) != hipSuccess) rc = fail("memset failed", __FILE__, __LINE__);
        hipLaunchKernelGGL(k_sum_rowlen, dim3(std::min(256, (n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, n,
                           (const int *)M.rowlen.p, acc.p);
        if (hipMemcpyAsync(&hacc, acc.p, sizeof(hacc), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = fail("copy failed", __FILE__, __LINE__);
      }
      if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess)
        rc = fail("assembly kernel failed", __FILE__, __LINE__);
      M.nnz = (long long)hacc;
    }
  }
  S.release();
  E.release();
  bdev.release();
  newlen.release();
  if (rc != ISPH_SUCCESS) { A->S.release(); delete A; return rc; }
  *A_out = A;
  return ISPH_SUCCESS;
}

inline int assemble_helmholtz(isph_ctx *ctx, const isph_particles *P, int antisym, double dt, double theta,
                              const double *nu, const double *rho, const double *pres, const double *force,
                              const double *gvec, int incremental, const double *vel, int ncol, isph_mat **A_out,
                              double *b_out, int lda, int on_device, int mode = 0, const double *field = nullptr,
                              const double *material = nullptr) {
  // mode 1 / 2: the scalar callers (HelmholtzArgs); dt then carries the Laplacian's alpha, field = c or phi [nall]
  ISPH_REQUIRE(mode == 0 || (field && A_out), "scalar assembly needs the field and a matrix handle");
  ISPH_REQUIRE(P->dim == 2 || P->dim == 3, "dim must be 2 or 3");
  ISPH_REQUIRE(P->x && P->type && (P->neigh_ptr || P->neigh_ptr64) && P->neigh_idx && P->colmap, "particle arrays missing");
  ISPH_REQUIRE(antisym || (P->Gc && P->Lc), "Symmetric family needs Gc and Lc");
  ISPH_REQUIRE(P->vfrac, "vfrac is required (isph_compute_volumes + forward comm first)");
  ISPH_REQUIRE(ncol >= P->nlocal && lda >= P->nlocal, "need ncol >= nlocal and lda >= nlocal");
  const int n = P->nlocal, dim = P->dim, dL = dim * (dim + 1) / 2;
  StagedParticles S;
  DevTmp<double> snu, sp, sf, sv;
  AsmTables T;
  HelmholtzArgs a;
  memset(&a, 0, sizeof(a));
  const bool rhs_only = A_out == nullptr;  // theta = 0 callers only need b (the reference then copies b into x)
  isph_mat *A = new isph_mat();
  DevTmp<double> bdev;
  DevTmp<int> newlen;
  int rc = stage_tables(ctx, P, S, T);
  long long nnb = 0;
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->x, (size_t)P->nall * 3, on_device, S.x, &a.x);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->type, (size_t)P->nall, on_device, S.type, &a.type);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->colmap, (size_t)P->nall, on_device, S.colmap, &a.colmap);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->vfrac, (size_t)P->nall, on_device, S.vfrac, &a.vfrac);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->Gc, (size_t)P->nlocal * dim * dim, on_device, S.Gc, &a.Gc);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->Lc, (size_t)P->nlocal * dL, on_device, S.Lc, &a.Lc);
  if (mode == 0) {
    if (rc == ISPH_SUCCESS) rc = stage(ctx, rho, (size_t)P->nall, on_device, S.rho, &a.rho);
    if (rc == ISPH_SUCCESS) rc = stage(ctx, nu, (size_t)P->nall, on_device, snu, &a.nu);
    if (rc == ISPH_SUCCESS) rc = stage(ctx, pres, (size_t)P->nall, on_device, sp, &a.p);
    if (rc == ISPH_SUCCESS) rc = stage(ctx, force, (size_t)P->nall * 3, on_device, sf, &a.f);
    if (rc == ISPH_SUCCESS) rc = stage(ctx, vel, (size_t)P->nall * 3, on_device, sv, &a.v);
  } else {
    if (rc == ISPH_SUCCESS) rc = stage(ctx, material, (size_t)P->nall, on_device, snu, &a.nu);   // NULL stays NULL
    if (rc == ISPH_SUCCESS) rc = stage(ctx, field, (size_t)P->nall, on_device, sv, &a.v);
  }
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
  if (rc == ISPH_SUCCESS) rc = build_neigh_ell(ctx, n, np, a.nidx, E, T, A_out ? a.colmap : nullptr);
  if (rc == ISPH_SUCCESS)
    for (int t = 1; t <= P->ntypes; ++t)
      if (P->kind[t] != KIND_FLUID && P->kind[t] != KIND_SOLID &&
          !(mode != 0 && (P->kind[t] == KIND_BUFFER_DIRICHLET || P->kind[t] == KIND_BUFFER_NEUMANN)))
        rc = fail(mode ? "only fluid/solid/buffer particle kinds are supported" : "only fluid/solid particle kinds are supported", __FILE__, __LINE__);
  Sell &M = A->S;
  M.nrow = n; M.ncol = ncol; M.nslices = (n + kSlice - 1) / kSlice;
  if (rc == ISPH_SUCCESS) rc = M.rowlen.reserve((size_t)(n > 0 ? n : 1));
  if (rc == ISPH_SUCCESS) rc = M.slice_off.reserve((size_t)M.nslices + 1);
  double *db = b_out;
  const int nrhs = mode ? 1 : dim;
  if (rc == ISPH_SUCCESS && !on_device) { rc = bdev.reserve((size_t)lda * nrhs); db = bdev.p; }
  if (rc == ISPH_SUCCESS && n > 0) {
    const int grid = (n + kBlock - 1) / kBlock;
    if (!rhs_only) {
      hipLaunchKernelGGL(k_asm_count, dim3(xcd_grid(grid)), dim3(kBlock), 0, ctx->stream, T, n, a.x, a.type, a.nptr, a.nidx, M.rowlen.p);
      hipLaunchKernelGGL(k_slicew_from_rowlen, dim3(grid), dim3(kBlock), 0, ctx->stream, n, M.rowlen.p, M.slice_off.p);
      rc = sell_finalize_offsets(ctx, M);
    }
    if (rc == ISPH_SUCCESS) {
      a.nlocal = n; a.antisym = antisym; a.incremental = incremental; a.lda = lda; a.dt = dt; a.theta = theta;
      a.morris = P->morris_holmes ? 1 : 0; a.safe = P->morris_safe_coeff;
      for (int k = 0; k < 3; ++k) a.g[k] = gvec ? gvec[k] : 0.0;
      const int gridp = M.nslices * kSlice / kBlock + ((M.nslices * kSlice) % kBlock ? 1 : 0);
      DevBuf<double4> pk1, pk2;
      DevBuf<int2> pk3;
      struct PackRelease { DevBuf<double4> &a, &b; DevBuf<int2> &c; ~PackRelease() { a.release(); b.release(); c.release(); } } pack_release{pk1, pk2, pk3};
      rc = pk1.reserve((size_t)P->nall);
      if (rc == ISPH_SUCCESS) rc = pk2.reserve((size_t)P->nall);
      if (rc == ISPH_SUCCESS) rc = pk3.reserve((size_t)P->nall);
      if (rc == ISPH_SUCCESS && mode != 0) {
        hipLaunchKernelGGL(k_pack_particles_scalar, dim3((P->nall + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream,
                           P->nall, a.x, a.vfrac, a.nu, a.v, a.type, a.colmap, pk1.p, pk2.p, pk3.p);
        a.r1 = pk1.p; a.r2 = pk2.p; a.r3 = pk3.p;
        a.filt_i = KIND_FLUID;
        a.filt_j = mode == 1 ? KIND_FLUID - KIND_BUFFER_NEUMANN : KIND_FLUID;
        if (mode == 1)
          hipLaunchKernelGGL((k_asm_helmholtz<0, -1, 1>), dim3(xcd_grid(gridp)), dim3(kBlock), 0, ctx->stream, T, a, M.slice_off.p, M.col.p, M.val.p, db);
        else
          hipLaunchKernelGGL((k_asm_helmholtz<0, -1, 2>), dim3(xcd_grid(gridp)), dim3(kBlock), 0, ctx->stream, T, a, M.slice_off.p, M.col.p, M.val.p, db);
      } else if (rc == ISPH_SUCCESS) {
        hipLaunchKernelGGL(k_pack_particles_helmholtz, dim3((P->nall + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream,
                           P->nall, a.x, a.vfrac, a.nu, a.rho, a.v, a.type, a.colmap, pk1.p, pk2.p, pk3.p);
        a.r1 = pk1.p; a.r2 = pk2.p; a.r3 = pk3.p;
        int *mcol = rhs_only ? (int *)nullptr : M.col.p;
        double *mval = rhs_only ? (double *)nullptr : M.val.p;
        if (T.dim == 3 && antisym)
          hipLaunchKernelGGL((k_asm_helmholtz<3, 1>), dim3(xcd_grid(gridp)), dim3(kBlock), 0, ctx->stream, T, a, M.slice_off.p, mcol, mval, db);
        else if (T.dim == 3)
          hipLaunchKernelGGL((k_asm_helmholtz<3, 0>), dim3(xcd_grid(gridp)), dim3(kBlock), 0, ctx->stream, T, a, M.slice_off.p, mcol, mval, db);
        else
          hipLaunchKernelGGL((k_asm_helmholtz<0, -1>), dim3(xcd_grid(gridp)), dim3(kBlock), 0, ctx->stream, T, a, M.slice_off.p, mcol, mval, db);
      }
      if (!rhs_only && n <= 32768) {
        rc = newlen.reserve((size_t)n);
        if (rc == ISPH_SUCCESS) {
          hipLaunchKernelGGL(k_sell_merge_duplicates, dim3(grid), dim3(kBlock), 0, ctx->stream, n, M.rowlen.p, M.slice_off.p,
                             M.col.p, M.val.p, newlen.p);
          if (hipMemcpyAsync(M.rowlen.p, newlen.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
            rc = fail("copy failed", __FILE__, __LINE__);
        }
      }
      // rows come out column-sorted when the neighbour lists were ordered; merged duplicates break that order
      if (rc == ISPH_SUCCESS && !rhs_only) rc = (T.sorted && n > 32768) ? sell_set_wmax(ctx, M) : sell_sort_rows(ctx, M);
      if (rc == ISPH_SUCCESS && !on_device &&
          hipMemcpyAsync(b_out, db, sizeof(double) * (size_t)lda * nrhs, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        rc = fail("copy failed", __FILE__, __LINE__);
      if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess)
        rc = fail("assembly kernel failed", __FILE__, __LINE__);
    }
    if (rc == ISPH_SUCCESS && !rhs_only) {
      rc = sell_sum_rowlen(ctx, M, &M.nnz);
    }
  }
  S.release(); E.release(); snu.release(); sp.release(); sf.release(); sv.release();
  bdev.release();
  newlen.release();
  if (rc != ISPH_SUCCESS || rhs_only) { A->S.release(); delete A; return rc; }
  *A_out = A;
  return ISPH_SUCCESS;
}

// FunctorOuterGradientCorrection + FunctorOuterLaplacianCorrection for the owned particles
inline int compute_corrections(isph_ctx *ctx, const isph_particles *P, double *Gc_out, double *Lc_out, int on_device) {
  ISPH_REQUIRE(P->dim == 2 || P->dim == 3, "dim must be 2 or 3");
  ISPH_REQUIRE(P->x && P->type && (P->neigh_ptr || P->neigh_ptr64) && P->neigh_idx && P->vfrac, "particle arrays missing (vfrac incl. ghosts)");
  const int n = P->nlocal, dim = P->dim, d2 = dim * dim, dL = dim * (dim + 1) / 2;
  StagedParticles S;
  AsmTables T;
  int rc = stage_tables(ctx, P, S, T);
  const double *dx = nullptr, *dvf = nullptr;
  const int *dt = nullptr, *dp = nullptr, *di = nullptr;
  long long nnb = 0;
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->x, (size_t)P->nall * 3, on_device, S.x, &dx);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->type, (size_t)P->nall, on_device, S.type, &dt);
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->vfrac, (size_t)P->nall, on_device, S.vfrac, &dvf);
  NeighPtr np;
  if (rc == ISPH_SUCCESS) rc = stage_neigh_ptr(ctx, P, n, on_device, S.nptr, S.nptr64, np, &nnb);
  dp = np.p32;
  if (rc == ISPH_SUCCESS) rc = stage(ctx, P->neigh_idx, (size_t)nnb, on_device, S.nidx, &di);
  NeighEll E;
  if (rc == ISPH_SUCCESS) rc = build_neigh_ell(ctx, n, np, di, E, T);
  DevTmp<double> g, l;
  DevTmp<int> nf;
  double *dG = Gc_out, *dLc = Lc_out;
  if (rc == ISPH_SUCCESS && !on_device) {
    rc = g.reserve((size_t)(n > 0 ? n : 1) * d2);
    if (rc == ISPH_SUCCESS) rc = l.reserve((size_t)(n > 0 ? n : 1) * dL);
    dG = g.p; dLc = l.p;
  }
  if (rc == ISPH_SUCCESS) rc = nf.reserve(1);
  int nfail = 0;
  if (rc == ISPH_SUCCESS && n > 0) {
    const int grid = (n + kBlock - 1) / kBlock;
    if (hipMemsetAsync(nf.p, 0, sizeof(int), ctx->stream) != hipSuccess) rc = fail("memset failed", __FILE__, __LINE__);
    hipLaunchKernelGGL(k_gradient_correction, dim3(xcd_grid(grid)), dim3(kBlock), 0, ctx->stream, T, n, dx, dt, dp, di, dvf, dG);
    if (P->dim == 3)
      hipLaunchKernelGGL((k_laplacian_correction<3>), dim3(xcd_grid(grid)), dim3(kBlock), 0, ctx->stream, T, n, dx, dt, dp, di, dvf,
                       (const double *)dG, dLc, nf.p);
    else
      hipLaunchKernelGGL((k_laplacian_correction<2>), dim3(xcd_grid(grid)), dim3(kBlock), 0, ctx->stream, T, n, dx, dt, dp, di, dvf,
                       (const double *)dG, dLc, nf.p);
    if (!on_device) {
      if (hipMemcpyAsync(Gc_out, dG, sizeof(double) * (size_t)n * d2, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipMemcpyAsync(Lc_out, dLc, sizeof(double) * (size_t)n * dL, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        rc = fail("copy failed", __FILE__, __LINE__);
    }
    if (hipMemcpyAsync(&nfail, nf.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess)
      rc = fail("correction kernels failed", __FILE__, __LINE__);
  }
  S.release(); E.release(); g.release(); l.release(); nf.release();
  if (rc == ISPH_SUCCESS && nfail > 0) return fail("singular Laplacian-correction system (DGESV failed in the reference)", __FILE__, __LINE__);
  return rc;
}

}  // namespace isph
