// order.hpp -- the row numbering the library owns.
//
// The reference numbers matrix rows by LAMMPS' atom order (the Epetra map is built from atom->tag[0..nlocal),
// ref: pair_isph.cpp:1258-1259) and Ifpack's subdomains are the bricks of LAMMPS' spatial decomposition (one per MPI rank,
// ref: precond_ifpack.h:60-74).  Both are the CALLER's: an atom order that is lexicographic, sorted by LAMMPS' sort bins
// or arbitrary after migration, and as many bricks as there are ranks.  The block-Jacobi ILU of this library wants
// subdomains of <= 1024 rows that are compact in space, and the sliced-ELL SpMV wants the 64 rows of a slice to share
// their x window -- so the assembly sorts the owned particles into bricks itself:
//
//   bounding box of the owned particles -> mean spacing d -> about extent / d fine cells per axis whose faces are the
//   QUANTILES of the coordinates (one histogram per axis) -> bricks of kTarget cells
//   (10 x 10 x 5 in 3-D, what profiles/r04_ilu_order.txt found best; 22 x 22 in 2-D)
//   key(i) = brick(i) * cells_per_brick + cell inside the brick (x fastest), sorted stably: ties keep the caller's order
//
// The permutation lives in the matrix (isph_mat::order).  Everything that crosses the C ABI stays in the caller's
// numbering: b / x / null mask of isph_solve, x / y of isph_spmv, r / z of isph_prec_apply, the send list of
// isph_mat_set_halo and the CSR export are gathered / scattered at the boundary (8 MB each way at 10^6 rows).  The brick
// table (over-full bricks split, empty ones dropped) is the subdomain table of "bjacobi-ilu<k>" with block_size 0.
// isph_mat_ordering exports permutation, table and geometry, so that a checker can restate all of it.
#pragma once
#include <rocprim/rocprim.hpp>

#include <cmath>
#include <memory>

#include "common.hpp"

namespace isph {

constexpr int kOrderBlockCap = 1024;  // rows per subdomain the block stream takes (ilu.hpp)

struct OrderGeom {
  int dim;
  double lo[3], inv_bin[3];  // the histogram grid the cell faces are picked from: bin = floor((x - lo) * inv_bin)
  int nbins[3];
  int ncell[3];   // fine cells per axis
  int cpb[3];     // fine cells per brick and axis
  int nbrick[3];  // bricks per axis
  // faces between the fine cells of axis a, ascending (device): cell(x) = number of faces <= x.  They are QUANTILES of
  // the coordinates (face k has k/ncell of the particles below it, rounded up to a bin edge of the histogram), not
  // multiples of a cell width: the planes of a lattice stay one cell each when a few particles have wrapped around a
  // periodic box and stretched the bounding box by a spacing (uniform cells then drift by one plane across the box and
  // the bricks hold 9 x 9 x 4 ... 11 x 11 x 6 planes instead of 10 x 10 x 5), and a non-uniform cloud gets bricks of
  // equal population along every axis.
  const double *face[3];
  // periodic axes (isph_ctx_set_periodic_box): every coordinate is taken as t = x - shift, + period when negative, before
  // anything else looks at it.  shift is a point of the widest empty stretch of the axis (circular histogram of the box):
  // particles that have just wrapped around the box end -- half a lattice plane at x = L - eps, the other half at
  // x = +eps -- are one plane again, and the bounding box does not grow by a spacing.  period 0: not periodic / not told.
  double shift[3], period[3];
};

struct OrderBox {  // what the caller said about its box (isph_ctx_set_periodic_box); periodic[a] = 0: nothing known
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  int periodic[3] = {0, 0, 0};
};

__device__ __host__ inline double order_coord(const OrderGeom &g, int a, double x) {
  double t = x - g.shift[a];
  if (g.period[a] > 0.0 && t < 0.0) t += g.period[a];
  return t;
}

struct RowOrder {
  int n = 0;
  OrderGeom g;
  DevBuf<int> perm;    // internal row r holds the caller's row perm[r]
  DevBuf<int> iperm;   // the caller's row i is internal row iperm[i]
  std::vector<int> block_ptr;  // subdomains: consecutive internal rows, each 1 .. kOrderBlockCap
  DevBuf<double> dface;        // the cell faces of the three axes, one after the other
  std::vector<double> hface[3];
  RowOrder() { memset(&g, 0, sizeof(g)); }
  RowOrder(const RowOrder &) = delete;
  RowOrder &operator=(const RowOrder &) = delete;
  ~RowOrder() { perm.release(); iperm.release(); dface.release(); }
  int nblocks() const { return (int)block_ptr.size() - 1; }
};
using RowOrderPtr = std::shared_ptr<RowOrder>;

// ---- gather / scatter at the boundary ----------------------------------------------------------------------------
// out[r][c] = in[perm[r]][c] for r < nperm, out[r][c] = in[r][c] for nperm <= r < ntotal (ghost particles keep their place)
template <class T>
__global__ void k_perm_gather(long long ntotal, int nperm, int ncomp, const int *__restrict__ perm, const T *__restrict__ in,
                              T *__restrict__ out) {
  const long long total = ntotal * ncomp;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long r = e / ncomp;
    const int c = (int)(e - r * ncomp);
    const long long s = r < nperm ? (long long)perm[r] : r;
    out[e] = in[s * ncomp + c];
  }
}
// out[perm[r]] = in[r], r < n
template <class T>
__global__ void k_perm_scatter(int n, const int *__restrict__ perm, const T *__restrict__ in, T *__restrict__ out) {
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) out[perm[r]] = in[r];
}
// matrix column of a particle in the internal numbering: owned columns move, ghost columns (>= nlocal) stay
__global__ void k_perm_colmap(int nall, int nlocal, const int *__restrict__ perm, const int *__restrict__ iperm,
                              const int *__restrict__ colmap, int *__restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nall) return;
  const int c = colmap[r < nlocal ? perm[r] : r];
  out[r] = c < nlocal ? iperm[c] : c;
}
// the same indexed by the CALLER's particle j: the sort key of a neighbour-list entry without going through idmap first
__global__ void k_perm_colkey(int nall, int nlocal, const int *__restrict__ iperm, const int *__restrict__ colmap, int *__restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nall) return;
  const int c = colmap[j];
  out[j] = c < nlocal ? iperm[c] : c;
}
// the particle index map of the neighbour lists: owned j -> iperm[j], ghosts keep their index
__global__ void k_perm_idmap(int nall, int nlocal, const int *__restrict__ iperm, int *__restrict__ idmap) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < nall) idmap[j] = j < nlocal ? iperm[j] : j;
}
__global__ void k_perm_map_indices(int n, int nlocal, const int *__restrict__ iperm, const int *__restrict__ in, int *__restrict__ out) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    const int i = in[k];
    out[k] = i < nlocal ? iperm[i] : i;
  }
}

inline int perm_grid(long long n) {
  const long long g = (n + kBlock - 1) / kBlock;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// ---- the brick sort ----------------------------------------------------------------------------------------------
// three coordinate arrays (PrecondWrapper_ML::setCoordinates hands them over like this, precond_ml.h:63-94) -> [n][3]
__global__ void k_order_soa_to_aos(int n, const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                   double *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[3 * (size_t)i] = x[i];
  out[3 * (size_t)i + 1] = y[i];
  out[3 * (size_t)i + 2] = z ? z[i] : 0.0;
}

// per-workgroup bounding box of x[0..n)[0..3): part[block][0..3) = min, [3..6) = max
__global__ __launch_bounds__(kBlock) void k_order_bbox(int n, OrderGeom g, const double *__restrict__ x, double *__restrict__ part) {
  __shared__ double red[kBlock / 64][6];
  double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    for (int a = 0; a < 3; ++a) {
      const double v = order_coord(g, a, x[3 * (size_t)i + a]);
      mn[a] = fmin(mn[a], v);
      mx[a] = fmax(mx[a], v);
    }
  for (int a = 0; a < 3; ++a)
    for (int o = 32; o > 0; o >>= 1) {
      mn[a] = fmin(mn[a], __shfl_xor(mn[a], o, 64));
      mx[a] = fmax(mx[a], __shfl_xor(mx[a], o, 64));
    }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
    for (int a = 0; a < 3; ++a) { red[wave][a] = mn[a]; red[wave][3 + a] = mx[a]; }
  __syncthreads();
  if (threadIdx.x < 6) {
    double v = red[0][threadIdx.x];
    for (int w = 1; w < kBlock / 64; ++w) v = threadIdx.x < 3 ? fmin(v, red[w][threadIdx.x]) : fmax(v, red[w][threadIdx.x]);
    part[(size_t)blockIdx.x * 6 + threadIdx.x] = v;
  }
}

// histogram bin of a coordinate; (x - lo) * inv_bin is a subtraction followed by a multiplication -- nothing a compiler
// can contract -- so a host restatement reaches the same bin bit for bit
__device__ __host__ inline int order_bin(const OrderGeom &g, int a, double x) {
  const double t = (x - g.lo[a]) * g.inv_bin[a];
  int q = (int)floor(t);
  return q < 0 ? 0 : (q >= g.nbins[a] ? g.nbins[a] - 1 : q);
}
__global__ void k_order_hist(int n, OrderGeom g, const double *__restrict__ x, int *__restrict__ h0, int *__restrict__ h1,
                             int *__restrict__ h2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  atomicAdd(&h0[order_bin(g, 0, order_coord(g, 0, x[3 * (size_t)i]))], 1);
  atomicAdd(&h1[order_bin(g, 1, order_coord(g, 1, x[3 * (size_t)i + 1]))], 1);
  if (g.dim == 3) atomicAdd(&h2[order_bin(g, 2, order_coord(g, 2, x[3 * (size_t)i + 2]))], 1);
}

// The same histogram through a workgroup-private copy in LDS: particles in lattice order hand a wave 64 times the same
// y and z bin, and a million global atomics on a few thousand words ran 0.46 ms; here a workgroup of 1024 threads counts
// its share of the particles in LDS and adds its non-zero bins to the global histogram once.  h = the three axes'
// histograms one behind the other (o1, o2: where axes 1 and 2 begin), nb_all ints of dynamic LDS.
constexpr int kHistGroups = 128;
__global__ __launch_bounds__(1024) void k_order_hist_lds(int n, OrderGeom g, const double *__restrict__ x, int *__restrict__ h,
                                                         int nb_all, int o1, int o2) {
  extern __shared__ int order_lh[];
  for (int s = threadIdx.x; s < nb_all; s += 1024) order_lh[s] = 0;
  __syncthreads();
  for (int i = blockIdx.x * 1024 + threadIdx.x; i < n; i += gridDim.x * 1024) {
    atomicAdd(&order_lh[order_bin(g, 0, order_coord(g, 0, x[3 * (size_t)i]))], 1);
    atomicAdd(&order_lh[o1 + order_bin(g, 1, order_coord(g, 1, x[3 * (size_t)i + 1]))], 1);
    if (g.dim == 3) atomicAdd(&order_lh[o2 + order_bin(g, 2, order_coord(g, 2, x[3 * (size_t)i + 2]))], 1);
  }
  __syncthreads();
  for (int s = threadIdx.x; s < nb_all; s += 1024) {
    const int v = order_lh[s];
    if (v) atomicAdd(&h[s], v);
  }
}

// occupancy of the caller's periodic box along its periodic axes, kCutBins bins per axis: bin = clamp(floor((x - lo) * inv))
constexpr int kCutBins = 4096;
// (workgroup-private LDS copy, like k_order_hist_lds)
__global__ __launch_bounds__(1024) void k_order_cut_hist(int n, int dim, OrderBox bx, const double *__restrict__ x, int *__restrict__ h) {
  __shared__ int lh[3 * kCutBins];
  for (int s = threadIdx.x; s < 3 * kCutBins; s += 1024) lh[s] = 0;
  __syncthreads();
  for (int i = blockIdx.x * 1024 + threadIdx.x; i < n; i += gridDim.x * 1024)
    for (int a = 0; a < dim; ++a) {
      if (!bx.periodic[a]) continue;
      const double inv = (double)kCutBins / (bx.hi[a] - bx.lo[a]);
      int q = (int)floor((x[3 * (size_t)i + a] - bx.lo[a]) * inv);
      q = q < 0 ? 0 : (q >= kCutBins ? kCutBins - 1 : q);
      atomicAdd(&lh[a * kCutBins + q], 1);
    }
  __syncthreads();
  for (int s = threadIdx.x; s < 3 * kCutBins; s += 1024) {
    const int v = lh[s];
    if (v) atomicAdd(&h[s], v);
  }
}

// key = brick * cells_per_brick + cell inside the brick, cell(a) = number of faces of axis a that are <= x_a
__device__ inline unsigned long long order_key(const OrderGeom &g, const double *xi) {
  int b[3] = {0, 0, 0}, c[3] = {0, 0, 0};
  for (int a = 0; a < g.dim; ++a) {
    const double *f = g.face[a];
    int lo = 0, hi = g.ncell[a] - 1;   // faces f[0 .. ncell-2]; q = first index with f[q] > x
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (f[mid] <= xi[a]) lo = mid + 1; else hi = mid;
    }
    const int q = lo;
    b[a] = q / g.cpb[a];
    c[a] = q - b[a] * g.cpb[a];
  }
  const unsigned long long brick = ((unsigned long long)b[2] * g.nbrick[1] + b[1]) * g.nbrick[0] + b[0];
  const unsigned long long cell = ((unsigned long long)c[2] * g.cpb[1] + c[1]) * g.cpb[0] + c[0];
  return brick * ((unsigned long long)g.cpb[0] * g.cpb[1] * g.cpb[2]) + cell;
}

__global__ void k_order_keys(int n, OrderGeom g, const double *__restrict__ x, unsigned long long *__restrict__ key,
                             int *__restrict__ val) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double xi[3] = {order_coord(g, 0, x[3 * (size_t)i]), order_coord(g, 1, x[3 * (size_t)i + 1]),
                        order_coord(g, 2, x[3 * (size_t)i + 2])};
  key[i] = order_key(g, xi);
  val[i] = i;
}

// inverse permutation + first internal row of every brick that holds particles (brick_start prefilled with -1)
__global__ void k_order_finish(int n, unsigned long long cells_per_brick, const unsigned long long *__restrict__ skey,
                               const int *__restrict__ perm, int *__restrict__ iperm, int *__restrict__ brick_start) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  iperm[perm[r]] = r;
  const unsigned long long b = skey[r] / cells_per_brick;
  if (r == 0 || skey[r - 1] / cells_per_brick != b) brick_start[b] = r;
}

// Brick geometry from the bounding box of the owned particles.  d solves d^dim = prod(extent_a + d) / n (the particles
// of a lattice sit at the cell centres of a box half a spacing wider than their bounding box on every side); every axis
// then gets round((extent + d) / d) fine cells and ceil(cells / target - 0.2) bricks of equally many cells.
inline void order_geometry(int dim, int n, const double mn[3], const double mx[3], OrderGeom &g) {
  static const int target3[3] = {10, 10, 5}, target2[3] = {22, 22, 1};
  const int *target = dim == 3 ? target3 : target2;
  const double keep_shift[3] = {g.shift[0], g.shift[1], g.shift[2]}, keep_period[3] = {g.period[0], g.period[1], g.period[2]};
  memset(&g, 0, sizeof(g));
  for (int a = 0; a < 3; ++a) { g.shift[a] = keep_shift[a]; g.period[a] = keep_period[a]; }
  g.dim = dim;
  double ext[3] = {0, 0, 0};
  double scale = 0.0;
  for (int a = 0; a < dim; ++a) { ext[a] = mx[a] - mn[a]; if (!(ext[a] >= 0.0)) ext[a] = 0.0; scale = std::max(scale, ext[a]); }
  // all particles in one point / on one line: any positive spacing serves
  if (!(scale > 0.0)) scale = 1.0;
  double d = scale / std::max(1.0, std::pow((double)n, 1.0 / dim));
  for (int it = 0; it < 32; ++it) {
    double vol = 1.0;
    for (int a = 0; a < dim; ++a) vol *= ext[a] + d;
    const double dn = std::pow(vol / (double)(n > 0 ? n : 1), 1.0 / dim);
    if (std::fabs(dn - d) <= 1e-14 * d) { d = dn; break; }
    d = dn;
  }
  for (int a = 0; a < 3; ++a) {
    g.lo[a] = 0.0; g.inv_bin[a] = 0.0; g.nbins[a] = 1; g.ncell[a] = 1; g.cpb[a] = 1; g.nbrick[a] = 1; g.face[a] = nullptr;
  }
  for (int a = 0; a < dim; ++a) {
    const double len = ext[a] + d;
    long long cells = std::llround(len / d);
    if (cells < 1) cells = 1;
    if (cells > (1LL << 20)) cells = 1LL << 20;
    long long nb = (long long)std::ceil((double)cells / target[a] - 0.2);
    if (nb < 1) nb = 1;
    const long long cpb = (cells + nb - 1) / nb;
    g.ncell[a] = (int)cells;
    g.cpb[a] = (int)cpb;
    g.nbrick[a] = (int)((cells + cpb - 1) / cpb);
    g.lo[a] = mn[a] - 0.5 * d;
    g.nbins[a] = (int)std::min<long long>(64 * cells, 1LL << 20);   // 64 bins per cell: a face lands between two lattice planes
    g.inv_bin[a] = (double)g.nbins[a] / len;
  }
}

// Faces of the fine cells of one axis from its histogram: face k (k = 1 .. ncell - 1) is the upper edge of the first bin
// at which the cumulative count reaches ceil(k n / ncell).  Bin edges are lo + j / inv_bin with integer j: a restatement
// that builds the same histogram gets the same doubles.
inline void order_faces(const OrderGeom &g, int a, int n, const int *hist, std::vector<double> &face) {
  const int nc = g.ncell[a];
  face.assign((size_t)(nc > 1 ? nc - 1 : 0), 0.0);
  long long cum = 0;
  int j = -1;
  for (int k = 1; k < nc; ++k) {
    const long long target = ((long long)k * n + nc - 1) / nc;
    while (cum < target && j + 1 < g.nbins[a]) cum += hist[++j];
    face[(size_t)k - 1] = g.lo[a] + (double)(j + 1) / g.inv_bin[a];
  }
}

// Subdomain table from the first rows of the occupied bricks: a brick above the capacity is cut into equal consecutive
// pieces (its rows are ordered z-slowest, so the pieces are slabs), bricks without particles have no entry.
inline void order_block_table(int n, const std::vector<int> &brick_start, std::vector<int> &block_ptr) {
  block_ptr.clear();
  block_ptr.push_back(0);
  int prev = -1;
  auto close = [&](int lo, int hi) {
    const int cnt = hi - lo;
    if (cnt <= 0) return;
    const int pieces = (cnt + kOrderBlockCap - 1) / kOrderBlockCap;
    const int each = (cnt + pieces - 1) / pieces;
    for (int s = lo; s < hi; s += each) block_ptr.push_back(std::min(hi, s + each));
  };
  for (size_t b = 0; b < brick_start.size(); ++b) {
    const int s = brick_start[b];
    if (s < 0) continue;
    if (prev >= 0) close(prev, s);
    prev = s;
  }
  if (prev >= 0) close(prev, n);
}

// x: device [>= n][3] positions of the owned particles in the caller's order.  Three host round trips (bounding box,
// histograms of the three axes, brick starts); everything else is queued on `st`.
inline int order_build(hipStream_t st, int dim, int n, const double *x, RowOrderPtr &out, const OrderBox *box = nullptr) {
  ISPH_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
  RowOrderPtr O = std::make_shared<RowOrder>();
  O->n = n;
  O->block_ptr.assign(1, 0);
  if (n == 0) { order_geometry(dim, 0, (const double[3]){0, 0, 0}, (const double[3]){0, 0, 0}, O->g); out = O; return ISPH_SUCCESS; }
  ISPH_CHECK(O->perm.reserve((size_t)n));
  ISPH_CHECK(O->iperm.reserve((size_t)n));
  // 0. periodic axes the caller told about: the cut goes into the widest empty stretch of the axis
  bool any_periodic = false;
  for (int a = 0; box && a < dim; ++a) any_periodic = any_periodic || (box->periodic[a] && box->hi[a] > box->lo[a]);
  if (any_periodic) {
    DevTmp<int> ch;
    ISPH_CHECK(ch.reserve((size_t)3 * kCutBins));
    ISPH_CHECK_HIP(hipMemsetAsync(ch.p, 0, sizeof(int) * 3 * kCutBins, st));
    OrderBox bx = *box;
    for (int a = 0; a < 3; ++a) if (a >= dim || !(bx.hi[a] > bx.lo[a])) bx.periodic[a] = 0;
    hipLaunchKernelGGL(k_order_cut_hist, dim3(std::min(kHistGroups, (n + 1023) / 1024)), dim3(1024), 0, st, n, dim, bx, x, ch.p);
    std::vector<int> hc((size_t)3 * kCutBins);
    ISPH_CHECK_HIP(hipMemcpyAsync(hc.data(), ch.p, sizeof(int) * hc.size(), hipMemcpyDeviceToHost, st));
    ISPH_CHECK_HIP(hipStreamSynchronize(st));
    for (int a = 0; a < dim; ++a) {
      if (!bx.periodic[a]) continue;
      // the first longest circular run of empty bins; no empty bin: the box end stays where it is
      const int *h = hc.data() + (size_t)a * kCutBins;
      int best_len = 0, best_start = 0;
      for (int s0 = 0; s0 < kCutBins; ++s0) {
        if (h[s0] != 0 || h[(s0 + kCutBins - 1) % kCutBins] == 0) continue;   // runs start behind an occupied bin
        int len = 0;
        while (len < kCutBins && h[(s0 + len) % kCutBins] == 0) ++len;
        if (len > best_len) { best_len = len; best_start = s0; }
      }
      const double L = bx.hi[a] - bx.lo[a];
      O->g.period[a] = L;
      O->g.shift[a] = bx.lo[a];
      if (best_len > 0 && best_len < kCutBins) {
        const int mid = (best_start + best_len / 2) % kCutBins;   // a bin edge in the middle of the run
        O->g.shift[a] = bx.lo[a] + (double)mid * (L / (double)kCutBins);
      }
    }
  }
  // 1. bounding box
  const int gb = std::min(256, (n + kBlock - 1) / kBlock);
  DevTmp<double> part;
  ISPH_CHECK(part.reserve((size_t)gb * 6));
  hipLaunchKernelGGL(k_order_bbox, dim3(gb), dim3(kBlock), 0, st, n, O->g, x, part.p);
  std::vector<double> hp((size_t)gb * 6);
  ISPH_CHECK_HIP(hipMemcpyAsync(hp.data(), part.p, sizeof(double) * hp.size(), hipMemcpyDeviceToHost, st));
  ISPH_CHECK_HIP(hipStreamSynchronize(st));
  double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
  for (int b = 0; b < gb; ++b)
    for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], hp[(size_t)b * 6 + a]); mx[a] = std::max(mx[a], hp[(size_t)b * 6 + 3 + a]); }
  for (int a = 0; a < 3; ++a) ISPH_REQUIRE(std::isfinite(mn[a]) && std::isfinite(mx[a]), "particle positions are not finite");
  order_geometry(dim, n, mn, mx, O->g);
  // 1b. cell faces = quantiles of the coordinates, from one histogram per axis
  {
    OrderGeom &gw = O->g;
    const size_t nb_all = (size_t)gw.nbins[0] + gw.nbins[1] + gw.nbins[2];
    DevTmp<int> hist;
    ISPH_CHECK(hist.reserve(nb_all));
    ISPH_CHECK_HIP(hipMemsetAsync(hist.p, 0, sizeof(int) * nb_all, st));
    if (nb_all * sizeof(int) <= (size_t)150 * 1024) {
      const size_t lds = nb_all * sizeof(int);
      ISPH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_order_hist_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k_order_hist_lds, dim3(std::min(kHistGroups, (n + 1023) / 1024)), dim3(1024), lds, st, n, gw, x, hist.p, (int)nb_all,
                         gw.nbins[0], gw.nbins[0] + gw.nbins[1]);
    } else {
      hipLaunchKernelGGL(k_order_hist, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, gw, x, hist.p, hist.p + gw.nbins[0],
                         hist.p + gw.nbins[0] + gw.nbins[1]);
    }
    std::vector<int> hh(nb_all);
    ISPH_CHECK_HIP(hipMemcpyAsync(hh.data(), hist.p, sizeof(int) * nb_all, hipMemcpyDeviceToHost, st));
    ISPH_CHECK_HIP(hipStreamSynchronize(st));
    size_t nf = 0, hoff = 0;
    for (int a = 0; a < 3; ++a) {
      if (a < dim) order_faces(gw, a, n, hh.data() + hoff, O->hface[a]);
      hoff += (size_t)gw.nbins[a];
      nf += O->hface[a].size();
    }
    ISPH_CHECK(O->dface.reserve(nf > 0 ? nf : 1));
    size_t at = 0;
    for (int a = 0; a < 3; ++a) {
      gw.face[a] = O->dface.p + at;
      if (!O->hface[a].empty())
        ISPH_CHECK_HIP(hipMemcpyAsync(O->dface.p + at, O->hface[a].data(), sizeof(double) * O->hface[a].size(), hipMemcpyHostToDevice, st));
      at += O->hface[a].size();
    }
  }
  const OrderGeom &g = O->g;
  const unsigned long long cpbt = (unsigned long long)g.cpb[0] * g.cpb[1] * g.cpb[2];
  const unsigned long long nbricks = (unsigned long long)g.nbrick[0] * g.nbrick[1] * g.nbrick[2];
  ISPH_REQUIRE(nbricks < (1ull << 31), "too many bricks");
  // 2. keys, stable radix sort over the bits in use
  DevTmp<unsigned long long> k0, k1;
  DevTmp<int> v0, bstart;
  DevTmp<char> tmp;
  ISPH_CHECK(k0.reserve((size_t)n));
  ISPH_CHECK(k1.reserve((size_t)n));
  ISPH_CHECK(v0.reserve((size_t)n));
  ISPH_CHECK(bstart.reserve((size_t)nbricks));
  hipLaunchKernelGGL(k_order_keys, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, g, x, k0.p, v0.p);
  unsigned bits = 1;
  while (bits < 64 && ((nbricks * cpbt) >> bits) != 0) ++bits;
  size_t bytes = 0;
  ISPH_CHECK_HIP(rocprim::radix_sort_pairs(nullptr, bytes, k0.p, k1.p, v0.p, O->perm.p, (size_t)n, 0u, bits, st));
  ISPH_CHECK(tmp.reserve(bytes > 0 ? bytes : 1));
  ISPH_CHECK_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, k0.p, k1.p, v0.p, O->perm.p, (size_t)n, 0u, bits, st));
  // 3. inverse permutation, brick starts -> subdomain table
  ISPH_CHECK_HIP(hipMemsetAsync(bstart.p, 0xff, sizeof(int) * (size_t)nbricks, st));
  hipLaunchKernelGGL(k_order_finish, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, cpbt, (const unsigned long long *)k1.p,
                     (const int *)O->perm.p, O->iperm.p, bstart.p);
  std::vector<int> hs((size_t)nbricks);
  ISPH_CHECK_HIP(hipMemcpyAsync(hs.data(), bstart.p, sizeof(int) * hs.size(), hipMemcpyDeviceToHost, st));
  ISPH_CHECK_HIP(hipStreamSynchronize(st));
  ISPH_CHECK_HIP(hipGetLastError());
  order_block_table(n, hs, O->block_ptr);
  ISPH_REQUIRE(O->block_ptr.back() == n, "brick table does not cover the rows");
  out = O;
  return ISPH_SUCCESS;
}

}  // namespace isph
