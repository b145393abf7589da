// isph_capi.hip -- the extern "C" surface of libisph_hip.so (include/isph_hip.h).
// gfx950 only; no CPU fallback anywhere in this library.
#include <algorithm>
#include <numeric>

#include "assemble.hpp"
#include "operators.hpp"
#include "block_helmholtz.hpp"
#include "core.hpp"
#include "ilu.hpp"
#include "krylov.hpp"
#include "sell.hpp"
#include "solver.hpp"
#include "amg.hpp"
#include "schwarz.hpp"
#include "gcrodr.hpp"
#include "ingress.hpp"

namespace isph {
thread_local std::string g_last_error;

}  // namespace isph (reopened below)

// Ifpack_AdditiveSchwarz<ILU(k)> with "Overlap Level" 1 on more than one rank (precond_ifpack.h:43,60-74): the rank's
// subdomain is its own rows plus the rows of its ghost columns (Ifpack_OverlappingRowMatrix); the extended matrix is
// built by the caller (Epetra_Import of the rows; dist.extend_rows in the Python plumbing) and factored as one block by
// the level-scheduled path of schwarz.hpp.  An application gathers the ghost part of r with the matrix' halo plan,
// solves on the extended vector and -- combine mode Add, the wrapper's default -- sends the ghost part of the result
// back to its owners, who add it; Zero keeps the owned part only (restricted additive Schwarz).
struct isph_overlap {
  int n = 0, next = 0, combine = 0;
  isph_schwarz *inner = nullptr;
  isph_halo H;
  isph::DevBuf<double> rext, zext, sbuf, rbuf;
};

namespace isph {

__global__ void k_scatter_add(int n, const int *__restrict__ idx, const double *__restrict__ v, double *__restrict__ z) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) z[idx[i]] += v[i];
}

inline void overlap_destroy(isph_overlap *O) {
  if (!O) return;
  if (O->inner) schwarz_destroy(O->inner);
  O->H.send_idx.release();
  O->rext.release(); O->zext.release(); O->sbuf.release(); O->rbuf.release();
  delete O;
}

inline int overlap_exchange(isph_ctx *ctx, const isph_halo &H, const double *send, double *recv, bool reverse) {
  // forward: send[send range p] -> peer p, recv[recv range p] <- peer p; reverse: the two roles swapped
  return comm_exchange(ctx, H, send, recv, 1, reverse, ctx->stream);
}

inline int overlap_apply(isph_ctx *ctx, const isph_overlap *O, const double *r, double *z) {
  const isph_halo &H = O->H;
  const int n = O->n;
  ISPH_CHECK_HIP(hipMemcpyAsync(O->rext.p, r, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
  if (H.nsend > 0)
    hipLaunchKernelGGL(k_gather, dim3(stream_grid(H.nsend)), dim3(kBlock), 0, ctx->stream, H.nsend, (const int *)H.send_idx.p, r,
                       O->sbuf.p);
  ISPH_CHECK(overlap_exchange(ctx, H, O->sbuf.p, O->rext.p + n, false));
  ISPH_CHECK(schwarz_apply(ctx, O->inner, O->rext.p, O->zext.p));
  ISPH_CHECK_HIP(hipMemcpyAsync(z, O->zext.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
  if (O->combine == 0 && H.npeers > 0) {  // Add: the neighbours' corrections of my rows come home (a rank may only send or only receive)
    ISPH_CHECK(overlap_exchange(ctx, H, O->zext.p + n, O->rbuf.p, true));
    for (int p = 0; p < H.npeers; ++p) {  // one launch per peer: a row can be in several peers' lists, never twice in one
      const int s0 = H.send_ptr[(size_t)p], cnt = H.send_ptr[(size_t)p + 1] - s0;
      if (cnt > 0)
        hipLaunchKernelGGL(k_scatter_add, dim3(stream_grid(cnt)), dim3(kBlock), 0, ctx->stream, cnt, (const int *)H.send_idx.p + s0,
                           (const double *)O->rbuf.p + s0, z);
    }
  }
  ISPH_CHECK_HIP(hipGetLastError());
  return ISPH_SUCCESS;
}

// The persistent triangular sweeps of the Schwarz path (schwarz.hpp k_gilu_solve_run) give up after a spin limit, store
// zeros and raise a word that the application copies to S->h_tmo.  Called with the stream DRAINED at the end of every
// solve / host-side application: a raised word fails the call -- on every rank together (one small max-reduction per
// solve; a rank that returned alone would leave the others waiting in their next collective) -- and is cleared, so one
// spurious time-out (a shared device, a debugger) does not poison the object.  A silently zeroed M^-1 r is never
// reported as a converged solve.
int prec_health(isph_ctx *ctx, const isph_prec *M) {
  if (!M || (M->type != 4 && M->type != 5)) return ISPH_SUCCESS;
  const isph_schwarz *S = M->type == 4 ? M->schwarz : (M->ovl ? M->ovl->inner : nullptr);
  double bad = 0.0;
  if (S && S->syncfree && S->h_tmo && *S->h_tmo != 0) {
    bad = 1.0;
    *S->h_tmo = 0;
    ISPH_CHECK_HIP(hipMemsetAsync(S->ctr.p + 3, 0, sizeof(int), ctx->stream));
  }
  if (comm_active(ctx) && ctx->nranks > 1) {
    ISPH_CHECK(ensure_scalars(ctx));
    double *d = ctx->dscal.p + SC_MISC + 30;
    ISPH_CHECK_HIP(hipMemcpyAsync(d, &bad, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    ISPH_CHECK(comm_allreduce(ctx, d, 1, /*max*/ 1, ctx->stream));
    ISPH_CHECK_HIP(hipMemcpyAsync(&bad, d, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  }
  if (bad != 0.0)
    return fail("Schwarz ILU sweep: a row waited for a dependency beyond the spin limit; the result of this call is not valid", __FILE__, __LINE__);
  return ISPH_SUCCESS;
}

int prec_apply_dev(isph_ctx *ctx, const isph_prec *M, const double *r, double *z) {
  ISPH_REQUIRE(M != nullptr, "preconditioner is NULL");
  const int n = M->n;
  ProfScope prof(ctx, PROF_PREC_APPLY);
  if (M->type == 0) {
    // identity: callers pass distinct buffers
    if (r != z) ISPH_CHECK_HIP(hipMemcpyAsync(z, r, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
    return ISPH_SUCCESS;
  }
  if (M->type == 1) {
    hipLaunchKernelGGL(k_mul_elem, dim3(stream_grid(n)), dim3(kBlock), 0, ctx->stream, n, r, M->invdiag.p, z);
    return ISPH_SUCCESS;
  }
  if (M->type == 3) return amg_apply(ctx, M->amg, r, z);
  if (M->type == 4) return schwarz_apply(ctx, M->schwarz, r, z);
  if (M->type == 5) return overlap_apply(ctx, M->ovl, r, z);
  return ilu_apply(ctx, M->ilu, r, z);
}

// K applications of one preconditioner: block ILU(k) sweeps its factor stream once for all vectors
int prec_apply_multi_dev(isph_ctx *ctx, const isph_prec *M, int K, const double *const *rs, double *const *zs) {
  ISPH_REQUIRE(M != nullptr, "preconditioner is NULL");
  if (M->type == 2 && M->ilu) return ilu_apply_multi(ctx, M->ilu, K, rs, zs);
  for (int k = 0; k < K; ++k) ISPH_CHECK(prec_apply_dev(ctx, M, rs[k], zs[k]));
  return ISPH_SUCCESS;
}

// upload helper: returns device pointer (either the caller's or a staged copy)
template <class T>
int stage_in(isph_ctx *ctx, const T *src, size_t n, int on_device, DevBuf<T> &tmp, const T **out) {
  if (on_device) { *out = src; return ISPH_SUCCESS; }
  ISPH_REQUIRE(!is_device_pointer(src), "device pointer passed with on_device = 0");
  ISPH_CHECK(tmp.reserve(n > 0 ? n : 1));
  ISPH_CHECK_HIP(hipMemcpyAsync(tmp.p, src, sizeof(T) * n, hipMemcpyHostToDevice, ctx->stream));
  *out = tmp.p;
  return ISPH_SUCCESS;
}

int sell_finalize_offsets(isph_ctx *ctx, Sell &S) {
  // slice_off currently holds per-slice entry counts in [0,nslices)
  hipLaunchKernelGGL(k_exclusive_scan_ll, dim3(1), dim3(1024), 0, ctx->stream, S.nslices, S.slice_off.p, S.slice_off.p);
  // the whole offset array comes back (8 B per slice): its last entry sizes the arrays, its differences give the
  // widest slice, which callers with sorted rows would otherwise fetch in a round trip of their own (sell_set_wmax)
  std::vector<long long> so((size_t)S.nslices + 1, 0);
  ISPH_CHECK_HIP(hipMemcpyAsync(so.data(), S.slice_off.p, sizeof(long long) * so.size(), hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  const long long total = so[(size_t)S.nslices];
  S.wmax = 0;
  for (int s2 = 0; s2 < S.nslices; ++s2) S.wmax = std::max(S.wmax, (int)((so[(size_t)s2 + 1] - so[(size_t)s2]) >> 6));
  S.stored = total;
  ISPH_CHECK(S.col.reserve((size_t)(total > 0 ? total : 1)));
  ISPH_CHECK(S.val.reserve((size_t)(total > 0 ? total : 1)));
  return ISPH_SUCCESS;
}

// widest slice (the row sort computes it on the way; callers with already sorted rows only need this)
int sell_set_wmax(isph_ctx *ctx, Sell &S) {
  S.wmax = 0;
  if (S.nslices == 0) return ISPH_SUCCESS;
  std::vector<long long> so((size_t)S.nslices + 1);
  ISPH_CHECK_HIP(hipMemcpyAsync(so.data(), S.slice_off.p, sizeof(long long) * so.size(), hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  for (int s = 0; s < S.nslices; ++s) S.wmax = std::max(S.wmax, (int)((so[(size_t)s + 1] - so[(size_t)s]) >> 6));
  return ISPH_SUCCESS;
}

// sort every row's entries by column (invariant of isph_mat, needs max width from host)
int sell_sort_rows(isph_ctx *ctx, Sell &S) {
  if (S.nslices == 0) return ISPH_SUCCESS;
  std::vector<long long> so((size_t)S.nslices + 1);
  ISPH_CHECK_HIP(hipMemcpyAsync(so.data(), S.slice_off.p, sizeof(long long) * so.size(), hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  int wmax = 0;
  for (int s = 0; s < S.nslices; ++s) wmax = std::max(wmax, (int)((so[(size_t)s + 1] - so[(size_t)s]) >> 6));
  S.wmax = wmax;
  if (wmax == 0) return ISPH_SUCCESS;
  const int Ws = wmax | 1;
  int R = 64;
  // <= 48 KiB per workgroup keeps >= 3 workgroups (12 waves) per CU in flight: the kernel is a
  // global-load/store stream, the LDS rank sort in between is cheap
  while (R > 1 && (size_t)R * Ws * 24 > 48 * 1024) R >>= 1;
  ISPH_REQUIRE((size_t)R * Ws * 24 <= 160 * 1024, "row too long for the LDS row sort");
  const size_t lds = (size_t)R * Ws * 24;
  ISPH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sell_sort_rows),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_sell_sort_rows, dim3(S.nslices), dim3(kBlock), lds, ctx->stream, S.nrow, S.nslices, R, Ws,
                     S.rowlen.p, S.slice_off.p, S.col.p, S.val.p);
  ISPH_CHECK_HIP(hipGetLastError());
  return ISPH_SUCCESS;
}

// ---- the library's own row numbering around one assembly call (order.hpp) ---------------------------------------
// Builds (or, while the neighbour list is held, reuses) the brick order of the owned particles, hands the assembly a
// device view of the particle arrays in that numbering -- owned particles permuted, ghosts in place, matrix columns and
// the wall / correction tensors following their particles -- and tells the neighbour-layout builder (ctx->nmap) to read
// the caller's list through the permutation.  The assembled right-hand side goes back in the caller's numbering.
struct OrderedAssembly {
  isph_ctx *ctx;
  const isph_particles *P;
  int on_device;
  RowOrderPtr O;
  isph_particles Q;
  std::vector<DevBuf<char> *> bufs;
  DevBuf<int> idmap, colmap, colkey;
  OrderedAssembly(isph_ctx *c, const isph_particles *p, int dev) : ctx(c), P(p), on_device(dev) { Q = *p; }
  OrderedAssembly(const OrderedAssembly &) = delete;
  ~OrderedAssembly() {
    ctx->nmap = isph_neigh_map();
    for (DevBuf<char> *b : bufs) { b->release(); delete b; }
    idmap.release(); colmap.release(); colkey.release();
  }
  template <class T>
  int alloc(size_t count, T **out) {
    DevBuf<char> *b = new DevBuf<char>();
    bufs.push_back(b);
    ISPH_CHECK(b->reserve((count > 0 ? count : 1) * sizeof(T)));
    *out = reinterpret_cast<T *>(b->p);
    return ISPH_SUCCESS;
  }
  // device copy of a caller array as it is (host arrays are staged; device arrays pass through)
  template <class T>
  int on_dev(const T *src, size_t count, const T **out) {
    if (!src) { *out = nullptr; return ISPH_SUCCESS; }
    if (on_device) { *out = src; return ISPH_SUCCESS; }
    ISPH_REQUIRE(!is_device_pointer(src), "device pointer passed with on_device = 0");
    T *d = nullptr;
    ISPH_CHECK(alloc(count, &d));
    if (count > 0) ISPH_CHECK_HIP(hipMemcpyAsync(d, src, sizeof(T) * count, hipMemcpyHostToDevice, ctx->stream));
    *out = d;
    return ISPH_SUCCESS;
  }
  // a per-particle field [rows][ncomp] in the internal numbering: rows = nlocal (owned only) or nall (ghosts in place)
  template <class T>
  int field(const T *src, int ncomp, bool with_ghosts, const T **out) {
    if (!src) { *out = nullptr; return ISPH_SUCCESS; }
    const long long rows = with_ghosts ? P->nall : P->nlocal;
    const T *d = nullptr;
    ISPH_CHECK(on_dev(src, (size_t)rows * ncomp, &d));
    T *q = nullptr;
    ISPH_CHECK(alloc((size_t)rows * ncomp, &q));
    if (rows > 0)
      hipLaunchKernelGGL((k_perm_gather<T>), dim3(perm_grid(rows * ncomp)), dim3(kBlock), 0, ctx->stream, rows, P->nlocal, ncomp,
                         (const int *)O->perm.p, d, q);
    *out = q;
    return ISPH_SUCCESS;
  }
  int begin(int ncol) {
    const int n = P->nlocal, nall = P->nall, dim = P->dim, dL = dim * (dim + 1) / 2;
    ISPH_REQUIRE(P->x && P->type && (P->neigh_ptr || P->neigh_ptr64) && P->neigh_idx && P->colmap, "particle arrays missing");
    ISPH_REQUIRE(n >= 0 && nall >= n, "need 0 <= nlocal <= nall");
    // neighbour list: stays the caller's (read through the permutation), on the device
    long long nnb = 0;
    if (P->neigh_ptr64) {
      ISPH_CHECK(on_dev(P->neigh_ptr64, (size_t)n + 1, &Q.neigh_ptr64));
      Q.neigh_ptr = nullptr;
      if (!on_device) nnb = P->neigh_ptr64[n];
    } else {
      ISPH_CHECK(on_dev(P->neigh_ptr, (size_t)n + 1, &Q.neigh_ptr));
      if (!on_device) nnb = P->neigh_ptr[n];
    }
    if (!on_device) {  // host-side shape checks before any kernel indexes with these (the assembly only sees device arrays)
      ISPH_REQUIRE(nnb >= 0, "negative neighbour count");
      for (long long k = 0; k < nnb; ++k) ISPH_REQUIRE(P->neigh_idx[k] >= 0 && P->neigh_idx[k] < nall, "neighbour index out of range");
      for (int j = 0; j < nall; ++j) ISPH_REQUIRE(P->colmap[j] >= 0 && P->colmap[j] < ncol, "colmap entry out of range");
      ISPH_CHECK(on_dev(P->neigh_idx, (size_t)nnb, &Q.neigh_idx));
    }
    const double *dx = nullptr;
    ISPH_CHECK(on_dev(P->x, (size_t)nall * 3, &dx));
    // the order: reused while the caller holds the neighbour list
    const void *k0 = (const void *)P->neigh_idx, *k1 = P->neigh_ptr64 ? (const void *)P->neigh_ptr64 : (const void *)P->neigh_ptr;
    if (ctx->neigh_hold && ctx->held_order && ctx->held_order->n == n && ctx->held_key[0] == k0 && ctx->held_key[1] == k1) {
      O = ctx->held_order;
    } else {
      ISPH_CHECK(order_build(ctx->stream, dim, n, dx, O, &ctx->box));
      if (ctx->neigh_hold) { ctx->held_order = O; ctx->held_key[0] = k0; ctx->held_key[1] = k1; }
    }
    const int *perm = O->perm.p, *iperm = O->iperm.p;
    // particle arrays in the internal numbering
    Q.x = nullptr;
    {
      double *q = nullptr;
      ISPH_CHECK(alloc((size_t)nall * 3, &q));
      if (nall > 0)
        hipLaunchKernelGGL((k_perm_gather<double>), dim3(perm_grid((long long)nall * 3)), dim3(kBlock), 0, ctx->stream, (long long)nall, n, 3, perm, dx, q);
      Q.x = q;
    }
    ISPH_CHECK(field(P->type, 1, true, &Q.type));
    ISPH_CHECK(field(P->vfrac, 1, true, &Q.vfrac));
    ISPH_CHECK(field(P->pnd, 1, true, &Q.pnd));
    ISPH_CHECK(field(P->normal, 3, true, &Q.normal));
    ISPH_CHECK(field(P->Gc, dim * dim, false, &Q.Gc));
    ISPH_CHECK(field(P->Lc, dL, false, &Q.Lc));
    const int *dcm = nullptr;
    ISPH_CHECK(on_dev(P->colmap, (size_t)nall, &dcm));
    ISPH_CHECK(colmap.reserve((size_t)(nall > 0 ? nall : 1)));
    ISPH_CHECK(idmap.reserve((size_t)(nall > 0 ? nall : 1)));
    ISPH_CHECK(colkey.reserve((size_t)(nall > 0 ? nall : 1)));
    if (nall > 0) {
      hipLaunchKernelGGL(k_perm_colkey, dim3((nall + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, nall, n, iperm, dcm, colkey.p);
      hipLaunchKernelGGL(k_perm_colmap, dim3((nall + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, nall, n, perm, iperm, dcm, colmap.p);
      hipLaunchKernelGGL(k_perm_idmap, dim3((nall + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, nall, n, iperm, idmap.p);
    }
    Q.colmap = colmap.p;
    ctx->nmap.rowsrc = perm;
    ctx->nmap.idmap = idmap.p;
    ctx->nmap.colmap_key = P->colmap;
    ctx->nmap.colkey = colkey.p;
    ctx->nmap.order = O.get();
    ISPH_CHECK_HIP(hipGetLastError());
    return ISPH_SUCCESS;
  }
  // the assembled right-hand side, internal [n x ncols] contiguous -> the caller's [lda x ncols], host or device
  int rhs_out(const double *internal, double *out, int ncols, int lda) {
    const int n = P->nlocal;
    if (n == 0) return ISPH_SUCCESS;
    double *d = out;
    if (!on_device) ISPH_CHECK(alloc((size_t)lda * ncols, &d));
    for (int c = 0; c < ncols; ++c)
      hipLaunchKernelGGL((k_perm_scatter<double>), dim3(perm_grid(n)), dim3(kBlock), 0, ctx->stream, n, (const int *)O->perm.p,
                         internal + (size_t)c * n, d + (size_t)c * lda);
    if (!on_device) {
      // rows lda > n of a column are not the library's to write: copy column by column
      for (int c = 0; c < ncols; ++c)
        ISPH_CHECK_HIP(hipMemcpyAsync(out + (size_t)c * lda, d + (size_t)c * lda, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    }
    ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    ISPH_CHECK_HIP(hipGetLastError());
    return ISPH_SUCCESS;
  }
};

// S0 (the caller's numbering) -> S (the numbering of O): rows perm[.], owned columns through iperm, rows column-sorted
int sell_permute(isph_ctx *ctx, const Sell &S0, const RowOrder &O, Sell &S) {
  S.nrow = S0.nrow; S.ncol = S0.ncol; S.nnz = S0.nnz; S.nslices = S0.nslices; S.wmax = S0.wmax;
  const int n = S.nrow;
  ISPH_CHECK(S.rowlen.reserve((size_t)(n > 0 ? n : 1)));
  ISPH_CHECK(S.slice_off.reserve((size_t)S.nslices + 1));
  if (n == 0) return ISPH_SUCCESS;
  const int grid = (n + kBlock - 1) / kBlock;
  hipLaunchKernelGGL((k_perm_gather<int>), dim3(perm_grid(n)), dim3(kBlock), 0, ctx->stream, (long long)n, n, 1, (const int *)O.perm.p,
                     (const int *)S0.rowlen.p, S.rowlen.p);
  hipLaunchKernelGGL(k_slicew_from_rowlen, dim3(grid), dim3(kBlock), 0, ctx->stream, n, (const int *)S.rowlen.p, S.slice_off.p);
  ISPH_CHECK(sell_finalize_offsets(ctx, S));
  const int wmax = S.wmax;
  if (wmax == 0) return ISPH_SUCCESS;
  const int Ws = wmax | 1;
  int R = 64;
  while (R > 1 && (size_t)R * Ws * 24 > 48 * 1024) R >>= 1;
  ISPH_REQUIRE((size_t)R * Ws * 24 <= 150 * 1024, "row too long for the LDS row sort");
  const size_t lds = (size_t)R * Ws * 24;
  ISPH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sell_permute_sort), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_sell_permute_sort, dim3(S.nslices), dim3(kBlock), lds, ctx->stream, n, S.nslices, R, Ws, (const int *)O.perm.p,
                     (const int *)O.iperm.p, (const int *)S0.rowlen.p, (const long long *)S0.slice_off.p, (const int *)S0.col.p,
                     (const double *)S0.val.p, (const int *)S.rowlen.p, (const long long *)S.slice_off.p, S.col.p, S.val.p);
  ISPH_CHECK_HIP(hipGetLastError());
  return ISPH_SUCCESS;
}

}  // namespace isph

using namespace isph;

extern "C" {

const char *isph_last_error(void) { return g_last_error.c_str(); }

int isph_comm_unique_id(char *uid) {
  static_assert(sizeof(ncclUniqueId) <= ISPH_UID_BYTES, "uid size");
  ncclUniqueId id;
  ISPH_CHECK_NCCL(ncclGetUniqueId(&id));
  memset(uid, 0, ISPH_UID_BYTES);
  memcpy(uid, &id, sizeof(id));
  return ISPH_SUCCESS;
}

static int ctx_create_common(int device, void *stream, isph_ctx **out) {
  ISPH_REQUIRE(out != nullptr, "ctx out pointer is NULL");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) return fail("no HIP device: libisph_hip has no CPU fallback", __FILE__, __LINE__);
  ISPH_REQUIRE(device >= 0 && device < ndev, "device index out of range");
  ISPH_CHECK_HIP(hipSetDevice(device));
  isph_ctx *c = new isph_ctx();
  c->device = device;
  if (stream) {
    c->stream = (hipStream_t)stream;
  } else {
    // hipStreamDefault: implicitly ordered against the legacy null stream, so a caller that issues its own
    // work (zero-fills, temporaries) on stream 0 stays ordered with the library's kernels
    ISPH_CHECK_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamDefault));
    c->own_stream = true;
  }
  ISPH_CHECK_HIP(hipEventCreate(&c->ev0));
  ISPH_CHECK_HIP(hipEventCreate(&c->ev1));
  ISPH_CHECK_HIP(hipEventCreateWithFlags(&c->ev_fetch, hipEventDisableTiming));
  ISPH_CHECK_HIP(hipEventCreateWithFlags(&c->ev_pack, hipEventDisableTiming));
  ISPH_CHECK_HIP(hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming));
  // the halo stream only exists on contexts with a communicator: HIP multiplexes its streams onto a handful of hardware
  // queues, and an idle extra stream costs the host ingress (ingress.hpp) a queue of its own
  ISPH_CHECK(ensure_scalars(c));
  *out = c;
  return ISPH_SUCCESS;
}

int isph_ctx_create(int device, void *stream, isph_ctx **ctx) {
  ISPH_CHECK(ctx_create_common(device, stream, ctx));
  // ISPH_POOL_CANARY=1 ISPH_POOL_CANARY_SELFTEST=1: one byte is written behind a 100-byte buffer on purpose; giving the
  // buffer back must abort the process (tests/test_gpu_parity.py shows the check bites)
  if (pool_canary() && getenv("ISPH_POOL_CANARY_SELFTEST")) {
    DevBuf<char> t;
    ISPH_CHECK(t.reserve(100));
    ISPH_CHECK_HIP(hipMemset(t.p + 104, 0, 1));
    t.release();
  }
  return ISPH_SUCCESS;
}

int isph_ctx_create_dist(int device, void *stream, int rank, int nranks, const char *uid, isph_ctx **ctx) {
  ISPH_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks && uid, "bad rank/nranks/uid");
  ISPH_CHECK(ctx_create_common(device, stream, ctx));
  isph_ctx *c = *ctx;
  c->rank = rank;
  c->nranks = nranks;
  ncclUniqueId id;
  memcpy(&id, uid, sizeof(id));
  ISPH_CHECK_HIP(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  ISPH_CHECK_NCCL(ncclCommInitRank(&c->comm, nranks, id, rank));
  return ISPH_SUCCESS;
}

int isph_ctx_create_hostcomm(int device, void *stream, int rank, int nranks, const isph_host_transport *t, isph_ctx **ctx) {
  ISPH_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank/nranks");
  ISPH_REQUIRE(t && t->exchange && t->allreduce, "host transport without its two callbacks");
  ISPH_CHECK(ctx_create_common(device, stream, ctx));
  isph_ctx *c = *ctx;
  c->rank = rank;
  c->nranks = nranks;
  c->host_tr = *t;
  // the same second stream as the RCCL contexts: the staged halo exchange is ordered exactly like the grouped send/recv
  ISPH_CHECK_HIP(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  return ISPH_SUCCESS;
}

int isph_device_identity(int device, char id[ISPH_DEVICE_ID_BYTES]) {
  ISPH_REQUIRE(id, "NULL argument");
  memset(id, 0, ISPH_DEVICE_ID_BYTES);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail("device index out of range", __FILE__, __LINE__);
  ISPH_CHECK_HIP(hipDeviceGetPCIBusId(id, ISPH_DEVICE_ID_BYTES - 1, device));
  return ISPH_SUCCESS;
}

int isph_ctx_sync(isph_ctx *ctx) {
  ISPH_REQUIRE(ctx, "ctx is NULL");
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return ISPH_SUCCESS;
}

int isph_ctx_set_profile(isph_ctx *ctx, int on) {
  ISPH_REQUIRE(ctx, "ctx is NULL");
  ctx->profile = on != 0;
  ctx->ev_used = 0;  // switching the mode starts a new collection
  ctx->hev_used = 0;
  return ISPH_SUCCESS;
}

int isph_ctx_halo_profile_read(isph_ctx *ctx, double ms[3], int *calls) {
  ISPH_REQUIRE(ctx && ms && calls, "NULL argument");
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->comm_stream) ISPH_CHECK_HIP(hipStreamSynchronize(ctx->comm_stream));
  ms[0] = ms[1] = ms[2] = 0.0;
  *calls = 0;
  for (size_t k = 0; k + 3 <= ctx->hev_used; k += 3) {
    float te = 0.f, ti = 0.f;
    ISPH_CHECK_HIP(hipEventElapsedTime(&te, ctx->hev[k], ctx->hev[k + 1]));
    ISPH_CHECK_HIP(hipEventElapsedTime(&ti, ctx->hev[k], ctx->hev[k + 2]));
    ms[0] += te; ms[1] += ti; ms[2] += te > ti ? te - ti : 0.0;
    *calls += 1;
  }
  ctx->hev_used = 0;
  return ISPH_SUCCESS;
}

int isph_ctx_comm_info(const isph_ctx *ctx, long long info[4]) {
  ISPH_REQUIRE(ctx && info, "NULL argument");
  info[0] = ctx->comm ? 1 : (ctx->host_tr.exchange ? 2 : 0);
  info[1] = ctx->nranks; info[2] = ctx->rank; info[3] = ctx->device;
  if (ctx->comm) {
    int cnt = 0, ur = 0, dv = 0;
    ISPH_CHECK_NCCL(ncclCommCount(ctx->comm, &cnt));
    ISPH_CHECK_NCCL(ncclCommUserRank(ctx->comm, &ur));
    ISPH_CHECK_NCCL(ncclCommCuDevice(ctx->comm, &dv));
    info[1] = cnt; info[2] = ur; info[3] = dv;
  }
  return ISPH_SUCCESS;
}

int isph_ctx_hold_neighbours(isph_ctx *ctx, int on) {
  ISPH_REQUIRE(ctx, "ctx is NULL");
  for (auto &c : ctx->neigh_cache) c.release();   // either way: what was kept belongs to the list of before
  ctx->held_order.reset();
  ctx->held_key[0] = ctx->held_key[1] = nullptr;
  ctx->neigh_hold = on != 0;
  return ISPH_SUCCESS;
}

int isph_ctx_set_ordering(isph_ctx *ctx, int mode) {
  ISPH_REQUIRE(ctx && (mode == ISPH_ORDER_CALLER || mode == ISPH_ORDER_BRICKS), "bad ordering mode");
  ctx->ordering = mode;
  ctx->held_order.reset();
  return ISPH_SUCCESS;
}

int isph_ctx_set_periodic_box(isph_ctx *ctx, const double lo[3], const double hi[3], const int periodic[3]) {
  ISPH_REQUIRE(ctx, "ctx is NULL");
  ctx->box = OrderBox();
  if (lo && hi && periodic)
    for (int a = 0; a < 3; ++a) { ctx->box.lo[a] = lo[a]; ctx->box.hi[a] = hi[a]; ctx->box.periodic[a] = periodic[a] ? 1 : 0; }
  ctx->held_order.reset();
  return ISPH_SUCCESS;
}

int isph_mat_ordering_info(const isph_mat *A, long long info[3], isph_order_geometry *geom) {
  ISPH_REQUIRE(A && info, "NULL argument");
  info[0] = A->order ? 1 : 0;
  info[1] = A->S.nrow;
  info[2] = A->order ? A->order->nblocks() : 0;
  if (geom) {
    memset(geom, 0, sizeof(*geom));
    if (A->order) {
      const OrderGeom &g = A->order->g;
      geom->dim = g.dim;
      for (int a = 0; a < 3; ++a) {
        geom->lo[a] = g.lo[a]; geom->inv_bin[a] = g.inv_bin[a]; geom->nbins[a] = g.nbins[a]; geom->ncell[a] = g.ncell[a];
        geom->cells_per_brick[a] = g.cpb[a]; geom->nbrick[a] = g.nbrick[a];
        geom->shift[a] = g.shift[a]; geom->period[a] = g.period[a];
      }
    }
  }
  return ISPH_SUCCESS;
}

int isph_mat_ordering_faces(const isph_mat *A, int axis, double *faces) {
  ISPH_REQUIRE(A && faces && axis >= 0 && axis < 3, "bad argument");
  ISPH_REQUIRE(A->order, "the matrix is in the caller's row numbering");
  const std::vector<double> &f = A->order->hface[axis];
  if (!f.empty()) memcpy(faces, f.data(), sizeof(double) * f.size());
  return ISPH_SUCCESS;
}

int isph_mat_ordering(isph_ctx *ctx, const isph_mat *A, int *perm, int *block_ptr) {
  ISPH_REQUIRE(ctx && A, "NULL argument");
  ISPH_REQUIRE(A->order, "the matrix is in the caller's row numbering");
  const RowOrder &O = *A->order;
  if (perm && O.n > 0) {
    ISPH_CHECK_HIP(hipMemcpyAsync(perm, O.perm.p, sizeof(int) * (size_t)O.n, hipMemcpyDeviceToHost, ctx->stream));
    ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  }
  if (block_ptr) memcpy(block_ptr, O.block_ptr.data(), sizeof(int) * O.block_ptr.size());
  return ISPH_SUCCESS;
}

int isph_ctx_profile_read(isph_ctx *ctx, double ms[8], int calls[8]) {
  ISPH_REQUIRE(ctx && ms && calls, "NULL argument");
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < 8; ++k) { ms[k] = 0.0; calls[k] = 0; }
  for (size_t k = 0; k + 1 < ctx->ev_used; k += 2) {
    float t = 0.f;
    ISPH_CHECK_HIP(hipEventElapsedTime(&t, ctx->ev[k], ctx->ev[k + 1]));
    const int c = ctx->ev_class[k / 2];
    if (c >= 0 && c < 8) { ms[c] += t; calls[c] += 1; }
  }
  ctx->ev_used = 0;
  return ISPH_SUCCESS;
}

int isph_pool_trim(void) {
  if (hipDeviceSynchronize() != hipSuccess) return fail("device synchronisation failed", __FILE__, __LINE__);
  DevPool::get().trim();
  HostPool::get().trim();  // the host work arrays of the Schwarz set-up (schwarz.hpp)
  return ISPH_SUCCESS;
}

int isph_set_exact_stream_threshold(long long bytes) {
  ilu_exact_stream_bytes() = bytes >= 0 ? bytes : (4LL << 30);
  return ISPH_SUCCESS;
}

int isph_pool_set_cap(long long bytes) {
  DevPool &p = DevPool::get();
  {
    std::lock_guard<std::mutex> lk(p.mu);
    p.cap_bytes = bytes > 0 ? (size_t)bytes : 0;  // 0: back to the default (80 % of the free memory at the next release)
  }
  if (bytes > 0 && (long long)p.cached > bytes) return isph_pool_trim();
  return ISPH_SUCCESS;
}

int isph_pool_info(long long info[4], int reset_peak) {
  ISPH_REQUIRE(info, "NULL argument");
  DevPool &p = DevPool::get();
  std::lock_guard<std::mutex> lk(p.mu);
  info[0] = (long long)p.cached; info[1] = (long long)p.live; info[2] = (long long)p.peak_live; info[3] = (long long)p.cap_bytes;
  if (reset_peak) p.peak_live = p.live;
  return ISPH_SUCCESS;
}

long long isph_pool_cached_bytes(void) {
  DevPool &p = DevPool::get();
  std::lock_guard<std::mutex> lk(p.mu);
  return (long long)p.cached;
}

void isph_ctx_destroy(isph_ctx *c) {
  if (!c) return;
  (void)hipStreamSynchronize(c->stream);
  if (c->comm_stream) { (void)hipStreamSynchronize(c->comm_stream); (void)hipStreamDestroy(c->comm_stream); }
  if (c->ev_pack) (void)hipEventDestroy(c->ev_pack);
  if (c->ev_halo) (void)hipEventDestroy(c->ev_halo);
  c->xghost.release();
  for (auto &nc : c->neigh_cache) nc.release();
  c->tables.release();
  if (c->comm) (void)ncclCommDestroy(c->comm);
  if (c->hsend) (void)hipHostFree(c->hsend);
  if (c->hrecv) (void)hipHostFree(c->hrecv);
  for (auto e : c->ev) (void)hipEventDestroy(e);
  for (auto e : c->hev) (void)hipEventDestroy(e);
  (void)hipEventDestroy(c->ev0);
  (void)hipEventDestroy(c->ev1);
  if (c->ev_fetch) (void)hipEventDestroy(c->ev_fetch);
  for (hipEvent_t e : c->ev_ls) (void)hipEventDestroy(e);
  c->partial.release(); c->dscal.release(); c->V.release(); c->Z.release(); c->wv.release(); c->tv.release();
  c->rv.release(); c->pv.release(); c->nvec.release(); c->xext.release(); c->sendbuf.release();
  c->bdev.release(); c->xdev.release(); c->imask.release();
  c->bint.release(); c->xint.release(); c->imask2.release();
  c->held_order.reset();
  if (c->hscal) (void)hipHostFree(c->hscal);
  delete c->stager;
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  DevPool::get().trim();  // cached device blocks go back to the driver with the context
}

/* ---- matrix ----------------------------------------------------------- */

}  // extern "C"

// CSR (device pointers, any shape) -> sliced-ELL matrix; rows end up column-sorted
template <class OFF>
static int mat_from_device_csr_t(isph_ctx *ctx, int nrow, int ncol, const OFF *drp, const int *dci, const double *dv,
                                 long long nnz, isph_mat **Aout, bool rows_sorted) {
  isph_mat *A = new isph_mat();
  Sell &S = A->S;
  S.nrow = nrow; S.ncol = ncol; S.nnz = nnz;
  S.nslices = (nrow + kSlice - 1) / kSlice;
  int rc = S.slice_off.reserve((size_t)S.nslices + 1);
  if (rc == ISPH_SUCCESS) rc = S.rowlen.reserve((size_t)(nrow > 0 ? nrow : 1));
  if (rc == ISPH_SUCCESS && nrow > 0) {
    hipLaunchKernelGGL(k_csr_rowlen_slicew<OFF>, dim3((nrow + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, nrow, drp,
                       S.rowlen.p, S.slice_off.p);
    rc = sell_finalize_offsets(ctx, S);
    if (rc == ISPH_SUCCESS) {
      hipLaunchKernelGGL(k_csr_to_sell<OFF>, dim3((S.nslices + 3) / 4), dim3(kBlock), 0, ctx->stream, nrow, drp, dci, dv,
                         (const long long *)S.slice_off.p, S.col.p, S.val.p, 0, S.nslices, (int *)nullptr, CsrChunks{});
      rc = rows_sorted ? sell_set_wmax(ctx, S) : sell_sort_rows(ctx, S);
      if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess)
        rc = fail("CSR->SELL conversion failed", __FILE__, __LINE__);
    }
  }
  if (rc != ISPH_SUCCESS) { isph_mat_destroy(A); return rc; }
  *Aout = A;
  return ISPH_SUCCESS;
}
int mat_from_device_csr(isph_ctx *ctx, int nrow, int ncol, const int *drp, const int *dci, const double *dv, long long nnz,
                        isph_mat **Aout, bool rows_sorted) {
  return mat_from_device_csr_t<int>(ctx, nrow, ncol, drp, dci, dv, nnz, Aout, rows_sorted);
}
int mat_from_device_csr(isph_ctx *ctx, int nrow, int ncol, const long long *drp, const int *dci, const double *dv,
                        long long nnz, isph_mat **Aout, bool rows_sorted) {
  return mat_from_device_csr_t<long long>(ctx, nrow, ncol, drp, dci, dv, nnz, Aout, rows_sorted);
}

extern "C" {

int isph_mat_create_csr(isph_ctx *ctx, int nrow, int ncol, const int *rowptr, const int *colidx,
                        const double *val, int on_device, isph_mat **Aout) {
  ISPH_REQUIRE(ctx && Aout && rowptr && colidx && val, "NULL argument");
  ISPH_REQUIRE(nrow >= 0 && ncol >= nrow, "need 0 <= nrow <= ncol");
  if (!on_device) return csr_ingress_host(ctx, nrow, ncol, rowptr, colidx, val, Aout);  // pipelined, ingress.hpp
  int last = 0;
  ISPH_CHECK_HIP(hipMemcpyAsync(&last, rowptr + nrow, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  isph_mat *A = nullptr;
  ISPH_CHECK(mat_from_device_csr(ctx, nrow, ncol, rowptr, colidx, val, (long long)last, &A));
  *Aout = A;
  return ISPH_SUCCESS;
}

static int order_from_host_coords(isph_ctx *ctx, int nrow, int dim, const double *x, const double *y, const double *z, RowOrderPtr &O);

int isph_mat_create_csr_coords(isph_ctx *ctx, int nrow, int ncol, const int *rowptr, const int *colidx, const double *val,
                               int dim, const double *x, const double *y, const double *z, isph_mat **Aout) {
  ISPH_REQUIRE(ctx && Aout && rowptr && colidx && val && x && y, "NULL argument");
  ISPH_REQUIRE(nrow >= 0 && ncol >= nrow, "need 0 <= nrow <= ncol");
  ISPH_REQUIRE((dim == 2) || (dim == 3 && z), "dim must be 2, or 3 with z");
  ISPH_REQUIRE(!is_device_pointer(x) && !is_device_pointer(y) && !is_device_pointer(z), "coordinates must be host arrays");
  if (nrow == 0) return isph_mat_create_csr(ctx, nrow, ncol, rowptr, colidx, val, 0, Aout);
  // 1. the row order from the coordinates (0.5 ms at 10^6 rows; the matrix is not needed for it)
  RowOrderPtr O;
  ISPH_CHECK(order_from_host_coords(ctx, nrow, dim, x, y, z, O));
  // 2. the matrix over the link as it is (the pipelined ingress: conversion hidden behind the copies)
  isph_mat *A0 = nullptr;
  ISPH_CHECK(csr_ingress_host(ctx, nrow, ncol, rowptr, colidx, val, &A0));
  // 3. one pass on the device: rows gathered in the new order, columns renamed, rows sorted
  isph_mat *A = new isph_mat();
  int rc = sell_permute(ctx, A0->S, *O, A->S);
  if (rc == ISPH_SUCCESS && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail("permutation of the matrix failed", __FILE__, __LINE__);
  isph_mat_destroy(A0);
  if (rc != ISPH_SUCCESS) { isph_mat_destroy(A); return rc; }
  A->order = O;
  *Aout = A;
  return ISPH_SUCCESS;
}

// the row order of a host CSR from the coordinates of its rows (three host arrays, PrecondWrapper_ML::setCoordinates' layout)
static int order_from_host_coords(isph_ctx *ctx, int nrow, int dim, const double *x, const double *y, const double *z, RowOrderPtr &O) {
  DevTmp<double> soa, aos;
  ISPH_CHECK(soa.reserve((size_t)3 * nrow));
  ISPH_CHECK(aos.reserve((size_t)3 * nrow));
  ISPH_CHECK_HIP(hipMemcpyAsync(soa.p, x, sizeof(double) * (size_t)nrow, hipMemcpyHostToDevice, ctx->stream));
  ISPH_CHECK_HIP(hipMemcpyAsync(soa.p + nrow, y, sizeof(double) * (size_t)nrow, hipMemcpyHostToDevice, ctx->stream));
  if (dim == 3) ISPH_CHECK_HIP(hipMemcpyAsync(soa.p + 2 * (size_t)nrow, z, sizeof(double) * (size_t)nrow, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_order_soa_to_aos, dim3((nrow + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, nrow, (const double *)soa.p,
                     (const double *)soa.p + nrow, dim == 3 ? (const double *)soa.p + 2 * (size_t)nrow : (const double *)nullptr, aos.p);
  return order_build(ctx->stream, dim, nrow, aos.p, O, &ctx->box);
}

int isph_mat_create_csr_coords_bjacobi(isph_ctx *ctx, int nrow, int ncol, const int *rowptr, const int *colidx, const double *val,
                                       int dim, const double *x, const double *y, const double *z, isph_mat **Aout, isph_prec **Mout) {
  ISPH_REQUIRE(ctx && Aout && Mout && rowptr && colidx && val && x && y, "NULL argument");
  ISPH_REQUIRE(nrow > 0 && ncol >= nrow, "need 0 < nrow <= ncol");
  ISPH_REQUIRE((dim == 2) || (dim == 3 && z), "dim must be 2, or 3 with z");
  ISPH_REQUIRE(!is_device_pointer(x) && !is_device_pointer(y) && !is_device_pointer(z), "coordinates must be host arrays");
  ISPH_REQUIRE(rowptr[0] == 0, "rowptr must start at 0");
  RowOrderPtr O;
  ISPH_CHECK(order_from_host_coords(ctx, nrow, dim, x, y, z, O));
  // the permuted row pointers on the host (the staging threads walk the rows in the new order): four threads gather the
  // row lengths, the prefix sum is one pass
  std::vector<int> hperm((size_t)nrow), rp2((size_t)nrow + 1);
  ISPH_CHECK_HIP(hipMemcpyAsync(hperm.data(), O->perm.p, sizeof(int) * (size_t)nrow, hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  {
    const int T = 4;
    std::vector<std::thread> th;
    std::vector<int> bad((size_t)T, 0);
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t]() {
        const size_t a = (size_t)nrow * t / T, b = (size_t)nrow * (t + 1) / T;
        for (size_t r = a; r < b; ++r) {
          const int sr = hperm[r];
          const int len = rowptr[sr + 1] - rowptr[sr];
          if (len < 0) bad[(size_t)t] = 1;
          rp2[r + 1] = len;
        }
      });
    for (auto &q : th) q.join();
    for (int t = 0; t < T; ++t) ISPH_REQUIRE(!bad[(size_t)t], "rowptr not monotone");
    rp2[0] = 0;
    long long run = 0;
    for (int r = 0; r < nrow; ++r) { run += rp2[(size_t)r + 1]; ISPH_REQUIRE(run <= 2147483647LL, "more than 2^31 entries"); rp2[(size_t)r + 1] = (int)run; }
    ISPH_REQUIRE(run == (long long)rowptr[nrow], "row pointers and entry count disagree");
  }
  const std::vector<int> &bp = O->block_ptr;
  int cap = 64;
  for (size_t b = 0; b + 1 < bp.size(); ++b) cap = std::max(cap, bp[b + 1] - bp[b]);
  cap = (cap + 63) / 64 * 64;
  IngressGather gat;
  gat.perm = hperm.data(); gat.src_rowptr = rowptr; gat.colren = O->iperm.p; gat.nren = nrow;
  isph_mat *A = nullptr;
  isph_ilu *F = nullptr;
  ISPH_CHECK(csr_ingress_host_bjacobi(ctx, nrow, ncol, rp2.data(), colidx, val, cap, &A, &F, (int)bp.size() - 1, bp.data(), &gat));
  A->order = O;
  isph_prec *M = new isph_prec();
  M->n = nrow;
  M->type = 2;
  M->ilu = F;
  M->order = O;
  *Aout = A;
  *Mout = M;
  return ISPH_SUCCESS;
}

int isph_mat_create_csr_blocks(isph_ctx *ctx, int nrow, int ncol, const int *rowptr, const int *colidx, const double *val,
                               int nblocks, const int *block_ptr, isph_mat **Aout, isph_prec **Mout) {
  ISPH_REQUIRE(ctx && Aout && Mout && rowptr && colidx && val && nblocks > 0 && block_ptr, "NULL argument or no subdomains");
  ISPH_REQUIRE(nrow >= 0 && ncol >= nrow, "need 0 <= nrow <= ncol");
  ISPH_REQUIRE(!is_device_pointer(block_ptr), "block_ptr must be a host array");
  ISPH_REQUIRE(block_ptr[0] == 0 && block_ptr[nblocks] == nrow, "subdomain table must run from 0 to the number of rows");
  int cap = 64;
  for (int b = 0; b < nblocks; ++b) cap = std::max(cap, block_ptr[b + 1] - block_ptr[b]);
  cap = (cap + 63) / 64 * 64;
  ISPH_REQUIRE(cap <= 1024, "a subdomain of the block stream holds at most 1024 rows");
  isph_mat *A = nullptr;
  isph_ilu *F = nullptr;
  ISPH_CHECK(csr_ingress_host_bjacobi(ctx, nrow, ncol, rowptr, colidx, val, cap, &A, &F, nblocks, block_ptr));
  isph_prec *M = new isph_prec();
  M->n = nrow;
  M->type = 2;
  M->ilu = F;
  *Aout = A;
  *Mout = M;
  return ISPH_SUCCESS;
}

int isph_mat_create_csr_bjacobi(isph_ctx *ctx, int nrow, int ncol, const int *rowptr, const int *colidx, const double *val,
                                int block_size, isph_mat **Aout, isph_prec **Mout) {
  ISPH_REQUIRE(ctx && Aout && Mout && rowptr && colidx && val, "NULL argument");
  ISPH_REQUIRE(nrow >= 0 && ncol >= nrow, "need 0 <= nrow <= ncol");
  isph_mat *A = nullptr;
  isph_ilu *F = nullptr;
  ISPH_CHECK(csr_ingress_host_bjacobi(ctx, nrow, ncol, rowptr, colidx, val, block_size, &A, &F));
  isph_prec *M = new isph_prec();
  M->n = nrow;
  M->type = 2;
  M->ilu = F;
  *Aout = A;
  *Mout = M;
  return ISPH_SUCCESS;
}

int isph_ingress_info(const isph_ctx *ctx, double info[8]) {
  ISPH_REQUIRE(ctx && info, "NULL argument");
  for (int k = 0; k < 8; ++k) info[k] = ctx->stager ? ctx->stager->stats[k] : 0.0;
  return ISPH_SUCCESS;
}

struct isph_halo_plan {
  isph_halo H;
  int nlocal = 0;
};

__global__ void k_gather_atoms(int n, int ncomp, const int *__restrict__ idx, const double *__restrict__ x,
                               double *__restrict__ out) {
  const long long total = (long long)n * ncomp;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(e / ncomp), c = (int)(e % ncomp);
    out[e] = x[(long long)idx[i] * ncomp + c];
  }
}

int isph_halo_create(isph_ctx *ctx, int nlocal, int npeers, const int *peer_rank, const int *send_ptr,
                     const int *send_idx, const int *recv_ptr, isph_halo_plan **plan) {
  ISPH_REQUIRE(ctx && plan && nlocal >= 0 && npeers >= 0, "NULL or negative argument");
  ISPH_REQUIRE(npeers == 0 || (peer_rank && send_ptr && recv_ptr), "NULL halo lists");
  ISPH_REQUIRE(npeers == 0 || comm_active(ctx), "a halo plan with peers needs a context made by isph_ctx_create_dist or isph_ctx_create_hostcomm");
  isph_halo_plan *P = new isph_halo_plan();
  P->nlocal = nlocal;
  isph_halo &H = P->H;
  H.npeers = npeers;
  H.peer.assign(peer_rank, peer_rank + npeers);
  if (npeers > 0) {
    H.send_ptr.assign(send_ptr, send_ptr + npeers + 1);
    H.recv_ptr.assign(recv_ptr, recv_ptr + npeers + 1);
  } else {
    H.send_ptr.assign(1, 0);
    H.recv_ptr.assign(1, 0);
  }
  H.nsend = H.send_ptr[(size_t)npeers];
  H.nrecv = H.recv_ptr[(size_t)npeers];
  int rc = ISPH_SUCCESS;
  for (int p = 0; p < npeers && rc == ISPH_SUCCESS; ++p) {
    if (peer_rank[p] < 0 || peer_rank[p] >= ctx->nranks) rc = fail("peer rank out of range", __FILE__, __LINE__);
    else if (H.send_ptr[(size_t)p + 1] < H.send_ptr[(size_t)p] || H.recv_ptr[(size_t)p + 1] < H.recv_ptr[(size_t)p])
      rc = fail("halo offsets must not decrease", __FILE__, __LINE__);
  }
  for (int k = 0; k < H.nsend && rc == ISPH_SUCCESS; ++k)
    if (send_idx[k] < 0 || send_idx[k] >= nlocal) rc = fail("send index out of range", __FILE__, __LINE__);
  if (rc == ISPH_SUCCESS) rc = H.send_idx.reserve((size_t)(H.nsend > 0 ? H.nsend : 1));
  if (rc == ISPH_SUCCESS && H.nsend > 0 &&
      (hipMemcpyAsync(H.send_idx.p, send_idx, sizeof(int) * (size_t)H.nsend, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
       hipStreamSynchronize(ctx->stream) != hipSuccess))
    rc = fail("copy of the send list failed", __FILE__, __LINE__);
  if (rc != ISPH_SUCCESS) { isph_halo_destroy(P); return rc; }
  *plan = P;
  return ISPH_SUCCESS;
}

void isph_halo_destroy(isph_halo_plan *P) {
  if (!P) return;
  P->H.send_idx.release();
  delete P;
}

int isph_halo_forward(isph_ctx *ctx, const isph_halo_plan *P, const double *x, double *ghosts, int ncomp, int on_device) {
  ISPH_REQUIRE(ctx && P && ncomp >= 1 && ncomp <= 16, "NULL plan or ncomp outside [1,16]");
  const isph_halo &H = P->H;
  if (H.nrecv == 0 && H.nsend == 0) return ISPH_SUCCESS;
  ISPH_REQUIRE(x && ghosts, "NULL field");
  ISPH_REQUIRE(comm_active(ctx), "no communicator");
  DevTmp<double> tx, tg;
  const double *dx = nullptr;
  int rc = stage_in(ctx, x, (size_t)P->nlocal * ncomp, on_device, tx, &dx);
  double *dg = ghosts;
  if (rc == ISPH_SUCCESS && !on_device) { rc = tg.reserve((size_t)(H.nrecv > 0 ? H.nrecv : 1) * ncomp); dg = tg.p; }
  if (rc == ISPH_SUCCESS && on_device && H.nrecv > 0 && !is_device_pointer(ghosts)) rc = fail("ghosts is not device memory (on_device = 1)", __FILE__, __LINE__);
  if (rc == ISPH_SUCCESS) rc = ctx->sendbuf.reserve((size_t)(H.nsend > 0 ? H.nsend : 1) * ncomp);
  if (rc == ISPH_SUCCESS) {
    if (H.nsend > 0)
      hipLaunchKernelGGL(k_gather_atoms, dim3(stream_grid((long long)H.nsend * ncomp)), dim3(kBlock), 0, ctx->stream, H.nsend,
                         ncomp, (const int *)H.send_idx.p, dx, ctx->sendbuf.p);
    rc = comm_exchange(ctx, H, ctx->sendbuf.p, dg, ncomp, false, ctx->stream);
  }
  if (rc == ISPH_SUCCESS && !on_device && H.nrecv > 0 &&
      hipMemcpyAsync(ghosts, dg, sizeof(double) * (size_t)H.nrecv * ncomp, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
    rc = fail("copy of the ghost values failed", __FILE__, __LINE__);
  if (rc == ISPH_SUCCESS && !on_device && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail("halo exchange failed", __FILE__, __LINE__);
  tx.release(); tg.release();
  return rc;
}

int isph_mat_set_halo(isph_ctx *ctx, isph_mat *A, int npeers, const int *peer_rank, const int *send_ptr,
                      const int *send_idx, const int *recv_ptr) {
  ISPH_REQUIRE(ctx && A, "NULL argument");
  isph_halo &H = A->halo;
  H.npeers = npeers;
  H.peer.assign(peer_rank, peer_rank + npeers);
  H.send_ptr.assign(send_ptr, send_ptr + npeers + 1);
  H.recv_ptr.assign(recv_ptr, recv_ptr + npeers + 1);
  H.nsend = H.send_ptr[npeers];
  H.nrecv = H.recv_ptr[npeers];
  ISPH_REQUIRE(H.nrecv == A->S.ncol - A->S.nrow, "halo receive count != number of ghost columns");
  for (int p = 0; p < npeers; ++p) ISPH_REQUIRE(peer_rank[p] >= 0 && peer_rank[p] < ctx->nranks, "peer rank out of range");
  for (int k = 0; k < H.nsend; ++k) ISPH_REQUIRE(send_idx[k] >= 0 && send_idx[k] < A->S.nrow, "send index out of range");
  ISPH_CHECK(H.send_idx.reserve((size_t)(H.nsend > 0 ? H.nsend : 1)));
  if (H.nsend > 0) {
    ISPH_CHECK_HIP(hipMemcpyAsync(H.send_idx.p, send_idx, sizeof(int) * (size_t)H.nsend, hipMemcpyHostToDevice, ctx->stream));
    if (A->order) {  // the caller lists its own rows; the packed values are read from vectors in the matrix' numbering
      DevTmp<int> t;
      ISPH_CHECK(t.reserve((size_t)H.nsend));
      ISPH_CHECK_HIP(hipMemcpyAsync(t.p, H.send_idx.p, sizeof(int) * (size_t)H.nsend, hipMemcpyDeviceToDevice, ctx->stream));
      hipLaunchKernelGGL(k_perm_map_indices, dim3(perm_grid(H.nsend)), dim3(kBlock), 0, ctx->stream, H.nsend, A->S.nrow,
                         (const int *)A->order->iperm.p, (const int *)t.p, H.send_idx.p);
      ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    }
    ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  }
  // interior / boundary slices: the interior ones (no ghost column) run while the exchange is in flight
  const Sell &S = A->S;
  H.n_int = H.n_bnd = 0;
  if (S.nslices > 0) {
    DevTmp<int> flag;
    ISPH_CHECK(flag.reserve((size_t)S.nslices));
    hipLaunchKernelGGL(k_sell_flag_ghost_slices, dim3((S.nslices + 3) / 4), dim3(kBlock), 0, ctx->stream, S.nrow, S.nslices,
                       (const long long *)S.slice_off.p, (const int *)S.col.p, flag.p);
    std::vector<int> hf((size_t)S.nslices), li, lb;
    ISPH_CHECK_HIP(hipMemcpyAsync(hf.data(), flag.p, sizeof(int) * (size_t)S.nslices, hipMemcpyDeviceToHost, ctx->stream));
    ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    flag.release();
    for (int s = 0; s < S.nslices; ++s) (hf[(size_t)s] ? lb : li).push_back(s);
    H.n_int = (int)li.size();
    H.n_bnd = (int)lb.size();
    ISPH_CHECK(H.list_int.reserve(li.size() > 0 ? li.size() : 1));
    ISPH_CHECK(H.list_bnd.reserve(lb.size() > 0 ? lb.size() : 1));
    if (!li.empty()) ISPH_CHECK_HIP(hipMemcpyAsync(H.list_int.p, li.data(), sizeof(int) * li.size(), hipMemcpyHostToDevice, ctx->stream));
    if (!lb.empty()) ISPH_CHECK_HIP(hipMemcpyAsync(H.list_bnd.p, lb.data(), sizeof(int) * lb.size(), hipMemcpyHostToDevice, ctx->stream));
    ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  }
  return ISPH_SUCCESS;
}

int isph_mat_info(const isph_mat *A, long long info[6]) {
  ISPH_REQUIRE(A && info, "NULL argument");
  info[0] = A->S.nrow; info[1] = A->S.ncol; info[2] = A->S.nnz; info[3] = A->S.nslices;
  info[4] = A->S.stored; info[5] = A->S.stored * 12 + ((long long)A->S.nslices + 1) * 8;
  return ISPH_SUCCESS;
}

int isph_mat_export_rows(isph_ctx *ctx, const isph_mat *A, int row_begin, int nrows, long long *rowptr, int *colidx,
                         double *val, long long capacity) {
  ISPH_REQUIRE(ctx && A && rowptr && colidx && val, "NULL argument");
  const Sell &S = A->S;
  ISPH_REQUIRE(row_begin >= 0 && nrows >= 0 && (long long)row_begin + nrows <= S.nrow, "row range outside the matrix");
  if (A->order && nrows > 0) {
    // the caller's rows are scattered in the matrix: one row at a time through the ranged kernel, columns translated
    // back and sorted (a host-side check of a few thousand rows)
    std::vector<int> hperm((size_t)S.nrow), hiperm((size_t)S.nrow);
    ISPH_CHECK_HIP(hipMemcpyAsync(hperm.data(), A->order->perm.p, sizeof(int) * hperm.size(), hipMemcpyDeviceToHost, ctx->stream));
    ISPH_CHECK_HIP(hipMemcpyAsync(hiperm.data(), A->order->iperm.p, sizeof(int) * hiperm.size(), hipMemcpyDeviceToHost, ctx->stream));
    ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    isph_mat plain;           // the same storage without the numbering: borrowed, never released
    plain.S = S;
    rowptr[0] = 0;
    std::vector<int> ord;
    std::vector<int> tc;
    std::vector<double> tv;
    for (int i = 0; i < nrows; ++i) {
      const int r = hiperm[(size_t)(row_begin + i)];
      long long rp1[2];
      const long long room = capacity - rowptr[i];
      ISPH_CHECK(isph_mat_export_rows(ctx, &plain, r, 1, rp1, colidx + rowptr[i], val + rowptr[i], room));
      const long long len1 = rp1[1];
      ord.resize((size_t)len1); tc.assign(colidx + rowptr[i], colidx + rowptr[i] + len1); tv.assign(val + rowptr[i], val + rowptr[i] + len1);
      for (long long k = 0; k < len1; ++k) { if (tc[(size_t)k] < S.nrow) tc[(size_t)k] = hperm[(size_t)tc[(size_t)k]]; ord[(size_t)k] = (int)k; }
      std::sort(ord.begin(), ord.end(), [&](int a, int c) { return tc[(size_t)a] < tc[(size_t)c]; });
      for (long long k = 0; k < len1; ++k) { colidx[rowptr[i] + k] = tc[(size_t)ord[(size_t)k]]; val[rowptr[i] + k] = tv[(size_t)ord[(size_t)k]]; }
      rowptr[i + 1] = rowptr[i] + len1;
    }
    return ISPH_SUCCESS;
  }
  std::vector<int> len((size_t)(nrows > 0 ? nrows : 1));
  if (nrows > 0) ISPH_CHECK_HIP(hipMemcpyAsync(len.data(), S.rowlen.p + row_begin, sizeof(int) * (size_t)nrows, hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  rowptr[0] = 0;
  for (int i = 0; i < nrows; ++i) rowptr[i + 1] = rowptr[i] + len[(size_t)i];
  const long long nnz = rowptr[nrows];
  ISPH_REQUIRE(nnz <= capacity, "isph_mat_export_rows: colidx / val too small for the requested rows");
  if (nnz == 0) return ISPH_SUCCESS;
  DevTmp<long long> drp;
  DevTmp<int> dci;
  DevTmp<double> dv;
  ISPH_CHECK(drp.reserve((size_t)nrows + 1));
  ISPH_CHECK(dci.reserve((size_t)nnz));
  ISPH_CHECK(dv.reserve((size_t)nnz));
  ISPH_CHECK_HIP(hipMemcpyAsync(drp.p, rowptr, sizeof(long long) * ((size_t)nrows + 1), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_sell_rows_to_csr, dim3((nrows + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, row_begin, nrows,
                     (const int *)S.rowlen.p, (const long long *)S.slice_off.p, (const int *)S.col.p, (const double *)S.val.p,
                     (const long long *)drp.p, dci.p, dv.p);
  ISPH_CHECK_HIP(hipMemcpyAsync(colidx, dci.p, sizeof(int) * (size_t)nnz, hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipMemcpyAsync(val, dv.p, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  ISPH_CHECK_HIP(hipGetLastError());
  return ISPH_SUCCESS;
}

int isph_mat_export_csr(isph_ctx *ctx, const isph_mat *A, int *rowptr, int *colidx, double *val) {
  ISPH_REQUIRE(ctx && A && rowptr && colidx && val, "NULL argument");
  const Sell &S = A->S;
  std::vector<int> len((size_t)S.nrow);
  ISPH_CHECK_HIP(hipMemcpyAsync(len.data(), S.rowlen.p, sizeof(int) * (size_t)S.nrow, hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  std::vector<long long> rp((size_t)S.nrow + 1, 0);
  for (int i = 0; i < S.nrow; ++i) rp[(size_t)i + 1] = rp[(size_t)i] + len[(size_t)i];
  const long long nnz = rp[(size_t)S.nrow];
  ISPH_REQUIRE(nnz < 2147483647LL, "nnz exceeds 32-bit CSR export");
  DevTmp<long long> drp;
  DevTmp<int> dci;
  DevTmp<double> dv;
  ISPH_CHECK(drp.reserve((size_t)S.nrow + 1));
  ISPH_CHECK(dci.reserve((size_t)(nnz > 0 ? nnz : 1)));
  ISPH_CHECK(dv.reserve((size_t)(nnz > 0 ? nnz : 1)));
  ISPH_CHECK_HIP(hipMemcpyAsync(drp.p, rp.data(), sizeof(long long) * ((size_t)S.nrow + 1), hipMemcpyHostToDevice, ctx->stream));
  if (S.nrow > 0)
    hipLaunchKernelGGL(k_sell_to_csr, dim3((S.nrow + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, S.nrow,
                       S.rowlen.p, S.slice_off.p, S.col.p, S.val.p, drp.p, dci.p, dv.p);
  std::vector<int> ci((size_t)nnz);
  std::vector<double> v((size_t)nnz);
  ISPH_CHECK_HIP(hipMemcpyAsync(ci.data(), dci.p, sizeof(int) * (size_t)nnz, hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipMemcpyAsync(v.data(), dv.p, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  drp.release(); dci.release(); dv.release();
  // the library's own row numbering: the caller sees its rows and columns (ghost columns are not renumbered)
  std::vector<int> hperm, hiperm;
  if (A->order && S.nrow > 0) {
    hperm.resize((size_t)S.nrow); hiperm.resize((size_t)S.nrow);
    ISPH_CHECK_HIP(hipMemcpyAsync(hperm.data(), A->order->perm.p, sizeof(int) * hperm.size(), hipMemcpyDeviceToHost, ctx->stream));
    ISPH_CHECK_HIP(hipMemcpyAsync(hiperm.data(), A->order->iperm.p, sizeof(int) * hiperm.size(), hipMemcpyDeviceToHost, ctx->stream));
    ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    for (long long k = 0; k < nnz; ++k)
      if (ci[(size_t)k] < S.nrow) ci[(size_t)k] = hperm[(size_t)ci[(size_t)k]];
  }
  // sort columns inside each row (Epetra OptimizeStorage order)
  std::vector<int> perm;
  long long q = 0;
  for (int i = 0; i < S.nrow; ++i) {
    const int r = hiperm.empty() ? i : hiperm[(size_t)i];
    const long long b = rp[(size_t)r], e = rp[(size_t)r + 1];
    perm.resize((size_t)(e - b));
    std::iota(perm.begin(), perm.end(), 0);
    std::sort(perm.begin(), perm.end(), [&](int a, int c) { return ci[(size_t)(b + a)] < ci[(size_t)(b + c)]; });
    rowptr[i] = (int)q;
    for (long long k = b; k < e; ++k, ++q) {
      colidx[q] = ci[(size_t)(b + perm[(size_t)(k - b)])];
      val[q] = v[(size_t)(b + perm[(size_t)(k - b)])];
    }
  }
  rowptr[S.nrow] = (int)nnz;
  return ISPH_SUCCESS;
}

void isph_mat_destroy(isph_mat *A) {
  if (!A) return;
  A->S.release();
  A->halo.send_idx.release();
  A->halo.list_int.release();
  A->halo.list_bnd.release();
  delete A;
}

int isph_spmv(isph_ctx *ctx, const isph_mat *A, const double *x, double *y, int on_device) {
  ISPH_REQUIRE(ctx && A && x && y, "NULL argument");
  const Sell &S = A->S;
  const size_t nx = (S.ncol > S.nrow && A->halo.npeers == 0) ? (size_t)S.ncol : (size_t)S.nrow;
  if (A->order) {
    // x and y are the caller's: x into the matrix' numbering (ghost entries, when the caller supplies them, in place),
    // the product back out
    DevTmp<double> xc, xi, yi;
    const double *dxc = x;
    if (!on_device) {
      ISPH_CHECK(xc.reserve(nx));
      ISPH_CHECK_HIP(hipMemcpyAsync(xc.p, x, sizeof(double) * nx, hipMemcpyHostToDevice, ctx->stream));
      dxc = xc.p;
    }
    ISPH_CHECK(xi.reserve((size_t)S.ncol + 64));
    ISPH_CHECK(yi.reserve((size_t)S.nrow + 64));
    hipLaunchKernelGGL((k_perm_gather<double>), dim3(perm_grid((long long)nx)), dim3(kBlock), 0, ctx->stream, (long long)nx, S.nrow, 1,
                       (const int *)A->order->perm.p, dxc, xi.p);
    if (S.ncol > S.nrow && A->halo.npeers == 0) {
      int nbp = 0;
      const int grid = spmv_grid(S.nslices, &nbp);
      hipLaunchKernelGGL((k_sell_spmv<8, false, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, S.nrow, S.nslices, nbp,
                         S.slice_off.p, S.col.p, S.val.p, (const double *)xi.p, yi.p, (const double *)nullptr, (double *)nullptr);
    } else {
      ISPH_CHECK(spmv_dev(ctx, A, xi.p, yi.p, nullptr));
    }
    double *dy = y;
    if (!on_device) { ISPH_CHECK(xc.reserve((size_t)S.nrow)); dy = xc.p; }
    hipLaunchKernelGGL((k_perm_scatter<double>), dim3(perm_grid(S.nrow)), dim3(kBlock), 0, ctx->stream, S.nrow,
                       (const int *)A->order->perm.p, (const double *)yi.p, dy);
    if (!on_device) ISPH_CHECK_HIP(hipMemcpyAsync(y, dy, sizeof(double) * (size_t)S.nrow, hipMemcpyDeviceToHost, ctx->stream));
    ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));  // the temporaries go back to the pool
    ISPH_CHECK_HIP(hipGetLastError());
    return ISPH_SUCCESS;
  }
  if (on_device) {
    ISPH_CHECK(spmv_dev(ctx, A, x, y, nullptr));
    return ISPH_SUCCESS;
  }
  ISPH_CHECK(ctx->xdev.reserve((size_t)S.ncol));
  ISPH_CHECK(ctx->bdev.reserve((size_t)(S.nrow > 0 ? S.nrow : 1)));
  ISPH_CHECK_HIP(hipMemcpyAsync(ctx->xdev.p, x, sizeof(double) * nx, hipMemcpyHostToDevice, ctx->stream));
  if (S.ncol > S.nrow && A->halo.npeers == 0) {
    // caller supplied all ncol entries (ghost values included): plain kernel
    int nbp = 0;
    const int grid = spmv_grid(S.nslices, &nbp);
    hipLaunchKernelGGL((k_sell_spmv<8, false, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, S.nrow, S.nslices, nbp,
                       S.slice_off.p, S.col.p, S.val.p, (const double *)ctx->xdev.p, ctx->bdev.p, (const double *)nullptr,
                       (double *)nullptr);
  } else {
    ISPH_CHECK(spmv_dev(ctx, A, ctx->xdev.p, ctx->bdev.p, nullptr));
  }
  ISPH_CHECK_HIP(hipMemcpyAsync(y, ctx->bdev.p, sizeof(double) * (size_t)S.nrow, hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  ISPH_CHECK_HIP(hipGetLastError());
  return ISPH_SUCCESS;
}

int isph_spmv_time(isph_ctx *ctx, const isph_mat *A, const double *x_dev, double *y_dev, int reps, int variant, double *avg_ms) {
  ISPH_REQUIRE(ctx && A && x_dev && y_dev && avg_ms && reps > 0, "bad argument");
  const Sell &S = A->S;
  ISPH_REQUIRE(S.ncol == S.nrow || A->halo.npeers > 0, "ghost columns without halo plan");
  if (S.ncol != S.nrow) {  // matrix with a halo: time the production path (exchange + interior/boundary launches)
    ISPH_CHECK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (int r = 0; r < reps; ++r) ISPH_CHECK(spmv_dev(ctx, A, x_dev, y_dev, nullptr));
    ISPH_CHECK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    ISPH_CHECK_HIP(hipEventSynchronize(ctx->ev1));
    float msh = 0.f;
    ISPH_CHECK_HIP(hipEventElapsedTime(&msh, ctx->ev0, ctx->ev1));
    *avg_ms = (double)msh / reps;
    return ISPH_SUCCESS;
  }
  const double *xuse = x_dev;
  int nbp = 0;
  const int grid = spmv_grid(S.nslices, &nbp);
  // kernel-tuning aid: `variant` selects an experimental instantiation for this timing call only (0: production)
  ISPH_CHECK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
#define ISPH_SPMV_LAUNCH(U, NTF)                                                                                      \
  hipLaunchKernelGGL((k_sell_spmv<U, false, NTF>), dim3(grid), dim3(kBlock), 0, ctx->stream, S.nrow, S.nslices, nbp, \
                     S.slice_off.p, S.col.p, S.val.p, xuse, y_dev, (const double *)nullptr, (double *)nullptr)
  for (int r = 0; r < reps; ++r) {
    switch (variant) {
      case 1: ISPH_SPMV_LAUNCH(8, false); break;
      case 2: ISPH_SPMV_LAUNCH(4, true); break;
      case 3: ISPH_SPMV_LAUNCH(12, true); break;
      case 4: ISPH_SPMV_LAUNCH(2, false); break;
      case 5: ISPH_SPMV_LAUNCH(6, false); break;
      case 6: ISPH_SPMV_LAUNCH(4, false); break;
      default: ISPH_SPMV_LAUNCH(8, true); break;  // production instantiation
    }
  }
#undef ISPH_SPMV_LAUNCH
  ISPH_CHECK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
  ISPH_CHECK_HIP(hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  ISPH_CHECK_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *avg_ms = (double)ms / reps;
  ISPH_CHECK_HIP(hipGetLastError());
  return ISPH_SUCCESS;
}

/* ---- preconditioner --------------------------------------------------- */

int isph_prec_create_blocks(isph_ctx *ctx, const isph_mat *A, int nblocks, const int *block_ptr, isph_prec **Mout) {
  return isph_prec_create_blocks_fill(ctx, A, nblocks, block_ptr, 0, Mout);
}

int isph_prec_create_blocks_fill(isph_ctx *ctx, const isph_mat *A, int nblocks, const int *block_ptr, int level_of_fill,
                                 isph_prec **Mout) {
  ISPH_REQUIRE(ctx && A && Mout && nblocks > 0 && block_ptr, "NULL argument or no subdomains");
  ISPH_REQUIRE(level_of_fill >= 0 && level_of_fill <= 8, "level of fill must be in [0,8]");
  ISPH_REQUIRE(!is_device_pointer(block_ptr), "block_ptr must be a host array");
  ISPH_REQUIRE(!A->order, "the matrix was assembled in the library's own row numbering: its subdomains are the library's "
                          "bricks (isph_prec_create with block_size 0); a table over the caller's rows needs "
                          "isph_ctx_set_ordering(ctx, 0) before the assembly");
  int cap = 64;
  for (int b = 0; b < nblocks; ++b) cap = std::max(cap, block_ptr[b + 1] - block_ptr[b]);
  cap = (cap + 63) / 64 * 64;
  ISPH_REQUIRE(cap <= 1024, "a subdomain of the block stream holds at most 1024 rows (isph_prec_create_schwarz takes larger ones)");
  isph_prec *M = new isph_prec();
  M->n = A->S.nrow;
  M->type = 2;
  const int rc = ilu_create(ctx, A, cap, &M->ilu, /*sgs=*/false, level_of_fill, nblocks, block_ptr);
  if (rc != ISPH_SUCCESS) { isph_prec_destroy(M); return rc; }
  *Mout = M;
  return ISPH_SUCCESS;
}

int isph_prec_create(isph_ctx *ctx, const isph_mat *A, const char *type, int block_size, isph_prec **Mout) {
  ISPH_REQUIRE(ctx && A && type && Mout, "NULL argument");
  isph_prec *M = new isph_prec();
  M->n = A->S.nrow;
  int rc = ISPH_SUCCESS;
  if (!strcmp(type, "none")) {
    M->type = 0;
  } else if (!strcmp(type, "jacobi")) {
    M->type = 1;
    rc = M->invdiag.reserve((size_t)(M->n > 0 ? M->n : 1));
    if (rc == ISPH_SUCCESS && M->n > 0) {
      hipLaunchKernelGGL(k_sell_inv_diag, dim3((M->n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, M->n,
                         A->S.rowlen.p, A->S.slice_off.p, A->S.col.p, A->S.val.p, M->invdiag.p);
      if (hipGetLastError() != hipSuccess) rc = fail("jacobi setup failed", __FILE__, __LINE__);
    }
  } else if (!strncmp(type, "bjacobi-ilu", 11) && type[11] >= '0' && type[11] <= '8' && type[12] == 0) {
    // "bjacobi-ilu<k>": "fact: level-of-fill" = k (precond_ifpack.h:35)
    M->type = 2;
    if (block_size <= 0) {
      // the matrix' own subdomains: the bricks the assembly sorted the particles into (order.hpp)
      if (!A->order) {
        rc = fail("block_size 0 selects the library's own subdomains: the matrix must come from an assembly entry point "
                  "with isph_ctx_set_ordering(ctx, 1)", __FILE__, __LINE__);
      } else {
        const std::vector<int> &bp = A->order->block_ptr;
        int cap = 64;
        for (size_t b = 0; b + 1 < bp.size(); ++b) cap = std::max(cap, bp[b + 1] - bp[b]);
        cap = (cap + 63) / 64 * 64;
        rc = ilu_create(ctx, A, cap, &M->ilu, /*sgs=*/false, /*fill=*/type[11] - '0', (int)bp.size() - 1, bp.data());
      }
    } else {
      rc = ilu_create(ctx, A, block_size, &M->ilu, /*sgs=*/false, /*fill=*/type[11] - '0');
    }
  } else if (!strncmp(type, "ilu", 3) && type[3] >= '0' && type[3] <= '8' && type[4] == 0) {
    // "ilu<k>": ILU(k) of the whole local matrix -- what Ifpack factors on one MPI rank (the overlap is a no-op there)
    M->type = 4;
    rc = schwarz_create(ctx, A, type[3] - '0', /*block_size=*/0, /*overlap=*/0, /*combine=*/0, &M->schwarz);
  } else if (!strcmp(type, "sa-amg")) {
    // PrecondWrapper_ML defaults without a null vector; block_size is the Gauss-Seidel block of the fine level.
    // isph_prec_create_amg takes the full parameter set and the null vector of a singular system.
    isph_amg_params prm;
    isph_amg_params_default(&prm);
    prm.block = block_size;
    M->type = 3;
    rc = amg_create(ctx, A, &prm, nullptr, &M->amg);
  } else {
    rc = fail("unknown preconditioner type (none|jacobi|bjacobi-ilu<k>|ilu<k>, k = 0..8|sa-amg)", __FILE__, __LINE__);
  }
  if (rc != ISPH_SUCCESS) { isph_prec_destroy(M); return rc; }
  M->order = A->order;
  *Mout = M;
  return ISPH_SUCCESS;
}

int isph_prec_apply(isph_ctx *ctx, const isph_prec *M, const double *r, double *z, int on_device) {
  ISPH_REQUIRE(ctx && M && r && z, "NULL argument");
  if (M->order && M->n > 0) {  // r and z are the caller's; the preconditioner lives in the numbering of its matrix
    const size_t n = (size_t)M->n;
    DevTmp<double> rc, ri, zi;
    const double *drc = r;
    if (!on_device) {
      ISPH_CHECK(rc.reserve(n));
      ISPH_CHECK_HIP(hipMemcpyAsync(rc.p, r, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
      drc = rc.p;
    }
    ISPH_CHECK(ri.reserve(n + 64));
    ISPH_CHECK(zi.reserve(n + 64));
    hipLaunchKernelGGL((k_perm_gather<double>), dim3(perm_grid((long long)n)), dim3(kBlock), 0, ctx->stream, (long long)n, M->n, 1,
                       (const int *)M->order->perm.p, drc, ri.p);
    ISPH_CHECK(prec_apply_dev(ctx, M, ri.p, zi.p));
    double *dz = on_device ? z : rc.p;
    hipLaunchKernelGGL((k_perm_scatter<double>), dim3(perm_grid(M->n)), dim3(kBlock), 0, ctx->stream, M->n,
                       (const int *)M->order->perm.p, (const double *)zi.p, dz);
    if (!on_device) ISPH_CHECK_HIP(hipMemcpyAsync(z, dz, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    ISPH_CHECK_HIP(hipGetLastError());
    return prec_health(ctx, M);
  }
  if (on_device) return prec_apply_dev(ctx, M, r, z);
  const size_t n = (size_t)M->n;
  ISPH_CHECK(ctx->xdev.reserve(n > 0 ? n : 1));
  ISPH_CHECK(ctx->bdev.reserve(n > 0 ? n : 1));
  ISPH_CHECK_HIP(hipMemcpyAsync(ctx->xdev.p, r, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
  ISPH_CHECK(prec_apply_dev(ctx, M, ctx->xdev.p, ctx->bdev.p));
  ISPH_CHECK_HIP(hipMemcpyAsync(z, ctx->bdev.p, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  ISPH_CHECK_HIP(hipGetLastError());
  return prec_health(ctx, M);
}

int isph_prec_export_ilu(isph_ctx *ctx, const isph_prec *M, int *rowptr, int *colidx, double *val) {
  ISPH_REQUIRE(ctx && M && M->type == 2 && M->ilu, "not an ILU preconditioner");
  return ilu_export(ctx, M->ilu, rowptr, colidx, val);
}

int isph_prec_info(isph_ctx *ctx, const isph_prec *M, long long info[4]) {
  ISPH_REQUIRE(ctx && M && info, "NULL argument");
  info[0] = info[1] = info[2] = info[3] = 0;
  if (M->type != 2 || !M->ilu) return ISPH_SUCCESS;
  const isph_ilu *F = M->ilu;
  std::vector<int> bi((size_t)4 * F->nblocks);  // per block: chunks of the L / U stream, rows in the L / U stream
  ISPH_CHECK_HIP(hipMemcpyAsync(bi.data(), F->blkinfo.p, sizeof(int) * bi.size(), hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  long long used = 0;
  for (size_t b = 0; b < (size_t)F->nblocks; ++b) used += bi[4 * b] + bi[4 * b + 1];
  info[0] = ilu_nnz(F); info[1] = used; info[2] = F->stream_chunks; info[3] = F->nblocks;
  return ISPH_SUCCESS;
}

long long isph_prec_nnz(const isph_prec *M) { return (M && M->type == 2 && M->ilu) ? ilu_nnz(M->ilu) : 0; }

void isph_prec_destroy(isph_prec *M) {
  if (!M) return;
  M->invdiag.release();
  if (M->ilu) ilu_destroy(M->ilu);
  if (M->amg) amg_destroy(M->amg);
  if (M->schwarz) schwarz_destroy(M->schwarz);
  if (M->ovl) overlap_destroy(M->ovl);
  delete M;
}

/* ---- additive Schwarz ILU(k) with the reference's Ifpack semantics ------ */

void isph_schwarz_params_default(isph_schwarz_params *p) {
  // PrecondWrapper_Ifpack::setParameters(NULL), ref: precond_ifpack.h:30-45
  p->level_of_fill = 1; p->overlap = 1; p->combine = 0; p->block_size = 0; p->level_launches = 0;
}

int isph_prec_create_schwarz(isph_ctx *ctx, const isph_mat *A, const isph_schwarz_params *prm, isph_prec **Mout) {
  ISPH_REQUIRE(ctx && A && prm && Mout, "NULL argument");
  isph_prec *M = new isph_prec();
  M->n = A->S.nrow;
  M->type = 4;
  const int rc = schwarz_create(ctx, A, prm->level_of_fill, prm->block_size, prm->overlap, prm->combine, &M->schwarz,
                                /*syncfree=*/prm->level_launches == 0);
  if (rc != ISPH_SUCCESS) { isph_prec_destroy(M); return rc; }
  M->order = A->order;
  *Mout = M;
  return ISPH_SUCCESS;
}

int isph_prec_schwarz_info(const isph_prec *M, long long info[7]) {
  ISPH_REQUIRE(M && M->type == 4 && M->schwarz && info, "not a Schwarz preconditioner");
  const isph_schwarz *S = M->schwarz;
  info[0] = S->nloc; info[1] = S->nnz; info[2] = S->nsub; info[3] = S->nlev_l; info[4] = S->nlev_u; info[5] = S->maxrow; info[6] = S->subsweep ? 2 : S->syncfree ? 1 : 0;
  return ISPH_SUCCESS;
}

int isph_prec_schwarz_timing(const isph_prec *M, double ms[6]) {
  ISPH_REQUIRE(M && M->type == 4 && M->schwarz && ms, "not a Schwarz preconditioner");
  for (int k = 0; k < 6; ++k) ms[k] = M->schwarz->t_ms[k];
  return ISPH_SUCCESS;
}

int isph_prec_schwarz_export(isph_ctx *ctx, const isph_prec *M, int *rows, int *loc_ptr, long long *rowptr, int *colidx,
                             double *val) {
  ISPH_REQUIRE(ctx && M && M->type == 4 && M->schwarz && rows && loc_ptr && rowptr && colidx && val, "bad argument");
  return schwarz_export(ctx, M->schwarz, rows, loc_ptr, rowptr, colidx, val);
}

int isph_prec_create_overlap(isph_ctx *ctx, const isph_mat *Aext, int nlocal, int level_of_fill, int combine, int npeers,
                             const int *peer_rank, const int *send_ptr, const int *send_idx, const int *recv_ptr,
                             isph_prec **Mout) {
  ISPH_REQUIRE(ctx && Aext && Mout && nlocal >= 0 && npeers >= 0 && (combine == 0 || combine == 1), "bad argument");
  ISPH_REQUIRE(npeers == 0 || (peer_rank && send_ptr && send_idx && recv_ptr), "NULL halo lists");
  ISPH_REQUIRE(npeers == 0 || comm_active(ctx), "overlap across ranks needs a context made by isph_ctx_create_dist or isph_ctx_create_hostcomm");
  const int nrecv = npeers > 0 ? recv_ptr[npeers] : 0, nsend = npeers > 0 ? send_ptr[npeers] : 0;
  ISPH_REQUIRE(Aext->S.nrow == nlocal + nrecv && Aext->S.ncol == Aext->S.nrow,
               "the extended matrix must be square with nlocal + (number of ghost columns) rows");
  for (int p = 0; p < npeers; ++p) ISPH_REQUIRE(peer_rank[p] >= 0 && peer_rank[p] < ctx->nranks, "peer rank out of range");
  for (int k = 0; k < nsend; ++k) ISPH_REQUIRE(send_idx[k] >= 0 && send_idx[k] < nlocal, "send index out of range");
  isph_prec *M = new isph_prec();
  M->n = nlocal;
  M->type = 5;
  isph_overlap *O = new isph_overlap();
  M->ovl = O;
  O->n = nlocal; O->next = nlocal + nrecv; O->combine = combine;
  isph_halo &H = O->H;
  H.npeers = npeers;
  H.peer.assign(peer_rank, peer_rank + npeers);
  H.send_ptr.assign(1, 0); H.recv_ptr.assign(1, 0);
  if (npeers > 0) { H.send_ptr.assign(send_ptr, send_ptr + npeers + 1); H.recv_ptr.assign(recv_ptr, recv_ptr + npeers + 1); }
  H.nsend = nsend; H.nrecv = nrecv;
  int rc = H.send_idx.reserve((size_t)(nsend > 0 ? nsend : 1));
  if (rc == ISPH_SUCCESS && nsend > 0 &&
      hipMemcpyAsync(H.send_idx.p, send_idx, sizeof(int) * (size_t)nsend, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
    rc = fail("copy of the send list failed", __FILE__, __LINE__);
  if (rc == ISPH_SUCCESS) rc = O->rext.reserve((size_t)(O->next > 0 ? O->next : 1));
  if (rc == ISPH_SUCCESS) rc = O->zext.reserve((size_t)(O->next > 0 ? O->next : 1));
  if (rc == ISPH_SUCCESS) rc = O->sbuf.reserve((size_t)(nsend > 0 ? nsend : 1));
  if (rc == ISPH_SUCCESS) rc = O->rbuf.reserve((size_t)(nsend > 0 ? nsend : 1));
  // one subdomain = the whole extended matrix, no further layers inside it
  if (rc == ISPH_SUCCESS) rc = schwarz_create(ctx, Aext, level_of_fill, /*block_size=*/0, /*overlap=*/0, /*combine=*/1, &O->inner);
  if (rc == ISPH_SUCCESS && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail("overlap set-up failed", __FILE__, __LINE__);
  if (rc != ISPH_SUCCESS) { isph_prec_destroy(M); return rc; }
  *Mout = M;
  return ISPH_SUCCESS;
}

/* ---- SA-AMG ----------------------------------------------------------- */

void isph_amg_params_default(isph_amg_params *p) {
  // PrecondWrapper_ML::setParameters(NULL), ref: precond_ml.h:44-55, plus ML's own defaults
  p->max_levels = 5; p->coarse_max = 128; p->omega = 4.0 / 3.0; p->block = 512; p->sweeps = 1; p->theta = 0.0;
  p->smoother = 0;
}

int isph_prec_create_amg(isph_ctx *ctx, const isph_mat *A, const isph_amg_params *prm, const double *nullvec,
                         int on_device, isph_prec **Mout) {
  ISPH_REQUIRE(ctx && A && Mout, "NULL argument");
  isph_amg_params def;
  isph_amg_params_default(&def);
  if (!prm) prm = &def;
  isph_prec *M = new isph_prec();
  M->n = A->S.nrow;
  M->type = 3;
  DevTmp<double> tn, tp;
  const double *dn = nullptr;
  int rc = nullvec ? stage_in(ctx, nullvec, (size_t)M->n, on_device, tn, &dn) : ISPH_SUCCESS;
  if (rc == ISPH_SUCCESS && dn && A->order && M->n > 0) {  // the caller's null vector in the matrix' numbering
    rc = tp.reserve((size_t)M->n);
    if (rc == ISPH_SUCCESS) {
      hipLaunchKernelGGL((k_perm_gather<double>), dim3(perm_grid(M->n)), dim3(kBlock), 0, ctx->stream, (long long)M->n, M->n, 1,
                         (const int *)A->order->perm.p, dn, tp.p);
      dn = tp.p;
    }
  }
  if (rc == ISPH_SUCCESS) rc = amg_create(ctx, A, prm, dn, &M->amg);
  M->order = A->order;
  tn.release(); tp.release();
  if (comm_active(ctx) && ctx->nranks > 1) {
    // Every level exchanges halos inside the cycle, and how often depends on the depth of the hierarchy and on the coarse
    // solver.  amg_create agrees on both between the ranks while it builds (amg.hpp); this is the last word on the outcome:
    // ranks that disagree would wait for each other forever -- fail on every rank together instead.
    double h[3] = {rc == ISPH_SUCCESS ? 1.0 : 0.0, (rc == ISPH_SUCCESS && M->amg->nlev > 1) ? 1.0 : 0.0,
                   (rc == ISPH_SUCCESS && M->amg->coarse_smooth) ? 1.0 : 0.0};
    DevTmp<double> d;
    int rc2 = d.reserve(3);
    if (rc2 == ISPH_SUCCESS && hipMemcpyAsync(d.p, h, sizeof(h), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc2 = ISPH_FAILURE;
    if (rc2 == ISPH_SUCCESS) rc2 = allreduce_inplace(ctx, d.p, 3);
    if (rc2 == ISPH_SUCCESS && (hipMemcpyAsync(h, d.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                                hipStreamSynchronize(ctx->stream) != hipSuccess)) rc2 = ISPH_FAILURE;
    d.release();
    const double nr = (double)ctx->nranks;
    if (rc2 != ISPH_SUCCESS) rc = fail("AMG: set-up consensus between the ranks failed", __FILE__, __LINE__);
    else if (h[0] != nr) rc = rc != ISPH_SUCCESS ? rc : fail("AMG: set-up failed on another rank", __FILE__, __LINE__);
    else if ((h[1] != 0.0 && h[1] != nr) || (h[1] == 0.0 && h[2] != 0.0 && h[2] != nr))
      rc = fail("AMG: the ranks built hierarchies of different depth; their fine-level halo exchanges would not match", __FILE__, __LINE__);
  }
  if (rc != ISPH_SUCCESS) { isph_prec_destroy(M); return rc; }
  *Mout = M;
  return ISPH_SUCCESS;
}

int isph_prec_amg_levels(const isph_prec *M) { return (M && M->type == 3 && M->amg) ? M->amg->nlev : 0; }

int isph_prec_amg_info(isph_ctx *ctx, const isph_prec *M, int level, long long info[3]) {
  ISPH_REQUIRE(ctx && M && M->type == 3 && M->amg && info, "not an AMG preconditioner");
  ISPH_REQUIRE(level >= 0 && level < M->amg->nlev, "level out of range");
  const AmgLevel *L = M->amg->L[(size_t)level];
  info[0] = L->A.n; info[1] = L->A.nnz; info[2] = level < M->amg->nlev - 1 ? L->P.nnz : 0;
  return ISPH_SUCCESS;
}

int isph_prec_amg_export(isph_ctx *ctx, const isph_prec *M, int level, int what, int *rowptr, int *colidx, double *val) {
  ISPH_REQUIRE(ctx && M && M->type == 3 && M->amg && rowptr && colidx && val, "not an AMG preconditioner");
  ISPH_REQUIRE(level >= 0 && level < M->amg->nlev && (what == 0 || (what == 1 && level < M->amg->nlev - 1)), "level out of range");
  const DCsr &Ck = what == 0 ? M->amg->L[(size_t)level]->A : M->amg->L[(size_t)level]->P;
  ISPH_REQUIRE(Ck.nnz < 2147483647LL, "level too large for the 32-bit test export");
  DCsr fine;  // the fine-level copy is dropped after the set-up: rebuilt from the SELL matrix here
  DevTmp<char> tmp;
  if (!Ck.ci.p && Ck.nnz > 0) {
    ISPH_REQUIRE(level == 0 && what == 0, "level operator not resident");
    ISPH_CHECK(amg_csr_from_sell(ctx, M->amg->L[0]->Am->S, fine, tmp));
  }
  struct Drop { DCsr &c; DevBuf<char> &t; ~Drop() { c.release(); t.release(); } } drop{fine, tmp};
  const DCsr &C = fine.ci.p ? fine : Ck;
  std::vector<int> ci((size_t)C.nnz);
  std::vector<double> v((size_t)C.nnz);
  std::vector<long long> rp64((size_t)C.n + 1);
  ISPH_CHECK_HIP(hipMemcpyAsync(rp64.data(), C.rp.p, sizeof(long long) * ((size_t)C.n + 1), hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipMemcpyAsync(ci.data(), C.ci.p, sizeof(int) * (size_t)C.nnz, hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipMemcpyAsync(v.data(), C.v.p, sizeof(double) * (size_t)C.nnz, hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i <= C.n; ++i) rowptr[i] = (int)rp64[(size_t)i];
  std::vector<int> perm;
  for (int i = 0; i < C.n; ++i) {  // coarse rows come out of the SpGEMM in table order: sort for the caller
    const int b = rowptr[i], e = rowptr[i + 1];
    perm.resize((size_t)(e - b));
    std::iota(perm.begin(), perm.end(), 0);
    std::sort(perm.begin(), perm.end(), [&](int a, int c) { return ci[(size_t)(b + a)] < ci[(size_t)(b + c)]; });
    for (int k = b; k < e; ++k) {
      colidx[k] = ci[(size_t)(b + perm[(size_t)(k - b)])];
      val[k] = v[(size_t)(b + perm[(size_t)(k - b)])];
    }
  }
  return ISPH_SUCCESS;
}

int isph_prec_amg_aggregates(isph_ctx *ctx, const isph_prec *M, int level, int *agg) {
  ISPH_REQUIRE(ctx && M && M->type == 3 && M->amg && agg, "not an AMG preconditioner");
  ISPH_REQUIRE(level >= 0 && level < M->amg->nlev - 1, "level out of range");
  const AmgLevel *L = M->amg->L[(size_t)level];
  ISPH_CHECK_HIP(hipMemcpyAsync(agg, L->agg.p, sizeof(int) * (size_t)L->A.n, hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return ISPH_SUCCESS;
}

/* ---- solve ------------------------------------------------------------ */

void isph_solver_params_default(isph_solver_params *p) {
  // SolverLin_Belos::setParameters(NULL), ref: solver_lin_belos.h:226-240
  p->solver_type = 0; p->flexible = 1; p->num_blocks = 50; p->max_iters = 500; p->max_restarts = 15;
  p->tol = 1.0e-8; p->ortho = 0; p->verbose = 0; p->num_recycled = 50;
}

int isph_solve(isph_ctx *ctx, const isph_mat *A, const isph_prec *M, double *b, double *x, int nvec, int lda,
               int is_singular, const int *null_mask, const isph_solver_params *prm_in, isph_solve_info *info,
               int on_device) {
  ISPH_REQUIRE(ctx && A && b && x && info, "NULL argument");
  const int n = A->S.nrow;
  ISPH_REQUIRE(nvec >= 1 && lda >= n, "need nvec >= 1 and lda >= nlocal");
  ISPH_REQUIRE(!M || M->n == n, "preconditioner / matrix size mismatch");
  ISPH_REQUIRE(!M || M->order.get() == A->order.get(),
               "preconditioner and matrix are in different row numberings (build the preconditioner from this matrix)");
  isph_solver_params prm;
  if (prm_in) prm = *prm_in; else isph_solver_params_default(&prm);
  memset(info, 0, sizeof(*info));
  ISPH_CHECK(ensure_scalars(ctx));
  hipStream_t st = ctx->stream;
  // profile mode: events accumulate over set-up and solve until isph_ctx_profile_read collects them; this solve's own
  // SpMV figures (info->spmv_ms) are summed from ev_start on
  if (ctx->ev_used > (1u << 16)) ctx->ev_used = 0;
  const size_t ev_start = ctx->ev_used;
  ctx->stat_reorth = 0;
  ISPH_CHECK_HIP(hipEventRecord(ctx->ev0, st));

  double *db = b, *dx = x;
  // a [lda x nvec] column-major view owns lda (nvec - 1) + n elements (Epetra_MultiVector(View, map, ptr, lda, nvec)): the
  // tail lda - n of the last column is not the caller's to give
  const size_t tot = (size_t)lda * (size_t)(nvec - 1) + (size_t)n;
  // Host operands: a context that has made a host-side matrix ingress owns a pinned ring (ingress.hpp); b, x and the null
  // mask then travel through it -- a streaming copy into a slot, an asynchronous DMA from there, the next operand staged
  // while the previous one is on the link -- instead of as synchronous copies of pageable memory (17 GB/s on this box).
  HostStager *ring = !on_device && ctx->stager && ctx->stager->nslots >= 3 && sizeof(double) * tot <= 12 * HostStager::kChunk ? ctx->stager : nullptr;
  auto ring_slot = [&](int k) { return ring->pslot + (size_t)k * 12 * HostStager::kChunk; };
  if (!on_device) {
    ISPH_CHECK(ctx->bdev.reserve(tot));
    ISPH_CHECK(ctx->xdev.reserve(tot));
    if (ring) {
      stage_copy(ring_slot(0), b, sizeof(double) * tot);
      ISPH_CHECK_HIP(hipMemcpyAsync(ctx->bdev.p, ring_slot(0), sizeof(double) * tot, hipMemcpyHostToDevice, st));
      stage_copy(ring_slot(1), x, sizeof(double) * tot);
      ISPH_CHECK_HIP(hipMemcpyAsync(ctx->xdev.p, ring_slot(1), sizeof(double) * tot, hipMemcpyHostToDevice, st));
    } else {
      ISPH_CHECK_HIP(hipMemcpyAsync(ctx->bdev.p, b, sizeof(double) * tot, hipMemcpyHostToDevice, st));
      ISPH_CHECK_HIP(hipMemcpyAsync(ctx->xdev.p, x, sizeof(double) * tot, hipMemcpyHostToDevice, st));
    }
    db = ctx->bdev.p;
    dx = ctx->xdev.p;
  }
  // b and x are the caller's vectors, in the caller's numbering.  A matrix in the library's own row numbering
  // (order.hpp) solves on copies in its numbering; the results are scattered back before they leave.
  double *const cb = db, *const cx = dx;  // device, caller's numbering
  const RowOrder *ord = A->order.get();
  if (ord && n > 0) {
    ISPH_CHECK(ctx->bint.reserve(tot + 64));
    ISPH_CHECK(ctx->xint.reserve(tot + 64));
    for (int c = 0; c < nvec; ++c) {
      hipLaunchKernelGGL((k_perm_gather<double>), dim3(perm_grid(n)), dim3(kBlock), 0, st, (long long)n, n, 1, (const int *)ord->perm.p,
                         (const double *)cb + (size_t)c * lda, ctx->bint.p + (size_t)c * lda);
      hipLaunchKernelGGL((k_perm_gather<double>), dim3(perm_grid(n)), dim3(kBlock), 0, st, (long long)n, n, 1, (const int *)ord->perm.p,
                         (const double *)cx + (size_t)c * lda, ctx->xint.p + (size_t)c * lda);
    }
    db = ctx->bint.p;
    dx = ctx->xint.p;
  }
  const int sg = stream_grid(n);
  const double *nv = nullptr;
  if (is_singular) {
    // SolverLin::createNullVector, ref: solver_lin.cpp:59-77
    ISPH_CHECK(ctx->nvec.reserve((size_t)(n > 0 ? n : 1)));
    if (null_mask) {
      ISPH_CHECK(ctx->imask.reserve((size_t)(n > 0 ? n : 1)));
      if (ring) {
        stage_copy(ring_slot(2), null_mask, sizeof(int) * (size_t)n);
        ISPH_CHECK_HIP(hipMemcpyAsync(ctx->imask.p, ring_slot(2), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
      } else {
        ISPH_CHECK_HIP(hipMemcpyAsync(ctx->imask.p, null_mask, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
      }
      if (ord && n > 0) {  // the caller's mask follows its rows
        ISPH_CHECK(ctx->imask2.reserve((size_t)n));
        hipLaunchKernelGGL((k_perm_gather<int>), dim3(perm_grid(n)), dim3(kBlock), 0, st, (long long)n, n, 1, (const int *)ord->perm.p,
                           (const int *)ctx->imask.p, ctx->imask2.p);
        hipLaunchKernelGGL(k_mask_to_double, dim3(sg), dim3(kBlock), 0, st, n, ctx->imask2.p, ctx->nvec.p);
      } else {
        hipLaunchKernelGGL(k_mask_to_double, dim3(sg), dim3(kBlock), 0, st, n, ctx->imask.p, ctx->nvec.p);
      }
    } else {
      hipLaunchKernelGGL(k_fill, dim3(sg), dim3(kBlock), 0, st, n, ctx->nvec.p, 1.0);
    }
    ISPH_CHECK(dot_dev(ctx, n, ctx->nvec.p, ctx->nvec.p, nullptr, nullptr, SC_MISC + 6));
    hipLaunchKernelGGL(k_scale_copy, dim3(sg), dim3(kBlock), 0, st, n, ctx->nvec.p, ctx->nvec.p, 1.0,
                       ctx->dscal.p + SC_MISC + 6, 1);
    nv = ctx->nvec.p;
  }
  LinOp op{ctx, A, M, nv, n};
  int iters = 0, restarts = 0, conv = 1;
  double worst_imp = 0.0, worst_exp = 0.0;
  // several right-hand sides of a non-singular system (the Helmholtz solve: one per velocity component) advance
  // together and share their matrix sweeps; the result is the one of solving them one after the other
  const bool lockstep = gmres_lockstep_ok(op, &prm, nvec);
  isph_solve_info cls[kMaxLockstep];
  memset(cls, 0, sizeof(cls));
  if (lockstep) {
    const double *bs[kMaxLockstep];
    double *xs[kMaxLockstep];
    for (int c = 0; c < nvec; ++c) { bs[c] = db + (size_t)c * lda; xs[c] = dx + (size_t)c * lda; }
    ISPH_CHECK(gmres_lockstep(op, nvec, bs, xs, &prm, cls));
  }
  for (int c = 0; c < nvec; ++c) {
    double *bc = db + (size_t)c * lda, *xc = dx + (size_t)c * lda;
    if (nv) ISPH_CHECK(project_dev(ctx, n, nv, bc));  // b -= (b.n) n   (:141-143)
    isph_solve_info ci;
    memset(&ci, 0, sizeof(ci));
    if (lockstep) ci = cls[c];
    else if (prm.solver_type == 1) ISPH_CHECK(pcg(op, bc, xc, &prm, &ci));
    else if (prm.solver_type == 2) ISPH_CHECK(gcrodr(op, bc, xc, &prm, &ci));
    else ISPH_CHECK(gmres(op, bc, xc, &prm, &ci));
    {  // ||b - A x|| / ||b|| with the unprojected A (:201-212)
      ISPH_CHECK(ctx->wv.reserve((size_t)n + 64));
      ISPH_CHECK(spmv_dev(ctx, A, xc, ctx->wv.p, nullptr));
      hipLaunchKernelGGL(k_residual, dim3(sg), dim3(kBlock), 0, st, n, bc, ctx->wv.p);
      ISPH_CHECK(dot_dev(ctx, n, ctx->wv.p, ctx->wv.p, bc, bc, SC_MISC + 16));
    }
    if (nv) ISPH_CHECK(project_dev(ctx, n, nv, xc));  // x -= (x.n) n   (:215-219)
    ISPH_CHECK(fetch_scalars(ctx, SC_MISC + 16, 2));
    const double bn = std::sqrt(ctx->hscal[SC_MISC + 17]);
    ci.rel_res_explicit = std::sqrt(ctx->hscal[SC_MISC + 16]) / (bn == 0.0 ? 1.0 : bn);
    iters += ci.iters;
    restarts += ci.restarts;
    conv = conv && ci.converged;
    worst_imp = std::max(worst_imp, ci.rel_res_implicit);
    worst_exp = std::max(worst_exp, ci.rel_res_explicit);
    if (ctx->rank == 0 && prm.verbose) {
      // non-convergence is reported, never raised (:192-213)
      if (ci.converged) printf(">> isph::Status - Passed! (%d iterations)\n", ci.iters);
      else printf(">> isph::Status - Failed to converge! ||r|| / ||b|| = %6.4e\n", ci.rel_res_explicit);
    }
  }
  if (ord && n > 0) {  // back into the caller's numbering (b holds its projection when the system is singular)
    for (int c = 0; c < nvec; ++c) {
      hipLaunchKernelGGL((k_perm_scatter<double>), dim3(perm_grid(n)), dim3(kBlock), 0, st, n, (const int *)ord->perm.p,
                         (const double *)dx + (size_t)c * lda, cx + (size_t)c * lda);
      hipLaunchKernelGGL((k_perm_scatter<double>), dim3(perm_grid(n)), dim3(kBlock), 0, st, n, (const int *)ord->perm.p,
                         (const double *)db + (size_t)c * lda, cb + (size_t)c * lda);
    }
    db = cb;
    dx = cx;
  }
  if (!on_device && ring) {  // x first (the caller's result), b's copy-out overlaps nothing but is half the pageable time
    ISPH_CHECK_HIP(hipMemcpyAsync(ring_slot(0), dx, sizeof(double) * tot, hipMemcpyDeviceToHost, st));
    ISPH_CHECK_HIP(hipEventRecord(ring->ev[0], st));
    ISPH_CHECK_HIP(hipMemcpyAsync(ring_slot(1), db, sizeof(double) * tot, hipMemcpyDeviceToHost, st));
    ISPH_CHECK_HIP(hipEventSynchronize(ring->ev[0]));
    memcpy(x, ring_slot(0), sizeof(double) * tot);  // while b is on the link
    ISPH_CHECK_HIP(hipStreamSynchronize(st));
    memcpy(b, ring_slot(1), sizeof(double) * tot);
  } else if (!on_device) {
    ISPH_CHECK_HIP(hipMemcpyAsync(b, db, sizeof(double) * tot, hipMemcpyDeviceToHost, st));
    ISPH_CHECK_HIP(hipMemcpyAsync(x, dx, sizeof(double) * tot, hipMemcpyDeviceToHost, st));
  }
  ISPH_CHECK_HIP(hipEventRecord(ctx->ev1, st));
  ISPH_CHECK_HIP(hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  ISPH_CHECK_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  info->converged = conv; info->iters = iters; info->restarts = restarts;
  info->rel_res_implicit = worst_imp; info->rel_res_explicit = worst_exp;
  info->solve_ms = ms;
  info->reorth = ctx->stat_reorth;
  if (ctx->profile) {
    double tot_ms = 0.0;
    int calls = 0;
    for (size_t k = ev_start; k + 1 < ctx->ev_used; k += 2) {
      if (ctx->ev_class[k / 2] != PROF_SPMV) continue;
      float t = 0.f;
      ISPH_CHECK_HIP(hipEventElapsedTime(&t, ctx->ev[k], ctx->ev[k + 1]));
      tot_ms += t;
      ++calls;
    }
    info->spmv_ms = tot_ms;
    info->spmv_calls = calls;
  }
  ISPH_CHECK_HIP(hipGetLastError());
  return prec_health(ctx, M);  // the stream is drained (ev1): every application of this solve has reported
}

// SolverLin_Belos::solveBlockProblem (ref: solver_lin_belos.h:53-128): GMRES / CG over the product vector
// [x_0; ..; x_{dim-1}] with the dim x dim blocked operator and the block-diagonal right preconditioner.
int isph_solve_block(isph_ctx *ctx, int dim, const isph_mat *const *blocks, const isph_prec *M, double *b, double *x,
                     int lda, const isph_solver_params *prm_in, isph_solve_info *info, int on_device) {
  ISPH_REQUIRE(ctx && blocks && b && x && info, "NULL argument");
  ISPH_REQUIRE(dim >= 1 && dim <= 3, "block dimension must be 1, 2 or 3");
  int n = -1;
  for (int k = 0; k < dim * dim; ++k)
    if (blocks[k]) {
      if (n < 0) n = blocks[k]->S.nrow;
      ISPH_REQUIRE(blocks[k]->S.nrow == n, "blocks must have the same number of rows");
    }
  ISPH_REQUIRE(n >= 0, "at least one block is needed");
  for (int k = 0; k < dim; ++k) ISPH_REQUIRE(blocks[k * dim + k] != nullptr, "diagonal blocks must be set");
  ISPH_REQUIRE(lda >= n, "need lda >= nlocal");
  ISPH_REQUIRE(!M || M->n == n, "preconditioner / block size mismatch");
  ISPH_REQUIRE(!M || M->order.get() == blocks[0]->order.get(),
               "preconditioner and blocks are in different row numberings (build the preconditioner from one of the blocks)");
  isph_solver_params prm;
  if (prm_in) prm = *prm_in; else isph_solver_params_default(&prm);
  memset(info, 0, sizeof(*info));
  ISPH_CHECK(ensure_scalars(ctx));
  hipStream_t st = ctx->stream;
  if (ctx->ev_used > (1u << 16)) ctx->ev_used = 0;
  ISPH_CHECK_HIP(hipEventRecord(ctx->ev0, st));
  // product vectors are contiguous [dim][n] on the device
  const size_t nt = (size_t)dim * (size_t)n;
  ISPH_CHECK(ctx->bdev.reserve(nt + 64));
  ISPH_CHECK(ctx->xdev.reserve(nt + 64));
  const hipMemcpyKind in = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  const hipMemcpyKind out = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  // blocks assembled in the library's own row numbering (all of them share it): the components are gathered on the way
  // in and scattered on the way out
  const RowOrder *ord = blocks[0]->order.get();
  for (int k = 0; k < dim * dim; ++k)
    if (blocks[k]) ISPH_REQUIRE(blocks[k]->order.get() == ord, "blocks must share one row numbering");
  DevTmp<double> stg;
  if (ord && n > 0) ISPH_CHECK(stg.reserve(2 * nt));
  for (int k = 0; k < dim; ++k) {
    if (ord && n > 0) {
      const double *sb = b + (size_t)k * lda, *sx = x + (size_t)k * lda;
      if (!on_device) {
        ISPH_CHECK_HIP(hipMemcpyAsync(stg.p + (size_t)k * n, sb, sizeof(double) * (size_t)n, in, st));
        ISPH_CHECK_HIP(hipMemcpyAsync(stg.p + nt + (size_t)k * n, sx, sizeof(double) * (size_t)n, in, st));
        sb = stg.p + (size_t)k * n; sx = stg.p + nt + (size_t)k * n;
      }
      hipLaunchKernelGGL((k_perm_gather<double>), dim3(perm_grid(n)), dim3(kBlock), 0, st, (long long)n, n, 1, (const int *)ord->perm.p,
                         sb, ctx->bdev.p + (size_t)k * n);
      hipLaunchKernelGGL((k_perm_gather<double>), dim3(perm_grid(n)), dim3(kBlock), 0, st, (long long)n, n, 1, (const int *)ord->perm.p,
                         sx, ctx->xdev.p + (size_t)k * n);
      continue;
    }
    ISPH_CHECK_HIP(hipMemcpyAsync(ctx->bdev.p + (size_t)k * n, b + (size_t)k * lda, sizeof(double) * (size_t)n, in, st));
    ISPH_CHECK_HIP(hipMemcpyAsync(ctx->xdev.p + (size_t)k * n, x + (size_t)k * lda, sizeof(double) * (size_t)n, in, st));
  }
  DevTmp<double> tmp, res;
  ISPH_CHECK(tmp.reserve((size_t)n + 64));
  LinOp op{ctx, nullptr, M, nullptr, (int)nt};
  op.dim = dim; op.nloc = n; op.blk = blocks; op.tmp = tmp.p;
  isph_solve_info ci;
  memset(&ci, 0, sizeof(ci));
  int rc = prm.solver_type == 1 ? pcg(op, ctx->bdev.p, ctx->xdev.p, &prm, &ci)
           : prm.solver_type == 2 ? gcrodr(op, ctx->bdev.p, ctx->xdev.p, &prm, &ci)
                                  : gmres(op, ctx->bdev.p, ctx->xdev.p, &prm, &ci);
  if (rc == ISPH_SUCCESS) rc = res.reserve(nt + 64);
  if (rc == ISPH_SUCCESS) rc = op.apply(ctx->xdev.p, res.p);
  if (rc == ISPH_SUCCESS) {
    hipLaunchKernelGGL(k_residual, dim3(stream_grid((long long)nt)), dim3(kBlock), 0, st, (int)nt, (const double *)ctx->bdev.p, res.p);
    rc = dot_dev(ctx, (int)nt, res.p, res.p, ctx->bdev.p, ctx->bdev.p, SC_MISC + 16);
  }
  if (rc == ISPH_SUCCESS) rc = fetch_scalars(ctx, SC_MISC + 16, 2);
  tmp.release(); res.release();
  ISPH_CHECK(rc);
  const double bn = std::sqrt(ctx->hscal[SC_MISC + 17]);
  ci.rel_res_explicit = std::sqrt(ctx->hscal[SC_MISC + 16]) / (bn == 0.0 ? 1.0 : bn);
  for (int k = 0; k < dim; ++k) {
    if (ord && n > 0) {
      double *dst = on_device ? x + (size_t)k * lda : stg.p + (size_t)k * n;
      hipLaunchKernelGGL((k_perm_scatter<double>), dim3(perm_grid(n)), dim3(kBlock), 0, st, n, (const int *)ord->perm.p,
                         (const double *)ctx->xdev.p + (size_t)k * n, dst);
      if (!on_device) ISPH_CHECK_HIP(hipMemcpyAsync(x + (size_t)k * lda, dst, sizeof(double) * (size_t)n, out, st));
      continue;
    }
    ISPH_CHECK_HIP(hipMemcpyAsync(x + (size_t)k * lda, ctx->xdev.p + (size_t)k * n, sizeof(double) * (size_t)n, out, st));
  }
  ISPH_CHECK_HIP(hipEventRecord(ctx->ev1, st));
  ISPH_CHECK_HIP(hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  ISPH_CHECK_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *info = ci;
  info->solve_ms = ms;
  if (ctx->rank == 0 && prm.verbose) {
    if (ci.converged) printf(">> isph::Status - Passed! (%d iterations)\n", ci.iters);
    else printf(">> isph::Status - Failed to converge! ||r|| / ||b|| = %6.4e\n", ci.rel_res_explicit);
  }
  ISPH_CHECK_HIP(hipGetLastError());
  return prec_health(ctx, M);
}

/* ---- assembly --------------------------------------------------------- */

int isph_assemble_poisson(isph_ctx *ctx, const isph_particles *P, int antisym, double dt, const double *rho,
                          const double *vstar, int singular_mode, int is_rank0, int ncol, isph_mat **A_out,
                          double *b_out, int on_device) {
  ISPH_REQUIRE(ctx && P && rho && vstar && A_out && b_out, "NULL argument");
  if (!ctx->ordering || P->nlocal <= 0)
    return assemble_poisson(ctx, P, antisym, dt, rho, vstar, singular_mode, is_rank0, ncol, A_out, b_out, on_device);
  OrderedAssembly W(ctx, P, on_device);
  ISPH_CHECK(W.begin(ncol));
  const double *drho = nullptr, *dvs = nullptr;
  ISPH_CHECK(W.field(rho, 1, true, &drho));
  ISPH_CHECK(W.field(vstar, 3, true, &dvs));
  double *bint = nullptr;
  ISPH_CHECK(W.alloc((size_t)P->nlocal, &bint));
  isph_mat *A = nullptr;
  ISPH_CHECK(assemble_poisson(ctx, &W.Q, antisym, dt, drho, dvs, singular_mode, is_rank0, ncol, &A, bint, 1));
  A->order = W.O;
  const int rc = W.rhs_out(bint, b_out, 1, P->nlocal);
  if (rc != ISPH_SUCCESS) { isph_mat_destroy(A); return rc; }
  *A_out = A;
  return ISPH_SUCCESS;
}

}  // extern "C"

// the three callers of assemble_helmholtz (velocity Helmholtz, solute transport, applied potential) in the library's
// own row numbering: fields permuted, b returned in the caller's numbering
static int assemble_helmholtz_ordered(isph_ctx *ctx, const isph_particles *P, int antisym, double dt, double theta,
                                      const double *nu, const double *rho, const double *pres, const double *force,
                                      const double *g, int incremental, const double *v, int ncol, isph_mat **A_out,
                                      double *b_out, int lda, int on_device, int mode, const double *field, const double *material) {
  if (!ctx->ordering || P->nlocal <= 0)
    return assemble_helmholtz(ctx, P, antisym, dt, theta, nu, rho, pres, force, g, incremental, v, ncol, A_out, b_out, lda,
                              on_device, mode, field, material);
  ISPH_REQUIRE(lda >= P->nlocal, "need lda >= nlocal");
  OrderedAssembly W(ctx, P, on_device);
  ISPH_CHECK(W.begin(ncol));
  const double *dnu = nullptr, *drho = nullptr, *dp = nullptr, *df = nullptr, *dv = nullptr, *dfield = nullptr, *dmat = nullptr;
  ISPH_CHECK(W.field(nu, 1, true, &dnu));
  ISPH_CHECK(W.field(rho, 1, true, &drho));
  ISPH_CHECK(W.field(pres, 1, true, &dp));
  ISPH_CHECK(W.field(force, 3, true, &df));
  ISPH_CHECK(W.field(v, 3, true, &dv));
  ISPH_CHECK(W.field(field, 1, true, &dfield));
  ISPH_CHECK(W.field(material, 1, true, &dmat));
  const int n = P->nlocal, nrhs = mode ? 1 : P->dim;
  double *bint = nullptr;
  ISPH_CHECK(W.alloc((size_t)n * nrhs, &bint));
  isph_mat *A = nullptr;
  ISPH_CHECK(assemble_helmholtz(ctx, &W.Q, antisym, dt, theta, dnu, drho, dp, df, g, incremental, dv, ncol, A_out ? &A : nullptr,
                                bint, n, 1, mode, dfield, dmat));
  if (A) A->order = W.O;
  const int rc = W.rhs_out(bint, b_out, nrhs, lda);
  if (rc != ISPH_SUCCESS) { if (A) isph_mat_destroy(A); return rc; }
  if (A_out) *A_out = A;
  return ISPH_SUCCESS;
}

extern "C" {


int isph_assemble_helmholtz(isph_ctx *ctx, const isph_particles *P, int antisym, double dt, double theta,
                            const double *nu, const double *rho, const double *pres, const double *force,
                            const double *g, int incremental_pressure, const double *v, int ncol, isph_mat **A_out,
                            double *b_out, int lda, int on_device) {
  ISPH_REQUIRE(ctx && P && nu && rho && pres && force && v && b_out, "NULL argument");  // A_out may be NULL: b only
  return assemble_helmholtz_ordered(ctx, P, antisym, dt, theta, nu, rho, pres, force, g, incremental_pressure, v, ncol, A_out,
                                    b_out, lda, on_device, 0, nullptr, nullptr);
}

int isph_assemble_solute_transport(isph_ctx *ctx, const isph_particles *P, int antisym, double dt, double theta,
                                   double dcoeff, const double *conc, int ncol, isph_mat **A_out, double *b_out,
                                   int on_device) {
  ISPH_REQUIRE(ctx && P && conc && A_out && b_out, "NULL argument");
  return assemble_helmholtz_ordered(ctx, P, antisym, dt * dcoeff, theta, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr,
                                    ncol, A_out, b_out, P->nlocal, on_device, 1, conc, nullptr);
}

int isph_assemble_applied_potential(isph_ctx *ctx, const isph_particles *P, int antisym, const double *sigma,
                                    const double *phi, int ncol, isph_mat **A_out, double *b_out, int on_device) {
  ISPH_REQUIRE(ctx && P && phi && A_out && b_out, "NULL argument");
  return assemble_helmholtz_ordered(ctx, P, antisym, -1.0, 0.0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, ncol,
                                    A_out, b_out, P->nlocal, on_device, 2, phi, sigma);
}

int isph_assemble_block_helmholtz(isph_ctx *ctx, const isph_particles *P, int antisym, double dt, double theta,
                                  double beta, const double *nu, const double *rho, const double *pres,
                                  const double *force, const double *g, int incremental_pressure, const double *v,
                                  const double *normal, int ncol, isph_mat **blocks_out, double *b_out, int lda,
                                  int on_device) {
  ISPH_REQUIRE(ctx && P && nu && rho && pres && force && v && blocks_out && b_out, "NULL argument");
  if (!ctx->ordering || P->nlocal <= 0)
    return assemble_block_helmholtz(ctx, P, ncol, antisym, dt, theta, beta, nu, rho, pres, force, g, incremental_pressure, v,
                                    normal, lda, blocks_out, b_out, on_device);
  ISPH_REQUIRE(lda >= P->nlocal, "need lda >= nlocal");
  OrderedAssembly W(ctx, P, on_device);
  ISPH_CHECK(W.begin(ncol));
  const double *dnu = nullptr, *drho = nullptr, *dp = nullptr, *df = nullptr, *dv = nullptr, *dn = nullptr;
  ISPH_CHECK(W.field(nu, 1, true, &dnu));
  ISPH_CHECK(W.field(rho, 1, true, &drho));
  ISPH_CHECK(W.field(pres, 1, true, &dp));
  ISPH_CHECK(W.field(force, 3, true, &df));
  ISPH_CHECK(W.field(v, 3, true, &dv));
  ISPH_CHECK(W.field(normal, 3, true, &dn));
  const int n = P->nlocal, dim = P->dim;
  double *bint = nullptr;
  ISPH_CHECK(W.alloc((size_t)n * dim, &bint));
  isph_mat *blk[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  ISPH_CHECK(assemble_block_helmholtz(ctx, &W.Q, ncol, antisym, dt, theta, beta, dnu, drho, dp, df, g, incremental_pressure, dv, dn,
                                      n, blk, bint, 1));
  for (int q = 0; q < dim * dim; ++q)
    if (blk[q]) blk[q]->order = W.O;
  const int rc = W.rhs_out(bint, b_out, dim, lda);
  if (rc != ISPH_SUCCESS) {
    for (int q = 0; q < dim * dim; ++q) isph_mat_destroy(blk[q]);
    return rc;
  }
  for (int q = 0; q < dim * dim; ++q) blocks_out[q] = blk[q];
  return ISPH_SUCCESS;
}

int isph_compute_volumes(isph_ctx *ctx, const isph_particles *P, double *vfrac_out, int on_device) {
  ISPH_REQUIRE(ctx && P && vfrac_out, "NULL argument");
  return compute_volumes(ctx, P, vfrac_out, on_device);
}

int isph_compute_pnd(isph_ctx *ctx, const isph_particles *P, double *pnd_out, int on_device) {
  ISPH_REQUIRE(ctx && P && pnd_out, "NULL argument");
  return compute_volumes(ctx, P, pnd_out, on_device, /*pnd=*/true);
}

int isph_compute_corrections(isph_ctx *ctx, const isph_particles *P, double *Gc_out, double *Lc_out, int on_device) {
  ISPH_REQUIRE(ctx && P && Gc_out && Lc_out, "NULL argument");
  return compute_corrections(ctx, P, Gc_out, Lc_out, on_device);
}

int isph_gradient(isph_ctx *ctx, const isph_particles *P, int antisym, const double *f, double alpha, int use_filter,
                  int filt_i, int filt_j, double *grad_out, int on_device) {
  ISPH_REQUIRE(ctx && P && f && grad_out, "NULL argument");
  return op_apply(ctx, P, 0, antisym, f, alpha, use_filter, filt_i, filt_j, grad_out, on_device);
}

int isph_divergence(isph_ctx *ctx, const isph_particles *P, int antisym, const double *f, double alpha, int use_filter,
                    int filt_i, int filt_j, double *div_out, int on_device) {
  ISPH_REQUIRE(ctx && P && f && div_out, "NULL argument");
  return op_apply(ctx, P, 1, antisym, f, alpha, use_filter, filt_i, filt_j, div_out, on_device);
}

int isph_correct_velocity_pressure(isph_ctx *ctx, const isph_particles *P, int antisym, double dt, const double *rho,
                                   const double *dp, double *vstar, double *p, int incremental_pressure, int on_device) {
  ISPH_REQUIRE(ctx && P && rho && dp && vstar && p, "NULL argument");
  return correct_velocity_pressure(ctx, P, antisym, dt, rho, dp, vstar, p, incremental_pressure, on_device);
}

int isph_advance_begin(isph_ctx *ctx, const isph_particles *P, int antisym, double dt, const double *p, const double *v,
                       const double *vnp1, double *dp_out, int on_device) {
  ISPH_REQUIRE(ctx && P && p && v && vnp1 && dp_out, "NULL argument");
  return advance_begin(ctx, P, antisym, dt, p, v, vnp1, dp_out, on_device);
}

int isph_advance_end(isph_ctx *ctx, int count, int dim, double dt, const double *dp, const double *vnp1, double *p,
                     double *x, double *v, int on_device) {
  ISPH_REQUIRE(ctx && dp && vnp1 && p && x && v && count >= 0, "bad argument");
  return advance_end(ctx, count, dim, dt, dp, vnp1, p, x, v, on_device);
}

int isph_compute_shift(isph_ctx *ctx, const isph_particles *P, double alpha, double shiftcut, double nonfluidweight,
                       double *dr, int on_device) {
  ISPH_REQUIRE(ctx && P && dr && shiftcut > 0.0, "bad argument");
  return compute_shift(ctx, P, alpha, shiftcut, nonfluidweight, dr, on_device);
}

int isph_apply_shift(isph_ctx *ctx, const isph_particles *P, int antisym, const int *fixed, const double *dr, double *x,
                     double *v, double *p, int on_device) {
  ISPH_REQUIRE(ctx && P && dr && x && v && p, "bad argument");
  return shift_apply(ctx, P, antisym, fixed, dr, 0.0, 1.0, 0.0, 0.0, x, v, p, nullptr, on_device);
}

int isph_shift_particles(isph_ctx *ctx, const isph_particles *P, int antisym, const int *fixed, double shift,
                         double shiftcut, double nonfluidweight, double dt, double *x, double *v, double *p,
                         double *vmax_out, int on_device) {
  ISPH_REQUIRE(ctx && P && x && v && p && shiftcut > 0.0, "bad argument");
  return shift_apply(ctx, P, antisym, fixed, nullptr, shift, shiftcut, nonfluidweight, dt, x, v, p, vmax_out, on_device);
}

}  // extern "C"
