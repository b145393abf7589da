// core.hpp -- context / matrix / preconditioner objects behind the C ABI.
#pragma once
#include "common.hpp"
#include "order.hpp"
#include "sell.hpp"

namespace isph { struct HostStager; }  // ingress.hpp: pinned ring of the host CSR ingress

// The layout of a neighbour list the row kernels read (slice-transposed, optionally ordered by matrix column:
// assemble.hpp build_neigh_ell), kept between operator calls while the caller holds the list (isph_ctx_hold_neighbours).
struct isph_neigh_layout {
  const void *nptr = nullptr;
  const int *nidx = nullptr, *colmap = nullptr;
  const void *order = nullptr;  // the RowOrder the layout was built in (NULL: the caller's numbering)
  int n = -1, is_sorted = 0;
  isph::DevBuf<long long> off;
  isph::DevBuf<int> idx, len, sorted;
  void release() { off.release(); idx.release(); len.release(); sorted.release(); nptr = nullptr; nidx = colmap = nullptr; order = nullptr; n = -1; is_sorted = 0; }
};

// What an ordered assembly (isph_capi.hip: OrderedAssembly) tells the neighbour-layout builder of assemble.hpp for the
// duration of one call: the list rows are read through rowsrc, the entries through idmap (order.hpp).
struct isph_neigh_map {
  const int *rowsrc = nullptr, *idmap = nullptr, *colmap_key = nullptr;
  const int *colkey = nullptr;  // internal matrix column of the CALLER's particle j (sort key of its list entries)
  const void *order = nullptr;
};

// The per-type-pair tables of the row kernels (kind, h, cutsq and what is derived from them) as they were staged last:
// a time step calls half a dozen operators with the same tables, each staging used to cost six small copies and a
// host synchronisation.
struct isph_table_cache {
  int ntypes = -1, kernel = -1, dim = -1;
  std::vector<int> kind;
  std::vector<double> h, cutsq;
  isph::DevBuf<int> dkind;
  isph::DevBuf<double> dh, dcutsq, dhinv, dknorm, dkdnorm;
  void release() { dkind.release(); dh.release(); dcutsq.release(); dhinv.release(); dknorm.release(); dkdnorm.release(); ntypes = -1; }
};

struct isph_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int rank = 0, nranks = 1;
  ncclComm_t comm = nullptr;
  // host-staged transport (isph_ctx_create_hostcomm; comm.hpp): callbacks of the caller + pinned staging buffers
  isph_host_transport host_tr = {nullptr, nullptr, nullptr};
  double *hsend = nullptr, *hrecv = nullptr;
  size_t hsend_cap = 0, hrecv_cap = 0;
  bool profile = false;
  int stat_reorth = 0;  // Gram-Schmidt steps of the current solve whose second (DGKS) pass was applied
  // reduction scratch + small scalar mailboxes
  isph::DevBuf<double> partial;   // per-block partials
  isph::DevBuf<double> dscal;     // device scalars (dot results, Hessenberg column, ...)
  double *hscal = nullptr;        // pinned host mirror
  size_t hscal_cap = 0;
  // Krylov workspaces (grow-only, reused across solves)
  isph::DevBuf<double> V, Z, wv, tv, rv, pv, nvec, xext, sendbuf, bdev, xdev;
  isph::DevBuf<double> bint, xint;  // b / x of a solve in the matrix' own row numbering (order.hpp)
  isph::DevBuf<int> imask, imask2;
  // profiling events
  std::vector<hipEvent_t> ev;     // pairs (begin, end) of profile mode
  std::vector<int> ev_class;      // class of every pair (isph::ProfClass)
  size_t ev_used = 0;
  std::vector<hipEvent_t> hev;    // profile mode, products with a halo: triples (packed, ghosts landed, interior done)
  size_t hev_used = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t ev_fetch = nullptr;  // marks the scalar mailbox copy of a Krylov iteration (host waits on it only)
  std::vector<hipEvent_t> ev_ls;  // one such mark per right-hand side of a lockstep solve (solver.hpp gmres_lockstep)
  // halo exchange overlapped with the interior rows of the SpMV: the grouped send/recv runs on comm_stream between
  // ev_pack (send buffer packed on `stream`) and ev_halo (ghost values landed in xghost)
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_pack = nullptr, ev_halo = nullptr;
  isph::DevBuf<double> xghost;
  isph::HostStager *stager = nullptr;  // created by the first host-side isph_mat_create_csr
  bool neigh_hold = false;             // isph_ctx_hold_neighbours
  isph_neigh_layout neigh_cache[2];    // [0] list order, [1] ordered by matrix column
  isph_table_cache tables;             // assemble.hpp stage_tables
  // row numbering of the matrices the assembly entry points build (isph_ctx_set_ordering): 1 = the library's bricks
  // (order.hpp, default), 0 = the caller's atom order
  int ordering = 1;
  isph::OrderBox box;                  // isph_ctx_set_periodic_box: what the caller said about its periodic box (for the brick sort)
  isph_neigh_map nmap;                 // set around one ordered assembly call
  // while the neighbour list is held the row order of the first matrix assembly serves the following ones (LAMMPS'
  // atom order -- the reference's row map -- does not change between two re-neighbourings either)
  isph::RowOrderPtr held_order;
  const void *held_key[2] = {nullptr, nullptr};  // neigh_idx / neigh_ptr of the caller
};

namespace isph {
// profile mode (isph_ctx_set_profile): HIP events on the library's stream around the launches of one class of kernels;
// isph_ctx_profile_read sums them per class.  Recorded on ctx->stream only (side-stream launches are not bracketed).
enum ProfClass { PROF_SPMV = 0, PROF_PREC_APPLY, PROF_MULTI_DOT, PROF_MULTI_AXPY_DOT, PROF_MULTI_AXPY_NORM, PROF_ILU_EXTRACT,
                 PROF_ILU_SCHEDULE, PROF_ILU_FACTOR, PROF_NCLASS };
struct ProfScope {
  isph_ctx *c;
  size_t slot = (size_t)-1;
  ProfScope(isph_ctx *ctx, int cls, hipStream_t st = nullptr) : c(ctx) {
    if (!c || !c->profile || (st && st != c->stream)) return;
    if (c->ev_used + 2 > c->ev.size()) {
      hipEvent_t a = nullptr, b = nullptr;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      c->ev.push_back(a); c->ev.push_back(b);
      c->ev_class.push_back(cls);
    }
    slot = c->ev_used;
    c->ev_class[slot / 2] = cls;
    c->ev_used += 2;
    (void)hipEventRecord(c->ev[slot], c->stream);
  }
  ~ProfScope() { end(); }
  void end() {
    if (slot != (size_t)-1) (void)hipEventRecord(c->ev[slot + 1], c->stream);
    slot = (size_t)-1;
  }
};
}  // namespace isph

struct isph_halo {
  int npeers = 0;
  std::vector<int> peer, send_ptr, recv_ptr;
  isph::DevBuf<int> send_idx;
  int nsend = 0, nrecv = 0;
  // slices without / with ghost columns (k_sell_flag_ghost_slices), ascending
  isph::DevBuf<int> list_int, list_bnd;
  int n_int = 0, n_bnd = 0;
};

struct isph_mat {
  isph::Sell S;
  isph_halo halo;
  // the library's own row numbering (order.hpp) when the matrix was assembled in it: rows and owned columns are
  // internal, vectors cross the C ABI in the caller's numbering (gathered / scattered there)
  isph::RowOrderPtr order;
  bool local = false;  // rectangular operator on rank-local vectors (AMG transfer operators): no ghost columns
  bool aux = false;    // a level operator of the AMG hierarchy with its own halo plan: not in the caller's SpMV statistics
};

struct isph_ilu;  // ilu.hpp
struct isph_amg;  // amg.hpp
struct isph_schwarz;  // schwarz.hpp
struct isph_overlap;  // isph_capi.hip: overlap-1 Schwarz across ranks

struct isph_prec {
  int type = 0;  // 0 none, 1 jacobi, 2 bjacobi-ilu<k> (block stream), 3 sa-amg, 4 additive Schwarz ILU(k) (schwarz.hpp),
                 // 5 ILU(k) of the rank's rows + one layer of the neighbours' rows (isph_prec_create_overlap)
  int n = 0;
  isph::DevBuf<double> invdiag;
  isph_ilu *ilu = nullptr;
  isph_amg *amg = nullptr;
  isph_schwarz *schwarz = nullptr;
  isph_overlap *ovl = nullptr;
  isph::RowOrderPtr order;  // the numbering of the matrix it was built from (isph_prec_apply takes the caller's)
};
