// core.hpp -- context / matrix / preconditioner objects behind the C ABI.
#pragma once
#include "common.hpp"
#include "sell.hpp"

namespace isph { struct HostStager; }  // ingress.hpp: pinned ring of the host CSR ingress

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
  // reduction scratch + small scalar mailboxes
  isph::DevBuf<double> partial;   // per-block partials
  isph::DevBuf<double> dscal;     // device scalars (dot results, Hessenberg column, ...)
  double *hscal = nullptr;        // pinned host mirror
  size_t hscal_cap = 0;
  // Krylov workspaces (grow-only, reused across solves)
  isph::DevBuf<double> V, Z, wv, tv, rv, pv, nvec, xext, sendbuf, bdev, xdev;
  isph::DevBuf<int> imask;
  // profiling events
  std::vector<hipEvent_t> ev;
  size_t ev_used = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t ev_fetch = nullptr;  // marks the scalar mailbox copy of a Krylov iteration (host waits on it only)
  std::vector<hipEvent_t> ev_ls;  // one such mark per right-hand side of a lockstep solve (solver.hpp gmres_lockstep)
  // halo exchange overlapped with the interior rows of the SpMV: the grouped send/recv runs on comm_stream between
  // ev_pack (send buffer packed on `stream`) and ev_halo (ghost values landed in xghost)
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_pack = nullptr, ev_halo = nullptr;
  isph::DevBuf<double> xghost;
  isph::HostStager *stager = nullptr;  // created by the first host-side isph_mat_create_csr
};

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
  bool local = false;  // rectangular operator on rank-local vectors (AMG transfer operators): no ghost columns
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
};
