// solver_lin_hip.h -- SolverLin_HIP: the drop-in for SolverLin_Belos
// (ref: solver_lin_belos.h:34-50,130-264).  `typedef SolverLin_HIP
// SolverLin_Belos` keeps `typedef class SolverLin_Belos SolverLinear`
// (pair_isph.h:77) and the USER-REAXC-T call site compiling unchanged.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "halo_lists.h"
#include "isph_hip.h"
#include "mpi_transport.h"
#include "solver_lin.h"

namespace LAMMPS_NS {

class SolverLin_HIP : public SolverLin {
 public:
  SolverLin_HIP(MPI_Comm &comm, int device = 0) : SolverLin(comm), _ctx(nullptr), _device(device) {}
  virtual ~SolverLin_HIP() {
    if (_ctx) isph_ctx_destroy(_ctx);
  }

  // same keys/defaults as SolverLin_Belos::setParameters, ref: solver_lin_belos.h:224-264
  void setParameters(Teuchos::ParameterList *param = NULL) {
    if (param == NULL) {
      _param = Teuchos::rcp(new Teuchos::ParameterList);
      _param->set("Flexible Gmres", true);
      _param->set("Num Blocks", 50);
      _param->set("Block Size", 1);
      _param->set("Maximum Iterations", 500);
      _param->set("Maximum Restarts", 15);
      _param->set("Convergence Tolerance", 1.0e-8);
      _param->set("Orthogonalization", "DGKS");
      _param->set("Solver Type", "Block GMRES");
      _param->set("Num Recycled Blocks", 50);
      _param->set("Output Frequency", 5);
      _param->set("Output Style", 1);
      _param->set("Verbosity", 33);
    } else if (_param.get() != param) {
      _param = Teuchos::rcp(param, false);
    }
  }

  // Belos-style keys -> isph_solver_params
  isph_solver_params solverParams() {
    isph_solver_params p;
    isph_solver_params_default(&p);
    const std::string type = _param->get("Solver Type", "Block GMRES");
    if (type == "Block CG") p.solver_type = 1;
    else if (type == "Recycling GMRES") p.solver_type = 2;  // Belos::GCRODRSolMgr, solver_lin_belos.h:178-179
    else if (type != "Block GMRES" && _comm.MyPID() == 0)
      std::printf(">> SolverLin_HIP: Solver Type '%s' not available, using Block GMRES\n", type.c_str());
    p.num_recycled = _param->get("Num Recycled Blocks", 50);
    p.flexible = _param->get("Flexible Gmres", true) ? 1 : 0;
    p.num_blocks = _param->get("Num Blocks", 50);
    p.max_iters = _param->get("Maximum Iterations", 500);
    p.max_restarts = _param->get("Maximum Restarts", 15);
    p.tol = _param->get("Convergence Tolerance", 1.0e-8);
    const std::string ortho = _param->get("Orthogonalization", "DGKS");
    p.ortho = ortho == "ICGS" ? 1 : ortho == "IMGS" ? 2 : 0;
    p.verbose = 0;
    return p;
  }

  int solveProblem(PrecondWrapper *prec = NULL, const char *name = NULL) {
    if (_comm.MyPID() == 0 && name != NULL) std::cout << ">> Belos::Label - " << name << std::endl;
    setParameters(_param.get());
    if (ensureContext() != ISPH_SUCCESS) return LAMMPS_FAILURE;
    if (!_A || !_x || !_b) return LAMMPS_FAILURE;

    // matrix ingress: the three CRS arrays of the filled Epetra matrix
    int *rp = nullptr, *ci = nullptr;
    double *v = nullptr;
    _A->ExtractCrsDataPointers(rp, ci, v);
    isph_mat *A = nullptr;
    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    // a preconditioner that is local to 512-row subdomains is set up while the matrix is still on the link
    const int fused = prec != NULL ? prec->fusedIngressBlockRows() : 0;
    isph_prec *Mfused = nullptr;
    int nsub = 0;
    const int *subptr = nullptr;
    const bool table = fused > 0 && prec->fusedIngressSubdomains(nsub, subptr);
    // a wrapper that was given the coordinates of the rows (setCoordinates): the library numbers the rows itself
    int cdim = 0;
    const double *cx = nullptr, *cy = nullptr, *cz = nullptr;
    const bool ordered = prec != NULL && prec->ingressCoordinates(cdim, cx, cy, cz);
    const char *fuse_env = std::getenv("ISPH_DROPIN_FUSED_ORDER");   // "1": the fused form of the ordered ingress (see precond_ifpack.h)
    const bool fuse_ordered = ordered && fuse_env && fuse_env[0] == '1' && _A->NumMyRows() > 0 && prec->orderedIngressFusable();
    if ((fuse_ordered
             ? isph_mat_create_csr_coords_bjacobi(_ctx, _A->NumMyRows(), _A->NumMyCols(), rp, ci, v, cdim, cx, cy, cz, &A, &Mfused)
         : ordered ? isph_mat_create_csr_coords(_ctx, _A->NumMyRows(), _A->NumMyCols(), rp, ci, v, cdim, cx, cy, cz, &A)
         : table ? isph_mat_create_csr_blocks(_ctx, _A->NumMyRows(), _A->NumMyCols(), rp, ci, v, nsub, subptr, &A, &Mfused)
         : fused > 0 ? isph_mat_create_csr_bjacobi(_ctx, _A->NumMyRows(), _A->NumMyCols(), rp, ci, v, fused, &A, &Mfused)
                     : isph_mat_create_csr(_ctx, _A->NumMyRows(), _A->NumMyCols(), rp, ci, v, 0, &A)) != ISPH_SUCCESS)
      return report_failure();
    if (attachHalo(A, *_A) != ISPH_SUCCESS) { isph_mat_destroy(A); isph_prec_destroy(Mfused); return report_failure(); }
    const std::chrono::steady_clock::time_point t1 = std::chrono::steady_clock::now();  // the ingress returns synchronised

    int rc = ISPH_SUCCESS;
    // the norm of the null vector is all-reduced here, where every rank is (8 bytes); getNullVector() then fills the
    // host copy locally when somebody asks -- a getter that only rank 0 calls must not be a collective
    if (_is_singular && !(prec != NULL && prec->usesNullVector())) { _n_sumsq = globalMaskSumSq(); _n_stale = true; }
    if (prec != NULL) {
      if (_is_singular && prec->usesNullVector()) { createNullVector(); prec->setNullVector(_n->Values()); }  // :149-151
      prec->create();
      if (fused > 0 || fuse_ordered) prec->adoptDevice(Mfused);
      else rc = prec->createOnDevice(_ctx, A);  // Ifpack Initialize+Compute happen here (:153)
    }
    if (_timing && rc == ISPH_SUCCESS) rc = isph_ctx_sync(_ctx);  // only to attribute the set-up; the solve queues behind it anyway
    const std::chrono::steady_clock::time_point t2 = std::chrono::steady_clock::now();
    isph_solver_params p = solverParams();

    isph_solve_info info;
    if (rc == ISPH_SUCCESS)
      rc = isph_solve(_ctx, A, prec ? prec->_M : nullptr, _b->Values(), _x->Values(), _x->NumVectors(), _x->Stride(),
                      _is_singular ? 1 : 0, _null_mask ? _null_mask->Values() : nullptr, &p, &info, 0);
    const std::chrono::steady_clock::time_point t3 = std::chrono::steady_clock::now();
    if (prec != NULL) {
      if (_is_singular && prec->usesNullVector()) prec->setNullVector(NULL);
      prec->free();  // :186-191
    }
    isph_mat_destroy(A);
    const std::chrono::steady_clock::time_point t4 = std::chrono::steady_clock::now();
    _ms[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
    _ms[1] = std::chrono::duration<double, std::milli>(t2 - t1).count();
    _ms[2] = std::chrono::duration<double, std::milli>(t3 - t2).count();
    _ms[3] = std::chrono::duration<double, std::milli>(t4 - t3).count();
    if (rc != ISPH_SUCCESS) return report_failure();
    _last = info;
    if (_comm.MyPID() == 0) {  // :194-213: non-convergence is reported, never raised
      if (info.converged) std::cout << ">> Belos::Status - Passed! " << (name == NULL ? " " : name) << std::endl;
      else {
        std::cout << ">> Belos::Status - Failed to converge! " << (name == NULL ? " " : name) << std::endl;
        std::printf(">> Belos:: ||r|| / ||b|| = %6.4e\n", info.rel_res_explicit);
      }
    }
    return LAMMPS_SUCCESS;
  }

  // ref: solver_lin_belos.h:53-128
  int solveBlockProblem(PrecondWrapper *prec = NULL, const char *name = NULL) {
    if (_comm.MyPID() == 0 && name != NULL) std::cout << ">> Belos(Block)::Label - " << name << std::endl;
    if (!_x || !_b || _b->NumVectors() != _dim) {
      std::fprintf(stderr, ">> SolverLin_Belos::solveBlockProblem, dimension of rhs does not match to the block matrix\n");
      return LAMMPS_FAILURE;
    }
    if (_is_singular) {
      std::fprintf(stderr, ">> SolverLin_Belos::solveBlockProblem does not support singular problems\n");
      return LAMMPS_FAILURE;
    }
    setParameters(_param.get());
    if (ensureContext() != ISPH_SUCCESS) return LAMMPS_FAILURE;
    isph_mat *blk[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const isph_mat *cblk[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int rc = ISPH_SUCCESS;
    for (int i = 0; i < _dim && rc == ISPH_SUCCESS; ++i)
      for (int j = 0; j < _dim && rc == ISPH_SUCCESS; ++j) {
        const Epetra_CrsMatrix *B = _blk[i * 3 + j];
        if (!B) continue;
        int *rp = nullptr, *ci = nullptr;
        double *v = nullptr;
        B->ExtractCrsDataPointers(rp, ci, v);
        rc = isph_mat_create_csr(_ctx, B->NumMyRows(), B->NumMyCols(), rp, ci, v, 0, &blk[i * _dim + j]);
        if (rc == ISPH_SUCCESS) rc = attachHalo(blk[i * _dim + j], *B);
        cblk[i * _dim + j] = blk[i * _dim + j];
      }
    // diagonal preconditioner: one operator, built from the matrix the wrapper was given (precond_ml.h:138-155:
    // "assume that all diagonals are same now"), or from block (0,0) when none was set
    isph_mat *Aprec = nullptr;
    if (rc == ISPH_SUCCESS && prec != NULL) {
      prec->create(_dim);
      const Epetra_CrsMatrix *P = prec->_A.get() ? prec->_A.get() : _blk[0];
      if (!P) throw std::runtime_error("SolverLin_Belos::solveProblem getBlockPrecondOperator failed");
      int *rp = nullptr, *ci = nullptr;
      double *v = nullptr;
      P->ExtractCrsDataPointers(rp, ci, v);
      rc = isph_mat_create_csr(_ctx, P->NumMyRows(), P->NumMyCols(), rp, ci, v, 0, &Aprec);
      if (rc == ISPH_SUCCESS) rc = attachHalo(Aprec, *P);
      if (rc == ISPH_SUCCESS) rc = prec->createOnDevice(_ctx, Aprec);
      if (rc == ISPH_SUCCESS && !prec->_M) {
        isph_mat_destroy(Aprec);
        for (int k = 0; k < 9; ++k) isph_mat_destroy(blk[k]);
        throw std::runtime_error("SolverLin_Belos::solveProblem getBlockPrecondOperator failed");
      }
    }
    isph_solver_params p = solverParams();
    isph_solve_info info;
    if (rc == ISPH_SUCCESS)
      rc = isph_solve_block(_ctx, _dim, cblk, prec ? prec->_M : nullptr, _b->Values(), _x->Values(), _x->Stride(), &p,
                            &info, 0);
    if (prec != NULL) prec->free();
    isph_mat_destroy(Aprec);
    for (int k = 0; k < 9; ++k) isph_mat_destroy(blk[k]);
    if (rc != ISPH_SUCCESS) return report_failure();
    _last = info;
    if (_comm.MyPID() == 0) {
      if (info.converged) std::cout << ">> Belos::Status - Passed! " << (name == NULL ? " " : name) << std::endl;
      else {
        std::cout << ">> Belos::Status - Failed to converge! " << (name == NULL ? " " : name) << std::endl;
        std::printf(">> Belos:: ||r|| / ||b|| = %6.4e\n", info.rel_res_explicit);
      }
    }
    return LAMMPS_SUCCESS;
  }

  const isph_solve_info &lastSolveInfo() const { return _last; }
  // wall time of the last solveProblem in milliseconds: [0] matrix ingress (host CSR -> device), [1] preconditioner
  // set-up, [2] Krylov solve incl. the transfers of b and x, [3] release.  [1] and [2] are only separated when
  // setTiming(true) put a synchronisation between them (not a reference method; diagnostics of the drop-in path)
  const double *lastTimingsMs() const { return _ms; }
  int lastIngressInfo(double info[8]) const { return _ctx ? isph_ingress_info(_ctx, info) : ISPH_FAILURE; }
  void setTiming(bool on) { _timing = on; }

 private:
  // One context per solver object.  More than one rank (SolverLin(MPI_Comm&) is multi-rank by construction,
  // solver_lin.cpp:30-31; row map = the rank's atoms, pair_isph.cpp:1258-1259), or a matrix that carries ghost
  // columns, needs the RCCL communicator: rank 0 draws the unique id and broadcasts it over the caller's MPI
  // communicator.  One GPU per rank: `device` = the rank's local device index (constructor argument).
  bool needComm() const {
    if (_comm.NumProc() > 1) return true;
    if (_A.get() && _A->Importer() != NULL) return true;
    for (int k = 0; k < 9; ++k)
      if (_blk[k] && _blk[k]->Importer() != NULL) return true;
    return false;
  }
  int ensureContext() {
    if (_ctx) return ISPH_SUCCESS;
    if (!needComm()) return isph_ctx_create(_device, nullptr, &_ctx);
#ifdef ISPH_HAVE_MPI
    // Which transport: RCCL needs one device per rank.  Ranks that share a device (several MPI ranks of a LAMMPS run
    // per GPU) exchange through the caller's MPI communicator instead, staged through pinned memory (csrc/comm.hpp).
    // "auto": every rank gathers the PCI bus ids of the devices of the ranks on its node; one shared device anywhere => MPI
    // everywhere (the choice must be the same on all ranks).  ISPH_TRANSPORT=rccl|mpi overrides.
    if (useMpiTransport()) {
      _mpi.comm = _comm.Comm();
      const isph_host_transport t = _mpi.table();
      return isph_ctx_create_hostcomm(_device, nullptr, _comm.MyPID(), _comm.NumProc(), &t, &_ctx);
    }
#endif
    // rank 0 draws the RCCL id; its status travels with the id so that every rank fails together instead of waiting in
    // a broadcast (or in ncclCommInitRank) for a rank that has already returned
    struct { int status; char uid[ISPH_UID_BYTES]; } msg;
    std::memset(&msg, 0, sizeof(msg));
    if (_comm.MyPID() == 0) msg.status = isph_comm_unique_id(msg.uid);
#ifdef ISPH_HAVE_MPI
    MPI_Bcast(&msg, (int)sizeof(msg), MPI_BYTE, 0, _comm.Comm());
#else
    if (_comm.NumProc() > 1) {  // cannot happen with the stand-in communicator (one rank); kept for a real Epetra_MpiComm
      std::fprintf(stderr, ">> SolverLin_HIP: %d ranks need a build with -DISPH_HAVE_MPI (the RCCL id is broadcast over MPI)\n", _comm.NumProc());
      return ISPH_FAILURE;
    }
#endif
    if (msg.status != ISPH_SUCCESS) return ISPH_FAILURE;
    return isph_ctx_create_dist(_device, nullptr, _comm.MyPID(), _comm.NumProc(), msg.uid, &_ctx);
  }
#ifdef ISPH_HAVE_MPI
  bool useMpiTransport() {
    const char *e = std::getenv("ISPH_TRANSPORT");
    int choice = -1;  // -1 auto, 0 rccl, 1 mpi
    if (e && std::string(e) == "mpi") choice = 1;
    else if (e && std::string(e) == "rccl") choice = 0;
    if (choice < 0) {
      MPI_Comm node;
      int shared = 0;
      if (MPI_Comm_split_type(_comm.Comm(), MPI_COMM_TYPE_SHARED, 0, MPI_INFO_NULL, &node) == MPI_SUCCESS) {
        int nn = 1, me = 0;
        MPI_Comm_size(node, &nn);
        MPI_Comm_rank(node, &me);
        // the PHYSICAL device, not the ordinal: in the usual one-GPU-per-rank launch every rank runs under its own
        // ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES mask and sees "device 0" -- ordinals would call that sharing and
        // put every exchange on the host; the PCI bus id of the device the ordinal maps to tells them apart
        char mine[ISPH_DEVICE_ID_BYTES];
        std::memset(mine, 0, sizeof(mine));
        if (isph_device_identity(_device, mine) != ISPH_SUCCESS) std::snprintf(mine, sizeof(mine), "ordinal-%d", _device);
        std::vector<char> ids((size_t)nn * ISPH_DEVICE_ID_BYTES, 0);
        MPI_Allgather(mine, ISPH_DEVICE_ID_BYTES, MPI_CHAR, ids.data(), ISPH_DEVICE_ID_BYTES, MPI_CHAR, node);
        for (int r = 0; r < nn; ++r)
          shared = shared || (r != me && std::memcmp(ids.data() + (size_t)r * ISPH_DEVICE_ID_BYTES, mine, ISPH_DEVICE_ID_BYTES) == 0);
        MPI_Comm_free(&node);
      }
      choice = _comm.MaxAll(shared);
      if (choice && _comm.MyPID() == 0)
        std::printf(">> SolverLin_HIP: ranks share a device -- halo exchange and dot products go through MPI (host-staged), not RCCL\n");
    }
    return choice == 1;
  }
  MpiTransport _mpi;
#endif
  // Epetra_Import of the matrix -> isph_mat_set_halo: peers = ProcsTo U ProcsFrom (ascending), per peer the owned rows
  // to send (ExportLIDs, grouped by destination) and the number of ghost values to receive (ghost columns are stored
  // grouped by source rank, Epetra's column-map order).
  int attachHalo(isph_mat *A, const Epetra_CrsMatrix &E) {
    if (E.NumMyCols() == E.NumMyRows()) return ISPH_SUCCESS;
    HaloLists H;
    if (halo_lists_from_import(E, H) != ISPH_SUCCESS) return ISPH_FAILURE;
    return isph_mat_set_halo(_ctx, A, H.npeers(), H.peers.data(), H.send_ptr.data(), H.send_idx.data(), H.recv_ptr.data());
  }
  int report_failure() {
    if (_comm.MyPID() == 0) std::fprintf(stderr, ">> SolverLin_HIP: %s\n", isph_last_error());
    return LAMMPS_FAILURE;
  }
  isph_ctx *_ctx;
  int _device;
  isph_solve_info _last{};
  double _ms[4] = {0.0, 0.0, 0.0, 0.0};
  bool _timing = false;
};

typedef SolverLin_HIP SolverLin_Belos;  // pair_isph.h:77 keeps compiling

}  // namespace LAMMPS_NS
