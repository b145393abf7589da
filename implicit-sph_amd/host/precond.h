// precond.h -- PrecondWrapper: header-compatible with the reference
// (ref: precond.h:17-46) so PairISPH::initializeSolvers compiles unchanged,
// forwarding to libisph_hip through the C ABI (include/isph_hip.h).
#pragma once
#include <stdexcept>

#include "isph_compat.h"
#include "isph_hip.h"

namespace LAMMPS_NS {

class SolverLin_HIP;

class PrecondWrapper {
 protected:
  Epetra_MpiComm _comm;
  Teuchos::RCP<Teuchos::ParameterList> _param;
  Teuchos::RCP<Epetra_CrsMatrix> _A;
  // device side (owned): created by create(), released by free()
  isph_prec *_M = nullptr;
  friend class SolverLin_HIP;
  // hook used by SolverLin_HIP: build the device preconditioner for matrix A
  virtual int createOnDevice(isph_ctx *, const isph_mat *) { return ISPH_SUCCESS; }
  // > 0: this wrapper's device object is the block-Jacobi ILU(0) on subdomains of that many rows, which the host matrix
  // ingress can set up while the matrix is still crossing PCIe (isph_mat_create_csr_bjacobi); SolverLin_HIP then hands
  // the finished object over with adoptDevice() instead of calling createOnDevice()
  virtual int fusedIngressBlockRows() { return 0; }
  // the wrapper's own table of subdomains for that fused set-up (isph_mat_create_csr_blocks), if it has one
  virtual bool fusedIngressSubdomains(int &, const int *&) { return false; }
  void adoptDevice(isph_prec *M) { free(); _M = M; }
  // Coordinates of the rows, as PrecondWrapper_ML::setCoordinates receives them (ref: precond_ml.h:63-94; borrowed
  // pointers into the adapter's Epetra_MultiVector, pair_isph.cpp:1290-1303).  When a wrapper has them, SolverLin_HIP
  // brings the host matrix in through isph_mat_create_csr_coords: the library then numbers the rows itself (bricks of
  // about 500 particles) instead of keeping LAMMPS' atom order.
  int _cdim = 0;
  const double *_cx = nullptr, *_cy = nullptr, *_cz = nullptr;
  void storeCoordinates(int dim, const double *x, const double *y, const double *z) {
    const bool ok = x != NULL && y != NULL && (dim == 2 || (dim == 3 && z != NULL));
    _cdim = ok ? dim : 0; _cx = ok ? x : nullptr; _cy = ok ? y : nullptr; _cz = ok ? z : nullptr;
  }
  // true when this wrapper's device object is the block ILU(0) on the library's bricks (fill 0), which
  // isph_mat_create_csr_coords_bjacobi can set up during the ordered ingress
  virtual bool orderedIngressFusable() { return false; }
  virtual bool ingressCoordinates(int &dim, const double *&x, const double *&y, const double *&z) {
    if (_cx == nullptr) return false;
    dim = _cdim; x = _cx; y = _cy; z = _cz;
    return true;
  }

 public:
  PrecondWrapper(MPI_Comm comm) : _comm(comm) {}
  virtual ~PrecondWrapper() { free(); }

  virtual void setMatrix(Epetra_CrsMatrix *A) {
    if (A != NULL) _A = Teuchos::rcp(A, false);
  }
  // ref: precond.h:33-36.  The blocked operator is assembled from the scalar blocks SolverLin::setBlock was given, so
  // the Thyra object is accepted for interface compatibility and not used.
  virtual void setBlockMatrix(Thyra::PhysicallyBlockedLinearOpBase<double> *) { return; }
  virtual Teuchos::ParameterList *setParameters(Teuchos::ParameterList *param = NULL) { return _param.get(); }
  virtual void setNullVector(double *) { return; }  // base no-op, ref: precond.h:40
  // true when setNullVector() is more than the base no-op: SolverLin_HIP then forms the host copy of the null vector
  // before create() like the reference (solver_lin_belos.h:149-151); for the others the 8 n bytes are not touched per
  // solve (the device forms its own null vector from the mask) and getNullVector() builds the host copy on demand
  virtual bool usesNullVector() const { return false; }
  // The reference builds the Ifpack/ML object here; the device object needs the
  // device matrix, which SolverLin_HIP owns, so create() only records the request
  // and the build happens inside solveProblem (same place in the timeline:
  // solver_lin_belos.h:147-156), via createOnDevice().
  virtual void create() { return; }
  virtual void create(const int) { return; }
  virtual void free() {
    if (_M) { isph_prec_destroy(_M); _M = nullptr; }
  }
  virtual Epetra_Operator *getPrecondOperator() { return NULL; }
  // ref: precond.h:45.  No Thyra operator crosses the boundary: solveBlockProblem applies the device preconditioner to
  // every component itself (precond_ml.h:138-155) and raises the reference's error when there is none.
  virtual Thyra::LinearOpBase<double> *getBlockPrecondOperator() { return NULL; }
};

}  // namespace LAMMPS_NS
