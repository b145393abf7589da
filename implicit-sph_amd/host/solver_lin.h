// solver_lin.h -- SolverLin with the reference's exact public surface
// (ref: solver_lin.h:23-98, solver_lin.cpp:30-160); state only, non-owning
// views over the caller's map / matrix / vectors.
#pragma once
#include <algorithm>
#include <cmath>

#include "isph_compat.h"
#include "precond.h"

#ifndef LAMMPS_SUCCESS
#define LAMMPS_SUCCESS 0   // ref: macrodef.h:23-24
#define LAMMPS_FAILURE -1  // ref: macrodef.h:20-21
#endif

namespace LAMMPS_NS {

class SolverLin {
 public:
  enum SolutionInitType { Random, Zero, Value };

  SolverLin(MPI_Comm &comm) : _comm(comm), _dim(0), _is_blocked(false), _is_singular(false) {}
  virtual ~SolverLin() {}

  int createLinearMap(int num_global_nodes, int index_base) {
    _map = Teuchos::rcp(new Epetra_Map(num_global_nodes, index_base, _comm));
    return LAMMPS_SUCCESS;
  }
  int createNodalMap(int num_local_nodes, int *gID) {
    _map = Teuchos::rcp(new Epetra_Map(-1, num_local_nodes, gID, 1, _comm));
    return LAMMPS_SUCCESS;
  }
  int createLoadMultiVector(int num_vectors) { return createLoadMultiVector(NULL, 0, num_vectors); }
  int createLoadMultiVector(double *b, int lda, int num_vectors) {
    _b = (b == NULL) ? Teuchos::rcp(new Epetra_MultiVector(*_map, num_vectors))
                     : Teuchos::rcp(new Epetra_MultiVector(View, *_map, b, lda, num_vectors));
    return LAMMPS_SUCCESS;
  }
  int createSolutionMultiVector(int num_vectors) { return createSolutionMultiVector(NULL, 0, num_vectors); }
  int createSolutionMultiVector(double *x, int lda, int num_vectors) {
    _x = (x == NULL) ? Teuchos::rcp(new Epetra_MultiVector(*_map, num_vectors))
                     : Teuchos::rcp(new Epetra_MultiVector(View, *_map, x, lda, num_vectors));
    return LAMMPS_SUCCESS;
  }
  // n = mask (or ones) / ||.||_2, ref: solver_lin.cpp:59-77.  The norm is the GLOBAL one (Epetra's Norm2 is an
  // all-reduce, solver_lin.cpp:72-74): with unequal local counts -- any masked null vector -- a rank-local norm would
  // scale the pieces differently and the result would not be in the null space.  The vector is kept between solves
  // (8 MB of first-touch page faults per solve at 10^6 rows otherwise).
  int createNullVector() {
    _n_sumsq = globalMaskSumSq();
    return fillNullVector();
  }
  // the collective half of createNullVector: sum of squares of the mask over ALL ranks (8 bytes all-reduced)
  double globalMaskSumSq() {
    const int len = _map->NumMyElements();
    const int m = _null_mask ? std::min(_null_mask->Length(), len) : len;
    const int *mask = _null_mask ? _null_mask->Values() : NULL;
    double sum = 0.0;
    for (int i = 0; i < m; ++i) { const double a = mask ? (double)mask[i] : 1.0; sum += a * a; }
    return _comm.SumAll(sum);
  }
  // the local half: n = mask / sqrt(_n_sumsq); no communication (getNullVector() may be called by one rank alone)
  int fillNullVector() {
    if (!_n || _n->MyLength() != _map->NumMyElements()) _n = Teuchos::rcp(new Epetra_Vector(*_map, true));
    double *v = _n->Values();
    const int len = _n->MyLength();
    const int m = _null_mask ? std::min(_null_mask->Length(), len) : len;
    const int *mask = _null_mask ? _null_mask->Values() : NULL;
    const double inv = 1.0 / std::sqrt(_n_sumsq);
    for (int i = 0; i < m; ++i) v[i] = (mask ? (double)mask[i] : 1.0) * inv;
    for (int i = m; i < len; ++i) v[i] = 0.0;
    _n_stale = false;
    return LAMMPS_SUCCESS;
  }
  int createBlockMatrix(const int dim, const char *) { _dim = dim; return LAMMPS_SUCCESS; }
  int freeBlockMatrix() { _dim = 0; return LAMMPS_SUCCESS; }

  void setNullVectorMask(Epetra_IntSerialDenseVector *mask) {
    if (mask != NULL) _null_mask = Teuchos::rcp(mask, false);
  }
  void setMatrixIsBlocked(const bool is_blocked) { _is_blocked = is_blocked; }
  void setMatrixIsSingular(const bool is_singular) { _is_singular = is_singular; }
  void setNodalMap(Epetra_Map *map) {
    if (map != NULL) _map = Teuchos::rcp(map, false);
  }
  void setMatrix(Epetra_CrsMatrix *A) {
    if (A != NULL) _A = Teuchos::rcp(A, false);
  }
  // ref: solver_lin.cpp:127-138 -- blocks are borrowed, NULL or out-of-range requests are ignored
  void setBlockBegin() { for (int k = 0; k < 9; ++k) _blk[k] = NULL; }
  void setBlock(const int i, const int j, const Epetra_CrsMatrix *A) {
    if ((A != NULL) && (i < _dim && j < _dim)) _blk[i * 3 + j] = A;
  }
  void setBlockEnd() {}

  void setInitialSolution(SolutionInitType init, double val = 0.0) {
    switch (init) {
      case Random: {  // Epetra Random(): uniform in (-1,1); any seed is acceptable for a start vector
        double *v = _x->Values();
        unsigned long long s = 88172645463325252ULL;
        for (int c = 0; c < _x->NumVectors(); ++c)
          for (int i = 0; i < _x->MyLength(); ++i) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            v[(size_t)c * _x->Stride() + i] = 2.0 * (double)(s >> 11) / 9007199254740992.0 - 1.0;
          }
        break;
      }
      case Zero: _x->PutScalar(0.0); break;
      case Value: _x->PutScalar(val); break;
    }
  }

  Teuchos::RCP<Epetra_Map> getNodalMap() const { return _map; }
  Teuchos::RCP<Epetra_MultiVector> getLoadMultiVector() { return _b; }
  Teuchos::RCP<Epetra_MultiVector> getSolutionMultiVector() { return _x; }
  Teuchos::RCP<Epetra_Vector> getNullVector() {
    if (_n_stale) { fillNullVector(); }  // local: the norm was all-reduced inside solveProblem, where every rank is
    return _n;
  }

  virtual void setParameters(Teuchos::ParameterList *param = NULL) {}
  virtual int solveProblem(PrecondWrapper *prec = NULL, const char *name = NULL) { return 0; }
  virtual int solveBlockProblem(PrecondWrapper *prec = NULL, const char *name = NULL) { return 0; }

 protected:
  Epetra_MpiComm _comm;
  Teuchos::RCP<Epetra_Map> _map;
  Teuchos::RCP<Teuchos::ParameterList> _param;
  Teuchos::RCP<Epetra_CrsMatrix> _A;
  int _dim;
  const Epetra_CrsMatrix *_blk[9] = {NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL};
  bool _is_blocked;
  Teuchos::RCP<Epetra_MultiVector> _x, _b;
  Teuchos::RCP<Epetra_Vector> _n;
  Teuchos::RCP<Epetra_IntSerialDenseVector> _null_mask;
  bool _is_singular;
  bool _n_stale = false;  // a singular solve ran without needing the host copy of the null vector
  double _n_sumsq = 1.0;  // global sum of squares of the mask, all-reduced by the last singular solve
};

}  // namespace LAMMPS_NS
