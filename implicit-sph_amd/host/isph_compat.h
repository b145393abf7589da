// isph_compat.h -- the small slice of the Epetra / Teuchos vocabulary that the
// SolverLin / PrecondWrapper signatures use (ref: solver_lin.h:23-98,
// precond.h:17-46), for builds WITHOUT Trilinos.  With -DHAVE_EPETRA the real
// Trilinos headers are used instead and these types are not defined.
//
// These are the product's own host-side value types (non-owning views over the
// caller's arrays); they carry exactly the members PairISPH and fix_qeq_reax
// touch on the objects they pass to / get back from the solver.
#pragma once
#ifdef HAVE_EPETRA
#include "Epetra_CrsMatrix.h"
#include "Epetra_IntSerialDenseVector.h"
#include "Epetra_Map.h"
#include "Epetra_MpiComm.h"
#include "Epetra_MultiVector.h"
#include "Epetra_Vector.h"
#include "Teuchos_ParameterList.hpp"
#include "Teuchos_RCP.hpp"
#else
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#ifdef ISPH_HAVE_MPI
#include <mpi.h>
#else
typedef int MPI_Comm;  // single-process builds: the communicator is a placeholder
#endif

namespace Teuchos {
template <class T>
using RCP = std::shared_ptr<T>;
// rcp(p,false): non-owning, like the reference's borrowed pointers (solver_lin.cpp:109-126)
template <class T>
RCP<T> rcp(T *p, bool owns = true) {
  return owns ? RCP<T>(p) : RCP<T>(p, [](T *) {});
}
const std::nullptr_t null = nullptr;

// typed key/value list with the get(name, default) idiom of Teuchos::ParameterList
class ParameterList {
 public:
  void set(const std::string &k, int v) { i_[k] = v; }
  void set(const std::string &k, bool v) { i_[k] = v ? 1 : 0; }
  void set(const std::string &k, double v) { d_[k] = v; }
  void set(const std::string &k, const char *v) { s_[k] = v; }
  void set(const std::string &k, const std::string &v) { s_[k] = v; }
  int get(const std::string &k, int def) const { auto it = i_.find(k); return it == i_.end() ? def : it->second; }
  bool get(const std::string &k, bool def) const { auto it = i_.find(k); return it == i_.end() ? def : it->second != 0; }
  double get(const std::string &k, double def) const {
    auto it = d_.find(k);
    if (it != d_.end()) return it->second;
    auto jt = i_.find(k);
    return jt == i_.end() ? def : (double)jt->second;
  }
  std::string get(const std::string &k, const char *def) const { auto it = s_.find(k); return it == s_.end() ? std::string(def) : it->second; }
  bool isParameter(const std::string &k) const { return i_.count(k) || d_.count(k) || s_.count(k); }
 private:
  std::map<std::string, int> i_;
  std::map<std::string, double> d_;
  std::map<std::string, std::string> s_;
};
}  // namespace Teuchos

// Thyra types that only appear in PrecondWrapper's signatures (precond.h:22,33-36,45); never instantiated here
namespace Thyra {
template <class Scalar> class PhysicallyBlockedLinearOpBase;
template <class Scalar> class LinearOpBase;
}  // namespace Thyra

enum Epetra_DataAccess { Copy, View };

// With -DISPH_HAVE_MPI (and <mpi.h> on the include path) the communicator is real: rank, size and the reductions the
// mirror classes need.  Without it this is a single-process build: one rank, and anything that needs a second rank
// fails loudly (SolverLin_HIP refuses NumProc() > 1, halo_lists.h returns an error for a foreign peer).
class Epetra_MpiComm {
 public:
#ifdef ISPH_HAVE_MPI
  explicit Epetra_MpiComm(MPI_Comm c = MPI_COMM_WORLD) : comm_(c) {
    MPI_Comm_rank(comm_, &rank_);
    MPI_Comm_size(comm_, &size_);
  }
  double SumAll(double x) const {
    double g = 0.0;
    MPI_Allreduce(&x, &g, 1, MPI_DOUBLE, MPI_SUM, comm_);
    return g;
  }
  int MaxAll(int x) const {
    int g = 0;
    MPI_Allreduce(&x, &g, 1, MPI_INT, MPI_MAX, comm_);
    return g;
  }
#else
  explicit Epetra_MpiComm(MPI_Comm c = 0) : comm_(c) {}
  double SumAll(double x) const { return x; }
  int MaxAll(int x) const { return x; }
#endif
  int MyPID() const { return rank_; }
  int NumProc() const { return size_; }
  MPI_Comm Comm() const { return comm_; }
 private:
  MPI_Comm comm_;
  int rank_ = 0, size_ = 1;
};

class Epetra_IntSerialDenseVector {
 public:
  Epetra_IntSerialDenseVector() {}
  explicit Epetra_IntSerialDenseVector(int n) : own_((size_t)n, 0), p_(own_.data()), n_(n) {}
  Epetra_IntSerialDenseVector(Epetra_DataAccess a, int *v, int n) : n_(n) {
    if (a == Copy) { own_.assign(v, v + n); p_ = own_.data(); } else p_ = v;
  }
  int Length() const { return n_; }
  int *Values() { return p_; }
  const int *Values() const { return p_; }
  int &operator[](int i) { return p_[i]; }
 private:
  std::vector<int> own_;
  int *p_ = nullptr;
  int n_ = 0;
};

// row map: locally owned global ids (atom tags), ref: pair_isph.cpp:1258-1259
class Epetra_Map {
 public:
  Epetra_Map(int nglobal, int nlocal, const int *gids, int index_base, const Epetra_MpiComm &)
      : nglobal_(nglobal < 0 ? nlocal : nglobal), base_(index_base), gid_(gids, gids + nlocal) {}
  Epetra_Map(int nglobal, int index_base, const Epetra_MpiComm &) : nglobal_(nglobal), base_(index_base), gid_((size_t)nglobal) {
    for (int i = 0; i < nglobal; ++i) gid_[(size_t)i] = i + index_base;
  }
  int NumMyElements() const { return (int)gid_.size(); }
  int NumGlobalElements() const { return nglobal_; }
  int GID(int lid) const { return gid_[(size_t)lid]; }
 private:
  int nglobal_, base_;
  std::vector<int> gid_;
};

// The import plan FillComplete builds for the ghost columns (Epetra_Import + its Epetra_MpiDistributor): which owned
// rows go to which rank (ProcsTo / LengthsTo / ExportLIDs, grouped by destination) and how many ghost values arrive
// from which rank (ProcsFrom / LengthsFrom; the ghost columns nrow.. are stored grouped by source rank in that order).
class Epetra_Import {
 public:
  Epetra_Import(int nto, const int *procs_to, const int *lengths_to, const int *export_lids, int nfrom,
                const int *procs_from, const int *lengths_from)
      : to_(procs_to, procs_to + nto), lto_(lengths_to, lengths_to + nto), from_(procs_from, procs_from + nfrom),
        lfrom_(lengths_from, lengths_from + nfrom) {
    int ns = 0;
    for (int k = 0; k < nto; ++k) ns += lengths_to[k];
    exp_.assign(export_lids, export_lids + ns);
    nrem_ = 0;
    for (int k = 0; k < nfrom; ++k) nrem_ += lengths_from[k];
  }
  int NumSends() const { return (int)to_.size(); }
  const int *ProcsTo() const { return to_.data(); }
  const int *LengthsTo() const { return lto_.data(); }
  int NumExportIDs() const { return (int)exp_.size(); }
  const int *ExportLIDs() const { return exp_.data(); }
  int NumReceives() const { return (int)from_.size(); }
  const int *ProcsFrom() const { return from_.data(); }
  const int *LengthsFrom() const { return lfrom_.data(); }
  int NumRemoteIDs() const { return nrem_; }
 private:
  std::vector<int> to_, lto_, exp_, from_, lfrom_;
  int nrem_ = 0;
};

// the filled matrix as Epetra hands it over after FillComplete+OptimizeStorage:
// contiguous CSR with local column ids (ExtractCrsDataPointers)
class Epetra_CrsMatrix {
 public:
  Epetra_CrsMatrix(int nrow, int ncol, int *rowptr, int *colidx, double *val, const Epetra_Import *importer = nullptr)
      : nrow_(nrow), ncol_(ncol), rp_(rowptr), ci_(colidx), v_(val), imp_(importer) {}
  const Epetra_Import *Importer() const { return imp_; }  // NULL when the matrix has no ghost columns
  int NumMyRows() const { return nrow_; }
  int NumMyCols() const { return ncol_; }
  int NumMyNonzeros() const { return rp_[nrow_]; }
  bool Filled() const { return true; }
  int ExtractCrsDataPointers(int *&rowptr, int *&colidx, double *&val) const {
    rowptr = rp_; colidx = ci_; val = v_;
    return 0;
  }
 private:
  int nrow_, ncol_;
  int *rp_, *ci_;
  double *v_;
  const Epetra_Import *imp_ = nullptr;
};

// column-major [lda x nvec] view or owned storage (solver_lin.cpp:45-58)
class Epetra_MultiVector {
 public:
  Epetra_MultiVector(const Epetra_Map &map, int nvec) : n_(map.NumMyElements()), lda_(map.NumMyElements()), nvec_(nvec),
                                                        own_((size_t)n_ * (size_t)nvec, 0.0), p_(own_.data()) {}
  Epetra_MultiVector(Epetra_DataAccess, const Epetra_Map &map, double *v, int lda, int nvec)
      : n_(map.NumMyElements()), lda_(lda), nvec_(nvec), p_(v) {}
  double *Values() { return p_; }
  const double *Values() const { return p_; }
  int NumVectors() const { return nvec_; }
  int MyLength() const { return n_; }
  int Stride() const { return lda_; }
  void PutScalar(double a) { for (int c = 0; c < nvec_; ++c) for (int i = 0; i < n_; ++i) p_[(size_t)c * lda_ + i] = a; }
  // *b = *x  (pair_isph.cpp:940-946)
  Epetra_MultiVector &operator=(const Epetra_MultiVector &o) {
    for (int c = 0; c < nvec_ && c < o.nvec_; ++c)
      std::memcpy(p_ + (size_t)c * lda_, o.p_ + (size_t)c * o.lda_, sizeof(double) * (size_t)(n_ < o.n_ ? n_ : o.n_));
    return *this;
  }
 protected:
  int n_, lda_, nvec_;
  std::vector<double> own_;
  double *p_;
};

class Epetra_Vector : public Epetra_MultiVector {
 public:
  explicit Epetra_Vector(const Epetra_Map &map, bool = true) : Epetra_MultiVector(map, 1) {}
};

class Epetra_Operator;  // only ever passed around as an opaque pointer here
#endif  // HAVE_EPETRA
