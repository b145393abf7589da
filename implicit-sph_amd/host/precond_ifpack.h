// precond_ifpack.h -- PrecondWrapper_Ifpack over the HIP block-Jacobi ILU(k)
// (ref: precond_ifpack.h:28-85: same parameter keys and defaults).
#pragma once
#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

#include "halo_lists.h"
#include "precond.h"

namespace LAMMPS_NS {

class PrecondWrapper_Ifpack : public PrecondWrapper {
 public:
  PrecondWrapper_Ifpack(MPI_Comm comm) : PrecondWrapper(comm) {}
  virtual ~PrecondWrapper_Ifpack() {}

  virtual Teuchos::ParameterList *setParameters(Teuchos::ParameterList *param = NULL) {
    if (param == NULL) {
      _param = Teuchos::rcp(new Teuchos::ParameterList);
      _param->set("fact: drop tolerance", 1e-9);   // ref: precond_ifpack.h:34 (unused by ILU)
      _param->set("fact: level-of-fill", 1);       // :35
      _param->set("schwarz: combine mode", "Add"); // :39
      _param->set("Precond Type", "ILU");          // :42
      _param->set("Overlap Level", 1);             // :43
      // device-side extension (not a reference key): rows per additive-Schwarz subdomain inside a rank.  512 = the
      // block stream of csrc/ilu.hpp (throughput path); 0 = one subdomain per rank = the whole local matrix, what the
      // reference factors (level-scheduled, csrc/schwarz.hpp); > 1024 = large subdomains with "Overlap Level" layers.
      _param->set("isph: block rows", 512);
    } else if (_param.get() != param) {
      _param = Teuchos::rcp(param, false);
    }
    return _param.get();
  }

  // Device-side extension (not a reference method): the additive-Schwarz subdomains INSIDE this rank as consecutive row
  // ranges, block b = rows block_ptr[b] .. block_ptr[b+1] (at most 1024 rows each).  The reference has one subdomain per
  // rank, a brick of LAMMPS' decomposition; an adapter that numbers the rank's particles brick by brick passes the brick
  // boundaries here and gets compact subdomains instead of a cut every "isph: block rows" rows.  Any level of fill,
  // overlap 0 (isph_prec_create_blocks_fill).  nblocks = 0 clears the table.
  void setSubdomains(int nblocks, const int *block_ptr) {
    _bptr.clear();
    if (nblocks > 0 && block_ptr != NULL) _bptr.assign(block_ptr, block_ptr + nblocks + 1);
  }

  // Device-side extension with the signature of PrecondWrapper_ML::setCoordinates (ref: precond_ml.h:25,63-94): the
  // coordinates of the rank's rows (three arrays of NumMyRows doubles, borrowed until the solve; z unused in 2-D).  The
  // reference hands coordinates to ML only (pair_isph.cpp:1290-1303); an adapter that makes the same three-line call for
  // this wrapper gets the library's own row numbering for the host matrix: subdomains = compact bricks of about 500
  // particles, whatever the atom order (on the 100^3 TGV system in create_atoms order 71 iterations instead of 164).
  // INTEGRATION.md shows the call.  NULL pointers clear them.
  void setCoordinates(const int dim, double *x, double *y, double *z) { storeCoordinates(dim, x, y, z); }

 protected:
  // the throughput path (block-Jacobi ILU(0), overlap 0 inside the rank) can be built during the matrix ingress
  virtual int fusedIngressBlockRows() {
    setParameters(_param.get());
    const int block = _param->get("isph: block rows", 512);
    // coordinates: the matrix crosses the link as it lies in the caller's memory and is permuted on the device; the set-up
    // follows in createOnDevice.  (isph_mat_create_csr_coords_bjacobi stages the rows in the new order on the host and
    // runs the set-up behind the link; on the 16-core share of the GPU box the gathered staging reaches 35 GB/s where
    // the flat copy reaches 54, so the fused form loses 5-10 ms against this one: profiles/r05_dropin.txt)
    if (_cx != nullptr) return 0;
    if (_param->get("Precond Type", "ILU") != "ILU" || _param->get("fact: level-of-fill", 1) != 0) return 0;
    if (tableUsable()) {
      int cap = 64;
      for (size_t b = 0; b + 1 < _bptr.size(); ++b) cap = std::max(cap, _bptr[b + 1] - _bptr[b]);
      noticeOnce(0, cap, _param->get("Overlap Level", 1));
      return (cap + 63) / 64 * 64;
    }
    if (block < 64 || block > 1024 || block % 64 != 0) return 0;
    noticeOnce(0, block, _param->get("Overlap Level", 1));
    return block;
  }
  // A table that does not fit the matrix at hand -- left over from a matrix with another row count, a subdomain above
  // the 1024 rows of the block stream, offsets that do not ascend -- is set aside with a notice and the solve goes on
  // with "isph: block rows" consecutive rows; it never fails the solve.
  bool tableUsable() {
    if (_bptr.empty()) return false;
    const int nrows = _A.get() != NULL ? _A->NumMyRows() : -1;
    bool ok = _bptr.front() == 0 && (nrows < 0 || _bptr.back() == nrows);
    for (size_t b = 0; ok && b + 1 < _bptr.size(); ++b) ok = _bptr[b + 1] > _bptr[b] && _bptr[b + 1] - _bptr[b] <= 1024;
    if (!ok && _comm.MyPID() == 0 && !_warned_table) {
      std::printf(">> PrecondWrapper_Ifpack(HIP): the subdomain table of setSubdomains does not fit this matrix (rows %d, table "
                  "ends at %d, blocks of 1..1024 rows required): ignored, subdomains are \"isph: block rows\" consecutive rows\n",
                  nrows, _bptr.back());
      _warned_table = true;
    }
    return ok;
  }
  virtual bool orderedIngressFusable() {
    setParameters(_param.get());
    const int block = _param->get("isph: block rows", 512);
    return _param->get("Precond Type", "ILU") == "ILU" && _param->get("fact: level-of-fill", 1) == 0 && block > 0 && block <= 1024;
  }
  virtual bool fusedIngressSubdomains(int &nblocks, const int *&bptr) {
    if (_cx != nullptr || !tableUsable()) return false;   // with coordinates the library's own bricks win over a table
    nblocks = (int)_bptr.size() - 1;
    bptr = _bptr.data();
    return true;
  }
  void noticeOnce(int fill, int block, int overlap) {
    if (_comm.MyPID() == 0 && !_warned) {
      if (block < 0)
        std::printf(">> PrecondWrapper_Ifpack(HIP): block-Jacobi ILU(%d) on the library's bricks of about 500 particles (rows "
                    "numbered by the coordinates of setCoordinates), overlap 0 (reference: one subdomain per rank, Overlap Level %d)\n",
                    fill, overlap);
      else
        std::printf(">> PrecondWrapper_Ifpack(HIP): block-Jacobi ILU(%d) on %d-row subdomains, overlap 0 (reference: one "
                    "subdomain per rank, Overlap Level %d); set \"isph: block rows\" = 0 for the reference's decomposition\n",
                    fill, block, overlap);
      _warned = true;
    }
  }
  virtual int createOnDevice(isph_ctx *ctx, const isph_mat *A) {
    setParameters(_param.get());
    const std::string type = _param->get("Precond Type", "ILU");
    const int fill = _param->get("fact: level-of-fill", 1);
    const int overlap = _param->get("Overlap Level", 1);
    if (type != "ILU") {
      std::fprintf(stderr, ">> PrecondWrapper_Ifpack(HIP): Precond Type '%s' is not available; only ILU\n", type.c_str());
      return ISPH_FAILURE;
    }
    if (fill < 0 || fill > 8) {
      std::fprintf(stderr, ">> PrecondWrapper_Ifpack(HIP): fact: level-of-fill %d is outside [0,8]\n", fill);
      return ISPH_FAILURE;
    }
    free();
    const int block = _param->get("isph: block rows", 512);
    const std::string mode = _param->get("schwarz: combine mode", "Add");
    // "isph: block rows" = 0: the reference's own decomposition -- one subdomain per rank = the whole local matrix,
    // ILU(fill) level-scheduled on the device (isph_prec_create_schwarz).  "Overlap Level" extends a subdomain by rows
    // of the neighbouring RANKS (Ifpack ignores it on one rank): level 1 = the rows of the matrix' ghost columns,
    // fetched with the matrix' importer (halo_lists.h) and factored with the rank's own (isph_prec_create_overlap);
    // level L > 1 = L such layers, each gathered by one all-to-all round (rows of non-neighbour ranks included).
    // The choice is made from rank-uniform information (the parameters and "does ANY rank have ghost columns"): the row
    // import below is collective, so a rank without ghost columns -- an isolated subdomain -- must enter it as well,
    // with empty lists, and a failure on one rank must fail all of them.
    const int any_ghosts = _A.get() != NULL ? _comm.MaxAll(_A->NumMyCols() > _A->NumMyRows() ? 1 : 0) : 0;
    if (block == 0 && overlap >= 1 && any_ghosts) {
      HaloLists H, XH;
      std::vector<int> rp, ci;
      std::vector<double> v;
      int bad = halo_lists_from_import(*_A, H) != ISPH_SUCCESS ? 1 : 0;
      bool layered = false;
#ifdef ISPH_HAVE_MPI
      // more than one rank: L layers of imported rows, each its own all-to-all round (halo_lists.h extend_rows_levels)
      if (_comm.NumProc() > 1) {
        layered = true;
        if (_comm.MaxAll(bad) == 0) bad = extend_rows_levels(*_A, _comm, H, overlap, rp, ci, v, XH) != ISPH_SUCCESS ? 1 : 0;
      }
#endif
      if (!layered) {
        if (_comm.MaxAll(bad) == 0) bad = extend_rows_one_layer(*_A, _comm, H, rp, ci, v) != ISPH_SUCCESS ? 1 : 0;
        XH = H;
        if (overlap > 1 && _comm.MyPID() == 0 && !_warned) {  // the self-peer importer of a one-rank run: one layer
          std::printf(">> PrecondWrapper_Ifpack(HIP): Overlap Level %d on one rank with a self importer is factored as Overlap Level 1\n", overlap);
          _warned = true;
        }
      }
      if (_comm.MaxAll(bad)) {
        if (_comm.MyPID() == 0)
          std::fprintf(stderr, ">> PrecondWrapper_Ifpack(HIP): cannot import the rows of the ghost columns (Overlap Level %d)\n", overlap);
        return ISPH_FAILURE;
      }
      const int next = (int)rp.size() - 1;
      isph_mat *Aext = NULL;
      int ierr = isph_mat_create_csr(ctx, next, next, rp.data(), ci.data(), v.data(), 0, &Aext);
      if (ierr != ISPH_SUCCESS) return ierr;
      ierr = isph_prec_create_overlap(ctx, Aext, _A->NumMyRows(), fill, (mode == "Zero") ? 1 : 0, XH.npeers(), XH.peers.data(),
                                      XH.send_ptr.data(), XH.send_idx.data(), XH.recv_ptr.data(), &_M);
      isph_mat_destroy(Aext);
      return ierr;
    }
    if (tableUsable()) {  // the caller's subdomains (setSubdomains), any level of fill
      noticeOnce(fill, 0, overlap);
      return isph_prec_create_blocks_fill(ctx, A, (int)_bptr.size() - 1, _bptr.data(), fill, &_M);
    }
    if (block == 0 || block > 1024) {
      isph_schwarz_params sp;
      isph_schwarz_params_default(&sp);
      sp.level_of_fill = fill;
      sp.block_size = block;
      sp.overlap = block == 0 ? 0 : overlap;    // subdomains inside one rank are extended by schwarz.hpp itself
      sp.combine = (mode == "Zero") ? 1 : 0;
      return isph_prec_create_schwarz(ctx, A, &sp, &_M);
    }
    {  // a matrix the library numbered itself (setCoordinates): its bricks are the subdomains
      long long oi[3] = {0, 0, 0};
      if (isph_mat_ordering_info(A, oi, NULL) == ISPH_SUCCESS && oi[0] == 1) {
        noticeOnce(fill, -1, overlap);
        const std::string kind = "bjacobi-ilu" + std::to_string(fill);
        return isph_prec_create(ctx, A, kind.c_str(), 0, &_M);
      }
    }
    // default: the throughput path -- block-Jacobi ILU(fill) on subdomains of `block` rows inside the rank, overlap 0.
    // This is NOT what the reference factors (one subdomain per rank, overlap 1): iteration counts differ (on the 100^3
    // TGV system 116 iterations against 49 for the whole-matrix ILU(0)); the notice is printed once, on every build.
    noticeOnce(fill, block, overlap);
    const std::string kind = "bjacobi-ilu" + std::to_string(fill);
    return isph_prec_create(ctx, A, kind.c_str(), block, &_M);
  }
  bool _warned = false, _warned_table = false;
  std::vector<int> _bptr;  // setSubdomains
};

}  // namespace LAMMPS_NS
