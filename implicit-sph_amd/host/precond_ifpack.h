// precond_ifpack.h -- PrecondWrapper_Ifpack over the HIP block-Jacobi ILU(k)
// (ref: precond_ifpack.h:28-85: same parameter keys and defaults).
#pragma once
#include <cstdio>
#include <string>

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
      // device-side extension: rows per additive-Schwarz subdomain (one Ifpack
      // rank's worth of rows).  Not a reference key.
      _param->set("isph: block rows", 512);
    } else if (_param.get() != param) {
      _param = Teuchos::rcp(param, false);
    }
    return _param.get();
  }

 protected:
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
    // "Overlap Level" extends an Ifpack subdomain by rows of the neighbouring RANKS (no effect on one rank).  The device
    // subdomains are blocks of "isph: block rows" rows inside a rank and are not extended: overlap 0 semantics.
    if (overlap != 0 && _comm.NumProc() > 1 && _comm.MyPID() == 0 && !_warned) {
      std::printf(">> PrecondWrapper_Ifpack(HIP): Overlap Level %d requested; this build provides overlap 0 "
                  "(block-Jacobi ILU(%d)) -- using that\n", overlap, fill);
      _warned = true;
    }
    free();
    const std::string kind = "bjacobi-ilu" + std::to_string(fill);
    return isph_prec_create(ctx, A, kind.c_str(), _param->get("isph: block rows", 512), &_M);
  }
  bool _warned = false;
};

}  // namespace LAMMPS_NS
