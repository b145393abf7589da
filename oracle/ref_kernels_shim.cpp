// TEST INFRASTRUCTURE.  C entry points around the reference's OWN kernel classes, compiled from the sources where they
// lie under /root/reference (kernel.h, kernel_wendland.h, kernel_quintic.h, kernel_cubic.h, filter.h, and the free templates of functor.h and mirror.h: the only files
// of the path that need nothing but the standard library).  Built by oracle/build.py into oracle/_ref/libisph_refkernels.so when
// /root/reference is present; nothing of the reference is copied into the repository -- the headers are found through
// the compiler's include path.  The library is a checker for the oracle's and the device's W and dW/dr
// (tests/test_oracle.py, tests/test_gpu_reference_tables.py); everything else of the reference needs Trilinos + LAMMPS.
#include "kernel_wendland.h"
#include "kernel_quintic.h"
#include "kernel_cubic.h"
#include "filter.h"
#include "functor.h"   // sphOperator<AntiSymmetric>: the pair operator of every gradient / divergence / Laplacian functor
#include "mirror.h"    // MirrorNothing, the mirror of the boundary conditions without a wall model

namespace {
LAMMPS_NS::KernelFunction *make(int kernel, int dim) {
  switch (kernel) {
  case 0: return new LAMMPS_NS::KernelFuncWendland((unsigned)dim);
  case 1: return new LAMMPS_NS::KernelFuncQuintic((unsigned)dim);
  default: return new LAMMPS_NS::KernelFuncCubic((unsigned)dim);
  }
}
}  // namespace

extern "C" {
// FilterBinary with setPairYes(filt_i, filt_j): the particle-kind tests every functor makes (filter.h:33-57)
int ref_filter_yes1(int filt_i, int ikind) {
  LAMMPS_NS::FilterBinary f;
  f.setPairYes(filt_i);
  return f.yes(ikind) ? 1 : 0;
}
int ref_filter_yes2(int filt_i, int filt_j, int ikind, int jkind) {
  LAMMPS_NS::FilterBinary f;
  f.setPairYes(filt_i, filt_j);
  return f.yes(ikind, jkind) ? 1 : 0;
}
// FilterMatchBinary (filter.h:83-104): what the solute-transport and applied-potential functors use
int ref_filter_match_yes1(int filt_i, int ikind) {
  LAMMPS_NS::FilterMatchBinary f;
  f.setPairYes(filt_i);
  return f.yes(ikind) ? 1 : 0;
}
int ref_filter_match_yes2(int filt_i, int filt_j, int ikind, int jkind) {
  LAMMPS_NS::FilterMatchBinary f;
  f.setPairYes(filt_i, filt_j);
  return f.yes(ikind, jkind) ? 1 : 0;
}
// sphOperator<true>(fi, fj) = fi + fj, sphOperator<false> = fj - fi (functor.h:9-20)
double ref_sph_operator(int antisym, double fi, double fj) {
  return antisym ? LAMMPS_NS::sphOperator<true>(fi, fj) : LAMMPS_NS::sphOperator<false>(fi, fj);
}
// MirrorNothing::computeMirrorCoefficient (mirror.h:19)
double ref_mirror_nothing(double r) {
  LAMMPS_NS::MirrorNothing m(2, nullptr, nullptr);
  return m.computeMirrorCoefficient(r);
}
// kernel: 0 Wendland, 1 Quintic, 2 Cubic (the oracle's numbering); the calls the functors make: kernel->val(r, h)
double ref_kernel_val(int kernel, int dim, double r, double h) {
  LAMMPS_NS::KernelFunction *k = make(kernel, dim);
  const double v = k->val(r, h);
  delete k;
  return v;
}
double ref_kernel_dval(int kernel, int dim, double r, double h) {
  LAMMPS_NS::KernelFunction *k = make(kernel, dim);
  const double v = k->dval(r, h);
  delete k;
  return v;
}
void ref_kernel_table(int kernel, int dim, double h, int n, const double *r, double *w, double *dw) {
  LAMMPS_NS::KernelFunction *k = make(kernel, dim);
  for (int i = 0; i < n; ++i) { w[i] = k->val(r[i], h); dw[i] = k->dval(r[i], h); }
  delete k;
}
}
