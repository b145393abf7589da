// ilu.hpp -- block-Jacobi ILU(0) on the GPU (placeholder until the kernels land)
#pragma once
#include "core.hpp"
namespace isph {
inline int ilu_create(isph_ctx *, const isph_mat *, int, isph_ilu **) { return fail("bjacobi-ilu0 not built yet", __FILE__, __LINE__); }
inline int ilu_apply(isph_ctx *, const isph_ilu *, const double *, double *) { return fail("bjacobi-ilu0 not built yet", __FILE__, __LINE__); }
inline int ilu_export(isph_ctx *, const isph_ilu *, int *, int *, double *) { return fail("bjacobi-ilu0 not built yet", __FILE__, __LINE__); }
inline long long ilu_nnz(const isph_ilu *) { return 0; }
inline void ilu_destroy(isph_ilu *) {}
}  // namespace isph
