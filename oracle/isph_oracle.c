/*
 * isph_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See isph_oracle.h for scope and parity-pinning status.
 *
 * All "ref:" citations are relative to /root/reference/IMPLICIT-SPH/.
 * The code below is a from-scratch restatement of the algorithms in those
 * files (flat arrays, no Epetra/Teuchos), written to be read side by side
 * with them.  OpenMP is used only in the linear-algebra half so the oracle
 * can also serve as the all-core CPU baseline (one ILU block per thread ==
 * Ifpack additive Schwarz, overlap 0, on that many MPI ranks).
 */
#include "isph_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_EPS 1.0e-24 /* ISPH_EPSILON, ref: macrodef.h:6 */
#define G2(A, dim, i, j) ((A)[(j) * (dim) + (i)]) /* VIEW2, ref: macrodef.h:61-62 */

static double now_sec(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void orc_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ===================================================================== *
 *  SPH kernels
 * ===================================================================== */

/* ref: kernel_wendland.h:28-63, kernel_quintic.h:28-80, kernel_cubic.h:28-70.
 * The reference caches _h,_C inside the object; the value is the same
 * function of (r,h) restated here without the cache. */
static double kernel_norm(int kernel, int dim, double h) {
  switch (kernel) {
  case ORC_WENDLAND: return dim == 3 ? 21.0 / (16 * M_PI * pow(h, 3)) : 7.0 / (4 * M_PI * pow(h, 2));
  case ORC_QUINTIC:  return dim == 3 ? 14.0 / (pow(h, 3) * 1745.0 * M_PI) : 7.0 / (pow(h, 2) * 478.0 * M_PI);
  default:           return dim == 3 ? 1.0 / (pow(h, 3) * M_PI) : 10.0 / (pow(h, 2) * 7.0 * M_PI);
  }
}

double orc_kernel_val(int kernel, int dim, double r, double h) {
  const double C = kernel_norm(kernel, dim, h);
  const double s = fabs(r / h);
  double v = 0.0;
  switch (kernel) {
  case ORC_WENDLAND: /* kernel_wendland.h:50-58 */
    v = pow(1 - 0.5 * s, 4) * (2 * s + 1.) * (s < 2);
    break;
  case ORC_QUINTIC: /* kernel_quintic.h:48-66, fall-through switch */
    switch ((int)floor(s)) {
    case 0: v += 15.0 * pow(1.0 - s, 5); /* fallthrough */
    case 1: v -= 6.0 * pow(2.0 - s, 5);  /* fallthrough */
    case 2: v += pow(3.0 - s, 5);
    }
    break;
  default: /* kernel_cubic.h:45-56 */
    switch ((int)floor(s)) {
    case 0: v = 1.0 - 0.75 * (2 - s) * s * s; break;
    case 1: v = 0.25 * pow(2.0 - s, 3);
    }
  }
  return v * C;
}

double orc_kernel_dval(int kernel, int dim, double r, double h) {
  const double C = kernel_norm(kernel, dim, h);
  const double s = fabs(r / h);
  double v = 0.0;
  switch (kernel) {
  case ORC_WENDLAND: /* kernel_wendland.h:60-68 */
    v = -5.0 * s * pow(1 - 0.5 * s, 3) * (s < 2);
    break;
  case ORC_QUINTIC: /* kernel_quintic.h:68-82 */
    switch ((int)floor(s)) {
    case 0: v -= 75.0 * pow(1 - s, 4); /* fallthrough */
    case 1: v += 30.0 * pow(2 - s, 4); /* fallthrough */
    case 2: v -= 5 * pow(3 - s, 4);
    }
    break;
  default: /* kernel_cubic.h:58-70 */
    switch ((int)floor(s)) {
    case 0: v = (2.25 * s - 3) * s; break;
    case 1: v = -0.75 * pow(2 - s, 2);
    }
  }
  return v * C / h;
}

/* ===================================================================== *
 *  helpers shared by the functors
 * ===================================================================== */

static inline int kind_of(const orc_particles *P, int i) { return P->kind[P->type[i]]; }
static inline double tab(const orc_particles *P, const double *T, int it, int jt) {
  return T[it * (P->ntypes + 1) + jt];
}
/* FilterBinary::yes, ref: filter.h:47-53; with ORC_FILTER_MATCH set in fi: FilterMatchBinary::yes, ref: filter.h:97-103
 * (the row kind must EQUAL the first kind, the neighbour kind is still a mask) */
static inline int fyes1(int fi, int ikind) {
  if (fi & ORC_FILTER_MATCH) return ikind == (fi & ~ORC_FILTER_MATCH);
  return (ikind & fi) != 0;
}
static inline int fyes2(int fi, int fj, int ikind, int jkind) {
  if (fi & ORC_FILTER_MATCH) return ikind == (fi & ~ORC_FILTER_MATCH) && (jkind & fj);
  return (ikind & fi) && (jkind & fj);
}
/* the two tests as the functors see them, for the check against the reference's FilterBinary (oracle/_ref) */
int orc_filter_yes1(int filt_i, int ikind) { return fyes1(filt_i, ikind); }
int orc_filter_yes2(int filt_i, int filt_j, int ikind, int jkind) { return fyes2(filt_i, filt_j, ikind, jkind); }
/* sphOperator<AntiSymmetric>, ref: functor.h:9-20 */
static inline double sph_op(int antisym, double fi, double fj) {
  return antisym ? (fi + fj) : (fj - fi);
}

double orc_sph_operator(int antisym, double fi, double fj) { return sph_op(antisym, fi, fj); }  /* for the check against oracle/_ref */

/* MirrorMorrisHolmes::computeMirrorCoefficient, ref: mirror_morris_holmes.h:39-52
 *   d = 2 cut (pnd V - 1/2) + eps ; coeff = 1 + d_j / max(d_i, safe*h) */
static double mirror_coeff(const orc_particles *P, int morris, int i, int j, double cut) {
  if (!morris || P->pnd == NULL) return 1.0; /* MirrorNothing, ref: mirror.h:19 */
  const double hij = tab(P, P->h, P->type[i], P->type[j]);
  double di = 2.0 * cut * (P->pnd[i] * P->vfrac[i] - 0.5) + ORC_EPS;
  double dj = 2.0 * cut * (P->pnd[j] * P->vfrac[j] - 0.5) + ORC_EPS;
  const double dmin = P->morris_safe_coeff * hij;
  if (di < dmin) di = dmin;
  return 1.0 + dj / di;
}

/* forward_comm_pair for a per-atom array on one rank: ghosts take the
 * owner's value (ref: pair_isph.cpp pack/unpack_forward_comm). */
void orc_forward_comm(const orc_particles *P, double *arr, int ncomp) {
  for (int j = P->nlocal; j < P->nall; ++j) {
    const int o = P->owner[j];
    if (o >= 0)
      for (int k = 0; k < ncomp; ++k) arr[j * ncomp + k] = arr[o * ncomp + k];
  }
}

/* ===================================================================== *
 *  pre-computation
 * ===================================================================== */

/* ref: functor_volume.h:40-80.  V_i = 1 / (W(0) + sum_j W(r_ij)). */
void orc_compute_volumes(const orc_particles *P) {
  const int dim = P->dim;
  for (int i = 0; i < P->nlocal; ++i) {
    const int it = P->type[i];
    double w = orc_kernel_val(P->kernel, dim, 0.0, tab(P, P->h, it, it));
    for (int jj = P->neigh_ptr[i]; jj < P->neigh_ptr[i + 1]; ++jj) {
      const int j = P->neigh_idx[jj];
      const int jt = P->type[j];
      double rsq = 0.0;
      for (int k = 0; k < dim; ++k) {
        const double r = P->x[3 * i + k] - P->x[3 * j + k];
        rsq += r * r;
      }
      if (rsq < tab(P, P->cutsq, it, jt))
        w += orc_kernel_val(P->kernel, dim, sqrt(rsq), tab(P, P->h, it, jt));
    }
    P->vfrac[i] = 1.0 / w;
  }
  orc_forward_comm(P, P->vfrac, 1); /* functor_volume.h:76-80 */
}

/* Particle number density of the MorrisHolmes mirror: the pnd part of FunctorOuterNormal::operator()
 * (ref: functor_normal.h:57-133) as PairISPH_Corrected::computeNormals runs it, once with the filter (Fluid,Solid)
 * and once with (Solid,Fluid) (pair_isph_corrected.cpp:396-419): a neighbour inside the cut that does NOT pass the
 * pair filter -- i.e. is not of the opposite phase -- adds W(r), the particle itself W(0) last (:109,:115). */
void orc_compute_pnd(const orc_particles *P, double *pnd) {
  const int dim = P->dim;
  for (int i = 0; i < P->nlocal; ++i) {
    const int it = P->type[i], ikind = kind_of(P, i);
    pnd[i] = 0.0;
    if (!(ikind & (ORC_KIND_FLUID | ORC_KIND_SOLID))) continue;
    const int opposite = (ikind & ORC_KIND_SOLID) ? ORC_KIND_FLUID : ORC_KIND_SOLID;
    double w = 0.0;
    for (int jj = P->neigh_ptr[i]; jj < P->neigh_ptr[i + 1]; ++jj) {
      const int j = P->neigh_idx[jj];
      const int jt = P->type[j];
      double rsq = 0.0;
      for (int k = 0; k < dim; ++k) {
        const double r = P->x[3 * i + k] - P->x[3 * j + k];
        rsq += r * r;
      }
      if (rsq < tab(P, P->cutsq, it, jt) && !(kind_of(P, j) & opposite))
        w += orc_kernel_val(P->kernel, dim, sqrt(rsq) + ORC_EPS, tab(P, P->h, it, jt));
    }
    pnd[i] = w + orc_kernel_val(P->kernel, dim, 0.0, tab(P, P->h, it, it));
  }
  orc_forward_comm(P, pnd, 1); /* pair_isph_corrected.cpp:1360,1374: pnd travels with the normal */
}

/* dense LU with partial pivoting, column-major n x n, nrhs right-hand sides
 * (what LAPACK dgesv computes; ref: utils_reference.cpp:398-407). */
static int dense_gesv(int n, double *A, int nrhs, double *B) {
  for (int k = 0; k < n; ++k) {
    int p = k;
    double amax = fabs(G2(A, n, k, k));
    for (int i = k + 1; i < n; ++i)
      if (fabs(G2(A, n, i, k)) > amax) { amax = fabs(G2(A, n, i, k)); p = i; }
    if (amax == 0.0) return -1;
    if (p != k) {
      for (int j = 0; j < n; ++j) { double t = G2(A, n, k, j); G2(A, n, k, j) = G2(A, n, p, j); G2(A, n, p, j) = t; }
      for (int j = 0; j < nrhs; ++j) { double t = G2(B, n, k, j); G2(B, n, k, j) = G2(B, n, p, j); G2(B, n, p, j) = t; }
    }
    const double piv = 1.0 / G2(A, n, k, k);
    for (int i = k + 1; i < n; ++i) {
      const double l = G2(A, n, i, k) * piv;
      G2(A, n, i, k) = l;
      if (l != 0.0) {
        for (int j = k + 1; j < n; ++j) G2(A, n, i, j) -= l * G2(A, n, k, j);
        for (int j = 0; j < nrhs; ++j) G2(B, n, i, j) -= l * G2(B, n, k, j);
      }
    }
  }
  for (int j = 0; j < nrhs; ++j)
    for (int i = n - 1; i >= 0; --i) {
      double s = G2(B, n, i, j);
      for (int k = i + 1; k < n; ++k) s -= G2(A, n, i, k) * G2(B, n, k, j);
      G2(B, n, i, j) = s / G2(A, n, i, i);
    }
  return 0;
}

/* UtilsReference::invertDenseMatrix, ref: utils_reference.cpp:251-326:
 * closed-form adjugate for dim 1..3. */
static void invert_small(int dim, const double *A, double *B) {
  if (dim == 1) { B[0] = 1.0 / A[0]; return; }
  if (dim == 2) {
    const double det = G2(A, 2, 0, 0) * G2(A, 2, 1, 1) - G2(A, 2, 0, 1) * G2(A, 2, 1, 0);
    G2(B, 2, 0, 0) = G2(A, 2, 1, 1) / det;
    G2(B, 2, 1, 1) = G2(A, 2, 0, 0) / det;
    G2(B, 2, 1, 0) = -G2(A, 2, 1, 0) / det;
    G2(B, 2, 0, 1) = -G2(A, 2, 0, 1) / det;
    return;
  }
  const double c00 = G2(A, 3, 1, 1) * G2(A, 3, 2, 2) - G2(A, 3, 2, 1) * G2(A, 3, 1, 2);
  const double c01 = -G2(A, 3, 1, 0) * G2(A, 3, 2, 2) + G2(A, 3, 2, 0) * G2(A, 3, 1, 2);
  const double c02 = G2(A, 3, 1, 0) * G2(A, 3, 2, 1) - G2(A, 3, 2, 0) * G2(A, 3, 1, 1);
  const double det = G2(A, 3, 0, 0) * c00 + G2(A, 3, 0, 1) * c01 + G2(A, 3, 0, 2) * c02;
  G2(B, 3, 0, 0) = c00 / det;
  G2(B, 3, 1, 0) = c01 / det;
  G2(B, 3, 2, 0) = c02 / det;
  G2(B, 3, 0, 1) = (-G2(A, 3, 0, 1) * G2(A, 3, 2, 2) + G2(A, 3, 2, 1) * G2(A, 3, 0, 2)) / det;
  G2(B, 3, 1, 1) = (G2(A, 3, 0, 0) * G2(A, 3, 2, 2) - G2(A, 3, 2, 0) * G2(A, 3, 0, 2)) / det;
  G2(B, 3, 2, 1) = (-G2(A, 3, 0, 0) * G2(A, 3, 2, 1) + G2(A, 3, 2, 0) * G2(A, 3, 0, 1)) / det;
  G2(B, 3, 0, 2) = (G2(A, 3, 0, 1) * G2(A, 3, 1, 2) - G2(A, 3, 1, 1) * G2(A, 3, 0, 2)) / det;
  G2(B, 3, 1, 2) = (-G2(A, 3, 0, 0) * G2(A, 3, 1, 2) + G2(A, 3, 1, 0) * G2(A, 3, 0, 2)) / det;
  G2(B, 3, 2, 2) = (G2(A, 3, 0, 0) * G2(A, 3, 1, 1) - G2(A, 3, 1, 0) * G2(A, 3, 0, 1)) / det;
}

/* ref: functor_gradient_correction.h:23-71.
 * G_i = ( - sum_j r_ij (x) r_ij  W'/r  V_j )^-1 , stored column-major. */
void orc_compute_gradient_correction(const orc_particles *P) {
  const int dim = P->dim, d2 = dim * dim;
  for (int i = 0; i < P->nlocal; ++i) {
    const int it = P->type[i];
    double G[9] = {0};
    for (int jj = P->neigh_ptr[i]; jj < P->neigh_ptr[i + 1]; ++jj) {
      const int j = P->neigh_idx[jj];
      const int jt = P->type[j];
      double rsq = 0.0, rij[3] = {0, 0, 0};
      for (int k = 0; k < dim; ++k) {
        rij[k] = P->x[3 * i + k] - P->x[3 * j + k];
        rsq += rij[k] * rij[k];
      }
      if (rsq < tab(P, P->cutsq, it, jt)) {
        const double r = sqrt(rsq) + ORC_EPS;
        const double dwdr = orc_kernel_dval(P->kernel, dim, r, tab(P, P->h, it, jt));
        for (int k2 = 0; k2 < dim; ++k2)
          for (int k1 = 0; k1 < dim; ++k1)
            G2(G, dim, k1, k2) -= rij[k1] * rij[k2] * dwdr / r * P->vfrac[j];
      }
    }
    invert_small(dim, G, &P->Gc[(size_t)i * d2]);
  }
}

/* ref: functor_laplacian_correction.h:24-153.
 * Builds the dimL x dimL system for the packed symmetric tensor L_i and
 * solves it with LU (dgesv). Returns number of singular solves. */
int orc_compute_laplacian_correction(const orc_particles *P) {
  const int dim = P->dim, d2 = dim * dim, dL = dim * (dim + 1) / 2;
  int nfail = 0;
  for (int i = 0; i < P->nlocal; ++i) {
    const int it = P->type[i];
    double A[27] = {0}, L[36] = {0};
    const double *G = &P->Gc[(size_t)i * d2];
    /* pass 1: third-order tensor A^{kmn} = sum_j a_ij^k r^m r^n  (:45-85) */
    for (int jj = P->neigh_ptr[i]; jj < P->neigh_ptr[i + 1]; ++jj) {
      const int j = P->neigh_idx[jj];
      const int jt = P->type[j];
      double rsq = 0.0, rij[3] = {0, 0, 0};
      for (int k = 0; k < dim; ++k) {
        rij[k] = P->x[3 * i + k] - P->x[3 * j + k];
        rsq += rij[k] * rij[k];
      }
      if (rsq < tab(P, P->cutsq, it, jt)) {
        const double r = sqrt(rsq) + ORC_EPS;
        const double dwdr = orc_kernel_dval(P->kernel, dim, r, tab(P, P->h, it, jt));
        double aij[3] = {0, 0, 0};
        for (int k2 = 0; k2 < dim; ++k2) {
          for (int k1 = 0; k1 < dim; ++k1) aij[k2] += G2(G, dim, k1, k2) * rij[k1];
          aij[k2] *= dwdr / r * P->vfrac[j];
        }
        for (int k3 = 0; k3 < dim; ++k3) {
          double *slice = &A[k3 * d2];
          for (int k2 = 0; k2 < dim; ++k2)
            for (int k1 = 0; k1 < k2 + 1; ++k1)
              G2(slice, dim, k1, k2) += aij[k3] * rij[k1] * rij[k2];
        }
      }
    }
    /* pass 2: linear system (:87-141) */
    for (int jj = P->neigh_ptr[i]; jj < P->neigh_ptr[i + 1]; ++jj) {
      const int j = P->neigh_idx[jj];
      const int jt = P->type[j];
      double rsq = 0.0, rij[3] = {0, 0, 0};
      for (int k = 0; k < dim; ++k) {
        rij[k] = P->x[3 * i + k] - P->x[3 * j + k];
        rsq += rij[k] * rij[k];
      }
      if (rsq < tab(P, P->cutsq, it, jt)) {
        const double r = sqrt(rsq) + ORC_EPS;
        const double dwdr = orc_kernel_dval(P->kernel, dim, r, tab(P, P->h, it, jt));
        double eij[3] = {0, 0, 0};
        for (int k = 0; k < dim; ++k) eij[k] = rij[k] / r;
        double C[9] = {0};
        for (int k3 = 0; k3 < dim; ++k3) {
          const double *slice = &A[k3 * d2];
          for (int k2 = 0; k2 < dim; ++k2)
            for (int k1 = 0; k1 < k2 + 1; ++k1)
              G2(C, dim, k1, k2) += G2(slice, dim, k1, k2) * eij[k3];
        }
        for (int k2 = 0; k2 < dim; ++k2)
          for (int k1 = 0; k1 < k2 + 1; ++k1) {
            G2(C, dim, k1, k2) += rij[k1] * eij[k2];
            G2(C, dim, k1, k2) *= dwdr * P->vfrac[j];
          }
        const double scale[2] = {2.0, 1.0};
        for (int k4 = 0, op = 0; k4 < dim; ++k4)
          for (int k3 = 0; k3 < k4 + 1; ++k3, ++op)
            for (int k2 = 0, mn = 0; k2 < dim; ++k2)
              for (int k1 = 0; k1 < k2 + 1; ++k1, ++mn)
                G2(L, dL, mn, op) += G2(C, dim, k1, k2) * eij[k3] * eij[k4] * scale[k3 == k4];
      }
    }
    double *Lc = &P->Lc[(size_t)i * dL];
    for (int k2 = 0, op = 0; k2 < dim; ++k2)
      for (int k1 = 0; k1 < k2 + 1; ++k1, ++op) Lc[op] = -(double)(k1 == k2);
    if (dense_gesv(dL, L, 1, Lc) != 0) ++nfail;
  }
  return nfail;
}

/* ===================================================================== *
 *  graph
 * ===================================================================== */

static int cmp_int(const void *a, const void *b) {
  const int x = *(const int *)a, y = *(const int *)b;
  return (x > y) - (x < y);
}

/* ref: functor_graph.h:38-99.  Row i <- {col(j): r_ij^2 < cutsq} U {col(i)};
 * FillComplete sorts the indices and merges duplicates (periodic images
 * share a tag).  Returns nnz, or -1 if cap is too small. */
int orc_graph(const orc_particles *P, int *rowptr, int *colidx, int cap) {
  const int dim = P->dim;
  int nnz = 0;
  rowptr[0] = 0;
  for (int i = 0; i < P->nlocal; ++i) {
    const int it = P->type[i];
    const int start = nnz;
    for (int jj = P->neigh_ptr[i]; jj < P->neigh_ptr[i + 1]; ++jj) {
      const int j = P->neigh_idx[jj];
      double rsq = 0.0;
      for (int k = 0; k < dim; ++k) {
        const double r = P->x[3 * i + k] - P->x[3 * j + k];
        rsq += r * r;
      }
      if (rsq < tab(P, P->cutsq, it, P->type[j])) {
        if (nnz >= cap) return -1;
        colidx[nnz++] = P->colmap[j];
      }
    }
    if (nnz >= cap) return -1;
    colidx[nnz++] = P->colmap[i];
    qsort(colidx + start, (size_t)(nnz - start), sizeof(int), cmp_int);
    int w = start;
    for (int k = start; k < nnz; ++k)
      if (w == start || colidx[k] != colidx[w - 1]) colidx[w++] = colidx[k];
    nnz = w;
    rowptr[i + 1] = nnz;
  }
  return nnz;
}

/* Epetra SumIntoGlobalValues on a filled static graph: per-entry column
 * search, values accumulate on duplicates. */
static int row_find(const int *colidx, int lo, int hi, int col) {
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (colidx[mid] < col) lo = mid + 1; else hi = mid;
  }
  return lo;
}
static int sum_into(const int *rowptr, const int *colidx, double *val, int row,
                    int cnt, const double *v, const int *c) {
  int bad = 0;
  for (int k = 0; k < cnt; ++k) {
    const int p = row_find(colidx, rowptr[row], rowptr[row + 1], c[k]);
    if (p < rowptr[row + 1] && colidx[p] == c[k]) val[p] += v[k]; else ++bad;
  }
  return bad;
}

/* ===================================================================== *
 *  Laplacian matrix rows
 * ===================================================================== */

/* ref: functor_laplacian_matrix.h:73-316 (scalar matrix branch, _iblock<0).
 * Row i of alpha * div(m grad .) in two neighbour sweeps; see SURVEY
 * Appendix A.  Returns the number of entries that were not in the graph. */
int orc_laplacian_matrix(const orc_particles *P, int antisym, double alpha,
                         const double *material, int filt_i, int filt_j,
                         int morris_holmes,
                         const int *rowptr, const int *colidx, double *val_out) {
  const int dim = P->dim, d2 = dim * dim, dL = dim * (dim + 1) / 2;
  int maxn = 0;
  for (int i = 0; i < P->nlocal; ++i) {
    const int n = P->neigh_ptr[i + 1] - P->neigh_ptr[i] + 1;
    if (n > maxn) maxn = n;
  }
  double *val = (double *)malloc(sizeof(double) * (size_t)maxn);
  int *idx = (int *)malloc(sizeof(int) * (size_t)maxn);
  double Gi[9] = {0}, Li[6] = {0};
  for (int k = 0; k < dim; ++k) G2(Gi, dim, k, k) = 1.0; /* pair_isph_corrected.cpp:343-346 */
  for (int k2 = 0, op = 0; k2 < dim; ++k2)
    for (int k1 = 0; k1 < k2 + 1; ++k1, ++op) Li[op] = (k1 == k2); /* :363-366 */
  int bad = 0;

  for (int i = 0; i < P->nlocal; ++i) {
    const int it = P->type[i], ikind = kind_of(P, i);
    const double mi = material ? material[i] : 1.0;

    if (!fyes1(filt_i, ikind)) { /* :88-96  ReplaceGlobalValues(tag_i,{tag_i:0}) */
      const int p = row_find(colidx, rowptr[i], rowptr[i + 1], P->colmap[i]);
      if (p < rowptr[i + 1] && colidx[p] == P->colmap[i]) val_out[p] = 0.0; else ++bad;
      continue;
    }
    const double *L = antisym ? Li : &P->Lc[(size_t)i * dL];
    const double *G = antisym ? Gi : &P->Gc[(size_t)i * d2];
    double grad_m[3] = {0, 0, 0}, ci[3] = {0, 0, 0};
    int cnt = 0;
    { /* pass 1, :127-201 */
      double diag = 0.0;
      for (int jj = P->neigh_ptr[i]; jj < P->neigh_ptr[i + 1]; ++jj) {
        const int j = P->neigh_idx[jj];
        const int jt = P->type[j], jkind = kind_of(P, j);
        const double mj = material ? material[j] : 1.0;
        double rsq = 0.0, rij[3] = {0, 0, 0};
        for (int k = 0; k < dim; ++k) {
          rij[k] = P->x[3 * i + k] - P->x[3 * j + k];
          rsq += rij[k] * rij[k];
        }
        const double cutsq = tab(P, P->cutsq, it, jt);
        if (rsq < cutsq) {
          double coeff = fyes2(filt_i, filt_j, ikind, ikind);
          if (!(ikind & ORC_KIND_SOLID) && (jkind & ORC_KIND_SOLID))
            coeff = fyes2(filt_i, filt_j, ikind, jkind) ? mirror_coeff(P, morris_holmes, i, j, sqrt(cutsq)) : 0.0;
          const double r = sqrt(rsq) + ORC_EPS;
          const double dwdr = orc_kernel_dval(P->kernel, dim, r, tab(P, P->h, it, jt));
          double eij[3] = {0, 0, 0};
          for (int k = 0; k < dim; ++k) eij[k] = rij[k] / r;
          const double vfrac = antisym ? sqrt(P->vfrac[i] * P->vfrac[j]) : P->vfrac[j];
          const double vjtmp = dwdr * vfrac;
          for (int k2 = 0; k2 < dim; ++k2) {
            double gitmp = 0.0;
            for (int k1 = 0; k1 < dim; ++k1) gitmp += G2(G, dim, k1, k2) * eij[k1];
            const double ijtmp = gitmp * vjtmp;
            if (ikind & jkind) grad_m[k2] += ijtmp * sph_op(antisym, mi, mj);
          }
          double aij = 0.0;
          const double scale_a[2] = {2.0, 1.0};
          for (int k2 = 0, op = 0; k2 < dim; ++k2)
            for (int k1 = 0; k1 < k2 + 1; ++k1, ++op)
              aij += L[op] * eij[k1] * eij[k2] * scale_a[k1 == k2];
          aij *= 2.0 * dwdr * vfrac;
          if (!antisym)
            for (int k = 0; k < dim; ++k) ci[k] += aij * eij[k];
          aij *= mi * coeff / r;
          val[cnt] = -aij;
          diag += aij;
          idx[cnt] = P->colmap[j];
          ++cnt;
        }
      }
      val[cnt] = diag;
      idx[cnt] = P->colmap[i];
      ++cnt;
    }
    { /* pass 2, :204-264 */
      double diag = 0.0;
      cnt = 0;
      for (int jj = P->neigh_ptr[i]; jj < P->neigh_ptr[i + 1]; ++jj) {
        const int j = P->neigh_idx[jj];
        const int jt = P->type[j], jkind = kind_of(P, j);
        double rsq = 0.0, rij[3] = {0, 0, 0};
        for (int k = 0; k < dim; ++k) {
          rij[k] = P->x[3 * i + k] - P->x[3 * j + k];
          rsq += rij[k] * rij[k];
        }
        if (rsq < tab(P, P->cutsq, it, jt)) {
          double coeff = fyes2(filt_i, filt_j, ikind, ikind);
          if (!(ikind & ORC_KIND_SOLID) && (jkind & ORC_KIND_SOLID))
            coeff = fyes2(filt_i, filt_j, ikind, jkind);
          /* morris_holmes == 2: the OPERATOR form FunctorOuterLaplacian_MorrisHolmes (functor_laplacian.h:143-160,
           * 240-243) carries the mirror coefficient in its gradient part too; the matrix functor does not (:225-227).
           * Only used to reproduce the reference's Poisson-Boltzmann channel table (oracle/pb_channel.py). */
          if (morris_holmes == 2 && !(ikind & ORC_KIND_SOLID) && (jkind & ORC_KIND_SOLID) && coeff != 0.0)
            coeff = mirror_coeff(P, morris_holmes, i, j, sqrt(tab(P, P->cutsq, it, jt)));
          const double r = sqrt(rsq) + ORC_EPS;
          const double dwdr = orc_kernel_dval(P->kernel, dim, r, tab(P, P->h, it, jt));
          const double vfrac = antisym ? sqrt(P->vfrac[i] * P->vfrac[j]) : P->vfrac[j];
          const double vjtmp = dwdr * vfrac;
          double eij[3] = {0, 0, 0}, bij[3] = {0, 0, 0};
          for (int k = 0; k < dim; ++k) eij[k] = rij[k] / r;
          for (int k2 = 0; k2 < dim; ++k2)
            for (int k1 = 0; k1 < dim; ++k1) bij[k2] += G2(G, dim, k1, k2) * eij[k1];
          double bc = 0.0, bg = 0.0;
          for (int k = 0; k < dim; ++k) { bc += bij[k] * ci[k]; bg += bij[k] * grad_m[k]; }
          const double tmp = coeff * (mi * bc * vjtmp - bg * vjtmp);
          val[cnt] -= tmp;
          diag += tmp;
          ++cnt;
        }
      }
      val[cnt] += diag;
      ++cnt;
    }
    for (int k = 0; k < cnt; ++k) val[k] *= alpha; /* :267 _val.Scale(alpha) */
    bad += sum_into(rowptr, colidx, val_out, i, cnt, val, idx); /* :269-270 */
  }
  free(val);
  free(idx);
  return bad;
}

/* ===================================================================== *
 *  divergence / gradient / matrix-free laplacian
 * ===================================================================== */

/* ref: functor_divergence.h:54-124. f is [nall][3]. */
void orc_divergence(const orc_particles *P, int antisym, const double *f,
                    double alpha, int use_filter, int filt_i, int filt_j,
                    int morris_holmes, double *div) {
  const int dim = P->dim, d2 = dim * dim;
  double Gi[9] = {0};
  for (int k = 0; k < dim; ++k) G2(Gi, dim, k, k) = 1.0;
  for (int i = 0; i < P->nlocal; ++i) {
    const int it = P->type[i], ikind = kind_of(P, i);
    double d = 0.0;
    div[i] = 0.0;
    if (use_filter && !fyes1(filt_i, ikind)) continue;
    const double *G = antisym ? Gi : &P->Gc[(size_t)i * d2];
    for (int jj = P->neigh_ptr[i]; jj < P->neigh_ptr[i + 1]; ++jj) {
      const int j = P->neigh_idx[jj];
      const int jt = P->type[j], jkind = kind_of(P, j);
      if (use_filter && !fyes2(filt_i, filt_j, ikind, jkind)) continue;
      double rsq = 0.0, rij[3] = {0, 0, 0};
      for (int k = 0; k < dim; ++k) {
        rij[k] = P->x[3 * i + k] - P->x[3 * j + k];
        rsq += rij[k] * rij[k];
      }
      const double cutsq = tab(P, P->cutsq, it, jt);
      if (rsq < cutsq) {
        double coeff = 1.0;
        if (!(ikind & ORC_KIND_SOLID) && (jkind & ORC_KIND_SOLID))
          coeff = mirror_coeff(P, morris_holmes, i, j, sqrt(cutsq));
        const double r = sqrt(rsq) + ORC_EPS;
        const double dwdr = orc_kernel_dval(P->kernel, dim, r, tab(P, P->h, it, jt));
        const double vfrac = antisym ? sqrt(P->vfrac[i] * P->vfrac[j]) : P->vfrac[j];
        const double vjtmp = dwdr / r * vfrac * coeff;
        for (int k2 = 0; k2 < dim; ++k2) {
          double gitmp = 0.0;
          for (int k1 = 0; k1 < dim; ++k1) gitmp += G2(G, dim, k1, k2) * rij[k1];
          d += gitmp * sph_op(antisym, f[3 * i + k2], f[3 * j + k2]) * vjtmp;
        }
      }
    }
    div[i] = d * alpha;
  }
}

/* ref: functor_gradient.h:78-170 (scalar field). f is [nall]; grad [nlocal][3]. */
void orc_gradient(const orc_particles *P, int antisym, const double *f,
                  double alpha, int use_filter, int filt_i, int filt_j, double *grad) {
  const int dim = P->dim, d2 = dim * dim;
  double Gi[9] = {0};
  for (int k = 0; k < dim; ++k) G2(Gi, dim, k, k) = 1.0;
  for (int i = 0; i < P->nlocal; ++i) {
    const int it = P->type[i], ikind = kind_of(P, i);
    double g[3] = {0, 0, 0};
    grad[3 * i] = grad[3 * i + 1] = grad[3 * i + 2] = 0.0;
    if (use_filter && !fyes1(filt_i, ikind)) continue;
    const double *G = antisym ? Gi : &P->Gc[(size_t)i * d2];
    for (int jj = P->neigh_ptr[i]; jj < P->neigh_ptr[i + 1]; ++jj) {
      const int j = P->neigh_idx[jj];
      const int jt = P->type[j], jkind = kind_of(P, j);
      if (use_filter && !fyes2(filt_i, filt_j, ikind, jkind)) continue;
      double rsq = 0.0, rij[3] = {0, 0, 0};
      for (int k = 0; k < dim; ++k) {
        rij[k] = P->x[3 * i + k] - P->x[3 * j + k];
        rsq += rij[k] * rij[k];
      }
      if (rsq < tab(P, P->cutsq, it, jt)) {
        const double r = sqrt(rsq) + ORC_EPS;
        const double dwdr = orc_kernel_dval(P->kernel, dim, r, tab(P, P->h, it, jt));
        const double vfrac = antisym ? sqrt(P->vfrac[i] * P->vfrac[j]) : P->vfrac[j];
        const double vjtmp = dwdr / r * vfrac;
        for (int k2 = 0; k2 < dim; ++k2) {
          double gitmp = 0.0;
          for (int k1 = 0; k1 < dim; ++k1) gitmp += G2(G, dim, k1, k2) * rij[k1];
          g[k2] += gitmp * vjtmp * sph_op(antisym, f[i], f[j]);
        }
      }
    }
    for (int k = 0; k < dim; ++k) grad[3 * i + k] = g[k] * alpha;
  }
}

/* FunctorOuterComputeShift (ref: functor_compute_shift.h:48-113): dr_i for
 * fluid particles, filter (Fluid, All); alpha = shift*dt*vmax. */
void orc_compute_shift(const orc_particles *P, double alpha, double shiftcut, double nonfluidweight, double *dr) {
  const int dim = P->dim;
  const double shiftcutsq = shiftcut * shiftcut;
  for (int i = 0; i < P->nlocal; ++i) {
    const int it = P->type[i], ikind = kind_of(P, i);
    dr[3 * i] = dr[3 * i + 1] = dr[3 * i + 2] = 0.0;
    if (!fyes1(ORC_KIND_FLUID, ikind)) continue;
    int cnt = 0;
    double ri = 0.0;
    for (int jj = P->neigh_ptr[i]; jj < P->neigh_ptr[i + 1]; ++jj) {
      const int j = P->neigh_idx[jj], jt = P->type[j], jkind = kind_of(P, j);
      if (!fyes2(ORC_KIND_FLUID, ORC_KIND_ALL, ikind, jkind)) continue;
      double rsq = 0.0;
      for (int k = 0; k < dim; ++k) rsq += pow(P->x[3 * i + k] - P->x[3 * j + k], 2);
      const double c = tab(P, P->cutsq, it, jt), rth = c < shiftcutsq ? c : shiftcutsq;
      if (rsq < rth) { ++cnt; ri += sqrt(rsq); }
    }
    if (cnt) ri /= (double)cnt;
    for (int jj = P->neigh_ptr[i]; jj < P->neigh_ptr[i + 1]; ++jj) {
      const int j = P->neigh_idx[jj], jt = P->type[j], jkind = kind_of(P, j);
      if (!fyes2(ORC_KIND_FLUID, ORC_KIND_ALL, ikind, jkind)) continue;
      double rsq = 0.0, rij[3] = {0, 0, 0};
      for (int k = 0; k < dim; ++k) { rij[k] = P->x[3 * i + k] - P->x[3 * j + k]; rsq += rij[k] * rij[k]; }
      const double c = tab(P, P->cutsq, it, jt), rth = c < shiftcutsq ? c : shiftcutsq;
      if (rsq < rth) {
        const double r = sqrt(rsq) + ORC_EPS, rir2 = pow(ri / r, 2);
        const double beta = alpha / r * rir2 * (1.0 + (!(jkind & ORC_KIND_FLUID)) * nonfluidweight * rir2);
        for (int k = 0; k < dim; ++k) dr[3 * i + k] += beta * rij[k];
      }
    }
  }
}

/* corrected gradients of p and of the dim components of v at row i with the
 * (Fluid, All) filter, geometry from x (functor_gradient.h:78-170 as used by
 * functor_apply_shift.h:60-66) */
static void shift_gradients(const orc_particles *P, int antisym, const double *x, const double *v, const double *p,
                            int i, double gp[3], double gv[3][3]) {
  const int dim = P->dim, d2 = dim * dim;
  double Gi[9] = {0};
  for (int k = 0; k < dim; ++k) G2(Gi, dim, k, k) = 1.0;
  const int it = P->type[i], ikind = kind_of(P, i);
  for (int k = 0; k < 3; ++k) { gp[k] = 0.0; gv[0][k] = gv[1][k] = gv[2][k] = 0.0; }
  if (!fyes1(ORC_KIND_FLUID, ikind)) return;
  const double *G = antisym ? Gi : &P->Gc[(size_t)i * d2];
  for (int jj = P->neigh_ptr[i]; jj < P->neigh_ptr[i + 1]; ++jj) {
    const int j = P->neigh_idx[jj], jt = P->type[j], jkind = kind_of(P, j);
    if (!fyes2(ORC_KIND_FLUID, ORC_KIND_ALL, ikind, jkind)) continue;
    double rsq = 0.0, rij[3] = {0, 0, 0};
    for (int k = 0; k < dim; ++k) { rij[k] = x[3 * i + k] - x[3 * j + k]; rsq += rij[k] * rij[k]; }
    if (rsq < tab(P, P->cutsq, it, jt)) {
      const double r = sqrt(rsq) + ORC_EPS;
      const double dwdr = orc_kernel_dval(P->kernel, dim, r, tab(P, P->h, it, jt));
      const double vfrac = antisym ? sqrt(P->vfrac[i] * P->vfrac[j]) : P->vfrac[j];
      const double vjtmp = dwdr / r * vfrac;
      for (int k2 = 0; k2 < dim; ++k2) {
        double gitmp = 0.0;
        for (int k1 = 0; k1 < dim; ++k1) gitmp += G2(G, dim, k1, k2) * rij[k1];
        const double ijtmp = gitmp * vjtmp;
        gp[k2] += ijtmp * sph_op(antisym, p[i], p[j]);
        for (int k1 = 0; k1 < dim; ++k1) gv[k1][k2] += ijtmp * sph_op(antisym, v[3 * i + k1], v[3 * j + k1]);
      }
    }
  }
}

/* FunctorOuterApplyShift (ref: functor_apply_shift.h:76-108) */
void orc_apply_shift(const orc_particles *P, int antisym, const int *fixed, const double *dr, double *x, double *v,
                     double *p, int sequential) {
  const int dim = P->dim, n = P->nlocal;
  double *x0 = x, *v0 = v, *p0 = p;
  if (!sequential) {  /* every row reads the pre-shift state */
    x0 = (double *)malloc(sizeof(double) * 3 * (size_t)P->nall);
    v0 = (double *)malloc(sizeof(double) * 3 * (size_t)P->nall);
    p0 = (double *)malloc(sizeof(double) * (size_t)P->nall);
    memcpy(x0, x, sizeof(double) * 3 * (size_t)P->nall);
    memcpy(v0, v, sizeof(double) * 3 * (size_t)P->nall);
    memcpy(p0, p, sizeof(double) * (size_t)P->nall);
  }
  for (int i = 0; i < n; ++i) {
    if (fixed && fixed[P->type[i]]) continue;
    double gp[3], gv[3][3];
    shift_gradients(P, antisym, x0, v0, p0, i, gp, gv);
    double s = 0.0;
    for (int k = 0; k < dim; ++k) s += gp[k] * dr[3 * i + k];
    p[i] += s;
    for (int k = 0; k < dim; ++k) {
      double t = 0.0;
      for (int q = 0; q < dim; ++q) t += gv[k][q] * dr[3 * i + q];
      v[3 * i + k] += t;
    }
    for (int k = 0; k < dim; ++k) x[3 * i + k] += dr[3 * i + k];
  }
  if (!sequential) { free(x0); free(v0); free(p0); }
}

/* Matrix-free application of the Laplacian rows to a field with ncomp
 * components: lap_i = sum_j A_ij f_j using exactly the row values of
 * orc_laplacian_matrix (the Helmholtz functor forms w = A v through the
 * assembled matrix, ref: functor_incomp_navier_stokes_helmholtz.h:88-94).
 * f is [nall][ncomp] (ghost values present), lap [nlocal][ncomp]. */
void orc_laplacian_apply(const orc_particles *P, int antisym, const double *f,
                         int ncomp, double alpha, const double *material,
                         int filt_i, int filt_j, double *lap) {
  /* assemble into a private graph keyed by particle index (not colmap) so
   * ghost copies keep their own f values */
  const int n = P->nlocal;
  int *rp = (int *)malloc(sizeof(int) * (size_t)(n + 1));
  const int cap = P->neigh_ptr[n] + n;
  int *ci = (int *)malloc(sizeof(int) * (size_t)cap);
  double *v = (double *)calloc((size_t)cap, sizeof(double));
  orc_particles Q = *P;
  int *ident = (int *)malloc(sizeof(int) * (size_t)P->nall);
  for (int j = 0; j < P->nall; ++j) ident[j] = j;
  Q.colmap = ident;
  orc_graph(&Q, rp, ci, cap);
  orc_laplacian_matrix(&Q, antisym, alpha, material, filt_i, filt_j, 0, rp, ci, v);
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < ncomp; ++c) {
      double s = 0.0;
      for (int p = rp[i]; p < rp[i + 1]; ++p) s += v[p] * f[(size_t)ci[p] * ncomp + c];
      lap[(size_t)i * ncomp + c] = s;
    }
  free(rp); free(ci); free(v); free(ident);
}

/* ===================================================================== *
 *  Poisson system builder
 * ===================================================================== */

/* ref: functor_incomp_navier_stokes_poisson.h:52-181 and
 * PairISPH::modifySingularMatrix, pair_isph.cpp:493-520.
 * val must be zero-initialised by the caller or is zeroed here (PutScalar(0)).
 * normal may be NULL (no wall normals).  With normals the Solid rows receive the
 * homogeneous-Neumann operator -dt n.grad (functor_gradient_dot_operator_matrix.h). */
int orc_poisson(const orc_particles *P, int antisym, int morris_holmes,
                double dt, const double *rho, const double *vstar,
                const double *normal, double solid_normal_diag, int singular_mode, int is_rank0,
                const int *rowptr, const int *colidx, double *val,
                double *b, double *work) {
  const int n = P->nlocal, dim = P->dim;
  memset(val, 0, sizeof(double) * (size_t)rowptr[n]); /* :61 PutScalar(0) */
  int filt_j, neumann;
  if (singular_mode == ORC_NOT_SINGULAR) { filt_j = ORC_KIND_ALL; neumann = 0; }   /* :73-78 */
  else                                   { filt_j = ORC_KIND_FLUID; neumann = 1; } /* :79-85 */
  for (int i = 0; i < P->nall; ++i) work[i] = 1.0 / rho[i];                          /* :88-91 */
  /* the MorrisHolmes Poisson variant keeps the plain Laplacian and only swaps the
   * divergence's mirror (pair_isph_corrected.cpp:174-178) */
  int bad = orc_laplacian_matrix(P, antisym, -dt, work, ORC_KIND_FLUID, filt_j,
                                 0, rowptr, colidx, val);                           /* :93-96 */
  if (bad) return -1;
  if (neumann && normal != NULL) {
    /* homogeneous Neumann rows on the wall particles (:98-107):
     * FunctorOuterGradientDotOperatorMatrix(normal, alpha=-dt), filter (Solid, All)
     * (ref: functor_gradient_dot_operator_matrix.h:39-79) on top of
     * FunctorOuterGradientOperator (ref: functor_gradient_operator.h:74-169): always G_i and V_j,
     * self entry first. */
    const int d2 = dim * dim;
    int maxn = 0;
    for (int i = 0; i < n; ++i) {
      const int m = P->neigh_ptr[i + 1] - P->neigh_ptr[i] + 1;
      if (m > maxn) maxn = m;
    }
    double *gv = (double *)malloc(sizeof(double) * (size_t)maxn);
    int *gi = (int *)malloc(sizeof(int) * (size_t)maxn);
    for (int i = 0; i < n; ++i) {
      const int it = P->type[i], ikind = kind_of(P, i);
      if (!fyes1(ORC_KIND_SOLID, ikind)) continue;
      const double *G = &P->Gc[(size_t)i * d2];
      int cnt = 0;
      gi[cnt] = P->colmap[i];
      gv[cnt++] = 0.0;
      for (int jj = P->neigh_ptr[i]; jj < P->neigh_ptr[i + 1]; ++jj) {
        const int j = P->neigh_idx[jj];
        const int jt = P->type[j];
        double rsq = 0.0, rij[3] = {0, 0, 0};
        for (int k = 0; k < dim; ++k) {
          rij[k] = P->x[3 * i + k] - P->x[3 * j + k];
          rsq += rij[k] * rij[k];
        }
        if (rsq < tab(P, P->cutsq, it, jt)) {
          const double r = sqrt(rsq) + ORC_EPS;
          const double dwdr = orc_kernel_dval(P->kernel, dim, r, tab(P, P->h, it, jt));
          const double vjtmp = dwdr / r * P->vfrac[j];   /* coeff = 1: particle i is solid */
          double v = 0.0;
          for (int k2 = 0; k2 < dim; ++k2) {
            double gitmp = 0.0;
            for (int k1 = 0; k1 < dim; ++k1) gitmp += G2(G, dim, k1, k2) * rij[k1];
            v += (gitmp * vjtmp) * (-dt) * normal[3 * i + k2];
          }
          gi[cnt] = P->colmap[j];
          gv[cnt++] = v;
          gv[0] -= v;
        }
      }
      if (sum_into(rowptr, colidx, val, i, cnt, gv, gi)) { free(gv); free(gi); return -1; }
    }
    free(gv);
    free(gi);
  }
  /* diag bookkeeping: ExtractDiagonalCopy ... ReplaceDiagonalValues (:109,:179-181) */
  orc_divergence(P, antisym, vstar, 1.0, 1, ORC_KIND_FLUID, ORC_KIND_ALL, morris_holmes, b); /* :113-116 */
  int once = 0;
  for (int i = 0; i < n; ++i) {
    const int ikind = kind_of(P, i);
    const int pd = row_find(colidx, rowptr[i], rowptr[i + 1], P->colmap[i]);
    double diag = val[pd];
    if (ikind == ORC_KIND_SOLID) { /* :137-147 */
      if (neumann) {
        double nn = 0.0;
        if (normal) for (int k = 0; k < dim; ++k) nn += normal[3 * i + k] * normal[3 * i + k];
        /* with a wall normal the functor leaves A.diagonal[i] untouched: the row keeps whatever
         * the vector held before (1 after the scalar Helmholtz pass of the same step, 0 when only
         * block matrices were built; pair_isph.cpp:1269, functor_incomp_navier_stokes_helmholtz.h:114-117) */
        diag = nn < 0.5 ? 1.0 : solid_normal_diag;
      } else diag = 1.0;
      b[i] = 0.0;
    } else { /* Fluid / buffers, :150-164 */
      b[i] = -b[i];
      if (is_rank0 && !once) { /* modifySingularMatrix */
        if (singular_mode == ORC_PINZERO) {
          for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) val[p] = 0.0;
          diag = -1.0;
          b[i] = 0.0;
        } else if (singular_mode == ORC_DOUBLEDIAG) diag *= 1.5;
        once = 1;
      }
    }
    val[pd] = diag;
  }
  return 0;
}

/* ===================================================================== *
 *  Helmholtz system builder
 * ===================================================================== */

/* ref: functor_incomp_navier_stokes_helmholtz.h:52-159.
 *   A = Laplacian(dt, mu = nu rho), filter (Fluid, All)      (:64-79)
 *   A <- diag(1/rho) A                                       (:81-85)
 *   w = (1-theta) A v                                        (:87-94)
 *   A <- -theta A ; diag_i = 1 + A_ii (fluid) | 1 (solid)    (:96-98,:114-121)
 *   b_i = v_i + w_i + dt (f_i/rho_i + g) - dt/rho_i grad p_i (:124-135)
 * vall [nall][3] are the velocities (ghosts included; the reference reads the
 * ghost values through Epetra's import of b).  b is column-major [lda x dim]
 * and holds v^n of the owned particles on entry. */
int orc_helmholtz(const orc_particles *P, int antisym, int morris_holmes, double dt, double theta,
                  const double *nu, const double *rho, const double *p,
                  const double *f, const double *g, int incremental_pressure,
                  const double *vall, const int *rowptr, const int *colidx,
                  double *val, double *b, int lda, double *work) {
  const int n = P->nlocal, dim = P->dim;
  memset(val, 0, sizeof(double) * (size_t)rowptr[n]);
  for (int i = 0; i < P->nall; ++i) work[i] = nu[i] * rho[i];
  /* MorrisHolmes variant: the Laplacian carries the mirror, the pressure gradient does not
   * (pair_isph_corrected.cpp:157-161) */
  if (orc_laplacian_matrix(P, antisym, dt, work, ORC_KIND_FLUID, ORC_KIND_ALL, morris_holmes, rowptr, colidx, val)) return -1;
  int ncol = 0;
  for (int j = 0; j < P->nall; ++j) if (P->colmap[j] + 1 > ncol) ncol = P->colmap[j] + 1;
  double *vext = (double *)calloc((size_t)ncol * 3, sizeof(double));
  for (int j = 0; j < P->nall; ++j)
    for (int k = 0; k < 3; ++k) vext[(size_t)P->colmap[j] * 3 + k] = vall[(size_t)j * 3 + k];
  double *grad = (double *)calloc((size_t)n * 3, sizeof(double));
  if (incremental_pressure) orc_gradient(P, antisym, p, 1.0, 1, ORC_KIND_FLUID, ORC_KIND_FLUID, grad);
  for (int i = 0; i < n; ++i) {
    const int ikind = kind_of(P, i);
    const double invrho = 1.0 / rho[i];
    double w[3] = {0, 0, 0};
    int pd = -1;
    for (int q = rowptr[i]; q < rowptr[i + 1]; ++q) {
      val[q] *= invrho;                                   /* LeftScale */
      for (int k = 0; k < dim; ++k) w[k] += val[q] * vext[(size_t)colidx[q] * 3 + k];
      val[q] *= -theta;                                   /* Scale(-theta) */
      if (colidx[q] == P->colmap[i]) pd = q;
    }
    if (pd < 0) { free(vext); free(grad); return -1; }
    if (ikind == ORC_KIND_SOLID) {
      val[pd] = 1.0;
    } else {
      val[pd] = 1.0 + val[pd];
      for (int k = 0; k < dim; ++k) {
        double *bk = &b[(size_t)k * lda + i];
        *bk += w[k] * (1.0 - theta);
        *bk += dt * (f[(size_t)i * 3 + k] / rho[i] + g[k]);
        if (incremental_pressure) *bk += dt * (-1.0 / rho[i] * grad[(size_t)i * 3 + k]);
      }
    }
  }
  free(vext); free(grad);
  return 0;
}

/* FunctorOuterSoluteTransport, ref: functor_solute_transport.h:47-138 (called by
 * PairISPH_Corrected::computeSoluteTransportSpecies, pair_isph_corrected.cpp:844-861):
 *   A = Laplacian(dt*dcoeff), FilterMatchBinary(Fluid, Fluid - BufferNeumann)           (:60-67)
 *   w = (1-theta) A b  with b = the concentration (Epetra Multiply: ghost columns carry the owners' values)  (:85-87)
 *   A *= -theta                                                                        (:90)
 *   Fluid rows: diag = 1 + A_ii, b += w;  Solid / BufferDirichlet / BufferNeumann rows: diag = 1        (:106-124)
 * conc: [nall]; b: [nlocal], returned. */
int orc_solute_transport(const orc_particles *P, int antisym, double dt, double theta, double dcoeff, const double *conc,
                         const int *rowptr, const int *colidx, double *val, double *b) {
  const int n = P->nlocal;
  memset(val, 0, sizeof(double) * (size_t)rowptr[n]);
  if (orc_laplacian_matrix(P, antisym, dt * dcoeff, NULL, ORC_KIND_FLUID | ORC_FILTER_MATCH,
                           ORC_KIND_FLUID - ORC_KIND_BUFFER_NEUMANN, 0, rowptr, colidx, val)) return -1;
  int ncol = 0;
  for (int j = 0; j < P->nall; ++j) if (P->colmap[j] + 1 > ncol) ncol = P->colmap[j] + 1;
  double *cext = (double *)calloc((size_t)ncol, sizeof(double));
  for (int j = 0; j < P->nall; ++j) cext[P->colmap[j]] = conc[j];
  for (int i = 0; i < n; ++i) {
    const int ikind = kind_of(P, i);
    double w = 0.0;
    int pd = -1;
    for (int q = rowptr[i]; q < rowptr[i + 1]; ++q) {
      w += val[q] * cext[colidx[q]];
      val[q] *= -theta;
      if (colidx[q] == P->colmap[i]) pd = q;
    }
    if (pd < 0) { free(cext); return -1; }
    b[i] = conc[i];
    if (ikind == ORC_KIND_FLUID) {
      val[pd] = 1.0 + val[pd];
      b[i] += w * (1.0 - theta);
    } else if (ikind == ORC_KIND_SOLID || ikind == ORC_KIND_BUFFER_DIRICHLET || ikind == ORC_KIND_BUFFER_NEUMANN) {
      val[pd] = 1.0;
    } else { free(cext); return -2; }   /* "Particle types are not supported" */
  }
  free(cext);
  return 0;
}

/* FunctorOuterAppliedElectricPotential, ref: functor_applied_electric_potential.h:36-98 (called by
 * PairISPH_Corrected::computeAppliedElectricPotential, pair_isph_corrected.cpp:569-620):
 *   A = Laplacian(-1, sigma), FilterMatchBinary(Fluid, Fluid)                                          (:49-58)
 *   b = 0; Solid rows: diag = 1; BufferNeumann / BufferDirichlet rows: diag = 1, b = phi; Fluid rows as assembled (:72-90)
 * sigma: [nall] or NULL; phi: [nall]. */
int orc_applied_potential(const orc_particles *P, int antisym, const double *sigma, const double *phi,
                          const int *rowptr, const int *colidx, double *val, double *b) {
  const int n = P->nlocal;
  memset(val, 0, sizeof(double) * (size_t)rowptr[n]);
  if (orc_laplacian_matrix(P, antisym, -1.0, sigma, ORC_KIND_FLUID | ORC_FILTER_MATCH, ORC_KIND_FLUID, 0, rowptr, colidx, val))
    return -1;
  for (int i = 0; i < n; ++i) {
    const int ikind = kind_of(P, i);
    int pd = -1;
    for (int q = rowptr[i]; q < rowptr[i + 1]; ++q)
      if (colidx[q] == P->colmap[i]) pd = q;
    if (pd < 0) return -1;
    b[i] = 0.0;
    if (ikind == ORC_KIND_SOLID) val[pd] = 1.0;
    else if (ikind == ORC_KIND_BUFFER_DIRICHLET || ikind == ORC_KIND_BUFFER_NEUMANN) { val[pd] = 1.0; b[i] = phi[i]; }
  }
  return 0;
}

/* ref: functor_incomp_navier_stokes_block_helmholtz.h:57-187 with
 * Corrected::FunctorOuterLaplacianMatrix's block branch (functor_laplacian_matrix.h:269-314) and
 * FunctorOuterBoundaryNavierSlip (functor_boundary_navier_slip.h:53-184), restated AS WRITTEN:
 *   for ib = 0: Laplacian(dt, nu), filter (Fluid, Solid), with the wall normals: a fluid row whose own normal
 *     is set (|n_i|^2 > 0.5) is distributed into blocks (ib*, jb) with weights n^_jb n^_ib*, n^ = the normalised
 *     sum of the normals of every row entry (neighbours in the cut + self), ib* = first component with
 *     n^_ib^2 >= 1/dim (last one otherwise); other fluid rows go to block (0,0)                (:71-76, :269-292)
 *   for ib = 1..dim-1: NavierSlip(-beta dt): robin_i = -sum_{j solid} (-beta dt) W'/r V_j / rho_i
 *     (n_i + n_j).(G_i^T r_ij) is added to the DIAGONAL ENTRY of EVERY block (ib', jb') with weight
 *     delta - n^_jb' n^_ib' (n^ = n_i/|n_i|), once per pass, i.e. dim-1 times                 (:77-79, :150-162)
 *   for every ib: Laplacian(dt, nu), filter (Fluid, Fluid) into block (ib, ib)                  (:82-85)
 *   w_ib = (1-theta) sum_jb A(ib,jb) b_ib   -- the SAME column ib of b for every jb, as written (:94-97)
 *   A(ib,jb) *= -theta ; diag(ib,ib) = 1 + A(ib,ib)_ii (fluid) | 1 (solid)                      (:101-104,:143-151)
 *   b_i,ib += w_i,ib + dt (f/rho + g) - dt/rho grad p (incremental pressure)                    (:153-162)
 * Every block is stored on the scalar pattern (rowptr/colidx of orc_graph): vals[(ib*dim+jb)*nnz + q].
 * G_i of the slip term is Gc[i] for BOTH operator families (functor_boundary_navier_slip.h:79).
 * material is nu (not nu*rho) and there is no 1/rho row scaling, unlike the scalar builder. */
int orc_block_helmholtz(const orc_particles *P, int antisym, int morris_holmes, double dt, double theta, double beta,
                        const double *nu, const double *rho, const double *p, const double *f, const double *g,
                        int incremental_pressure, const double *normal, const double *vall,
                        const int *rowptr, const int *colidx, double *vals, double *b, int lda) {
  const int n = P->nlocal, dim = P->dim, d2 = dim * dim;
  const size_t nnz = (size_t)rowptr[n];
  memset(vals, 0, sizeof(double) * nnz * (size_t)d2);
  double *ff = (double *)calloc(nnz, sizeof(double)), *fs = (double *)calloc(nnz, sizeof(double));
  if (orc_laplacian_matrix(P, antisym, dt, nu, ORC_KIND_FLUID, ORC_KIND_FLUID, morris_holmes, rowptr, colidx, ff)) { free(ff); free(fs); return -1; }
  if (orc_laplacian_matrix(P, antisym, dt, nu, ORC_KIND_FLUID, ORC_KIND_SOLID, morris_holmes, rowptr, colidx, fs)) { free(ff); free(fs); return -1; }
  int ncol = 0;
  for (int j = 0; j < P->nall; ++j) if (P->colmap[j] + 1 > ncol) ncol = P->colmap[j] + 1;
  double *vext = (double *)calloc((size_t)ncol * 3, sizeof(double));
  for (int j = 0; j < P->nall; ++j)
    for (int k = 0; k < 3; ++k) vext[(size_t)P->colmap[j] * 3 + k] = vall[(size_t)j * 3 + k];
  double *grad = (double *)calloc((size_t)n * 3, sizeof(double));
  if (incremental_pressure) orc_gradient(P, antisym, p, 1.0, 1, ORC_KIND_FLUID, ORC_KIND_FLUID, grad);
#define BLK(ib, jb) (vals + ((size_t)(ib) * dim + (jb)) * nnz)
  for (int i = 0; i < n; ++i) {
    const int it = P->type[i], ikind = kind_of(P, i);
    const int pd = row_find(colidx, rowptr[i], rowptr[i + 1], P->colmap[i]);
    if (pd >= rowptr[i + 1] || colidx[pd] != P->colmap[i]) { free(ff); free(fs); free(vext); free(grad); return -1; }
    if (!(ikind & ORC_KIND_FLUID)) {  /* solid row: unit diagonal in the diagonal blocks, b untouched */
      for (int ib = 0; ib < dim; ++ib) BLK(ib, ib)[pd] = 1.0;
      continue;
    }
    /* (Fluid,Fluid) rows into every diagonal block */
    for (int ib = 0; ib < dim; ++ib)
      for (int q = rowptr[i]; q < rowptr[i + 1]; ++q) BLK(ib, ib)[q] += ff[q];
    /* (Fluid,Solid) row: plain or distributed by the wall normal */
    const double *ni = normal ? &normal[3 * (size_t)i] : NULL;
    double nn = 0.0;
    if (ni) for (int k = 0; k < dim; ++k) nn += ni[k] * ni[k];
    if (!(ni && nn > 0.5)) {
      for (int q = rowptr[i]; q < rowptr[i + 1]; ++q) BLK(0, 0)[q] += fs[q];
    } else {
      double nh[3] = {0, 0, 0};
      for (int jj = P->neigh_ptr[i]; jj < P->neigh_ptr[i + 1]; ++jj) {
        const int j = P->neigh_idx[jj];
        double rsq = 0.0;
        for (int k = 0; k < dim; ++k) { const double d = P->x[3 * i + k] - P->x[3 * j + k]; rsq += d * d; }
        if (rsq < tab(P, P->cutsq, it, P->type[j]))
          for (int k = 0; k < dim; ++k) nh[k] += normal[3 * (size_t)j + k];
      }
      for (int k = 0; k < dim; ++k) nh[k] += ni[k];
      double norm = 0.0;
      for (int k = 0; k < dim; ++k) norm += nh[k] * nh[k];
      norm = sqrt(norm);
      for (int k = 0; k < dim; ++k) nh[k] /= norm;
      int ibp = 0;
      for (; ibp < dim - 1 && (nh[ibp] * nh[ibp] < 1.0 / dim); ++ibp);
      for (int jb = 0; jb < dim; ++jb)
        for (int q = rowptr[i]; q < rowptr[i + 1]; ++q) BLK(ibp, jb)[q] += fs[q] * nh[jb] * nh[ibp];
    }
    /* Navier slip: dim-1 passes, each adds the robin term to the diagonal entry of every block */
    if (dim > 1 && normal) {
      const double *G = &P->Gc[(size_t)i * d2];
      double robin = 0.0;
      for (int jj = P->neigh_ptr[i]; jj < P->neigh_ptr[i + 1]; ++jj) {
        const int j = P->neigh_idx[jj], jt = P->type[j];
        if (kind_of(P, j) != ORC_KIND_SOLID) continue;
        double rsq = 0.0, rij[3] = {0, 0, 0};
        for (int k = 0; k < dim; ++k) { rij[k] = P->x[3 * i + k] - P->x[3 * j + k]; rsq += rij[k] * rij[k]; }
        if (rsq < tab(P, P->cutsq, it, jt)) {
          const double r = sqrt(rsq) + ORC_EPS;
          const double dwdr = orc_kernel_dval(P->kernel, dim, r, tab(P, P->h, it, jt));
          double aij[3] = {0, 0, 0};
          for (int k2 = 0; k2 < dim; ++k2)
            for (int k1 = 0; k1 < dim; ++k1) aij[k2] += G2(G, dim, k1, k2) * rij[k1];
          double tmp = 0.0;
          for (int k = 0; k < dim; ++k) tmp += (normal[3 * (size_t)i + k] + normal[3 * (size_t)j + k]) * aij[k];
          robin -= (-beta * dt) * dwdr / r * P->vfrac[j] / rho[i] * tmp;
        }
      }
      double nh[3] = {0, 0, 0}, norm = 0.0;
      for (int k = 0; k < dim; ++k) { nh[k] = normal[3 * (size_t)i + k]; norm += nh[k] * nh[k]; }
      norm = sqrt(norm);
      if (norm != 0) for (int k = 0; k < dim; ++k) nh[k] /= norm;
      for (int pass = 1; pass < dim; ++pass)
        for (int ib = 0; ib < dim; ++ib)
          for (int jb = 0; jb < dim; ++jb) BLK(ib, jb)[pd] += robin * ((double)(ib == jb) - nh[jb] * nh[ib]);
    }
    /* w, theta scaling, diagonal, right-hand side */
    for (int ib = 0; ib < dim; ++ib) {
      double w = 0.0;
      for (int jb = 0; jb < dim; ++jb)
        for (int q = rowptr[i]; q < rowptr[i + 1]; ++q) {
          w += BLK(ib, jb)[q] * vext[(size_t)colidx[q] * 3 + ib];
          BLK(ib, jb)[q] *= -theta;
        }
      BLK(ib, ib)[pd] = 1.0 + BLK(ib, ib)[pd];
      double *bk = &b[(size_t)ib * lda + i];
      *bk += w * (1.0 - theta);
      *bk += dt * (f[(size_t)i * 3 + ib] / rho[i] + g[ib]);
      if (incremental_pressure) *bk += dt * (-1.0 / rho[i] * grad[(size_t)i * 3 + ib]);
    }
  }
#undef BLK
  free(ff); free(fs); free(vext); free(grad);
  return 0;
}

/* ===================================================================== *
 *  linear algebra: Epetra / Belos / Ifpack semantics
 * ===================================================================== */

void orc_spmv(int n, const int *rowptr, const int *colidx, const double *val,
              const double *x, double *y) {
#pragma omp parallel for schedule(static) if (n > 16384)
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
    for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) s += val[p] * x[colidx[p]];
    y[i] = s;
  }
}

static double vdot(int n, const double *a, const double *b) {
  double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static) if (n > 16384)
  for (int i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}
static void vaxpy(int n, double a, const double *x, double *y) {
#pragma omp parallel for schedule(static) if (n > 16384)
  for (int i = 0; i < n; ++i) y[i] += a * x[i];
}
static void vscale_copy(int n, double a, const double *x, double *y) {
#pragma omp parallel for schedule(static) if (n > 16384)
  for (int i = 0; i < n; ++i) y[i] = a * x[i];
}

/* SolverLin::createNullVector, ref: solver_lin.cpp:59-77 */
void orc_null_vector(int n, const int *mask, double *nvec) {
  for (int i = 0; i < n; ++i) nvec[i] = mask ? (double)mask[i] : 1.0;
  const double nrm = sqrt(vdot(n, nvec, nvec));
  for (int i = 0; i < n; ++i) nvec[i] /= nrm;
}

/* -------- ILU(k), block-Jacobi == Ifpack AdditiveSchwarz<ILU>, overlap 0 ---
 * ref: precond_ifpack.h:28-75 (parameters + create()); algorithm = the
 * level-of-fill ILU(k) of Ifpack_ILU (relax 0, athresh 0, rthresh 1):
 * symbolic  lev(i,j) = min_k lev(i,k)+lev(k,j)+1 <= k ; numeric IKJ.
 * Storage here: one CSR holding strict-L (unit diagonal implied), D and
 * strict-U, columns sorted; entries outside a row's block are dropped. */
struct orc_ilu {
  int n, nblocks;
  int *block_ptr;
  int *rowptr, *colidx, *diag;
  double *val;
};

static void ilu_block(orc_ilu *F, int b, const int *arp, const int *aci, const double *av,
                      int lof, int **cols_out, double **vals_out, int *cnt_out) {
  const int lo = F->block_ptr[b], hi = F->block_ptr[b + 1], m = hi - lo;
  /* per-row dynamic pattern with levels */
  int **rcols = (int **)malloc(sizeof(int *) * (size_t)m);
  int **rlevs = (int **)malloc(sizeof(int *) * (size_t)m);
  double **rvals = (double **)malloc(sizeof(double *) * (size_t)m);
  int *rcnt = (int *)malloc(sizeof(int) * (size_t)m);
  int *rdiag = (int *)malloc(sizeof(int) * (size_t)m);
  int *lev = (int *)malloc(sizeof(int) * (size_t)m);
  int *next = (int *)malloc(sizeof(int) * (size_t)(m + 1));
  double *w = (double *)calloc((size_t)m, sizeof(double));
  for (int c = 0; c < m; ++c) lev[c] = -1;
  for (int r = 0; r < m; ++r) {
    const int i = lo + r;
    /* linked list of columns in increasing order; head = next[m] */
    int head = m, count = 0;
    {
      int prev = m;
      next[m] = m;
      /* A's columns are sorted: append in order */
      int have_diag = 0;
      for (int p = arp[i]; p < arp[i + 1]; ++p) {
        const int c = aci[p] - lo;
        if (c < 0 || c >= m) continue;
        if (c == r) have_diag = 1;
        if (!have_diag && c > r) { /* structurally missing diagonal: insert */
          lev[r] = 0; w[r] = 0.0; next[prev] = r; next[r] = m; prev = r; ++count; have_diag = 1;
        }
        lev[c] = 0; w[c] = av[p];
        next[prev] = c; next[c] = m; prev = c; ++count;
      }
      if (!have_diag) { lev[r] = 0; w[r] = 0.0; next[prev] = r; next[r] = m; ++count; }
      head = next[m];
    }
    /* Symbolic pass (Ifpack_IlukGraph::ConstructFilledGraph): merge the level patterns of the rows k < r in
     * increasing order; the numeric pass below then works on the FINAL pattern of the row, as Ifpack_ILU::Compute
     * does -- a pivot's update of an entry counts even when that entry only entered the pattern through a later
     * pivot (a one-pass factorisation with dynamic insertion would drop those updates). */
    for (int k = head; k < r; k = next[k]) {
      int pos = k; /* insertion cursor in linked list */
      for (int q = rdiag[k] + 1; q < rcnt[k]; ++q) {
        const int j = rcols[k][q];
        const int newlev = lev[k] + rlevs[k][q] + 1;
        if (lev[j] >= 0) {
          if (newlev < lev[j]) lev[j] = newlev;
        } else if (newlev <= lof) {
          while (next[pos] < j) pos = next[pos];
          next[j] = next[pos]; next[pos] = j;
          lev[j] = newlev; w[j] = 0.0; ++count;
        }
      }
    }
    /* numeric pass: IKJ on the fixed pattern */
    for (int k = head; k < r; k = next[k]) {
      const double lik = w[k] / rvals[k][rdiag[k]];
      w[k] = lik;
      for (int q = rdiag[k] + 1; q < rcnt[k]; ++q) {
        const int j = rcols[k][q];
        if (lev[j] >= 0) w[j] -= lik * rvals[k][q];
      }
    }
    rcols[r] = (int *)malloc(sizeof(int) * (size_t)count);
    rlevs[r] = (int *)malloc(sizeof(int) * (size_t)count);
    rvals[r] = (double *)malloc(sizeof(double) * (size_t)count);
    int q = 0;
    for (int c = next[m]; c < m; c = next[c]) {
      rcols[r][q] = c; rlevs[r][q] = lev[c]; rvals[r][q] = w[c];
      if (c == r) rdiag[r] = q;
      ++q;
    }
    rcnt[r] = q;
    for (int c = next[m]; c < m;) { const int nx = next[c]; lev[c] = -1; w[c] = 0.0; c = nx; }
  }
  int total = 0;
  for (int r = 0; r < m; ++r) total += rcnt[r];
  int *cols = (int *)malloc(sizeof(int) * (size_t)(total > 0 ? total : 1));
  double *vals = (double *)malloc(sizeof(double) * (size_t)(total > 0 ? total : 1));
  int q = 0;
  for (int r = 0; r < m; ++r) {
    cnt_out[lo + r] = rcnt[r];
    for (int k = 0; k < rcnt[r]; ++k) { cols[q] = rcols[r][k] + lo; vals[q] = rvals[r][k]; ++q; }
    free(rcols[r]); free(rlevs[r]); free(rvals[r]);
  }
  cols_out[b] = cols; vals_out[b] = vals;
  free(rcols); free(rlevs); free(rvals); free(rcnt); free(rdiag); free(lev); free(next); free(w);
}

orc_ilu *orc_ilu_create(int n, const int *rowptr, const int *colidx,
                        const double *val, int level_of_fill,
                        int nblocks, const int *block_ptr) {
  orc_ilu *F = (orc_ilu *)calloc(1, sizeof(orc_ilu));
  F->n = n;
  F->nblocks = nblocks > 0 ? nblocks : 1;
  F->block_ptr = (int *)malloc(sizeof(int) * (size_t)(F->nblocks + 1));
  if (nblocks > 0 && block_ptr) memcpy(F->block_ptr, block_ptr, sizeof(int) * (size_t)(nblocks + 1));
  else { F->block_ptr[0] = 0; F->block_ptr[1] = n; }
  int **bc = (int **)calloc((size_t)F->nblocks, sizeof(int *));
  double **bv = (double **)calloc((size_t)F->nblocks, sizeof(double *));
  int *cnt = (int *)calloc((size_t)n, sizeof(int));
#pragma omp parallel for schedule(dynamic, 1)
  for (int b = 0; b < F->nblocks; ++b)
    ilu_block(F, b, rowptr, colidx, val, level_of_fill, bc, bv, cnt);
  F->rowptr = (int *)malloc(sizeof(int) * (size_t)(n + 1));
  F->rowptr[0] = 0;
  for (int i = 0; i < n; ++i) F->rowptr[i + 1] = F->rowptr[i] + cnt[i];
  F->colidx = (int *)malloc(sizeof(int) * (size_t)(F->rowptr[n] > 0 ? F->rowptr[n] : 1));
  F->val = (double *)malloc(sizeof(double) * (size_t)(F->rowptr[n] > 0 ? F->rowptr[n] : 1));
  F->diag = (int *)malloc(sizeof(int) * (size_t)n);
#pragma omp parallel for schedule(dynamic, 1)
  for (int b = 0; b < F->nblocks; ++b) {
    const int lo = F->block_ptr[b], hi = F->block_ptr[b + 1];
    const int base = F->rowptr[lo], len = F->rowptr[hi] - base;
    memcpy(F->colidx + base, bc[b], sizeof(int) * (size_t)len);
    memcpy(F->val + base, bv[b], sizeof(double) * (size_t)len);
    free(bc[b]); free(bv[b]);
    for (int i = lo; i < hi; ++i)
      for (int p = F->rowptr[i]; p < F->rowptr[i + 1]; ++p)
        if (F->colidx[p] == i) F->diag[i] = p;
  }
  /* Ifpack_ILU inverts the diagonal and clamps |D| below Epetra_MinDouble;
   * we keep D itself and divide at apply time (same arithmetic up to 1 ulp). */
  free(bc); free(bv); free(cnt);
  return F;
}

/* z = U^-1 D^-1 L^-1 r, block by block */
void orc_ilu_apply(const orc_ilu *F, const double *r, double *z) {
#pragma omp parallel for schedule(dynamic, 1) if (F->n > 16384)
  for (int b = 0; b < F->nblocks; ++b) {
    const int lo = F->block_ptr[b], hi = F->block_ptr[b + 1];
    for (int i = lo; i < hi; ++i) {
      double s = r[i];
      for (int p = F->rowptr[i]; p < F->diag[i]; ++p) s -= F->val[p] * z[F->colidx[p]];
      z[i] = s;
    }
    for (int i = hi - 1; i >= lo; --i) {
      double s = z[i];
      for (int p = F->diag[i] + 1; p < F->rowptr[i + 1]; ++p) s -= F->val[p] * z[F->colidx[p]];
      z[i] = s / F->val[F->diag[i]];
    }
  }
}

int orc_ilu_nnz(const orc_ilu *F) { return F->rowptr[F->n]; }
void orc_ilu_export(const orc_ilu *F, int *rowptr, int *colidx, double *val) {
  memcpy(rowptr, F->rowptr, sizeof(int) * (size_t)(F->n + 1));
  memcpy(colidx, F->colidx, sizeof(int) * (size_t)F->rowptr[F->n]);
  memcpy(val, F->val, sizeof(double) * (size_t)F->rowptr[F->n]);
}
void orc_ilu_destroy(orc_ilu *F) {
  if (!F) return;
  free(F->block_ptr); free(F->rowptr); free(F->colidx); free(F->diag); free(F->val); free(F);
}

/* -------- operator / preconditioner application ---------------------- */

typedef struct {
  int n;
  const int *rowptr, *colidx;
  const double *val;
  const double *nvec; /* NULL unless singular */
  int prec_type;
  const orc_ilu *F;
  const orc_amg *G;
  const orc_schwarz *W;
  double *invdiag;
  int ncomp; /* > 1: block-diagonal preconditioner, the same operator on every component (precond_ml.h:138-155) */
} lin_ctx;

/* PoissonProjection::Apply, ref: solver_lin.h:131-140: y = A x; y -= (y.n) n */
static void op_apply(const lin_ctx *c, const double *x, double *y) {
  orc_spmv(c->n, c->rowptr, c->colidx, c->val, x, y);
  if (c->nvec) {
    const double d = vdot(c->n, y, c->nvec);
    vaxpy(c->n, -d, c->nvec, y);
  }
}
static void prec_apply(const lin_ctx *c, const double *r, double *z) {
  if (c->ncomp > 1 && (c->prec_type == 2 || c->prec_type == 3)) {
    const int m = c->n / c->ncomp;
    for (int k = 0; k < c->ncomp; ++k) {
      if (c->prec_type == 2) orc_ilu_apply(c->F, r + (size_t)k * m, z + (size_t)k * m);
      else orc_amg_apply(c->G, r + (size_t)k * m, z + (size_t)k * m);
    }
    return;
  }
  if (c->prec_type == 2 && c->F) orc_ilu_apply(c->F, r, z);
  else if (c->prec_type == 3 && c->G) orc_amg_apply(c->G, r, z);
  else if (c->prec_type == 4 && c->W) orc_schwarz_apply(c->W, r, z);
  else if (c->prec_type == 1) {
#pragma omp parallel for schedule(static) if (c->n > 16384)
    for (int i = 0; i < c->n; ++i) z[i] = r[i] * c->invdiag[i];
  } else memcpy(z, r, sizeof(double) * (size_t)c->n);
}

/* Belos::DGKSOrthoManager / ICGS / IMGS, block size 1.
 * DGKS: one classical Gram-Schmidt pass (h = V^T w as one block product),
 * a second pass only if ||w_new|| < dep_tol ||w_old||, dep_tol = 1/sqrt(2). */
static double orthogonalize(int n, int j, double **V, double *w, double *h, int ortho) {
  double *c = (double *)malloc(sizeof(double) * (size_t)(j + 1));
  for (int k = 0; k <= j; ++k) h[k] = 0.0;
  if (ortho == 2) { /* IMGS, 2 sweeps */
    for (int pass = 0; pass < 2; ++pass)
      for (int k = 0; k <= j; ++k) {
        const double d = vdot(n, V[k], w);
        vaxpy(n, -d, V[k], w);
        h[k] += d;
      }
  } else {
    const double old = sqrt(vdot(n, w, w));
    for (int k = 0; k <= j; ++k) c[k] = vdot(n, V[k], w);
    for (int k = 0; k <= j; ++k) { vaxpy(n, -c[k], V[k], w); h[k] += c[k]; }
    double nw = sqrt(vdot(n, w, w));
    if (ortho == 1 || nw < M_SQRT1_2 * old) {
      for (int k = 0; k <= j; ++k) c[k] = vdot(n, V[k], w);
      for (int k = 0; k <= j; ++k) { vaxpy(n, -c[k], V[k], w); h[k] += c[k]; }
    }
  }
  free(c);
  return sqrt(vdot(n, w, w));
}

/* Belos::BlockGmresSolMgr, block size 1, optional "Flexible Gmres";
 * right preconditioning; convergence = implicit residual / ||r0|| <= tol
 * (flexible / no left preconditioner => no explicit-residual test);
 * restart length "Num Blocks"; <= "Maximum Restarts" restarts and
 * <= "Maximum Iterations" iterations in total.
 * ref: solver_lin_belos.h:161-184,224-264; SURVEY Appendix C. */
static void gmres(const lin_ctx *c, const double *b, double *x,
                  const orc_solver_params *prm, orc_solve_info *info) {
  const int n = c->n, m = prm->num_blocks;
  double **V = (double **)malloc(sizeof(double *) * (size_t)(m + 1));
  double **Z = (double **)malloc(sizeof(double *) * (size_t)m);
  for (int k = 0; k <= m; ++k) V[k] = (double *)malloc(sizeof(double) * (size_t)n);
  for (int k = 0; k < m; ++k) Z[k] = prm->flexible ? (double *)malloc(sizeof(double) * (size_t)n) : NULL;
  double *H = (double *)calloc((size_t)(m + 1) * (size_t)m, sizeof(double)); /* column-major (m+1) x m */
  double *cs = (double *)calloc((size_t)m, sizeof(double)), *sn = (double *)calloc((size_t)m, sizeof(double));
  double *g = (double *)calloc((size_t)(m + 1), sizeof(double)), *y = (double *)calloc((size_t)m, sizeof(double));
  double *w = (double *)malloc(sizeof(double) * (size_t)n), *t = (double *)malloc(sizeof(double) * (size_t)n);

  op_apply(c, x, w);
#pragma omp parallel for schedule(static) if (n > 16384)
  for (int i = 0; i < n; ++i) w[i] = b[i] - w[i];
  double beta = sqrt(vdot(n, w, w));
  const double scale = beta == 0.0 ? 1.0 : beta; /* Belos: zero scale -> 1 */
  info->iters = 0; info->restarts = 0; info->converged = 0;
  info->rel_res_implicit = beta / scale;
  if (beta / scale <= prm->tol) info->converged = 1;

  while (!info->converged && info->iters < prm->max_iters) {
    vscale_copy(n, 1.0 / beta, w, V[0]);
    memset(g, 0, sizeof(double) * (size_t)(m + 1));
    g[0] = beta;
    int j = 0;
    for (; j < m;) {
      const double *zj;
      if (prm->flexible) { prec_apply(c, V[j], Z[j]); zj = Z[j]; }
      else { prec_apply(c, V[j], t); zj = t; }
      op_apply(c, zj, w);
      double *h = &H[(size_t)j * (size_t)(m + 1)];
      h[j + 1] = orthogonalize(n, j, V, w, h, prm->ortho);
      if (h[j + 1] != 0.0) vscale_copy(n, 1.0 / h[j + 1], w, V[j + 1]);
      for (int k = 0; k < j; ++k) { /* previous Givens rotations */
        const double a = cs[k] * h[k] + sn[k] * h[k + 1];
        h[k + 1] = -sn[k] * h[k] + cs[k] * h[k + 1];
        h[k] = a;
      }
      { /* new rotation */
        const double a = h[j], bb = h[j + 1], rr = hypot(a, bb);
        cs[j] = rr == 0.0 ? 1.0 : a / rr;
        sn[j] = rr == 0.0 ? 0.0 : bb / rr;
        h[j] = rr; h[j + 1] = 0.0;
        g[j + 1] = -sn[j] * g[j];
        g[j] = cs[j] * g[j];
      }
      ++j;
      ++info->iters;
      info->rel_res_implicit = fabs(g[j]) / scale;
      if (prm->verbose && (info->iters % 10 == 0))
        printf("[oracle gmres] iter %d  rel res %.3e\n", info->iters, info->rel_res_implicit);
      if (info->rel_res_implicit <= prm->tol) { info->converged = 1; break; }
      if (info->iters >= prm->max_iters) break;
    }
    /* y = R^-1 g ; x += Z y  (or M^-1 V y) */
    for (int k = j - 1; k >= 0; --k) {
      double s = g[k];
      for (int l = k + 1; l < j; ++l) s -= H[(size_t)l * (size_t)(m + 1) + k] * y[l];
      y[k] = s / H[(size_t)k * (size_t)(m + 1) + k];
    }
    if (prm->flexible) {
      for (int k = 0; k < j; ++k) vaxpy(n, y[k], Z[k], x);
    } else {
      memset(w, 0, sizeof(double) * (size_t)n);
      for (int k = 0; k < j; ++k) vaxpy(n, y[k], V[k], w);
      prec_apply(c, w, t);
      vaxpy(n, 1.0, t, x);
    }
    if (info->converged || info->iters >= prm->max_iters) break;
    if (info->restarts >= prm->max_restarts) break;
    ++info->restarts;
    op_apply(c, x, w);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) w[i] = b[i] - w[i];
    beta = sqrt(vdot(n, w, w));
  }
  for (int k = 0; k <= m; ++k) free(V[k]);
  for (int k = 0; k < m; ++k) free(Z[k]);
  free(V); free(Z); free(H); free(cs); free(sn); free(g); free(y); free(w); free(t);
}

/* Belos::BlockCGSolMgr, block size 1 (CGIter): the "right" preconditioner
 * slot is applied as z = M^-1 r.  ref: solver_lin_belos.h:180-181,
 * USER-REAXC-T/solver_lin_belos.h:236-245. */
static void pcg(const lin_ctx *c, const double *b, double *x,
                const orc_solver_params *prm, orc_solve_info *info) {
  const int n = c->n;
  double *r = (double *)malloc(sizeof(double) * (size_t)n), *z = (double *)malloc(sizeof(double) * (size_t)n);
  double *p = (double *)malloc(sizeof(double) * (size_t)n), *ap = (double *)malloc(sizeof(double) * (size_t)n);
  op_apply(c, x, ap);
#pragma omp parallel for schedule(static) if (n > 16384)
  for (int i = 0; i < n; ++i) r[i] = b[i] - ap[i];
  const double r0 = sqrt(vdot(n, r, r));
  const double scale = r0 == 0.0 ? 1.0 : r0;
  info->iters = 0; info->restarts = 0;
  info->rel_res_implicit = r0 / scale;
  info->converged = (r0 / scale <= prm->tol);
  prec_apply(c, r, z);
  memcpy(p, z, sizeof(double) * (size_t)n);
  double rz = vdot(n, r, z);
  while (!info->converged && info->iters < prm->max_iters) {
    op_apply(c, p, ap);
    const double pap = vdot(n, p, ap);
    const double alpha = rz / pap;
    vaxpy(n, alpha, p, x);
    vaxpy(n, -alpha, ap, r);
    ++info->iters;
    info->rel_res_implicit = sqrt(vdot(n, r, r)) / scale;
    if (info->rel_res_implicit <= prm->tol) { info->converged = 1; break; }
    prec_apply(c, r, z);
    const double rz_new = vdot(n, r, z);
    const double betak = rz_new / rz;
    rz = rz_new;
#pragma omp parallel for schedule(static) if (n > 16384)
    for (int i = 0; i < n; ++i) p[i] = z[i] + betak * p[i];
  }
  free(r); free(z); free(p); free(ap);
}

/* SolverLin_Belos::solveProblem, ref: solver_lin_belos.h:130-222.
 *  1. singular: n = mask/||mask||; b -= (b.n) n               (:138-144)
 *  2. preconditioner built from the unprojected A              (:147-156)
 *  3. operator = PoissonProjection(A,n) if singular            (:161-167)
 *  4. right preconditioning, solver by "Solver Type"           (:168-184)
 *  5. x -= (x.n) n ; non-convergence is reported, not raised   (:192-219)
 * b is modified in place exactly as the reference modifies *_b. */
static int solve_impl(int n, const int *rowptr, const int *colidx, const double *val,
                      double *b, double *x, int is_singular, const int *null_mask,
                      int prec_type, const void *prec_obj, int ncomp,
                      const orc_solver_params *prm, orc_solve_info *info);

int orc_solve(int n, const int *rowptr, const int *colidx, const double *val,
              double *b, double *x, int is_singular, const int *null_mask,
              int prec_type, const void *prec_obj,
              const orc_solver_params *prm, orc_solve_info *info) {
  return solve_impl(n, rowptr, colidx, val, b, x, is_singular, null_mask, prec_type, prec_obj, 1, prm, info);
}

/* SolverLin_Belos::solveBlockProblem, ref: solver_lin_belos.h:53-128: the dim x dim blocked operator is given
 * here as one CSR over the product vector [x_0; ..; x_{dim-1}] (n = dim * nlocal rows); the preconditioner
 * object (built for one nlocal x nlocal block) is applied to every component; singular systems are refused
 * like the reference does (:60-61). */
int orc_solve_block(int n, int dim, const int *rowptr, const int *colidx, const double *val,
                    double *b, double *x, int prec_type, const void *prec_obj,
                    const orc_solver_params *prm, orc_solve_info *info) {
  return solve_impl(n, rowptr, colidx, val, b, x, 0, NULL, prec_type, prec_obj, dim, prm, info);
}

static int solve_impl(int n, const int *rowptr, const int *colidx, const double *val,
                      double *b, double *x, int is_singular, const int *null_mask,
                      int prec_type, const void *prec_obj, int ncomp,
                      const orc_solver_params *prm, orc_solve_info *info) {
  const double t0 = now_sec();
  lin_ctx c;
  memset(&c, 0, sizeof(c));
  c.ncomp = ncomp;
  c.n = n; c.rowptr = rowptr; c.colidx = colidx; c.val = val;
  c.prec_type = prec_type;
  c.F = prec_type == 2 ? (const orc_ilu *)prec_obj : NULL;
  c.G = prec_type == 3 ? (const orc_amg *)prec_obj : NULL;
  c.W = prec_type == 4 ? (const orc_schwarz *)prec_obj : NULL;
  double *nvec = NULL;
  if (is_singular) {
    nvec = (double *)malloc(sizeof(double) * (size_t)n);
    orc_null_vector(n, null_mask, nvec);
    const double d = vdot(n, b, nvec);
    vaxpy(n, -d, nvec, b);
    c.nvec = nvec;
  }
  if (prec_type == 1) {
    c.invdiag = (double *)malloc(sizeof(double) * (size_t)n);
    for (int i = 0; i < n; ++i) {
      double d = 1.0;
      for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) if (colidx[p] == i) d = val[p];
      c.invdiag[i] = 1.0 / d;
    }
  }
  if (prm->solver_type == 1) pcg(&c, b, x, prm, info);
  else gmres(&c, b, x, prm, info);
  { /* ||b - A x|| / ||b|| with the unprojected A (:201-212) */
    double *r = (double *)malloc(sizeof(double) * (size_t)n);
    orc_spmv(n, rowptr, colidx, val, x, r);
#pragma omp parallel for schedule(static) if (n > 16384)
    for (int i = 0; i < n; ++i) r[i] = b[i] - r[i];
    const double bn = sqrt(vdot(n, b, b));
    info->rel_res_explicit = sqrt(vdot(n, r, r)) / (bn == 0.0 ? 1.0 : bn);
    free(r);
  }
  if (is_singular) {
    const double d = vdot(n, x, nvec);
    vaxpy(n, -d, nvec, x);
    free(nvec);
  }
  free(c.invdiag);
  info->solve_seconds = now_sec() - t0;
  return 0; /* LAMMPS_SUCCESS even when not converged */
}
