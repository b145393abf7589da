/*
 * isph_amg_oracle.c -- CPU ORACLE (test infrastructure, NOT product code) for the
 * smoothed-aggregation AMG preconditioner that stands in for PrecondWrapper_ML
 * (ref: IMPLICIT-SPH/precond_ml.h:40-171).
 *
 * PARITY PINNING STATUS: UNPINNED.  The arithmetic of the reference lives in
 * Trilinos/ML (un-vendored, no pinned version); the reference holds no numeric
 * vectors at this boundary.  What is restated here is the configuration the
 * wrapper sets (precond_ml.h:44-55,97-127) on top of ML's published
 * smoothed-aggregation method (Vanek, Mandel, Brezina 1996):
 *   "max levels" 5, "aggregation: type" Uncoupled (aggregates never cross the
 *   rank), threshold 0, damping 4/3, "smoother: type" symmetric Gauss-Seidel,
 *   1 sweep, pre and post, V cycle, coarse "Amesos-KLU" (direct) -- or, when a
 *   null vector is injected (setNullVector, :97-127), a one-dimensional
 *   pre-computed null space and the smoother as the coarse solver.
 * Two documented departures, shared with the GPU implementation so that the two
 * can be compared entry by entry:
 *   (1) ML forms uncoupled aggregates with a sequential greedy sweep; here the
 *       roots are a distance-2 maximal independent set found with synchronous
 *       rounds and hashed priorities (Bell, Dalton, Olson 2012) -- the same
 *       "root + all its neighbours" aggregates, order-independent.
 *   (2) ML's Gauss-Seidel is processor-local; here it is local to blocks of
 *       `block` consecutive rows on the fine level and to 64 rows on the coarse ones: x += M_B^-1 (b - A x),
 *       M_B = blockdiag[(D+L_B) D^-1 (D+U_B)], one residual per sweep.
 *   The damping uses rho = ||D^-1 A||_inf ("eigen-analysis: type" Anorm)
 *   instead of ML's default 10 CG-Lanczos steps.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "isph_oracle.h"

#define AMG_MAXLEV 8
#define AMG_DENSE_MAX 2048 /* largest coarsest level the dense LU is formed for (shared with the GPU) */
#define AMG_COARSE_BLOCK 64 /* rows the Gauss-Seidel sweeps are local to on levels >= 1 (shared with the GPU) */

typedef struct { int n, m; int *rp, *ci; double *v; } csr_t; /* n rows, m columns */

struct orc_amg {
  int nlev, block, sweeps, singular;
  int gs_eff; /* 1: "ML Gauss-Seidel" with "smoother: Gauss-Seidel efficient symmetric" (bench-script ml.xml): forward sweeps before the
                 coarse correction, backward sweeps after it, instead of symmetric sweeps on both sides */
  int coarse_smooth; /* coarsest level solved by the smoother: singular, or larger than AMG_DENSE_MAX rows */
  int whole_sgs; /* 1: Gauss-Seidel over the whole level (= ML's processor-local sweep on one rank) */
  csr_t A[AMG_MAXLEV], P[AMG_MAXLEV], R[AMG_MAXLEV];
  int *agg[AMG_MAXLEV];
  double *nv[AMG_MAXLEV];
  double *dinv[AMG_MAXLEV];
  double *lu; int *piv;            /* dense LU of the coarsest level (non-singular case) */
  double *x[AMG_MAXLEV], *b[AMG_MAXLEV], *r[AMG_MAXLEV];
};

static void csr_free(csr_t *a) { free(a->rp); free(a->ci); free(a->v); memset(a, 0, sizeof(*a)); }

static uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
enum { ST_COVERED = 0, ST_UNDECIDED = 1, ST_ROOT = 3 };
static uint64_t mkkey(int state, int i) {
  return ((uint64_t)state << 62) | ((uint64_t)hash32((uint32_t)i) << 30) | (uint64_t)i;
}

/* distance-2 MIS roots + "root and its neighbours" aggregates; returns the number of aggregates */
static double diag_of(const csr_t *A, int i);
/* ML's strength test: a_ij is strong iff a_ij^2 > theta^2 |a_ii a_jj| (theta = "aggregation: threshold", default 0) */
#define STRONG(p, i, j) ((j) < n && (j) != (i) && A->v[p] * A->v[p] > th2 * fabs(dg[i] * dg[j]))
static int aggregate(const csr_t *A, double theta, int *agg) {
  const int n = A->n;
  const double th2 = theta * theta;
  double *dg = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) dg[i] = diag_of(A, i);
  uint64_t *key = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n);
  uint64_t *t1 = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n);
  uint64_t *t2 = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n);
  int undecided = 0;
  for (int i = 0; i < n; ++i) {
    int deg = 0;
    for (int p = A->rp[i]; p < A->rp[i + 1]; ++p) deg += STRONG(p, i, A->ci[p]);
    key[i] = mkkey(deg ? ST_UNDECIDED : ST_COVERED, i);
    undecided += deg != 0;
  }
  while (undecided) {
#pragma omp parallel for schedule(static) if (n > 16384)
    for (int i = 0; i < n; ++i) {
      uint64_t m = key[i];
      for (int p = A->rp[i]; p < A->rp[i + 1]; ++p) { const int j = A->ci[p]; if (STRONG(p, i, j) && key[j] > m) m = key[j]; }
      t1[i] = m;
    }
#pragma omp parallel for schedule(static) if (n > 16384)
    for (int i = 0; i < n; ++i) {
      uint64_t m = t1[i];
      for (int p = A->rp[i]; p < A->rp[i + 1]; ++p) { const int j = A->ci[p]; if (STRONG(p, i, j) && t1[j] > m) m = t1[j]; }
      t2[i] = m;
    }
    undecided = 0;
    for (int i = 0; i < n; ++i) {
      if ((key[i] >> 62) != ST_UNDECIDED) continue;
      if (t2[i] == key[i]) key[i] = mkkey(ST_ROOT, i);
      else if ((t2[i] >> 62) == ST_ROOT) key[i] = mkkey(ST_COVERED, i);
      else ++undecided;
    }
  }
  int nagg = 0;
  for (int i = 0; i < n; ++i) agg[i] = ((key[i] >> 62) == ST_ROOT) ? nagg++ : -1;
  /* pass 1: neighbours of a root */
  int *a1 = (int *)malloc(sizeof(int) * (size_t)n);
  for (int i = 0; i < n; ++i) {
    a1[i] = agg[i];
    if (agg[i] >= 0) continue;
    for (int p = A->rp[i]; p < A->rp[i + 1]; ++p) {
      const int j = A->ci[p];
      if (STRONG(p, i, j) && (key[j] >> 62) == ST_ROOT) { a1[i] = agg[j]; break; }
    }
  }
  /* pass 2: the rest joins the pass-1 neighbour it is most strongly coupled to (ties: smallest index) */
  for (int i = 0; i < n; ++i) {
    agg[i] = a1[i];
    if (a1[i] >= 0) continue;
    double best = -1.0;
    for (int p = A->rp[i]; p < A->rp[i + 1]; ++p) {
      const int j = A->ci[p];
      if (!STRONG(p, i, j) || a1[j] < 0) continue;
      const double w = fabs(A->v[p]);
      if (w > best) { best = w; agg[i] = a1[j]; }
    }
  }
  /* pass 3 (unsymmetric patterns only): leftover connected nodes become singletons */
  for (int i = 0; i < n; ++i) {
    if (agg[i] >= 0) continue;
    int deg = 0;
    for (int p = A->rp[i]; p < A->rp[i + 1]; ++p) deg += STRONG(p, i, A->ci[p]);
    if (deg) agg[i] = nagg++;
  }
  free(key); free(t1); free(t2); free(a1); free(dg);
  return nagg;
}

static double diag_of(const csr_t *A, int i) {
  for (int p = A->rp[i]; p < A->rp[i + 1]; ++p) if (A->ci[p] == i) return A->v[p];
  return 1.0;
}

/* ML's "Uncoupled" aggregation as ML defines it (ML_Aggregate_CoarsenUncoupled; Tuminaro & Tong, "Parallel smoothed
 * aggregation multigrid: aggregation strategies on massively parallel machines", SC 2000, section 3), restated for
 * one rank, natural ordering:
 *   phase 1  sweep the nodes in row order; a node whose strong neighbourhood is still entirely unaggregated becomes a
 *            root and forms an aggregate with ALL its strong neighbours;
 *   phase 2  every node left over that has a strong neighbour in a phase-1 aggregate joins the aggregate it is most
 *            strongly coupled to (ties: the first in row order);
 *   phase 3  what is still left (no aggregated neighbour) is swept in row order: a node forms a new aggregate with its
 *            unaggregated strong neighbours.
 * Isolated nodes (no strong neighbour) stay out of every aggregate, like in the MIS-2 variant (Dirichlet-like rows).
 * This is the sequential algorithm the device's MIS-2 variant departs from; orc_amg_create_ex(aggregation = 1)
 * selects it so that the iteration-count gap between the two can be measured (tests/test_oracle.py). */
static int aggregate_ml_uncoupled(const csr_t *A, double theta, int *agg) {
  const int n = A->n;
  const double th2 = theta * theta;
  double *dg = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) dg[i] = diag_of(A, i);
  for (int i = 0; i < n; ++i) agg[i] = -1;
  int nagg = 0;
  for (int i = 0; i < n; ++i) { /* phase 1 */
    if (agg[i] >= 0) continue;
    int deg = 0, free_nb = 1;
    for (int p = A->rp[i]; p < A->rp[i + 1] && free_nb; ++p) {
      const int j = A->ci[p];
      if (!STRONG(p, i, j)) continue;
      ++deg;
      if (agg[j] >= 0) free_nb = 0;
    }
    if (!deg || !free_nb) continue;
    agg[i] = nagg;
    for (int p = A->rp[i]; p < A->rp[i + 1]; ++p) { const int j = A->ci[p]; if (STRONG(p, i, j)) agg[j] = nagg; }
    ++nagg;
  }
  const int nphase1 = nagg;
  int *a1 = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  memcpy(a1, agg, sizeof(int) * (size_t)n);
  for (int i = 0; i < n; ++i) { /* phase 2: against the phase-1 state only */
    if (a1[i] >= 0) continue;
    double best = -1.0;
    for (int p = A->rp[i]; p < A->rp[i + 1]; ++p) {
      const int j = A->ci[p];
      if (!STRONG(p, i, j) || a1[j] < 0 || a1[j] >= nphase1) continue;
      const double w = fabs(A->v[p]);
      if (w > best) { best = w; agg[i] = a1[j]; }
    }
  }
  for (int i = 0; i < n; ++i) { /* phase 3 */
    if (agg[i] >= 0) continue;
    int deg = 0;
    for (int p = A->rp[i]; p < A->rp[i + 1]; ++p) deg += STRONG(p, i, A->ci[p]);
    if (!deg) continue;
    agg[i] = nagg;
    for (int p = A->rp[i]; p < A->rp[i + 1]; ++p) { const int j = A->ci[p]; if (STRONG(p, i, j) && agg[j] < 0) agg[j] = nagg; }
    ++nagg;
  }
  free(a1); free(dg);
  return nagg;
}

/* P = (I - omega/rho D^-1 A) P_tent, P_tent[i, agg i] = nv_i / |nv restricted to the aggregate| */
static void build_prolongator(const csr_t *A, const int *agg, int nagg, const double *nv, double omega, csr_t *P,
                              double *nvc) {
  const int n = A->n;
  for (int a = 0; a < nagg; ++a) nvc[a] = 0.0;
  for (int i = 0; i < n; ++i) if (agg[i] >= 0) nvc[agg[i]] += nv[i] * nv[i];
  for (int a = 0; a < nagg; ++a) nvc[a] = sqrt(nvc[a]);
  double *pt = (double *)malloc(sizeof(double) * (size_t)n);
  for (int i = 0; i < n; ++i) pt[i] = (agg[i] >= 0 && nvc[agg[i]] > 0.0) ? nv[i] / nvc[agg[i]] : 0.0;
  double rho = 0.0;
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
    for (int p = A->rp[i]; p < A->rp[i + 1]; ++p) if (A->ci[p] < n) s += fabs(A->v[p]);
    const double d = fabs(diag_of(A, i));
    if (d == 0.0) continue; /* empty row of a coarse operator (see the smoother's pivots) */
    s /= d;
    if (s > rho) rho = s;
  }
  const double damp = rho > 0.0 ? omega / rho : 0.0;
  P->n = n; P->m = nagg;
  P->rp = (int *)malloc(sizeof(int) * (size_t)(n + 1));
  int cap = A->rp[n] + n + 1;
  P->ci = (int *)malloc(sizeof(int) * (size_t)cap);
  P->v = (double *)malloc(sizeof(double) * (size_t)cap);
  double *acc = (double *)calloc((size_t)(nagg > 0 ? nagg : 1), sizeof(double));
  int *mark = (int *)malloc(sizeof(int) * (size_t)(nagg > 0 ? nagg : 1));
  for (int a = 0; a < nagg; ++a) mark[a] = -1;
  int nnz = 0;
  for (int i = 0; i < n; ++i) {
    P->rp[i] = nnz;
    const int start = nnz;
    const double di = diag_of(A, i);
    if (agg[i] >= 0) { mark[agg[i]] = i; P->ci[nnz++] = agg[i]; acc[agg[i]] = pt[i]; }
    for (int p = A->rp[i]; p < A->rp[i + 1]; ++p) {
      const int j = A->ci[p];
      if (j >= n || agg[j] < 0) continue;
      const int a = agg[j];
      if (mark[a] != i) { mark[a] = i; P->ci[nnz++] = a; acc[a] = 0.0; }
      acc[a] -= (di != 0.0 ? damp / di : 0.0) * A->v[p] * pt[j];
    }
    /* ascending columns */
    for (int p = start + 1; p < nnz; ++p) {
      const int c = P->ci[p];
      int q = p - 1;
      while (q >= start && P->ci[q] > c) { P->ci[q + 1] = P->ci[q]; --q; }
      P->ci[q + 1] = c;
    }
    for (int p = start; p < nnz; ++p) P->v[p] = acc[P->ci[p]];
  }
  P->rp[n] = nnz;
  free(acc); free(mark); free(pt);
}

static void transpose(const csr_t *P, csr_t *R) {
  const int n = P->n, m = P->m, nnz = P->rp[n];
  R->n = m; R->m = n;
  R->rp = (int *)calloc((size_t)(m + 1), sizeof(int));
  R->ci = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
  R->v = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
  for (int p = 0; p < nnz; ++p) R->rp[P->ci[p] + 1]++;
  for (int a = 0; a < m; ++a) R->rp[a + 1] += R->rp[a];
  int *pos = (int *)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
  for (int a = 0; a < m; ++a) pos[a] = R->rp[a];
  for (int i = 0; i < n; ++i)
    for (int p = P->rp[i]; p < P->rp[i + 1]; ++p) { const int q = pos[P->ci[p]]++; R->ci[q] = i; R->v[q] = P->v[p]; }
  free(pos);
}

/* C = X * Y (Gustavson), columns ascending, x columns >= Y->n ignored */
static void spgemm(const csr_t *X, const csr_t *Y, csr_t *C) {
  const int n = X->n, m = Y->m;
  C->n = n; C->m = m;
  C->rp = (int *)malloc(sizeof(int) * (size_t)(n + 1));
  int cap = 1024, nnz = 0;
  C->ci = (int *)malloc(sizeof(int) * (size_t)cap);
  C->v = (double *)malloc(sizeof(double) * (size_t)cap);
  double *acc = (double *)calloc((size_t)(m > 0 ? m : 1), sizeof(double));
  int *mark = (int *)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
  for (int a = 0; a < m; ++a) mark[a] = -1;
  for (int i = 0; i < n; ++i) {
    C->rp[i] = nnz;
    const int start = nnz;
    for (int p = X->rp[i]; p < X->rp[i + 1]; ++p) {
      const int k = X->ci[p];
      if (k >= Y->n) continue;
      const double xv = X->v[p];
      for (int q = Y->rp[k]; q < Y->rp[k + 1]; ++q) {
        const int c = Y->ci[q];
        if (mark[c] != i) {
          mark[c] = i;
          if (nnz == cap) { cap *= 2; C->ci = (int *)realloc(C->ci, sizeof(int) * (size_t)cap); C->v = (double *)realloc(C->v, sizeof(double) * (size_t)cap); }
          C->ci[nnz++] = c;
          acc[c] = 0.0;
        }
        acc[c] += xv * Y->v[q];
      }
    }
    for (int p = start + 1; p < nnz; ++p) {
      const int c = C->ci[p];
      int q = p - 1;
      while (q >= start && C->ci[q] > c) { C->ci[q + 1] = C->ci[q]; --q; }
      C->ci[q + 1] = c;
    }
    for (int p = start; p < nnz; ++p) C->v[p] = acc[C->ci[p]];
  }
  C->rp[n] = nnz;
  free(acc); free(mark);
}

static void spmv(const csr_t *A, const double *x, double *y) {
#pragma omp parallel for schedule(static) if (A->n > 16384)
  for (int i = 0; i < A->n; ++i) {
    double s = 0.0;
    for (int p = A->rp[i]; p < A->rp[i + 1]; ++p) if (A->ci[p] < A->m) s += A->v[p] * x[A->ci[p]];
    y[i] = s;
  }
}

/* z = M_B^-1 r, M_B = blockdiag[(D+L_B) D^-1 (D+U_B)] over blocks of `block` rows */
static void sgs_solve(const csr_t *A, const double *dinv, int block, const double *r, double *z) {
  const int n = A->n, nb = (n + block - 1) / block;
#pragma omp parallel for schedule(dynamic, 4) if (n > 16384)
  for (int b = 0; b < nb; ++b) {
    const int lo = b * block, hi = lo + block < n ? lo + block : n;
    for (int i = lo; i < hi; ++i) {        /* (D+L) t = r, z holds y = D t */
      double s = r[i];
      for (int p = A->rp[i]; p < A->rp[i + 1]; ++p) {
        const int j = A->ci[p];
        if (j >= lo && j < i) s -= A->v[p] * dinv[j] * z[j];
      }
      z[i] = s;
    }
    for (int i = hi - 1; i >= lo; --i) {   /* (D+U) x = y */
      double s = z[i];
      for (int p = A->rp[i]; p < A->rp[i + 1]; ++p) {
        const int j = A->ci[p];
        if (j > i && j < hi) s -= A->v[p] * z[j];
      }
      z[i] = s * dinv[i];
    }
  }
}

/* one direction only: dir > 0: z = (D+L_B)^-1 r (as y = D (D+L_B)^-1 r, then z = D^-1 y: the order the device works in),
 * dir < 0: z = (D+U_B)^-1 r */
static void gs_solve(const csr_t *A, const double *dinv, int block, const double *r, double *z, int dir) {
  const int n = A->n, nb = (n + block - 1) / block;
#pragma omp parallel for schedule(dynamic, 4) if (n > 16384)
  for (int b = 0; b < nb; ++b) {
    const int lo = b * block, hi = lo + block < n ? lo + block : n;
    if (dir > 0) {
      for (int i = lo; i < hi; ++i) {
        double s = r[i];
        for (int p = A->rp[i]; p < A->rp[i + 1]; ++p) {
          const int j = A->ci[p];
          if (j >= lo && j < i) s -= A->v[p] * dinv[j] * z[j];
        }
        z[i] = s;
      }
      for (int i = lo; i < hi; ++i) z[i] *= dinv[i];
    } else {
      for (int i = hi - 1; i >= lo; --i) {
        double s = r[i];
        for (int p = A->rp[i]; p < A->rp[i + 1]; ++p) {
          const int j = A->ci[p];
          if (j > i && j < hi) s -= A->v[p] * z[j];
        }
        z[i] = s * dinv[i];
      }
    }
  }
}

/* x += M_B^-1 (b - A x); zero_guess: x = M_B^-1 b.  dir: 0 symmetric sweep, +1 forward only, -1 backward only */
static void smooth_dir(const orc_amg *G, int l, const double *b, double *x, int zero_guess, int dir) {
  const csr_t *A = &G->A[l];
  double *r = G->r[l];
  const int blk = G->whole_sgs ? (A->n > 0 ? A->n : 1) : (l == 0 ? G->block : AMG_COARSE_BLOCK);
  if (zero_guess) {
    if (dir == 0) sgs_solve(A, G->dinv[l], blk, b, x); else gs_solve(A, G->dinv[l], blk, b, x, dir);
    return;
  }
  spmv(A, x, r);
  for (int i = 0; i < A->n; ++i) r[i] = b[i] - r[i];
  double *z = (double *)malloc(sizeof(double) * (size_t)A->n);
  if (dir == 0) sgs_solve(A, G->dinv[l], blk, r, z); else gs_solve(A, G->dinv[l], blk, r, z, dir);
  for (int i = 0; i < A->n; ++i) x[i] += z[i];
  free(z);
}

/* x += M_B^-1 (b - A x); zero_guess: x = M_B^-1 b */
static void smooth(const orc_amg *G, int l, const double *b, double *x, int zero_guess) {
  const csr_t *A = &G->A[l];
  double *r = G->r[l];
  const int blk = G->whole_sgs ? (A->n > 0 ? A->n : 1) : (l == 0 ? G->block : AMG_COARSE_BLOCK);
  if (zero_guess) { sgs_solve(A, G->dinv[l], blk, b, x); return; }
  spmv(A, x, r);
  for (int i = 0; i < A->n; ++i) r[i] = b[i] - r[i];
  double *z = (double *)malloc(sizeof(double) * (size_t)A->n);
  sgs_solve(A, G->dinv[l], blk, r, z);
  for (int i = 0; i < A->n; ++i) x[i] += z[i];
  free(z);
}

static void dense_lu(int n, double *a, int *piv) {
  for (int k = 0; k < n; ++k) {
    int p = k;
    for (int i = k + 1; i < n; ++i) if (fabs(a[i * n + k]) > fabs(a[p * n + k])) p = i;
    piv[k] = p;
    if (p != k) for (int j = 0; j < n; ++j) { const double t = a[k * n + j]; a[k * n + j] = a[p * n + j]; a[p * n + j] = t; }
    const double d = a[k * n + k];
    for (int i = k + 1; i < n; ++i) {
      const double f = a[i * n + k] / d;
      a[i * n + k] = f;
      for (int j = k + 1; j < n; ++j) a[i * n + j] -= f * a[k * n + j];
    }
  }
}
static void dense_solve(int n, const double *a, const int *piv, double *x) {
  for (int k = 0; k < n; ++k) { const double t = x[k]; x[k] = x[piv[k]]; x[piv[k]] = t; for (int i = k + 1; i < n; ++i) x[i] -= a[i * n + k] * x[k]; }
  for (int i = n - 1; i >= 0; --i) { double s = x[i]; for (int j = i + 1; j < n; ++j) s -= a[i * n + j] * x[j]; x[i] = s / a[i * n + i]; }
}

orc_amg *orc_amg_create(int n, const int *rowptr, const int *colidx, const double *val, const double *nullvec,
                        int max_levels, int coarse_max, double omega, int block, int sweeps, double theta) {
  return orc_amg_create_ex(n, rowptr, colidx, val, nullvec, max_levels, coarse_max, omega, block, sweeps, theta, 0, 0);
}

/* aggregation: 0 = distance-2 MIS (the device algorithm), 1 = ML's sequential Uncoupled sweep;
 * whole_sgs: 1 = Gauss-Seidel over the whole level (ML on one rank) instead of block-local */
orc_amg *orc_amg_create_ex(int n, const int *rowptr, const int *colidx, const double *val, const double *nullvec,
                           int max_levels, int coarse_max, double omega, int block, int sweeps, double theta,
                           int aggregation, int whole_sgs) {
  orc_amg *G = (orc_amg *)calloc(1, sizeof(orc_amg));
  G->block = block; G->sweeps = sweeps; G->singular = nullvec != NULL; G->whole_sgs = whole_sgs;
  if (max_levels > AMG_MAXLEV) max_levels = AMG_MAXLEV;
  /* level 0: the local square part of A */
  csr_t *A0 = &G->A[0];
  A0->n = A0->m = n;
  A0->rp = (int *)malloc(sizeof(int) * (size_t)(n + 1));
  int nnz = 0;
  for (int i = 0; i < n; ++i) for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) nnz += colidx[p] < n;
  A0->ci = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
  A0->v = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
  nnz = 0;
  for (int i = 0; i < n; ++i) {
    A0->rp[i] = nnz;
    for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) if (colidx[p] < n) { A0->ci[nnz] = colidx[p]; A0->v[nnz++] = val[p]; }
  }
  A0->rp[n] = nnz;
  G->nv[0] = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) G->nv[0][i] = nullvec ? nullvec[i] : 1.0;
  G->nlev = 1;
  while (G->nlev < max_levels) {
    const int l = G->nlev - 1;
    const csr_t *A = &G->A[l];
    if (A->n <= coarse_max) break;
    G->agg[l] = (int *)malloc(sizeof(int) * (size_t)A->n);
    const int nagg = aggregation == 1 ? aggregate_ml_uncoupled(A, theta, G->agg[l]) : aggregate(A, theta, G->agg[l]);
    /* no coarsening, or a coarse space too small to carry anything but the null vector: stop here */
    if (nagg < 8 || nagg >= A->n) { free(G->agg[l]); G->agg[l] = NULL; break; }
    G->nv[l + 1] = (double *)malloc(sizeof(double) * (size_t)nagg);
    build_prolongator(A, G->agg[l], nagg, G->nv[l], omega, &G->P[l], G->nv[l + 1]);
    transpose(&G->P[l], &G->R[l]);
    csr_t AP;
    spgemm(A, &G->P[l], &AP);
    spgemm(&G->R[l], &AP, &G->A[l + 1]);
    csr_free(&AP);
    ++G->nlev;
  }
  for (int l = 0; l < G->nlev; ++l) {
    const size_t m = (size_t)(G->A[l].n > 0 ? G->A[l].n : 1);
    G->x[l] = (double *)calloc(m, sizeof(double));
    G->b[l] = (double *)calloc(m, sizeof(double));
    G->r[l] = (double *)calloc(m, sizeof(double));
    G->dinv[l] = (double *)malloc(sizeof(double) * m);
    /* a coarse unknown whose aggregate carries none of the null vector (a masked null vector: solid rows) has an
     * empty column in P and an empty row in R A P: it stays at zero instead of dividing by its zero pivot */
    for (int i = 0; i < G->A[l].n; ++i) { const double d = diag_of(&G->A[l], i); G->dinv[l][i] = d != 0.0 ? 1.0 / d : 0.0; }
  }
  G->coarse_smooth = G->singular || G->A[G->nlev - 1].n > AMG_DENSE_MAX;
  if (!G->coarse_smooth) {
    const csr_t *A = &G->A[G->nlev - 1];
    const int m = A->n;
    G->lu = (double *)calloc((size_t)m * (size_t)m + 1, sizeof(double));
    G->piv = (int *)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
    for (int i = 0; i < m; ++i) for (int p = A->rp[i]; p < A->rp[i + 1]; ++p) G->lu[(size_t)i * m + A->ci[p]] = A->v[p];
    dense_lu(m, G->lu, G->piv);
  }
  return G;
}

static void vcycle(const orc_amg *G, int l, const double *b, double *x) {
  const csr_t *A = &G->A[l];
  if (l == G->nlev - 1) {
    if (G->coarse_smooth) {
      smooth(G, l, b, x, 1);
      for (int s = 1; s < G->sweeps; ++s) smooth(G, l, b, x, 0);
    } else {
      memcpy(x, b, sizeof(double) * (size_t)A->n);
      dense_solve(A->n, G->lu, G->piv, x);
    }
    return;
  }
  const int pre = G->gs_eff ? 1 : 0, post = G->gs_eff ? -1 : 0;
  smooth_dir(G, l, b, x, 1, pre);
  for (int s = 1; s < G->sweeps; ++s) smooth_dir(G, l, b, x, 0, pre);
  double *r = G->r[l];
  spmv(A, x, r);
  for (int i = 0; i < A->n; ++i) r[i] = b[i] - r[i];
  spmv(&G->R[l], r, G->b[l + 1]);
  vcycle(G, l + 1, G->b[l + 1], G->x[l + 1]);
  spmv(&G->P[l], G->x[l + 1], r);
  for (int i = 0; i < A->n; ++i) x[i] += r[i];
  for (int s = 0; s < G->sweeps; ++s) smooth_dir(G, l, b, x, 0, post);
}

/* kind: 0 = symmetric Gauss-Seidel sweeps before and after the coarse correction (precond_ml.h:50), 1 = Gauss-Seidel,
 * "efficient symmetric" (ml.xml of the benchmark protocol) */
void orc_amg_set_smoother(orc_amg *G, int kind) { G->gs_eff = kind == 1; }

void orc_amg_apply(const orc_amg *G, const double *r, double *z) { vcycle(G, 0, r, z); }

int orc_amg_levels(const orc_amg *G) { return G->nlev; }
/* info: rows, nnz(A_l), nnz(P_l) (0 on the last level) */
void orc_amg_level_info(const orc_amg *G, int l, int *info) {
  info[0] = G->A[l].n; info[1] = G->A[l].rp[G->A[l].n];
  info[2] = l < G->nlev - 1 ? G->P[l].rp[G->P[l].n] : 0;
}
/* what: 0 = A_l, 1 = P_l */
void orc_amg_export(const orc_amg *G, int l, int what, int *rowptr, int *colidx, double *val) {
  const csr_t *M = what == 0 ? &G->A[l] : &G->P[l];
  memcpy(rowptr, M->rp, sizeof(int) * (size_t)(M->n + 1));
  memcpy(colidx, M->ci, sizeof(int) * (size_t)M->rp[M->n]);
  memcpy(val, M->v, sizeof(double) * (size_t)M->rp[M->n]);
}
void orc_amg_export_aggregates(const orc_amg *G, int l, int *agg) {
  memcpy(agg, G->agg[l], sizeof(int) * (size_t)G->A[l].n);
}
void orc_amg_destroy(orc_amg *G) {
  if (!G) return;
  for (int l = 0; l < AMG_MAXLEV; ++l) {
    csr_free(&G->A[l]); csr_free(&G->P[l]); csr_free(&G->R[l]);
    free(G->agg[l]); free(G->nv[l]); free(G->dinv[l]); free(G->x[l]); free(G->b[l]); free(G->r[l]);
  }
  free(G->lu); free(G->piv); free(G);
}
