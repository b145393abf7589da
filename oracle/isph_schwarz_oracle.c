/* isph_schwarz_oracle.c -- TEST INFRASTRUCTURE (CPU oracle), never shipped.
 *
 * Ifpack_AdditiveSchwarz<Ifpack_ILU> as PrecondWrapper_Ifpack configures it
 * (ref: precond_ifpack.h:28-75):
 *   "Precond Type" = "ILU", "Overlap Level" = 1 (:43), "fact: level-of-fill" = 1 (:35),
 *   "schwarz: combine mode" = "Add" (:37), Factory.Create(type, A, overlap) (:63).
 * Trilinos (Ifpack) is not vendored with the reference and no version is pinned
 * (README:9-11), so the algorithm restated here is Ifpack's published one:
 *   - every MPI rank is one subdomain; its local matrix is extended by `overlap`
 *     layers of rows (Ifpack_OverlappingRowMatrix: layer l = the off-subdomain
 *     columns referenced by the rows of the layers < l; the new rows are appended
 *     after the rows already present),
 *   - Ifpack_LocalFilter drops every entry whose column is not a row of the
 *     extended subdomain,
 *   - ILU(k) of that local matrix in its local row order (Ifpack_ILU: level-of-fill
 *     pattern, relax 0, absolute threshold 0, relative threshold 1),
 *   - apply: import r on the extended rows, solve L U z_s = r_s, export with the
 *     combine mode: Add sums the contributions of all subdomains on shared rows,
 *     Zero keeps the owner's value only (restricted additive Schwarz).
 *   On ONE rank there is a single subdomain = the whole matrix and the overlap is
 *   a no-op (Ifpack_AdditiveSchwarz: IsOverlapping_ = false when NumProc() == 1).
 * One detail is not defined by the reference's own sources: the order in which
 * Ifpack_OverlappingRowMatrix appends the rows of one layer (it follows Epetra's
 * column map).  Here a layer's rows are appended in ascending global row number.
 *
 * The subdomains of this restatement are given as consecutive row ranges
 * own_ptr[s]..own_ptr[s+1] (rank s of the reference's linear/nodal map, or one
 * block of the GPU's block decomposition). */
#include <stdlib.h>
#include <string.h>

#include "isph_oracle.h"

struct orc_schwarz {
  int n, nsub, nloc, combine;
  int *loc_ptr;  /* [nsub+1] first local row of every subdomain */
  int *nown;     /* [nsub] owned rows (they come first)          */
  int *rows;     /* [nloc] global row of every local row         */
  orc_ilu *F;    /* ILU(k) of the block-diagonal local matrix    */
  double *rl, *zl;
};

static int cmp_int_s(const void *a, const void *b) {
  const int x = *(const int *)a, y = *(const int *)b;
  return (x > y) - (x < y);
}

orc_schwarz *orc_schwarz_create(int n, const int *rowptr, const int *colidx, const double *val, int level_of_fill,
                                int nsub, const int *own_ptr, int overlap, int combine) {
  orc_schwarz *S = (orc_schwarz *)calloc(1, sizeof(orc_schwarz));
  S->n = n; S->nsub = nsub; S->combine = combine;
  S->loc_ptr = (int *)calloc((size_t)nsub + 1, sizeof(int));
  S->nown = (int *)calloc((size_t)nsub, sizeof(int));
  int **srows = (int **)calloc((size_t)nsub, sizeof(int *));
  int *scnt = (int *)calloc((size_t)nsub, sizeof(int));
  int *loc = (int *)malloc(sizeof(int) * (size_t)n); /* global row -> local row of the current subdomain, -1 */
  for (int i = 0; i < n; ++i) loc[i] = -1;
  for (int s = 0; s < nsub; ++s) {
    const int lo = own_ptr[s], hi = own_ptr[s + 1];
    int cap = (hi - lo) * 2 + 16, cnt = 0;
    int *rows = (int *)malloc(sizeof(int) * (size_t)cap);
    for (int i = lo; i < hi; ++i) { rows[cnt] = i; loc[i] = cnt++; }
    S->nown[s] = hi - lo;
    int layer_lo = 0;
    for (int l = 0; l < overlap && nsub > 1; ++l) { /* one rank: overlap is a no-op */
      const int layer_hi = cnt;
      int ncand = 0, ccap = 1024;
      int *cand = (int *)malloc(sizeof(int) * (size_t)ccap);
      for (int q = layer_lo; q < layer_hi; ++q) {
        const int i = rows[q];
        for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) {
          const int c = colidx[p];
          if (c < 0 || c >= n) continue; /* ghost column of another rank: not part of this restatement */
          if (loc[c] == -1) {
            loc[c] = -2; /* candidate */
            if (ncand == ccap) { ccap *= 2; cand = (int *)realloc(cand, sizeof(int) * (size_t)ccap); }
            cand[ncand++] = c;
          }
        }
      }
      qsort(cand, (size_t)ncand, sizeof(int), cmp_int_s);
      if (cnt + ncand > cap) { cap = (cnt + ncand) * 2; rows = (int *)realloc(rows, sizeof(int) * (size_t)cap); }
      for (int k = 0; k < ncand; ++k) { rows[cnt] = cand[k]; loc[cand[k]] = cnt++; }
      free(cand);
      layer_lo = layer_hi;
    }
    srows[s] = rows; scnt[s] = cnt;
    S->loc_ptr[s + 1] = S->loc_ptr[s] + cnt;
    for (int q = 0; q < cnt; ++q) loc[rows[q]] = -1;
  }
  S->nloc = S->loc_ptr[nsub];
  S->rows = (int *)malloc(sizeof(int) * (size_t)(S->nloc > 0 ? S->nloc : 1));
  for (int s = 0; s < nsub; ++s) memcpy(S->rows + S->loc_ptr[s], srows[s], sizeof(int) * (size_t)scnt[s]);
  /* local block-diagonal matrix (Ifpack_LocalFilter), columns in local numbering, sorted */
  int *lrp = (int *)calloc((size_t)S->nloc + 1, sizeof(int));
  for (int s = 0; s < nsub; ++s) {
    for (int q = 0; q < scnt[s]; ++q) loc[srows[s][q]] = q;
    for (int q = 0; q < scnt[s]; ++q) {
      const int i = srows[s][q];
      int c = 0;
      for (int p = rowptr[i]; p < rowptr[i + 1]; ++p)
        if (colidx[p] >= 0 && colidx[p] < n && loc[colidx[p]] >= 0) ++c;
      lrp[S->loc_ptr[s] + q + 1] = c;
    }
    for (int q = 0; q < scnt[s]; ++q) loc[srows[s][q]] = -1;
  }
  for (int i = 0; i < S->nloc; ++i) lrp[i + 1] += lrp[i];
  const int lnnz = lrp[S->nloc];
  int *lci = (int *)malloc(sizeof(int) * (size_t)(lnnz > 0 ? lnnz : 1));
  double *lv = (double *)malloc(sizeof(double) * (size_t)(lnnz > 0 ? lnnz : 1));
  for (int s = 0; s < nsub; ++s) {
    const int base = S->loc_ptr[s];
    for (int q = 0; q < scnt[s]; ++q) loc[srows[s][q]] = q;
    for (int q = 0; q < scnt[s]; ++q) {
      const int i = srows[s][q];
      int w = lrp[base + q];
      for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) {
        const int c = colidx[p];
        if (c >= 0 && c < n && loc[c] >= 0) { lci[w] = base + loc[c]; lv[w] = val[p]; ++w; }
      }
      /* sort the row by local column (insertion sort: rows are ~100 entries and nearly sorted) */
      for (int a = lrp[base + q] + 1; a < w; ++a) {
        const int kc = lci[a]; const double kv = lv[a];
        int b = a - 1;
        while (b >= lrp[base + q] && lci[b] > kc) { lci[b + 1] = lci[b]; lv[b + 1] = lv[b]; --b; }
        lci[b + 1] = kc; lv[b + 1] = kv;
      }
    }
    for (int q = 0; q < scnt[s]; ++q) loc[srows[s][q]] = -1;
    free(srows[s]);
  }
  S->F = orc_ilu_create(S->nloc, lrp, lci, lv, level_of_fill, nsub, S->loc_ptr);
  S->rl = (double *)malloc(sizeof(double) * (size_t)(S->nloc > 0 ? S->nloc : 1));
  S->zl = (double *)malloc(sizeof(double) * (size_t)(S->nloc > 0 ? S->nloc : 1));
  free(lrp); free(lci); free(lv); free(srows); free(scnt); free(loc);
  return S;
}

void orc_schwarz_apply(const orc_schwarz *S, const double *r, double *z) {
  for (int q = 0; q < S->nloc; ++q) S->rl[q] = r[S->rows[q]];
  orc_ilu_apply(S->F, S->rl, S->zl);
  memset(z, 0, sizeof(double) * (size_t)S->n);
  for (int s = 0; s < S->nsub; ++s) {
    const int base = S->loc_ptr[s];
    const int cnt = S->combine == 0 ? S->loc_ptr[s + 1] - base : S->nown[s]; /* Add : Zero (restricted) */
    for (int q = 0; q < cnt; ++q) z[S->rows[base + q]] += S->zl[base + q];
  }
}

int orc_schwarz_nloc(const orc_schwarz *S) { return S->nloc; }
int orc_schwarz_nnz(const orc_schwarz *S) { return orc_ilu_nnz(S->F); }
/* rows[nloc], loc_ptr[nsub+1], and the factor of the block-diagonal local matrix as CSR in local numbering */
void orc_schwarz_export(const orc_schwarz *S, int *rows, int *loc_ptr, int *rowptr, int *colidx, double *val) {
  memcpy(rows, S->rows, sizeof(int) * (size_t)S->nloc);
  memcpy(loc_ptr, S->loc_ptr, sizeof(int) * (size_t)(S->nsub + 1));
  orc_ilu_export(S->F, rowptr, colidx, val);
}
void orc_schwarz_destroy(orc_schwarz *S) {
  if (!S) return;
  orc_ilu_destroy(S->F);
  free(S->loc_ptr); free(S->nown); free(S->rows); free(S->rl); free(S->zl); free(S);
}
