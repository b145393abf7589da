/*
 * isph_workload.h -- synthetic Taylor-Green-vortex particle sets (host only).
 *
 * Replaces, for benchmarking and tests, what LAMMPS hands PairISPH::compute
 * in the reference: atom->x / v / tag / type, the full neighbour list
 * (list->numneigh / firstneigh) and the ghost atoms of the rank's brick.
 * Geometry follows sph-script/taylor-green-vortex-{2d,3d}.lmp: box [0,2pi)^d
 * periodic, simple-cubic lattice dx = 2pi/N, h = 1.5 dx, cut = "cut over h"*h.
 *
 * No GPU code and no oracle code in here: this is the input side of the
 * drop-in boundary (what the LAMMPS adapter would pass in).
 */
#ifndef ISPH_WORKLOAD_H
#define ISPH_WORKLOAD_H

#ifdef __cplusplus
extern "C" {
#endif

enum { ISPH_TGV_LATTICE = 0, ISPH_TGV_JITTER = 1, ISPH_TGV_ADVECT = 2 };

typedef struct {
  int dim;               /* 2 or 3                                            */
  int ncell[3];          /* global lattice cells per axis (ncell[2]=1 in 2-D) */
  int pgrid[3];          /* rank grid (domain decomposition bricks)           */
  int rank;              /* this rank, x fastest                              */
  int brick[3];          /* in-rank particle ordering: bricks of this many cells
                            (0 => plain lexicographic order)                  */
  double origin[3];      /* lattice origin as a fraction of dx (0.5 in the 2-D
                            script, 0 in the 3-D script)                      */
  double h_over_dx;      /* 1.5                                               */
  double cut_over_h;     /* 2.0 Wendland, 3.0 Quintic                         */
  double skin;           /* neighbour-list skin (absolute length)             */
  int mode;              /* ISPH_TGV_*                                        */
  double jitter_amp;     /* JITTER: positions += U(-a,a)*h per axis           */
  unsigned long long seed;
  double umax;           /* 0.1                                               */
  double advect_dt;      /* ADVECT: positions += advect_dt * v_tgv(lattice)   */
  int basis;             /* particles per cell: 0/1 simple cubic (`lattice sc`), 2 body-centred cubic
                            (`lattice bcc ${dx}`, sph-script/pore-scale-flow-3d.lmp): second site at the
                            cell centre; the cell's particles are numbered consecutively */
  int colour_period;     /* numbering INSIDE a brick: 0/1 lexicographic (x fastest); c > 1: multi-colour -- the cells of
                            a brick are numbered colour by colour, colour = (ix mod c, iy mod c, iz mod c) (x fastest), and
                            lexicographically inside a colour.  With c = 3 cells of one colour are 3 spacings apart, i.e.
                            beyond the Wendland cut of the undisturbed lattice: the rows of one colour of a block-Jacobi
                            ILU(0) subdomain do not depend on each other and the triangular sweeps have ~c^3 levels
                            instead of one per lattice diagonal */
} isph_tgv_spec;

/* Sizes needed to allocate the arrays of isph_tgv_fill. */
int isph_tgv_count(const isph_tgv_spec *s, int *nlocal, int *nghost, long long *neigh_cap);

/* x,v: [nall][3]; tag: [nall] global 1-based id (ghost images share the
 * owner's tag); owner_rank/owner_index: [nall] owning rank and its local index
 * there; neigh_ptr: [nlocal+1]; neigh_idx: [<=neigh_cap] indices into
 * [0,nall).  Returns the number of neighbour entries, <0 on error. */
long long isph_tgv_fill(const isph_tgv_spec *s, double *x, double *v, int *tag,
                        int *owner_rank, int *owner_index,
                        int *neigh_ptr, int *neigh_idx);
/* Same with 64-bit list offsets (lists beyond 2^31 entries: bcc lattice + Quintic cut 3h at 4 M particles).
 * neigh_idx == NULL: only neigh_ptr is filled (count pass), so the caller can size neigh_idx exactly. */
long long isph_tgv_fill64(const isph_tgv_spec *s, double *x, double *v, int *tag,
                          int *owner_rank, int *owner_index,
                          long long *neigh_ptr, int *neigh_idx);

/* A general particle cloud in a periodic box (one rank): wrap-around ghost atoms + full neighbour list for owned
 * positions x[nlocal][3] already wrapped into [0, L) -- what LAMMPS rebuilds between two PairISPH::compute calls once the
 * particles have moved (Neighbor::build with `neighbor ${skin} bin`, bench-script/hopper/tgv/1728/tgv-3d-p24.lmp:95-96).
 * x_all == NULL: returns the number of ghosts.  Otherwise fills x_all[nall][3] (owned first), owner_index[nall],
 * neigh_ptr[nlocal+1] and -- when neigh_idx != NULL -- the lists (ascending particle index per row); returns the
 * number of list entries (call once with neigh_idx == NULL to size it).  < 0: bad arguments (L < 2 cut). */
long long isph_cloud_build(int dim, int nlocal, const double *x, const double *L, double cut, double *x_all,
                           int *owner_index, long long *neigh_ptr, int *neigh_idx);

#ifdef __cplusplus
}
#endif
#endif
