/*
 * isph_lammps.h -- host-side converters from the data formats LAMMPS hands PairISPH::compute to the flat arrays of
 * isph_particles (include/isph_hip.h).  Plain C, no GPU code: the adapter calls them once per neighbour-list build.
 *
 * Replaces what every reference functor does on the fly:
 *   FunctorOuter's members  _inum, _ilist, _numneigh, _firstneigh = list->{inum, ilist, numneigh, firstneigh}
 *   (functor.h:65,83-86) and the per-entry  j = jlist[jj] & NEIGHMASK  (functor_laplacian_matrix.h:131 and every other
 *   neighbour loop; NEIGHMASK = 0x3FFFFFFF strips LAMMPS' special-bond bits), and
 *   the tag -> local matrix column look-up Epetra does inside SumIntoGlobalValues / FillComplete with the node map built
 *   from atom->tag (pair_isph.cpp:1258-1270, functor_graph.h:61-97).
 */
#ifndef ISPH_LAMMPS_H
#define ISPH_LAMMPS_H

#ifdef __cplusplus
extern "C" {
#endif

#define ISPH_NEIGHMASK 0x3FFFFFFF

/* Full neighbour list -> CSR.  ilist[0..inum) are the owned atoms in list order (LAMMPS guarantees every owned atom
 * appears once for a full list); numneigh / firstneigh are indexed by atom.  neigh_ptr64 [nlocal+1] always receives the
 * offsets; neigh_ptr [nlocal+1] too when it is non-NULL and the list has < 2^31 entries (the int offsets LAMMPS-sized
 * runs use); neigh_idx may be NULL to count only.  Returns the number of entries, or -1 for a bad list (an atom listed
 * twice, an index outside [0, nall)). */
long long isph_flatten_neighbor_list(int inum, const int *ilist, const int *numneigh, int *const *firstneigh, int nlocal,
                                     int nall, int *neigh_ptr, long long *neigh_ptr64, int *neigh_idx);

/* Matrix column of every local + ghost atom from the tags: owned atoms get their local index; a ghost whose tag is owned
 * by this rank (periodic image) gets the owner's column; every other ghost tag gets one ghost column, numbered from
 * nlocal in order of first appearance -- the column map Epetra builds in FillComplete.  Returns the number of columns
 * (nlocal + distinct remote tags) or -1 if two owned atoms share a tag.  ghost_tag_out (may be NULL) receives the tag of
 * every ghost column, [ncol - nlocal]. */
int isph_colmap_from_tags(int nlocal, int nall, const int *tag, int *colmap, int *ghost_tag_out);

#ifdef __cplusplus
}
#endif
#endif
