// lammps_formats.cpp -- host-side converters declared in include/isph_lammps.h (part of libisph_host.so).
#include <cstddef>
#include <unordered_map>
#include <vector>

#include "isph_lammps.h"

extern "C" {

long long isph_flatten_neighbor_list(int inum, const int *ilist, const int *numneigh, int *const *firstneigh, int nlocal,
                                     int nall, int *neigh_ptr, long long *neigh_ptr64, int *neigh_idx) {
  if (inum < 0 || nlocal < 0 || nall < nlocal || !neigh_ptr64 || (inum > 0 && (!ilist || !numneigh || !firstneigh))) return -1;
  std::vector<int> len((size_t)nlocal, -1);
  for (int ii = 0; ii < inum; ++ii) {
    const int i = ilist[ii];
    if (i < 0 || i >= nlocal || len[(size_t)i] >= 0 || numneigh[i] < 0) return -1;
    len[(size_t)i] = numneigh[i];
  }
  long long run = 0;
  for (int i = 0; i < nlocal; ++i) {
    neigh_ptr64[i] = run;
    if (len[(size_t)i] > 0) run += len[(size_t)i];
  }
  neigh_ptr64[nlocal] = run;
  if (neigh_ptr && run < 2147483647LL)
    for (int i = 0; i <= nlocal; ++i) neigh_ptr[i] = (int)neigh_ptr64[i];
  if (!neigh_idx) return run;
  for (int ii = 0; ii < inum; ++ii) {
    const int i = ilist[ii];
    const int *jlist = firstneigh[i];
    int *out = neigh_idx + neigh_ptr64[i];
    for (int jj = 0; jj < numneigh[i]; ++jj) {
      const int j = jlist[jj] & ISPH_NEIGHMASK;
      if (j >= nall) return -1;
      out[jj] = j;
    }
  }
  return run;
}

int isph_colmap_from_tags(int nlocal, int nall, const int *tag, int *colmap, int *ghost_tag_out) {
  if (nlocal < 0 || nall < nlocal || (nall > 0 && (!tag || !colmap))) return -1;
  std::unordered_map<int, int> col;
  col.reserve((size_t)nall * 2);
  for (int i = 0; i < nlocal; ++i) {
    if (!col.emplace(tag[i], i).second) return -1;
    colmap[i] = i;
  }
  int ncol = nlocal;
  for (int g = nlocal; g < nall; ++g) {
    auto it = col.find(tag[g]);
    if (it == col.end()) {
      if (ghost_tag_out) ghost_tag_out[ncol - nlocal] = tag[g];
      it = col.emplace(tag[g], ncol++).first;
    }
    colmap[g] = it->second;
  }
  return ncol;
}

}  // extern "C"
