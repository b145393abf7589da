// workload.cpp -- synthetic Taylor-Green-vortex particle bricks (host C++).
//
// Stands in for the LAMMPS side of the boundary: per-rank atoms + ghost atoms
// (periodic images / off-rank neighbours) and the full neighbour list that
// PairISPH::compute receives (ref: pair_isph.cpp:1241-1260, functor.h:86-104),
// for the lattices of sph-script/taylor-green-vortex-{2d,3d}.lmp.
#include "isph_workload.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace {

struct Layout {
  int dim;
  int N[3], P[3], rc[3], lo[3], hi[3], n[3], g[3], ext[3], bs[3];
  double dx[3], h, cut, cutn;
  int nlocal, nghost;
  int nb;  // particles per lattice cell: 1 = simple cubic / square, 2 = bcc (second site at the cell centre)
};

inline int wrap(int c, int n) { int r = c % n; return r < 0 ? r + n : r; }

inline void split(int N, int P, int r, int &lo, int &hi) {
  lo = (int)((long long)r * N / P);
  hi = (int)((long long)(r + 1) * N / P);
}

// local index of owned cell c (coordinates relative to the rank's lo) in a
// rank whose owned extent is n, ordered in bricks of bs (x fastest everywhere)
// rank of cell (i0,i1,i2) of a brick of bd cells in the multi-colour numbering of period cp (isph_workload.h):
// cells before it = all cells of smaller colours + cells of its colour that precede it lexicographically
inline long long colour_rank(const int bd[3], const int i[3], int cp) {
  auto cnt = [&](int extent, int colour) { return colour < extent ? (extent - 1 - colour) / cp + 1 : 0; };  // cells with coordinate = colour mod cp
  const int col[3] = {i[0] % cp, i[1] % cp, i[2] % cp};
  long long before = 0;
  for (int c2 = 0; c2 < cp; ++c2)
    for (int c1 = 0; c1 < cp; ++c1)
      for (int c0 = 0; c0 < cp; ++c0) {
        const bool smaller = c2 < col[2] || (c2 == col[2] && (c1 < col[1] || (c1 == col[1] && c0 < col[0])));
        if (smaller) before += (long long)cnt(bd[0], c0) * cnt(bd[1], c1) * cnt(bd[2], c2);
      }
  const long long n0 = cnt(bd[0], col[0]), n1 = cnt(bd[1], col[1]);
  return before + ((long long)(i[2] / cp) * n1 + i[1] / cp) * n0 + i[0] / cp;
}

inline int brick_index(const int n[3], const int bs[3], const int c[3], int cp = 0) {
  int b[3], i[3], bd[3];
  for (int a = 0; a < 3; ++a) {
    const int s = bs[a] > 0 ? bs[a] : n[a];
    b[a] = c[a] / s;
    i[a] = c[a] % s;
    const int rem = n[a] - b[a] * s;
    bd[a] = rem < s ? rem : s;
  }
  const int s0 = bs[0] > 0 ? bs[0] : n[0], s1 = bs[1] > 0 ? bs[1] : n[1], s2 = bs[2] > 0 ? bs[2] : n[2];
  long long off = (long long)b[2] * s2 * n[0] * n[1];
  off += (long long)b[1] * s1 * n[0] * bd[2];
  off += (long long)b[0] * s0 * bd[1] * bd[2];
  off += cp > 1 ? colour_rank(bd, i, cp) : ((long long)i[2] * bd[1] + i[1]) * bd[0] + i[0];
  return (int)off;
}

inline uint64_t splitmix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ULL;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}
inline double u01(uint64_t seed, uint64_t gid, int axis) {
  const uint64_t r = splitmix64(splitmix64(seed) ^ (gid * 3u + (uint64_t)axis));
  return (double)(r >> 11) * (1.0 / 9007199254740992.0);
}

bool make_layout(const isph_tgv_spec *s, Layout &L) {
  L.dim = s->dim;
  if (s->dim != 2 && s->dim != 3) return false;
  L.nb = s->basis == 2 ? 2 : 1;
  double maxdisp = 0.0;
  for (int a = 0; a < 3; ++a) {
    L.N[a] = (a < s->dim) ? s->ncell[a] : 1;
    L.P[a] = (a < s->dim) ? (s->pgrid[a] > 0 ? s->pgrid[a] : 1) : 1;
    L.bs[a] = (a < s->dim) ? s->brick[a] : 0;
    if (L.N[a] < 1 || L.P[a] > L.N[a]) return false;
  }
  // one lattice spacing in every direction: dx = 2 pi / (fewest cells of a side), so a box of (2N, N, N) cells is 4 pi x
  // 2 pi x 2 pi (every side a whole number of periods of the vortex) -- weak scaling over 2 or 4 bricks keeps the
  // stencil of the single brick instead of stretching the lattice along the short sides
  int nmin = L.N[0];
  for (int a = 1; a < s->dim; ++a) nmin = std::min(nmin, L.N[a]);
  for (int a = 0; a < 3; ++a) L.dx[a] = 2.0 * M_PI / nmin;
  int r = s->rank;
  if (r < 0 || r >= L.P[0] * L.P[1] * L.P[2]) return false;
  L.rc[0] = r % L.P[0]; r /= L.P[0];
  L.rc[1] = r % L.P[1]; r /= L.P[1];
  L.rc[2] = r;
  L.h = s->h_over_dx * L.dx[0];
  L.cut = s->cut_over_h * L.h;
  L.cutn = L.cut + s->skin;
  if (s->mode == ISPH_TGV_JITTER) maxdisp = s->jitter_amp * L.h;
  if (s->mode == ISPH_TGV_ADVECT) maxdisp = std::fabs(s->advect_dt) * s->umax;
  L.nlocal = 1;
  long long nall = 1;
  for (int a = 0; a < 3; ++a) {
    split(L.N[a], L.P[a], L.rc[a], L.lo[a], L.hi[a]);
    L.n[a] = L.hi[a] - L.lo[a];
    L.g[a] = (a < s->dim) ? (int)std::floor((L.cutn + 2.0 * maxdisp) / L.dx[a] + 1e-6) : 0;
    L.ext[a] = L.n[a] + 2 * L.g[a];
    L.nlocal *= L.n[a];
    nall *= L.ext[a];
  }
  if (nall * L.nb > 2000000000LL) return false;
  L.nghost = (int)(nall - L.nlocal) * L.nb;
  L.nlocal *= L.nb;
  return true;
}

inline void tgv_velocity(double umax, const double p[3], double v[3]) {
  v[0] = umax * std::sin(p[0]) * std::cos(p[1]);
  v[1] = -umax * std::cos(p[0]) * std::sin(p[1]);
  v[2] = 0.0;
}

// position / velocity of the particle born in wrapped global cell gw, placed
// at unwrapped cell gc (gc == gw for owned cells)
void particle_state(const isph_tgv_spec *s, const Layout &L, const int gc[3], const int gw[3], int b,
                    double x[3], double v[3]) {
  double base[3], disp[3] = {0, 0, 0};
  const uint64_t gid = (((uint64_t)gw[2] * L.N[1] + gw[1]) * L.N[0] + gw[0]) * (uint64_t)L.nb + (uint64_t)b;
  const double boff = b ? 0.5 : 0.0;  // bcc: the second site sits at the cell centre
  for (int a = 0; a < 3; ++a)
    base[a] = (a < L.dim) ? (gw[a] + s->origin[a] + boff) * L.dx[a] : 0.0;
  if (s->mode == ISPH_TGV_JITTER) {
    for (int a = 0; a < L.dim; ++a) disp[a] = s->jitter_amp * L.h * (2.0 * u01(s->seed, gid, a) - 1.0);
    double p[3] = {base[0] + disp[0], base[1] + disp[1], base[2] + disp[2]};
    tgv_velocity(s->umax, p, v);
  } else {
    tgv_velocity(s->umax, base, v);
    if (s->mode == ISPH_TGV_ADVECT)
      for (int a = 0; a < L.dim; ++a) disp[a] = s->advect_dt * v[a];
  }
  for (int a = 0; a < 3; ++a)
    x[a] = (a < L.dim) ? (gc[a] + s->origin[a] + boff) * L.dx[a] + disp[a] : 0.0;
}

}  // namespace

extern "C" int isph_tgv_count(const isph_tgv_spec *s, int *nlocal, int *nghost, long long *neigh_cap) {
  Layout L;
  if (!make_layout(s, L)) return -1;
  *nlocal = L.nlocal;
  *nghost = L.nghost;
  long long stencil = 1;
  for (int a = 0; a < 3; ++a) stencil *= (2 * L.g[a] + 1);
  *neigh_cap = (long long)L.nlocal * (stencil * L.nb - 1);
  return 0;
}

namespace {
template <class OFF>
long long tgv_fill_t(const isph_tgv_spec *s, double *x, double *v, int *tag, int *owner_rank, int *owner_index,
                     OFF *neigh_ptr, int *neigh_idx) {
  Layout L;
  if (!make_layout(s, L)) return -1;
  const long long next = (long long)L.ext[0] * L.ext[1] * L.ext[2];
  std::vector<int> pidx((size_t)next);  // extended cell -> index of its first particle (nb consecutive particles per cell)
  // owned cells -> brick order, ghosts -> nlocal + running counter
  const int nb = L.nb;
  int ghost = L.nlocal;
  for (int ez = 0; ez < L.ext[2]; ++ez)
    for (int ey = 0; ey < L.ext[1]; ++ey)
      for (int ex = 0; ex < L.ext[0]; ++ex) {
        const int e[3] = {ex, ey, ez};
        int c[3], gc[3], gw[3];
        bool own = true;
        for (int a = 0; a < 3; ++a) {
          c[a] = e[a] - L.g[a];
          gc[a] = L.lo[a] + c[a];
          gw[a] = wrap(gc[a], L.N[a]);
          if (c[a] < 0 || c[a] >= L.n[a]) own = false;
        }
        const long long ec = ((long long)ez * L.ext[1] + ey) * L.ext[0] + ex;
        int p, orank_p, oidx_p;
        if (own) {
          p = brick_index(L.n, L.bs, c, s->colour_period) * nb;
          orank_p = s->rank;
          oidx_p = p;
        } else {
          p = ghost;
          ghost += nb;
          int orc[3], olo[3], ohi[3], on[3], oc[3];
          for (int a = 0; a < 3; ++a) {
            // owner rank coordinate along a: invert split()
            int r = (int)(((long long)(gw[a] + 1) * L.P[a] - 1) / L.N[a]);
            split(L.N[a], L.P[a], r, olo[a], ohi[a]);
            while (gw[a] < olo[a]) { --r; split(L.N[a], L.P[a], r, olo[a], ohi[a]); }
            while (gw[a] >= ohi[a]) { ++r; split(L.N[a], L.P[a], r, olo[a], ohi[a]); }
            orc[a] = r;
            on[a] = ohi[a] - olo[a];
            oc[a] = gw[a] - olo[a];
          }
          orank_p = (orc[2] * L.P[1] + orc[1]) * L.P[0] + orc[0];
          oidx_p = brick_index(on, L.bs, oc, s->colour_period) * nb;
        }
        pidx[(size_t)ec] = p;
        for (int b = 0; b < nb; ++b) {
          owner_rank[p + b] = orank_p;
          owner_index[p + b] = oidx_p + b;
          tag[p + b] = 1 + (int)((((long long)gw[2] * L.N[1] + gw[1]) * L.N[0] + gw[0]) * nb + b);
          particle_state(s, L, gc, gw, b, &x[3 * (size_t)(p + b)], &v[3 * (size_t)(p + b)]);
        }
      }
  // full neighbour list, candidates in ascending extended-cell order
  const double cutnsq = L.cutn * L.cutn;
  std::vector<int> rowcnt((size_t)L.nlocal, 0);
  // pass 1 counts, pass 2 fills (rows are in brick order, cells are visited
  // lexicographically, so we need the counts first)
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) {
      neigh_ptr[0] = 0;
      for (int i = 0; i < L.nlocal; ++i) neigh_ptr[i + 1] = neigh_ptr[i] + (OFF)rowcnt[(size_t)i];
      if (!neigh_idx) break;  // count-only call: the caller sizes neigh_idx from neigh_ptr[nlocal]
    }
#pragma omp parallel for schedule(static)
    for (int cz = 0; cz < L.n[2]; ++cz)
      for (int cy = 0; cy < L.n[1]; ++cy)
        for (int cx = 0; cx < L.n[0]; ++cx) {
          const int ex = cx + L.g[0], ey = cy + L.g[1], ez = cz + L.g[2];
          const int i0 = pidx[(size_t)(((long long)ez * L.ext[1] + ey) * L.ext[0] + ex)];
          for (int bi = 0; bi < nb; ++bi) {
            const int i = i0 + bi;
            const double *xi = &x[3 * (size_t)i];
            int cnt = 0;
            int *out = pass == 1 ? &neigh_idx[neigh_ptr[i]] : nullptr;
            for (int oz = -L.g[2]; oz <= L.g[2]; ++oz)
              for (int oy = -L.g[1]; oy <= L.g[1]; ++oy)
                for (int ox = -L.g[0]; ox <= L.g[0]; ++ox) {
                  const int j0 = pidx[(size_t)(((long long)(ez + oz) * L.ext[1] + (ey + oy)) * L.ext[0] + (ex + ox))];
                  for (int bj = 0; bj < nb; ++bj) {
                    const int j = j0 + bj;
                    if (j == i) continue;
                    const double *xj = &x[3 * (size_t)j];
                    double rsq = 0.0;
                    for (int a = 0; a < L.dim; ++a) {
                      const double d = xi[a] - xj[a];
                      rsq += d * d;
                    }
                    if (rsq < cutnsq) {
                      if (out) out[cnt] = j;
                      ++cnt;
                    }
                  }
                }
            if (pass == 0) rowcnt[(size_t)i] = cnt;
          }
        }
  }
  return (long long)neigh_ptr[L.nlocal];
}
}  // namespace

extern "C" long long isph_tgv_fill(const isph_tgv_spec *s, double *x, double *v, int *tag, int *owner_rank,
                                   int *owner_index, int *neigh_ptr, int *neigh_idx) {
  return tgv_fill_t<int>(s, x, v, tag, owner_rank, owner_index, neigh_ptr, neigh_idx);
}

extern "C" long long isph_tgv_fill64(const isph_tgv_spec *s, double *x, double *v, int *tag, int *owner_rank,
                                     int *owner_index, long long *neigh_ptr, int *neigh_idx) {
  return tgv_fill_t<long long>(s, x, v, tag, owner_rank, owner_index, neigh_ptr, neigh_idx);
}

// ---------------------------------------------------------------------------------------------------------------------
// General particle cloud in a periodic box: what LAMMPS does between two calls of PairISPH::compute when the particles
// have moved -- wrap into the box, create the ghost atoms (periodic images within the cut of a face) and rebuild the full
// neighbour list with cells of the cut's size (Neighbor::build; `neighbor ${skin} bin`, `neigh_modify every 1`,
// bench-script/hopper/tgv/1728/tgv-3d-p24.lmp:95-96).  One rank, periodic in every direction.  Host only; not part of the
// path (the reference gets these arrays from LAMMPS), used by bench.py --workload step to run consecutive time steps.
namespace {
struct Cloud {
  int dim, nlocal;
  double L[3], cut;
  std::vector<double> gx;        // ghost positions [nghost][3]
  std::vector<int> gowner;       // ghost -> owner
};

inline void cloud_ghosts(int dim, int nlocal, const double *x, const double *L, double cut, Cloud &C) {
  C.dim = dim; C.nlocal = nlocal; C.cut = cut;
  for (int a = 0; a < 3; ++a) C.L[a] = L[a];
  const int s2 = dim == 3 ? 1 : 0;
  for (int i = 0; i < nlocal; ++i) {
    const double *p = &x[3 * (size_t)i];
    int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};  // image shifts that can land within the cut of the box
    for (int a = 0; a < dim; ++a) {
      if (p[a] < cut) hi[a] = 1;            // image at p + L lies within cut beyond the upper face
      if (p[a] >= L[a] - cut) lo[a] = -1;   // image at p - L lies within cut below the lower face
    }
    for (int sz = (s2 ? lo[2] : 0); sz <= (s2 ? hi[2] : 0); ++sz)
      for (int sy = lo[1]; sy <= hi[1]; ++sy)
        for (int sx = lo[0]; sx <= hi[0]; ++sx) {
          if (!sx && !sy && !sz) continue;
          C.gx.push_back(p[0] + sx * L[0]);
          C.gx.push_back(p[1] + sy * L[1]);
          C.gx.push_back(dim == 3 ? p[2] + sz * L[2] : 0.0);
          C.gowner.push_back(i);
        }
  }
}
}  // namespace

// Pass 1 (x_all == NULL): returns the number of ghosts for the wrapped owned positions x[nlocal][3].
// Pass 2: fills x_all[nall][3] (owned first, then ghosts), owner_index[nall], neigh_ptr[nlocal+1] (64-bit) and, when
// neigh_idx != NULL, the lists; returns the number of list entries.  Call with neigh_idx == NULL to size it.
extern "C" long long isph_cloud_build(int dim, int nlocal, const double *x, const double *L, double cut, double *x_all,
                                      int *owner_index, long long *neigh_ptr, int *neigh_idx) {
  if ((dim != 2 && dim != 3) || nlocal < 0 || cut <= 0.0) return -1;
  for (int a = 0; a < dim; ++a)
    if (L[a] < 2.0 * cut) return -1;  // a particle would see two images of a neighbour
  Cloud C;
  cloud_ghosts(dim, nlocal, x, L, cut, C);
  const int nghost = (int)C.gowner.size(), nall = nlocal + nghost;
  if (!x_all) return nghost;
  for (size_t k = 0; k < (size_t)nlocal * 3; ++k) x_all[k] = x[k];
  for (size_t k = 0; k < C.gx.size(); ++k) x_all[(size_t)nlocal * 3 + k] = C.gx[k];
  for (int i = 0; i < nlocal; ++i) owner_index[i] = i;
  for (int g = 0; g < nghost; ++g) owner_index[nlocal + g] = C.gowner[(size_t)g];
  // cells of at least the cut's size over [-cut, L + cut)
  int nc[3] = {1, 1, 1};
  double cs[3] = {1, 1, 1}, org[3] = {0, 0, 0};
  for (int a = 0; a < dim; ++a) {
    nc[a] = std::max(1, (int)std::floor((L[a] + 2.0 * cut) / cut));
    cs[a] = (L[a] + 2.0 * cut) / nc[a];
    org[a] = -cut;
  }
  const long long ncell = (long long)nc[0] * nc[1] * nc[2];
  auto cell_of = [&](const double *p, int c[3]) {
    for (int a = 0; a < 3; ++a) {
      c[a] = a < dim ? (int)std::floor((p[a] - org[a]) / cs[a]) : 0;
      if (c[a] < 0) c[a] = 0;
      if (c[a] >= nc[a]) c[a] = nc[a] - 1;
    }
  };
  std::vector<int> cstart((size_t)ncell + 1, 0), order((size_t)nall), cid((size_t)nall);
  for (int j = 0; j < nall; ++j) {
    int c[3];
    cell_of(&x_all[3 * (size_t)j], c);
    cid[(size_t)j] = (c[2] * nc[1] + c[1]) * nc[0] + c[0];
    ++cstart[(size_t)cid[(size_t)j] + 1];
  }
  for (long long c = 0; c < ncell; ++c) cstart[(size_t)c + 1] += cstart[(size_t)c];
  {
    std::vector<int> fill(cstart.begin(), cstart.end() - 1);
    for (int j = 0; j < nall; ++j) order[(size_t)fill[(size_t)cid[(size_t)j]]++] = j;  // ascending particle index inside a cell
  }
  const double cutsq = cut * cut;
  std::vector<int> cnt((size_t)nlocal, 0);
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) {
      neigh_ptr[0] = 0;
      for (int i = 0; i < nlocal; ++i) neigh_ptr[i + 1] = neigh_ptr[i] + cnt[(size_t)i];
      if (!neigh_idx) break;
    }
#pragma omp parallel for schedule(dynamic, 1024)
    for (int i = 0; i < nlocal; ++i) {
      const double *xi = &x_all[3 * (size_t)i];
      int c[3];
      cell_of(xi, c);
      int n = 0;
      int *out = pass == 1 ? &neigh_idx[neigh_ptr[i]] : nullptr;
      for (int oz = (dim == 3 ? -1 : 0); oz <= (dim == 3 ? 1 : 0); ++oz) {
        const int cz = c[2] + oz;
        if (cz < 0 || cz >= nc[2]) continue;
        for (int oy = -1; oy <= 1; ++oy) {
          const int cy = c[1] + oy;
          if (cy < 0 || cy >= nc[1]) continue;
          for (int ox = -1; ox <= 1; ++ox) {
            const int cx = c[0] + ox;
            if (cx < 0 || cx >= nc[0]) continue;
            const long long cc = ((long long)cz * nc[1] + cy) * nc[0] + cx;
            for (int q = cstart[(size_t)cc]; q < cstart[(size_t)cc + 1]; ++q) {
              const int j = order[(size_t)q];
              if (j == i) continue;
              const double *xj = &x_all[3 * (size_t)j];
              double rsq = 0.0;
              for (int a = 0; a < dim; ++a) { const double d = xi[a] - xj[a]; rsq += d * d; }
              if (rsq < cutsq) { if (out) out[n] = j; ++n; }
            }
          }
        }
      }
      if (pass == 0) cnt[(size_t)i] = n;
      else if (out) std::sort(out, out + n);  // ascending particle index: a fixed summation order whatever the cell walk
    }
  }
  return neigh_ptr[nlocal];
}
