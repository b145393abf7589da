// ilu.hpp -- block-Jacobi ILU(0) on the GPU.
//
// Replaces PrecondWrapper_Ifpack::create() + Belos::EpetraPrecOp::Apply
// (ref: precond_ifpack.h:52-75, solver_lin_belos.h:147-156) for the setting
// "Precond Type"=ILU, "Overlap Level"=0, "fact: level-of-fill"=0 with one
// additive-Schwarz subdomain per block of B consecutive rows (what Ifpack gives
// with one MPI rank per block).  Entries that couple different blocks are
// dropped, exactly like the off-rank columns of the reference's local matrix.
//
// Data layout: the factor F shares A's sliced-ELL geometry (same slice
// offsets): row i keeps its in-block entries, columns ascending, in slots
// [0, flen[i]); fdiag[i] is the slot of the diagonal.  Strict-L (unit
// diagonal implied), D and strict-U live in one pattern == A's in-block pattern.
//
// Parallelism: no symbolic level analysis.  Both the numeric factorisation and
// the triangular solves are "sync-free": a block is owned by one workgroup, a
// row advances as soon as the rows it depends on have published their result
// through a flag in LDS.  Dependencies always point to lower (L) / higher (U)
// rows of the same workgroup, every wait loop is wave-uniform and re-polled, so
// every wave reaches its exit.
#pragma once
#include "core.hpp"
#include "sell.hpp"

struct isph_ilu {
  int n = 0, B = 0, nblocks = 0, wmax = 0;
  const isph::Sell *S = nullptr;  // geometry shared with A (A must outlive the factor)
  isph::DevBuf<int> fcol, flen, fdiag, err;
  isph::DevBuf<double> fval;
  // statically scheduled triangular solves (see k_ilu_schedule / k_ilu_solve_stream)
  isph::DevBuf<double> sv;        // chunk stream values   [nchunks*64]
  isph::DevBuf<unsigned> sc;      // chunk stream words    col | row<<16 | END<<31
  isph::DevBuf<int> fdst;         // factor slot -> stream index (-1: diagonal)
  isph::DevBuf<int> blkinfo;      // [nblocks][2] chunks in the L / U stream
  isph::DevBuf<double> dinv;      // [n] 1/d_i
  long long stream_chunks = 0;
  long long nnz = 0;
};

namespace isph {

// in-block entries of row i, compacted in order (rows of A are column-sorted)
__global__ void k_ilu_extract(int n, int B, const int *__restrict__ rowlen, const long long *__restrict__ slice_off,
                              const int *__restrict__ scol, const double *__restrict__ sval, int *__restrict__ fcol,
                              double *__restrict__ fval, int *__restrict__ flen, int *__restrict__ fdiag,
                              int *__restrict__ err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int blo = (i / B) * B, bhi = min(blo + B, n);
  const long long off = slice_off[i >> 6];
  const int lane = i & 63;
  int cnt = 0, dg = -1;
  for (int k = 0; k < rowlen[i]; ++k) {
    const long long p = sell_pos(off, lane, k);
    const int c = scol[p];
    if (c >= blo && c < bhi) {
      const long long q = sell_pos(off, lane, cnt);
      fcol[q] = c;
      fval[q] = sval[p];
      if (c == i) dg = cnt;
      ++cnt;
    }
  }
  flen[i] = cnt;
  fdiag[i] = dg;
  if (dg < 0) atomicOr(err, 1);  // structurally missing diagonal
}

// wave-uniform wait on an LDS flag written by another wave of this workgroup
// Bounded: after kSpinCap polls the wave gives up, raises the error word and
// carries on, so the grid always drains (a stuck dependency shows up as a
// failed preconditioner build, never as a hung GPU).
constexpr int kSpinCap = 1 << 20;
__device__ __forceinline__ void wait_flag(volatile int *flag, int *err) {
  int spins = 0;
  while (*flag == 0) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > kSpinCap) {
      atomicOr(err, 2);
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// IKJ ILU(0), one workgroup per block, one wave per row (round-robin), LDS:
//   done[B] flags, diag[B], per wave: wval[W], wcol[W], pos[B] (slot+1 of a column in the current row)
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_ilu_factor(int n, int B, int W, const long long *__restrict__ slice_off,
                                                           const int *__restrict__ fcol, double *__restrict__ fval,
                                                           const int *__restrict__ flen,
                                                           const int *__restrict__ fdiag, const int *__restrict__ fdst,
                                                           double *__restrict__ sv, double *__restrict__ dinv,
                                                           int *__restrict__ err) {
  extern __shared__ double lds_f[];
  double *diag = lds_f;                                   // [B]
  double *wval = diag + B;                                // [WAVES][W]
  int *wcol = reinterpret_cast<int *>(wval + WAVES * W);  // [WAVES][W]
  volatile int *done = wcol + WAVES * W;                  // [B]
  unsigned short *pos = reinterpret_cast<unsigned short *>(const_cast<int *>(done) + B);  // [WAVES][B]
  const int blo = blockIdx.x * B, bhi = min(blo + B, n), m = bhi - blo;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int t = threadIdx.x; t < m; t += blockDim.x) done[t] = 0;
  for (int t = threadIdx.x; t < WAVES * B; t += blockDim.x) pos[t] = 0;
  __syncthreads();
  double *mv = wval + wave * W;
  int *mc = wcol + wave * W;
  unsigned short *mp = pos + wave * B;
  for (int r = wave; r < m; r += WAVES) {
    const int i = blo + r;
    const long long off = slice_off[i >> 6];
    const int li = i & 63, len = flen[i], dg = fdiag[i];
    for (int s = lane; s < len; s += 64) {
      const long long p = sell_pos(off, li, s);
      const int c = fcol[p];
      mc[s] = c;
      mv[s] = fval[p];
      mp[c - blo] = (unsigned short)(s + 1);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int s = 0; s < dg; ++s) {  // lower entries, ascending column
      const int k = mc[s];
      wait_flag(&done[k - blo], err);
      const double lik = mv[s] / diag[k - blo];
      const long long koff = slice_off[k >> 6];
      const int kl = k & 63, kd = fdiag[k], klen = flen[k];
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) mv[s] = lik;
      for (int t = kd + 1 + lane; t < klen; t += 64) {  // U-row k
        const long long p = sell_pos(koff, kl, t);
        const int ps = mp[fcol[p] - blo];
        if (ps) mv[ps - 1] -= lik * fval[p];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    for (int s = lane; s < len; s += 64) {
      const long long p = sell_pos(off, li, s);
      fval[p] = mv[s];
      const int d = fdst[p];
      if (d >= 0) sv[d] = mv[s];  // triangular-solve stream
      mp[mc[s] - blo] = 0;
    }
    if (lane == 0) { diag[r] = mv[dg]; dinv[i] = 1.0 / mv[dg]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // row i (global) + diag (LDS) before the flag
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) done[r] = 1;
  }
}

// z = U^-1 D^-1 L^-1 r per block; one thread per row, sync-free through LDS flags.
__global__ __launch_bounds__(1024) void k_ilu_solve(int n, int B, const long long *__restrict__ slice_off,
                                                    const int *__restrict__ fcol, const double *__restrict__ fval,
                                                    const int *__restrict__ flen, const int *__restrict__ fdiag,
                                                    const double *__restrict__ r, double *__restrict__ z,
                                                    int *__restrict__ err) {
  extern __shared__ double lds_s[];
  volatile double *y = lds_s;                             // [B] L-solve result
  volatile double *x = lds_s + B;                         // [B] U-solve result
  volatile int *doneL = reinterpret_cast<volatile int *>(lds_s + 2 * B);
  volatile int *doneU = doneL + B;
  const int blo = blockIdx.x * B, bhi = min(blo + B, n), m = bhi - blo;
  const int t = threadIdx.x;
  if (t < B) { doneL[t] = 0; doneU[t] = 0; }
  __syncthreads();
  const bool active = t < m;
  const int i = blo + t;
  long long off = 0;
  int li = 0, len = 0, dg = 0;
  double sum = 0.0;
  if (active) {
    off = slice_off[i >> 6];
    li = i & 63;
    len = flen[i];
    dg = fdiag[i];
    sum = r[i];
  }
  // ---- forward: y_i = r_i - sum_{k<i} l_ik y_k
  {
    int p = 0;
    bool fin = !active;
    int c = 0;
    double v = 0.0;
    bool have = false;
    int polls = 0;
    while (!__all(fin)) {
      if (++polls > kSpinCap) {  // bounded: give up, flag the error, let the grid drain
        if (!fin) { atomicOr(err, 4); y[t] = sum; doneL[t] = 1; fin = true; }
        continue;
      }
      if (!fin) {
        while (p < dg) {
          if (!have) {
            const long long q = sell_pos(off, li, p);
            c = fcol[q] - blo;
            v = fval[q];
            have = true;
          }
          if (doneL[c] == 0) break;
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          sum -= v * y[c];
          have = false;
          ++p;
        }
        if (p == dg) {
          y[t] = sum;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          doneL[t] = 1;
          fin = true;
        }
      }
    }
  }
  // ---- backward: x_i = (y_i - sum_{j>i} u_ij x_j) / d_i
  {
    int p = len - 1;
    bool fin = !active;
    int c = 0;
    double v = 0.0, d = 1.0;
    bool have = false;
    if (active) d = fval[sell_pos(off, li, dg)];
    int polls = 0;
    while (!__all(fin)) {
      if (++polls > kSpinCap) {
        if (!fin) { atomicOr(err, 8); x[t] = sum; doneU[t] = 1; fin = true; }
        continue;
      }
      if (!fin) {
        while (p > dg) {
          if (!have) {
            const long long q = sell_pos(off, li, p);
            c = fcol[q] - blo;
            v = fval[q];
            have = true;
          }
          if (doneU[c] == 0) break;
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          sum -= v * x[c];
          have = false;
          --p;
        }
        if (p == dg) {
          const double xi = sum / d;
          x[t] = xi;
          z[i] = xi;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          doneU[t] = 1;
          fin = true;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Static schedule of the triangular solves (built per factorisation, on the GPU)
//
// For each block and each direction (L: dependencies = strictly-lower entries,
// U: strictly-upper) rows are levelled (lev = 1 + max lev of dependencies; level
// 0 rows need no work), sorted by level, and packed 8 rows per "step" with 8
// lanes per row.  A step is a run of chunks; chunk = 64 x (value, word) with
//   word = local column (11 bits) | local row << 11 (11 bits) | log2(G/8) << 22 | END << 31
// (END: last chunk of step; G = lanes per row in this step: 8 for a full step of
// 8 rows, 16/32/64 when a level leaves only <=4/2/1 rows for its last step)
// so the solve kernel is a pure coalesced stream: per chunk one fma against x in
// LDS, per step one 8-lane reduction and one LDS update.  No flags, no waiting.
// Stream capacity per block = kCapFactor x the block's sliced-ELL region + slack
// (+ prefetch pad); running out of it fails the build loudly.
constexpr unsigned kRowInvalid = 0x7FFu;
constexpr int kRowShift = 11, kGShift = 22;
constexpr unsigned kEndBit = 0x80000000u;
constexpr int kPrefetch = 8;    // chunks kept in flight per wave
constexpr int kPadChunks = 16;    // per-block tail pad so the prefetch never leaves the buffer
constexpr int kSlackChunks = 64;  // per-block slack on top of kCapFactor x the block's ELL region
constexpr int kCapFactor = 3;

__device__ __forceinline__ long long ilu_base_chunk(const long long *slice_off, int b, int B) {
  return kCapFactor * (slice_off[(long long)b * (B / 64)] >> 6) + (long long)(kPadChunks + kSlackChunks) * b;
}

__global__ __launch_bounds__(1024) void k_ilu_schedule(int n, int B, const long long *__restrict__ slice_off,
                                                       const int *__restrict__ fcol, const int *__restrict__ flen,
                                                       const int *__restrict__ fdiag, double *__restrict__ sv,
                                                       unsigned *__restrict__ sc, int *__restrict__ fdst,
                                                       int *__restrict__ blkinfo, int *__restrict__ err) {
  extern __shared__ int lds_i[];
  int *lev = lds_i;              // [B]
  int *cnt = lev + B;            // [B+1]
  int *stepstart = cnt + B + 1;  // [B+1]
  int *tmaxs = stepstart + B + 1;  // [B+1]
  int *choff = tmaxs + B + 1;    // [B+2]
  __shared__ int s_nlev, s_nsteps, s_nch;
  const int b = blockIdx.x, blo = b * B, bhi = min(blo + B, n), m = bhi - blo;
  const int t = threadIdx.x;
  const bool active = t < m;
  const int i = blo + t;
  const int nslices_blk = (m + 63) / 64;
  const long long region = slice_off[(long long)b * (B / 64) + nslices_blk] - slice_off[(long long)b * (B / 64)];
  const long long cap = kCapFactor * (region >> 6) + kSlackChunks;
  const long long base = ilu_base_chunk(slice_off, b, B);
  long long off = 0;
  int li = 0, len = 0, dg = 0;
  if (active) {
    off = slice_off[i >> 6];
    li = i & 63;
    len = flen[i];
    dg = fdiag[i];
    fdst[sell_pos(off, li, dg)] = -1;
  }
  int used = 0;  // chunks used so far in this block (L then U)
  for (int dir = 0; dir < 2; ++dir) {
    const int d0 = dir == 0 ? 0 : dg + 1, d1 = dir == 0 ? dg : len;  // dependency slots
    const int ndep = active ? d1 - d0 : 0;
    // ---- levels by relaxation (monotone, converges in #levels sweeps)
    if (t < B) lev[t] = 0;
    __syncthreads();
    for (int sweep = 0; sweep <= B; ++sweep) {
      int nl = 0;
      for (int e = 0; e < ndep; ++e) nl = max(nl, lev[fcol[sell_pos(off, li, d0 + e)] - blo] + 1);
      const int changed = active && nl != lev[t];
      if (!__syncthreads_or(changed)) break;
      if (active) lev[t] = nl;
      __syncthreads();
    }
    // ---- histogram of levels
    if (t == 0) s_nlev = 0;
    for (int k = t; k <= B; k += blockDim.x) { cnt[k] = 0; tmaxs[k] = 0; }
    __syncthreads();
    const int mylev = active ? lev[t] : 0;
    if (active) { atomicAdd(&cnt[mylev], 1); atomicMax(&s_nlev, mylev + 1); }
    __syncthreads();
    // deterministic rank inside the level (row order)
    int rk = 0;
    if (active && mylev > 0)
      for (int q = 0; q < t; ++q) rk += (lev[q] == mylev);
    if (t == 0) {
      int run = 0;
      stepstart[0] = 0;
      for (int l = 1; l < s_nlev; ++l) { stepstart[l] = run; run += (cnt[l] + 7) >> 3; }
      s_nsteps = run;
    }
    __syncthreads();
    const int step = (active && mylev > 0) ? stepstart[mylev] + (rk >> 3) : -1;
    const int g = rk & 7;
    const int rows_in_step = (active && mylev > 0) ? min(8, cnt[mylev] - ((rk >> 3) << 3)) : 8;
    const int gcode = rows_in_step > 4 ? 0 : rows_in_step > 2 ? 1 : rows_in_step > 1 ? 2 : 3;
    const int G = 8 << gcode;           // lanes per row in this step
    const int tneed = (ndep + G - 1) / G;  // >= 1 for scheduled rows
    if (step >= 0) atomicMax(&tmaxs[step], tneed);
    __syncthreads();
    if (t == 0) {
      int run = 0;
      for (int q = 0; q < s_nsteps; ++q) { choff[q] = run; run += tmaxs[q]; }
      choff[s_nsteps] = run;
      s_nch = run;
      blkinfo[2 * b + dir] = run;
      if ((long long)used + run > cap) atomicOr(err, 16);  // stream capacity exceeded
    }
    __syncthreads();
    if ((long long)used + s_nch > cap) return;  // uniform exit; host reports the error
    if (step >= 0) {
      const int tm = tmaxs[step];
      const unsigned gbits = (unsigned)gcode << kGShift;
      for (int c = 0; c < tm; ++c) {
        const long long chunk = base + used + choff[step] + c;
        const unsigned endbit = ((c == tm - 1) ? kEndBit : 0u) | gbits;
        for (int j = 0; j < G; ++j) {
          const int e = c * G + j;
          const long long idx = chunk * 64 + g * G + j;
          if (e < ndep) {
            const long long slot = sell_pos(off, li, d0 + e);
            sc[idx] = (unsigned)(fcol[slot] - blo) | ((unsigned)t << kRowShift) | endbit;
            fdst[slot] = (int)idx;
          } else {
            sc[idx] = ((unsigned)t << kRowShift) | endbit;
            sv[idx] = 0.0;
          }
        }
        if (g == 0)  // lanes of the groups this step does not use
          for (int gg = rows_in_step; gg < 64 / G; ++gg)
            for (int j = 0; j < G; ++j) {
              const long long idx = chunk * 64 + gg * G + j;
              sc[idx] = (kRowInvalid << kRowShift) | endbit;
              sv[idx] = 0.0;
            }
      }
    }
    used += s_nch;
    __syncthreads();
  }
}

// z = U^-1 D^-1 L^-1 r : one wave per block streams the block's chunk list.
// LDS per wave: y[B].  kPrefetch chunks (values + words) are kept in flight.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_ilu_solve_stream(int n, int B, int nblocks,
                                                                 const long long *__restrict__ slice_off,
                                                                 const double *__restrict__ sv,
                                                                 const unsigned *__restrict__ sc,
                                                                 const int *__restrict__ blkinfo,
                                                                 const int *__restrict__ flen,
                                                                 const int *__restrict__ fdiag,
                                                                 const double *__restrict__ dinv,
                                                                 const double *__restrict__ r, double *__restrict__ z) {
  extern __shared__ double lds_y[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * WAVES + wave;
  if (b >= nblocks) return;
  double *y = lds_y + (size_t)wave * B;
  const int blo = b * B, bhi = min(blo + B, n), m = bhi - blo;
  for (int t = lane; t < m; t += 64) y[t] = r[blo + t];
  const long long base = ilu_base_chunk(slice_off, b, B);
  const int nL = blkinfo[2 * b], nU = blkinfo[2 * b + 1];
  const double *__restrict__ pv = sv + base * 64 + lane;
  const unsigned *__restrict__ pc = sc + base * 64 + lane;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  double vq[kPrefetch];
  unsigned cq[kPrefetch];
  const int ntot = nL + nU;
#pragma unroll
  for (int u = 0; u < kPrefetch; ++u) {
    vq[u] = pv[(long long)u * 64];
    cq[u] = pc[(long long)u * 64];
  }
  double acc = 0.0;
  bool upper = false;
  for (int c0 = 0; c0 < ntot; c0 += kPrefetch) {
#pragma unroll
    for (int u = 0; u < kPrefetch; ++u) {
      const int c = c0 + u;
      const double v = vq[u];
      const unsigned cw = cq[u];
      vq[u] = pv[(long long)(c + kPrefetch) * 64];  // stays inside the padded buffer
      cq[u] = pc[(long long)(c + kPrefetch) * 64];
      if (c < ntot) {
        if (c == nL && !upper) {
          // switch to the U phase: rows without upper dependencies finish here
          upper = true;
          for (int t = lane; t < m; t += 64)
            if (fdiag[blo + t] == flen[blo + t] - 1) y[t] *= dinv[blo + t];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
        acc = fma(v, y[cw & 0x7FFu], acc);
        const unsigned cw0 = __builtin_amdgcn_readfirstlane(cw);
        if (cw0 & kEndBit) {
          const int gcode = (cw0 >> kGShift) & 3;  // wave-uniform
          double s = group8_sum(acc);
          if (gcode > 0) s += __shfl_xor(s, 8, 64);
          if (gcode > 1) s += __shfl_xor(s, 16, 64);
          if (gcode > 2) s += __shfl_xor(s, 32, 64);
          const unsigned row = (cw >> kRowShift) & 0x7FFu;
          if ((lane & ((8 << gcode) - 1)) == 0 && row != kRowInvalid) {
            const double yr = y[row] - s;
            y[row] = upper ? yr * dinv[blo + row] : yr;
          }
          acc = 0.0;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
  }
  if (!upper) {  // no U chunks at all (or empty stream): still scale the rows without upper deps
    for (int t = lane; t < m; t += 64)
      if (fdiag[blo + t] == flen[blo + t] - 1) y[t] *= dinv[blo + t];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  for (int t = lane; t < m; t += 64) z[blo + t] = y[t];
}

inline void ilu_destroy(isph_ilu *F) {
  if (!F) return;
  F->fcol.release(); F->flen.release(); F->fdiag.release(); F->err.release(); F->fval.release();
  F->sv.release(); F->sc.release(); F->fdst.release(); F->blkinfo.release(); F->dinv.release();
  delete F;
}

constexpr int kIluWaves = 8;

inline int ilu_create(isph_ctx *ctx, const isph_mat *A, int block_size, isph_ilu **out) {
  const Sell &S = A->S;
  ISPH_REQUIRE(block_size >= 64 && block_size <= 1024 && block_size % 64 == 0,
               "bjacobi-ilu0 block size must be a multiple of 64 in [64,1024]");
  isph_ilu *F = new isph_ilu();
  F->n = S.nrow; F->B = block_size; F->S = &S; F->wmax = S.wmax;
  F->nblocks = (S.nrow + block_size - 1) / block_size;
  const size_t stored = (size_t)(S.stored > 0 ? S.stored : 1), n1 = (size_t)(S.nrow > 0 ? S.nrow : 1);
  int rc = F->fcol.reserve(stored);
  if (rc == ISPH_SUCCESS) rc = F->fval.reserve(stored);
  if (rc == ISPH_SUCCESS) rc = F->flen.reserve(n1);
  if (rc == ISPH_SUCCESS) rc = F->fdiag.reserve(n1);
  if (rc == ISPH_SUCCESS) rc = F->err.reserve(1);
  F->stream_chunks = kCapFactor * (S.stored >> 6) + (long long)(kPadChunks + kSlackChunks) * (F->nblocks + 1) + kPrefetch;
  if (rc == ISPH_SUCCESS) rc = F->sv.reserve((size_t)F->stream_chunks * 64);
  if (rc == ISPH_SUCCESS) rc = F->sc.reserve((size_t)F->stream_chunks * 64);
  if (rc == ISPH_SUCCESS) rc = F->fdst.reserve(stored);
  if (rc == ISPH_SUCCESS) rc = F->blkinfo.reserve((size_t)2 * (F->nblocks > 0 ? F->nblocks : 1));
  if (rc == ISPH_SUCCESS) rc = F->dinv.reserve(n1);
  if (rc == ISPH_SUCCESS && (long long)F->stream_chunks * 64 >= 2147483647LL)
    rc = fail("ILU stream exceeds 32-bit indexing", __FILE__, __LINE__);
  if (rc == ISPH_SUCCESS && S.nrow > 0) {
    const int W = ((S.wmax + 63) / 64) * 64;
    const size_t lds = sizeof(double) * (size_t)block_size + (size_t)kIluWaves * W * 12 + 4 * (size_t)block_size +
                       2 * (size_t)kIluWaves * block_size + 16;
    if (lds > 160 * 1024) rc = fail("ILU factor kernel needs too much LDS for this row width", __FILE__, __LINE__);
    if (rc == ISPH_SUCCESS && hipMemsetAsync(F->err.p, 0, sizeof(int), ctx->stream) != hipSuccess)
      rc = fail("memset failed", __FILE__, __LINE__);
    if (rc == ISPH_SUCCESS) {
      hipLaunchKernelGGL(k_ilu_extract, dim3((S.nrow + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, S.nrow,
                         block_size, S.rowlen.p, S.slice_off.p, S.col.p, S.val.p, F->fcol.p, F->fval.p, F->flen.p,
                         F->fdiag.p, F->err.p);
      int herr = 0;
      if (hipMemcpyAsync(&herr, F->err.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipStreamSynchronize(ctx->stream) != hipSuccess)
        rc = fail("ILU extract failed", __FILE__, __LINE__);
      else if (herr)
        rc = fail("matrix row without a diagonal entry: ILU(0) undefined", __FILE__, __LINE__);
    }
    if (rc == ISPH_SUCCESS) {
      const size_t lds_s = sizeof(int) * (5 * (size_t)block_size + 8);
      hipLaunchKernelGGL(k_ilu_schedule, dim3(F->nblocks), dim3(block_size), lds_s, ctx->stream, S.nrow, block_size,
                         S.slice_off.p, F->fcol.p, F->flen.p, F->fdiag.p, F->sv.p, F->sc.p, F->fdst.p, F->blkinfo.p,
                         F->err.p);
      int herr = 0;
      if (hipGetLastError() != hipSuccess ||
          hipMemcpyAsync(&herr, F->err.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipStreamSynchronize(ctx->stream) != hipSuccess)
        rc = fail("ILU schedule kernel failed", __FILE__, __LINE__);
      else if (herr)  // the factor kernel must not run on a partial schedule
        rc = fail("ILU triangular-solve stream capacity exceeded", __FILE__, __LINE__);
    }
    if (rc == ISPH_SUCCESS) {
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_ilu_factor<kIluWaves>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        rc = fail("LDS attribute failed", __FILE__, __LINE__);
      else
        hipLaunchKernelGGL((k_ilu_factor<kIluWaves>), dim3(F->nblocks), dim3(kIluWaves * 64), lds, ctx->stream, S.nrow,
                           block_size, W, S.slice_off.p, F->fcol.p, F->fval.p, F->flen.p, F->fdiag.p, F->fdst.p, F->sv.p,
                           F->dinv.p, F->err.p);
      if (rc == ISPH_SUCCESS && hipGetLastError() != hipSuccess) rc = fail("ILU factor launch failed", __FILE__, __LINE__);
      if (rc == ISPH_SUCCESS) {
        int herr = 0;
        if (hipMemcpyAsync(&herr, F->err.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
          rc = fail("ILU factor kernel failed", __FILE__, __LINE__);
        else if (herr & 16)
          rc = fail("ILU triangular-solve stream capacity exceeded", __FILE__, __LINE__);
        else if (herr)
          rc = fail("ILU factor: dependency wait timed out", __FILE__, __LINE__);
      }
    }
  }
  if (rc != ISPH_SUCCESS) { ilu_destroy(F); return rc; }
  *out = F;
  return ISPH_SUCCESS;
}

inline int ilu_apply(isph_ctx *ctx, const isph_ilu *F, const double *r, double *z) {
  ISPH_REQUIRE(F != nullptr, "ILU factor is NULL");
  if (F->n == 0) return ISPH_SUCCESS;
  constexpr int WV = 4;
  const size_t lds = sizeof(double) * (size_t)F->B * WV;
  hipLaunchKernelGGL((k_ilu_solve_stream<WV>), dim3((F->nblocks + WV - 1) / WV), dim3(WV * 64), lds, ctx->stream, F->n,
                     F->B, F->nblocks, F->S->slice_off.p, F->sv.p, F->sc.p, F->blkinfo.p, F->flen.p, F->fdiag.p,
                     F->dinv.p, r, z);
  ISPH_CHECK_HIP(hipGetLastError());
  return ISPH_SUCCESS;
}

// number of stored factor entries (test/export helper: synchronous copy)
inline long long ilu_nnz(const isph_ilu *F) {
  if (!F || F->n == 0) return 0;
  std::vector<int> len((size_t)F->n);
  if (hipMemcpy(len.data(), F->flen.p, sizeof(int) * len.size(), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  long long s = 0;
  for (int v : len) s += v;
  return s;
}

// CSR export (device -> host) for parity tests
inline int ilu_export(isph_ctx *ctx, const isph_ilu *F, int *rowptr, int *colidx, double *val) {
  const Sell &S = *F->S;
  std::vector<int> len((size_t)F->n);
  std::vector<long long> so((size_t)S.nslices + 1);
  std::vector<int> col((size_t)S.stored);
  std::vector<double> v((size_t)S.stored);
  ISPH_CHECK_HIP(hipMemcpyAsync(len.data(), F->flen.p, sizeof(int) * len.size(), hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipMemcpyAsync(so.data(), S.slice_off.p, sizeof(long long) * so.size(), hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipMemcpyAsync(col.data(), F->fcol.p, sizeof(int) * col.size(), hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipMemcpyAsync(v.data(), F->fval.p, sizeof(double) * v.size(), hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  long long q = 0;
  for (int i = 0; i < F->n; ++i) {
    rowptr[i] = (int)q;
    const long long off = so[(size_t)(i >> 6)];
    for (int s = 0; s < len[(size_t)i]; ++s, ++q) {
      const long long p = off + (long long)(s >> 1) * 128 + (i & 63) * 2 + (s & 1);
      colidx[q] = col[(size_t)p];
      val[q] = v[(size_t)p];
    }
  }
  rowptr[F->n] = (int)q;
  return ISPH_SUCCESS;
}

}  // namespace isph
