// ilu.hpp -- block-Jacobi ILU(0) on the GPU.
//
// Replaces PrecondWrapper_Ifpack::create() + Belos::EpetraPrecOp::Apply
// (ref: precond_ifpack.h:52-75, solver_lin_belos.h:147-156) for the setting
// "Precond Type"=ILU, "Overlap Level"=0, "fact: level-of-fill"=0 with one
// additive-Schwarz subdomain per block of B consecutive rows (what Ifpack gives
// with one MPI rank per block).  Entries that couple different blocks are
// dropped, exactly like the off-rank columns of the reference's local matrix.
//
// Data layout
//   factor F : block-local row-major storage.  Block b owns the entry range of
//              its rows' sliced-ELL region in A (capacity >= its in-block nnz),
//              row i keeps its in-block entries contiguously at
//              frp[i] .. frp[i]+flen[i], columns ascending, fdiag[i] = slot of the
//              diagonal.  Strict-L (unit diagonal implied), D and strict-U live in
//              one pattern == A's in-block pattern.  Row-major because both the
//              factorisation (U-row k is read by every row that depends on k) and
//              the schedule walk single rows: a row is 3-6 cache lines here, ~60
//              in the lane-interleaved ELL layout.
//   stream   : the triangular solves never touch F; they stream a per-block list
//              of 64-entry chunks laid out in execution order (see k_ilu_schedule).
//
// Pipeline per build:  k_ilu_extract -> k_ilu_schedule -> k_ilu_factor
// Apply:               k_ilu_solve_stream (one wave per block)
#pragma once
#include "core.hpp"
#include "sell.hpp"

struct isph_ilu {
  int n = 0, B = 0, nblocks = 0, wmax = 0;
  const isph::Sell *S = nullptr;  // geometry shared with A (A must outlive the factor)
  isph::DevBuf<long long> frp;    // [n] first entry of row i in fcol/fval
  isph::DevBuf<int> fcol, flen, fdiag, err;
  isph::DevBuf<double> fval;
  isph::DevBuf<double> sv;        // chunk stream values   [nchunks*64]
  isph::DevBuf<unsigned> sc;      // chunk stream words
  isph::DevBuf<int> fdst;         // factor entry -> stream index (-1: diagonal)
  isph::DevBuf<int> blkinfo;      // [nblocks][2] chunks in the L / U stream
  isph::DevBuf<int> llev;         // [n] L-level of every row (level-synchronous factorisation)
  isph::DevBuf<double> dinv;      // [n] 1/d_i
  long long stream_chunks = 0;
  int capf = 0, slack = 0;  // stream capacity rule in force (see kCapFactorSafe)
};

namespace isph {

// ---------------------------------------------------------------------------
// extract: one workgroup (B threads) per block.  Thread t counts the in-block
// entries of row blo+t (A rows are column-sorted, lane==row reads are
// coalesced), a block scan places the rows back to back inside the block's
// region, then every thread copies its entries.
__global__ __launch_bounds__(1024) void k_ilu_extract(int n, int B, const int *__restrict__ rowlen,
                                                      const long long *__restrict__ slice_off,
                                                      const int *__restrict__ scol, const double *__restrict__ sval,
                                                      long long *__restrict__ frp, int *__restrict__ fcol,
                                                      double *__restrict__ fval, int *__restrict__ flen,
                                                      int *__restrict__ fdiag, int *__restrict__ err) {
  __shared__ int wsum[16];
  const int b = blockIdx.x, blo = b * B, bhi = min(blo + B, n);
  const int t = threadIdx.x, i = blo + t;
  const bool active = i < bhi;
  const long long base = slice_off[(long long)b * (B / 64)];
  long long off = 0;
  int lane = 0, len = 0, cnt = 0, dg = -1;
  if (active) {
    off = slice_off[i >> 6];
    lane = i & 63;
    len = rowlen[i];
    for (int k = 0; k < len; ++k) {
      const int c = scol[sell_pos(off, lane, k)];
      if (c >= blo && c < bhi) {
        if (c == i) dg = cnt;
        ++cnt;
      }
    }
  }
  // exclusive scan of cnt over the workgroup
  int s = cnt;
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(s, o, 64);
    if ((t & 63) >= o) s += v;
  }
  if ((t & 63) == 63) wsum[t >> 6] = s;
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < (t >> 6); ++w) woff += wsum[w];
  const long long start = base + woff + s - cnt;
  if (active) {
    frp[i] = start;
    flen[i] = cnt;
    fdiag[i] = dg;
    if (dg < 0) atomicOr(err, 1);  // structurally missing diagonal
    long long q = start;
    for (int k = 0; k < len; ++k) {
      const long long p = sell_pos(off, lane, k);
      const int c = scol[p];
      if (c >= blo && c < bhi) {
        fcol[q] = c;
        fval[q] = sval[p];
        ++q;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Static schedule of the triangular solves (built per factorisation, on the GPU)
//
// For each block and each direction (L: dependencies = strictly-lower entries,
// U: strictly-upper) rows are levelled (lev = 1 + max lev of dependencies; level
// 0 rows need no work), sorted by level, and packed up to 8 rows per "step" with
// G lanes per row.  A step is a run of chunks; chunk = 64 x (value, word) with
//   word = local column (11 bits) | local row << 11 (11 bits) | log2(G/8) << 22 | END << 31
// (END: last chunk of step; G = lanes per row in this step: 8 for a full step of
// 8 rows, 16/32/64 when a level leaves only <=4/2/1 rows for its last step)
// so the solve kernel is a pure coalesced stream: per chunk one fma against x in
// LDS, per step one G-lane reduction and one LDS update.  No flags, no waiting.
// Stream capacity per block = kCapFactor x the block's sliced-ELL region + slack
// (+ prefetch pad); running out of it fails the build loudly.
constexpr unsigned kRowInvalid = 0x7FFu;
constexpr int kRowShift = 11, kGShift = 22;
constexpr unsigned kEndBit = 0x80000000u;
constexpr int kPrefetch = 8;      // chunks kept in flight per wave
constexpr int kPadChunks = 16;    // per-block tail pad so the prefetch never leaves the buffer
// Capacity is (capf x the block's ELL region + slack) chunks per block.  First attempt: capf 3, slack 64
// (the 3-D production matrices use ~1.75 x).  A step of r rows uses max_r ceil(ndep/G) <= sum(ndep)/8 + 1
// chunks and there are at most 2 m steps, so capf 8 with slack 2 B can never overflow: the build falls back
// to that bound when the first attempt reports an overflow (small 2-D blocks with one row per level).
constexpr int kSlackChunks = 64;
constexpr int kCapFactor = 3;
constexpr int kCapFactorSafe = 8;

__device__ __forceinline__ long long ilu_base_chunk(const long long *slice_off, int b, int B, int capf, int slack) {
  return capf * (slice_off[(long long)b * (B / 64)] >> 6) + (long long)(kPadChunks + slack) * b;
}

__global__ __launch_bounds__(1024) void k_ilu_schedule(int n, int B, const long long *__restrict__ slice_off,
                                                       const long long *__restrict__ frp,
                                                       const int *__restrict__ fcol, const int *__restrict__ flen,
                                                       const int *__restrict__ fdiag, double *__restrict__ sv,
                                                       unsigned *__restrict__ sc, int *__restrict__ fdst,
                                                       int *__restrict__ blkinfo, int *__restrict__ llev,
                                                       int ccap, int capf, int slack, int *__restrict__ err) {
  extern __shared__ int lds_i[];
  int *lev = lds_i;              // [B]
  int *cnt = lev + B;            // [B+1]
  int *stepstart = cnt + B + 1;  // [B+1]
  int *tmaxs = stepstart + B + 1;  // [B+1]
  int *choff = tmaxs + B + 1;    // [B+2]
  // optional cache of every row's in-block local columns, [slot][row] (conflict-free across rows)
  unsigned short *ccol = reinterpret_cast<unsigned short *>(choff + B + 2);  // [ccap][B], ccap == 0: read global
  __shared__ int s_nlev, s_nsteps, s_nch;
  const int b = blockIdx.x, blo = b * B, bhi = min(blo + B, n), m = bhi - blo;
  const int t = threadIdx.x;
  const bool active = t < m;
  const int i = blo + t;
  const int nslices_blk = (m + 63) / 64;
  const long long region = slice_off[(long long)b * (B / 64) + nslices_blk] - slice_off[(long long)b * (B / 64)];
  const long long cap = capf * (region >> 6) + slack;
  const long long base = ilu_base_chunk(slice_off, b, B, capf, slack);
  long long rp = 0;
  int len = 0, dg = 0;
  if (active) {
    rp = frp[i];
    len = flen[i];
    dg = fdiag[i];
    fdst[rp + dg] = -1;
    if (ccap > 0)
      for (int s = 0; s < len; ++s) ccol[s * B + t] = (unsigned short)(fcol[rp + s] - blo);
  }
  __syncthreads();
  int used = 0;  // chunks used so far in this block (L then U)
  for (int dir = 0; dir < 2; ++dir) {
    const int d0 = dir == 0 ? 0 : dg + 1, d1 = dir == 0 ? dg : len;  // dependency slots
    const int ndep = active ? d1 - d0 : 0;
    // ---- levels by relaxation (monotone, converges in #levels sweeps)
    if (t < B) lev[t] = 0;
    __syncthreads();
    for (int sweep = 0; sweep <= B; ++sweep) {
      int nl = 0;
      if (ccap > 0) {
        // 8 independent column/level look-ups in flight (a plain loop serialises two LDS latencies per dependency)
        for (int e0 = 0; e0 < ndep; e0 += 8) {
          int lv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int e = min(e0 + u, ndep - 1);
            lv[u] = lev[ccol[(d0 + e) * B + t]];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) nl = max(nl, lv[u] + 1);
        }
      } else {
        for (int e = 0; e < ndep; ++e) nl = max(nl, lev[fcol[rp + d0 + e] - blo] + 1);
      }
      const int changed = active && nl != lev[t];
      if (!__syncthreads_or(changed)) break;
      if (active) lev[t] = nl;
      __syncthreads();
    }
    if (dir == 0 && active) llev[i] = lev[t];  // the numeric factorisation walks the same levels
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
            const long long slot = rp + d0 + e;
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

// ---------------------------------------------------------------------------
// IKJ ILU(0), one workgroup per block, level-synchronous: rows of one L-level are
// independent, the WAVES waves take them round-robin, one __syncthreads per
// level.  Every row a row depends on is final before its level starts, so the
// U-rows it eliminates with are plain contiguous loads that are prefetched
// kFacPrefetch steps ahead (no flags, no waiting inside a row).  LDS: diag[B],
// frp/flen/fdiag[B], level order + offsets, per wave a row image (values,
// columns) and a column->slot map.
constexpr int kFacPrefetch = 4;
constexpr int kIluWaves = 8;

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_ilu_factor(int n, int B, int W, const long long *__restrict__ frp,
                                                           const int *__restrict__ fcol, double *__restrict__ fval,
                                                           const int *__restrict__ flen,
                                                           const int *__restrict__ fdiag, const int *__restrict__ fdst,
                                                           const int *__restrict__ llev, double *__restrict__ sv,
                                                           double *__restrict__ dinv) {
  extern __shared__ double lds_f[];
  double *diag = lds_f;                                   // [B] 1/d_k of the finished rows
  double *wval = diag + B;                                // [WAVES][W]
  long long *rpL = reinterpret_cast<long long *>(wval + WAVES * W);  // [B] frp
  int *wcol = reinterpret_cast<int *>(rpL + B);           // [WAVES][W]
  int *lenL = wcol + WAVES * W;                           // [B] flen
  int *dgL = lenL + B;                                    // [B] fdiag
  int *lstart = dgL + B;                                  // [B+2] first position of every level in `order`
  int *lcnt = lstart + B + 2;                             // [B+1]
  int *levs = lcnt + B + 1;                               // [B] L-level of every row
  unsigned short *order = reinterpret_cast<unsigned short *>(levs + B);  // [B] rows sorted by (level,row)
  unsigned short *pos = order + B;                        // [WAVES][B] slot+1 of a column in the current row
  __shared__ int s_nlev;
  const int blo = blockIdx.x * B, bhi = min(blo + B, n), m = bhi - blo;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) s_nlev = 0;
  for (int t = threadIdx.x; t <= B; t += blockDim.x) lcnt[t] = 0;
  for (int t = threadIdx.x; t < WAVES * B; t += blockDim.x) pos[t] = 0;
  __syncthreads();
  for (int t = threadIdx.x; t < m; t += blockDim.x) {
    rpL[t] = frp[blo + t];
    lenL[t] = flen[blo + t];
    dgL[t] = fdiag[blo + t];
    const int l = llev[blo + t];
    levs[t] = l;
    atomicAdd(&lcnt[l], 1);
    atomicMax(&s_nlev, l + 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int l = 0; l < s_nlev; ++l) { lstart[l] = run; run += lcnt[l]; }
    lstart[s_nlev] = run;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < m; t += blockDim.x) {  // deterministic position: rows of a level in row order
    const int l = levs[t];
    int rk = 0;
    for (int q = 0; q < t; ++q) rk += (levs[q] == l);
    order[lstart[l] + rk] = (unsigned short)t;
  }
  __syncthreads();
  double *mv = wval + wave * W;
  int *mc = wcol + wave * W;
  unsigned short *mp = pos + wave * B;
  const int nlev = s_nlev;
  for (int l = 0; l < nlev; ++l) {
    for (int q = lstart[l] + wave; q < lstart[l + 1]; q += WAVES) {
      const int r = order[q];
      const int i = blo + r;
      const long long rp = rpL[r];
      const int len = lenL[r], dg = dgL[r];
      for (int s = lane; s < len; s += 64) {
        const int c = fcol[rp + s] - blo;
        mc[s] = c;
        mv[s] = fval[rp + s];
        mp[c] = (unsigned short)(s + 1);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // software-pipelined elimination: the U-row of step s+kFacPrefetch is
      // requested while step s is applied (one entry per lane per 64 columns)
      int psq[kFacPrefetch];      // slot+1 in row i of the prefetched U-row entry (0: not in the pattern)
      double pvq[kFacPrefetch];   // its value
      double pdq[kFacPrefetch];   // 1/d_k of that step
      auto request = [&](int s, int &ps, double &pvv, double &pdv) {
        ps = 0;
        pvv = 0.0;
        pdv = 0.0;
        if (s < dg) {
          const int k = mc[s];
          pdv = diag[k];
          const int t = dgL[k] + 1 + lane;
          if (t < lenL[k]) {
            const long long p = rpL[k] + t;
            ps = mp[fcol[p] - blo];
            pvv = fval[p];
          }
        }
      };
#pragma unroll
      for (int u = 0; u < kFacPrefetch; ++u) request(u, psq[u], pvq[u], pdq[u]);
      for (int s0 = 0; s0 < dg; s0 += kFacPrefetch) {
#pragma unroll
        for (int u = 0; u < kFacPrefetch; ++u) {
          const int s = s0 + u;
          if (s < dg) {
            const int ps = psq[u];
            const double lik = mv[s] * pdq[u];   // l_ik = a_ik / d_k (reciprocal stored once per pivot)
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) mv[s] = lik;
            if (ps) mv[ps - 1] -= lik * pvq[u];
            const int k = mc[s];
            // U-rows wider than one wave (rare: > 64 in-block upper entries)
            for (int t = dgL[k] + 1 + 64 + lane; t < lenL[k]; t += 64) {
              const long long p = rpL[k] + t;
              const int ps2 = mp[fcol[p] - blo];
              if (ps2) mv[ps2 - 1] -= lik * fval[p];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            request(s + kFacPrefetch, psq[u], pvq[u], pdq[u]);
          }
        }
      }
      for (int s = lane; s < len; s += 64) {
        fval[rp + s] = mv[s];
        const int d = fdst[rp + s];
        if (d >= 0) sv[d] = mv[s];  // triangular-solve stream
        mp[mc[s]] = 0;
      }
      if (lane == 0) {
        const double rd = 1.0 / mv[dg];
        diag[r] = rd;
        dinv[i] = rd;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();  // level l (global rows + diag) complete before level l+1 reads it
  }
}

// ---------------------------------------------------------------------------
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
                                                                 const double *__restrict__ r, double *__restrict__ z,
                                                                 int capf, int slack) {
  extern __shared__ double lds_y[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * WAVES + wave;
  if (b >= nblocks) return;
  double *y = lds_y + (size_t)wave * B;
  const int blo = b * B, bhi = min(blo + B, n), m = bhi - blo;
  for (int t = lane; t < m; t += 64) y[t] = r[blo + t];
  const long long base = ilu_base_chunk(slice_off, b, B, capf, slack);
  const int nL = blkinfo[2 * b], nU = blkinfo[2 * b + 1];
  const double *__restrict__ pv = sv + base * 64 + lane;
  const unsigned *__restrict__ pc = sc + base * 64 + lane;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  double vq[kPrefetch];
  unsigned cq[kPrefetch];
  const int ntot = nL + nU;
#pragma unroll
  for (int u = 0; u < kPrefetch; ++u) {  // non-temporal: the stream is read once per apply
    vq[u] = __builtin_nontemporal_load(&pv[(long long)u * 64]);
    cq[u] = __builtin_nontemporal_load(&pc[(long long)u * 64]);
  }
  double acc = 0.0;
  bool upper = false;
  for (int c0 = 0; c0 < ntot; c0 += kPrefetch) {
#pragma unroll
    for (int u = 0; u < kPrefetch; ++u) {
      const int c = c0 + u;
      const double v = vq[u];
      const unsigned cw = cq[u];
      vq[u] = __builtin_nontemporal_load(&pv[(long long)(c + kPrefetch) * 64]);  // stays inside the padded buffer
      cq[u] = __builtin_nontemporal_load(&pc[(long long)(c + kPrefetch) * 64]);
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

// ---------------------------------------------------------------------------
inline void ilu_destroy(isph_ilu *F) {
  if (!F) return;
  F->frp.release(); F->fcol.release(); F->flen.release(); F->fdiag.release(); F->err.release(); F->fval.release();
  F->sv.release(); F->sc.release(); F->fdst.release(); F->blkinfo.release(); F->dinv.release(); F->llev.release();
  delete F;
}

inline int ilu_check_err(isph_ctx *ctx, isph_ilu *F, const char *what, bool *overflow = nullptr) {
  int herr = 0;
  if (hipGetLastError() != hipSuccess ||
      hipMemcpyAsync(&herr, F->err.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess)
    return fail(what, __FILE__, __LINE__);
  if (herr & 1) return fail("matrix row without a diagonal entry: ILU(0) undefined", __FILE__, __LINE__);
  if ((herr & 16) && overflow) { *overflow = true; return ISPH_SUCCESS; }
  if (herr & 16) return fail("ILU triangular-solve stream capacity exceeded", __FILE__, __LINE__);
  if (herr) return fail(what, __FILE__, __LINE__);
  return ISPH_SUCCESS;
}

inline int ilu_create(isph_ctx *ctx, const isph_mat *A, int block_size, isph_ilu **out) {
  const Sell &S = A->S;
  ISPH_REQUIRE(block_size >= 64 && block_size <= 1024 && block_size % 64 == 0,
               "bjacobi-ilu0 block size must be a multiple of 64 in [64,1024]");
  isph_ilu *F = new isph_ilu();
  F->n = S.nrow; F->B = block_size; F->S = &S; F->wmax = S.wmax;
  F->nblocks = (S.nrow + block_size - 1) / block_size;
  const size_t stored = (size_t)(S.stored > 0 ? S.stored : 1), n1 = (size_t)(S.nrow > 0 ? S.nrow : 1);
  F->capf = kCapFactor; F->slack = kSlackChunks;
  auto size_stream = [&]() {
    F->stream_chunks = F->capf * (S.stored >> 6) + (long long)(kPadChunks + F->slack) * (F->nblocks + 1) + kPrefetch;
    int r = F->sv.reserve((size_t)F->stream_chunks * 64);
    if (r == ISPH_SUCCESS) r = F->sc.reserve((size_t)F->stream_chunks * 64);
    if (r == ISPH_SUCCESS && (long long)F->stream_chunks * 64 >= 2147483647LL)
      r = fail("ILU stream exceeds 32-bit indexing", __FILE__, __LINE__);
    return r;
  };
  int rc = F->fcol.reserve(stored);
  if (rc == ISPH_SUCCESS) rc = F->fval.reserve(stored);
  if (rc == ISPH_SUCCESS) rc = F->fdst.reserve(stored);
  if (rc == ISPH_SUCCESS) rc = F->frp.reserve(n1);
  if (rc == ISPH_SUCCESS) rc = F->flen.reserve(n1);
  if (rc == ISPH_SUCCESS) rc = F->fdiag.reserve(n1);
  if (rc == ISPH_SUCCESS) rc = F->dinv.reserve(n1);
  if (rc == ISPH_SUCCESS) rc = F->llev.reserve(n1);
  if (rc == ISPH_SUCCESS) rc = F->err.reserve(1);
  if (rc == ISPH_SUCCESS) rc = size_stream();
  if (rc == ISPH_SUCCESS) rc = F->blkinfo.reserve((size_t)2 * (F->nblocks > 0 ? F->nblocks : 1));
  if (rc == ISPH_SUCCESS && S.nrow > 0) {
    const int W = ((S.wmax + 63) / 64) * 64;
    const size_t Bz = (size_t)block_size;
    const size_t lds_f = 8 * Bz + (size_t)kIluWaves * W * 12 + 8 * Bz + 4 * (5 * Bz + 4) + 2 * Bz + 2 * (size_t)kIluWaves * Bz + 16;
    if (lds_f > 160 * 1024) rc = fail("ILU factor kernel needs too much LDS for this row width", __FILE__, __LINE__);
    if (rc == ISPH_SUCCESS && hipMemsetAsync(F->err.p, 0, sizeof(int), ctx->stream) != hipSuccess)
      rc = fail("memset failed", __FILE__, __LINE__);
    if (rc == ISPH_SUCCESS) {
      hipLaunchKernelGGL(k_ilu_extract, dim3(F->nblocks), dim3(block_size), 0, ctx->stream, S.nrow, block_size,
                         S.rowlen.p, S.slice_off.p, S.col.p, S.val.p, F->frp.p, F->fcol.p, F->fval.p, F->flen.p,
                         F->fdiag.p, F->err.p);
      size_t lds_s = sizeof(int) * (5 * Bz + 8);
      int ccap = S.wmax;  // cache every row's local columns in LDS when it fits
      if (lds_s + 2 * (size_t)ccap * Bz > 150 * 1024) ccap = 0;
      lds_s += 2 * (size_t)ccap * Bz;
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_ilu_schedule), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_s) != hipSuccess)
        rc = fail("LDS attribute failed", __FILE__, __LINE__);
      for (int attempt = 0; attempt < 2 && rc == ISPH_SUCCESS; ++attempt) {
        hipLaunchKernelGGL(k_ilu_schedule, dim3(F->nblocks), dim3(block_size), lds_s, ctx->stream, S.nrow, block_size,
                           S.slice_off.p, F->frp.p, F->fcol.p, F->flen.p, F->fdiag.p, F->sv.p, F->sc.p, F->fdst.p,
                           F->blkinfo.p, F->llev.p, ccap, F->capf, F->slack, F->err.p);
        // the factor kernel must not run on a partial schedule: check now (one sync per build)
        bool overflow = false;
        rc = ilu_check_err(ctx, F, "ILU extract/schedule kernel failed", attempt == 0 ? &overflow : nullptr);
        if (!overflow) break;
        F->capf = kCapFactorSafe; F->slack = 2 * block_size;  // proven bound, see kCapFactorSafe
        rc = size_stream();
        if (rc == ISPH_SUCCESS && hipMemsetAsync(F->err.p, 0, sizeof(int), ctx->stream) != hipSuccess)
          rc = fail("memset failed", __FILE__, __LINE__);
      }
    }
    if (rc == ISPH_SUCCESS) {
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_ilu_factor<kIluWaves>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f) != hipSuccess)
        rc = fail("LDS attribute failed", __FILE__, __LINE__);
      else
        hipLaunchKernelGGL((k_ilu_factor<kIluWaves>), dim3(F->nblocks), dim3(kIluWaves * 64), lds_f, ctx->stream, S.nrow,
                           block_size, W, F->frp.p, F->fcol.p, F->fval.p, F->flen.p, F->fdiag.p, F->fdst.p, F->llev.p,
                           F->sv.p, F->dinv.p);
      if (rc == ISPH_SUCCESS && hipGetLastError() != hipSuccess) rc = fail("ILU factor launch failed", __FILE__, __LINE__);
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
                     F->dinv.p, r, z, F->capf, F->slack);
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
  std::vector<long long> rp((size_t)F->n);
  std::vector<int> col((size_t)S.stored);
  std::vector<double> v((size_t)S.stored);
  ISPH_CHECK_HIP(hipMemcpyAsync(len.data(), F->flen.p, sizeof(int) * len.size(), hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipMemcpyAsync(rp.data(), F->frp.p, sizeof(long long) * rp.size(), hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipMemcpyAsync(col.data(), F->fcol.p, sizeof(int) * col.size(), hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipMemcpyAsync(v.data(), F->fval.p, sizeof(double) * v.size(), hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  long long q = 0;
  for (int i = 0; i < F->n; ++i) {
    rowptr[i] = (int)q;
    for (int s = 0; s < len[(size_t)i]; ++s, ++q) {
      colidx[q] = col[(size_t)(rp[(size_t)i] + s)];
      val[q] = v[(size_t)(rp[(size_t)i] + s)];
    }
  }
  rowptr[F->n] = (int)q;
  return ISPH_SUCCESS;
}

}  // namespace isph
