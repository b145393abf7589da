// sell.hpp -- sliced-ELL (SELL-64, pair-interleaved) matrix + SpMV for gfx950.
//
// Layout (DESIGN.md "data layout"): rows are cut into slices of 64 (= one
// wavefront, lane == row).  A slice of width w (max row length, rounded up to
// even) stores its entries pair-interleaved:
//     entry (row lane, k)  ->  off + (k>>1)*128 + lane*2 + (k&1)
// so one wave-instruction loads 64 x double2 = 1 KiB contiguous values
// (global_load_dwordx4) and 64 x int2 = 512 B contiguous column ids.
// Padding entries carry value 0 and the row's own column.
// Replaces Epetra_CrsMatrix's CSR + Apply (ref: solver_lin.h:133).
#pragma once
#include "common.hpp"

namespace isph {

struct Sell {
  int nrow = 0, ncol = 0, nslices = 0, wmax = 0;
  long long nnz = 0, stored = 0;
  DevBuf<long long> slice_off;  // [nslices+1] entry offsets (multiples of 128)
  DevBuf<int> rowlen;           // [nrow]
  DevBuf<int> col;              // [stored]
  DevBuf<double> val;           // [stored]
  // windowed 16-bit copy of col for the SpMV (built on first use, see k_sell_compress_cols):
  // col = wtab[slice*64 + (c16 >> 10)] << 10 | (c16 & 1023).  c16_state: 0 not tried, 1 usable, -1 some slice
  // touches more than 64 windows of 1024 columns (then the 32-bit kernel stays in use)
  mutable DevBuf<unsigned short> col16;
  mutable DevBuf<int> wtab;
  mutable int c16_state = 0;
  void release() { slice_off.release(); rowlen.release(); col.release(); val.release(); col16.release(); wtab.release(); c16_state = 0; }
};

__device__ __forceinline__ unsigned amg_like_hash(int x) {
  unsigned h = (unsigned)x;
  h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16;
  return h;
}

__device__ __forceinline__ long long sell_pos(long long off, int lane, int k) {
  return off + (long long)(k >> 1) * 128 + lane * 2 + (k & 1);
}

// ---- CSR -> SELL ---------------------------------------------------------
template <class OFF>
__global__ void k_csr_rowlen_slicew(int nrow, const OFF *__restrict__ rowptr, int *__restrict__ rowlen,
                                    long long *__restrict__ slice_cnt) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  int len = 0;
  if (row < nrow) {
    len = (int)(rowptr[row + 1] - rowptr[row]);
    rowlen[row] = len;
  }
  const int w = wave_max_i32(len);
  if ((threadIdx.x & 63) == 0 && row < nrow) slice_cnt[row >> 6] = (long long)((w + 1) & ~1) * kSlice;
}

// generic: slice entry counts from a row-length array
__global__ void k_slicew_from_rowlen(int nrow, const int *__restrict__ rowlen, long long *__restrict__ slice_cnt) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  const int len = row < nrow ? rowlen[row] : 0;
  const int w = wave_max_i32(len);
  if ((threadIdx.x & 63) == 0 && row < nrow) slice_cnt[row >> 6] = (long long)((w + 1) & ~1) * kSlice;
}

// single-block exclusive scan of n long longs (n ~ 1e4..1e5): in[i] counts ->
// out[i] offsets, out[n] total.  In-place allowed.
__global__ void k_exclusive_scan_ll(int n, const long long *in, long long *out) {
  __shared__ long long wsum[16];
  __shared__ long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += blockDim.x) {
    const int i = base + threadIdx.x;
    const long long v = i < n ? in[i] : 0;
    long long s = v;  // inclusive scan inside the wave
    for (int o = 1; o < 64; o <<= 1) {
      const long long t = __shfl_up(s, o, 64);
      if ((threadIdx.x & 63) >= o) s += t;
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) wsum[wv] = s;
    __syncthreads();
    long long woff = 0;
    for (int k = 0; k < wv; ++k) woff += wsum[k];
    const long long c = carry;
    if (i < n) out[i] = c + woff + s - v;
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) carry = c + woff + s;
    __syncthreads();
  }
  if (threadIdx.x == 0) out[n] = carry;
}

// rowlen only (the host ingress knows the slice offsets from the host row pointers)
__global__ void k_csr_rowlen(int nrow, const int *__restrict__ rowptr, int *__restrict__ rowlen) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row < nrow) rowlen[row] = rowptr[row + 1] - rowptr[row];
}

// slices [slice_begin, slice_end); unsorted (optional) is raised when a row's columns are not ascending.
// Chunked image (host ingress, ingress.hpp; ck.stage != nullptr, colidx / cval unused): the entry range is cut into
// chunks [cstart[c], cstart[c+1]), and chunk c -- cnt entries -- lives at stage + 12 cstart[c] as it crossed the link in
// ONE copy:   [ cnt fp64 values | columns | first columns of the rows that start in the chunk ]
// with the columns either cnt 32-bit indices (mode bit 0; no first-column table) or cnt 16-bit differences to the
// previous column of the same row (mode bit 1: 10 instead of 12 bytes per entry on the link), the table then at the
// next 4-byte boundary, indexed by row - crow[c].  A lane walks its row front to back anyway, so the running sum of the
// differences costs nothing.
constexpr int kModeWords = 8;  // 512 chunks
struct CsrChunks {
  const char *stage;
  const long long *cstart;
  const int *crow;
  unsigned long long mode16[kModeWords];
  int nchunks;
};
struct ChunkView {
  const double *vals;
  const unsigned short *d16;
  const int *c32, *rowfirst;
  long long cbase, cend;
  int row0;
  bool m16;
};
__device__ __forceinline__ ChunkView chunk_view(const CsrChunks &ck, int ch) {
  ChunkView v;
  v.cbase = ck.cstart[ch];
  v.cend = ck.cstart[ch + 1];
  const long long cnt = v.cend - v.cbase;
  const char *base = ck.stage + 12 * v.cbase;
  v.vals = reinterpret_cast<const double *>(base);
  v.m16 = (ck.mode16[ch >> 6] >> (ch & 63)) & 1ull;
  v.d16 = reinterpret_cast<const unsigned short *>(base + 8 * cnt);
  v.c32 = reinterpret_cast<const int *>(base + 8 * cnt);
  v.rowfirst = reinterpret_cast<const int *>(base + ((10 * cnt + 3) & ~3ll));
  v.row0 = ck.crow[ch];
  return v;
}
template <class OFF>
__global__ void k_csr_to_sell(int nrow, const OFF *__restrict__ rowptr, const int *__restrict__ colidx,
                              const double *__restrict__ cval, const long long *__restrict__ slice_off,
                              int *__restrict__ scol, double *__restrict__ sval, int slice_begin, int slice_end,
                              int *__restrict__ unsorted, CsrChunks ck, const int *__restrict__ colren = nullptr, int nren = 0) {
  // colren (may be NULL): columns < nren are stored as colren[c] -- the library's own row numbering for a host CSR whose
  // rows arrive in that numbering but whose columns still carry the caller's (ingress.hpp, IngressGather); the rows then
  // come out unsorted and the caller sorts the slice range afterwards
  const int slice = slice_begin + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int row = slice * kSlice + lane;
  if (slice >= slice_end) return;
  const long long off = slice_off[slice];
  const int w = (int)((slice_off[slice + 1] - off) >> 6);
  OFF beg = 0;
  int len = 0;
  if (row < nrow) {
    beg = rowptr[row];
    len = (int)(rowptr[row + 1] - beg);
  }
  const bool chunked = ck.stage != nullptr;
  int ch = 0;
  ChunkView cv{};
  int padcol = 0;  // padding repeats a column the row already reads (always in range, also for rectangular operators)
  if (len > 0) {
    if (chunked) {  // chunk of the row's first entry
      int lo = 0, hi = ck.nchunks - 1;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (ck.cstart[mid] <= (long long)beg) lo = mid; else hi = mid - 1;
      }
      ch = lo;
      cv = chunk_view(ck, ch);
      padcol = cv.m16 ? cv.rowfirst[row - cv.row0] : cv.c32[(long long)beg - cv.cbase];
    } else {
      padcol = colidx[beg];
    }
  }
  const int first_col = padcol;            // the row's first column as it was sent (16-bit mode adds differences to it)
  if (colren && padcol < nren) padcol = colren[padcol];
  int prev = -1;
  bool desc = false;
  for (int k = 0; k < w; ++k) {
    const long long p = sell_pos(off, lane, k);
    if (k < len) {
      const long long q = (long long)beg + k;
      int c;
      double v;
      if (!chunked) {
        c = colidx[q];
        v = cval[q];
      } else {
        if (q >= cv.cend) cv = chunk_view(ck, ++ch);
        const long long e = q - cv.cbase;
        v = cv.vals[e];
        if (!cv.m16) c = cv.c32[e];
        else if (k == 0) c = first_col;
        else c = prev + (int)cv.d16[e];
      }
      desc = desc || c < prev;
      prev = c;
      scol[p] = (colren && c < nren) ? colren[c] : c;
      sval[p] = v;
    } else {
      scol[p] = padcol;
      sval[p] = 0.0;
    }
  }
  if (unsorted != nullptr && __ballot(desc) != 0ull && lane == 0) atomicOr(unsorted, 1);
}

// ---- row sort: columns ascending inside every row (Epetra OptimizeStorage order)
// One workgroup per slice.  R rows at a time are staged through LDS (coalesced
// k-pair segments), ranked by counting (all lanes read the same LDS word ->
// broadcast), written to a second LDS image in sorted order and streamed back.
// LDS: 24 bytes * R * Ws.
__global__ __launch_bounds__(kBlock) void k_sell_sort_rows(int nrow, int nslices, int R, int Ws,
                                                           const int *__restrict__ rowlen,
                                                           const long long *__restrict__ slice_off,
                                                           int *__restrict__ scol, double *__restrict__ sval, int slice0 = 0) {
  extern __shared__ double lds_raw[];
  const int slice = slice0 + blockIdx.x;   // a ranged launch sorts slices [slice0, slice0 + gridDim.x)
  if (slice >= nslices) return;
  double *valA = lds_raw;                 // [R][Ws]
  double *valB = valA + (size_t)R * Ws;   // [R][Ws]
  int *colA = reinterpret_cast<int *>(valB + (size_t)R * Ws);
  int *colB = colA + (size_t)R * Ws;
  const long long off = slice_off[slice];
  const int w = (int)((slice_off[slice + 1] - off) >> 6);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int r0 = 0; r0 < kSlice; r0 += R) {
    const int total = R * w;  // w is even
    for (int e = threadIdx.x; e < total; e += kBlock) {
      const int kk = e & 1, r = (e >> 1) % R, kp = (e >> 1) / R;
      const long long g = off + (long long)kp * 128 + (r0 + r) * 2 + kk;
      colA[r * Ws + 2 * kp + kk] = scol[g];
      valA[r * Ws + 2 * kp + kk] = sval[g];
    }
    __syncthreads();
    for (int r = wave; r < R; r += kBlock / kWave) {
      const int row = slice * kSlice + r0 + r;
      const int len = row < nrow ? rowlen[row] : 0;
      for (int k = lane; k < w; k += kWave) {
        int dst = k;  // padding keeps its slot
        if (k < len) {
          const int c = colA[r * Ws + k];
          int rank = 0;
          for (int q = 0; q < len; ++q) {
            const int cq = colA[r * Ws + q];
            rank += (cq < c) || (cq == c && q < k);
          }
          dst = rank;
        }
        colB[r * Ws + dst] = colA[r * Ws + k];
        valB[r * Ws + dst] = valA[r * Ws + k];
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < total; e += kBlock) {
      const int kk = e & 1, r = (e >> 1) % R, kp = (e >> 1) / R;
      const long long g = off + (long long)kp * 128 + (r0 + r) * 2 + kk;
      scol[g] = colB[r * Ws + 2 * kp + kk];
      sval[g] = valB[r * Ws + 2 * kp + kk];
    }
    __syncthreads();
  }
}

// The same sort fused with a symmetric permutation of the matrix (order.hpp: the library's own row numbering for a
// matrix that arrived as a host CSR in the caller's atom order): slice s of the result holds the rows perm[64 s ..] of the
// source, their owned columns renamed through iperm (ghost columns >= nrow keep their number), every row sorted by its
// new columns.  The load phase gathers from the source rows (12 B per entry wherever they lie), the rest is
// k_sell_sort_rows; one pass over both matrices instead of a copy and a sort.
__global__ __launch_bounds__(kBlock) void k_sell_permute_sort(int nrow, int nslices, int R, int Ws,
                                                              const int *__restrict__ perm, const int *__restrict__ iperm,
                                                              const int *__restrict__ rowlen0,
                                                              const long long *__restrict__ slice_off0,
                                                              const int *__restrict__ scol0, const double *__restrict__ sval0,
                                                              const int *__restrict__ rowlen,
                                                              const long long *__restrict__ slice_off,
                                                              int *__restrict__ scol, double *__restrict__ sval) {
  extern __shared__ double lds_raw[];
  __shared__ long long s_off0[kSlice];
  __shared__ int s_lane0[kSlice], s_len[kSlice];
  const int slice = blockIdx.x;
  if (slice >= nslices) return;
  double *valA = lds_raw;                 // [R][Ws]
  double *valB = valA + (size_t)R * Ws;   // [R][Ws]
  int *colA = reinterpret_cast<int *>(valB + (size_t)R * Ws);
  int *colB = colA + (size_t)R * Ws;
  const long long off = slice_off[slice];
  const int w = (int)((slice_off[slice + 1] - off) >> 6);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x < kSlice) {
    const int row = slice * kSlice + threadIdx.x;
    const int src = row < nrow ? perm[row] : 0;
    s_off0[threadIdx.x] = slice_off0[src >> 6];
    s_lane0[threadIdx.x] = src & 63;
    s_len[threadIdx.x] = row < nrow ? rowlen0[src] : 0;
  }
  __syncthreads();
  for (int r0 = 0; r0 < kSlice; r0 += R) {
    const int total = R * w;  // w is even
    for (int e = threadIdx.x; e < total; e += kBlock) {
      const int kk = e & 1, r = (e >> 1) % R, kp = (e >> 1) / R;
      const int k = 2 * kp + kk, row = slice * kSlice + r0 + r;
      int c = row < nrow ? row : 0;        // padding: the row's own column, value 0
      double v = 0.0;
      if (k < s_len[r0 + r]) {
        const long long g0 = sell_pos(s_off0[r0 + r], s_lane0[r0 + r], k);
        c = scol0[g0];
        v = sval0[g0];
        if (c < nrow) c = iperm[c];
      }
      colA[r * Ws + k] = c;
      valA[r * Ws + k] = v;
    }
    __syncthreads();
    for (int r = wave; r < R; r += kBlock / kWave) {
      const int len = s_len[r0 + r];
      for (int k = lane; k < w; k += kWave) {
        int dst = k;  // padding keeps its slot
        if (k < len) {
          const int c = colA[r * Ws + k];
          int rank = 0;
          for (int q = 0; q < len; ++q) {
            const int cq = colA[r * Ws + q];
            rank += (cq < c) || (cq == c && q < k);
          }
          dst = rank;
        }
        colB[r * Ws + dst] = colA[r * Ws + k];
        valB[r * Ws + dst] = valA[r * Ws + k];
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < total; e += kBlock) {
      const int kk = e & 1, r = (e >> 1) % R, kp = (e >> 1) / R;
      const long long g = off + (long long)kp * 128 + (r0 + r) * 2 + kk;
      scol[g] = colB[r * Ws + 2 * kp + kk];
      sval[g] = valB[r * Ws + 2 * kp + kk];
    }
    __syncthreads();
  }
}

// ---- SELL -> CSR (export for tests; unsorted within the row) -------------
__global__ void k_sell_to_csr(int nrow, const int *__restrict__ rowlen, const long long *__restrict__ slice_off,
                              const int *__restrict__ scol, const double *__restrict__ sval,
                              const long long *__restrict__ rowptr, int *__restrict__ colidx,
                              double *__restrict__ cval) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= nrow) return;
  const long long off = slice_off[row >> 6];
  const int lane = row & 63;
  const long long beg = rowptr[row];
  for (int k = 0; k < rowlen[row]; ++k) {
    const long long p = sell_pos(off, lane, k);
    colidx[beg + k] = scol[p];
    cval[beg + k] = sval[p];
  }
}

// rows [row0, row0 + n) only, row pointers relative to the range (ranged export for host-side checks at sizes where
// the whole matrix does not fit a 32-bit CSR)
__global__ void k_sell_rows_to_csr(int row0, int n, const int *__restrict__ rowlen, const long long *__restrict__ slice_off,
                                   const int *__restrict__ scol, const double *__restrict__ sval,
                                   const long long *__restrict__ rowptr, int *__restrict__ colidx,
                                   double *__restrict__ cval) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int row = row0 + t;
  const long long off = slice_off[row >> 6];
  const int lane = row & 63;
  const long long beg = rowptr[t];
  for (int k = 0; k < rowlen[row]; ++k) {
    const long long p = sell_pos(off, lane, k);
    colidx[beg + k] = scol[p];
    cval[beg + k] = sval[p];
  }
}

// ---- SpMV ----------------------------------------------------------------
// One wavefront per slice, lane == row.  UNROLL pair-columns are issued
// back-to-back so each lane keeps UNROLL 16-B value loads, UNROLL 8-B index
// loads and 2*UNROLL x-gathers in flight.  Production: UNROLL=8 with
// non-temporal matrix loads (the matrix is streamed exactly once per SpMV and
// must not displace x from the XCD's L2): 0.213 ms vs 0.235 ms (UNROLL=4,
// default cache policy) on the 1M-row bench matrix.  HBM-bound: 12 B/entry streamed,
// x gathered through the XCD-local L2 (xcd_remap keeps an XCD on one
// contiguous row range).  Optionally accumulates per-slice partials of y.n
// for the PoissonProjection (ref: solver_lin.h:131-140).
// LIST: the launch covers the slices slice_list[0..nslices) (interior / boundary split of a matrix with a halo: the
// interior slices have no ghost column and run while the halo exchange is in flight).  GHOST: columns >= nrow are
// read from the ghost buffer xg the exchange received into (no copy of x into an extended vector).
template <bool GHOST>
__device__ __forceinline__ double x_at(const double *__restrict__ x, const double *__restrict__ xg, int nrow, int c) {
  if (GHOST) return c < nrow ? x[c] : xg[c - nrow];
  return x[c];
}

template <int UNROLL, bool DOT, bool NT = false, bool LIST = false, bool GHOST = false>
__global__ __launch_bounds__(kBlock) void k_sell_spmv(int nrow, int nslices, int nblocks_padded,
                                                      const long long *__restrict__ slice_off,
                                                      const int *__restrict__ scol,
                                                      const double *__restrict__ sval,
                                                      const double *__restrict__ x, double *__restrict__ y,
                                                      const double *__restrict__ nvec,
                                                      double *__restrict__ dot_partial,
                                                      const int *__restrict__ slice_list = nullptr,
                                                      const double *__restrict__ xg = nullptr,
                                                      const double *badd = nullptr, double alpha = 1.0) {
  const int b = xcd_remap(blockIdx.x, nblocks_padded);
  int slice = b * (kBlock / kWave) + (threadIdx.x >> 6);
  if (slice >= nslices) return;
  if (LIST) slice = slice_list[slice];
  const int lane = threadIdx.x & 63;
  const long long off = slice_off[slice];
  const int npair = (int)((slice_off[slice + 1] - off) >> 7);
  const double2 *__restrict__ v = reinterpret_cast<const double2 *>(sval + off) + lane;
  const int2 *__restrict__ c = reinterpret_cast<const int2 *>(scol + off) + lane;
  double acc0 = 0.0, acc1 = 0.0;
  int q = 0;
  for (; q + UNROLL <= npair; q += UNROLL) {
    double2 vv[UNROLL];
    int2 cc[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (NT) {  // the matrix is read exactly once per SpMV: keep it from displacing x in L2
        vv[u].x = __builtin_nontemporal_load(&v[(q + u) * 64].x);
        vv[u].y = __builtin_nontemporal_load(&v[(q + u) * 64].y);
        cc[u].x = __builtin_nontemporal_load(&c[(q + u) * 64].x);
        cc[u].y = __builtin_nontemporal_load(&c[(q + u) * 64].y);
      } else {
        vv[u] = v[(q + u) * 64];
        cc[u] = c[(q + u) * 64];
      }
    }
    double xa[UNROLL], xb[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      xa[u] = x_at<GHOST>(x, xg, nrow, cc[u].x);
      xb[u] = x_at<GHOST>(x, xg, nrow, cc[u].y);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      acc0 = fma(vv[u].x, xa[u], acc0);
      acc1 = fma(vv[u].y, xb[u], acc1);
    }
  }
  for (; q < npair; ++q) {
    const double2 vv = v[q * 64];
    const int2 cc = c[q * 64];
    acc0 = fma(vv.x, x_at<GHOST>(x, xg, nrow, cc.x), acc0);
    acc1 = fma(vv.y, x_at<GHOST>(x, xg, nrow, cc.y), acc1);
  }
  const int row = slice * kSlice + lane;
  double r = acc0 + acc1;
  // epilogue of the AMG cycle: y = badd + alpha (A x) -- residual b - A x, corrections x += P e, r -= (A P) e -- with the
  // bits of the product followed by the vector kernel it replaces (badd may be y itself)
  if (alpha != 1.0) r *= alpha;
  if (badd != nullptr && row < nrow) r += badd[row];
  if (row < nrow) y[row] = r;
  if (DOT) {
    const double d = wave_sum(row < nrow ? r * nvec[row] : 0.0);
    if (lane == 0) dot_partial[slice] = d;
  }
}

// ---- 16-bit windowed column indices ------------------------------------------------------------------
// The SpMV streams 8 B of value and 4 B of column per entry; a slice's rows are 64 neighbouring particles whose
// columns fall into a few dozen aligned windows of 1024 indices (the bricks around them), so a per-slice table of
// <= 64 window numbers plus 6+10 bits per entry carries the same information in 2 B: 10 B instead of 12 B per
// entry for the HBM-bound kernel.  One wave per slice builds the table (LDS set with atomicCAS) and re-encodes.
__global__ __launch_bounds__(kBlock) void k_sell_compress_cols(int slice_begin, int slice_end,
                                                               const long long *__restrict__ slice_off,
                                                               const int *__restrict__ scol,
                                                               unsigned short *__restrict__ c16, int *__restrict__ wtab,
                                                               int *__restrict__ fail) {
  __shared__ int tab[kBlock / kWave][64];
  __shared__ int cnt[kBlock / kWave];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slice = slice_begin + blockIdx.x * (kBlock / kWave) + wave;
  if (slice >= slice_end) return;
  tab[wave][lane] = -1;
  if (lane == 0) cnt[wave] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const long long off = slice_off[slice];
  const long long nent = slice_off[slice + 1] - off;
  bool bad = false;
  // Position e of a slice belongs to row (e % 128) / 2 (pair-interleaved layout), so a lane that walks e = lane, lane + 64,
  // .. alternates between TWO rows and sees every second entry of each, in ascending column order: the window of an
  // entry is nearly always the window of the lane's entry two trips earlier.  The lane remembers one (window, slot) per
  // row; the table in LDS (compare-and-swap while it is built, probing when it is read) is only asked when the window
  // changes (0.53 -> see docs/kernels_detail.md; the table and the codes are what they were).
  int last_win[2] = {-2, -2}, last_slot[2] = {0, 0};
  for (long long e0 = lane; e0 < nent; e0 += 128) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long long e = e0 + 64 * h;
      if (e >= nent) break;
      const int win = scol[off + e] >> 10;
      if (win == last_win[h]) continue;
      int slot = (int)(amg_like_hash(win) & 63u);
      for (int tries = 0; tries < 64; ++tries) {
        int old = __atomic_load_n(&tab[wave][slot], __ATOMIC_RELAXED);
        if (old == -1) old = atomicCAS(&tab[wave][slot], -1, win);
        if (old == -1 || old == win) break;
        slot = (slot + 1) & 63;
        if (tries == 63) bad = true;
      }
      last_win[h] = win;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (__ballot(bad)) { if (lane == 0) atomicOr(fail, 1); return; }
  wtab[(long long)slice * 64 + lane] = tab[wave][lane];
  last_win[0] = last_win[1] = -2;
  for (long long e0 = lane; e0 < nent; e0 += 128) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long long e = e0 + 64 * h;
      if (e >= nent) break;
      const int c = scol[off + e], win = c >> 10;
      if (win != last_win[h]) {
        int slot = (int)(amg_like_hash(win) & 63u);
        while (tab[wave][slot] != win) slot = (slot + 1) & 63;
        last_win[h] = win;
        last_slot[h] = slot;
      }
      c16[off + e] = (unsigned short)((last_slot[h] << 10) | (c & 1023));
    }
  }
}

template <int UNROLL, bool DOT, bool LIST = false, bool GHOST = false>
__global__ __launch_bounds__(kBlock) void k_sell_spmv16(int nrow, int nslices, int nblocks_padded,
                                                        const long long *__restrict__ slice_off,
                                                        const unsigned short *__restrict__ c16,
                                                        const int *__restrict__ wtab,
                                                        const double *__restrict__ sval,
                                                        const double *__restrict__ x, double *__restrict__ y,
                                                        const double *__restrict__ nvec,
                                                        double *__restrict__ dot_partial,
                                                        const int *__restrict__ slice_list = nullptr,
                                                        const double *__restrict__ xg = nullptr,
                                                        const double *badd = nullptr, double alpha = 1.0) {
  __shared__ int tab[kBlock / kWave][64];
  const int b = xcd_remap(blockIdx.x, nblocks_padded);
  const int wave = threadIdx.x >> 6;
  int slice = b * (kBlock / kWave) + wave;
  if (slice >= nslices) return;
  if (LIST) slice = slice_list[slice];
  const int lane = threadIdx.x & 63;
  tab[wave][lane] = wtab[(long long)slice * 64 + lane] << 10;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int *__restrict__ tw = tab[wave];
  const long long off = slice_off[slice];
  const int npair = (int)((slice_off[slice + 1] - off) >> 7);
  const double2 *__restrict__ v = reinterpret_cast<const double2 *>(sval + off) + lane;
  const unsigned *__restrict__ c = reinterpret_cast<const unsigned *>(c16 + off) + lane;  // two 16-bit columns
  double acc0 = 0.0, acc1 = 0.0;
  int q = 0;
  for (; q + UNROLL <= npair; q += UNROLL) {
    double2 vv[UNROLL];
    unsigned cc[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      vv[u].x = __builtin_nontemporal_load(&v[(q + u) * 64].x);
      vv[u].y = __builtin_nontemporal_load(&v[(q + u) * 64].y);
      cc[u] = __builtin_nontemporal_load(&c[(q + u) * 64]);
    }
    double xa[UNROLL], xb[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const unsigned lo = cc[u] & 0xffffu, hi = cc[u] >> 16;
      xa[u] = x_at<GHOST>(x, xg, nrow, tw[lo >> 10] | (int)(lo & 1023u));
      xb[u] = x_at<GHOST>(x, xg, nrow, tw[hi >> 10] | (int)(hi & 1023u));
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      acc0 = fma(vv[u].x, xa[u], acc0);
      acc1 = fma(vv[u].y, xb[u], acc1);
    }
  }
  for (; q < npair; ++q) {
    const double2 vv = v[q * 64];
    const unsigned cc = c[q * 64];
    const unsigned lo = cc & 0xffffu, hi = cc >> 16;
    acc0 = fma(vv.x, x_at<GHOST>(x, xg, nrow, tw[lo >> 10] | (int)(lo & 1023u)), acc0);
    acc1 = fma(vv.y, x_at<GHOST>(x, xg, nrow, tw[hi >> 10] | (int)(hi & 1023u)), acc1);
  }
  const int row = slice * kSlice + lane;
  double r = acc0 + acc1;
  // epilogue of the AMG cycle: y = badd + alpha (A x) -- residual b - A x, corrections x += P e, r -= (A P) e -- with the
  // bits of the product followed by the vector kernel it replaces (badd may be y itself)
  if (alpha != 1.0) r *= alpha;
  if (badd != nullptr && row < nrow) r += badd[row];
  if (row < nrow) y[row] = r;
  if (DOT) {
    const double d = wave_sum(row < nrow ? r * nvec[row] : 0.0);
    if (lane == 0) dot_partial[slice] = d;
  }
}

// y_k = A x_k for NV vectors in one sweep of the matrix (the Helmholtz system has one right-hand side per velocity
// component and one matrix: SURVEY section 7 step 9).  Same accumulation order per vector as k_sell_spmv16 -- the
// results are bit-identical to NV separate products.
struct SpmmVecs {
  const double *x[4];
  double *y[4];
};
template <int NV>
__global__ __launch_bounds__(kBlock) void k_sell_spmm16(int nrow, int nslices, int nblocks_padded,
                                                        const long long *__restrict__ slice_off,
                                                        const unsigned short *__restrict__ c16,
                                                        const int *__restrict__ wtab, const double *__restrict__ sval,
                                                        SpmmVecs V) {
  constexpr int UNROLL = 4;  // 426 us for three vectors of the 1 M-row matrix (8: 467 us); one vector alone takes 190 us:
                             // the NV gathers per entry, not the matrix stream, set the pace
  __shared__ int tab[kBlock / kWave][64];
  const int b = xcd_remap(blockIdx.x, nblocks_padded);
  const int wave = threadIdx.x >> 6;
  const int slice = b * (kBlock / kWave) + wave;
  if (slice >= nslices) return;
  const int lane = threadIdx.x & 63;
  tab[wave][lane] = wtab[(long long)slice * 64 + lane] << 10;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int *__restrict__ tw = tab[wave];
  const long long off = slice_off[slice];
  const int npair = (int)((slice_off[slice + 1] - off) >> 7);
  const double2 *__restrict__ v = reinterpret_cast<const double2 *>(sval + off) + lane;
  const unsigned *__restrict__ c = reinterpret_cast<const unsigned *>(c16 + off) + lane;
  double acc0[NV], acc1[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) { acc0[k] = 0.0; acc1[k] = 0.0; }
  int q = 0;
  for (; q + UNROLL <= npair; q += UNROLL) {
    double2 vv[UNROLL];
    int ca[UNROLL], cb[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      vv[u].x = __builtin_nontemporal_load(&v[(q + u) * 64].x);
      vv[u].y = __builtin_nontemporal_load(&v[(q + u) * 64].y);
      const unsigned cc = __builtin_nontemporal_load(&c[(q + u) * 64]);
      const unsigned lo = cc & 0xffffu, hi = cc >> 16;
      ca[u] = tw[lo >> 10] | (int)(lo & 1023u);
      cb[u] = tw[hi >> 10] | (int)(hi & 1023u);
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      double xa[UNROLL], xb[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) { xa[u] = V.x[k][ca[u]]; xb[u] = V.x[k][cb[u]]; }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        acc0[k] = fma(vv[u].x, xa[u], acc0[k]);
        acc1[k] = fma(vv[u].y, xb[u], acc1[k]);
      }
    }
  }
  for (; q < npair; ++q) {
    const double2 vv = v[q * 64];
    const unsigned cc = c[q * 64];
    const unsigned lo = cc & 0xffffu, hi = cc >> 16;
    const int ca = tw[lo >> 10] | (int)(lo & 1023u), cb = tw[hi >> 10] | (int)(hi & 1023u);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      acc0[k] = fma(vv.x, V.x[k][ca], acc0[k]);
      acc1[k] = fma(vv.y, V.x[k][cb], acc1[k]);
    }
  }
  const int row = slice * kSlice + lane;
  if (row < nrow) {
#pragma unroll
    for (int k = 0; k < NV; ++k) V.y[k][row] = acc0[k] + acc1[k];
  }
}

// invdiag[row] = 1 / A(row,row)  (point-Jacobi debug preconditioner)
__global__ void k_sell_inv_diag(int nrow, const int *__restrict__ rowlen, const long long *__restrict__ slice_off,
                                const int *__restrict__ scol, const double *__restrict__ sval,
                                double *__restrict__ invdiag) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= nrow) return;
  const long long off = slice_off[row >> 6];
  double d = 0.0;
  for (int k = 0; k < rowlen[row]; ++k) {
    const long long p = sell_pos(off, row & 63, k);
    if (scol[p] == row) d += sval[p];
  }
  invdiag[row] = d != 0.0 ? 1.0 / d : 1.0;
}

// flag[slice] = 1 when the slice reads a ghost column (column >= nrow): boundary slice of a matrix with a halo
__global__ __launch_bounds__(kBlock) void k_sell_flag_ghost_slices(int nrow, int nslices, const long long *__restrict__ slice_off,
                                                                   const int *__restrict__ scol, int *__restrict__ flag) {
  const int slice = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  if (slice >= nslices) return;
  const int lane = threadIdx.x & 63;
  const long long off = slice_off[slice], nent = slice_off[slice + 1] - off;
  bool g = false;
  for (long long e = lane; e < nent; e += 64) g = g || (scol[off + e] >= nrow);
  const unsigned long long m = __ballot(g);
  if (lane == 0) flag[slice] = m != 0ull;
}

inline int spmv_grid(int nslices, int *nblocks_padded) {
  const int nb = (nslices + 3) / 4;
  const int per = (nb + kXcd - 1) / kXcd;
  *nblocks_padded = per * kXcd;
  return per * kXcd;
}

}  // namespace isph
