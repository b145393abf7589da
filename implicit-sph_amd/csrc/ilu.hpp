// ilu.hpp -- block-Jacobi ILU(k) on the GPU.
//
// Replaces PrecondWrapper_Ifpack::create() + Belos::EpetraPrecOp::Apply
// (ref: precond_ifpack.h:52-75, solver_lin_belos.h:147-156) for the setting
// "Precond Type"=ILU, "Overlap Level"=0, "fact: level-of-fill"=k with one
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
//              one pattern == A's in-block pattern (k = 0) or the level-of-fill
//              pattern built by ilu_symbolic (k > 0; then the block regions are
//              the factor's own, boff[]).  Row-major because both the
//              factorisation (U-row k is read by every row that depends on k) and
//              the schedule walk single rows: a row is 3-6 cache lines here, ~60
//              in the lane-interleaved ELL layout.
//   stream   : the triangular solves never touch F; they stream a per-block list
//              of 64-entry chunks laid out in execution order (see k_ilu_schedule).
//
// Pipeline per build:  k_ilu_extract [-> k x (k_iluk_merge count, write)] -> k_ilu_schedule -> k_ilu_factor
// Apply:               k_ilu_solve_stream (one wave per block)
#pragma once
#include "core.hpp"
#include "sell.hpp"

struct isph_ilu {
  int n = 0, B = 0, nblocks = 0, wmax = 0;
  isph::DevBuf<long long> frp;    // [n] first entry of row i in fcol/fval
  isph::DevBuf<int> fcol, flen, fdiag, err;
  isph::DevBuf<double> fval;
  isph::DevBuf<double> sv;        // chunk stream values   [nchunks*64]
  isph::DevBuf<unsigned short> sc;  // chunk stream words (16 bit per entry)
  isph::DevBuf<unsigned char> si;   // one byte per chunk: END | need << 1
  isph::DevBuf<unsigned short> sperm;  // [nblocks][2][B] row -> position in the solve order of the L / U direction
  isph::DevBuf<int> fdst;         // factor entry -> stream index RELATIVE to the block's first stream entry (-1: diagonal)
  isph::DevBuf<int> blkinfo;      // [nblocks][4] chunks in the L / U stream, rows the L / U stream completes
  isph::DevBuf<int> llev;         // [n] L-level of every row (level-synchronous factorisation)
  isph::DevBuf<double> dinv;      // [n] 1/d_i
  isph::DevBuf<long long> boff;   // [nblocks+1] first factor entry of every block (multiples of 64)
  // exact stream sizing (large streams: a counting pass of the schedule, then the fill): 64 x the running chunk count of
  // the blocks, used in place of boff -- with capacity factor 1 and no slack -- wherever the stream is addressed
  isph::DevBuf<long long> sboff;
  bool exact = false;
  long long stream_entries = 0;  // exact mode: 64 x the chunks the schedule counted
  bool compact = false;          // exact mode, ILU(0) / Gauss-Seidel: the factor regions hold the in-block entries only
  const long long *stream_off() const { return exact ? sboff.p : boff.p; }
  isph::DevBuf<unsigned char> flev;  // level of fill of every factor entry (ILU(k) symbolic phase only)
  long long stream_chunks = 0;
  long long total = 0;      // entries reserved for the factor (ILU(0): A's sliced-ELL size; ILU(k): sum of the blocks)
  int fill = 0;             // level of fill
  int capf = 0, slack = 0;  // stream capacity rule in force (see kCapFactorSafe)
  // caller-defined subdomains (isph_prec_create_blocks): block b = rows bptr[b] .. bptr[b+1], at most B of them (B is then
  // the CAPACITY of a block: LDS sizes, threads per workgroup, stride of the per-block tables); NULL = B rows each
  isph::DevBuf<int> bptr;
  const int *blocks() const { return bptr.p; }
};

namespace isph {

// rows of block b: consecutive ranges of B rows, or the caller's table
__device__ __forceinline__ void ilu_block_rows(const int *__restrict__ bptr, int b, int B, int n, int &blo, int &bhi) {
  if (bptr) { blo = bptr[b]; bhi = bptr[b + 1]; }
  else { blo = b * B; bhi = min(blo + B, n); }
}

// ---------------------------------------------------------------------------
// extract: one workgroup (B threads) per block.  Thread t counts the in-block
// entries of row blo+t (A rows are column-sorted, lane==row reads are
// coalesced), a block scan places the rows back to back inside the block's
// region, then every thread copies its entries.
constexpr int kExtChunk = 8;  // entries of a row staged per round (dynamic LDS: waves x 64 rows x 8 x 12 B)
constexpr int kExtAhead = 8;  // slots of a row whose loads are issued together

__global__ __launch_bounds__(1024) void k_ilu_extract(int n, int B, const int *__restrict__ rowlen,
                                                      const long long *__restrict__ slice_off,
                                                      const int *__restrict__ scol, const double *__restrict__ sval,
                                                      long long *__restrict__ frp, int *__restrict__ fcol,
                                                      double *__restrict__ fval, int *__restrict__ flen,
                                                      int *__restrict__ fdiag, int *__restrict__ err, int b0,
                                                      const long long *__restrict__ base_off,
                                                      const int *__restrict__ bptr) {
  __shared__ int wsum[16];
  extern __shared__ double ext_lds[];
  const int nwv = blockDim.x >> 6;
  double (*stage_val)[64][kExtChunk] = reinterpret_cast<double (*)[64][kExtChunk]>(ext_lds);
  long long (*stage_start)[64] = reinterpret_cast<long long (*)[64]>(ext_lds + (size_t)nwv * 64 * kExtChunk);
  int (*stage_col)[64][kExtChunk] = reinterpret_cast<int (*)[64][kExtChunk]>(ext_lds + (size_t)nwv * 64 * (kExtChunk + 1));
  int (*stage_cnt)[64] = reinterpret_cast<int (*)[64]>(reinterpret_cast<int *>(ext_lds + (size_t)nwv * 64 * (kExtChunk + 1)) +
                                                       (size_t)nwv * 64 * kExtChunk);
  const int b = blockIdx.x + b0;  // b0: first block of a ranged launch
  int blo, bhi;
  ilu_block_rows(bptr, b, B, n, blo, bhi);
  const int t = threadIdx.x, i = blo + t;
  const bool active = i < bhi;
  // the block's rows go back to back into its region: A's own sliced-ELL region of these rows, or -- compact mode, large
  // operators -- a region of exactly the block's in-block entries (k_ilu_count_inblock)
  const long long base = base_off ? base_off[b] : slice_off[(long long)b * (B / 64)];
  long long off = 0;
  int lane = 0, len = 0, cnt = 0, dg = -1;
  if (active) {
    off = slice_off[i >> 6];
    lane = i & 63;
    len = rowlen[i];
    int prevc = -1;
    bool disorder = false;
    for (int k0 = 0; k0 < len; k0 += kExtAhead) {   // the loads of kExtAhead slots first, then the dependent counting
      int cq[kExtAhead];
#pragma unroll
      for (int u = 0; u < kExtAhead; ++u) cq[u] = scol[sell_pos(off, lane, k0 + u < len ? k0 + u : 0)];   // every slot loads: countable
#pragma unroll
      for (int u = 0; u < kExtAhead; ++u) {
        const int c = cq[u];
        if (k0 + u < len && c >= blo && c < bhi) {
          if (c <= prevc) disorder = true;  // every later stage relies on strictly ascending columns (one diagonal)
          prevc = c;
          if (c == i) dg = cnt;
          ++cnt;
        }
      }
    }
    if (disorder) atomicOr(err, 32);
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
  }
  // copy: the wave walks its 64 rows of A column slot by column slot (every load is one coalesced wave access of the
  // sliced-ELL image); a lane keeps the in-block entries of its row in an LDS buffer of kExtChunk entries, and as soon
  // as one lane's buffer is full the wave writes all buffers out together -- kExtChunk consecutive lanes write one
  // row's piece of the row-major destination (64-B / 32-B segments instead of 64 scattered 8-B stores).
  const int wv = t >> 6, ln = t & 63;
  int *scnt = &stage_cnt[wv][0];
  long long *sstart = &stage_start[wv][0];
  sstart[ln] = start;
  int got = 0;
  const int lenmax = wave_max_i32(len);
  int cq[kExtAhead];
  double vq[kExtAhead];
  for (int k = 0; k < lenmax; ++k) {
    const int u0 = k % kExtAhead;
    if (u0 == 0) {   // columns and values of the next kExtAhead slots are requested together (the values of entries outside
                     // the block travel in vain: 4 of 10 on the bench matrix, against a load per slot that waited for its column)
#pragma unroll
      for (int u = 0; u < kExtAhead; ++u) {
        const long long p = sell_pos(off, lane, k + u < len ? k + u : 0);   // every slot loads (a slot behind the row: its first)
        cq[u] = scol[p];
        vq[u] = sval[p];
      }
    }
    int c = -1;
    double v = 0.0;
#pragma unroll
    for (int u = 0; u < kExtAhead; ++u) if (u == u0) { c = cq[u]; v = vq[u]; }
    if (k < len) {
      if (c >= blo && c < bhi) {
        stage_col[wv][ln][got] = c;
        stage_val[wv][ln][got] = v;
        ++got;
      }
    }
    if (__ballot(got == kExtChunk) == 0 && k + 1 < lenmax) continue;  // wave-uniform
    scnt[ln] = got;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < kExtChunk; ++it) {
      const int e = ln + 64 * it, row = e / kExtChunk, slot = e % kExtChunk;
      if (slot < scnt[row]) {
        const long long q = sstart[row] + slot;
        fcol[q] = stage_col[wv][row][slot];
        fval[q] = stage_val[wv][row][slot];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    sstart[ln] += got;  // read again only behind the next flush's barrier
    got = 0;
  }
}

// in-block entries of every block, rounded up to a multiple of 64 (compact factor regions of large operators: a row of
// the 4 M x 749 operator of BASELINE configs[4] keeps ~15 % of its entries inside its 512-row block)
__global__ __launch_bounds__(1024) void k_ilu_count_inblock(int n, int B, const int *__restrict__ rowlen,
                                                            const long long *__restrict__ slice_off,
                                                            const int *__restrict__ scol, int *__restrict__ blktot,
                                                            const int *__restrict__ bptr) {
  __shared__ int s_sum;
  const int b = blockIdx.x;
  int blo, bhi;
  ilu_block_rows(bptr, b, B, n, blo, bhi);
  const int i = blo + threadIdx.x;
  if (threadIdx.x == 0) s_sum = 0;
  __syncthreads();
  int cnt = 0;
  if (i < bhi) {
    const long long off = slice_off[i >> 6];
    const int lane = i & 63, len = rowlen[i];
    for (int k = 0; k < len; ++k) {
      const int c = scol[sell_pos(off, lane, k)];
      cnt += c >= blo && c < bhi;
    }
  }
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s_sum, cnt);
  __syncthreads();
  if (threadIdx.x == 0) blktot[b] = (s_sum + 63) & ~63;
}

// ---------------------------------------------------------------------------
// ILU(k) symbolic phase ("fact: level-of-fill" = k > 0, precond_ifpack.h:35; Ifpack_IlukGraph rule
// lev(i,j) = min over k < min(i,j) of lev(i,k) + lev(k,j) + 1, entries of A at level 0, kept while <= k).
//
// The sequential definition walks the rows in order because row i merges the FINAL patterns of the rows it
// eliminates with.  Here the same levels are reached as the fixed point of
//     lev_{t+1}(i,j) = min( lev_t(i,j), min_{k<min(i,j)} lev_t(i,k) + lev_t(k,j) + 1 ),   lev_0 = pattern of A:
// a level-l entry is the sum of two entries of lower level, so after t sweeps every entry of level <= t is exact
// (induction on the level; the recursion is well-founded in min(i,j), so the fixed point is unique) and K sweeps
// give the ILU(K) pattern.  Inside a sweep all rows are independent: one wave per row scatters the row into a
// dense level array of the block's columns in LDS, merges the upper parts of the rows in its lower part (the
// previous sweep's pattern, read only), and compacts the columns with level <= K.  ILU(1) is a single sweep.
// Every sweep runs twice: once to count (per-row lengths, in-block offsets, block totals), once to write.
constexpr int kSymWaves = 8;
constexpr int kSymPrefetch = 4;
constexpr int kLevNone = 255;

template <bool FILL>
__global__ __launch_bounds__(kSymWaves * 64) void k_iluk_merge(
    int n, int B, int K, const long long *__restrict__ frp, const int *__restrict__ fcol,
    const unsigned char *__restrict__ flev, const int *__restrict__ flen, const int *__restrict__ fdiag,
    const double *__restrict__ fval, int *__restrict__ nlen, int *__restrict__ inoff, int *__restrict__ blktot,
    int *__restrict__ maxlen, const long long *__restrict__ boff, long long *__restrict__ nrp,
    int *__restrict__ ncol, unsigned char *__restrict__ nlev, int *__restrict__ ndiag, double *__restrict__ nval,
    const int *__restrict__ bptr) {
  extern __shared__ double lds_m[];
  long long *rpL = reinterpret_cast<long long *>(lds_m);       // [B]
  double *dvals = lds_m + B;                                   // [waves][B] values of the row, by local column
  int *lenL = reinterpret_cast<int *>(dvals + kSymWaves * B);  // [B]
  int *dgL = lenL + B;                                         // [B]
  int *cntL = dgL + B;                                         // [B] new row lengths (count pass)
  int *dlev = cntL + B;                                        // [waves][B] level of the row's entry, by local column
  unsigned short *mcol = reinterpret_cast<unsigned short *>(dlev + kSymWaves * B);  // [waves][B] the row's columns
  unsigned char *mlev = reinterpret_cast<unsigned char *>(mcol + kSymWaves * B);    // [waves][B] ... and levels
  const int b = blockIdx.x;
  int blo, bhi;
  ilu_block_rows(bptr, b, B, n, blo, bhi);  // B rows each, or the caller's table (B is then the capacity)
  const int m = bhi - blo;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int t = threadIdx.x; t < m; t += blockDim.x) {
    rpL[t] = frp[blo + t];
    lenL[t] = flen[blo + t];
    dgL[t] = fdiag[blo + t];
  }
  __syncthreads();
  int *dl = dlev + wave * B;
  double *dv = dvals + wave * B;
  unsigned short *mc = mcol + wave * B;
  unsigned char *ml = mlev + wave * B;
  const long long bbase = FILL ? boff[b] : 0;
  for (int r = wave; r < m; r += kSymWaves) {
    const int i = blo + r;
    const long long rp = rpL[r];
    const int len = lenL[r], dg = dgL[r];
    for (int t = lane; t < m; t += 64) { dl[t] = kLevNone; dv[t] = 0.0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int s = lane; s < len; s += 64) {
      const int c = fcol[rp + s] - blo;
      const int l = flev[rp + s];
      dl[c] = l;
      dv[c] = fval[rp + s];
      mc[s] = (unsigned short)c;
      ml[s] = (unsigned char)l;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // merge the upper parts of the rows k in the lower part of row i (prefetched like k_ilu_factor)
    int pcq[kSymPrefetch], plq[kSymPrefetch], paq[kSymPrefetch];
    auto request = [&](int s, int &pc, int &pl, int &pa) {
      pc = -1; pl = 0; pa = kLevNone;
      if (s < dg) {
        const int k = mc[s];
        pa = ml[s];
        const int t = dgL[k] + 1 + lane;
        if (pa < K && t < lenL[k]) {
          const long long p = rpL[k] + t;
          pc = fcol[p];
          pl = flev[p];
        }
      }
    };
#pragma unroll
    for (int u = 0; u < kSymPrefetch; ++u) request(u, pcq[u], plq[u], paq[u]);
    for (int s0 = 0; s0 < dg; s0 += kSymPrefetch) {
#pragma unroll
      for (int u = 0; u < kSymPrefetch; ++u) {
        const int s = s0 + u;
        if (s < dg) {
          if (pcq[u] >= 0) {
            const int nl = paq[u] + plq[u] + 1;
            const int j = pcq[u] - blo;
            if (nl <= K && nl < dl[j]) dl[j] = nl;
          }
          if (paq[u] < K) {  // upper parts wider than one wave
            const int k = mc[s];
            for (int t = dgL[k] + 1 + 64 + lane; t < lenL[k]; t += 64) {
              const long long p = rpL[k] + t;
              const int nl = paq[u] + (int)flev[p] + 1;
              const int j = fcol[p] - blo;
              if (nl <= K && nl < dl[j]) dl[j] = nl;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          request(s + kSymPrefetch, pcq[u], plq[u], paq[u]);
        }
      }
    }
    // compact the columns with level <= K (ascending by construction)
    int run = 0;
    const long long q0 = FILL ? bbase + inoff[i] : 0;
    for (int c0 = 0; c0 < m; c0 += 64) {
      const int t = c0 + lane;
      const int l = t < m ? dl[t] : kLevNone;
      const bool keep = l <= K;
      const unsigned long long mask = __ballot(keep);
      if (FILL && keep) {
        const int pos = run + __popcll(mask & ((1ull << lane) - 1ull));
        ncol[q0 + pos] = blo + t;
        nlev[q0 + pos] = (unsigned char)l;
        nval[q0 + pos] = dv[t];
        if (t == r) ndiag[i] = pos;
      }
      run += __popcll(mask);
    }
    if (lane == 0) {
      if (FILL) nrp[i] = q0;
      else cntL[r] = run;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (FILL) return;
  __syncthreads();
  if (wave == 0) {  // in-block exclusive scan of the new row lengths
    const int per = (m + 63) / 64, t0 = lane * per, t1 = min(t0 + per, m);
    int sum = 0, mx = 0;
    for (int t = t0; t < t1; ++t) { sum += cntL[t]; mx = max(mx, cntL[t]); }
    int inc = sum;
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(inc, o, 64);
      if (lane >= o) inc += v;
    }
    int run = inc - sum;
    for (int t = t0; t < t1; ++t) {
      inoff[blo + t] = run;
      nlen[blo + t] = cntL[t];
      run += cntL[t];
    }
    const int total = __shfl(inc, 63, 64);
    mx = wave_max_i32(mx);
    if (lane == 0) {
      blktot[b] = ((total + 63) / 64) * 64;
      atomicMax(maxlen, mx);
    }
  }
}

// exclusive scan of the block totals (a few thousand values: one workgroup)
__global__ __launch_bounds__(1024) void k_iluk_block_offsets(int nblocks, const int *__restrict__ blktot,
                                                             long long *__restrict__ boff) {
  __shared__ long long part[1024];
  const int t = threadIdx.x, per = (nblocks + 1023) / 1024, b0 = t * per, b1 = min(b0 + per, nblocks);
  long long sum = 0;
  for (int b = b0; b < b1; ++b) sum += blktot[b];
  part[t] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const long long v = t >= o ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  long long run = part[t] - sum;
  for (int b = b0; b < b1; ++b) { boff[b] = run; run += blktot[b]; }
  if (t == 1023) boff[nblocks] = part[1023];
}

// ---------------------------------------------------------------------------
// Static schedule of the triangular solves (built per factorisation, on the GPU)
//
// For each block and each direction (L: dependencies = strictly-lower entries,
// U: strictly-upper) rows are levelled (lev = 1 + max lev of dependencies; level
// 0 rows need no work) and the rows of a level are packed into "steps".  A step
// is a run of T chunks of 64 x (value, word); row i of the step owns g_i =
// ceil(d_i / T) CONTIGUOUS lanes (d_i = its dependency count) and streams its
// entries through them, so a step of rows with 5..50 dependencies fills the 64
// lanes instead of padding every row to the longest one.  Rows of a level are
// taken longest first; a step grows while its chunk count T stays, or -- when the
// next row would raise T -- while it stays at least half full afterwards (on the
// production 3-D matrices: 1.21 x the entry count instead of 1.73 x for fixed
// 8-lane groups, and one step per level; scripts/ilu_pack_model.py, ilu_pack_model2.py).
//   word (16 bit) = local column (10 bits) | TAIL << 10 | CONT << 11 | p << 12      per entry
//   info (8 bit)  = END | need << 1                                                per chunk
// END: last chunk of the step.  The step ends with a segmented scan in registers: p = distance of the lane from
// the start of its row's lanes inside its 16-lane DPP row (row_shr:1,2,4,8 add when p >= shift), CONT = the row's
// lanes began in the previous DPP row (row_bcast:15 carries the partial sum over, DPP row by DPP row; `need` says
// which of the three carries the step uses), TAIL = last lane of the row: it holds the row's sum and updates y.
// Which row a TAIL lane belongs to is not stored per entry: the rows appear in the stream in the order of the
// schedule's sorted positions, so the k-th TAIL lane of a direction belongs to row sperm[k] (a 2 x B table of 16-bit
// row numbers per block that the solve keeps in LDS, with a running count of the rows done).  10 B per stream entry
// instead of 12: the solve is bound by the stream bytes as much as by its dependency chain.
// Per chunk one fma against y in LDS; no flags, no waiting.
// Every step is at least half full or one chunk long, so a block never needs more
// than (entries/32 + rows per direction) chunks: capacity = 2 x its sliced-ELL
// region + 2 B, checked by the kernel all the same.
constexpr int kTail16 = 10, kCont16 = 11, kPos16 = 12;
constexpr int kPrefetch = 16;     // chunks kept in flight per wave (8, 12 and 24 measured slower or equal)
constexpr int kPadChunks = 32;    // per-block tail pad so the prefetch never leaves the buffer (>= deepest prefetch)
constexpr int kCapFactor = 2;
constexpr int kCapFactorSafe = 8;

__device__ __forceinline__ long long ilu_base_chunk(const long long *boff, int b, int capf, int slack) {
  return capf * (boff[b] >> 6) + (long long)(kPadChunks + slack) * b;
}

// ILU(0): a block's factor lives in the entry range of its rows' sliced-ELL region in A
__global__ void k_ilu_boff0(int nblocks, int B, int nslices, const long long *__restrict__ slice_off,
                            long long *__restrict__ boff) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b <= nblocks) boff[b] = slice_off[min((long long)b * (B / 64), (long long)nslices)];
}

// exact stream offsets from the counting pass: sboff[b] = 64 x (chunks of the blocks before b), sboff[nblocks] = 64 x total
__global__ __launch_bounds__(1024) void k_ilu_exact_offsets(int nblocks, const int *__restrict__ blkinfo, long long *__restrict__ sboff) {
  __shared__ long long part[1024];
  const int t = threadIdx.x;
  const int per = (nblocks + 1023) / 1024;
  const int b0 = min(t * per, nblocks), b1 = min(b0 + per, nblocks);
  long long sum = 0;
  for (int b = b0; b < b1; ++b) sum += (long long)blkinfo[4 * b] + blkinfo[4 * b + 1];
  part[t] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const long long v = t >= o ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  long long run = t > 0 ? part[t - 1] : 0;
  for (int b = b0; b < b1; ++b) { sboff[b] = run * 64; run += (long long)blkinfo[4 * b] + blkinfo[4 * b + 1]; }
  if (t == 1023) sboff[nblocks] = part[1023] * 64;
}

// (second launch bound: eight waves per SIMD.  The kernel needed 106 SGPRs -- seven waves per SIMD, THREE 512-thread blocks per
//  CU -- with 17 of them spilled to VGPR lanes it has four like its LDS allows: 1.60 -> 1.34 ms at 100^3)
// NARROW: no row of the matrix has more than 128 entries beside its diagonal (the host knows the longest row), so a row's
// dependencies in a direction fit two loads per lane: the level walk then has nothing conditional around its loads.
template <bool NARROW>
__global__ __launch_bounds__(1024, 8) void k_ilu_schedule(int n, int B, const long long *__restrict__ boff,
                                                       const long long *__restrict__ frp,
                                                       const int *__restrict__ fcol, const int *__restrict__ flen,
                                                       const int *__restrict__ fdiag, double *__restrict__ sv,
                                                       unsigned short *__restrict__ sc, unsigned char *__restrict__ si,
                                                       unsigned short *__restrict__ sperm, int *__restrict__ fdst,
                                                       int *__restrict__ blkinfo, int *__restrict__ llev,
                                                       int capf, int slack, int *__restrict__ err,
                                                       const double *__restrict__ sgs_fval,
                                                       const double *__restrict__ sgs_dinv, int b0, int count_only,
                                                       const int *__restrict__ bptr) {
  extern __shared__ int lds_i[];
  int *levL = lds_i;             // [B] level of every row in the L solve
  int *levU = levL + B;          // [B] ... in the U solve
  int *cnt = levU + B;           // [B+1] rows per level
  int *lvoff = cnt + B + 1;      // [B+1] sorted position of the first row of a level
  int *choff = lvoff + B + 1;    // [B+2] first chunk of every step
  int *skey = choff + B + 2;     // [B] sort key (level, -ndep) of the rank computation
  int *sd = skey + B;            // [B] dependency count, by sorted position
  int *slane = sd + B;           // [B] first lane of the row inside its step
  int *sT = slane + B;           // [B] chunk count of the row's step
  int *sinfo = sT + B;           // [B] bit 0: first row of its step, bits 1-3: carries needed, bits 8-14: lanes used
  int *sstep = sinfo + B;        // [B] step index
  int *slen = sstep + B;         // [B] row length / diagonal slot / first entry, for the level walk
  int *sdg = slen + B;           // [B]
  int *spos = sdg + B;           // [B] position of a row in the current direction's solve order
  long long *srp = reinterpret_cast<long long *>(spos + B + (B & 1));  // [B]
  __shared__ int s_nlev, s_nsched, s_nch, s_err;
  // rows without a diagonal or with duplicate / unsorted columns (flagged by k_ilu_extract) would send the level
  // walk through uninitialised levels: leave the block alone, the host reports the error.  The word is read ONCE per
  // workgroup (kernels of the next ingress batch may be raising bits of it meanwhile): the exit is uniform
  if (threadIdx.x == 0) s_err = *err;
  __syncthreads();
  if (s_err & (1 | 32)) return;
  const int b = blockIdx.x + b0;
  int blo, bhi;
  ilu_block_rows(bptr, b, B, n, blo, bhi);
  const int m = bhi - blo;
  const int t = threadIdx.x;
  const bool active = t < m;
  const int i = blo + t;
  // count_only: the pass that sizes the stream exactly -- levels, ranks and steps as below, the chunk counts go to
  // blkinfo, nothing is written to the stream (which does not exist yet), nothing can overflow
  const long long region = count_only ? 0 : boff[b + 1] - boff[b];
  const long long cap = count_only ? 0x7fffffffffffffffLL : capf * (region >> 6) + slack;
  const long long base = count_only ? 0 : ilu_base_chunk(boff, b, capf, slack);
  long long rp = 0;
  int len = 0, dg = 0;
  if (active) {
    rp = frp[i];
    len = flen[i];
    dg = fdiag[i];
    if (!count_only) fdst[rp + dg] = -1;
    srp[t] = rp; slen[t] = len; sdg[t] = dg;
  }
  __syncthreads();
  // ---- levels: lev = 1 + max lev of the dependencies (0: no dependencies).  Dependencies of the L solve have
  // smaller, those of the U solve larger indices, so ONE walk over the rows in index order settles a direction:
  // wave 0 walks upwards for L, wave 1 downwards for U, lanes over the row's dependencies, the columns of the
  // next kLevAhead rows already requested.  (A relaxation over all rows needed one sweep per level, ~175 of them:
  // half of this kernel's time, and its LDS column cache kept the kernel at one workgroup per CU.)
  {
    constexpr int kLevAhead = 16;  // a row is ~0.2 us of work and a column load ~2 us away: 4 rows ahead left the walk waiting (1.85 -> ms measured below)
    const int wave = t >> 6, lane = t & 63;
    const int nwalk = blockDim.x >= 128 ? 2 : 1;  // 64-row blocks have a single wave: it walks both directions
    for (int wdir = wave; wdir < 2; wdir += nwalk) {
      if (wave >= nwalk) break;
      int *lv = wdir == 0 ? levL : levU;
      const int step = wdir == 0 ? 1 : -1;
      const int r = wdir == 0 ? 0 : m - 1;
      if constexpr (NARROW) {
        // every request reads for every lane (a lane without a dependency, a request outside the block: the row's first
        // entry, ignored): the compiler's s_waitcnt pass can count loads that are always issued, and only those
        // (k_ilu_factor has the story); eight rows ahead, two columns per lane and row
        constexpr int kA = 8;
        int ca[kA], cb[kA];
        auto req = [&](int row, int &c0, int &c1) {
          const int rr = row < 0 ? 0 : (row >= m ? m - 1 : row);
          const int dgr = sdg[rr], lnr = slen[rr];
          const int d0r = wdir == 0 ? 0 : dgr + 1, nd = row == rr ? (wdir == 0 ? dgr : lnr - dgr - 1) : 0;
          const long long b0r = srp[rr];
          c0 = fcol[b0r + (lane < nd ? d0r + lane : 0)];
          c1 = fcol[b0r + (lane + 64 < nd ? d0r + lane + 64 : 0)];
        };
        if (m > 0) {
#pragma unroll
          for (int u = 0; u < kA; ++u) req(r + u * step, ca[u], cb[u]);
          for (int k0 = 0; k0 < m; k0 += kA) {
#pragma unroll
            for (int u = 0; u < kA; ++u) {
              const int row = r + (k0 + u) * step;
              if (k0 + u < m) {
                const int nd = wdir == 0 ? sdg[row] : slen[row] - sdg[row] - 1;
                int nl = lane < nd ? lv[ca[u] - blo] + 1 : 0;
                if (lane + 64 < nd) nl = max(nl, lv[cb[u] - blo] + 1);
                nl = wave_max_i32(nl);
                if (lane == 0) lv[row] = nl;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
              }
              req(row + kA * step, ca[u], cb[u]);
            }
          }
        }
        continue;
      }
      int cq[kLevAhead];
      auto request = [&](int row, int &c) {
        c = -1;
        if (row >= 0 && row < m) {
          const int d0r = wdir == 0 ? 0 : sdg[row] + 1, d1r = wdir == 0 ? sdg[row] : slen[row];
          if (lane < d1r - d0r) c = fcol[srp[row] + d0r + lane];
        }
      };
#pragma unroll
      for (int u = 0; u < kLevAhead; ++u) request(r + u * step, cq[u]);
      for (int k0 = 0; k0 < m; k0 += kLevAhead) {
#pragma unroll
        for (int u = 0; u < kLevAhead; ++u) {
          const int row = r + (k0 + u) * step;
          if (k0 + u < m) {
            int nl = cq[u] >= 0 ? lv[cq[u] - blo] + 1 : 0;
            const int d0r = wdir == 0 ? 0 : sdg[row] + 1, d1r = wdir == 0 ? sdg[row] : slen[row];
            for (int e = lane + 64; e < d1r - d0r; e += 64) nl = max(nl, lv[fcol[srp[row] + d0r + e] - blo] + 1);
            nl = wave_max_i32(nl);
            if (lane == 0) lv[row] = nl;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            request(row + kLevAhead * step, cq[u]);
          }
        }
      }
    }
  }
  __syncthreads();
  int used = 0;  // chunks used so far in this block (L then U)
  for (int dir = 0; dir < 2; ++dir) {
    const int d0 = dir == 0 ? 0 : dg + 1, d1 = dir == 0 ? dg : len;  // dependency slots
    const int ndep = active ? d1 - d0 : 0;
    const int *lev = dir == 0 ? levL : levU;
    if (dir == 0 && active) llev[i] = lev[t];  // the numeric factorisation walks the same levels
    // ---- histogram of levels
    if (t == 0) s_nlev = 0;
    for (int k = t; k <= B; k += blockDim.x) cnt[k] = 0;
    __syncthreads();
    const int mylev = active ? lev[t] : 0;
    if (active) { atomicAdd(&cnt[mylev], 1); atomicMax(&s_nlev, mylev + 1); }
    // deterministic rank inside the level: longest rows first (ties in row order)
    if (t < B) skey[t] = active ? ((mylev << 12) | (4095 - min(ndep, 4095))) : -1;
    __syncthreads();
    const bool sched = active && mylev > 0;
    int rk = 0;
    if (active) {  // rows without dependencies (level 0) are ranked too: they take the positions behind the stream's rows
      const int mykey = skey[t];
      for (int q = 0; q < m; ++q) {
        const int kq = skey[q];
        rk += (kq < mykey && (kq >> 12) == mylev) || (kq == mykey && q < t);
      }
    }
    if (t == 0) {
      int run = 0;
      for (int l = 1; l < s_nlev; ++l) { lvoff[l] = run; run += cnt[l]; }
      s_nsched = run;
      lvoff[0] = run;
    }
    __syncthreads();
    const int sp = active ? lvoff[mylev] + rk : 0;  // position of the row in this direction's solve order
    if (sched) sd[sp] = ndep;
    if (active) {
      spos[t] = sp;
      sperm[((size_t)b * 2 + dir) * B + t] = (unsigned short)sp;
    }
    __syncthreads();
    // ---- steps: the first row of every run of 64 ranks packs its run greedily
    if (sched && (rk & 63) == 0) {
      const int k = min(64, cnt[mylev] - rk);
      int a = 0;
      while (a < k) {
        int e = a + 1;
        int S = sd[sp + a];
        int T = (S + 63) >> 6;
        int lanes = (S + T - 1) / T;
        while (e < k) {
          const int de = sd[sp + e];
          int T2 = T, lanes2 = lanes + (de + T - 1) / T;
          if (lanes2 > 64) {
            do {
              ++T2;
              lanes2 = 0;
              for (int q = a; q <= e; ++q) lanes2 += (sd[sp + q] + T2 - 1) / T2;
            } while (lanes2 > 64);
            // a higher chunk count is accepted while the step stays >= 50 % full: a level then nearly always is ONE
            // step (the solve is bound by the number of steps, each ends in a scan, as much as by the stream bytes:
            // 300 -> 175 steps per block for 3 % more chunks; scripts/ilu_pack_model2.py)
            if (!(2 * (S + de) >= 64 * T2)) break;
          }
          T = T2; lanes = lanes2; S += de; ++e;
        }
        int run = 0, need = 0;
        for (int q = a; q < e; ++q) {
          const int g = (sd[sp + q] + T - 1) / T;
          slane[sp + q] = run;
          for (int r = (run >> 4) + 1; r <= ((run + g - 1) >> 4); ++r) need |= 1 << (r - 1);
          run += g;
        }
        for (int q = a; q < e; ++q) {
          sT[sp + q] = T;
          sinfo[sp + q] = (q == a ? 1 : 0) | (need << 1) | (run << 8);
        }
        a = e;
      }
    }
    __syncthreads();
    if (t == 0) {
      int step = -1, run = 0;
      for (int q = 0; q < s_nsched; ++q) {
        if (sinfo[q] & 1) { ++step; choff[step] = run; run += sT[q]; }
        sstep[q] = step;
      }
      s_nch = run;
      blkinfo[4 * b + dir] = run;
      blkinfo[4 * b + 2 + dir] = s_nsched;
      if ((long long)used + run > cap) atomicOr(err, 16);  // stream capacity exceeded
    }
    __syncthreads();
    if ((long long)used + s_nch > cap) return;  // uniform exit; host reports the error
    if (count_only) { used += s_nch; __syncthreads(); continue; }
    // ---- emission.  A wave writes the rows of its own 64 threads one after the other, lanes over the row's stream
    // entries: entry x of a row sits at factor slot rp + d0 + x (consecutive lanes read consecutive slots) and goes to
    // chunk x / g, lane ls + x % g (runs of g consecutive stream words).  (One thread per row made every access of a
    // wave hit 64 different cache lines: most of this kernel's time.)
    {
      int eT = 0, els = 0, einfo = 0, eg = 0;
      long long ech0 = 0;
      if (sched) {
        eT = sT[sp]; els = slane[sp]; einfo = sinfo[sp];
        eg = (ndep + eT - 1) / eT;
        ech0 = base + used + choff[sstep[sp]];
      }
      const int lane = t & 63;
      // src is wave-uniform: v_readlane instead of a ds_bpermute round trip per value
      auto from = [&](int v, int src) { return __builtin_amdgcn_readlane(v, src); };
      auto from64 = [&](long long v, int src) {
        return (long long)(((unsigned long long)(unsigned)from((int)(v >> 32), src) << 32) | (unsigned)from((int)v, src));
      };
      // the factor column of this lane's first entry of row `src` of the wave (row 64: none) -- requested one row ahead, for
      // every lane (a lane without an entry reads the wave's first slot), so that a row does not start with a load it waits for
      const long long rp_safe = from64(rp, 0);
      auto colreq = [&](int src) -> int {
        const int sl = src < 64 ? src : 63;
        const int nd = src < 64 ? from(ndep, sl) : 0;
        const long long slot = lane < nd ? from64(rp, sl) + from(d0, sl) + lane : rp_safe;
        return fcol[slot];
      };
      auto emit = [&](int src, int cj_first) {
        const int rT = from(eT, src);
        if (rT == 0) return;  // row without work in this direction (wave-uniform)
        const int rls = from(els, src), rinfo = from(einfo, src), rg = from(eg, src);
        const int rnd = from(ndep, src), rd0 = from(d0, src);
        const long long rrp = from64(rp, src), rch0 = from64(ech0, src);
        const int tot = rT * rg;
        for (int x = lane; x < tot; x += 64) {
          const int c = x / rg, q = x - c * rg, ln = rls + q;
          const long long idx = (rch0 + c) * 64 + ln;
          const int segstart = max(rls, ln & ~15);
          const unsigned flags = ((unsigned)(ln - segstart) << kPos16) | ((ln >> 4) > (rls >> 4) ? (1u << kCont16) : 0u) |
                                 (q == rg - 1 ? (1u << kTail16) : 0u);
          if (x < rnd) {
            const long long slot = rrp + rd0 + x;
            const int cj = x < 64 ? cj_first : fcol[slot];
            sc[idx] = (unsigned short)((unsigned)spos[cj - blo] | flags);
            fdst[slot] = (int)(idx - base * 64);  // block-relative: the stream as a whole may exceed 2^31 entries
            // Gauss-Seidel mode: the stream values are A's own entries (L part scaled by the column's pivot)
            if (sgs_dinv) sv[idx] = dir == 0 ? sgs_fval[slot] * sgs_dinv[cj] : sgs_fval[slot];
          } else {
            sc[idx] = (unsigned short)flags;
            sv[idx] = 0.0;
          }
        }
        if (rinfo & 1) {  // first row of its step: the chunk info bytes and the lanes the step does not use
          const unsigned need = (unsigned)((rinfo >> 1) & 7);
          for (int c = lane; c < rT; c += 64) si[rch0 + c] = (unsigned char)((c == rT - 1 ? 1u : 0u) | (need << 1));
          const int first = (rinfo >> 8) & 127, nun = 64 - first;
          for (int x = lane; x < rT * nun; x += 64) {
            const int c = x / nun, ln = first + x - c * nun;
            const long long idx = (rch0 + c) * 64 + ln;
            sc[idx] = 0;  // position 0, not a TAIL: adds 0 * y[0] to a sum nobody reads
            sv[idx] = 0.0;
          }
        }
      };
      // two rows per trip, their requests in two registers that take turns (a copy of a requested value would wait for it)
      int cja = colreq(0), cjb;
      for (int src = 0; src < 64; src += 2) {
        cjb = colreq(src + 1);
        emit(src, cja);
        cja = colreq(src + 2);
        emit(src + 1, cjb);
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
constexpr int kFacPrefetch = 4;  // r4, with the readlane requests: 4 -> 3.62 ms, 6 -> 6.27, 8 -> 5.48 (the queue registers cost a wave per SIMD)
constexpr int kFacPrefetchWide = 8;
constexpr int kIluWaves = 8;

// WIDE: two U-row entries per lane are prefetched (U-rows of up to 128 entries: ILU(k > 0), wide kernels); the narrow
// variant keeps one (the bench matrix has ~30 upper entries per row and the second queue only costs there).
// (second launch bound of the narrow variant: eight waves per SIMD = four 512-thread blocks per CU, which its LDS allows)
template <int WAVES, bool WIDE>
__global__ __launch_bounds__(WAVES * 64, WIDE ? 1 : 8) void k_ilu_factor(int n, int B, int W, const long long *__restrict__ frp,
                                                           const int *__restrict__ fcol, double *__restrict__ fval,
                                                           const int *__restrict__ flen,
                                                           const int *__restrict__ fdiag, const int *__restrict__ fdst,
                                                           const int *__restrict__ llev, double *__restrict__ sv,
                                                           double *__restrict__ dinv, const long long *__restrict__ boff,
                                                           int capf, int slack, int b0, const int *__restrict__ err,
                                                           const int *__restrict__ bptr) {
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
  __shared__ int s_nlev, s_err;
  // a ranged launch (host CSR ingress, ingress.hpp) queues this kernel before the host has seen the schedule's error
  // word: a failed extraction or an overflowed stream leaves the blocks alone and the host redoes the set-up.  One
  // read per workgroup, then a uniform exit (the next batch's kernels may be raising bits concurrently)
  if (threadIdx.x == 0) s_err = err != nullptr ? *err : 0;
  __syncthreads();
  if (s_err & (1 | 16 | 32)) return;
  const int bid = blockIdx.x + b0;
  int blo, bhi;
  ilu_block_rows(bptr, bid, B, n, blo, bhi);
  const int m = bhi - blo;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double *svb = sv + ilu_base_chunk(boff, bid, capf, slack) * 64;  // this block's part of the solve stream
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
  // first entry of the block's factor region (its rows lie back to back from there), as a wave-uniform value
  const long long rp_block = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(rpL[0] >> 32)) << 32) |
                                         (unsigned)__builtin_amdgcn_readfirstlane((int)rpL[0]));
  const int *__restrict__ fcol_b = fcol + rp_block;
  const double *__restrict__ fval_b = fval + rp_block;
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
      constexpr int kPF = WIDE ? kFacPrefetchWide : kFacPrefetch;
      // software-pipelined elimination: the U-row of step s+kPF is
      // requested while step s is applied (one entry per lane per 64 columns)
      // The prefetch only ISSUES the global loads (column and value of the U-row entry this lane will apply);
      // the column -> slot look-up in LDS happens when the step is applied.  (Looking the slot up inside the
      // request made every request wait for its own global load: no overlap at all.)
      // two entries of the U-row per lane are prefetched (rows of up to 128 upper entries: ILU(1) rows are that wide)
      int pcq[kPF], pcq2[kPF];      // column of the prefetched U-row entries (-1: none for this lane)
      double pvq[kPF], pvq2[kPF];   // their values
      double pdq[kPF];   // 1/d_k of that step
      int pnq[kPF];      // entries of that pivot row's upper part
      // What a step needs to know about its pivot row -- where its upper part starts, how long it is, the reciprocal pivot --
      // is looked up ONCE per row, one pivot per lane, and handed to the step by v_readlane (the step index is wave-uniform):
      // the four dependent LDS reads this replaces sat on every step's path between its fence and its loads.  (All pivot
      // rows belong to earlier levels: their tables are final.)  Pivots beyond the 64th of a row take the tables.
      long long pbase = 0;
      int pcnt = 0;
      double pdk = 0.0;
      if (lane < dg) {
        const int k = mc[lane];
        pbase = rpL[k] + dgL[k] + 1;
        pcnt = lenL[k] - dgL[k] - 1;
        pdk = diag[k];
      }
      // Rows whose pivots all have at most 64 upper entries and that have at most 64 pivots -- all rows of the narrow variant
      // on the bench matrices -- take a loop with NOTHING conditional around its loads: every request issues exactly two
      // loads for every lane (a lane without an entry, or a request behind the last pivot, reads the row's own first entry
      // again and the step ignores it).  Loads under a branch, or an inner loop with loads of its own, cannot be counted by
      // the compiler's s_waitcnt pass: it then waits for (nearly) everything in flight at some steps -- vmcnt(1) / vmcnt(0)
      // in the ISA where vmcnt(2 (kPF - 1)) belongs -- and the queue drains.  The other rows take the plain loop below.
      const bool narrow_row = !WIDE && dg <= 64 && __ballot(lane < dg && pcnt > 64) == 0;   // wave-uniform
      if (narrow_row) {
        // the loads of a request are addressed as (uniform base of the block's factor region) + (32-bit offset): the
        // region of a block holds some 10^4 entries, and a 64-bit position per lane cost six vector instructions per
        // request where two do (r5: the kernel issues vector instructions 68 % of the time, profiles/r05_pmc_ilu_setup.txt)
        const unsigned prel = (unsigned)(pbase - rp_block);   // this lane's pivot row, relative to the block's first entry
        const unsigned rprel = (unsigned)(rp - rp_block);
        auto request1 = [&](int s, int &pc, double &pvv, double &pdv, int &pn) {
          const int sl = s < dg ? s : 0;   // (dg >= 1 here: the loop below does not run otherwise)
          const unsigned p0 = (unsigned)__builtin_amdgcn_readlane((int)prel, sl);
          const int cnt = s < dg ? __builtin_amdgcn_readlane(pcnt, sl) : 0;
          pdv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(pdk), sl), __builtin_amdgcn_readlane(__double2loint(pdk), sl));
          pn = cnt;
          const unsigned p = lane < cnt ? p0 + (unsigned)lane : rprel;
          pc = fcol_b[p];  // raw column: no arithmetic on the loaded value here, or the wait moves up to the request
          pvv = fval_b[p];
        };
        if (dg > 0) {
#pragma unroll
          for (int u = 0; u < kPF; ++u) request1(u, pcq[u], pvq[u], pdq[u], pnq[u]);
          for (int s0 = 0; s0 < dg; s0 += kPF) {
#pragma unroll
            for (int u = 0; u < kPF; ++u) {
              const int s = s0 + u;
              if (s < dg) {
                const int ps = lane < pnq[u] ? mp[pcq[u] - blo] : 0;  // slot+1 in row i (0: not in the pattern)
                const double lik = mv[s] * pdq[u];   // l_ik = a_ik / d_k (reciprocal stored once per pivot)
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) mv[s] = lik;
                if (ps) mv[ps - 1] -= lik * pvq[u];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
              }
              request1(s + kPF, pcq[u], pvq[u], pdq[u], pnq[u]);   // unconditionally: the loads of every slot stay countable
            }
          }
        }
      } else {
      auto request = [&](int s, int &pc, double &pvv, int &pc2, double &pvv2, double &pdv, int &pn) {
        pc2 = -1;
        pvv2 = 0.0;
        pdv = 0.0;
        pn = 0;
        long long pbeg = rp;
        if (s < dg) {
          long long p0;
          int cnt;
          if (s < 64) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)pbase, s);
            const int hi = __builtin_amdgcn_readlane((int)(pbase >> 32), s);
            p0 = (long long)(((unsigned long long)(unsigned)hi << 32) | lo);
            cnt = __builtin_amdgcn_readlane(pcnt, s);
            pdv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(pdk), s), __builtin_amdgcn_readlane(__double2loint(pdk), s));
          } else {
            const int k = mc[s];
            p0 = rpL[k] + dgL[k] + 1;
            cnt = lenL[k] - dgL[k] - 1;
            pdv = diag[k];
          }
          pn = cnt;
          pbeg = p0;
        }
        // EVERY request issues its loads, for every lane (a lane without an entry, or a request behind the last pivot, reads
        // the row's own first entry again and the step ignores it): loads under a branch cannot be counted by the compiler's
        // s_waitcnt pass, which then waits for (nearly) everything in flight at each step -- vmcnt(1) / vmcnt(0) in the ISA
        // instead of vmcnt(2 (kPF - 1)) -- and the queue is a queue of one.
        const long long p = lane < pn ? pbeg + lane : rp;
        pc = fcol[p];  // raw column: no arithmetic on the loaded value here, or the wait moves up to the request
        pvv = fval[p];
        if (WIDE) {
          const long long p2 = lane + 64 < pn ? pbeg + 64 + lane : rp;
          pc2 = fcol[p2];
          pvv2 = fval[p2];
        }
      };
#pragma unroll
      for (int u = 0; u < kPF; ++u) request(u, pcq[u], pvq[u], pcq2[u], pvq2[u], pdq[u], pnq[u]);
      for (int s0 = 0; s0 < dg; s0 += kPF) {
#pragma unroll
        for (int u = 0; u < kPF; ++u) {
          const int s = s0 + u;
          if (s < dg) {
            const int ps = lane < pnq[u] ? mp[pcq[u] - blo] : 0;  // slot+1 in row i (0: not in the pattern)
            const int ps2 = WIDE && lane + 64 < pnq[u] ? mp[pcq2[u] - blo] : 0;
            const double lik = mv[s] * pdq[u];   // l_ik = a_ik / d_k (reciprocal stored once per pivot)
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) mv[s] = lik;
            if (ps) mv[ps - 1] -= lik * pvq[u];
            if (WIDE && ps2) mv[ps2 - 1] -= lik * pvq2[u];
            if (pnq[u] > (WIDE ? 128 : 64)) {  // U-rows wider than the prefetched part (wave-uniform)
              const int k = mc[s];
              for (int t = dgL[k] + 1 + (WIDE ? 128 : 64) + lane; t < lenL[k]; t += 64) {
                const long long p = rpL[k] + t;
                const int ps3 = mp[fcol[p] - blo];
                if (ps3) mv[ps3 - 1] -= lik * fval[p];
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            request(s + kPF, pcq[u], pvq[u], pcq2[u], pvq2[u], pdq[u], pnq[u]);
          }
        }
      }
      }
      for (int s = lane; s < len; s += 64) {
        fval[rp + s] = mv[s];
        const int d = fdst[rp + s];
        if (d >= 0) svb[d] = mv[s];  // triangular-solve stream
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
// Block-local symmetric Gauss-Seidel in the same stream form (no factorisation): with
// M_B = (D+L_B) D^-1 (D+U_B),  M_B^-1 r = (D+U_B)^-1 [ D (D+L_B)^-1 r ]  and  y = D (D+L_B)^-1 r  solves the
// unit-lower system  y_i = r_i - sum_j (a_ij / a_jj) y_j,  so the "L factor" is a_ij / a_jj, the "U factor" a_ij
// and the pivots a_ii: k_sgs_pivots + the sgs_* arguments of k_ilu_schedule.  Used as the AMG smoother (amg.hpp).
__global__ void k_sgs_pivots(int n, const long long *__restrict__ frp, const double *__restrict__ fval,
                             const int *__restrict__ fdiag, double *__restrict__ dinv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  // an empty row (a coarse unknown whose aggregate carries none of a masked null vector: zero column in P, zero row
  // in R A P) keeps its unknown at zero instead of dividing by its zero pivot
  if (i < n) { const double d = fval[frp[i] + fdiag[i]]; dinv[i] = d != 0.0 ? 1.0 / d : 0.0; }
}
// ---------------------------------------------------------------------------
// z = U^-1 D^-1 L^-1 r : one wave per block streams the block's chunk list.
// LDS per wave: y[B] and the block's reciprocal pivots.  kPrefetch chunks (values + words) are kept in flight.
// PART (Gauss-Seidel streams only, amg.hpp): 0 = both directions; 1 = the L part alone, z = (D+L_B)^-1 r (the stream
// leaves y = D (D+L_B)^-1 r, scaled on the way out); 2 = the U part alone on r, z = (D+U_B)^-1 r -- the forward and the
// backward sweep of ML's "Gauss-Seidel, efficient symmetric".  The instantiation the ILU preconditioner uses is PART 0.
template <int WAVES, int PF, int PART = 0>
__global__ __launch_bounds__(WAVES * 64) void k_ilu_solve_stream(int n, int B, int nblocks,
                                                                 const long long *__restrict__ boff,
                                                                 const double *__restrict__ sv,
                                                                 const unsigned short *__restrict__ sc,
                                                                 const unsigned char *__restrict__ si,
                                                                 const unsigned short *__restrict__ sperm,
                                                                 const int *__restrict__ blkinfo,
                                                                 const double *__restrict__ dinv,
                                                                 const double *__restrict__ r, double *__restrict__ z,
                                                                 int capf, int slack, const int *__restrict__ bptr,
                                                                 int accumulate = 0) {
  // accumulate: z += M^-1 r instead of z = M^-1 r (the smoother's x += M_B^-1 (b - A x): no vector kernel behind the sweep)
  extern __shared__ double lds_y[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + wave);  // wave-uniform: keep what follows scalar
  if (b >= nblocks) return;
  // per wave: y in the L solve's row order, y in the U solve's row order, the reciprocal pivots in U order (the U
  // steps must not wait on a global load each)
  double *yl = lds_y + (size_t)wave * 3 * B;
  double *yu = yl + B;
  double *dv = yu + B;
  int blo, bhi;
  ilu_block_rows(bptr, b, B, n, blo, bhi);
  const int m = bhi - blo;
  const unsigned short *__restrict__ posl = sperm + (size_t)b * 2 * B;  // row -> position in the L / U order
  const unsigned short *__restrict__ posu = posl + B;
  for (int t = lane; t < m; t += 64) {
    yl[posl[t]] = r[blo + t];
    dv[posu[t]] = dinv[blo + t];
  }
  const int nL0 = __builtin_amdgcn_readfirstlane(blkinfo[4 * b]), nU0 = __builtin_amdgcn_readfirstlane(blkinfo[4 * b + 1]);
  // PART 2 starts behind the L chunks and sees a stream without them; PART 1 ends before the U chunks
  const long long base = ilu_base_chunk(boff, b, capf, slack) + (PART == 2 ? nL0 : 0);
  const int nL = PART == 2 ? 0 : nL0, nU = PART == 1 ? 0 : nU0;
  const int nsU = __builtin_amdgcn_readfirstlane(blkinfo[4 * b + 3]);  // rows the U stream completes
  const double *__restrict__ pv = sv + base * 64 + lane;
  const unsigned short *__restrict__ pc = sc + base * 64 + lane;
  const unsigned char *__restrict__ pi = si + base;
  const unsigned long long below = (1ull << lane) - 1ull;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  double vq[PF];
  unsigned cq[PF], iq[PF];
  const int ntot = nL + nU;
#pragma unroll
  for (int u = 0; u < PF; ++u) {  // non-temporal: the stream is read once per apply
    vq[u] = __builtin_nontemporal_load(&pv[(long long)u * 64]);
    cq[u] = __builtin_nontemporal_load(&pc[(long long)u * 64]);
    iq[u] = pi[u];
  }
  // L -> U: the rows change places (U order), rows without upper dependencies are finished by their pivot
  auto to_upper = [&]() {
    for (int t = lane; t < m; t += 64) yu[posu[t]] = yl[posl[t]];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int t = nsU + lane; t < m; t += 64) yu[t] *= dv[t];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  double acc = 0.0;
  bool upper = false;
  int done = 0;      // rows of the current direction finished so far (wave-uniform)
  double *y = yl;
  for (int c0 = 0; c0 < ntot; c0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int c = c0 + u;
      const double v = vq[u];
      const unsigned cw = cq[u];
      const unsigned iw = __builtin_amdgcn_readfirstlane(iq[u]);
      vq[u] = __builtin_nontemporal_load(&pv[(long long)(c + PF) * 64]);  // stays inside the padded buffer
      cq[u] = __builtin_nontemporal_load(&pc[(long long)(c + PF) * 64]);
      iq[u] = pi[c + PF];
      if (c < ntot) {
        if (c == nL && !upper) {
          upper = true;
          done = 0;
          to_upper();
          y = yu;
        }
        acc = fma(v, y[cw & 0x3FFu], acc);
        if (iw & 1u) {
          // the k-th TAIL lane of a direction finishes the row at position k of that direction's order: no look-up.
          // The row's current value and pivot are fetched before the scan, not after it: they do not depend on it
          const bool tail = (cw >> kTail16) & 1u;
          const unsigned long long tails = __ballot(tail);
          const int pos = tail ? done + __popcll(tails & below) : 0;
          done += __popcll(tails);
          const double yold = y[pos];
          const double dvr = upper ? dv[pos] : 1.0;
          // segmented inclusive scan towards the last lane of every row (see k_ilu_schedule)
          const int p = (cw >> kPos16) & 15;
          double s = acc, q;
          q = dpp_move<0x111>(s); s += p >= 1 ? q : 0.0;  // row_shr:1
          q = dpp_move<0x112>(s); s += p >= 2 ? q : 0.0;  // row_shr:2
          q = dpp_move<0x114>(s); s += p >= 4 ? q : 0.0;  // row_shr:4
          q = dpp_move<0x118>(s); s += p >= 8 ? q : 0.0;  // row_shr:8
          const unsigned need = (iw >> 1) & 7u;  // wave-uniform
          if (need) {
            const bool cont = (cw >> kCont16) & 1u;
            if (need & 1u) { q = dpp_move<0x142, 0x2>(s); s += cont ? q : 0.0; }  // lane 15 -> DPP row 1
            if (need & 2u) { q = dpp_move<0x142, 0x4>(s); s += cont ? q : 0.0; }  // lane 31 -> DPP row 2
            if (need & 4u) { q = dpp_move<0x142, 0x8>(s); s += cont ? q : 0.0; }  // lane 47 -> DPP row 3
          }
          if (tail) y[pos] = (yold - s) * dvr;
          acc = 0.0;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
  }
  if (PART == 1) {  // forward sweep alone: y = D (D+L_B)^-1 r sits in the L order
    for (int t = lane; t < m; t += 64) {
      const double v = yl[posl[t]] * dinv[blo + t];
      z[blo + t] = accumulate ? z[blo + t] + v : v;
    }
    return;
  }
  if (!upper) to_upper();  // no U chunks at all (or an empty stream)
  if (accumulate) {
    for (int t = lane; t < m; t += 64) z[blo + t] += yu[posu[t]];
  } else {
    for (int t = lane; t < m; t += 64) z[blo + t] = yu[posu[t]];
  }
}

// ---------------------------------------------------------------------------
inline void ilu_destroy(isph_ilu *F) {
  if (!F) return;
  F->frp.release(); F->fcol.release(); F->flen.release(); F->fdiag.release(); F->err.release(); F->fval.release();
  F->bptr.release();
  F->boff.release(); F->sboff.release(); F->flev.release();
  F->sv.release(); F->sc.release(); F->si.release(); F->sperm.release(); F->fdst.release(); F->blkinfo.release(); F->dinv.release(); F->llev.release();
  delete F;
}

inline int ilu_check_err(isph_ctx *ctx, isph_ilu *F, const char *what, bool *overflow = nullptr) {
  int herr = 0;
  if (hipGetLastError() != hipSuccess ||
      hipMemcpyAsync(&herr, F->err.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess)
    return fail(what, __FILE__, __LINE__);
  if (herr & 1) return fail("matrix row without a diagonal entry: ILU(0) undefined", __FILE__, __LINE__);
  if (herr & 32) return fail("matrix row with duplicate or unsorted columns: ILU pattern undefined", __FILE__, __LINE__);
  if ((herr & 16) && overflow) { *overflow = true; return ISPH_SUCCESS; }
  if (herr & 16) return fail("ILU triangular-solve stream capacity exceeded", __FILE__, __LINE__);
  if (herr) return fail(what, __FILE__, __LINE__);
  return ISPH_SUCCESS;
}

// symbolic ILU(k): K sweeps of count + write over the extracted level-0 pattern (see k_iluk_merge); on return the
// factor arrays hold the ILU(K) pattern (fill entries zero), boff the 64-aligned block offsets
inline int ilu_symbolic(isph_ctx *ctx, isph_ilu *F, int K) {
  const int n = F->n, B = F->B, nb = F->nblocks;
  const size_t n1 = (size_t)n;
  DevBuf<int> nlen, inoff, blktot, meta;
  int rc = nlen.reserve(n1);
  if (rc == ISPH_SUCCESS) rc = inoff.reserve(n1);
  if (rc == ISPH_SUCCESS) rc = blktot.reserve((size_t)nb);
  if (rc == ISPH_SUCCESS) rc = meta.reserve(1);
  if (rc == ISPH_SUCCESS) rc = F->flev.reserve((size_t)(F->total > 0 ? F->total : 1));
  if (rc == ISPH_SUCCESS && hipMemsetAsync(F->flev.p, 0, (size_t)F->total, ctx->stream) != hipSuccess)
    rc = fail("memset failed", __FILE__, __LINE__);
  const size_t Bz = (size_t)B;
  const size_t lds = 8 * Bz + 8 * kSymWaves * Bz + 4 * 3 * Bz + 4 * kSymWaves * Bz + 2 * kSymWaves * Bz + kSymWaves * Bz;
  if (rc == ISPH_SUCCESS && lds > 160 * 1024) rc = fail("ILU(k) symbolic phase: block too large for LDS", __FILE__, __LINE__);
  if (rc == ISPH_SUCCESS &&
      (hipFuncSetAttribute(reinterpret_cast<const void *>(k_iluk_merge<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                           (int)lds) != hipSuccess ||
       hipFuncSetAttribute(reinterpret_cast<const void *>(k_iluk_merge<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                           (int)lds) != hipSuccess))
    rc = fail("LDS attribute failed", __FILE__, __LINE__);
  for (int sweep = 0; sweep < K && rc == ISPH_SUCCESS; ++sweep) {
    if (hipMemsetAsync(meta.p, 0, sizeof(int), ctx->stream) != hipSuccess) { rc = fail("memset failed", __FILE__, __LINE__); break; }
    hipLaunchKernelGGL((k_iluk_merge<false>), dim3(nb), dim3(kSymWaves * 64), lds, ctx->stream, n, B, K, F->frp.p, F->fcol.p,
                       F->flev.p, F->flen.p, F->fdiag.p, F->fval.p, nlen.p, inoff.p, blktot.p, meta.p,
                       (const long long *)nullptr, (long long *)nullptr, (int *)nullptr, (unsigned char *)nullptr,
                       (int *)nullptr, (double *)nullptr, F->blocks());
    hipLaunchKernelGGL(k_iluk_block_offsets, dim3(1), dim3(1024), 0, ctx->stream, nb, blktot.p, F->boff.p);
    long long total = 0;
    int wmax = 0, herr = 0;
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(&total, F->boff.p + nb, sizeof(long long), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(&wmax, meta.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(&herr, F->err.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
      rc = fail("ILU(k) symbolic count failed", __FILE__, __LINE__);
      break;
    }
    if (herr & 1) { rc = fail("matrix row without a diagonal entry: ILU undefined", __FILE__, __LINE__); break; }
    const size_t tot1 = (size_t)(total > 0 ? total : 1);
    DevBuf<long long> nrp;
    DevBuf<int> ncol, ndiag;
    DevBuf<unsigned char> nlev;
    DevBuf<double> nval;
    rc = nrp.reserve(n1);
    if (rc == ISPH_SUCCESS) rc = ncol.reserve(tot1);
    if (rc == ISPH_SUCCESS) rc = ndiag.reserve(n1);
    if (rc == ISPH_SUCCESS) rc = nlev.reserve(tot1);
    if (rc == ISPH_SUCCESS) rc = nval.reserve(tot1);
    if (rc == ISPH_SUCCESS) {
      hipLaunchKernelGGL((k_iluk_merge<true>), dim3(nb), dim3(kSymWaves * 64), lds, ctx->stream, n, B, K, F->frp.p, F->fcol.p,
                         F->flev.p, F->flen.p, F->fdiag.p, F->fval.p, nlen.p, inoff.p, blktot.p, meta.p,
                         (const long long *)F->boff.p, nrp.p, ncol.p, nlev.p, ndiag.p, nval.p, F->blocks());
      if (hipGetLastError() != hipSuccess) rc = fail("ILU(k) symbolic fill failed", __FILE__, __LINE__);
    }
    if (rc != ISPH_SUCCESS) { nrp.release(); ncol.release(); ndiag.release(); nlev.release(); nval.release(); break; }
    // the new pattern replaces the old one (stream order keeps the old buffers alive until the kernel is done:
    // the pool hands a released block out again only to later work on the same stream)
    F->frp.release(); F->fcol.release(); F->fdiag.release(); F->flev.release(); F->fval.release(); F->flen.release();
    F->frp = nrp; F->fcol = ncol; F->fdiag = ndiag; F->flev = nlev; F->fval = nval;
    F->flen = nlen;
    nlen = DevBuf<int>();
    rc = nlen.reserve(n1);
    F->total = total;
    F->wmax = wmax;
  }
  nlen.release(); inoff.release(); blktot.release(); meta.release();
  F->flev.release();  // only the symbolic sweeps need the levels
  return rc;
}

// ---- set-up in pieces: allocations that only need the matrix' SHAPE (ilu_begin), then extract / schedule / factor over
// ranges of blocks.  ilu_create runs them over all blocks with the host check between schedule and factorisation; the
// host CSR ingress (ingress.hpp) queues them range by range behind the rows that have arrived over PCIe.
// above this size (10 bytes per stream entry at the capacity rule's factor 2) the stream is sized exactly by a counting
// pass of the schedule: the 4 M x 749 operator of BASELINE configs[4] would reserve 110 GB for a 55 GB Gauss-Seidel stream
inline long long &ilu_exact_stream_bytes() {
  static long long v = 4LL << 30;  // isph_set_exact_stream_threshold
  return v;
}
inline bool ilu_wants_exact_stream(const isph_ilu *F) {
  const long long rule_chunks = kCapFactor * (F->total >> 6) + (long long)(kPadChunks + 2 * F->B) * (F->nblocks + 1) + kPadChunks;
  return rule_chunks * 64 * 10 > ilu_exact_stream_bytes();
}
inline int ilu_size_stream(isph_ilu *F) {
  const long long entries = F->exact ? F->stream_entries : F->total;
  F->stream_chunks = F->capf * (entries >> 6) + (long long)(kPadChunks + F->slack) * (F->nblocks + 1) + kPadChunks;
  int r = F->sv.reserve((size_t)F->stream_chunks * 64);
  if (r == ISPH_SUCCESS) r = F->sc.reserve((size_t)F->stream_chunks * 64);
  if (r == ISPH_SUCCESS) r = F->si.reserve((size_t)F->stream_chunks + 64);
  return r;
}

// needs S.nrow, S.stored, S.wmax, S.nslices and S.slice_off on the device (stream-ordered); not S.col / S.val
// nblocks_tab > 0: the caller's subdomains, host_bptr[0 .. nblocks_tab] (ascending from 0 to nrow, every block at most
// block_size rows -- block_size is then the capacity); ILU(0) / Gauss-Seidel only
inline int ilu_begin(isph_ctx *ctx, const Sell &S, int block_size, bool sgs, int fill, isph_ilu **out, bool defer_factor_arrays = false,
                     int nblocks_tab = 0, const int *host_bptr = nullptr) {
  ISPH_REQUIRE(block_size >= 64 && block_size <= 1024 && block_size % 64 == 0,
               "block-Jacobi ILU block size must be a multiple of 64 in [64,1024]");
  ISPH_REQUIRE(fill >= 0 && fill <= 8 && !(sgs && fill), "level of fill must be in [0,8]");
  ISPH_REQUIRE(nblocks_tab == 0 || host_bptr, "a table of subdomains without its offsets");
  isph_ilu *F = new isph_ilu();
  F->n = S.nrow; F->B = block_size; F->wmax = S.wmax; F->fill = fill;
  F->nblocks = nblocks_tab > 0 ? nblocks_tab : (S.nrow + block_size - 1) / block_size;
  if (nblocks_tab > 0) {
    if (F->bptr.reserve((size_t)nblocks_tab + 1) != ISPH_SUCCESS ||
        hipMemcpyAsync(F->bptr.p, host_bptr, sizeof(int) * ((size_t)nblocks_tab + 1), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {  // the caller's array need not outlive the call
      ilu_destroy(F);
      return fail("copy of the subdomain table failed", __FILE__, __LINE__);
    }
  }
  // caller-defined subdomains: a block's factor region is its in-block entry count rounded up to 64 (k_ilu_count_inblock),
  // so the regions of all blocks together may exceed A's stored entries by up to 63 per block (a table of 1-row blocks on
  // a banded matrix does); the arrays and the stream are sized for that bound
  F->total = S.stored + (nblocks_tab > 0 ? 64LL * nblocks_tab : 0);
  const size_t stored = (size_t)(F->total > 0 ? F->total : 1), n1 = (size_t)(S.nrow > 0 ? S.nrow : 1);
  F->capf = kCapFactor; F->slack = 2 * block_size;
  int rc = defer_factor_arrays ? ISPH_SUCCESS : F->fcol.reserve(stored);
  if (rc == ISPH_SUCCESS && !defer_factor_arrays) rc = F->fval.reserve(stored);
  if (rc == ISPH_SUCCESS) rc = F->frp.reserve(n1);
  if (rc == ISPH_SUCCESS) rc = F->flen.reserve(n1);
  if (rc == ISPH_SUCCESS) rc = F->fdiag.reserve(n1);
  if (rc == ISPH_SUCCESS) rc = F->dinv.reserve(n1);
  if (rc == ISPH_SUCCESS) rc = F->llev.reserve(n1);
  if (rc == ISPH_SUCCESS) rc = F->err.reserve(1);
  if (rc == ISPH_SUCCESS) rc = F->boff.reserve((size_t)F->nblocks + 1);
  if (rc == ISPH_SUCCESS) rc = F->blkinfo.reserve((size_t)4 * (F->nblocks > 0 ? F->nblocks : 1));
  if (rc == ISPH_SUCCESS) rc = F->sperm.reserve((size_t)2 * block_size * (F->nblocks > 0 ? F->nblocks : 1));
  if (rc == ISPH_SUCCESS && hipMemsetAsync(F->err.p, 0, sizeof(int), ctx->stream) != hipSuccess)
    rc = fail("memset failed", __FILE__, __LINE__);
  if (rc == ISPH_SUCCESS && S.nrow > 0) {
    const size_t lds_e = (size_t)(block_size / 64) * 64 * ((kExtChunk + 1) * 8 + (kExtChunk + 1) * 4);
    const size_t lds_s = sizeof(int) * (14 * (size_t)block_size + 10) + sizeof(long long) * (size_t)block_size;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_ilu_extract), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_e) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_ilu_schedule<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_s) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_ilu_schedule<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_s) != hipSuccess)
      rc = fail("LDS attribute failed", __FILE__, __LINE__);
  }
  if (rc != ISPH_SUCCESS) { ilu_destroy(F); return rc; }
  *out = F;
  return ISPH_SUCCESS;
}

// ILU(0) only: block regions = the sliced-ELL regions of the blocks' rows, the stream sized from them
inline int ilu_begin_fill0(isph_ctx *ctx, isph_ilu *F, const Sell &S, bool size_stream = true) {
  if (S.nrow == 0) return ISPH_SUCCESS;
  hipLaunchKernelGGL(k_ilu_boff0, dim3((F->nblocks + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, F->nblocks,
                     F->B, S.nslices, (const long long *)S.slice_off.p, F->boff.p);
  ISPH_CHECK(F->fdst.reserve((size_t)(F->total > 0 ? F->total : 1)));
  return size_stream ? ilu_size_stream(F) : ISPH_SUCCESS;
}

// st: the stream of a ranged launch (ingress.hpp runs consecutive ranges on alternating streams); default ctx->stream
inline void ilu_launch_extract(isph_ctx *ctx, isph_ilu *F, const Sell &S, int b0, int nb, hipStream_t st = nullptr) {
  const size_t lds_e = (size_t)(F->B / 64) * 64 * ((kExtChunk + 1) * 8 + (kExtChunk + 1) * 4);
  ProfScope prof(ctx, PROF_ILU_EXTRACT, st);
  hipLaunchKernelGGL(k_ilu_extract, dim3(nb), dim3(F->B), lds_e, st ? st : ctx->stream, S.nrow, F->B, (const int *)S.rowlen.p,
                     (const long long *)S.slice_off.p, (const int *)S.col.p, (const double *)S.val.p, F->frp.p, F->fcol.p,
                     F->fval.p, F->flen.p, F->fdiag.p, F->err.p, b0, F->compact ? (const long long *)F->boff.p : (const long long *)nullptr,
                     F->blocks());
}

inline void ilu_launch_schedule(isph_ctx *ctx, isph_ilu *F, const Sell &S, int b0, int nb, bool sgs, hipStream_t st = nullptr,
                                bool count_only = false) {
  const size_t Bz = (size_t)F->B;
  const size_t lds_s = sizeof(int) * (14 * Bz + 10) + sizeof(long long) * Bz;
  ProfScope prof(ctx, PROF_ILU_SCHEDULE, st);
  if (F->wmax <= 129)   // a row of at most 129 entries has at most 128 dependencies
    hipLaunchKernelGGL(k_ilu_schedule<true>, dim3(nb), dim3(F->B), lds_s, st ? st : ctx->stream, S.nrow, F->B, F->stream_off(), F->frp.p, F->fcol.p,
                       F->flen.p, F->fdiag.p, F->sv.p, F->sc.p, F->si.p, F->sperm.p, F->fdst.p, F->blkinfo.p, F->llev.p, F->capf,
                       F->slack, F->err.p, sgs ? (const double *)F->fval.p : (const double *)nullptr,
                       sgs ? (const double *)F->dinv.p : (const double *)nullptr, b0, count_only ? 1 : 0, F->blocks());
  else
    hipLaunchKernelGGL(k_ilu_schedule<false>, dim3(nb), dim3(F->B), lds_s, st ? st : ctx->stream, S.nrow, F->B, F->stream_off(), F->frp.p, F->fcol.p,
                       F->flen.p, F->fdiag.p, F->sv.p, F->sc.p, F->si.p, F->sperm.p, F->fdst.p, F->blkinfo.p, F->llev.p, F->capf,
                       F->slack, F->err.p, sgs ? (const double *)F->fval.p : (const double *)nullptr,
                       sgs ? (const double *)F->dinv.p : (const double *)nullptr, b0, count_only ? 1 : 0, F->blocks());
}

// err_dev != nullptr: the kernel itself skips its blocks when the set-up so far has raised an error (ranged launches)
inline int ilu_launch_factor(isph_ctx *ctx, isph_ilu *F, const Sell &S, int b0, int nb, const int *err_dev, hipStream_t st_in = nullptr) {
  hipStream_t st = st_in ? st_in : ctx->stream;
  const size_t Bz = (size_t)F->B;
  const int W = ((F->wmax + 63) / 64) * 64;
  const size_t lds_f = 8 * Bz + (size_t)kIluWaves * W * 12 + 8 * Bz + 4 * (5 * Bz + 4) + 2 * Bz + 2 * (size_t)kIluWaves * Bz + 16;
  ISPH_REQUIRE(lds_f <= 160 * 1024, "ILU factor kernel needs too much LDS for this row width");
  const bool wide = F->wmax > 128;  // rows this long have U parts beyond one wave
  const void *fk = wide ? reinterpret_cast<const void *>(k_ilu_factor<kIluWaves, true>)
                        : reinterpret_cast<const void *>(k_ilu_factor<kIluWaves, false>);
  ISPH_CHECK_HIP(hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
  ProfScope prof(ctx, PROF_ILU_FACTOR, st_in);
  if (wide)
    hipLaunchKernelGGL((k_ilu_factor<kIluWaves, true>), dim3(nb), dim3(kIluWaves * 64), lds_f, st, S.nrow, F->B, W,
                       F->frp.p, F->fcol.p, F->fval.p, F->flen.p, F->fdiag.p, F->fdst.p, F->llev.p, F->sv.p, F->dinv.p, F->stream_off(),
                       F->capf, F->slack, b0, err_dev, F->blocks());
  else
    hipLaunchKernelGGL((k_ilu_factor<kIluWaves, false>), dim3(nb), dim3(kIluWaves * 64), lds_f, st, S.nrow, F->B, W,
                       F->frp.p, F->fcol.p, F->fval.p, F->flen.p, F->fdiag.p, F->fdst.p, F->llev.p, F->sv.p, F->dinv.p, F->stream_off(),
                       F->capf, F->slack, b0, err_dev, F->blocks());
  ISPH_CHECK_HIP(hipGetLastError());
  return ISPH_SUCCESS;
}

// schedule (all blocks) with the host check and the one retry at the proven stream capacity, then the factorisation
// (or, for the Gauss-Seidel stream, nothing more).  Expects the extraction (and the symbolic phase) queued.
inline int ilu_schedule_and_factor(isph_ctx *ctx, isph_ilu *F, const Sell &S, bool sgs) {
  int rc = ISPH_SUCCESS;
  if (sgs)  // pivots first: the schedule writes the Gauss-Seidel stream values itself
    hipLaunchKernelGGL(k_sgs_pivots, dim3((S.nrow + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, S.nrow, F->frp.p,
                       F->fval.p, F->fdiag.p, F->dinv.p);
  if (F->exact) {  // counting pass: chunk counts per block -> exact offsets -> a stream of exactly that size
    rc = F->sboff.reserve((size_t)F->nblocks + 1);
    if (rc == ISPH_SUCCESS) {
      ilu_launch_schedule(ctx, F, S, 0, F->nblocks, sgs, nullptr, /*count_only=*/true);
      hipLaunchKernelGGL(k_ilu_exact_offsets, dim3(1), dim3(1024), 0, ctx->stream, F->nblocks, (const int *)F->blkinfo.p, F->sboff.p);
      long long total = 0;
      if (hipMemcpyAsync(&total, F->sboff.p + F->nblocks, sizeof(long long), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        rc = fail("copy failed", __FILE__, __LINE__);
      if (rc == ISPH_SUCCESS) rc = ilu_check_err(ctx, F, "ILU extract/schedule (counting pass) failed");
      F->stream_entries = total;
      F->capf = 1; F->slack = 0;
      if (rc == ISPH_SUCCESS) rc = ilu_size_stream(F);
    }
  }
  for (int attempt = 0; attempt < 2 && rc == ISPH_SUCCESS; ++attempt) {
    ilu_launch_schedule(ctx, F, S, 0, F->nblocks, sgs);
    // the factor kernel must not run on a partial schedule: check now (one sync per build)
    bool overflow = false;
    rc = ilu_check_err(ctx, F, "ILU extract/schedule kernel failed", attempt == 0 ? &overflow : nullptr);
    if (!overflow) break;
    if (F->exact) { rc = fail("ILU stream: the fill pass needed more chunks than the counting pass found", __FILE__, __LINE__); break; }
    F->capf = kCapFactorSafe; F->slack = 2 * F->B;  // proven bound, see kCapFactorSafe
    rc = ilu_size_stream(F);
    if (rc == ISPH_SUCCESS && hipMemsetAsync(F->err.p, 0, sizeof(int), ctx->stream) != hipSuccess)
      rc = fail("memset failed", __FILE__, __LINE__);
  }
  if (rc == ISPH_SUCCESS && sgs) {
    // nothing left to do: k_ilu_schedule filled the stream; the row-major copy is not read again (no export of a
    // smoother) and is 16 B per stored entry -- 48 GB on the 4 M x 749 operator of BASELINE configs[4]
    F->fcol.release(); F->fval.release(); F->fdst.release();
  } else if (rc == ISPH_SUCCESS) {
    rc = ilu_launch_factor(ctx, F, S, 0, F->nblocks, nullptr);
  }
  return rc;
}

inline int ilu_create(isph_ctx *ctx, const isph_mat *A, int block_size, isph_ilu **out, bool sgs = false, int fill = 0,
                      int nblocks_tab = 0, const int *host_bptr = nullptr) {
  const Sell &S = A->S;
  isph_ilu *F = nullptr;
  const bool var = nblocks_tab > 0;
  if (var) {
    ISPH_REQUIRE(host_bptr && host_bptr[0] == 0 && host_bptr[nblocks_tab] == S.nrow, "subdomain table must run from 0 to the number of rows");
    for (int b = 0; b < nblocks_tab; ++b)
      ISPH_REQUIRE(host_bptr[b + 1] > host_bptr[b] && host_bptr[b + 1] - host_bptr[b] <= block_size,
                   "every subdomain needs between 1 and `capacity` rows");
  }
  // large operators (see ilu_wants_exact_stream): the factor regions are sized from a count of the in-block entries
  // instead of A's sliced-ELL regions, and the stream from a counting pass of the schedule
  bool big = false;
  {
    isph_ilu probe;
    probe.total = S.stored; probe.B = block_size; probe.nblocks = var ? nblocks_tab : (S.nrow + block_size - 1) / block_size;
    big = fill == 0 && S.nrow > 0 && ilu_wants_exact_stream(&probe);
  }
  ISPH_CHECK(ilu_begin(ctx, S, block_size, sgs, fill, &F, /*defer_factor_arrays=*/big, nblocks_tab, host_bptr));
  int rc = ISPH_SUCCESS;
  if (var && !big && S.nrow > 0) {
    // caller-defined subdomains: their rows are not whole 64-row slices of A, so a block's factor region cannot be A's
    // own sliced-ELL region -- the regions are the in-block entry counts (64-aligned), formed on the device without a
    // host round trip; the arrays keep the upper bound (all of A's entries) ilu_begin reserved
    DevTmp<int> blktot;
    rc = blktot.reserve((size_t)F->nblocks);
    if (rc == ISPH_SUCCESS) {
      hipLaunchKernelGGL(k_ilu_count_inblock, dim3(F->nblocks), dim3(block_size), 0, ctx->stream, S.nrow, block_size,
                         (const int *)S.rowlen.p, (const long long *)S.slice_off.p, (const int *)S.col.p, blktot.p, F->blocks());
      hipLaunchKernelGGL(k_iluk_block_offsets, dim3(1), dim3(1024), 0, ctx->stream, F->nblocks, (const int *)blktot.p, F->boff.p);
      F->compact = true;
    }
    if (rc == ISPH_SUCCESS) {
      ilu_launch_extract(ctx, F, S, 0, F->nblocks);
      if (fill > 0) {  // "fact: level-of-fill" > 0 on the table's blocks: the symbolic sweeps re-lay the factor out
        rc = ilu_symbolic(ctx, F, fill);
        F->exact = ilu_wants_exact_stream(F);
      }
    }
    if (rc == ISPH_SUCCESS) rc = F->fdst.reserve((size_t)(F->total > 0 ? F->total : 1));
    if (rc == ISPH_SUCCESS && !F->exact) rc = ilu_size_stream(F);
    if (rc == ISPH_SUCCESS) rc = ilu_schedule_and_factor(ctx, F, S, sgs);
    blktot.release();
  } else if (big) {
    DevTmp<int> blktot;
    rc = blktot.reserve((size_t)F->nblocks);
    long long total = 0;
    if (rc == ISPH_SUCCESS) {
      hipLaunchKernelGGL(k_ilu_count_inblock, dim3(F->nblocks), dim3(block_size), 0, ctx->stream, S.nrow, block_size,
                         (const int *)S.rowlen.p, (const long long *)S.slice_off.p, (const int *)S.col.p, blktot.p, F->blocks());
      hipLaunchKernelGGL(k_iluk_block_offsets, dim3(1), dim3(1024), 0, ctx->stream, F->nblocks, (const int *)blktot.p, F->boff.p);
      if (hipMemcpyAsync(&total, F->boff.p + F->nblocks, sizeof(long long), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipStreamSynchronize(ctx->stream) != hipSuccess)
        rc = fail("count of the in-block entries failed", __FILE__, __LINE__);
    }
    F->total = total;
    F->compact = true;
    F->exact = true;
    const size_t tot1 = (size_t)(total > 0 ? total : 1);
    if (rc == ISPH_SUCCESS) rc = F->fcol.reserve(tot1);
    if (rc == ISPH_SUCCESS) rc = F->fval.reserve(tot1);
    if (rc == ISPH_SUCCESS) rc = F->fdst.reserve(tot1);
    if (rc == ISPH_SUCCESS) {
      ilu_launch_extract(ctx, F, S, 0, F->nblocks);
      rc = ilu_schedule_and_factor(ctx, F, S, sgs);
    }
  } else if (S.nrow > 0) {
    ilu_launch_extract(ctx, F, S, 0, F->nblocks);
    if (fill > 0) {
      rc = ilu_symbolic(ctx, F, fill);
      if (rc == ISPH_SUCCESS) rc = F->fdst.reserve((size_t)(F->total > 0 ? F->total : 1));
      F->exact = ilu_wants_exact_stream(F);
      if (rc == ISPH_SUCCESS && !F->exact) rc = ilu_size_stream(F);
    } else {
      F->exact = ilu_wants_exact_stream(F);
      rc = ilu_begin_fill0(ctx, F, S, /*size_stream=*/!F->exact);
    }
    if (rc == ISPH_SUCCESS) rc = ilu_schedule_and_factor(ctx, F, S, sgs);
  }
  if (rc != ISPH_SUCCESS) { ilu_destroy(F); return rc; }
  *out = F;
  return ISPH_SUCCESS;
}

// The same sweep for NV right-hand sides at once (the lockstep Helmholtz solve, solver.hpp gmres_lockstep): the stream
// -- values, column words, step flags -- is read once and applied to NV vectors; every vector goes through exactly the
// operations of k_ilu_solve_stream, in the same order, so the results are bit-identical to NV separate applications.
// LDS per wave: y[NV][B] (the L -> U reordering happens in place, through registers) + the pivots in U order.
struct IluVecs {
  const double *r[4];
  double *z[4];
};
template <int WAVES, int PF, int NV>
__global__ __launch_bounds__(WAVES * 64) void k_ilu_solve_stream_multi(int n, int B, int nblocks,
                                                                       const long long *__restrict__ boff,
                                                                       const double *__restrict__ sv,
                                                                       const unsigned short *__restrict__ sc,
                                                                       const unsigned char *__restrict__ si,
                                                                       const unsigned short *__restrict__ sperm,
                                                                       const int *__restrict__ blkinfo,
                                                                       const double *__restrict__ dinv, IluVecs X,
                                                                       int capf, int slack, const int *__restrict__ bptr) {
  extern __shared__ double lds_y[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + wave);
  if (b >= nblocks) return;
  double *yb = lds_y + (size_t)wave * (NV + 1) * B;  // y[k] = yb + k * B
  double *dv = yb + (size_t)NV * B;
  int blo, bhi;
  ilu_block_rows(bptr, b, B, n, blo, bhi);
  const int m = bhi - blo;
  const unsigned short *__restrict__ posl = sperm + (size_t)b * 2 * B;
  const unsigned short *__restrict__ posu = posl + B;
  for (int t = lane; t < m; t += 64) {
    const int pl = posl[t];
#pragma unroll
    for (int k = 0; k < NV; ++k) yb[k * B + pl] = X.r[k][blo + t];
    dv[posu[t]] = dinv[blo + t];
  }
  const long long base = ilu_base_chunk(boff, b, capf, slack);
  const int nL = __builtin_amdgcn_readfirstlane(blkinfo[4 * b]), nU = __builtin_amdgcn_readfirstlane(blkinfo[4 * b + 1]);
  const int nsU = __builtin_amdgcn_readfirstlane(blkinfo[4 * b + 3]);
  const double *__restrict__ pv = sv + base * 64 + lane;
  const unsigned short *__restrict__ pc = sc + base * 64 + lane;
  const unsigned char *__restrict__ pi = si + base;
  const unsigned long long below = (1ull << lane) - 1ull;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  double vq[PF];
  unsigned cq[PF], iq[PF];
  const int ntot = nL + nU;
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    vq[u] = __builtin_nontemporal_load(&pv[(long long)u * 64]);
    cq[u] = __builtin_nontemporal_load(&pc[(long long)u * 64]);
    iq[u] = pi[u];
  }
  // L -> U: every vector changes from the L order to the U order in place (values through registers: B <= 1024 is at
  // most 16 per lane), rows without upper dependencies are finished by their pivot
  auto to_upper = [&]() {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      double hold[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int t = lane + 64 * q;
        hold[q] = t < m ? yb[k * B + posl[t]] : 0.0;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int t = lane + 64 * q;
        if (t < m) yb[k * B + posu[t]] = hold[q];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int t = nsU + lane; t < m; t += 64) {
      const double d = dv[t];
#pragma unroll
      for (int k = 0; k < NV; ++k) yb[k * B + t] *= d;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  double acc[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) acc[k] = 0.0;
  bool upper = false;
  int done = 0;
  for (int c0 = 0; c0 < ntot; c0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int c = c0 + u;
      const double v = vq[u];
      const unsigned cw = cq[u];
      const unsigned iw = __builtin_amdgcn_readfirstlane(iq[u]);
      vq[u] = __builtin_nontemporal_load(&pv[(long long)(c + PF) * 64]);
      cq[u] = __builtin_nontemporal_load(&pc[(long long)(c + PF) * 64]);
      iq[u] = pi[c + PF];
      if (c < ntot) {
        if (c == nL && !upper) {
          upper = true;
          done = 0;
          to_upper();
        }
        const int col = (int)(cw & 0x3FFu);
#pragma unroll
        for (int k = 0; k < NV; ++k) acc[k] = fma(v, yb[k * B + col], acc[k]);
        if (iw & 1u) {
          const bool tail = (cw >> kTail16) & 1u;
          const unsigned long long tails = __ballot(tail);
          const int pos = tail ? done + __popcll(tails & below) : 0;
          done += __popcll(tails);
          const double dvr = upper ? dv[pos] : 1.0;
          const int p = (cw >> kPos16) & 15;
          const unsigned need = (iw >> 1) & 7u;
          const bool cont = (cw >> kCont16) & 1u;
#pragma unroll
          for (int k = 0; k < NV; ++k) {
            const double yold = yb[k * B + pos];
            double s = acc[k], q;
            q = dpp_move<0x111>(s); s += p >= 1 ? q : 0.0;
            q = dpp_move<0x112>(s); s += p >= 2 ? q : 0.0;
            q = dpp_move<0x114>(s); s += p >= 4 ? q : 0.0;
            q = dpp_move<0x118>(s); s += p >= 8 ? q : 0.0;
            if (need) {
              if (need & 1u) { q = dpp_move<0x142, 0x2>(s); s += cont ? q : 0.0; }
              if (need & 2u) { q = dpp_move<0x142, 0x4>(s); s += cont ? q : 0.0; }
              if (need & 4u) { q = dpp_move<0x142, 0x8>(s); s += cont ? q : 0.0; }
            }
            if (tail) yb[k * B + pos] = (yold - s) * dvr;
            acc[k] = 0.0;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
  }
  if (!upper) to_upper();
  for (int t = lane; t < m; t += 64) {
    const int pu = posu[t];
#pragma unroll
    for (int k = 0; k < NV; ++k) X.z[k][blo + t] = yb[k * B + pu];
  }
}

// z_k = U^-1 D^-1 L^-1 r_k for K = 2..4 vectors in one sweep of the factor stream (falls back to K sweeps otherwise)
inline int ilu_apply(isph_ctx *ctx, const isph_ilu *F, const double *r, double *z, int part = 0, bool accumulate = false);
inline int ilu_apply_multi(isph_ctx *ctx, const isph_ilu *F, int K, const double *const *rs, double *const *zs) {
  ISPH_REQUIRE(F != nullptr, "ILU factor is NULL");
  if (F->n == 0) return ISPH_SUCCESS;
  constexpr int WV = 4;
  const size_t lds = sizeof(double) * (size_t)(K + 1) * (size_t)F->B * WV;
  // K sweeps when one sweep for all vectors does not fit the LDS this device gives a workgroup (asked, not assumed)
  static const size_t lds_max = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0)
      return (size_t)64 * 1024;
    return (size_t)v;
  }();
  if (K < 2 || K > 4 || lds > lds_max) {
    for (int k = 0; k < K; ++k) ISPH_CHECK(ilu_apply(ctx, F, rs[k], zs[k]));
    return ISPH_SUCCESS;
  }
  IluVecs X;
  for (int k = 0; k < 4; ++k) { X.r[k] = rs[k < K ? k : 0]; X.z[k] = zs[k < K ? k : 0]; }
#define ISPH_ILU_LAUNCH_MULTI(NV)                                                                                          \
  do {                                                                                                                      \
    ISPH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ilu_solve_stream_multi<WV, kPrefetch, NV>),         \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                              \
    hipLaunchKernelGGL((k_ilu_solve_stream_multi<WV, kPrefetch, NV>), dim3((F->nblocks + WV - 1) / WV), dim3(WV * 64), lds, \
                       ctx->stream, F->n, F->B, F->nblocks, F->stream_off(), F->sv.p, F->sc.p, F->si.p, F->sperm.p,               \
                       F->blkinfo.p, F->dinv.p, X, F->capf, F->slack, F->blocks());                                         \
  } while (0)
  if (K == 2) ISPH_ILU_LAUNCH_MULTI(2);
  else if (K == 3) ISPH_ILU_LAUNCH_MULTI(3);
  else ISPH_ILU_LAUNCH_MULTI(4);
#undef ISPH_ILU_LAUNCH_MULTI
  ISPH_CHECK_HIP(hipGetLastError());
  return ISPH_SUCCESS;
}

inline int ilu_apply(isph_ctx *ctx, const isph_ilu *F, const double *r, double *z, int part, bool accumulate) {
  ISPH_REQUIRE(F != nullptr, "ILU factor is NULL");
  ISPH_REQUIRE(part >= 0 && part <= 2, "ilu_apply: part must be 0 (both sweeps), 1 (lower) or 2 (upper)");
  if (F->n == 0) return ISPH_SUCCESS;
  constexpr int WV = 4;
  const size_t lds = sizeof(double) * 3 * (size_t)F->B * WV;
  constexpr int pf = kPrefetch;
#define ISPH_ILU_LAUNCH(PF, PART)                                                                                       \
  do {                                                                                                                   \
    if (lds > 48 * 1024)                                                                                                 \
      ISPH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ilu_solve_stream<WV, PF, PART>),               \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                         \
    hipLaunchKernelGGL((k_ilu_solve_stream<WV, PF, PART>), dim3((F->nblocks + WV - 1) / WV), dim3(WV * 64), lds, ctx->stream, \
                       F->n, F->B, F->nblocks, F->stream_off(), F->sv.p, F->sc.p, F->si.p, F->sperm.p, F->blkinfo.p,          \
                       F->dinv.p, r, z, F->capf, F->slack, F->blocks(), accumulate ? 1 : 0);                            \
  } while (0)
  if (part == 1) ISPH_ILU_LAUNCH(16, 1);
  else if (part == 2) ISPH_ILU_LAUNCH(16, 2);
  else if (pf == 12) ISPH_ILU_LAUNCH(12, 0);
  else if (pf == 16) ISPH_ILU_LAUNCH(16, 0);
  else if (pf == 24) ISPH_ILU_LAUNCH(24, 0);
  else ISPH_ILU_LAUNCH(8, 0);
#undef ISPH_ILU_LAUNCH
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
  std::vector<int> len((size_t)F->n);
  std::vector<long long> rp((size_t)F->n);
  std::vector<int> col((size_t)F->total);
  std::vector<double> v((size_t)F->total);
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
