// amg.hpp -- smoothed-aggregation AMG on the GPU, standing in for PrecondWrapper_ML
// (ref: IMPLICIT-SPH/precond_ml.h:40-171).
//
// Configuration restated from the wrapper: "max levels" 5, "aggregation: type" Uncoupled (aggregates never
// leave the rank), threshold 0, prolongator damping 4/3, symmetric Gauss-Seidel, 1 sweep pre and post, V cycle,
// direct coarse solve -- or, when a null vector is injected (setNullVector, :97-127), a one-dimensional
// pre-computed null space and the smoother as the coarse solver.  ML itself is an un-vendored Trilinos package:
// the method is the published smoothed aggregation (Vanek/Mandel/Brezina) with two GPU-minded choices that the
// oracle (oracle/isph_amg_oracle.c) shares entry by entry:
//   * aggregate roots = distance-2 maximal independent set, synchronous rounds with hashed priorities
//     (Bell/Dalton/Olson 2012) instead of ML's sequential greedy sweep: same "root + all neighbours" aggregates;
//   * Gauss-Seidel local to blocks of `block` rows (ML: local to the processor), one residual per sweep:
//     x += M_B^-1 (b - A x), M_B = blockdiag[(D+L_B) D^-1 (D+U_B)], run through the ILU chunk stream (k_ilu_schedule in Gauss-Seidel mode);
//   * damping from rho = ||D^-1 A||_inf ("eigen-analysis: type" Anorm).
// Set-up kernels work on device CSR copies, one wave per row (rows hold ~100 entries on the fine level); the
// cycle itself uses the SELL SpMV and the chunk-stream triangular solves of the rest of the library.
#pragma once
#include <rocprim/rocprim.hpp>

#include "ilu.hpp"
#include "solver.hpp"

int mat_from_device_csr(isph_ctx *ctx, int nrow, int ncol, const int *drp, const int *dci, const double *dv, long long nnz,
                        isph_mat **Aout, bool rows_sorted = false);  // isph_capi.hip
int mat_from_device_csr(isph_ctx *ctx, int nrow, int ncol, const long long *drp, const int *dci, const double *dv,
                        long long nnz, isph_mat **Aout, bool rows_sorted = false);
void isph_mat_destroy(isph_mat *A);
extern "C" int isph_mat_set_halo(isph_ctx *ctx, isph_mat *A, int npeers, const int *peer_rank, const int *send_ptr,
                                 const int *send_idx, const int *recv_ptr);

namespace isph {

// row offsets of every device CSR of the set-up are 64-bit: the fine-level copy of the operator can exceed 2^31
// entries (BASELINE configs[4]: 4 M rows x 749); the derived operators are small but share the kernels
typedef long long rp_t;

struct DCsr {
  int n = 0, m = 0;
  long long nnz = 0;
  DevBuf<rp_t> rp;
  DevBuf<int> ci;
  DevBuf<double> v;
  void release() { rp.release(); ci.release(); v.release(); n = m = 0; nnz = 0; }
};

struct AmgLevel {
  DCsr A;                    // device CSR of the level operator (fine level: a copy of the SELL matrix)
  const isph_mat *Am = nullptr;
  isph_mat *Aown = nullptr;  // coarse levels own their SELL matrix
  DCsr P;                    // prolongator (kept in CSR for export)
  isph_mat *Pm = nullptr;
  isph_mat *APm = nullptr;   // A P of the set-up, kept when the level has no halo: residual update after the coarse correction
  DCsr R;                    // restriction P^T in CSR: its rows are hundreds of entries long, one wave per row
  isph_ilu *sgs = nullptr;   // block-local symmetric Gauss-Seidel in stream form
  DevBuf<double> wsgs;       // small coarse levels: the same smoother as dense 64 x 64 inverses (k_sgs_dense_build)
  DevBuf<double> wfwd, wbwd; // ... of the forward / backward sweep alone (isph_amg::gs_eff)
  DevBuf<int> agg;
  DevBuf<double> nv, x, b, r, z;
};

}  // namespace isph

struct isph_amg {
  int nlev = 0, block = 512, sweeps = 1, singular = 0;
  int gs_eff = 0;         // isph_amg_params::smoother == 1: forward sweeps before, backward sweeps after the coarse correction
  int coarse_smooth = 0;  // coarsest level solved by the smoother: singular system (precond_ml.h:97-127) or a level too
                          // large for the dense inverse (no coarsening possible: isolated / Dirichlet rows dominate)
  std::vector<isph::AmgLevel *> L;
  isph::DevBuf<double> cinv;  // dense inverse of the coarsest operator (non-singular case)
  int nc = 0;
  // more than one rank (coarse levels across ranks, see amg_extend_pack / amg_extend_finish): the coarsest operator of ALL ranks is
  // inverted on every rank (nc = its global size, rows nc_off .. nc_off + nc_loc are this rank's)
  int dist = 0, nc_off = 0, nc_loc = 0;
  isph::DevBuf<double> bglob;
};

namespace isph {

constexpr int kAmgCoarseBlock = 64;  // rows the Gauss-Seidel sweeps are local to on levels >= 1
constexpr int kAmgWaves = 4;  // rows per 256-thread workgroup in the wave-per-row kernels
enum { AMG_COVERED = 0, AMG_UNDECIDED = 1, AMG_ROOT = 3 };

__device__ __forceinline__ unsigned amg_hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned long long amg_key(int state, int i) {
  return ((unsigned long long)state << 62) | ((unsigned long long)amg_hash32((unsigned)i) << 30) | (unsigned long long)i;
}
// wave-wide max / min, result in every lane: DPP moves (a lane without a source keeps its own value) + v_readlane, not the
// six (twelve for 64 bits) dependent ds_bpermute round trips of a __shfl_xor butterfly -- these sit once per matrix row in the
// MIS sweeps
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ unsigned long long dpp_keep_u64(unsigned long long v) {
  const int lo = dpp_keep_i32<CTRL, ROW_MASK>((int)(unsigned)v), hi = dpp_keep_i32<CTRL, ROW_MASK>((int)(unsigned)(v >> 32));
  return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  unsigned long long t;
  t = dpp_keep_u64<0x111>(v); v = t > v ? t : v;
  t = dpp_keep_u64<0x112>(v); v = t > v ? t : v;
  t = dpp_keep_u64<0x114>(v); v = t > v ? t : v;
  t = dpp_keep_u64<0x118>(v); v = t > v ? t : v;
  t = dpp_keep_u64<0x142, 0xa>(v); v = t > v ? t : v;
  t = dpp_keep_u64<0x143, 0xc>(v); v = t > v ? t : v;
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ int wave_min_i32(int v) { return -wave_max_i32(-v); }   // callers pass row lengths / indices >= 0
__device__ __forceinline__ double wave_max_f64(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
// ML's strength test: a_ij strong iff a_ij^2 > theta^2 |a_ii a_jj| ("aggregation: threshold", default 0)
// (threshold 0: every non-zero off-diagonal entry is strong -- the diagonal gathers are skipped)
#define AMG_STRONG(val, i, j) ((j) < n && (j) != (i) && (th2 == 0.0 ? (val) != 0.0 : (val) * (val) > th2 * fabs(dg[i] * dg[j])))

// SELL -> CSR, one wave per slice: a lane reads its own row (coalesced across the wave), kCsrChunk entries per
// round are staged in LDS and kCsrChunk consecutive lanes write one row's piece of the row-major destination
constexpr int kCsrChunk = 16;  // (8 until r5: 64-byte row pieces; 16 gives 128-byte value pieces per row: 1.17 -> see docs/kernels_detail.md)
// PREP (aggregation threshold 0): the sweep also leaves what k_amg_prepare would compute from the copy -- diagonal,
// strong columns, first MIS keys, rho -- the lane that owns a row sees its entries go by (the row sum of rho is added in
// entry order here, across lanes first there: the last bit of rho may differ between the two)
template <bool PREP>
__global__ __launch_bounds__(256) void k_sell_to_csr_i32(int nrow, const int *__restrict__ rowlen,
                                                         const long long *__restrict__ slice_off,
                                                         const int *__restrict__ scol, const double *__restrict__ sval,
                                                         const rp_t *__restrict__ rowptr, int *__restrict__ colidx,
                                                         double *__restrict__ cval, double *__restrict__ dg = nullptr,
                                                         int *__restrict__ sc = nullptr, unsigned long long *__restrict__ key = nullptr,
                                                         volatile unsigned long long *rho_bits = nullptr) {
  // (+1: a lane writes its row's piece, rows kCsrChunk words apart would all fall on two LDS banks)
  __shared__ int stc[4][64][kCsrChunk + 1];
  __shared__ double stv[4][64][kCsrChunk + 1];
  __shared__ int slen[4][64];
  __shared__ rp_t sbeg[4][64];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slice = blockIdx.x * 4 + wv;
  const int row = slice * 64 + lane;
  if (slice * 64 >= nrow) return;
  const long long off = slice_off[slice];
  const int len = row < nrow ? rowlen[row] : 0;
  slen[wv][lane] = len;
  sbeg[wv][lane] = row < nrow ? rowptr[row] : 0;
  const int rounds = (wave_max_i32(len) + kCsrChunk - 1) / kCsrChunk;
  double pd = 0.0, ps = 0.0;
  bool pfound = false, pstrong = false;
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int u = 0; u < kCsrChunk; ++u) {
      const int k = r * kCsrChunk + u;
      if (k < len) {
        const long long p = sell_pos(off, lane, k);
        const int c = scol[p];
        const double a = sval[p];
        stc[wv][lane][u] = c;
        stv[wv][lane][u] = a;
        if (PREP) {
          if (c == row && !pfound) { pd = a; pfound = true; }
          pstrong |= c < nrow && c != row && a != 0.0;
          if (c < nrow) ps += fabs(a);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < kCsrChunk; ++it) {
      const int e = lane + 64 * it, rr = e / kCsrChunk, u = e % kCsrChunk, k = r * kCsrChunk + u;
      if (k < slen[wv][rr]) {
        const int c = stc[wv][rr][u];
        const double a = stv[wv][rr][u];
        colidx[sbeg[wv][rr] + k] = c;
        cval[sbeg[wv][rr] + k] = a;
        if (PREP) sc[sbeg[wv][rr] + k] = (c < nrow && c != slice * 64 + rr && a != 0.0) ? c : -1;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (PREP) {
    const double d = pfound ? pd : 1.0;
    if (row < nrow) { dg[row] = d; key[row] = amg_key(pstrong ? AMG_UNDECIDED : AMG_COVERED, row); }
    const double q = (row < nrow && d != 0.0) ? ps / fabs(d) : 0.0;
    unsigned long long bits = wave_max_u64((unsigned long long)__double_as_longlong(q));
    if (lane == 0 && bits > *rho_bits) atomicMax(const_cast<unsigned long long *>(rho_bits), bits);
  }
}

__global__ __launch_bounds__(256) void k_amg_diag(int n, const rp_t *__restrict__ rp, const int *__restrict__ ci,
                                                  const double *__restrict__ v, double *__restrict__ dg) {
  const int i = blockIdx.x * kAmgWaves + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  double d = 0.0;
  bool found = false;
  for (rp_t p = rp[i] + lane; p < rp[i + 1]; p += 64)
    if (ci[p] == i && !found) { d = v[p]; found = true; }
  const unsigned long long any = __ballot(found);
  if (any) d = __shfl(d, __ffsll((long long)any) - 1, 64);
  if (lane == 0) dg[i] = any ? d : 1.0;
}

// sc[p] = column of entry p when it is a strong off-diagonal coupling, -1 otherwise: the ~30 graph sweeps of the
// aggregation then read 4 B per entry instead of column + value + two diagonal gathers
__global__ void k_strong_cols(int n, const rp_t *__restrict__ rp, const int *__restrict__ ci, const double *__restrict__ v,
                              const double *__restrict__ dg, double th2, int *__restrict__ sc) {
  const int i = blockIdx.x * kAmgWaves + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  for (rp_t p = rp[i] + lane; p < rp[i + 1]; p += 64) {
    const int j = ci[p];
    sc[p] = AMG_STRONG(v[p], i, j) ? j : -1;
  }
}

// threshold 0 (the wrapper's default): ONE sweep over A instead of four -- the diagonal (k_amg_diag), the strong
// columns (k_strong_cols), the first MIS keys (k_mis_init) and rho (k_amg_rho), each with the arithmetic of the kernel
// it stands for
__global__ __launch_bounds__(256) void k_amg_prepare(int n, const rp_t *__restrict__ rp, const int *__restrict__ ci,
                                                     const double *__restrict__ v, double *__restrict__ dg, int *__restrict__ sc,
                                                     unsigned long long *__restrict__ key, volatile unsigned long long *rho_bits) {
  const int i = blockIdx.x * kAmgWaves + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  double d = 0.0, s = 0.0;
  bool found = false, strong = false;
  for (rp_t p = rp[i] + lane; p < rp[i + 1]; p += 64) {
    const int j = ci[p];
    const double a = v[p];
    if (j == i && !found) { d = a; found = true; }
    const bool st = j < n && j != i && a != 0.0;
    sc[p] = st ? j : -1;
    strong |= st;
    if (j < n) s += fabs(a);
  }
  const unsigned long long any = __ballot(found);
  d = any ? __shfl(d, __ffsll((long long)any) - 1, 64) : 1.0;
  const bool any_strong = __ballot(strong) != 0;
  if (lane == 0) { dg[i] = d; key[i] = amg_key(any_strong ? AMG_UNDECIDED : AMG_COVERED, i); }
  if (d == 0.0) return;  // wave-uniform
  s = wave_sum(s) / fabs(d);
  const unsigned long long bits = (unsigned long long)__double_as_longlong(s);
  if (lane == 0 && bits > *rho_bits) atomicMax(const_cast<unsigned long long *>(rho_bits), bits);
}

__global__ __launch_bounds__(256) void k_mis_init(int n, const rp_t *__restrict__ rp, const int *__restrict__ sc,
                                                  unsigned long long *__restrict__ key) {
  const int i = blockIdx.x * kAmgWaves + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  bool strong = false;
  for (rp_t p = rp[i] + lane; p < rp[i + 1]; p += 64) strong |= sc[p] >= 0;
  const bool any_strong = __ballot(strong) != 0;  // every lane takes part in the vote
  if (lane == 0) key[i] = amg_key(any_strong ? AMG_UNDECIDED : AMG_COVERED, i);
}

// out[i] = max(in[i], max over strong neighbours in[j])
__global__ __launch_bounds__(256) void k_mis_max(int n, const rp_t *__restrict__ rp, const int *__restrict__ sc,
                                                 const unsigned long long *__restrict__ in,
                                                 unsigned long long *__restrict__ out,
                                                 const unsigned long long *__restrict__ only_undecided) {
  const int i = blockIdx.x * kAmgWaves + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  // second sweep: only undecided rows read their result
  if (only_undecided && (only_undecided[i] >> 62) != AMG_UNDECIDED) return;
  unsigned long long m = in[i];
  for (rp_t p = rp[i] + lane; p < rp[i + 1]; p += 64) {
    const int j = sc[p];
    if (j >= 0) { const unsigned long long kj = in[j]; m = kj > m ? kj : m; }
  }
  m = wave_max_u64(m);
  if (lane == 0) out[i] = m;
}

__global__ void k_mis_decide(int n, unsigned long long *__restrict__ key, const unsigned long long *__restrict__ t2,
                             int *__restrict__ undecided) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool und = false;
  if (i < n && (key[i] >> 62) == AMG_UNDECIDED) {
    if (t2[i] == key[i]) key[i] = amg_key(AMG_ROOT, i);
    else if ((t2[i] >> 62) == AMG_ROOT) key[i] = amg_key(AMG_COVERED, i);
    else und = true;
  }
  const unsigned long long votes = __ballot(und);  // every lane takes part in the vote
  if (votes && (threadIdx.x & 63) == 0) atomicAdd(undecided, __popcll(votes));
}

// ---- MIS rounds on work lists.  After the first rounds only a few rows are still undecided; a round then needs the
// neighbourhood maximum t1 only on the rows next to an undecided row and t2 only on the undecided rows themselves.
// k_mis_decide_list appends the rows it leaves undecided to the next round's list, k_mis_mark collects their
// neighbourhood (each row once: stamp), and the list kernels read their length from device memory (no extra sync).
// The lists are filled in atomic order; every value computed from them is order independent.
constexpr int kMisGrid = 2048;

__global__ __launch_bounds__(256) void k_mis_max_list(const int *__restrict__ cntp, const int *__restrict__ list,
                                                      const rp_t *__restrict__ rp, const int *__restrict__ sc,
                                                      const unsigned long long *__restrict__ in,
                                                      unsigned long long *__restrict__ out) {
  const int nw = gridDim.x * kAmgWaves, lane = threadIdx.x & 63, cnt = *cntp;
  for (int w = blockIdx.x * kAmgWaves + (threadIdx.x >> 6); w < cnt; w += nw) {
    const int i = list[w];
    unsigned long long m = in[i];
    for (rp_t p = rp[i] + lane; p < rp[i + 1]; p += 64) {
      const int j = sc[p];
      if (j >= 0) { const unsigned long long kj = in[j]; m = kj > m ? kj : m; }
    }
    m = wave_max_u64(m);
    if (lane == 0) out[i] = m;
  }
}

constexpr int kMarkBuf = 1024;  // fresh rows a wave collects before it claims list space (one atomic per flush)

__global__ __launch_bounds__(256) void k_mis_mark(const int *__restrict__ cntp, const int *__restrict__ list,
                                                  const rp_t *__restrict__ rp, const int *__restrict__ sc,
                                                  int *__restrict__ stamp, int round, int *__restrict__ list1,
                                                  int *__restrict__ cnt1) {
  __shared__ int sbuf[kAmgWaves][kMarkBuf];
  int *buf = sbuf[threadIdx.x >> 6];
  const int nw = gridDim.x * kAmgWaves, lane = threadIdx.x & 63, cnt = *cntp;
  int fill = 0;  // wave-uniform
  auto flush = [&]() {
    if (fill == 0) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(cnt1, fill);
    base = __shfl(base, 0, 64);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < fill; k += 64) list1[base + k] = buf[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    fill = 0;
  };
  for (int w = blockIdx.x * kAmgWaves + (threadIdx.x >> 6); w < cnt; w += nw) {
    const int i = list[w];
    const rp_t lo = rp[i], hi = rp[i + 1];
    for (rp_t p0 = lo - 1; p0 < hi; p0 += 64) {  // slot lo-1 stands for the row itself
      const rp_t p = p0 + lane;
      int j = -1;
      if (p < hi) j = p < lo ? i : sc[p];
      bool fresh = false;
      if (j >= 0 && stamp[j] != round) fresh = atomicExch(&stamp[j], round) != round;
      const unsigned long long votes = __ballot(fresh);
      if (votes) {
        if (fill + 64 > kMarkBuf) flush();
        if (fresh) buf[fill + __popcll(votes & ((1ull << lane) - 1ull))] = j;
        fill += __popcll(votes);
      }
    }
  }
  flush();
}

// list == nullptr: the identity list 0..n_all-1 (first round)
__global__ __launch_bounds__(256) void k_mis_decide_list(const int *__restrict__ cntp, const int *__restrict__ list,
                                                         int n_all, unsigned long long *__restrict__ key,
                                                         const unsigned long long *__restrict__ t2,
                                                         int *__restrict__ next, int *__restrict__ nextcnt) {
  const int total = gridDim.x * blockDim.x, lane = threadIdx.x & 63, cnt = list ? *cntp : n_all;
  for (int base0 = blockIdx.x * blockDim.x + threadIdx.x - lane; base0 < cnt; base0 += total) {
    const int idx = base0 + lane;
    bool und = false;
    int i = -1;
    if (idx < cnt) {
      i = list ? list[idx] : idx;
      if ((key[i] >> 62) == AMG_UNDECIDED) {
        const unsigned long long m2 = t2[i];
        if (m2 == key[i]) key[i] = amg_key(AMG_ROOT, i);
        else if ((m2 >> 62) == AMG_ROOT) key[i] = amg_key(AMG_COVERED, i);
        // the largest key within distance 2 belongs to an undecided row m: when m is the largest of ITS distance-2
        // neighbourhood too (t2[m] == its key == m2) it becomes a root in this very round, and i is covered by it.
        // (Without this every second round only spreads the "covered" state; the roots are the same either way:
        // the lexicographically first distance-2 independent set of the priorities.)
        else if (t2[(int)(m2 & 0x3FFFFFFFull)] == m2) key[i] = amg_key(AMG_COVERED, i);
        else und = true;
      }
    }
    const unsigned long long votes = __ballot(und);  // every lane takes part in the vote
    if (votes) {
      int base = 0;
      if (lane == 0) base = atomicAdd(nextcnt, __popcll(votes));
      base = __shfl(base, 0, 64);
      if (und) next[base + __popcll(votes & ((1ull << lane) - 1ull))] = i;
    }
  }
}

__global__ void k_flag_roots(int n, const unsigned long long *__restrict__ key, int *__restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = (key[i] >> 62) == AMG_ROOT;
}

// pass 1: roots take their scan id, a non-root takes the id of the first strong neighbour that is a root
__global__ __launch_bounds__(256) void k_agg_pass1(int n, const rp_t *__restrict__ rp, const int *__restrict__ sc,
                                                   const unsigned long long *__restrict__ key,
                                                   const int *__restrict__ rootid, int *__restrict__ a1) {
  const int i = blockIdx.x * kAmgWaves + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  if ((key[i] >> 62) == AMG_ROOT) { if (lane == 0) a1[i] = rootid[i]; return; }
  int best = 0x7fffffff;  // position within the row
  const rp_t lo = rp[i];
  for (rp_t p = lo + lane; p < rp[i + 1]; p += 64) {
    const int j = sc[p];
    if (j >= 0 && (key[j] >> 62) == AMG_ROOT && (int)(p - lo) < best) best = (int)(p - lo);
  }
  best = wave_min_i32(best);
  if (lane == 0) a1[i] = best == 0x7fffffff ? -1 : rootid[sc[lo + best]];
}

// pass 2: the rest joins the pass-1 neighbour it is most strongly coupled to (ties: first in the row)
__global__ __launch_bounds__(256) void k_agg_pass2(int n, const rp_t *__restrict__ rp, const int *__restrict__ sc,
                                                   const double *__restrict__ v, const int *__restrict__ a1, int *__restrict__ agg,
                                                   int *__restrict__ leftover) {
  const int i = blockIdx.x * kAmgWaves + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  if (a1[i] >= 0) { if (lane == 0) { agg[i] = a1[i]; leftover[i] = 0; } return; }
  double bw = -1.0;
  int bp = 0x7fffffff;
  bool strong = false;
  const rp_t lo = rp[i];
  for (rp_t p = lo + lane; p < rp[i + 1]; p += 64) {
    const int j = sc[p];
    if (j < 0) continue;
    strong = true;
    if (a1[j] < 0) continue;
    const double w = fabs(v[p]);
    if (w > bw) { bw = w; bp = (int)(p - lo); }
  }
  const double wmax = wave_max_f64(bw);
  const int pbest = wave_min_i32((bw == wmax && wmax >= 0.0) ? bp : 0x7fffffff);
  const bool any_strong = __ballot(strong) != 0;
  if (lane == 0) {
    const int a = pbest == 0x7fffffff ? -1 : a1[sc[lo + pbest]];
    agg[i] = a;
    leftover[i] = (a < 0 && any_strong) ? 1 : 0;  // unsymmetric patterns only: becomes a singleton (pass 3)
  }
}

__global__ void k_agg_pass3(int n, const int *__restrict__ leftover, const int *__restrict__ leftid, int base,
                            int *__restrict__ agg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && leftover[i]) agg[i] = base + leftid[i];
}

__global__ void k_member_keys(int n, const int *__restrict__ agg, unsigned long long *__restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keys[i] = agg[i] >= 0 ? (((unsigned long long)agg[i] << 32) | (unsigned)i) : ~0ull;
}

// start[a] = first sorted position of aggregate a, start[nagg] = number of aggregated nodes
__global__ void k_segment_starts(int n, int nagg, const unsigned long long *__restrict__ keys, int *__restrict__ start) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const unsigned long long k = keys[p];
  if (k == ~0ull) {
    if (p == 0 || keys[p - 1] != ~0ull) start[nagg] = p;
    return;
  }
  const int a = (int)(k >> 32);
  if (p == 0 || (int)(keys[p - 1] >> 32) != a) start[a] = p;
  if (p == n - 1) start[nagg] = n;
}

// nvc[a] = |nv restricted to aggregate a|: one wave per aggregate (a hundred members on the fine level), lane l sums
// members l, l + 64, .. in index order and the lanes are added in a fixed order (reproducible)
__global__ __launch_bounds__(256) void k_agg_norm(int nagg, const int *__restrict__ start, const unsigned long long *__restrict__ keys,
                                                  const double *__restrict__ nv, double *__restrict__ nvc) {
  const int a = blockIdx.x * kAmgWaves + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (a >= nagg) return;
  double s = 0.0;
  for (int p = start[a] + lane; p < start[a + 1]; p += 64) {
    const double x = nv[(unsigned)(keys[p] & 0xffffffffu)];
    s += x * x;
  }
  s = wave_sum(s);
  if (lane == 0) nvc[a] = sqrt(s);
}

__global__ void k_ptent(int n, const int *__restrict__ agg, const double *__restrict__ nv, const double *__restrict__ nvc,
                        double *__restrict__ pt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pt[i] = (agg[i] >= 0 && nvc[agg[i]] > 0.0) ? nv[i] / nvc[agg[i]] : 0.0;
}

// rho = max_i sum_j |a_ij| / |a_ii|  (bit pattern of a non-negative double orders like the number)
__global__ __launch_bounds__(256) void k_amg_rho(int n, const rp_t *__restrict__ rp, const int *__restrict__ ci,
                                                 const double *__restrict__ v, const double *__restrict__ dg,
                                                 volatile unsigned long long *rho_bits) {
  const int i = blockIdx.x * kAmgWaves + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  double s = 0.0;
  for (rp_t p = rp[i] + lane; p < rp[i + 1]; p += 64)
    if (ci[p] < n) s += fabs(v[p]);
  if (dg[i] == 0.0) return;  // empty row of a coarse operator (wave-uniform: see k_sgs_pivots)
  s = wave_sum(s) / fabs(dg[i]);
  // one atomic per row would serialise a million updates of one word: only candidates above the running maximum try
  const unsigned long long bits = (unsigned long long)__double_as_longlong(s);
  if (lane == 0 && bits > *rho_bits) atomicMax(const_cast<unsigned long long *>(rho_bits), bits);
}

// Row i of P = (I - damp D^-1 A) P_tent.  The row's distinct aggregates live in registers, entry e in slot e / 64 of
// lane e % 64 (up to 64 * kProlongSlots of them; a row of the bench matrix touches ~10, a row of a strongly thresholded
// graph more than 64): entries are taken 64 at a time, every distinct aggregate of the chunk is summed with a
// fixed-order wave sum and merged into the table, so the result does not depend on scheduling.  FILL = false only counts.
constexpr int kProlongSlots = 8;

// MODE 0 counts, MODE 1 fills the CSR rows behind prp (the two passes of rounds 2-4: rows of more than 64
// aggregates, and ISPH_AMG_PROLONG_TWO_PASS=1)

// The same row in ONE pass: LPR lanes per row (16: four rows per wave -- the chain row offsets -> columns -> aggregates ->
// table is four dependent round trips whatever the row holds, so four rows in flight per wave finish in the time of one),
// a per-row LDS table of 2 LPR slots (key = aggregate, open addressing) takes the terms as they are read, and the row
// goes, columns ascending, to a scratch row of LPR slots and its length to pcnt (k_rows_compact packs the rows).  A row of more than LPR aggregates raises err bit 8: the caller
// repeats with LPR = 64 and then falls back to the two passes.  The terms of one aggregate are added in the order the
// lanes reach the table (like the products of the SpGEMM that consumes P): the last bits of P are not reproducible from
// run to run, those of the two-pass kernel are.
template <int LPR>
__global__ __launch_bounds__(256) void k_prolong_rows(int n, const rp_t *__restrict__ rp, const int *__restrict__ ci,
                                                      const double *__restrict__ v, const double *__restrict__ dg,
                                                      const int *__restrict__ agg, const double *__restrict__ pt, double damp,
                                                      int *__restrict__ pcnt, int *__restrict__ tci, double *__restrict__ tcv,
                                                      int *__restrict__ err) {
  constexpr int RPW = 64 / LPR, TABLE = 2 * LPR;
  __shared__ int tk[kAmgWaves * RPW][TABLE];
  __shared__ double tv[kAmgWaves * RPW][TABLE];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane / LPR, sub = lane % LPR;
  const int tr = w * RPW + g;
  const int i = (blockIdx.x * kAmgWaves + w) * RPW + g;
  const bool live = i < n;   // the lanes of a row past the end walk along: the wave barriers below are for all 64
  tk[tr][sub] = -1; tk[tr][sub + LPR] = -1;
  tv[tr][sub] = 0.0; tv[tr][sub + LPR] = 0.0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  bool over = false;
  auto insert = [&](int c, double val) {
    int slot = (int)(amg_hash32((unsigned)c) & (TABLE - 1));
    for (int tries = 0;; ++tries) {
      int cur = __atomic_load_n(&tk[tr][slot], __ATOMIC_RELAXED);
      if (cur == -1) cur = atomicCAS(&tk[tr][slot], -1, c);
      if (cur == -1 || cur == c) break;
      slot = (slot + 1) & (TABLE - 1);
      if (tries >= TABLE) { over = true; slot = -1; break; }
    }
    if (slot >= 0) atomicAdd(&tv[tr][slot], val);
  };
  if (live) {
    const int ai = agg[i];
    if (ai >= 0 && sub == 0) insert(ai, pt[i]);
    const double f = dg[i] != 0.0 ? damp / dg[i] : 0.0;
    const rp_t lo = rp[i], hi = rp[i + 1];
    constexpr int U = LPR == 64 ? 2 : 4;   // rounds of loads in flight
    for (rp_t p0 = lo; p0 < hi; p0 += U * LPR) {
      int a[U];
      double term[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const rp_t p = p0 + LPR * u + sub;
        a[u] = -1;
        term[u] = 0.0;
        if (p < hi) {
          const int j = ci[p];
          if (j < n) {
            const int aj = agg[j];
            if (aj >= 0) { a[u] = aj; term[u] = -(f * v[p] * pt[j]); }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (a[u] >= 0) insert(a[u], term[u]);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int k0 = tk[tr][sub], k1 = tk[tr][sub + LPR];
  int r0 = 0, r1 = 0, cnt = 0;
  bool bad = over;
  for (int s2 = 0; s2 < TABLE; ++s2) {
    const int ks = tk[tr][s2];   // the same address for the lanes of a row
    if (ks >= 0) { ++cnt; r0 += ks < k0; r1 += ks < k1; }
  }
  // a lane that met a full table: the row's other lanes learn it from the vote
  unsigned long long votes = __ballot(over);
  if (LPR < 64) votes = (votes >> (g * LPR)) & ((1ull << (LPR & 63)) - 1ull);
  bad = votes != 0 || cnt > LPR;
  if (!live) return;
  if (bad) { if (sub == 0) { atomicOr(err, 8); pcnt[i] = 0; } return; }
  if (k0 >= 0) { tci[(size_t)i * LPR + r0] = k0; tcv[(size_t)i * LPR + r0] = tv[tr][sub]; }
  if (k1 >= 0) { tci[(size_t)i * LPR + r1] = k1; tcv[(size_t)i * LPR + r1] = tv[tr][sub + LPR]; }
  if (sub == 0) pcnt[i] = cnt;
}

// the two-pass kernel (see above)
template <int MODE>
__global__ __launch_bounds__(256) void k_prolong(int n, const rp_t *__restrict__ rp, const int *__restrict__ ci,
                                                 const double *__restrict__ v, const double *__restrict__ dg,
                                                 const int *__restrict__ agg, const double *__restrict__ pt, double damp,
                                                 int *__restrict__ pcnt, const rp_t *__restrict__ prp,
                                                 int *__restrict__ pci, double *__restrict__ pv, int *__restrict__ err) {
  const int i = blockIdx.x * kAmgWaves + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  int tkey[kProlongSlots], cnt = 0;
  double tval[kProlongSlots];
#pragma unroll
  for (int t = 0; t < kProlongSlots; ++t) { tkey[t] = 0x7fffffff; tval[t] = 0.0; }
  const int ai = agg[i];
  if (ai >= 0) { if (lane == 0) { tkey[0] = ai; tval[0] = pt[i]; } cnt = 1; }
  const double f = dg[i] != 0.0 ? damp / dg[i] : 0.0;
  for (rp_t p0 = rp[i]; p0 < rp[i + 1]; p0 += 64) {
    const rp_t p = p0 + lane;
    int a = -1;
    double term = 0.0;
    if (p < rp[i + 1]) {
      const int j = ci[p];
      if (j < n && agg[j] >= 0) { a = agg[j]; term = f * v[p] * pt[j]; }
    }
    unsigned long long live = __ballot(a >= 0);
    while (live) {
      const int src = __ffsll((long long)live) - 1;
      const int acur = __shfl(a, src, 64);
      const bool mine = a == acur;
      const double s = wave_sum(mine ? term : 0.0);
      bool found = false;
#pragma unroll
      for (int t = 0; t < kProlongSlots; ++t) {
        if (64 * t < cnt && !found) {  // wave-uniform
          const unsigned long long hit = __ballot(lane + 64 * t < cnt && tkey[t] == acur);
          if (hit) {
            if (lane == __ffsll((long long)hit) - 1) tval[t] -= s;
            found = true;
          }
        }
      }
      if (!found) {
        if (cnt < 64 * kProlongSlots) {
#pragma unroll
          for (int t = 0; t < kProlongSlots; ++t)
            if (t == (cnt >> 6) && lane == (cnt & 63)) { tkey[t] = acur; tval[t] = -s; }
          ++cnt;
        } else if (lane == 0) {
          atomicOr(err, 1);  // a row touching more aggregates than the table holds
        }
      }
      live &= ~__ballot(mine);
    }
  }
  if (MODE == 0) { if (lane == 0) pcnt[i] = cnt; return; }
  const rp_t obeg = prp[i];
  // rank sort by aggregate id (ascending columns)
  int rank[kProlongSlots];
#pragma unroll
  for (int t = 0; t < kProlongSlots; ++t) rank[t] = 0;
  for (int e = 0; e < cnt; ++e) {
    int ke = 0;
#pragma unroll
    for (int t = 0; t < kProlongSlots; ++t)
      if (t == (e >> 6)) ke = __shfl(tkey[t], e & 63, 64);
#pragma unroll
    for (int t = 0; t < kProlongSlots; ++t) rank[t] += ke < tkey[t];
  }
#pragma unroll
  for (int t = 0; t < kProlongSlots; ++t)
    if (lane + 64 * t < cnt) { pci[obeg + rank[t]] = tkey[t]; pv[obeg + rank[t]] = tval[t]; }
}

// row pointers of the transpose from the sorted (column, row) keys: position p opens every column after its
// predecessor's up to its own (empty columns included); the last position closes the rest
__global__ void k_transpose_rowptr(long long nnz, int ncol, const unsigned long long *__restrict__ keys, rp_t *__restrict__ rp) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nnz) return;
  const int c = (int)(keys[p] >> 32), cprev = p > 0 ? (int)(keys[p - 1] >> 32) : -1;
  for (int cc = cprev + 1; cc <= c; ++cc) rp[cc] = p;
  if (p == nnz - 1)
    for (int cc = c + 1; cc <= ncol; ++cc) rp[cc] = nnz;
}
__global__ void k_transpose_keys(int n, const rp_t *__restrict__ rp, const int *__restrict__ ci,
                                 unsigned long long *__restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (rp_t p = rp[i]; p < rp[i + 1]; ++p) keys[p] = ((unsigned long long)ci[p] << 32) | (unsigned)i;
}
__global__ void k_low32(long long nnz, const unsigned long long *__restrict__ keys, int *__restrict__ out) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < nnz) out[p] = (int)(keys[p] & 0xffffffffu);
}

// C = X * Y, one workgroup per row of C, LDS accumulator: direct-addressed when Y has <= TABLE columns, open
// addressing otherwise.  Thread t walks entries t, t+BS, .. of X's row and the whole Y row behind each.
// FILL = false counts the row's entries.  Output columns are in table order (the SELL conversion sorts rows).
template <int TABLE, int BS, bool FILL>
__global__ __launch_bounds__(BS) void k_spgemm(int n, int ycols, int yrows, const rp_t *__restrict__ xrp,
                                               const int *__restrict__ xci, const double *__restrict__ xv,
                                               const rp_t *__restrict__ yrp, const int *__restrict__ yci,
                                               const double *__restrict__ yv, int *__restrict__ ccnt,
                                               const rp_t *__restrict__ crp, int *__restrict__ cci,
                                               double *__restrict__ cv, int *__restrict__ err) {
  __shared__ int tk[TABLE];
  __shared__ double tv[TABLE];
  __shared__ int s_cnt;
  const int i = blockIdx.x;
  if (i >= n) return;
  const bool dense = ycols <= TABLE;
  for (int s = threadIdx.x; s < TABLE; s += BS) { tk[s] = -1; if (FILL) tv[s] = 0.0; }
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  for (rp_t p = xrp[i] + threadIdx.x; p < xrp[i + 1]; p += BS) {
    const int k = xci[p];
    if (k >= yrows) continue;
    const double xa = xv[p];
    for (rp_t q = yrp[k]; q < yrp[k + 1]; ++q) {
      const int c = yci[q];
      int slot = dense ? c : (int)((amg_hash32((unsigned)c)) & (TABLE - 1));
      if (dense) {
        tk[slot] = c;  // benign race: every writer stores the same value
      } else {
        int tries = 0;
        while (true) {
          const int old = atomicCAS(&tk[slot], -1, c);
          if (old == -1 || old == c) break;
          slot = (slot + 1) & (TABLE - 1);
          if (++tries >= TABLE) { atomicOr(err, 2); slot = -1; break; }  // table full
        }
        if (slot < 0) continue;
      }
      if (FILL) atomicAdd(&tv[slot], xa * yv[q]);
    }
  }
  __syncthreads();
  if (!FILL) {
    int c = 0;
    for (int s = threadIdx.x; s < TABLE; s += BS) c += tk[s] >= 0;
    if (c) atomicAdd(&s_cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) ccnt[i] = s_cnt;
    return;
  }
  const rp_t beg = crp[i];
  for (int s = threadIdx.x; s < TABLE; s += BS)
    if (tk[s] >= 0) {
      const int pos = atomicAdd(&s_cnt, 1);
      cci[beg + pos] = tk[s];
      cv[beg + pos] = tv[s];
    }
}

// inclusive prefix sum over the 64 lanes (DPP: Hillis-Steele inside a row of 16, then the row totals are passed on)
__device__ __forceinline__ int wave_scan_incl_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1 (a lane without a source adds 0)
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);   // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);   // row_bcast:31 into rows 2 and 3
  return v;
}

// C = X * Y for rows of X around a hundred entries and short rows of Y (A * P: ~8 entries): ONE pass, one wave per row of C.
// The products of 64 entries of X's row are laid out flat (prefix sum of the Y row lengths; own[] names the X entry behind
// every product), so a lane takes products x, x + 64, .. whatever the lengths of the Y rows are, and the loads of one
// round do not depend on the round before.  Accumulation in a per-wave LDS hash table of 2 CAP slots; a row of C is
// written to a scratch row of CAP slots with ascending columns (k_rows_compact packs the rows once their lengths are
// scanned).  A row with more than CAP distinct columns raises err bit 2: the caller repeats with a larger CAP.
// (The two-pass kernel below it replaces walked every Y row with one thread, count pass and fill pass: 2.2 + 4.5 ms
// per set-up at 100^3.)
constexpr int kOwnCap = 2048;
template <int CAP>
__global__ __launch_bounds__(256) void k_spgemm_rows(int n, int yrows, const rp_t *__restrict__ xrp, const int *__restrict__ xci,
                                                     const double *__restrict__ xv, const rp_t *__restrict__ yrp,
                                                     const int *__restrict__ yci, const double *__restrict__ yv,
                                                     int *__restrict__ ccnt, int *__restrict__ tci, double *__restrict__ tcv,
                                                     int *__restrict__ err) {
  constexpr int TABLE = 2 * CAP, NT = TABLE / 64;
  __shared__ int tk[4][TABLE];
  __shared__ double tv[4][TABLE];
  __shared__ unsigned char own[4][kOwnCap];
  __shared__ rp_t sbase[4][64];
  __shared__ double sxa[4][64];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + w;
  if (i >= n) return;   // wave-uniform; the kernel has no workgroup barrier
  for (int s = lane; s < TABLE; s += 64) { tk[w][s] = -1; tv[w][s] = 0.0; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  bool over = false;
  auto insert = [&](int c, double val) {
    int slot = (int)(amg_hash32((unsigned)c) & (TABLE - 1));
    for (int tries = 0;; ++tries) {
      // most products meet a slot that holds their column already: a plain read (one broadcast for all the lanes of an
      // address) settles those, the compare-and-swap -- serialised lane by lane on one address -- only runs on an empty slot
      int cur = __atomic_load_n(&tk[w][slot], __ATOMIC_RELAXED);
      if (cur == -1) cur = atomicCAS(&tk[w][slot], -1, c);
      if (cur == -1 || cur == c) break;
      slot = (slot + 1) & (TABLE - 1);
      if (tries >= TABLE) { over = true; slot = -1; break; }
    }
    if (slot >= 0) atomicAdd(&tv[w][slot], val);
  };
  const rp_t x0 = xrp[i], x1 = xrp[i + 1];
  for (rp_t pp = x0; pp < x1; pp += 128) {
    // the X entries and Y row extents of two rounds of 64 are requested together (one dependent chain for both)
    int kk[2], len2[2];
    rp_t qb2[2];
    double xa2[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const rp_t p = pp + 64 * h + lane;
      kk[h] = -1;
      xa2[h] = 0.0;
      if (p < x1) { const int k = xci[p]; if (k < yrows) { kk[h] = k; xa2[h] = xv[p]; } }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      len2[h] = 0;
      qb2[h] = 0;
      if (kk[h] >= 0) { qb2[h] = yrp[kk[h]]; len2[h] = (int)(yrp[kk[h] + 1] - qb2[h]); }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (pp + 64 * h >= x1) break;   // wave-uniform
      const int len = len2[h];
      const rp_t qb = qb2[h];
      const double xa = xa2[h];
      const int incl = wave_scan_incl_i32(len);
      const int T = __builtin_amdgcn_readlane(incl, 63), start = incl - len;
      if (T <= kOwnCap) {
        for (int t = 0; t < len; ++t) own[w][start + t] = (unsigned char)lane;
        sbase[w][lane] = qb - start;
        sxa[w][lane] = xa;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // four rounds of loads are issued before the first insertion waits on its table slot (the compiler keeps loads
        // behind the LDS atomics of the round before when the loop is written round by round)
        for (int xx = 0; xx < T; xx += 256) {
          int c[4];
          double pv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int x = xx + 64 * u + lane;
            c[u] = -1;
            pv[u] = 0.0;
            if (x < T) {
              const int e = own[w][x];
              const rp_t q = sbase[w][e] + x;
              c[u] = yci[q];
              pv[u] = sxa[w][e] * yv[q];
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (c[u] >= 0) insert(c[u], pv[u]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      } else {   // Y rows of hundreds of entries: every lane walks its own
        for (int t = 0; t < len; ++t) insert(yci[qb + t], xa * yv[qb + t]);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  int kt[NT], rank[NT], cnt = 0;
  unsigned long long occ[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    kt[t] = tk[w][lane + 64 * t];
    rank[t] = 0;
    occ[t] = __ballot(kt[t] >= 0);
    cnt += __popcll(occ[t]);
  }
  if (__ballot(over) != 0 || cnt > CAP) {
    if (lane == 0) { atomicOr(err, 2); ccnt[i] = 0; }
    return;
  }
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    unsigned long long m = occ[u];
    while (m) {
      const int src = __ffsll((long long)m) - 1;
      const int ks = __builtin_amdgcn_readlane(kt[u], src);
#pragma unroll
      for (int t = 0; t < NT; ++t) rank[t] += ks < kt[t];
      m &= m - 1;
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (kt[t] >= 0) {
      tci[(size_t)i * CAP + rank[t]] = kt[t];
      tcv[(size_t)i * CAP + rank[t]] = tv[w][lane + 64 * t];
    }
  if (lane == 0) ccnt[i] = cnt;
}

// scratch rows of 1 << cap_shift slots -> CSR, sixteen lanes per row
__global__ void k_rows_compact(int n, int cap_shift, const rp_t *__restrict__ rp, const int *__restrict__ tci,
                               const double *__restrict__ tcv, int *__restrict__ ci, double *__restrict__ v) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int r = (int)(idx >> 4), s0 = (int)(idx & 15);
  if (r >= n) return;
  const rp_t b = rp[r];
  const int len = (int)(rp[r + 1] - b);
  const size_t src = (size_t)r << cap_shift;
  for (int s2 = s0; s2 < len; s2 += 16) { ci[b + s2] = tci[src + s2]; v[b + s2] = tcv[src + s2]; }
}

// y = R x for a CSR matrix with long rows: one wave per row, coalesced reads of the row, fixed-order wave sum
__global__ __launch_bounds__(256) void k_csr_spmv_wave(int n, const rp_t *__restrict__ rp, const int *__restrict__ ci,
                                                       const double *__restrict__ v, const double *__restrict__ x,
                                                       double *__restrict__ y) {
  const int i = blockIdx.x * kAmgWaves + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  double s = 0.0;
  for (rp_t p = rp[i] + lane; p < rp[i + 1]; p += 64) s = fma(v[p], x[ci[p]], s);
  s = wave_sum(s);
  if (lane == 0) y[i] = s;
}

constexpr int kAmgDenseMax = 2048;  // largest coarsest level the dense inverse is formed for
// ---- dense direct solve of the coarsest level (non-singular case) ---------------------------------------
__global__ void k_dense_from_csr(int n, const rp_t *__restrict__ rp, const int *__restrict__ ci,
                                 const double *__restrict__ v, double *__restrict__ aug) {
  // aug = [A | I], row-major n x 2n
  const int i = blockIdx.x;
  for (int c = threadIdx.x; c < 2 * n; c += blockDim.x) aug[(size_t)i * 2 * n + c] = c == n + i ? 1.0 : 0.0;
  __syncthreads();
  for (rp_t p = rp[i] + threadIdx.x; p < rp[i + 1]; p += blockDim.x)
    if (ci[p] < n) aug[(size_t)i * 2 * n + ci[p]] = v[p];
}
// Gauss-Jordan with partial pivoting, ONE launch per elimination step (a level of ~10^3 rows took 2 x 10^3 launches of
// 7 us with a pivot kernel and an elimination kernel per step -- 14 ms of every set-up of the reference's benchmark
// protocol, tgv.xml: Quintic kernel, ML, DoubleDiag).  What makes one launch enough:
//   * every workgroup (= matrix row) finds the pivot itself, from a compact copy of the pivot column that the step
//     before left behind (cur / next take turns);
//   * rows are not swapped and the pivot row is not scaled: pivrow[k] names the row that served column k, and the
//     elimination stored_r -= (stored_r[k] / stored_p[k]) stored_p is the same whether p was scaled or not.  Nobody writes
//     what another workgroup of the same launch reads (the pivot row stays as it is; column k itself is left alone, it is
//     never read again; columns < k of the pivot row are zero);
//   * k_gj_finish divides row pivrow[k] of the right half by its pivot and stores it as row k of the LEFT half, which is
//     where the application kernels read the inverse.
// usedat[r] = the step row r served as pivot (-1: not yet); a value >= k read during step k still means "not yet".
__global__ void k_gj_init(int n, const double *__restrict__ aug, double *__restrict__ col0, int *__restrict__ usedat) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) { col0[r] = aug[(size_t)r * 2 * n]; usedat[r] = -1; }
}
__global__ __launch_bounds__(256) void k_gj_step(int n, int k, double *__restrict__ aug, const double *__restrict__ cur,
                                                 double *__restrict__ next, int *__restrict__ usedat, int *__restrict__ pivrow,
                                                 int *__restrict__ err) {
  __shared__ double sval[256];
  __shared__ int sidx[256];
  __shared__ int s_err;
  const int r = blockIdx.x, t = threadIdx.x, w = 2 * n;
  if (t == 0) s_err = *err;   // read once per workgroup: the exit below is uniform whatever another row raises meanwhile
  __syncthreads();
  if (s_err & 4) return;
  double best = -1.0;
  int bi = 0x7fffffff;
  for (int q = t; q < n; q += 256) {
    const int u = usedat[q];
    if (u >= 0 && u < k) continue;
    const double a = fabs(cur[q]);
    if (a > best) { best = a; bi = q; }   // ascending q per thread: the first maximal row of the thread's share
  }
  sval[t] = best; sidx[t] = bi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) {
      const double a = sval[t + o];
      const int ai = sidx[t + o];
      if (a > sval[t] || (a == sval[t] && ai < sidx[t])) { sval[t] = a; sidx[t] = ai; }
    }
    __syncthreads();
  }
  const int piv = sidx[0];
  if (!(sval[0] > 0.0)) { if (t == 0 && r == 0) atomicOr(err, 4); return; }  // singular coarse operator (every row sees it)
  double *row = aug + (size_t)r * w;
  if (r == piv) {
    if (t == 0) { usedat[r] = k; pivrow[k] = r; if (k + 1 < n) next[r] = row[k + 1]; }
    return;
  }
  const double *prow = aug + (size_t)piv * w;
  const double f = cur[r] / cur[piv];
  if (f != 0.0)
    for (int c = k + 1 + t; c < w; c += 256) row[c] -= f * prow[c];
  __syncthreads();
  if (t == 0 && k + 1 < n) next[r] = row[k + 1];
}
__global__ void k_gj_pivots(int n, const double *__restrict__ aug, const int *__restrict__ pivrow, double *__restrict__ pv) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) pv[k] = aug[(size_t)pivrow[k] * 2 * n + k];
}
__global__ __launch_bounds__(256) void k_gj_finish(int n, double *__restrict__ aug, const int *__restrict__ pivrow,
                                                   const double *__restrict__ pv) {
  const int k = blockIdx.x, w = 2 * n;
  const double *src = aug + (size_t)pivrow[k] * w + n;
  const double d = 1.0 / pv[k];
  for (int c = threadIdx.x; c < n; c += 256) aug[(size_t)k * w + c] = src[c] * d;
}
__global__ __launch_bounds__(256) void k_dense_apply(int n, const double *__restrict__ aug, const double *__restrict__ b,
                                                     double *__restrict__ x) {
  const int i = blockIdx.x * kAmgWaves + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  double s = 0.0;
  for (int c = lane; c < n; c += 64) s += aug[(size_t)i * 2 * n + c] * b[c];   // the inverse: left half (k_gj_finish)
  s = wave_sum(s);
  if (lane == 0) x[i] = s;
}

// ---- the smoother of small coarse levels as dense block inverses ---------------------------------------------------
// On a level of a few thousand rows the chunk-stream sweeps are one wave per 64-row block walking 64 dependent rows twice
// (37 us per application at 4 014 rows, 16 us at 24 rows -- 140 us of every V cycle on levels that hold 0.3 % of the
// entries).  The smoother is linear: M_B^-1 = (D+U_B)^-1 D (D+L_B)^-1 of one block is a dense 64 x 64 matrix, built once
// per set-up (thread t solves for column t: forward sweep, scaling, backward sweep, with the conventions of
// k_sgs_pivots: a zero pivot keeps its unknown at zero) and applied as 64 multiply-adds per row.
constexpr int kSgsDenseMaxRows = 32768;   // 16 MB of inverses at most
// mode 0: the symmetric sweep; 1: the forward sweep alone, (D+L_B)^-1; 2: the backward sweep alone, (D+U_B)^-1
__global__ __launch_bounds__(64) void k_sgs_dense_build(int n, const rp_t *__restrict__ rp, const int *__restrict__ ci,
                                                        const double *__restrict__ v, double *__restrict__ W, int mode) {
  __shared__ double Bm[64][65];   // the block's entries, row i at Bm[i][.]
  __shared__ double Y[64][65];    // Y[i][t]: unknown i of thread t's column
  const int t = threadIdx.x, r0 = blockIdx.x * 64;
  const int m = min(64, n - r0);
  for (int j = 0; j < 64; ++j) Bm[t][j] = 0.0;
  if (t < m)
    for (rp_t p = rp[r0 + t]; p < rp[r0 + t + 1]; ++p) {
      const int j = ci[p] - r0;
      if (j >= 0 && j < m) Bm[t][j] = v[p];
    }
  __syncthreads();
  if (mode == 2) {
    for (int i = 0; i < m; ++i) Y[i][t] = i == t ? 1.0 : 0.0;
  } else {
    // w = (D + L)^-1 e_t, then y = D w
    for (int i = 0; i < m; ++i) {
      double s = i == t ? 1.0 : 0.0;
      for (int j = 0; j < i; ++j) s -= Bm[i][j] * Y[j][t];
      const double d = Bm[i][i];
      Y[i][t] = d != 0.0 ? s / d : 0.0;
    }
    if (mode == 0)
      for (int i = 0; i < m; ++i) Y[i][t] *= Bm[i][i];
  }
  if (mode != 1) {
    // x = (D + U)^-1 y
    for (int i = m - 1; i >= 0; --i) {
      double s = Y[i][t];
      for (int j = i + 1; j < m; ++j) s -= Bm[i][j] * Y[j][t];
      const double d = Bm[i][i];
      Y[i][t] = d != 0.0 ? s / d : 0.0;
    }
  }
  __syncthreads();
  // column-major: W[t * 64 + i] = (M^-1)[i][t]; thread t writes row t of every column (coalesced)
  double *Wb = W + (size_t)blockIdx.x * 4096;
  for (int c = 0; c < 64; ++c) Wb[c * 64 + t] = (t < m && c < m) ? Y[t][c] : 0.0;
}

// z = M_B^-1 r (ACC: z += M_B^-1 r), one wave per block: lane i adds column t times r[t] for t = 0..63
template <bool ACC>
__global__ __launch_bounds__(256) void k_sgs_dense_apply(int n, const double *__restrict__ W, const double *__restrict__ r,
                                                         double *__restrict__ z) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b * 64 >= n) return;
  const int row = b * 64 + lane;
  const double rv = row < n ? r[row] : 0.0;
  const int lo = __double2loint(rv), hi = __double2hiint(rv);
  const double *Wb = W + (size_t)b * 4096 + lane;
  double acc = 0.0;
#pragma unroll
  for (int t = 0; t < 64; ++t) {
    const double rt = __hiloint2double(__builtin_amdgcn_readlane(hi, t), __builtin_amdgcn_readlane(lo, t));
    acc = fma(Wb[t * 64], rt, acc);
  }
  if (row < n) z[row] = ACC ? z[row] + acc : acc;
}

// ---- host side ------------------------------------------------------------------------------------------
inline int amg_scan(isph_ctx *ctx, const int *in, rp_t *out, int n, DevBuf<char> &tmp) {
  size_t bytes = 0;
  ISPH_CHECK_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, (rp_t)0, (size_t)n, rocprim::plus<rp_t>(), ctx->stream));
  ISPH_CHECK(tmp.reserve(bytes > 0 ? bytes : 1));
  ISPH_CHECK_HIP(rocprim::exclusive_scan(tmp.p, bytes, in, out, (rp_t)0, (size_t)n, rocprim::plus<rp_t>(), ctx->stream));
  return ISPH_SUCCESS;
}

// int -> int scan (ids, flags)
inline int amg_scan(isph_ctx *ctx, const int *in, int *out, int n, DevBuf<char> &tmp) {
  size_t bytes = 0;
  ISPH_CHECK_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, 0, (size_t)n, rocprim::plus<int>(), ctx->stream));
  ISPH_CHECK(tmp.reserve(bytes > 0 ? bytes : 1));
  ISPH_CHECK_HIP(rocprim::exclusive_scan(tmp.p, bytes, in, out, 0, (size_t)n, rocprim::plus<int>(), ctx->stream));
  return ISPH_SUCCESS;
}

inline int amg_read_int(isph_ctx *ctx, const int *dev, int *host) {
  ISPH_CHECK_HIP(hipMemcpyAsync(host, dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return ISPH_SUCCESS;
}

inline int amg_read_off(isph_ctx *ctx, const rp_t *dev, long long *host) {
  ISPH_CHECK_HIP(hipMemcpyAsync(host, dev, sizeof(rp_t), hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return ISPH_SUCCESS;
}

inline int amg_wave_grid(int n) { return (n + kAmgWaves - 1) / kAmgWaves; }

// device CSR copy of a SELL matrix (rows stay column-sorted)
// prep (dg, sc, key, rho all given): see k_sell_to_csr_i32<true>
inline int amg_csr_from_sell(isph_ctx *ctx, const Sell &S, DCsr &A, DevBuf<char> &tmp, double *dg = nullptr, int *sc = nullptr,
                             unsigned long long *key = nullptr, unsigned long long *rho = nullptr) {
  A.n = S.nrow; A.m = S.ncol; A.nnz = S.nnz;
  ISPH_CHECK(A.rp.reserve((size_t)S.nrow + 1));
  ISPH_CHECK(A.ci.reserve((size_t)(S.nnz > 0 ? S.nnz : 1)));
  ISPH_CHECK(A.v.reserve((size_t)(S.nnz > 0 ? S.nnz : 1)));
  // exclusive scan over n+1 entries: the last input is ignored by giving the scan n+1 items of a padded copy
  DevTmp<int> len1;
  ISPH_CHECK(len1.reserve((size_t)S.nrow + 1));
  ISPH_CHECK_HIP(hipMemcpyAsync(len1.p, S.rowlen.p, sizeof(int) * (size_t)S.nrow, hipMemcpyDeviceToDevice, ctx->stream));
  ISPH_CHECK_HIP(hipMemsetAsync(len1.p + S.nrow, 0, sizeof(int), ctx->stream));
  int rc = amg_scan(ctx, len1.p, A.rp.p, S.nrow + 1, tmp);
  if (rc == ISPH_SUCCESS && S.nrow > 0) {
    if (dg && sc && key && rho) {
      if (hipMemsetAsync(rho, 0, sizeof(unsigned long long), ctx->stream) != hipSuccess) rc = fail("memset failed", __FILE__, __LINE__);
      hipLaunchKernelGGL((k_sell_to_csr_i32<true>), dim3((S.nslices + 3) / 4), dim3(256), 0, ctx->stream, S.nrow,
                         S.rowlen.p, S.slice_off.p, S.col.p, S.val.p, (const rp_t *)A.rp.p, A.ci.p, A.v.p, dg, sc, key, rho);
    } else {
      hipLaunchKernelGGL((k_sell_to_csr_i32<false>), dim3((S.nslices + 3) / 4), dim3(256), 0, ctx->stream, S.nrow,
                         S.rowlen.p, S.slice_off.p, S.col.p, S.val.p, (const rp_t *)A.rp.p, A.ci.p, A.v.p);
    }
  }
  if (rc == ISPH_SUCCESS && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail("SELL->CSR failed", __FILE__, __LINE__);
  len1.release();
  return rc;
}

// R = P^T through a stable radix sort of (column, row) keys
inline int amg_transpose(isph_ctx *ctx, const DCsr &P, DCsr &R, DevBuf<char> &tmp) {
  R.n = P.m; R.m = P.n; R.nnz = P.nnz;
  const size_t nnz1 = (size_t)(P.nnz > 0 ? P.nnz : 1);
  ISPH_CHECK(R.rp.reserve((size_t)R.n + 1));
  ISPH_CHECK(R.ci.reserve(nnz1));
  ISPH_CHECK(R.v.reserve(nnz1));
  DevTmp<unsigned long long> k0, k1;
  int rc = k0.reserve(nnz1);
  if (rc == ISPH_SUCCESS) rc = k1.reserve(nnz1);
  if (rc == ISPH_SUCCESS && P.nnz > 0) {
    const int gn = (int)((P.nnz + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_transpose_keys, dim3((P.n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, P.n,
                       (const rp_t *)P.rp.p, (const int *)P.ci.p, k0.p);
    // the keys are written row by row, so a STABLE sort by the column bits alone leaves every column's rows ascending
    int cbits = 1;
    while (cbits < 31 && (1ll << cbits) < (long long)R.n) ++cbits;
    const unsigned b0 = 32, b1 = 32 + (unsigned)cbits;
    size_t bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, bytes, k0.p, k1.p, P.v.p, R.v.p, (size_t)P.nnz, b0, b1, ctx->stream) != hipSuccess)
      rc = fail("radix sort sizing failed", __FILE__, __LINE__);
    if (rc == ISPH_SUCCESS) rc = tmp.reserve(bytes > 0 ? bytes : 1);
    if (rc == ISPH_SUCCESS &&
        rocprim::radix_sort_pairs(tmp.p, bytes, k0.p, k1.p, P.v.p, R.v.p, (size_t)P.nnz, b0, b1, ctx->stream) != hipSuccess)
      rc = fail("radix sort failed", __FILE__, __LINE__);
    if (rc == ISPH_SUCCESS) {
      hipLaunchKernelGGL(k_low32, dim3(gn), dim3(kBlock), 0, ctx->stream, P.nnz, (const unsigned long long *)k1.p, R.ci.p);
      hipLaunchKernelGGL(k_transpose_rowptr, dim3(gn), dim3(kBlock), 0, ctx->stream, P.nnz, R.n, (const unsigned long long *)k1.p, R.rp.p);
    }
  } else if (rc == ISPH_SUCCESS) {
    if (hipMemsetAsync(R.rp.p, 0, sizeof(rp_t) * ((size_t)R.n + 1), ctx->stream) != hipSuccess) rc = fail("memset failed", __FILE__, __LINE__);
  }
  if (rc == ISPH_SUCCESS && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail("transpose failed", __FILE__, __LINE__);
  k0.release(); k1.release();
  return rc;
}

template <int TABLE, int BS>
inline int amg_spgemm_t(isph_ctx *ctx, const DCsr &X, const DCsr &Y, DCsr &C, DevBuf<char> &tmp, int *derr) {
  C.n = X.n; C.m = Y.m;
  ISPH_CHECK(C.rp.reserve((size_t)C.n + 1));
  DevTmp<int> cnt;
  ISPH_CHECK(cnt.reserve((size_t)C.n + 1));
  ISPH_CHECK_HIP(hipMemsetAsync(cnt.p, 0, sizeof(int) * ((size_t)C.n + 1), ctx->stream));
  if (C.n > 0)
    hipLaunchKernelGGL((k_spgemm<TABLE, BS, false>), dim3(C.n), dim3(BS), 0, ctx->stream, X.n, Y.m, Y.n, (const rp_t *)X.rp.p,
                       (const int *)X.ci.p, (const double *)X.v.p, (const rp_t *)Y.rp.p, (const int *)Y.ci.p,
                       (const double *)Y.v.p, cnt.p, (const rp_t *)nullptr, (int *)nullptr, (double *)nullptr, derr);
  int rc = amg_scan(ctx, cnt.p, C.rp.p, C.n + 1, tmp);
  long long nnz = 0;
  if (rc == ISPH_SUCCESS) rc = amg_read_off(ctx, C.rp.p + C.n, &nnz);
  C.nnz = nnz;
  if (rc == ISPH_SUCCESS) rc = C.ci.reserve((size_t)(nnz > 0 ? nnz : 1));
  if (rc == ISPH_SUCCESS) rc = C.v.reserve((size_t)(nnz > 0 ? nnz : 1));
  if (rc == ISPH_SUCCESS && C.n > 0)
    hipLaunchKernelGGL((k_spgemm<TABLE, BS, true>), dim3(C.n), dim3(BS), 0, ctx->stream, X.n, Y.m, Y.n, (const rp_t *)X.rp.p,
                       (const int *)X.ci.p, (const double *)X.v.p, (const rp_t *)Y.rp.p, (const int *)Y.ci.p,
                       (const double *)Y.v.p, (int *)nullptr, (const rp_t *)C.rp.p, C.ci.p, C.v.p, derr);
  if (rc == ISPH_SUCCESS && hipGetLastError() != hipSuccess) rc = fail("SpGEMM launch failed", __FILE__, __LINE__);
  cnt.release();
  return rc;
}

// A*P: rows see few distinct aggregates, so small tables are tried first (clearing the table is most of the cost);
// an overflow of the table is reported by the kernel and the next size is used
template <int TABLE>
inline int amg_spgemm_try(isph_ctx *ctx, const DCsr &A, const DCsr &P, DCsr &AP, DevBuf<char> &tmp, int *derr, bool *overflow) {
  int before = 0, after = 0;
  ISPH_CHECK(amg_read_int(ctx, derr, &before));
  ISPH_CHECK((amg_spgemm_t<TABLE, 64>(ctx, A, P, AP, tmp, derr)));
  ISPH_CHECK(amg_read_int(ctx, derr, &after));
  *overflow = (after & 2) && !(before & 2);
  if (*overflow) {
    after &= ~2;
    ISPH_CHECK_HIP(hipMemcpyAsync(derr, &after, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    AP.release();
  }
  return ISPH_SUCCESS;
}
// the one-pass kernel with scratch rows of CAP slots; *overflow: a row has more distinct columns (nothing is kept)
template <int CAP>
inline int amg_spgemm_rows(isph_ctx *ctx, const DCsr &X, const DCsr &Y, DCsr &C, DevBuf<char> &tmp, int *derr, bool *overflow) {
  constexpr int SHIFT = CAP == 64 ? 6 : 8;
  static_assert(CAP == 64 || CAP == 256, "scratch rows of 64 or 256 slots");
  *overflow = false;
  C.n = X.n; C.m = Y.m;
  ISPH_CHECK(C.rp.reserve((size_t)C.n + 1));
  DevTmp<int> cnt, tci;
  DevTmp<double> tcv;
  const size_t slots = (size_t)(C.n > 0 ? C.n : 1) * CAP;
  ISPH_CHECK(cnt.reserve((size_t)C.n + 1));
  ISPH_CHECK(tci.reserve(slots));
  ISPH_CHECK(tcv.reserve(slots));
  ISPH_CHECK_HIP(hipMemsetAsync(cnt.p + C.n, 0, sizeof(int), ctx->stream));
  if (C.n > 0)
    hipLaunchKernelGGL((k_spgemm_rows<CAP>), dim3((C.n + 3) / 4), dim3(256), 0, ctx->stream, X.n, Y.n, (const rp_t *)X.rp.p,
                       (const int *)X.ci.p, (const double *)X.v.p, (const rp_t *)Y.rp.p, (const int *)Y.ci.p,
                       (const double *)Y.v.p, cnt.p, tci.p, tcv.p, derr);
  ISPH_CHECK(amg_scan(ctx, cnt.p, C.rp.p, C.n + 1, tmp));
  long long nnz = 0;
  int herr = 0;
  ISPH_CHECK_HIP(hipMemcpyAsync(&nnz, C.rp.p + C.n, sizeof(rp_t), hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK(amg_read_int(ctx, derr, &herr));
  if (herr & 2) {
    herr &= ~2;
    ISPH_CHECK_HIP(hipMemcpyAsync(derr, &herr, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    C.release();
    *overflow = true;
    return ISPH_SUCCESS;
  }
  C.nnz = nnz;
  ISPH_CHECK(C.ci.reserve((size_t)(nnz > 0 ? nnz : 1)));
  ISPH_CHECK(C.v.reserve((size_t)(nnz > 0 ? nnz : 1)));
  if (C.n > 0) {
    const long long total = (long long)C.n * 16;
    hipLaunchKernelGGL(k_rows_compact, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, C.n, SHIFT,
                       (const rp_t *)C.rp.p, (const int *)tci.p, (const double *)tcv.p, C.ci.p, C.v.p);
  }
  return ISPH_SUCCESS;   // (the scratch rows go back to the pool, which hands a block out again only after a device synchronisation)
}

inline int amg_spgemm_ap(isph_ctx *ctx, const DCsr &A, const DCsr &P, DCsr &AP, DevBuf<char> &tmp, int *derr) {
  bool overflow = false;
  const char *two = getenv("ISPH_AMG_SPGEMM_TWO_PASS");   // the kernels of rounds 2-4, kept for comparisons
  if (!(two && two[0] == '1')) {
    ISPH_CHECK(amg_spgemm_rows<64>(ctx, A, P, AP, tmp, derr, &overflow));
    if (!overflow) return ISPH_SUCCESS;
    ISPH_CHECK(amg_spgemm_rows<256>(ctx, A, P, AP, tmp, derr, &overflow));
    if (!overflow) return ISPH_SUCCESS;
  }
  ISPH_CHECK(amg_spgemm_try<256>(ctx, A, P, AP, tmp, derr, &overflow));
  if (!overflow) return ISPH_SUCCESS;
  ISPH_CHECK(amg_spgemm_try<1024>(ctx, A, P, AP, tmp, derr, &overflow));
  if (!overflow) return ISPH_SUCCESS;
  return amg_spgemm_t<4096, 64>(ctx, A, P, AP, tmp, derr);
}

inline void amg_level_destroy(AmgLevel *L) {
  if (!L) return;
  L->A.release(); L->P.release(); L->R.release();
  if (L->Aown) isph_mat_destroy(L->Aown);
  if (L->Pm) isph_mat_destroy(L->Pm);
  if (L->APm) isph_mat_destroy(L->APm);
  if (L->sgs) ilu_destroy(L->sgs);
  L->wsgs.release(); L->wfwd.release(); L->wbwd.release();
  L->agg.release(); L->nv.release(); L->x.release(); L->b.release(); L->r.release(); L->z.release();
  delete L;
}

inline void amg_destroy(isph_amg *G) {
  if (!G) return;
  for (auto *L : G->L) amg_level_destroy(L);
  G->cinv.release(); G->bglob.release();
  delete G;
}

// aggregates of level L (sets L->agg) ; returns the number of aggregates in *nagg_out
// theta == 0: also fills dg and *rho (k_amg_prepare); otherwise dg is read and rho is left to amg_prolongator
// prep_sc / prep_key: the strong columns and first keys are already there (amg_csr_from_sell with its prep outputs)
inline int amg_aggregate(isph_ctx *ctx, AmgLevel *L, double *dg, unsigned long long *rho, double theta, DevBuf<char> &tmp, int *nagg_out,
                         int *prep_sc = nullptr, unsigned long long *prep_key = nullptr) {
  const DCsr &A = L->A;
  const int n = A.n;
  const double th2 = theta * theta;
  const bool prepared = prep_sc != nullptr && prep_key != nullptr;
  DevTmp<unsigned long long> key_own, t1, t2;
  DevTmp<int> flag, id, a1, cnt, scb_own, listA, listB, list1, stamp;
  int rc = prepared ? ISPH_SUCCESS : key_own.reserve((size_t)n);
  if (rc == ISPH_SUCCESS) rc = listA.reserve((size_t)n);
  if (rc == ISPH_SUCCESS) rc = listB.reserve((size_t)n);
  if (rc == ISPH_SUCCESS) rc = list1.reserve((size_t)n);
  if (rc == ISPH_SUCCESS) rc = stamp.reserve((size_t)n);
  if (rc == ISPH_SUCCESS && hipMemsetAsync(stamp.p, 0, sizeof(int) * (size_t)n, ctx->stream) != hipSuccess)
    rc = fail("memset failed", __FILE__, __LINE__);
  if (rc == ISPH_SUCCESS && !prepared) rc = scb_own.reserve((size_t)(A.nnz > 0 ? A.nnz : 1));
  unsigned long long *const keyp = prepared ? prep_key : key_own.p;
  int *const scp = prepared ? prep_sc : scb_own.p;
  if (rc == ISPH_SUCCESS) rc = t1.reserve((size_t)n);
  if (rc == ISPH_SUCCESS) rc = t2.reserve((size_t)n);
  if (rc == ISPH_SUCCESS) rc = flag.reserve((size_t)n + 1);
  if (rc == ISPH_SUCCESS) rc = id.reserve((size_t)n + 1);
  if (rc == ISPH_SUCCESS) rc = a1.reserve((size_t)n);
  if (rc == ISPH_SUCCESS) rc = cnt.reserve(4);
  if (rc == ISPH_SUCCESS) rc = L->agg.reserve((size_t)n);
  const int gw = amg_wave_grid(n), gt = (n + kBlock - 1) / kBlock;
  const rp_t *rp = A.rp.p;
  const int *ci = A.ci.p;
  const double *v = A.v.p;
  if (rc == ISPH_SUCCESS) {
    if (prepared) {
    } else if (th2 == 0.0) {
      if (hipMemsetAsync(rho, 0, sizeof(unsigned long long), ctx->stream) != hipSuccess) rc = fail("memset failed", __FILE__, __LINE__);
      hipLaunchKernelGGL(k_amg_prepare, dim3(gw), dim3(256), 0, ctx->stream, n, rp, ci, v, dg, scp, keyp, rho);
    } else {
      hipLaunchKernelGGL(k_strong_cols, dim3(gw), dim3(256), 0, ctx->stream, n, rp, ci, v, (const double *)dg, th2, scp);
      hipLaunchKernelGGL(k_mis_init, dim3(gw), dim3(256), 0, ctx->stream, n, rp, (const int *)scp, keyp);
    }
    // rounds: t1 = max over the strong neighbourhood, t2 = max of t1 (distance 2), decide.  From the second round on
    // the sweeps run over work lists (see k_mis_mark)
    int und = n, cur = 0;
    for (int round = 0; round < 1000 && rc == ISPH_SUCCESS; ++round) {
      const int *clist = round == 0 ? nullptr : (cur ? listB.p : listA.p);
      int *nlist = round == 0 ? listA.p : (cur ? listA.p : listB.p);
      const int *ccnt = cnt.p + cur;
      int *ncnt = cnt.p + (round == 0 ? 0 : 1 - cur);
      if (clist && (long long)und * 4 <= n) {
        if (hipMemsetAsync(cnt.p + 2, 0, sizeof(int), ctx->stream) != hipSuccess) { rc = fail("memset failed", __FILE__, __LINE__); break; }
        hipLaunchKernelGGL(k_mis_mark, dim3(std::min(kMisGrid, (und + 31) / 32)), dim3(256), 0, ctx->stream, ccnt, clist, rp,
                           (const int *)scp, stamp.p, round, list1.p, cnt.p + 2);
        hipLaunchKernelGGL(k_mis_max_list, dim3(kMisGrid), dim3(256), 0, ctx->stream, (const int *)(cnt.p + 2),
                           (const int *)list1.p, rp, (const int *)scp, (const unsigned long long *)keyp, t1.p);
      } else {
        hipLaunchKernelGGL(k_mis_max, dim3(gw), dim3(256), 0, ctx->stream, n, rp, (const int *)scp,
                           (const unsigned long long *)keyp, t1.p, (const unsigned long long *)nullptr);
      }
      if (clist)
        hipLaunchKernelGGL(k_mis_max_list, dim3(std::min(kMisGrid, amg_wave_grid(und))), dim3(256), 0, ctx->stream, ccnt, clist,
                           rp, (const int *)scp, (const unsigned long long *)t1.p, t2.p);
      else
        hipLaunchKernelGGL(k_mis_max, dim3(gw), dim3(256), 0, ctx->stream, n, rp, (const int *)scp,
                           (const unsigned long long *)t1.p, t2.p, (const unsigned long long *)keyp);
      if (hipMemsetAsync(ncnt, 0, sizeof(int), ctx->stream) != hipSuccess) { rc = fail("memset failed", __FILE__, __LINE__); break; }
      const int nd = clist ? und : n;
      hipLaunchKernelGGL(k_mis_decide_list, dim3(std::min(kMisGrid, (nd + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream,
                         ccnt, clist, n, keyp, (const unsigned long long *)t2.p, nlist, ncnt);
      rc = amg_read_int(ctx, ncnt, &und);
      if (round > 0) cur = 1 - cur;
      if (und == 0) break;
      if (round == 999) rc = fail("MIS did not terminate", __FILE__, __LINE__);
    }
  }
  int nroot = 0, nleft = 0, last = 0;
  if (rc == ISPH_SUCCESS) {
    hipLaunchKernelGGL(k_flag_roots, dim3(gt), dim3(kBlock), 0, ctx->stream, n, (const unsigned long long *)keyp, flag.p);
    if (hipMemsetAsync(flag.p + n, 0, sizeof(int), ctx->stream) != hipSuccess) rc = fail("memset failed", __FILE__, __LINE__);
  }
  if (rc == ISPH_SUCCESS) rc = amg_scan(ctx, flag.p, id.p, n + 1, tmp);
  if (rc == ISPH_SUCCESS) rc = amg_read_int(ctx, id.p + n, &nroot);
  if (rc == ISPH_SUCCESS) {
    hipLaunchKernelGGL(k_agg_pass1, dim3(gw), dim3(256), 0, ctx->stream, n, rp, (const int *)scp,
                       (const unsigned long long *)keyp, (const int *)id.p, a1.p);
    hipLaunchKernelGGL(k_agg_pass2, dim3(gw), dim3(256), 0, ctx->stream, n, rp, (const int *)scp, v, (const int *)a1.p,
                       L->agg.p, flag.p);
    rc = amg_scan(ctx, flag.p, id.p, n + 1, tmp);
  }
  if (rc == ISPH_SUCCESS) rc = amg_read_int(ctx, id.p + n, &nleft);
  (void)last;
  if (rc == ISPH_SUCCESS && nleft > 0)
    hipLaunchKernelGGL(k_agg_pass3, dim3(gt), dim3(kBlock), 0, ctx->stream, n, (const int *)flag.p, (const int *)id.p, nroot,
                       L->agg.p);
  if (rc == ISPH_SUCCESS && hipGetLastError() != hipSuccess) rc = fail("aggregation kernels failed", __FILE__, __LINE__);
  *nagg_out = nroot + nleft;
  key_own.release(); t1.release(); t2.release(); flag.release(); id.release(); a1.release(); cnt.release(); scb_own.release();
  listA.release(); listB.release(); list1.release(); stamp.release();
  return rc;
}

// P = (I - omega/rho D^-1 A) P_tent and the coarse null vector
// rho_ready: the device word already holds rho (amg_aggregate with threshold 0)
inline int amg_prolongator(isph_ctx *ctx, AmgLevel *L, const double *dg, unsigned long long *rho_dev, bool rho_ready, int nagg, double omega,
                           DevBuf<double> &nvc, DevBuf<char> &tmp, int *derr) {
  const DCsr &A = L->A;
  const int n = A.n;
  DevTmp<unsigned long long> k0, k1;
  DevTmp<int> start, cnt;
  DevTmp<double> pt;
  int rc = k0.reserve((size_t)n);
  if (rc == ISPH_SUCCESS) rc = k1.reserve((size_t)n);
  if (rc == ISPH_SUCCESS) rc = start.reserve((size_t)nagg + 1);
  if (rc == ISPH_SUCCESS) rc = cnt.reserve((size_t)n + 1);
  if (rc == ISPH_SUCCESS) rc = pt.reserve((size_t)n);
  if (rc == ISPH_SUCCESS) rc = nvc.reserve((size_t)nagg);
  const int gw = amg_wave_grid(n), gt = (n + kBlock - 1) / kBlock;
  if (rc == ISPH_SUCCESS) {
    hipLaunchKernelGGL(k_member_keys, dim3(gt), dim3(kBlock), 0, ctx->stream, n, (const int *)L->agg.p, k0.p);
    // keys are made in index order: a STABLE sort by the aggregate bits alone leaves the members of an aggregate ascending
    // (nodes outside every aggregate carry all ones there: one value more than the aggregates need)
    int abits = 1;
    while (abits < 32 && (1ll << abits) <= (long long)nagg) ++abits;
    const unsigned b0 = 32, b1 = 32 + (unsigned)abits;
    size_t bytes = 0;
    if (rocprim::radix_sort_keys(nullptr, bytes, k0.p, k1.p, (size_t)n, b0, b1, ctx->stream) != hipSuccess)
      rc = fail("radix sort sizing failed", __FILE__, __LINE__);
    if (rc == ISPH_SUCCESS) rc = tmp.reserve(bytes > 0 ? bytes : 1);
    if (rc == ISPH_SUCCESS && rocprim::radix_sort_keys(tmp.p, bytes, k0.p, k1.p, (size_t)n, b0, b1, ctx->stream) != hipSuccess)
      rc = fail("radix sort failed", __FILE__, __LINE__);
  }
  double rho_h = 0.0;
  if (rc == ISPH_SUCCESS) {
    hipLaunchKernelGGL(k_segment_starts, dim3(gt), dim3(kBlock), 0, ctx->stream, n, nagg, (const unsigned long long *)k1.p, start.p);
    hipLaunchKernelGGL(k_agg_norm, dim3(amg_wave_grid(nagg)), dim3(256), 0, ctx->stream, nagg, (const int *)start.p,
                       (const unsigned long long *)k1.p, (const double *)L->nv.p, nvc.p);
    hipLaunchKernelGGL(k_ptent, dim3(gt), dim3(kBlock), 0, ctx->stream, n, (const int *)L->agg.p, (const double *)L->nv.p,
                       (const double *)nvc.p, pt.p);
    if (!rho_ready) {
      if (hipMemsetAsync(rho_dev, 0, sizeof(unsigned long long), ctx->stream) != hipSuccess) rc = fail("memset failed", __FILE__, __LINE__);
      hipLaunchKernelGGL(k_amg_rho, dim3(gw), dim3(256), 0, ctx->stream, n, (const rp_t *)A.rp.p, (const int *)A.ci.p,
                         (const double *)A.v.p, dg, rho_dev);
    }
    if (rc == ISPH_SUCCESS &&
        (hipMemcpyAsync(&rho_h, rho_dev, sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
         hipStreamSynchronize(ctx->stream) != hipSuccess))
      rc = fail("rho read-back failed", __FILE__, __LINE__);
  }
  const double damp = rho_h > 0.0 ? omega / rho_h : 0.0;
  DCsr &P = L->P;
  P.n = n; P.m = nagg;
  if (rc == ISPH_SUCCESS) rc = P.rp.reserve((size_t)n + 1);
  const rp_t *arp = A.rp.p;
  const int *aci = A.ci.p;
  const double *av = A.v.p;
  const int *agg = L->agg.p;
  long long nnz = 0;
  bool one_pass = false;
  const char *two = getenv("ISPH_AMG_PROLONG_TWO_PASS");
  for (int attempt = 0; attempt < 2 && rc == ISPH_SUCCESS && !one_pass && !(two && two[0] == '1'); ++attempt) {
    // one pass into scratch rows of 16 slots (four rows per wave), then of 64 (one row per wave)
    const int shift = attempt == 0 ? 4 : 6, lpr = 1 << shift;
    int herr = 0;
    DevTmp<int> ps_ci;
    DevTmp<double> ps_v;
    rc = ps_ci.reserve((size_t)n << shift);
    if (rc == ISPH_SUCCESS) rc = ps_v.reserve((size_t)n << shift);
    if (rc != ISPH_SUCCESS) break;
    if (hipMemsetAsync(cnt.p + n, 0, sizeof(int), ctx->stream) != hipSuccess) rc = fail("memset failed", __FILE__, __LINE__);
    const int rows_wg = kAmgWaves * (64 / lpr), grid = (n + rows_wg - 1) / rows_wg;
    if (attempt == 0)
      hipLaunchKernelGGL((k_prolong_rows<16>), dim3(grid), dim3(256), 0, ctx->stream, n, arp, aci, av, dg, agg, (const double *)pt.p, damp,
                         cnt.p, ps_ci.p, ps_v.p, derr);
    else
      hipLaunchKernelGGL((k_prolong_rows<64>), dim3(grid), dim3(256), 0, ctx->stream, n, arp, aci, av, dg, agg, (const double *)pt.p, damp,
                         cnt.p, ps_ci.p, ps_v.p, derr);
    if (rc == ISPH_SUCCESS) rc = amg_scan(ctx, cnt.p, P.rp.p, n + 1, tmp);
    if (rc == ISPH_SUCCESS && hipMemcpyAsync(&nnz, P.rp.p + n, sizeof(rp_t), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
      rc = fail("copy failed", __FILE__, __LINE__);
    if (rc == ISPH_SUCCESS) rc = amg_read_int(ctx, derr, &herr);
    if (rc == ISPH_SUCCESS && (herr & 8)) {   // a row longer than the scratch rows
      herr &= ~8;
      if (hipMemcpyAsync(derr, &herr, sizeof(int), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
          hipStreamSynchronize(ctx->stream) != hipSuccess)
        rc = fail("copy failed", __FILE__, __LINE__);
    } else if (rc == ISPH_SUCCESS) {
      one_pass = true;
      P.nnz = nnz;
      rc = P.ci.reserve((size_t)(nnz > 0 ? nnz : 1));
      if (rc == ISPH_SUCCESS) rc = P.v.reserve((size_t)(nnz > 0 ? nnz : 1));
      if (rc == ISPH_SUCCESS && n > 0) {
        const long long total = (long long)n * 16;
        hipLaunchKernelGGL(k_rows_compact, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, n, shift,
                           (const rp_t *)P.rp.p, (const int *)ps_ci.p, (const double *)ps_v.p, P.ci.p, P.v.p);
      }
    }
  }
  if (rc == ISPH_SUCCESS && !one_pass) {
    if (hipMemsetAsync(cnt.p, 0, sizeof(int) * ((size_t)n + 1), ctx->stream) != hipSuccess) rc = fail("memset failed", __FILE__, __LINE__);
    hipLaunchKernelGGL((k_prolong<0>), dim3(gw), dim3(256), 0, ctx->stream, n, arp, aci, av, dg, agg, (const double *)pt.p, damp,
                       cnt.p, (const rp_t *)nullptr, (int *)nullptr, (double *)nullptr, derr);
    if (rc == ISPH_SUCCESS) rc = amg_scan(ctx, cnt.p, P.rp.p, n + 1, tmp);
    if (rc == ISPH_SUCCESS) rc = amg_read_off(ctx, P.rp.p + n, &nnz);
    P.nnz = nnz;
    if (rc == ISPH_SUCCESS) rc = P.ci.reserve((size_t)(nnz > 0 ? nnz : 1));
    if (rc == ISPH_SUCCESS) rc = P.v.reserve((size_t)(nnz > 0 ? nnz : 1));
    if (rc == ISPH_SUCCESS)
      hipLaunchKernelGGL((k_prolong<1>), dim3(gw), dim3(256), 0, ctx->stream, n, arp, aci, av, dg, agg, (const double *)pt.p, damp,
                         (int *)nullptr, (const rp_t *)P.rp.p, P.ci.p, P.v.p, derr);
  }
  if (rc == ISPH_SUCCESS && hipGetLastError() != hipSuccess) rc = fail("prolongator kernels failed", __FILE__, __LINE__);
  k0.release(); k1.release(); start.release(); cnt.release(); pt.release();
  return rc;
}


// ---- coarse levels across ranks --------------------------------------------------------------------------------------
// ML's "Uncoupled" aggregation keeps aggregates (and here the smoothed prolongator's columns) on the rank, but its
// Galerkin operator P^T A P is the product with the WHOLE A (precond_ml.h:49, ML_Gen_MGHierarchy): a coarse row couples
// to the neighbours' aggregates through A's ghost columns.  What that takes on top of the rank-local set-up:
//   * the rows of P that belong to A's ghost columns -- every rank sends the P rows of its send-list nodes with the fine
//     level's own halo plan (a fixed record per node: length, the peer's coarse list size, K column positions, K values;
//     K = longest such row over all ranks), the columns as positions in the ascending list of coarse unknowns the rank
//     will send to that peer from now on -- which IS the coarse level's send list;
//   * P extended by those rows (columns nagg + offset of the peer + position), so that A P and R (A P) come out of the
//     same SpGEMM kernels with ghost columns in the coarse operator;
//   * the coarse operator's halo plan (same peers; the send lists above; receive counts as announced by the peers).
// The cycle's coarse SpMVs then exchange like the fine one does (spmv_dev).
__global__ void k_amg_send_rowmax(int nsend, const int *__restrict__ send_idx, const rp_t *__restrict__ prp, int *__restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nsend) return;
  const int i = send_idx[k], len = (int)(prp[i + 1] - prp[i]);
  if (len > *(volatile int *)out) atomicMax(out, len);
}
__global__ void k_amg_mark_cols(int s0, int s1, const int *__restrict__ send_idx, const rp_t *__restrict__ prp,
                                const int *__restrict__ pci, int *__restrict__ flag) {
  const int k = s0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= s1) return;
  const int i = send_idx[k];
  for (rp_t p = prp[i]; p < prp[i + 1]; ++p) flag[pci[p]] = 1;
}
__global__ void k_amg_compact(int n, const int *__restrict__ flag, const int *__restrict__ pos, int *__restrict__ list) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n && flag[c]) list[pos[c]] = c;
}
// record of send slot k: [len, count, pos_0 .. pos_{K-1}, val_0 .. val_{K-1}]
__global__ void k_amg_pack_rows(int s0, int s1, const int *__restrict__ send_idx, const rp_t *__restrict__ prp,
                                const int *__restrict__ pci, const double *__restrict__ pv, const int *__restrict__ pos,
                                int K, int count, double *__restrict__ buf) {
  const int k = s0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= s1) return;
  const int i = send_idx[k];
  const rp_t b = prp[i];
  const int len = (int)(prp[i + 1] - b);
  double *rec = buf + (size_t)k * (2 + 2 * (size_t)K);
  rec[0] = (double)len; rec[1] = (double)count;
  for (int t = 0; t < K; ++t) {
    rec[2 + t] = t < len ? (double)pos[pci[b + t]] : 0.0;
    rec[2 + K + t] = t < len ? pv[b + t] : 0.0;
  }
}
__global__ void k_amg_ext_len(int n, int nrecv, const rp_t *__restrict__ prp, const double *__restrict__ rbuf, int K, int *__restrict__ len) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) len[q] = (int)(prp[q + 1] - prp[q]);
  else if (q < n + nrecv) len[q] = (int)rbuf[(size_t)(q - n) * (2 + 2 * (size_t)K)];
  else if (q == n + nrecv) len[q] = 0;
}
__global__ void k_amg_ext_fill(int n, int nrecv, const rp_t *__restrict__ prp, const int *__restrict__ pci, const double *__restrict__ pv,
                               const double *__restrict__ rbuf, int K, const int *__restrict__ slot_col0, const rp_t *__restrict__ erp,
                               int *__restrict__ eci, double *__restrict__ ev) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n + nrecv) return;
  const rp_t w = erp[q];
  if (q < n) {
    const rp_t b = prp[q];
    const int len = (int)(prp[q + 1] - b);
    for (int t = 0; t < len; ++t) { eci[w + t] = pci[b + t]; ev[w + t] = pv[b + t]; }
  } else {
    const double *rec = rbuf + (size_t)(q - n) * (2 + 2 * (size_t)K);
    const int len = (int)rec[0], c0 = slot_col0[q - n];
    for (int t = 0; t < len; ++t) { eci[w + t] = c0 + (int)rec[2 + t]; ev[w + t] = rec[2 + K + t]; }
  }
}

// in-place all-reduce of a few host numbers (op 0 sum, 1 max); every rank of the communicator calls it
inline int amg_host_allreduce(isph_ctx *ctx, double *h, int count, int op) {
  DevTmp<double> d;
  ISPH_CHECK(d.reserve((size_t)count));
  ISPH_CHECK_HIP(hipMemcpyAsync(d.p, h, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
  ISPH_CHECK(comm_allreduce(ctx, d.p, count, op, ctx->stream));
  ISPH_CHECK_HIP(hipMemcpyAsync(h, d.p, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return ISPH_SUCCESS;
}

// P (n x nagg, this rank's rows) -> Pext ((n + ghosts of A) x (nagg + ghosts of the coarse level)) and the coarse halo
// lists.  K: longest P row on any send list of any rank (agreed by the caller).
// Two halves, so that the ranks can agree between them: everything of the first half that can fail on one rank alone
// (allocations, scans, read-backs) happens BEFORE any rank enters the point-to-point exchange of the second half -- a rank
// that returned early from a single function left its peers waiting in that exchange for ever (amg_create holds a
// consensus all-reduce between the halves).
struct AmgExtendState {
  DevTmp<double> sbuf, rbuf;
  DevTmp<int> flag, pos, list;
};
inline int amg_extend_pack(isph_ctx *ctx, const isph_halo &H, const DCsr &P, int K, AmgExtendState &St, std::vector<int> &csend_ptr,
                           std::vector<int> &csend_idx, DevBuf<char> &tmp) {
  const int nagg = P.m, np = H.npeers, nsend = H.nsend, nrecv = H.nrecv;
  const size_t rec = 2 + 2 * (size_t)K;
  DevTmp<double> &sbuf = St.sbuf, &rbuf = St.rbuf;
  DevTmp<int> &flag = St.flag, &pos = St.pos, &list = St.list;
  ISPH_CHECK(sbuf.reserve(std::max<size_t>((size_t)nsend * rec, 1)));
  ISPH_CHECK(rbuf.reserve(std::max<size_t>((size_t)nrecv * rec, 1)));
  ISPH_CHECK(flag.reserve((size_t)nagg + 1));
  ISPH_CHECK(pos.reserve((size_t)nagg + 1));
  ISPH_CHECK(list.reserve((size_t)std::max(nagg, 1)));
  csend_ptr.assign((size_t)np + 1, 0);
  csend_idx.clear();
  std::vector<int> hl;
  for (int p = 0; p < np; ++p) {
    const int s0 = H.send_ptr[(size_t)p], s1 = H.send_ptr[(size_t)p + 1];
    int count = 0;
    if (s1 > s0) {
      ISPH_CHECK_HIP(hipMemsetAsync(flag.p, 0, sizeof(int) * ((size_t)nagg + 1), ctx->stream));
      hipLaunchKernelGGL(k_amg_mark_cols, dim3((s1 - s0 + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, s0, s1, (const int *)H.send_idx.p,
                         (const rp_t *)P.rp.p, (const int *)P.ci.p, flag.p);
      ISPH_CHECK(amg_scan(ctx, (const int *)flag.p, pos.p, nagg + 1, tmp));
      ISPH_CHECK(amg_read_int(ctx, pos.p + nagg, &count));
      if (count > 0) {
        hipLaunchKernelGGL(k_amg_compact, dim3((nagg + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, nagg, (const int *)flag.p,
                           (const int *)pos.p, list.p);
        hl.resize((size_t)count);
        ISPH_CHECK_HIP(hipMemcpyAsync(hl.data(), list.p, sizeof(int) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
      }
      hipLaunchKernelGGL(k_amg_pack_rows, dim3((s1 - s0 + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, s0, s1, (const int *)H.send_idx.p,
                         (const rp_t *)P.rp.p, (const int *)P.ci.p, (const double *)P.v.p, (const int *)pos.p, K, count, sbuf.p);
      ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      if (count > 0) csend_idx.insert(csend_idx.end(), hl.begin(), hl.end());
    }
    csend_ptr[(size_t)p + 1] = csend_ptr[(size_t)p] + count;
  }
  return ISPH_SUCCESS;
}
inline int amg_extend_finish(isph_ctx *ctx, const isph_halo &H, const DCsr &P, int K, AmgExtendState &St, DCsr &Pext,
                             std::vector<int> &crecv_ptr, DevBuf<char> &tmp) {
  const int n = P.n, nagg = P.m, np = H.npeers, nrecv = H.nrecv;
  const size_t rec = 2 + 2 * (size_t)K;
  DevTmp<double> &sbuf = St.sbuf, &rbuf = St.rbuf;
  DevTmp<int> slot0, len;
  ISPH_CHECK(comm_exchange(ctx, H, sbuf.p, rbuf.p, (int)rec, false, ctx->stream));
  // what every peer will send from now on: announced in each of its records
  crecv_ptr.assign((size_t)np + 1, 0);
  std::vector<double> cnt((size_t)np, 0.0);
  for (int p = 0; p < np; ++p) {
    const int r0 = H.recv_ptr[(size_t)p], r1 = H.recv_ptr[(size_t)p + 1];
    if (r1 > r0) ISPH_CHECK_HIP(hipMemcpyAsync(&cnt[(size_t)p], rbuf.p + (size_t)r0 * rec + 1, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  }
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  std::vector<int> hslot((size_t)std::max(nrecv, 1), 0);
  for (int p = 0; p < np; ++p) {
    crecv_ptr[(size_t)p + 1] = crecv_ptr[(size_t)p] + (int)cnt[(size_t)p];
    for (int k = H.recv_ptr[(size_t)p]; k < H.recv_ptr[(size_t)p + 1]; ++k) hslot[(size_t)k] = nagg + crecv_ptr[(size_t)p];
  }
  const int ncg = crecv_ptr[(size_t)np];
  ISPH_CHECK(slot0.reserve((size_t)std::max(nrecv, 1)));
  if (nrecv > 0) ISPH_CHECK_HIP(hipMemcpyAsync(slot0.p, hslot.data(), sizeof(int) * (size_t)nrecv, hipMemcpyHostToDevice, ctx->stream));
  const int ne = n + nrecv;
  ISPH_CHECK(len.reserve((size_t)ne + 1));
  hipLaunchKernelGGL(k_amg_ext_len, dim3((ne + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, n, nrecv, (const rp_t *)P.rp.p,
                     (const double *)rbuf.p, K, len.p);
  Pext.n = ne; Pext.m = nagg + ncg;
  ISPH_CHECK(Pext.rp.reserve((size_t)ne + 1));
  ISPH_CHECK(amg_scan(ctx, (const int *)len.p, Pext.rp.p, ne + 1, tmp));
  long long nnz = 0;
  ISPH_CHECK(amg_read_off(ctx, Pext.rp.p + ne, &nnz));
  Pext.nnz = nnz;
  ISPH_CHECK(Pext.ci.reserve((size_t)(nnz > 0 ? nnz : 1)));
  ISPH_CHECK(Pext.v.reserve((size_t)(nnz > 0 ? nnz : 1)));
  hipLaunchKernelGGL(k_amg_ext_fill, dim3((ne + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, n, nrecv, (const rp_t *)P.rp.p,
                     (const int *)P.ci.p, (const double *)P.v.p, (const double *)rbuf.p, K, (const int *)slot0.p, (const rp_t *)Pext.rp.p,
                     Pext.ci.p, Pext.v.p);
  ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));   // hslot and the temporaries go out of scope
  ISPH_CHECK_HIP(hipGetLastError());
  return ISPH_SUCCESS;
}

// the coarsest operator of all ranks as one dense [A | I] on every rank: own rows written, the rest summed in
__global__ void k_dense_from_csr_glob(int nc, int N, int off, const rp_t *__restrict__ rp, const int *__restrict__ ci,
                                      const double *__restrict__ v, const double *__restrict__ ghost_gid, double *__restrict__ aug) {
  const int i = blockIdx.x;
  double *row = aug + (size_t)(off + i) * 2 * N;
  if (threadIdx.x == 0) row[N + off + i] = 1.0;
  for (rp_t p = rp[i] + threadIdx.x; p < rp[i + 1]; p += blockDim.x) {
    const int c = ci[p];
    row[c < nc ? off + c : (int)ghost_gid[c - nc]] = v[p];
  }
}
__global__ void k_amg_gid(int n, int off, const int *__restrict__ idx, double *__restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[k] = (double)(off + idx[k]);
}
__global__ __launch_bounds__(256) void k_dense_apply_rows(int nloc, int N, int off, const double *__restrict__ aug,
                                                          const double *__restrict__ b, double *__restrict__ x) {
  const int i = blockIdx.x * kAmgWaves + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= nloc) return;
  double s = 0.0;
  for (int c = lane; c < N; c += 64) s += aug[(size_t)(off + i) * 2 * N + c] * b[c];   // the inverse: left half (k_gj_finish)
  s = wave_sum(s);
  if (lane == 0) x[i] = s;
}

// A rank that cannot coarsen any further while others still can (few rows, an empty rank) passes its level on unchanged:
// every row its own aggregate, P = I -- so that it still takes part in the exchanges of the level and its neighbours'
// coarse rows find its unknowns.
__global__ void k_amg_identity_p(int n, int *__restrict__ agg, rp_t *__restrict__ prp, int *__restrict__ pci, double *__restrict__ pv,
                                 const double *__restrict__ nv, double *__restrict__ nvc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { agg[i] = i; prp[i] = i; pci[i] = i; pv[i] = 1.0; nvc[i] = nv[i]; }
  if (i == n) prp[n] = n;
}
inline int amg_identity_prolongator(isph_ctx *ctx, AmgLevel *L, DevBuf<double> &nvc) {
  const int n = L->A.n;
  DCsr &P = L->P;
  P.n = n; P.m = n; P.nnz = n;
  ISPH_CHECK(L->agg.reserve((size_t)std::max(n, 1)));
  ISPH_CHECK(P.rp.reserve((size_t)n + 1));
  ISPH_CHECK(P.ci.reserve((size_t)std::max(n, 1)));
  ISPH_CHECK(P.v.reserve((size_t)std::max(n, 1)));
  ISPH_CHECK(nvc.reserve((size_t)std::max(n, 1)));
  hipLaunchKernelGGL(k_amg_identity_p, dim3((n + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, n, L->agg.p, P.rp.p, P.ci.p, P.v.p,
                     (const double *)L->nv.p, nvc.p);
  ISPH_CHECK_HIP(hipGetLastError());
  return ISPH_SUCCESS;
}

// aug = [A | I] (n x 2n, row-major) -> its left half holds A^-1; err bit 4: singular
inline int amg_dense_invert(isph_ctx *ctx, double *aug, int n, int *derr) {
  if (n <= 0) return ISPH_SUCCESS;
  DevTmp<double> col;
  DevTmp<int> idx;
  ISPH_CHECK(col.reserve((size_t)3 * n));
  ISPH_CHECK(idx.reserve((size_t)2 * n));
  double *c0 = col.p, *c1 = col.p + n, *pv = col.p + 2 * (size_t)n;
  int *usedat = idx.p, *pivrow = idx.p + n;
  hipLaunchKernelGGL(k_gj_init, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, n, (const double *)aug, c0, usedat);
  for (int k = 0; k < n; ++k)
    hipLaunchKernelGGL(k_gj_step, dim3(n), dim3(256), 0, ctx->stream, n, k, aug, (const double *)((k & 1) ? c1 : c0), (k & 1) ? c0 : c1,
                       usedat, pivrow, derr);
  hipLaunchKernelGGL(k_gj_pivots, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, n, (const double *)aug, (const int *)pivrow, pv);
  hipLaunchKernelGGL(k_gj_finish, dim3(n), dim3(256), 0, ctx->stream, n, aug, (const int *)pivrow, (const double *)pv);
  ISPH_CHECK_HIP(hipGetLastError());
  return ISPH_SUCCESS;   // (col / idx go back to the pool, which hands them out again only after a device synchronisation)
}

inline int amg_level_buffers(AmgLevel *L) {
  const size_t m = (size_t)(L->A.n > 0 ? L->A.n : 1) + 64;
  ISPH_CHECK(L->x.reserve(m));
  ISPH_CHECK(L->b.reserve(m));
  ISPH_CHECK(L->r.reserve(m));
  ISPH_CHECK(L->z.reserve(m));
  return ISPH_SUCCESS;
}

inline int amg_create(isph_ctx *ctx, const isph_mat *Am, const isph_amg_params *prm, const double *nullvec_dev,
                      isph_amg **out) {
  ISPH_REQUIRE(prm->max_levels >= 1 && prm->max_levels <= 8, "max_levels must be in [1,8]");
  ISPH_REQUIRE(prm->block >= 64 && prm->block <= 1024 && prm->block % 64 == 0, "smoother block must be a multiple of 64 in [64,1024]");
  ISPH_REQUIRE(prm->sweeps >= 1, "smoother sweeps must be >= 1");
  isph_amg *G = new isph_amg();
  G->block = prm->block; G->sweeps = prm->sweeps; G->singular = nullvec_dev != nullptr;
  ISPH_REQUIRE(prm->smoother == 0 || prm->smoother == 1, "smoother must be 0 (symmetric Gauss-Seidel) or 1 (Gauss-Seidel, efficient symmetric)");
  G->gs_eff = prm->smoother == 1;
  DevTmp<char> tmp;
  DevTmp<int> derr;
  DevTmp<double> dg;
  DevTmp<unsigned long long> rho;
  int rc = derr.reserve(1);
  if (rc == ISPH_SUCCESS) rc = rho.reserve(1);
  if (rc == ISPH_SUCCESS && hipMemsetAsync(derr.p, 0, sizeof(int), ctx->stream) != hipSuccess) rc = fail("memset failed", __FILE__, __LINE__);
  AmgLevel *L0 = new AmgLevel();
  G->L.push_back(L0);
  G->nlev = 1;
  L0->Am = Am;
  const int n0 = Am->S.nrow;
  // threshold 0: the SELL -> CSR sweep of the fine level prepares the aggregation on the way (k_sell_to_csr_i32<true>)
  DevTmp<int> sc0;
  DevTmp<unsigned long long> key0;
  bool prep0 = prm->theta == 0.0 && n0 > 0 && prm->max_levels > 1 && n0 > prm->coarse_max && !(getenv("ISPH_AMG_NO_FUSED_PREP") != nullptr);
  if (rc == ISPH_SUCCESS && prep0) {
    rc = sc0.reserve((size_t)(Am->S.nnz > 0 ? Am->S.nnz : 1));
    if (rc == ISPH_SUCCESS) rc = key0.reserve((size_t)n0);
    if (rc == ISPH_SUCCESS) rc = dg.reserve((size_t)n0);
  }
  if (rc == ISPH_SUCCESS) rc = prep0 ? amg_csr_from_sell(ctx, Am->S, L0->A, tmp, dg.p, sc0.p, key0.p, rho.p) : amg_csr_from_sell(ctx, Am->S, L0->A, tmp);
  if (rc == ISPH_SUCCESS) rc = L0->nv.reserve((size_t)(n0 > 0 ? n0 : 1));
  if (rc == ISPH_SUCCESS) {
    if (nullvec_dev) {
      if (hipMemcpyAsync(L0->nv.p, nullvec_dev, sizeof(double) * (size_t)n0, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
        rc = fail("copy failed", __FILE__, __LINE__);
    } else if (n0 > 0) {
      hipLaunchKernelGGL(k_fill, dim3(stream_grid(n0)), dim3(kBlock), 0, ctx->stream, n0, L0->nv.p, 1.0);
    }
  }
  // more than one rank: every decision about the depth of the hierarchy is taken by all ranks together (the set-up and
  // the cycle exchange with the neighbours on every level), a failure on one rank is a failure on all
  // (ISPH_AMG_RANK_LOCAL=1: the hierarchy of rounds 1-3, P^T A P with the rank's own columns only -- kept for comparisons)
  const char *env_local = getenv("ISPH_AMG_RANK_LOCAL");
  const bool dist = comm_active(ctx) && ctx->nranks > 1 && !(env_local && env_local[0] == '1');
  G->dist = dist ? 1 : 0;
  while ((dist || rc == ISPH_SUCCESS) && G->nlev < prm->max_levels) {
    AmgLevel *L = G->L.back();
    const int n = L->A.n;
    if (dist) {   // go on while any rank is above coarse_max (and none has failed)
      double h[2] = {rc == ISPH_SUCCESS && n > prm->coarse_max ? 1.0 : 0.0, rc == ISPH_SUCCESS ? 0.0 : 1.0};
      if (amg_host_allreduce(ctx, h, 2, 1) != ISPH_SUCCESS) { rc = fail("AMG: consensus between the ranks failed", __FILE__, __LINE__); break; }
      if (h[1] != 0.0) { if (rc == ISPH_SUCCESS) rc = fail("AMG: set-up failed on another rank", __FILE__, __LINE__); break; }
      if (h[0] == 0.0) break;
    } else if (n <= prm->coarse_max) break;
    rc = dg.reserve((size_t)(n > 0 ? n : 1));
    if (rc != ISPH_SUCCESS && !dist) break;
    int nagg = 0;
    if (rc == ISPH_SUCCESS && n > 0) {
      if (prm->theta != 0.0)
        hipLaunchKernelGGL(k_amg_diag, dim3(amg_wave_grid(n)), dim3(256), 0, ctx->stream, n, (const rp_t *)L->A.rp.p,
                           (const int *)L->A.ci.p, (const double *)L->A.v.p, dg.p);
      const bool use_prep = prep0 && L == L0;
      rc = amg_aggregate(ctx, L, dg.p, rho.p, prm->theta, tmp, &nagg, use_prep ? sc0.p : nullptr, use_prep ? key0.p : nullptr);
      if (use_prep) { sc0.release(); key0.release(); }
    }
    if (rc != ISPH_SUCCESS && !dist) break;
    // no coarsening, or a coarse space too small to carry anything but the null vector: stop here
    bool stop = rc != ISPH_SUCCESS || nagg < 8 || nagg >= n;
    bool identity = false;
    if (dist) {   // ... when no rank can go on; a rank that cannot while others can passes its level on unchanged (P = I)
      double h[2] = {stop ? 0.0 : 1.0, rc == ISPH_SUCCESS ? 0.0 : 1.0};
      if (amg_host_allreduce(ctx, h, 2, 1) != ISPH_SUCCESS) { rc = fail("AMG: consensus between the ranks failed", __FILE__, __LINE__); break; }
      if (h[1] != 0.0 && rc == ISPH_SUCCESS) rc = fail("AMG: set-up failed on another rank", __FILE__, __LINE__);
      identity = stop && h[0] != 0.0 && rc == ISPH_SUCCESS;
      stop = h[0] == 0.0 || rc != ISPH_SUCCESS;
    }
    if (stop) { L->agg.release(); break; }
    AmgLevel *Lc = new AmgLevel();
    if (identity) { nagg = n; rc = amg_identity_prolongator(ctx, L, Lc->nv); }
    else rc = amg_prolongator(ctx, L, dg.p, rho.p, prm->theta == 0.0, nagg, prm->omega, Lc->nv, tmp, derr.p);
    DCsr AP, Pext;
    DCsr &R = L->R;
    const isph_halo &H = L->Am->halo;
    std::vector<int> cs_ptr, cs_idx, cr_ptr;
    bool extended = false;
    if (dist) {
      // the P rows behind A's ghost columns (amg_extend_pack / amg_extend_finish); K is agreed first, by all ranks
      DevTmp<int> kmax;
      int hk = 0;
      if (rc == ISPH_SUCCESS) rc = kmax.reserve(1);
      if (rc == ISPH_SUCCESS && hipMemsetAsync(kmax.p, 0, sizeof(int), ctx->stream) != hipSuccess) rc = fail("memset failed", __FILE__, __LINE__);
      if (rc == ISPH_SUCCESS && H.nsend > 0)
        hipLaunchKernelGGL(k_amg_send_rowmax, dim3((H.nsend + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, H.nsend,
                           (const int *)H.send_idx.p, (const rp_t *)L->P.rp.p, kmax.p);
      if (rc == ISPH_SUCCESS) rc = amg_read_int(ctx, kmax.p, &hk);
      double h[2] = {(double)hk, rc == ISPH_SUCCESS ? 0.0 : 1.0};
      if (amg_host_allreduce(ctx, h, 2, 1) != ISPH_SUCCESS) { rc = fail("AMG: consensus between the ranks failed", __FILE__, __LINE__); amg_level_destroy(Lc); break; }
      if (h[1] != 0.0) { if (rc == ISPH_SUCCESS) rc = fail("AMG: set-up failed on another rank", __FILE__, __LINE__); amg_level_destroy(Lc); break; }
      AmgExtendState est;
      const int Kx = std::max((int)h[0], 1);
      // (a rank without ghost columns of its own still owes its neighbours their rows)
      if (H.npeers > 0) rc = amg_extend_pack(ctx, H, L->P, Kx, est, cs_ptr, cs_idx, tmp);
      double h2 = rc == ISPH_SUCCESS ? 0.0 : 1.0;   // nobody enters the exchange unless everybody can
      if (amg_host_allreduce(ctx, &h2, 1, 1) != ISPH_SUCCESS) { rc = fail("AMG: consensus between the ranks failed", __FILE__, __LINE__); amg_level_destroy(Lc); break; }
      if (h2 != 0.0) { if (rc == ISPH_SUCCESS) rc = fail("AMG: set-up failed on another rank", __FILE__, __LINE__); amg_level_destroy(Lc); break; }
      if (H.npeers > 0) {
        rc = amg_extend_finish(ctx, H, L->P, Kx, est, Pext, cr_ptr, tmp);
        extended = rc == ISPH_SUCCESS;
      }
    }
    if (rc == ISPH_SUCCESS) rc = amg_transpose(ctx, L->P, R, tmp);
    if (rc == ISPH_SUCCESS) rc = amg_spgemm_ap(ctx, L->A, extended ? Pext : L->P, AP, tmp, derr.p);
    if (rc == ISPH_SUCCESS) rc = amg_spgemm_t<4096, 256>(ctx, R, AP, Lc->A, tmp, derr.p);
    Pext.release();
    int herr = 0;
    if (rc == ISPH_SUCCESS) rc = amg_read_int(ctx, derr.p, &herr);
    if (rc == ISPH_SUCCESS && (herr & 1)) rc = fail("AMG: a row touches more than 512 aggregates", __FILE__, __LINE__);
    if (rc == ISPH_SUCCESS && (herr & 2)) rc = fail("AMG: coarse operator row too dense for the SpGEMM table", __FILE__, __LINE__);
    if (rc == ISPH_SUCCESS) rc = mat_from_device_csr(ctx, L->P.n, L->P.m, L->P.rp.p, L->P.ci.p, L->P.v.p, L->P.nnz, &L->Pm, /*rows_sorted=*/true);
    if (rc == ISPH_SUCCESS) rc = mat_from_device_csr(ctx, Lc->A.n, Lc->A.m, Lc->A.rp.p, Lc->A.ci.p, Lc->A.v.p, Lc->A.nnz, &Lc->Aown);
    // without ghost columns A (x + P e) = A x + (A P) e: the cycle updates the residual with the product already at hand
    // (a fifth of A's entries) instead of a second sweep over A
    // (not on a rank that sends to neighbours: the product it saves is also this rank's part of their exchange)
    if (rc == ISPH_SUCCESS && ((L->Am->S.ncol == L->Am->S.nrow && L->Am->halo.nsend == 0) || L->Am->local)) {
      rc = mat_from_device_csr(ctx, AP.n, AP.m, AP.rp.p, AP.ci.p, AP.v.p, AP.nnz, &L->APm, /*rows_sorted=*/true);
      if (rc == ISPH_SUCCESS) L->APm->local = true;
    }
    AP.release();
    if (rc == ISPH_SUCCESS && extended) {   // the coarse operator exchanges with the same peers (lists: amg_extend_pack / amg_extend_finish)
      rc = isph_mat_set_halo(ctx, Lc->Aown, H.npeers, H.peer.data(), cs_ptr.data(), cs_idx.empty() ? nullptr : cs_idx.data(), cr_ptr.data());
      if (rc == ISPH_SUCCESS) Lc->Aown->aux = true;
    }
    if (rc != ISPH_SUCCESS) { amg_level_destroy(Lc); if (dist) continue; break; }   // dist: the next consensus ends the loop on all ranks
    L->Pm->local = true;
    if (!extended) Lc->Aown->local = true;
    Lc->Am = Lc->Aown;
    G->L.push_back(Lc);
    ++G->nlev;
  }
  // the fine-level CSR copy only serves the set-up (the cycle runs on the SELL matrix): 12 B per entry go back to the
  // pool before the smoother is built; isph_prec_amg_export rebuilds it on demand
  if (G->nlev > 1) { L0->A.ci.release(); L0->A.v.release(); L0->A.rp.release(); }
  // smoothers, work vectors, coarse solve
  G->coarse_smooth = G->singular || G->L.back()->A.n > kAmgDenseMax;
  int nc_off = 0, nc_glob = G->L.back()->A.n;
  if (dist) {
    // the coarsest operators of all ranks form ONE system (ML gathers it for its direct solver): every rank inverts it
    std::vector<double> h((size_t)ctx->nranks + 1, 0.0);
    h[(size_t)ctx->rank] = (double)G->L.back()->A.n;
    h[(size_t)ctx->nranks] = rc == ISPH_SUCCESS ? 0.0 : 1.0;
    if (amg_host_allreduce(ctx, h.data(), ctx->nranks + 1, 0) != ISPH_SUCCESS) rc = fail("AMG: consensus between the ranks failed", __FILE__, __LINE__);
    else if (h[(size_t)ctx->nranks] != 0.0 && rc == ISPH_SUCCESS) rc = fail("AMG: set-up failed on another rank", __FILE__, __LINE__);
    double tot = 0.0;
    for (int r = 0; r < ctx->nranks; ++r) { if (r == ctx->rank) nc_off = (int)tot; tot += h[(size_t)r]; }
    nc_glob = tot > 2.0e9 ? 2000000000 : (int)tot;
    G->coarse_smooth = G->singular || nc_glob > kAmgDenseMax;
  }
  for (int l = 0; l < G->nlev && rc == ISPH_SUCCESS; ++l) {
    AmgLevel *L = G->L[(size_t)l];
    rc = amg_level_buffers(L);
    const bool last = l == G->nlev - 1;
    // coarse levels are small: 64-row blocks keep enough waves busy (an 8-block level ran its sweeps on 8 waves)
    if (rc == ISPH_SUCCESS && (!last || G->coarse_smooth)) {
      const char *env_stream = getenv("ISPH_AMG_COARSE_STREAM");   // the chunk-stream sweeps on every level (rounds 2-4)
      if (l > 0 && L->A.n > 0 && L->A.n <= kSgsDenseMaxRows && kAmgCoarseBlock == 64 && !(env_stream && env_stream[0] == '1')) {
        const int nb = (L->A.n + 63) / 64;
        // symmetric sweeps: always on a coarsest level the smoother solves, otherwise unless the cycle uses the one-directional sweeps
        const bool need_sym = last || !G->gs_eff, need_dir = G->gs_eff && !last;
        auto build = [&](DevBuf<double> &W, int mode) {
          int r2 = W.reserve((size_t)nb * 4096);
          if (r2 == ISPH_SUCCESS)
            hipLaunchKernelGGL(k_sgs_dense_build, dim3(nb), dim3(64), 0, ctx->stream, L->A.n, (const rp_t *)L->A.rp.p,
                               (const int *)L->A.ci.p, (const double *)L->A.v.p, W.p, mode);
          return r2;
        };
        if (need_sym) rc = build(L->wsgs, 0);
        if (rc == ISPH_SUCCESS && need_dir) rc = build(L->wfwd, 1);
        if (rc == ISPH_SUCCESS && need_dir) rc = build(L->wbwd, 2);
      } else {
        rc = ilu_create(ctx, L->Am, l == 0 ? G->block : kAmgCoarseBlock, &L->sgs, /*sgs=*/true);
      }
    }
  }
  if (dist) {   // the smoothers are built per rank: a failure there must keep every rank out of the collective steps below
    double h = rc == ISPH_SUCCESS ? 0.0 : 1.0;
    if (amg_host_allreduce(ctx, &h, 1, 1) != ISPH_SUCCESS) rc = fail("AMG: consensus between the ranks failed", __FILE__, __LINE__);
    else if (h != 0.0 && rc == ISPH_SUCCESS) rc = fail("AMG: set-up failed on another rank", __FILE__, __LINE__);
  }
  if (rc == ISPH_SUCCESS && !G->coarse_smooth && dist && G->nlev > 1) {
    AmgLevel *L = G->L.back();
    const int ncl = L->A.n, N = nc_glob;
    G->nc = N; G->nc_off = nc_off; G->nc_loc = ncl;
    const isph_halo &Hc = L->Am->halo;
    DevTmp<double> gs, gr;
    rc = G->cinv.reserve((size_t)2 * N * N + (size_t)N + 1);
    if (rc == ISPH_SUCCESS) rc = G->bglob.reserve((size_t)std::max(N, 1));
    if (rc == ISPH_SUCCESS) rc = gs.reserve((size_t)std::max(Hc.nsend, 1));
    if (rc == ISPH_SUCCESS) rc = gr.reserve((size_t)std::max(Hc.nrecv, 1));
    if (rc == ISPH_SUCCESS && hipMemsetAsync(G->cinv.p, 0, sizeof(double) * (size_t)2 * N * N, ctx->stream) != hipSuccess) rc = fail("memset failed", __FILE__, __LINE__);
    if (rc == ISPH_SUCCESS && Hc.nsend > 0)
      hipLaunchKernelGGL(k_amg_gid, dim3((Hc.nsend + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, Hc.nsend, nc_off, (const int *)Hc.send_idx.p, gs.p);
    if (rc == ISPH_SUCCESS) rc = comm_exchange(ctx, Hc, gs.p, gr.p, 1, false, ctx->stream);   // global numbers of the ghost columns
    if (rc == ISPH_SUCCESS && ncl > 0)
      hipLaunchKernelGGL(k_dense_from_csr_glob, dim3(ncl), dim3(kBlock), 0, ctx->stream, ncl, N, nc_off, (const rp_t *)L->A.rp.p,
                         (const int *)L->A.ci.p, (const double *)L->A.v.p, (const double *)gr.p, G->cinv.p);
    if (rc == ISPH_SUCCESS) rc = comm_allreduce(ctx, G->cinv.p, 2 * N * N, 0, ctx->stream);
    if (rc == ISPH_SUCCESS && N > 0) {
      rc = amg_dense_invert(ctx, G->cinv.p, N, derr.p);
      int herr = 0;
      if (rc == ISPH_SUCCESS) rc = amg_read_int(ctx, derr.p, &herr);
      if (rc == ISPH_SUCCESS && (herr & 4)) rc = fail("AMG: coarsest operator is singular (pass the null vector)", __FILE__, __LINE__);
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == ISPH_SUCCESS) rc = fail("AMG setup failed", __FILE__, __LINE__);   // gs, gr leave scope
  } else if (rc == ISPH_SUCCESS && !G->coarse_smooth) {
    AmgLevel *L = G->L.back();
    const int nc = L->A.n;
    G->nc = nc;
    if (rc == ISPH_SUCCESS) rc = G->cinv.reserve((size_t)2 * nc * nc + (size_t)nc + 1);
    if (rc == ISPH_SUCCESS && nc > 0) {
      hipLaunchKernelGGL(k_dense_from_csr, dim3(nc), dim3(kBlock), 0, ctx->stream, nc, (const rp_t *)L->A.rp.p,
                         (const int *)L->A.ci.p, (const double *)L->A.v.p, G->cinv.p);
      rc = amg_dense_invert(ctx, G->cinv.p, nc, derr.p);
      int herr = 0;
      if (rc == ISPH_SUCCESS) rc = amg_read_int(ctx, derr.p, &herr);
      if (rc == ISPH_SUCCESS && (herr & 4)) rc = fail("AMG: coarsest operator is singular (pass the null vector)", __FILE__, __LINE__);
    }
  }
  if (rc == ISPH_SUCCESS && (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess))
    rc = fail("AMG setup failed", __FILE__, __LINE__);
  tmp.release(); derr.release(); dg.release();
  if (rc != ISPH_SUCCESS) { amg_destroy(G); return rc; }
  *out = G;
  return ISPH_SUCCESS;
}

// z = M_B^-1 r of level L (accumulate: z += M_B^-1 r).  part: 0 = the symmetric sweep, 1 = forward alone, 2 = backward alone
inline int amg_sgs_apply(isph_ctx *ctx, AmgLevel *L, const double *r, double *z, bool accumulate, int part = 0) {
  const int n = L->A.n;
  const double *W = part == 0 ? L->wsgs.p : (part == 1 ? L->wfwd.p : L->wbwd.p);
  if (W) {
    const int grid = ((n + 63) / 64 + 3) / 4;
    if (n <= 0) return ISPH_SUCCESS;
    if (accumulate) hipLaunchKernelGGL((k_sgs_dense_apply<true>), dim3(grid), dim3(256), 0, ctx->stream, n, W, r, z);
    else hipLaunchKernelGGL((k_sgs_dense_apply<false>), dim3(grid), dim3(256), 0, ctx->stream, n, W, r, z);
    return ISPH_SUCCESS;
  }
  ISPH_REQUIRE(L->sgs != nullptr, "AMG level without a smoother");
  return ilu_apply(ctx, L->sgs, r, z, part, accumulate);   // (accumulate: the stream kernel adds into z on its way out)
}

// x += M_B^-1 (b - A x); zero_guess: x = M_B^-1 b
inline int amg_smooth(isph_ctx *ctx, const isph_amg *G, int l, const double *b, double *x, bool zero_guess, int part = 0) {
  AmgLevel *L = G->L[(size_t)l];
  if (zero_guess) return amg_sgs_apply(ctx, L, b, x, false, part);
  ISPH_CHECK(spmv_dev(ctx, L->Am, x, L->r.p, nullptr, b, -1.0));   // r = b - A x
  return amg_sgs_apply(ctx, L, L->r.p, x, true, part);
}

inline int amg_vcycle(isph_ctx *ctx, const isph_amg *G, int l, const double *b, double *x) {
  AmgLevel *L = G->L[(size_t)l];
  const int n = L->A.n;
  if (l == G->nlev - 1) {
    if (G->coarse_smooth) {
      ISPH_CHECK(amg_smooth(ctx, G, l, b, x, true));
      for (int s = 1; s < G->sweeps; ++s) ISPH_CHECK(amg_smooth(ctx, G, l, b, x, false));
    } else if (G->dist && G->nlev > 1) {
      // the right-hand side of all ranks, then this rank's rows of the inverse
      const int N = G->nc;
      if (N > 0) {
        ISPH_CHECK_HIP(hipMemsetAsync(G->bglob.p, 0, sizeof(double) * (size_t)N, ctx->stream));
        if (n > 0) ISPH_CHECK_HIP(hipMemcpyAsync(G->bglob.p + G->nc_off, b, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
        ISPH_CHECK(comm_allreduce(ctx, G->bglob.p, N, 0, ctx->stream));
        if (n > 0)
          hipLaunchKernelGGL(k_dense_apply_rows, dim3(amg_wave_grid(n)), dim3(256), 0, ctx->stream, n, N, G->nc_off, (const double *)G->cinv.p,
                             (const double *)G->bglob.p, x);
      }
    } else if (n > 0) {
      hipLaunchKernelGGL(k_dense_apply, dim3(amg_wave_grid(n)), dim3(256), 0, ctx->stream, n, (const double *)G->cinv.p, b, x);
    }
    return ISPH_SUCCESS;
  }
  AmgLevel *Lc = G->L[(size_t)l + 1];
  const int pre = G->gs_eff ? 1 : 0, post = G->gs_eff ? 2 : 0;
  ISPH_CHECK(amg_smooth(ctx, G, l, b, x, true, pre));
  for (int s = 1; s < G->sweeps; ++s) ISPH_CHECK(amg_smooth(ctx, G, l, b, x, false, pre));
  ISPH_CHECK(spmv_dev(ctx, L->Am, x, L->r.p, nullptr, b, -1.0));   // r = b - A x
  if (L->R.n > 0)   // (a rank without rows still walks the cycle: its neighbours' exchanges and the all-reduces count on it)
    hipLaunchKernelGGL(k_csr_spmv_wave, dim3(amg_wave_grid(L->R.n)), dim3(256), 0, ctx->stream, L->R.n, (const rp_t *)L->R.rp.p,
                       (const int *)L->R.ci.p, (const double *)L->R.v.p, (const double *)L->r.p, Lc->b.p);
  ISPH_CHECK(amg_vcycle(ctx, G, l + 1, Lc->b.p, Lc->x.p));
  ISPH_CHECK(spmv_dev(ctx, L->Pm, Lc->x.p, x, nullptr, x, 1.0));   // x += P e
  int first = 0;
  if (L->APm) {
    // r still holds b - A x of before the correction: r -= (A P) e, then the first post-smoothing sweep uses it
    ISPH_CHECK(spmv_dev(ctx, L->APm, Lc->x.p, L->r.p, nullptr, L->r.p, -1.0));
    ISPH_CHECK(amg_sgs_apply(ctx, L, L->r.p, x, true, post));
    first = 1;
  }
  for (int s = first; s < G->sweeps; ++s) ISPH_CHECK(amg_smooth(ctx, G, l, b, x, false, post));
  return ISPH_SUCCESS;
}

inline int amg_apply(isph_ctx *ctx, const isph_amg *G, const double *r, double *z) {
  ISPH_REQUIRE(G != nullptr, "AMG hierarchy is NULL");
  ISPH_CHECK(amg_vcycle(ctx, G, 0, r, z));
  ISPH_CHECK_HIP(hipGetLastError());
  return ISPH_SUCCESS;
}

}  // namespace isph
