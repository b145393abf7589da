// ingress.hpp -- host CSR (the three arrays of Epetra_CrsMatrix::ExtractCrsDataPointers, pageable memory owned by the
// caller) -> sliced-ELL on the device, as a pipeline.  This is the drop-in path of SolverLin_HIP::solveProblem
// (ref: solver_lin_belos.h:130-222 receives the matrix PairISPH filled on the host, pair_isph.cpp:924-926,988-1011):
// everything here is inside what the reference times as "ISPH: solvePoisson", so the 12 B per stored entry have to
// cross PCIe at the link's rate and nothing else may be on the critical path.
//
//   worker threads : pageable -> pinned ring slot: values by memcpy, columns as 16-bit differences to the previous
//                    column of the row where the rows allow it (10 instead of 12 bytes per entry on the link; the
//                    range check rides along in the same pass), the first column of every row in a table behind them
//   main thread    : slot -> device image, ONE hipMemcpyAsync per chunk on a copy stream, event per slot;
//                    compute stream waits on the event and converts the slices whose rows have fully arrived
//                    (k_csr_to_sell over a slice range), so only the last chunk's conversion is exposed;
//   hooks          : work that only needs the rows that are already there is queued behind the conversion --
//                    the 16-bit column windows of the SpMV and the block-Jacobi ILU(0) set-up (extract, schedule,
//                    factorisation per range of 512-row blocks) run while the rest of the matrix is still on the link
//                    (csr_ingress_host_bjacobi = isph_mat_create_csr_bjacobi).
//
// Slice widths/offsets come from the host row pointers (no device round trip); rows that are already column-sorted
// (Epetra's OptimizeStorage order) are detected on the device while they are converted and skip the row sort.
// Measured on the MI355X box (scripts/pcie_probe.hip, profiles/r03_pcie_probe.txt): pinned H2D 57.6 GB/s with >= 12 MiB
// per copy (41 GB/s with 4 MiB copies), one memcpy thread moves 34 GB/s pageable -> pinned.
#pragma once
#include <atomic>
#include <chrono>
#include <cstdint>
#include <condition_variable>
#include <functional>
#include <thread>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "core.hpp"
#include "ilu.hpp"

namespace isph {

// Staging copies with non-temporal stores: the pinned ring is written by the staging threads and read by the DMA engine
// only.  With ordinary stores the freshly written lines sit dirty in the cores' caches and every DMA read has to be served
// from there: the more staging threads, the slower the link ran (copies of the 100^3 matrix done after 24.4 ms with 4
// threads, 26.5 with 8, 34 with 12).  Streaming stores put the data in memory behind the write-combining buffers.
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline void stage_copy_avx2(void *dst, const void *src, size_t bytes) {
  char *d = static_cast<char *>(dst);
  const char *s = static_cast<const char *>(src);
  size_t k = 0;
  while (k < bytes && (reinterpret_cast<uintptr_t>(d + k) & 31)) { d[k] = s[k]; ++k; }
  for (; k + 128 <= bytes; k += 128) {
    const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + k));
    const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + k + 32));
    const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + k + 64));
    const __m256i e = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + k + 96));
    _mm256_stream_si256(reinterpret_cast<__m256i *>(d + k), a);
    _mm256_stream_si256(reinterpret_cast<__m256i *>(d + k + 32), b);
    _mm256_stream_si256(reinterpret_cast<__m256i *>(d + k + 64), c);
    _mm256_stream_si256(reinterpret_cast<__m256i *>(d + k + 96), e);
  }
  for (; k < bytes; ++k) d[k] = s[k];
  _mm_sfence();
}
// copy of n column indices with the range check [0, ncol) riding along; returns nonzero when one is outside
__attribute__((target("avx2"))) inline unsigned stage_cols_avx2(int *dst, const int *src, size_t n, unsigned ncol) {
  unsigned over = 0;
  size_t k = 0;
  while (k < n && (reinterpret_cast<uintptr_t>(dst + k) & 31)) { dst[k] = src[k]; over |= (unsigned)((unsigned)src[k] >= ncol); ++k; }
  const __m256i sign = _mm256_set1_epi32((int)0x80000000u);
  const __m256i lim = _mm256_set1_epi32((int)((ncol - 1u) ^ 0x80000000u));  // unsigned x > ncol - 1  <=>  (x ^ sign) >s (ncol - 1) ^ sign
  __m256i bad = _mm256_setzero_si256();
  for (; k + 16 <= n; k += 16) {
    const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + k));
    const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + k + 8));
    _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + k), a);
    _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + k + 8), b);
    bad = _mm256_or_si256(bad, _mm256_or_si256(_mm256_cmpgt_epi32(_mm256_xor_si256(a, sign), lim), _mm256_cmpgt_epi32(_mm256_xor_si256(b, sign), lim)));
  }
  if (!_mm256_testz_si256(bad, bad)) over = 1;
  for (; k < n; ++k) { dst[k] = src[k]; over |= (unsigned)((unsigned)src[k] >= ncol); }
  _mm_sfence();
  return over;
}
inline bool stage_have_avx2() { static const bool v = __builtin_cpu_supports("avx2"); return v; }
#else
inline bool stage_have_avx2() { return false; }
inline void stage_copy_avx2(void *, const void *, size_t) {}
inline unsigned stage_cols_avx2(int *, const int *, size_t, unsigned) { return 0; }
#endif
inline void stage_copy(void *dst, const void *src, size_t bytes) {
  if (stage_have_avx2()) stage_copy_avx2(dst, src, bytes); else memcpy(dst, src, bytes);
}
// A row segment of the gathered ingress: a few hundred bytes, both pointers 4-byte aligned (8 for values).  Plain moves
// up to the next 32-byte boundary of the destination, streaming stores of 32 bytes, plain moves for the rest; no fence
// (the chunk's finisher fences once).  stage_copy's byte-wise head and tail (up to 158 bytes per call) cost more than
// the copy itself at this size.
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline void stage_copy_row_avx2(void *dst, const void *src, size_t bytes) {
  char *d = static_cast<char *>(dst);
  const char *s = static_cast<const char *>(src);
  size_t k = 0;
  while (k + 4 <= bytes && (reinterpret_cast<uintptr_t>(d + k) & 31)) { memcpy(d + k, s + k, 4); k += 4; }
  for (; k + 32 <= bytes; k += 32)
    _mm256_stream_si256(reinterpret_cast<__m256i *>(d + k), _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + k)));
  for (; k + 4 <= bytes; k += 4) memcpy(d + k, s + k, 4);
  for (; k < bytes; ++k) d[k] = s[k];
}
#endif
inline void stage_copy_row(void *dst, const void *src, size_t bytes) {
#if defined(__x86_64__)
  if (stage_have_avx2()) { stage_copy_row_avx2(dst, src, bytes); return; }
#endif
  memcpy(dst, src, bytes);
}
inline void stage_fence() {
#if defined(__x86_64__)
  _mm_sfence();
#endif
}

struct HostStager {
  static constexpr size_t kChunk = (size_t)4 << 20;  // entries per ring slot; a slot is 12 bytes per entry = 48 MiB
  int nslots = 0, nthreads = 0;
  char *pslot = nullptr;    // pinned [nslots][12 * kChunk]: one chunk as it crosses the link (layout: sell.hpp CsrChunks)
  hipStream_t copy_stream = nullptr;
  static constexpr int kAux = 2;
  hipStream_t aux[kAux] = {nullptr, nullptr};  // set-up batches of the fused path, round robin
  hipEvent_t ev_conv = nullptr, ev_aux[kAux] = {nullptr, nullptr};
  std::vector<hipEvent_t> ev;  // per slot: its last copy has completed
  DevBuf<int> flag;            // [0] some row was not column-sorted, [1] a slice has more than 64 column windows
  // wall-clock milestones of the last ingress in ms since its start (isph_ingress_info): [0] workers started and device
  // buffers reserved, [1] all chunks queued on the copy stream, [2] copy stream drained, [3] compute stream drained
  // (conversion + hooks), [4] end (incl. row sort), [5] of which the main thread waited for staged chunks, [6] bytes that
  // crossed the link, [7] threads
  double stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long bytes_sent = 0;    // over the link in the last ingress
  ~HostStager() {
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    for (int k = 0; k < kAux; ++k) {
      if (aux[k]) (void)hipStreamDestroy(aux[k]);
      if (ev_aux[k]) (void)hipEventDestroy(ev_aux[k]);
    }
    if (ev_conv) (void)hipEventDestroy(ev_conv);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    if (pslot) (void)hipHostFree(pslot);
    flag.release();
  }
};

inline int stager_get(isph_ctx *ctx, HostStager **out) {
  if (!ctx->stager) {
    HostStager *S = new HostStager();
    const unsigned hc = std::thread::hardware_concurrency();
    // a staging thread moves ~10-15 GB/s of entries into the ring (streaming stores); four to eight of them all end the
    // copies of the 100^3 matrix after 24.3-25.0 ms, twelve and more lose 1-2 ms (profiles/r03_dropin.txt)
    S->nthreads = (int)std::min(6u, std::max(1u, hc / 2));
    S->nslots = 4;  // all threads stage one chunk together: one on the link, one queued, one being staged, one spare
    bool ok = hipHostMalloc((void **)&S->pslot, 12 * HostStager::kChunk * (size_t)S->nslots, hipHostMallocDefault) == hipSuccess &&
              hipStreamCreateWithFlags(&S->copy_stream, hipStreamNonBlocking) == hipSuccess;
    for (int s = 0; ok && s < S->nslots; ++s) {
      hipEvent_t e = nullptr;
      ok = hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
      if (ok) S->ev.push_back(e);
    }
    for (int k = 0; ok && k < HostStager::kAux; ++k)
      ok = hipStreamCreateWithFlags(&S->aux[k], hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&S->ev_aux[k], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&S->ev_conv, hipEventDisableTiming) == hipSuccess;
    ok = ok && S->flag.reserve(2) == ISPH_SUCCESS;
    if (!ok) { delete S; return fail("pinned staging ring for the host CSR ingress could not be created", __FILE__, __LINE__); }
    ctx->stager = S;
  }
  *out = ctx->stager;
  return ISPH_SUCCESS;
}

int sell_sort_rows(isph_ctx *ctx, Sell &S);  // isph_capi.hip

// The host CSR in ANOTHER ROW ORDER than it lies in memory (the library's own numbering for the drop-in with coordinates,
// order.hpp): the `rowptr` csr_ingress_host is given is then the row pointer of the PERMUTED matrix, row r of which is
// row perm[r] of the caller's arrays (src_rowptr = the caller's own row pointers); the staging threads gather the row
// segments (1.2 KB each on the bench matrix) instead of copying one flat range, the columns travel in the caller's
// numbering (16-bit differences still work: they are taken inside a source row) and the conversion kernel renames the
// owned ones (colren, a DEVICE array: the caller's row -> its new number).  The rows arrive unsorted by their new
// columns: the `slices` hook sorts each converted range before anything else looks at it.
struct IngressGather {
  const int *perm = nullptr;        // host [nrow]
  const int *src_rowptr = nullptr;  // host [nrow + 1]
  const int *colren = nullptr;      // device [nren]
  int nren = 0;
};

struct IngressHooks {
  // the matrix has its shape, its buffers and (stream-ordered) its slice offsets and row lengths; no entries yet
  std::function<int(isph_mat *)> begin;
  // the conversion of slices [s0, s1) has been queued on the stream; s1 == nslices on the last call
  std::function<int(isph_mat *, int, int)> slices;
};

// host CSR -> isph_mat.  The caller's arrays are only read while this function runs.  was_unsorted: the rows had to be
// column-sorted after the conversion (what the hooks queued for the unsorted image is then void).
inline int csr_ingress_host(isph_ctx *ctx, int nrow, int ncol, const int *rowptr, const int *colidx, const double *val,
                            isph_mat **Aout, const IngressHooks *hooks = nullptr, bool *was_unsorted = nullptr,
                            const IngressGather *gat = nullptr) {
  ISPH_REQUIRE(!is_device_pointer(rowptr) && !is_device_pointer(colidx) && !is_device_pointer(val),
               "device pointer passed with on_device = 0");
  ISPH_REQUIRE(rowptr[0] == 0 && rowptr[nrow] >= 0, "rowptr must start at 0 (Epetra's ExtractCrsDataPointers does) and be monotone");
  {  // the staging threads walk the rows: the row pointers are checked before they start (0.3 ms at 10^6 rows)
    int bad = 0;
    for (int i = 0; i < nrow; ++i) bad |= rowptr[i + 1] < rowptr[i];
    ISPH_REQUIRE(!bad, "rowptr not monotone");
  }
  const long long nnz = rowptr[nrow];
  HostStager *H = nullptr;
  ISPH_CHECK(stager_get(ctx, &H));
  const std::chrono::steady_clock::time_point t_start = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
  double wait_fill = 0.0;

  // Chunks: a small first one (the link starts after ~0.1 ms of staging, and its rows decide whether this matrix travels
  // with 16-bit column differences), then whole slots.  EVERY chunk is staged by ALL staging threads together, each
  // taking a part of it: a 48 MiB chunk is in the ring after 0.7 ms and goes over the link in one copy at the link's best
  // rate, while the threads stage the next one.  (One thread per chunk -- the first version -- kept six chunks in flight
  // but delivered the first whole slot after 4 ms: the link idled for 3 of the first 4.5 ms.)
  std::vector<long long> cstart(1, 0);
  if (nnz > 0) cstart.push_back(std::min(nnz, (long long)HostStager::kChunk / 8));
  while (cstart.back() < nnz) cstart.push_back(std::min(nnz, cstart.back() + (long long)HostStager::kChunk));
  const long long nchunks = (long long)cstart.size() - 1;
  const bool packing = nchunks <= 64 * kModeWords;  // the mode bits travel as a kernel argument
  std::vector<int> crow((size_t)nchunks + 1, nrow);  // first row that starts at or behind the chunk's first entry
  for (long long c = 0; c <= nchunks; ++c)
    crow[(size_t)c] = (int)(std::lower_bound(rowptr, rowptr + nrow + 1, (int)cstart[(size_t)c]) - rowptr);

  // ---- the workers start on the caller's arrays at once; everything below overlaps with their first chunks
  const int P = (int)std::max<long long>(1, std::min<long long>(H->nthreads, nnz / 65536));  // parts per chunk = staging threads
  std::mutex mu;
  std::condition_variable cv;
  std::vector<char> filled((size_t)nchunks, 0), recorded((size_t)nchunks, 0);  // filled: 1 = 32-bit columns, 2 = 16-bit differences
  std::vector<std::atomic<int>> parts_done((size_t)(nchunks > 0 ? nchunks : 1));
  std::vector<std::atomic<long long>> nwide_sum((size_t)(nchunks > 0 ? nchunks : 1));
  for (auto &a : parts_done) a.store(0);
  for (auto &a : nwide_sum) a.store(0);
  std::atomic<long long> next(0);
  std::atomic<int> bad_col(0), hip_err(0);
  std::atomic<int> verdict(0);  // 16-bit differences for this matrix?  0: chunk 0 not staged yet, 1: yes, 2: no
  auto worker = [&]() {
    (void)hipSetDevice(ctx->device);
    for (;;) {
      const long long id = next.fetch_add(1);
      const long long c = id / P;
      const int part = (int)(id % P);
      if (c >= nchunks) break;
      const int slot = (int)(c % H->nslots);
      if (c >= H->nslots) {  // the slot's previous chunk must have left it
        char state = 0;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return recorded[(size_t)(c - H->nslots)] != 0; });
          state = recorded[(size_t)(c - H->nslots)];
        }
        if (state == 1 && hipEventSynchronize(H->ev[(size_t)slot]) != hipSuccess) hip_err.store(1);
      }
      const long long p0 = cstart[(size_t)c], p1 = cstart[(size_t)c + 1];
      const size_t cnt = (size_t)(p1 - p0);
      char *base = H->pslot + (size_t)slot * 12 * HostStager::kChunk;
      const int r0 = crow[(size_t)c], r1 = crow[(size_t)c + 1];
      const size_t off_rf = (10 * cnt + 3) & ~(size_t)3;
      const bool fits16 = packing && off_rf + 4 * (size_t)(r1 - r0) <= 12 * cnt;
      // columns as 16-bit differences?  chunk 0 tries (when they fit), the others follow its verdict
      bool try16 = fits16;
      if (c > 0) {
        int v = verdict.load();
        if (v == 0) {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return verdict.load() != 0; });
          v = verdict.load();
        }
        try16 = fits16 && v == 1;
      }
      // this thread's part of the chunk, boundaries on multiples of 32 entries (aligned streaming stores)
      auto cut = [&](int k) { return k >= P ? p1 : std::min(p1, p0 + (((long long)cnt * k / P) & ~31LL)); };
      const long long q_lo = cut(part), q_hi = cut(part + 1);
      unsigned over = 0;
      if (q_hi > q_lo && gat) {
        // gathered rows: the part's entries are the segments [max(rowptr[r], q_lo), min(rowptr[r + 1], q_hi)) of the permuted
        // rows r, each a contiguous piece of source row perm[r]
        int r = (int)(std::upper_bound(rowptr, rowptr + nrow + 1, (int)q_lo) - rowptr) - 1;
        unsigned nw = 0, ov = 0;
        unsigned short *d16 = reinterpret_cast<unsigned short *>(base + 8 * cnt);
        int *dc32 = reinterpret_cast<int *>(base + 8 * cnt);
        for (; r < nrow && (long long)rowptr[r] < q_hi; ++r) {
          const long long a = std::max<long long>(rowptr[r], q_lo), b = std::min<long long>(rowptr[r + 1], q_hi);
          if (b <= a) continue;
          const long long sq = (long long)gat->src_rowptr[gat->perm[r]] + (a - rowptr[r]);
          const size_t m = (size_t)(b - a);
          if (r + 3 < nrow) {  // the rows three ahead start somewhere else in the caller's arrays: ask for their first lines now
            const long long nq = gat->src_rowptr[gat->perm[r + 3]];
            __builtin_prefetch(val + nq); __builtin_prefetch(val + nq + 8); __builtin_prefetch(val + nq + 16); __builtin_prefetch(val + nq + 24);
            __builtin_prefetch(colidx + nq); __builtin_prefetch(colidx + nq + 16);
          }
          stage_copy_row(base + 8 * (size_t)(a - p0), val + sq, sizeof(double) * m);
          const int *__restrict__ src = colidx + sq;
          if (try16) {
            unsigned short *__restrict__ dst = d16 + (a - p0);
            size_t k0 = 0;
            if (a == (long long)rowptr[r]) { dst[0] = 0; ov |= (unsigned)((unsigned)src[0] >= (unsigned)ncol); k0 = 1; }  // row start: its column travels in the table
            for (size_t k = k0; k < m; ++k) {
              const int cc = src[k];
              const unsigned diff = (unsigned)(cc - src[(ptrdiff_t)k - 1]);   // the previous entry of the same source row
              nw += diff > 65535u;
              ov |= (unsigned)((unsigned)cc >= (unsigned)ncol);
              dst[k] = (unsigned short)diff;
            }
          } else {
            int *dc = dc32 + (a - p0);
            unsigned o2 = 0;
            for (size_t k = 0; k < m; ++k) o2 |= (unsigned)((unsigned)src[k] >= (unsigned)ncol);
            ov |= o2;
            stage_copy_row(dc, src, sizeof(int) * m);
          }
        }
        stage_fence();
        over |= ov;
        if (try16) nwide_sum[(size_t)c].fetch_add((long long)nw);
      } else if (q_hi > q_lo) {
        stage_copy(base + 8 * (size_t)(q_lo - p0), val + q_lo, sizeof(double) * (size_t)(q_hi - q_lo));
        if (try16) {
          // flat pass (vectorises, runs at copy speed): range check and the 16-bit difference to the previous entry, row
          // starts included; the chunk's finisher takes the row starts back
          unsigned short *d16 = reinterpret_cast<unsigned short *>(base + 8 * cnt);
          const long long qa = q_lo > 0 ? q_lo : 1;  // entry 0 has no predecessor (it is a row start)
          if (q_lo == 0) { d16[0] = 0; over |= (unsigned)((unsigned)colidx[0] >= (unsigned)ncol); }
          const int *src = colidx + qa;
          unsigned short *dst = d16 + (qa - p0);
          const size_t m = (size_t)(q_hi - qa);
          unsigned nw = 0, ov = 0;
          for (size_t k = 0; k < m; ++k) {
            const int cc = src[k];
            const unsigned diff = (unsigned)(cc - src[(ptrdiff_t)k - 1]);
            nw += diff > 65535u;
            ov |= (unsigned)((unsigned)cc >= (unsigned)ncol);
            dst[k] = (unsigned short)diff;
          }
          over |= ov;
          nwide_sum[(size_t)c].fetch_add((long long)nw);
        } else {
          int *dc = reinterpret_cast<int *>(base + 8 * cnt) + (q_lo - p0);
          const int *sc = colidx + q_lo;
          const size_t m = (size_t)(q_hi - q_lo);
          if (stage_have_avx2() && ncol > 0) {
            over |= stage_cols_avx2(dc, sc, m, (unsigned)ncol);
          } else {
            unsigned ov = 0;
            for (size_t k = 0; k < m; ++k) { const int cc = sc[k]; ov |= (unsigned)((unsigned)cc >= (unsigned)ncol); dc[k] = cc; }
            over |= ov;
          }
        }
      }
      if (over) bad_col.store(1);
      if (parts_done[(size_t)c].fetch_add(1) + 1 < P) continue;
      // ---- the thread that finishes the chunk's last part closes it
      bool pack16 = try16;
      if (try16) {
        // rows that start in this chunk: first column to the table behind the differences; a start the flat pass counted
        // as "too wide" (the previous row ended further right) is taken back.  A real difference outside [0, 65535] --
        // unsorted rows, or neighbours in a row more than 65535 columns apart -- sends the chunk as 32-bit columns.
        unsigned short *d16 = reinterpret_cast<unsigned short *>(base + 8 * cnt);
        int *rowfirst = reinterpret_cast<int *>(base + off_rf);
        long long nwide = nwide_sum[(size_t)c].load();
        for (int r = r0; r < r1; ++r) {
          const long long rs = rowptr[r];
          const long long rsrc = gat ? (long long)gat->src_rowptr[gat->perm[r]] : rs;   // where the row starts in the caller's arrays
          rowfirst[r - r0] = rs < nnz ? colidx[rsrc] : 0;  // (rows without entries: never read)
          if (rowptr[r + 1] == rs) continue;
          d16[rs - p0] = 0;
          if (!gat && rs > 0 && (unsigned)(colidx[rs] - colidx[rs - 1]) > 65535u) --nwide;   // (the gathered pass does not count row starts)
        }
        if (nwide > 0) {  // not this chunk: its columns once more, plain (this thread alone; rare after chunk 0)
          pack16 = false;
          int *dc = reinterpret_cast<int *>(base + 8 * cnt);
          if (gat) {
            int r = (int)(std::upper_bound(rowptr, rowptr + nrow + 1, (int)p0) - rowptr) - 1;
            for (; r < nrow && (long long)rowptr[r] < p1; ++r) {
              const long long a = std::max<long long>(rowptr[r], p0), b = std::min<long long>(rowptr[r + 1], p1);
              const long long sq = (long long)gat->src_rowptr[gat->perm[r]] + (a - rowptr[r]);
              for (long long k = 0; k < b - a; ++k) dc[a - p0 + k] = colidx[sq + k];
            }
          } else if (stage_have_avx2() && ncol > 0) (void)stage_cols_avx2(dc, colidx + p0, cnt, (unsigned)ncol);
          else for (size_t k = 0; k < cnt; ++k) dc[k] = colidx[p0 + (long long)k];
        }
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        if (c == 0) verdict.store(pack16 ? 1 : 2);
        filled[(size_t)c] = pack16 ? 2 : 1;
      }
      cv.notify_all();
    }
  };
  struct Pool {  // joins on every exit path; the workers always run to the end of the chunk list
    std::vector<std::thread> th;
    std::mutex &mu;
    std::condition_variable &cv;
    std::vector<char> &recorded;
    std::atomic<int> &verdict;
    ~Pool() {
      {
        std::lock_guard<std::mutex> lk(mu);
        for (char &r : recorded) if (r == 0) r = 2;  // nothing will be recorded any more: do not wait for it
        if (verdict.load() == 0) verdict.store(2);
      }
      cv.notify_all();
      for (std::thread &t : th) t.join();
    }
  } pool{{}, mu, cv, recorded, verdict};
  const int nthreads = nchunks > 0 ? P : 0;
  for (int t = 0; t < nthreads; ++t) pool.th.emplace_back(worker);

  struct Guard {  // every early return gives the matrix and the device image back
    isph_mat *A = nullptr;
    DevBuf<int> drp, dcrow;
    DevBuf<long long> dcstart;
    DevBuf<char> stage;
    hipStream_t copy_stream = nullptr;
    ~Guard() {
      (void)hipStreamSynchronize(copy_stream);  // no copy may still read the ring when the next call refills it
      drp.release(); dcrow.release(); dcstart.release(); stage.release();
      if (A) isph_mat_destroy(A);
    }
  } g;
  g.copy_stream = H->copy_stream;
  g.A = new isph_mat();
  Sell &S = g.A->S;
  S.nrow = nrow; S.ncol = ncol; S.nnz = nnz;
  S.nslices = (nrow + kSlice - 1) / kSlice;

  // slice offsets and the widest slice from the host row pointers
  std::vector<long long> so((size_t)S.nslices + 1, 0);
  int wmax = 0;
  for (int s = 0; s < S.nslices; ++s) {
    int w = 0;
    const int r1 = std::min(nrow, (s + 1) * kSlice);
    for (int r = s * kSlice; r < r1; ++r) w = std::max(w, rowptr[r + 1] - rowptr[r]);
    w = (w + 1) & ~1;
    wmax = std::max(wmax, w);
    so[(size_t)s + 1] = so[(size_t)s] + (long long)w * kSlice;
  }
  S.stored = so[(size_t)S.nslices];
  S.wmax = wmax;
  ISPH_CHECK(S.slice_off.reserve((size_t)S.nslices + 1));
  ISPH_CHECK(S.rowlen.reserve((size_t)(nrow > 0 ? nrow : 1)));
  ISPH_CHECK(S.col.reserve((size_t)(S.stored > 0 ? S.stored : 1)));
  ISPH_CHECK(S.val.reserve((size_t)(S.stored > 0 ? S.stored : 1)));
  ISPH_CHECK(g.drp.reserve((size_t)nrow + 1));
  ISPH_CHECK(g.stage.reserve((size_t)(nnz > 0 ? 12 * nnz : 16)));
  ISPH_CHECK(g.dcstart.reserve((size_t)nchunks + 1));
  ISPH_CHECK(g.dcrow.reserve((size_t)nchunks + 1));
  // small operands (synchronous copies of pageable memory: 4 MB + 125 kB at 1 M rows)
  ISPH_CHECK_HIP(hipMemcpyAsync(g.drp.p, rowptr, sizeof(int) * ((size_t)nrow + 1), hipMemcpyHostToDevice, ctx->stream));
  ISPH_CHECK_HIP(hipMemcpyAsync(S.slice_off.p, so.data(), sizeof(long long) * so.size(), hipMemcpyHostToDevice, ctx->stream));
  ISPH_CHECK_HIP(hipMemcpyAsync(g.dcstart.p, cstart.data(), sizeof(long long) * cstart.size(), hipMemcpyHostToDevice, ctx->stream));
  ISPH_CHECK_HIP(hipMemcpyAsync(g.dcrow.p, crow.data(), sizeof(int) * crow.size(), hipMemcpyHostToDevice, ctx->stream));
  ISPH_CHECK_HIP(hipMemsetAsync(H->flag.p, 0, 2 * sizeof(int), ctx->stream));
  if (nrow > 0)
    hipLaunchKernelGGL(k_csr_rowlen, dim3((nrow + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, nrow, (const int *)g.drp.p, S.rowlen.p);
  if (hooks && hooks->begin) ISPH_CHECK(hooks->begin(g.A));
  H->stats[0] = since();

  CsrChunks ck{};
  ck.stage = g.stage.p; ck.cstart = g.dcstart.p; ck.crow = g.dcrow.p; ck.nchunks = (int)nchunks;
  int rc = ISPH_SUCCESS;
  int slices_done = 0;
  long long sent = 0;
  auto convert_upto = [&](long long entries_arrived) {
    // slices whose 64 rows end at or before the last uploaded entry
    int lo = slices_done, hi = S.nslices;
    while (lo < hi) {
      const int mid = (lo + hi + 1) / 2;
      if ((long long)rowptr[std::min(nrow, mid * kSlice)] <= entries_arrived) lo = mid; else hi = mid - 1;
    }
    if (lo > slices_done) {
      const int cnt = lo - slices_done;
      hipLaunchKernelGGL(k_csr_to_sell<int>, dim3((cnt + 3) / 4), dim3(kBlock), 0, ctx->stream, nrow, (const int *)g.drp.p,
                         (const int *)nullptr, (const double *)nullptr, (const long long *)S.slice_off.p, S.col.p, S.val.p,
                         slices_done, lo, gat ? (int *)nullptr : H->flag.p, ck, gat ? gat->colren : (const int *)nullptr,
                         gat ? gat->nren : 0);
      const int s0 = slices_done;
      slices_done = lo;
      if (hooks && hooks->slices && rc == ISPH_SUCCESS) rc = hooks->slices(g.A, s0, lo);
    }
  };
  for (long long c = 0; c < nchunks; ++c) {
    char how = 0;
    {
      const double w0 = since();
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return filled[(size_t)c] != 0; });
      how = filled[(size_t)c];
      wait_fill += since() - w0;
    }
    const int slot = (int)(c % H->nslots);
    const long long p0 = cstart[(size_t)c];
    const size_t cnt = (size_t)(cstart[(size_t)c + 1] - p0);
    char state = 1;  // 1: event recorded, 2: nothing to wait for (a HIP call failed; workers must not block on the event)
    if (rc == ISPH_SUCCESS) {
      // ONE copy per chunk: values, columns and (16-bit mode) the first columns of the rows that start in it
      const size_t bytes = how == 2 ? ((10 * cnt + 3) & ~(size_t)3) + 4 * (size_t)(crow[(size_t)c + 1] - crow[(size_t)c]) : 12 * cnt;
      if (how == 2) ck.mode16[c >> 6] |= 1ull << (c & 63);
      if (hipMemcpyAsync(g.stage.p + 12 * p0, H->pslot + (size_t)slot * 12 * HostStager::kChunk, bytes, hipMemcpyHostToDevice, H->copy_stream) != hipSuccess ||
          hipEventRecord(H->ev[(size_t)slot], H->copy_stream) != hipSuccess ||
          hipStreamWaitEvent(ctx->stream, H->ev[(size_t)slot], 0) != hipSuccess)
        rc = fail("upload of a CSR chunk failed", __FILE__, __LINE__);
      else
        convert_upto(p0 + (long long)cnt);
      sent += (long long)bytes;
    }
    if (rc != ISPH_SUCCESS) state = 2;
    {
      std::lock_guard<std::mutex> lk(mu);
      recorded[(size_t)c] = state;
    }
    cv.notify_all();
  }
  if (rc == ISPH_SUCCESS) convert_upto(nnz);  // rows without entries at the end, or nnz == 0
  if (rc == ISPH_SUCCESS && slices_done < S.nslices) rc = fail("row pointers and entry count disagree", __FILE__, __LINE__);
  // the ring is reused by the next call and the caller may free its arrays: wait for the copies; the flags travel back
  // behind everything queued so far (conversion and hooks), which is the one synchronisation of the ingress
  int flags[2] = {0, 0};
  H->stats[1] = since();
  const bool copies_ok = hipStreamSynchronize(H->copy_stream) == hipSuccess;
  H->stats[2] = since();
  if (!copies_ok ||
      hipMemcpyAsync(flags, H->flag.p, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess || hip_err.load())
    if (rc == ISPH_SUCCESS) rc = fail("CSR->SELL conversion failed", __FILE__, __LINE__);
  H->stats[3] = since();
  ISPH_CHECK(rc);
  ISPH_REQUIRE(!bad_col.load(), "column index out of range");
  if (flags[0]) {
    ISPH_CHECK(sell_sort_rows(ctx, S));
    ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  }
  if (was_unsorted) *was_unsorted = flags[0] != 0;
  if (S.col16.p != nullptr) {  // a hook built the 16-bit columns slice by slice
    S.c16_state = flags[1] ? -1 : flags[0] ? 0 : 1;  // unsorted rows were permuted afterwards: rebuilt on first use
    if (S.c16_state != 1) { S.col16.release(); S.wtab.release(); }
  }
  ISPH_CHECK_HIP(hipGetLastError());
  H->bytes_sent = sent + 4 * ((long long)nrow + 1);
  H->stats[4] = since(); H->stats[5] = wait_fill; H->stats[6] = (double)H->bytes_sent; H->stats[7] = (double)nthreads;
  *Aout = g.A;
  g.A = nullptr;
  return ISPH_SUCCESS;
}

// Host CSR ingress fused with the set-up of the block-Jacobi ILU(0) preconditioner (block_size rows per subdomain):
// the 16-bit column windows and the ILU extraction / schedule / factorisation of a range of blocks are queued as soon
// as the range's rows have been converted, in batches of 1/kBatchDiv of the blocks on two alternating side streams (a
// factorisation launch lives ~2 ms whatever its size -- the per-block dependency chain -- while a twelfth of the 100^3
// matrix takes 1.9 ms on the link: on one stream the batches queue up behind each other).  Measured at 100^3
// (profiles/r03_dropin.txt): copies done after 24.3 ms, set-up done 3.1 ms later for every batch size from 1/8 to 1/24;
// with more side streams than hardware queues (HIP maps its streams onto 4) the copy stream ends up sharing a queue with
// a 2 ms factorisation kernel and the link idles: 32-38 ms.
constexpr int kBatchDiv = 12;
// nblocks_tab > 0: the caller's subdomains (host table bptr[0 .. nblocks_tab], isph_mat_create_csr_blocks); block_size is then
// the capacity of a block.  Their factor regions are the entry ranges of their rows in the host CSR, 64-aligned -- known
// from the row pointers before the first entry has crossed the link.
inline int csr_ingress_host_bjacobi(isph_ctx *ctx, int nrow, int ncol, const int *rowptr, const int *colidx, const double *val,
                                    int block_size, isph_mat **Aout, isph_ilu **Fout, int nblocks_tab = 0,
                                    const int *bptr = nullptr, const IngressGather *gat = nullptr) {
  ISPH_REQUIRE(block_size >= 64 && block_size <= 1024 && block_size % 64 == 0,
               "block-Jacobi ILU block size must be a multiple of 64 in [64,1024]");
  const bool var = nblocks_tab > 0;
  if (var) {
    ISPH_REQUIRE(bptr && bptr[0] == 0 && bptr[nblocks_tab] == nrow, "subdomain table must run from 0 to the number of rows");
    for (int b = 0; b < nblocks_tab; ++b)
      ISPH_REQUIRE(bptr[b + 1] > bptr[b] && bptr[b + 1] - bptr[b] <= block_size, "every subdomain needs between 1 and `capacity` rows");
  }
  HostStager *H = nullptr;
  ISPH_CHECK(stager_get(ctx, &H));
  isph_ilu *F = nullptr;
  int blocks_done = 0, nbatch = 0;
  const int spb = block_size / 64;
  IngressHooks hooks;
  hooks.begin = [&](isph_mat *A) -> int {
    const Sell &S = A->S;
    if (S.nrow == 0) return ISPH_SUCCESS;
    ISPH_CHECK(S.col16.reserve((size_t)(S.stored > 0 ? S.stored : 1)));
    ISPH_CHECK(S.wtab.reserve((size_t)S.nslices * 64));
    if (!var) {
      ISPH_CHECK(ilu_begin(ctx, S, block_size, false, 0, &F));
      return ilu_begin_fill0(ctx, F, S);
    }
    ISPH_CHECK(ilu_begin(ctx, S, block_size, false, 0, &F, /*defer_factor_arrays=*/true, nblocks_tab, bptr));
    std::vector<long long> hb((size_t)nblocks_tab + 1, 0);
    for (int b = 0; b < nblocks_tab; ++b)
      hb[(size_t)b + 1] = hb[(size_t)b] + (((long long)rowptr[bptr[b + 1]] - rowptr[bptr[b]] + 63) & ~63LL);
    F->total = hb[(size_t)nblocks_tab];
    F->compact = true;
    const size_t tot1 = (size_t)(F->total > 0 ? F->total : 1);
    ISPH_CHECK(F->fcol.reserve(tot1));
    ISPH_CHECK(F->fval.reserve(tot1));
    ISPH_CHECK(F->fdst.reserve(tot1));
    ISPH_CHECK_HIP(hipMemcpyAsync(F->boff.p, hb.data(), sizeof(long long) * hb.size(), hipMemcpyHostToDevice, ctx->stream));
    ISPH_CHECK_HIP(hipStreamSynchronize(ctx->stream));  // hb leaves scope
    return ilu_size_stream(F);
  };
  // blocks whose last row lies in a converted slice
  auto blocks_ready = [&](int s1, int nslices) {
    if (s1 == nslices) return F->nblocks;
    if (!var) return s1 / spb;
    const int rows = s1 * 64;
    int lo = 0, hi = nblocks_tab;
    while (lo < hi) { const int mid = (lo + hi + 1) / 2; if (bptr[mid] <= rows) lo = mid; else hi = mid - 1; }
    return lo;
  };
  // The hook kernels run BEFORE the host has reported an out-of-range column (bad_col is checked when the ingress ends):
  // none of them indexes memory by a column -- k_sell_compress_cols takes differences, the ILU extract / schedule /
  // factor kernels compare a column against the block's row range and only use it as an index inside that range.
  hooks.slices = [&](isph_mat *A, int s0, int s1) -> int {
    const Sell &S = A->S;
    const int ready = blocks_ready(s1, S.nslices);
    const int batch = std::max(1, F->nblocks / kBatchDiv);
    const bool fire = ready > blocks_done && (ready - blocks_done >= batch || s1 == S.nslices);
    if (gat && s1 > s0) {
      // gathered rows carry renamed columns: every row of the range is sorted by them first (k_sell_sort_rows, ranged)
      const int Ws = S.wmax | 1;
      int R = 64;
      while (R > 1 && (size_t)R * Ws * 24 > 48 * 1024) R >>= 1;
      ISPH_REQUIRE((size_t)R * Ws * 24 <= 150 * 1024, "row too long for the LDS row sort");
      const size_t lds = (size_t)R * Ws * 24;
      ISPH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sell_sort_rows), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k_sell_sort_rows, dim3(s1 - s0), dim3(kBlock), lds, ctx->stream, S.nrow, S.nslices, R, Ws, (const int *)S.rowlen.p,
                         (const long long *)S.slice_off.p, S.col.p, S.val.p, s0);
    }
    if (fire) ISPH_CHECK_HIP(hipEventRecord(H->ev_conv, ctx->stream));  // behind the conversion, in front of the column windows
    hipLaunchKernelGGL(k_sell_compress_cols, dim3((s1 - s0 + 3) / 4), dim3(kBlock), 0, ctx->stream, s0, s1,
                       (const long long *)S.slice_off.p, (const int *)S.col.p, S.col16.p, S.wtab.p, H->flag.p + 1);
    if (fire) {
      // consecutive batches run on alternating side streams behind the conversion of their rows, so their
      // dependency-chain latencies overlap each other and the link
      const int nb = ready - blocks_done;
      hipStream_t st = H->aux[nbatch % HostStager::kAux];
      ISPH_CHECK_HIP(hipStreamWaitEvent(st, H->ev_conv, 0));
      ilu_launch_extract(ctx, F, S, blocks_done, nb, st);
      ilu_launch_schedule(ctx, F, S, blocks_done, nb, false, st);
      ISPH_CHECK(ilu_launch_factor(ctx, F, S, blocks_done, nb, F->err.p, st));
      blocks_done = ready;
      ++nbatch;
    }
    if (s1 == S.nslices)  // the main stream (flags, the caller's next work) continues behind every batch
      for (int k = 0; k < HostStager::kAux; ++k) {
        ISPH_CHECK_HIP(hipEventRecord(H->ev_aux[k], H->aux[k]));
        ISPH_CHECK_HIP(hipStreamWaitEvent(ctx->stream, H->ev_aux[k], 0));
      }
    return ISPH_SUCCESS;
  };
  isph_mat *A = nullptr;
  bool unsorted = false;
  int rc = csr_ingress_host(ctx, nrow, ncol, rowptr, colidx, val, &A, &hooks, &unsorted, gat);
  bool redo = unsorted;
  if (rc == ISPH_SUCCESS && F && !unsorted && nrow > 0) {
    bool overflow = false;
    rc = ilu_check_err(ctx, F, "ILU set-up during the matrix ingress failed", &overflow);
    redo = overflow;
  }
  if (rc == ISPH_SUCCESS && (redo || !F)) {  // sorted now / stream at its proven capacity: the plain set-up
    if (F) ilu_destroy(F);
    F = nullptr;
    rc = ilu_create(ctx, A, block_size, &F, false, 0, nblocks_tab, bptr);
  }
  if (rc != ISPH_SUCCESS) {
    // a failure may have skipped the hook call that joins the side streams (it only runs while rc is good): batches
    // queued on them still read F and A -- drain them before the objects go back to the pool
    for (int k = 0; k < HostStager::kAux; ++k) (void)hipStreamSynchronize(H->aux[k]);
    (void)hipStreamSynchronize(ctx->stream);
    if (F) ilu_destroy(F);
    if (A) isph_mat_destroy(A);
    return rc;
  }
  *Aout = A;
  *Fout = F;
  return ISPH_SUCCESS;
}

}  // namespace isph
