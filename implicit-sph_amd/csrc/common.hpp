// common.hpp -- shared host/device helpers of libisph_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "isph_hip.h"

namespace isph {

extern thread_local std::string g_last_error;

inline int fail(const char *what, const char *file, int line) {
  char buf[512];
  snprintf(buf, sizeof(buf), "%s (%s:%d)", what, file, line);
  g_last_error = buf;
  return ISPH_FAILURE;
}

#define ISPH_CHECK_HIP(expr)                                                        \
  do {                                                                              \
    hipError_t e_ = (expr);                                                         \
    if (e_ != hipSuccess) return ::isph::fail(hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define ISPH_CHECK_NCCL(expr)                                                        \
  do {                                                                               \
    ncclResult_t e_ = (expr);                                                        \
    if (e_ != ncclSuccess) return ::isph::fail(ncclGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define ISPH_CHECK(expr)                                  \
  do {                                                    \
    int rc_ = (expr);                                     \
    if (rc_ != ISPH_SUCCESS) return rc_;                  \
  } while (0)
#define ISPH_REQUIRE(cond, msg)                                        \
  do {                                                                 \
    if (!(cond)) return ::isph::fail(msg, __FILE__, __LINE__);        \
  } while (0)

// Process-wide cache of freed device blocks.  The matrix, the ILU factor and its stream are rebuilt every
// time step (the reference deletes A / graph / map after every compute(), pair_isph.cpp:1351-1372); going
// back to hipMalloc/hipFree for multi-GB buffers each step costs milliseconds and an implicit device sync
// per call.  Blocks are reused when the request fits within 25 % slack; at most cap() bytes stay cached (isph_pool_trim returns them).
// A released block may still be read by kernels in flight (on any stream of the process), so release() only
// parks it on a PENDING list without synchronising; the first allocation that finds no ready block drains the
// device once (one hipDeviceSynchronize for all parked blocks -- in steady state one per solve set-up, none in the
// Krylov loop) and makes the whole list ready.  Nothing is ever handed out while work that was queued before
// its release can still touch it, whichever stream that work runs on.
struct DevPool {
  // freed blocks are kept for the next set-up (the preconditioner is rebuilt every solve): a fresh hipMalloc of tens of
  // GB costs seconds (1.7 s for 48 GB on MI355X), and the BASELINE configs[4] operator (36 GB) needs several such blocks
  // per set-up.  The cache may hold up to 80 % of the memory that was FREE when the library first released a block --
  // not of the device's total: torch, RCCL and other contexts of the process own memory this pool cannot see, and they
  // meet an out-of-memory error, not this pool's trim-and-retry.  isph_pool_set_cap() overrides the limit and
  // isph_pool_trim() gives everything back (call it before another allocator of the process needs the room).
  size_t cap_bytes = 0;
  size_t cap() {
    if (cap_bytes == 0) {
      size_t fr = 0, tot = 0;
      cap_bytes = (hipMemGetInfo(&fr, &tot) == hipSuccess && fr > 0) ? (fr + cached) / 5 * 4 : ((size_t)48 << 30);
    }
    return cap_bytes;
  }
  std::multimap<size_t, void *> free_blocks;   // ready: no kernel can still touch them
  std::vector<std::pair<size_t, void *>> pending;  // released since the last device synchronisation
  size_t cached = 0;
  size_t live = 0, peak_live = 0;  // bytes handed out and not yet given back; its high-water mark (isph_pool_info)
  std::mutex mu;
  static DevPool &get() {
    static DevPool p;
    return p;
  }
  void *take_ready(size_t bytes, size_t *got) {
    auto it = free_blocks.lower_bound(bytes);
    if (it != free_blocks.end() && it->first <= bytes + bytes / 4 + 4096) {
      void *p = it->second;
      *got = it->first;
      cached -= it->first;
      free_blocks.erase(it);
      return p;
    }
    return nullptr;
  }
  void note_live(size_t got) {
    live += got;
    if (live > peak_live) peak_live = live;
  }
  void *alloc(size_t bytes, size_t *got) {
    {
      std::lock_guard<std::mutex> lk(mu);
      if (void *p = take_ready(bytes, got)) { note_live(*got); return p; }
      bool fits = false;
      for (auto &kv : pending) fits = fits || (kv.first >= bytes && kv.first <= bytes + bytes / 4 + 4096);
      if (fits) {
        (void)hipDeviceSynchronize();
        for (auto &kv : pending) free_blocks.emplace(kv.first, kv.second);
        pending.clear();
        if (void *p = take_ready(bytes, got)) { note_live(*got); return p; }
      }
    }
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
      trim();  // give cached blocks back and retry once
      if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    }
    *got = bytes;
    {
      std::lock_guard<std::mutex> lk(mu);
      note_live(bytes);
    }
    return p;
  }
  void release(void *p, size_t bytes) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(mu);
    live -= bytes <= live ? bytes : live;
    if (bytes >= 4096 && cached + bytes <= cap()) {
      pending.emplace_back(bytes, p);
      cached += bytes;
    } else {
      (void)hipFree(p);  // synchronises by itself
    }
  }
  void trim() {
    std::lock_guard<std::mutex> lk(mu);
    for (auto &kv : free_blocks) (void)hipFree(kv.second);
    for (auto &kv : pending) (void)hipFree(kv.second);
    free_blocks.clear();
    pending.clear();
    cached = 0;
  }
};

// grow-only device buffer backed by the pool (never allocates inside the Krylov loop:
// workspaces are sized once per solve)
// ISPH_POOL_CANARY=1 (debugging aid; the GPU address sanitizer is not available on every system): every device buffer is
// followed by kCanaryBytes of a known pattern that is checked, after a device synchronisation, when the buffer goes back
// to the pool -- a kernel that wrote past the end of what was reserved aborts the process there.  Buffers then hold
// exactly the requested element count (no pool slack), so growing re-allocates.
constexpr size_t kCanaryBytes = 256;
inline bool pool_canary() {
  static const bool on = [] { const char *e = getenv("ISPH_POOL_CANARY"); return e && e[0] == '1'; }();
  return on;
}
inline void canary_arm(void *p, size_t at) { (void)hipMemset(static_cast<char *>(p) + at, 0xA5, kCanaryBytes); }
inline void canary_check(const void *p, size_t at) {
  unsigned char h[kCanaryBytes];
  (void)hipDeviceSynchronize();
  if (hipMemcpy(h, static_cast<const char *>(p) + at, kCanaryBytes, hipMemcpyDeviceToHost) != hipSuccess) return;
  for (size_t k = 0; k < kCanaryBytes; ++k)
    if (h[k] != 0xA5) {
      fprintf(stderr, "ISPH_POOL_CANARY: a kernel wrote %zu bytes past the end of a %zu-byte device buffer\n", k + 1, at);
      abort();
    }
}

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;       // elements usable
  size_t bytes = 0;     // size of the underlying block
  size_t guard_at = 0;  // canary mode: offset of the pattern (= bytes requested, rounded up to 8)
  int reserve(size_t n) {
    if (n <= cap) return ISPH_SUCCESS;
    release();
    size_t got = 0;
    const bool canary = pool_canary();
    const size_t want = (n * sizeof(T) + 7) / 8 * 8;
    p = static_cast<T *>(DevPool::get().alloc(canary ? want + kCanaryBytes : n * sizeof(T), &got));
    if (!p) return ::isph::fail("device allocation failed", __FILE__, __LINE__);
    bytes = got;
    cap = canary ? n : got / sizeof(T);
    guard_at = canary ? want : 0;
    if (canary) canary_arm(p, guard_at);
    return ISPH_SUCCESS;
  }
  void release() {
    if (p && guard_at) canary_check(p, guard_at);
    if (p) DevPool::get().release(p, bytes);
    p = nullptr;
    cap = 0;
    bytes = 0;
    guard_at = 0;
  }
};

// function-local temporary: gives its block back when the scope ends, on every path (an ISPH_CHECK that returns early
// from a set-up with ten work arrays would otherwise leak them exactly when memory is short).  release() stays
// available for giving a large buffer back before the scope ends.  Not copyable: buffers that move into an object
// (factor arrays, matrix storage) are plain DevBuf members released by the object's destroy function.
template <class T>
struct DevTmp : DevBuf<T> {
  DevTmp() = default;
  DevTmp(const DevTmp &) = delete;
  DevTmp &operator=(const DevTmp &) = delete;
  ~DevTmp() { this->release(); }
};

// true when p is device memory: a caller that says on_device = 0 but hands over a device pointer would make the host
// dereference it (a page fault that looks like a hang), so the staging helpers refuse such a pointer
inline bool is_device_pointer(const void *p) {
  if (!p) return false;
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }  // plain host memory
  return at.type == hipMemoryTypeDevice;
}

constexpr int kWave = 64;          // gfx950 wavefront
constexpr int kSlice = 64;         // SELL slice height = one wavefront
constexpr int kBlock = 256;        // 4 waves
constexpr int kXcd = 8;            // XCDs (per-XCD L2)
constexpr int kMaxRedBlocks = 2048;

// ---- wave-level reductions with DPP (no LDS traffic) --------------------
// row_shr / row_bcast patterns of the gfx9 DPP encoding; a double moves as two
// 32-bit halves.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_move(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}

// sum over the 64 lanes; result valid in lane 63, broadcast by readlane
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_move<0x111>(v);        // row_shr:1
  v += dpp_move<0x112>(v);        // row_shr:2
  v += dpp_move<0x114>(v);        // row_shr:4
  v += dpp_move<0x118>(v);        // row_shr:8   -> lane 15 of each row holds the row sum
  v += dpp_move<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v += dpp_move<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
  int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

// max over the 64 lanes, in every lane's return value.  DPP moves like wave_sum (a lane without a source keeps its own
// value: max(v, v)); the __shfl_xor butterfly this replaces is six dependent ds_bpermute round trips, ~100 cycles each, and
// sat in the sequential level walk of k_ilu_schedule once per row.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_keep_i32(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ int wave_max_i32(int v) {
  v = max(v, dpp_keep_i32<0x111>(v));        // row_shr:1
  v = max(v, dpp_keep_i32<0x112>(v));        // row_shr:2
  v = max(v, dpp_keep_i32<0x114>(v));        // row_shr:4
  v = max(v, dpp_keep_i32<0x118>(v));        // row_shr:8   -> lane 15 of each row holds the row's max
  v = max(v, dpp_keep_i32<0x142, 0xa>(v));   // row_bcast:15 into rows 1 and 3
  v = max(v, dpp_keep_i32<0x143, 0xc>(v));   // row_bcast:31 into rows 2 and 3
  return __builtin_amdgcn_readlane(v, 63);
}

// sum within aligned groups of 8 lanes (result in every lane of the group)
__device__ __forceinline__ double group8_sum(double v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  return v;
}

// sum within aligned groups of 16 lanes (a DPP row), result in every lane: the butterfly v += v(lane ^ 1), ^ 2, ^ 4,
// ^ 8 with the exchanges done by DPP moves instead of ds_bpermute, which queues behind other LDS traffic (the polling
// reads of the triangular sweeps).  quad_perm gives the partners of the first two steps; after them a quad holds one
// value, so the mirror of 8 lanes and the mirror of 16 deliver exactly the operands of the ^ 4 and ^ 8 steps --
// the same bits as the __shfl_xor butterfly.
__device__ __forceinline__ double group16_sum(double v) {
  v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);   // row_half_mirror
  v += dpp_move<0x140>(v);   // row_mirror
  return v;
}

// XCD-aware block remap: the dispatcher deals blocks round-robin over the 8
// XCDs, so blockIdx b lands on XCD b%8.  Give each XCD one contiguous range of
// work items so its private L2 only sees that range's slice of x.
__device__ __forceinline__ int xcd_remap(int b, int nblocks_padded) {
  const int per = nblocks_padded / kXcd;
  return (b % kXcd) * per + (b / kXcd);
}

// row kernels of the assembly: launched on xcd_grid(blocks) workgroups, block index through xcd_block().  The rows an
// XCD works on at one time are then neighbours in space, and the neighbour records they gather (72 B per particle,
// +-2 lattice planes around the active rows) fit its 4 MB L2; dealt round-robin the eight XCDs each sweep the whole
// domain at once and every gather misses.
__device__ __forceinline__ int xcd_block() { return xcd_remap((int)blockIdx.x, (int)gridDim.x); }
inline int xcd_grid(int blocks) { return (blocks + kXcd - 1) / kXcd * kXcd; }

}  // namespace isph
