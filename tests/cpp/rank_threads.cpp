// rank_threads.cpp -- TEST transport: N ranks as N threads of ONE process, all on one GPU.
//
// Implements the two callbacks of isph_host_transport (include/isph_hip.h) with MPI's semantics -- eager point-to-point
// messages matched per (source, destination) in order, an all-reduce every rank enters -- over mutex-protected mailboxes.
// The multi-rank tests (tests/test_gpu_ranks.py) give every rank thread its own isph_ctx through
// isph_ctx_create_hostcomm: RCCL cannot put two ranks on one device and the GPU box allows six processes on its card,
// so this is how the 8-rank decomposition of BASELINE configs[2] runs on one MI355X.  Not part of the product.
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "isph_hip.h"

namespace {

struct Group {
  int n = 0;
  std::mutex mu;
  std::condition_variable cv;
  std::map<std::pair<int, int>, std::deque<std::vector<double>>> box;  // (src, dst) -> messages in order
  // all-reduce: contributions by rank (summed in rank order by the last arrival: the result does not depend on timing)
  std::vector<std::vector<double>> slot;
  std::vector<double> result;
  int arrived = 0;
  long generation = 0;
  int ar_count = -1, ar_op = -1;
  bool aborted = false;
  double timeout_s = 300.0;
  long n_exchange = 0, n_allreduce = 0;  // call counters (all ranks together), for the tests
};

struct Rank {
  Group *g;
  int rank;
};

template <class Pred>
bool wait_for(Group *g, std::unique_lock<std::mutex> &lk, Pred pred) {
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(g->timeout_s);
  while (!pred()) {
    if (g->aborted) return false;
    if (g->cv.wait_until(lk, deadline) == std::cv_status::timeout && !pred()) {
      g->aborted = true;  // one rank gave up: release everybody
      g->cv.notify_all();
      return false;
    }
  }
  return !g->aborted;
}

int rt_exchange(void *user, int npeers, const int *peer, const double *send, const long long *so, double *recv,
                const long long *ro) {
  Rank *R = static_cast<Rank *>(user);
  Group *g = R->g;
  std::unique_lock<std::mutex> lk(g->mu);
  if (g->aborted) return 1;
  ++g->n_exchange;
  for (int p = 0; p < npeers; ++p) {
    if (peer[p] < 0 || peer[p] >= g->n) return 1;
    const long long n = so[p + 1] - so[p];
    if (n > 0) g->box[{R->rank, peer[p]}].emplace_back(send + so[p], send + so[p + 1]);
  }
  g->cv.notify_all();
  for (int p = 0; p < npeers; ++p) {
    const long long n = ro[p + 1] - ro[p];
    if (n <= 0) continue;
    std::deque<std::vector<double>> &q = g->box[{peer[p], R->rank}];
    if (!wait_for(g, lk, [&] { return !q.empty(); })) return 1;
    if ((long long)q.front().size() != n) { g->aborted = true; g->cv.notify_all(); return 1; }  // plans disagree
    std::memcpy(recv + ro[p], q.front().data(), sizeof(double) * (size_t)n);
    q.pop_front();
  }
  return 0;
}

int rt_allreduce(void *user, double *buf, int count, int op) {
  Rank *R = static_cast<Rank *>(user);
  Group *g = R->g;
  std::unique_lock<std::mutex> lk(g->mu);
  if (g->aborted) return 1;
  ++g->n_allreduce;
  if (g->arrived == 0) { g->ar_count = count; g->ar_op = op; }
  else if (g->ar_count != count || g->ar_op != op) { g->aborted = true; g->cv.notify_all(); return 1; }  // ranks out of step
  g->slot[(size_t)R->rank].assign(buf, buf + count);
  const long gen = g->generation;
  if (++g->arrived == g->n) {
    g->result = g->slot[0];
    for (int r = 1; r < g->n; ++r)
      for (int k = 0; k < count; ++k) {
        const double v = g->slot[(size_t)r][(size_t)k];
        g->result[(size_t)k] = op == 1 ? (v > g->result[(size_t)k] ? v : g->result[(size_t)k]) : g->result[(size_t)k] + v;
      }
    g->arrived = 0;
    ++g->generation;
    g->cv.notify_all();
  } else if (!wait_for(g, lk, [&] { return g->generation != gen; })) {
    return 1;
  }
  std::memcpy(buf, g->result.data(), sizeof(double) * (size_t)count);
  return 0;
}

}  // namespace

extern "C" {

void *rt_group_create(int nranks, double timeout_s) {
  Group *g = new Group();
  g->n = nranks;
  g->slot.resize((size_t)nranks);
  if (timeout_s > 0) g->timeout_s = timeout_s;
  return g;
}

void rt_group_destroy(void *group) { delete static_cast<Group *>(group); }

// a rank thread that fails for its own reasons calls this so that the others do not wait for it
void rt_group_abort(void *group) {
  Group *g = static_cast<Group *>(group);
  std::lock_guard<std::mutex> lk(g->mu);
  g->aborted = true;
  g->cv.notify_all();
}

int rt_group_aborted(void *group) {
  Group *g = static_cast<Group *>(group);
  std::lock_guard<std::mutex> lk(g->mu);
  return g->aborted ? 1 : 0;
}

void rt_group_counts(void *group, long long out[2]) {
  Group *g = static_cast<Group *>(group);
  std::lock_guard<std::mutex> lk(g->mu);
  out[0] = g->n_exchange;
  out[1] = g->n_allreduce;
}

// fills *out with the callbacks of rank `rank`; the returned handle owns the per-rank state (rt_rank_destroy)
void *rt_rank_create(void *group, int rank, isph_host_transport *out) {
  Rank *R = new Rank{static_cast<Group *>(group), rank};
  out->user = R;
  out->exchange = rt_exchange;
  out->allreduce = rt_allreduce;
  return R;
}

void rt_rank_destroy(void *rank) { delete static_cast<Rank *>(rank); }

}  // extern "C"
