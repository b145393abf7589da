// comm.hpp -- the two communication primitives of the multi-rank path behind one seam.
//
// Everything the ranks ever say to each other on this path is one of
//   * comm_exchange : for every peer p, send[send_ptr[p] .. send_ptr[p+1]) goes to rank peer[p] and
//                     recv[recv_ptr[p] .. recv_ptr[p+1]) arrives from it (the Epetra_Import inside
//                     Epetra_CrsMatrix::Apply, solver_lin.h:133; LAMMPS' forward_comm_pair, pair_isph.cpp:1924-2110;
//                     the row overlap of Ifpack_AdditiveSchwarz, precond_ifpack.h:43) and
//   * comm_allreduce: in-place sum / max of a few doubles (Epetra's Dot / Norm2, solver_lin.cpp:72-74; the MPI_Allreduce
//                     MAX of shiftParticles).
// Both take DEVICE buffers and a stream and are ordered on that stream like a kernel.
//
// Transports:
//   RCCL  (isph_ctx_create_dist)     grouped ncclSend/ncclRecv and ncclAllReduce enqueued on the stream: one process per
//                                    GPU over xGMI, the production transport; the host never waits.
//   host  (isph_ctx_create_hostcomm) the buffers are staged through pinned host memory and handed to two callbacks the
//                                    caller supplies (MPI_Isend/Irecv/Waitall + MPI_Allreduce in host/mpi_transport.h):
//                                    for ranks that SHARE a device, where RCCL refuses to form a communicator -- several
//                                    MPI ranks of a LAMMPS run on one GPU, and the multi-rank tests on a one-GPU box.
//                                    The host blocks in each call; kernels, streams and their order are the same.
#pragma once
#include "core.hpp"

namespace isph {

inline bool comm_active(const isph_ctx *ctx) { return ctx->comm != nullptr || ctx->host_tr.exchange != nullptr; }

inline int comm_stage_reserve(isph_ctx *ctx, size_t nsend, size_t nrecv) {
  if (nsend > ctx->hsend_cap) {
    if (ctx->hsend) (void)hipHostFree(ctx->hsend);
    ctx->hsend = nullptr; ctx->hsend_cap = 0;
    const size_t cap = nsend + nsend / 2 + 64;
    ISPH_CHECK_HIP(hipHostMalloc((void **)&ctx->hsend, cap * sizeof(double)));
    ctx->hsend_cap = cap;
  }
  if (nrecv > ctx->hrecv_cap) {
    if (ctx->hrecv) (void)hipHostFree(ctx->hrecv);
    ctx->hrecv = nullptr; ctx->hrecv_cap = 0;
    const size_t cap = nrecv + nrecv / 2 + 64;
    ISPH_CHECK_HIP(hipHostMalloc((void **)&ctx->hrecv, cap * sizeof(double)));
    ctx->hrecv_cap = cap;
  }
  return ISPH_SUCCESS;
}

// in-place all-reduce of `count` device doubles on stream s; op 0 = sum, 1 = max.  No communicator: nothing to do.
inline int comm_allreduce(isph_ctx *ctx, double *d, int count, int op, hipStream_t s) {
  if (ctx->comm) {
    ISPH_CHECK_NCCL(ncclAllReduce(d, d, (size_t)count, ncclDouble, op == 1 ? ncclMax : ncclSum, ctx->comm, s));
    return ISPH_SUCCESS;
  }
  if (!ctx->host_tr.allreduce) return ISPH_SUCCESS;
  ISPH_CHECK(comm_stage_reserve(ctx, (size_t)count, 0));
  ISPH_CHECK_HIP(hipMemcpyAsync(ctx->hsend, d, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, s));
  ISPH_CHECK_HIP(hipStreamSynchronize(s));
  if (ctx->host_tr.allreduce(ctx->host_tr.user, ctx->hsend, count, op) != 0)
    return fail("host transport: all-reduce failed", __FILE__, __LINE__);
  ISPH_CHECK_HIP(hipMemcpyAsync(d, ctx->hsend, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, s));
  ISPH_CHECK_HIP(hipStreamSynchronize(s));  // the pinned buffer is reused by the next call, whichever stream that is on
  return ISPH_SUCCESS;
}

// Point-to-point exchange with every peer of a halo plan, `ncomp` doubles per listed entry, on stream s.
// reverse = false: send ranges -> peers, recv ranges <- peers (owners to ghosts); true: the roles swapped (ghost
// contributions back to their owners, the "Add" of the overlapped Schwarz).
inline int comm_exchange(isph_ctx *ctx, const isph_halo &H, const double *send, double *recv, int ncomp, bool reverse,
                         hipStream_t s) {
  if (H.npeers == 0) return ISPH_SUCCESS;
  const std::vector<int> &sp = reverse ? H.recv_ptr : H.send_ptr, &rp = reverse ? H.send_ptr : H.recv_ptr;
  const size_t nc = (size_t)ncomp, np = (size_t)H.npeers;
  if (ctx->comm) {
    ncclResult_t nr = ncclGroupStart();
    for (size_t p = 0; p < np && nr == ncclSuccess; ++p) {
      const size_t ns = (size_t)(sp[p + 1] - sp[p]) * nc, nv = (size_t)(rp[p + 1] - rp[p]) * nc;
      if (ns > 0) nr = ncclSend(send + (size_t)sp[p] * nc, ns, ncclDouble, H.peer[p], ctx->comm, s);
      if (nr == ncclSuccess && nv > 0) nr = ncclRecv(recv + (size_t)rp[p] * nc, nv, ncclDouble, H.peer[p], ctx->comm, s);
    }
    const ncclResult_t ne = ncclGroupEnd();
    if (nr != ncclSuccess || ne != ncclSuccess) return fail("RCCL point-to-point exchange failed", __FILE__, __LINE__);
    return ISPH_SUCCESS;
  }
  ISPH_REQUIRE(ctx->host_tr.exchange, "an exchange with peers needs a context made by isph_ctx_create_dist or isph_ctx_create_hostcomm");
  const size_t nsend = (size_t)sp[np] * nc, nrecv = (size_t)rp[np] * nc;
  ISPH_CHECK(comm_stage_reserve(ctx, nsend, nrecv));
  std::vector<long long> so(np + 1), ro(np + 1);
  for (size_t p = 0; p <= np; ++p) { so[p] = (long long)sp[p] * ncomp; ro[p] = (long long)rp[p] * ncomp; }
  if (nsend > 0) ISPH_CHECK_HIP(hipMemcpyAsync(ctx->hsend, send, sizeof(double) * nsend, hipMemcpyDeviceToHost, s));
  ISPH_CHECK_HIP(hipStreamSynchronize(s));
  if (ctx->host_tr.exchange(ctx->host_tr.user, H.npeers, H.peer.data(), ctx->hsend, so.data(), ctx->hrecv, ro.data()) != 0)
    return fail("host transport: point-to-point exchange failed", __FILE__, __LINE__);
  if (nrecv > 0) ISPH_CHECK_HIP(hipMemcpyAsync(recv, ctx->hrecv, sizeof(double) * nrecv, hipMemcpyHostToDevice, s));
  ISPH_CHECK_HIP(hipStreamSynchronize(s));
  return ISPH_SUCCESS;
}

}  // namespace isph
