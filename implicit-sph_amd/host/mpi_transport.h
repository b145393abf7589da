// mpi_transport.h -- the two callbacks of isph_host_transport (include/isph_hip.h) over the caller's MPI communicator:
// what Epetra_MpiComm / Epetra_MpiDistributor do for the reference (ref: solver_lin.cpp:30-31, the Import inside
// Epetra_CrsMatrix::Apply solver_lin.h:133, Dot/Norm2 all-reduces).  The library stages device buffers through pinned
// host memory around these calls (csrc/comm.hpp).  Used when ranks share a device -- RCCL cannot put two ranks of one
// communicator on one GPU -- which is how LAMMPS is commonly run (several MPI ranks per GPU) and how the multi-rank
// tests run on a one-GPU box.  One GPU per rank keeps RCCL over xGMI (SolverLin_HIP::ensureContext picks).
#pragma once
#ifdef ISPH_HAVE_MPI
#include <mpi.h>

#include <vector>

#include "isph_hip.h"

namespace LAMMPS_NS {

struct MpiTransport {
  MPI_Comm comm;
  std::vector<MPI_Request> req;
  isph_host_transport table() {
    isph_host_transport t;
    t.user = this;
    t.exchange = &MpiTransport::exchange;
    t.allreduce = &MpiTransport::allreduce;
    return t;
  }
  // all receives, then all sends, then one wait: no ordering between the peers, a peer may be this rank
  static int exchange(void *user, int npeers, const int *peer, const double *send, const long long *send_off, double *recv,
                      const long long *recv_off) {
    MpiTransport *T = static_cast<MpiTransport *>(user);
    T->req.clear();
    const int tag = 4711;
    for (int p = 0; p < npeers; ++p) {
      const long long n = recv_off[p + 1] - recv_off[p];
      if (n <= 0) continue;
      if (n > 0x7fffffffLL) return 1;
      T->req.push_back(MPI_REQUEST_NULL);
      if (MPI_Irecv(recv + recv_off[p], (int)n, MPI_DOUBLE, peer[p], tag, T->comm, &T->req.back()) != MPI_SUCCESS) return 1;
    }
    for (int p = 0; p < npeers; ++p) {
      const long long n = send_off[p + 1] - send_off[p];
      if (n <= 0) continue;
      if (n > 0x7fffffffLL) return 1;
      T->req.push_back(MPI_REQUEST_NULL);
      if (MPI_Isend(send + send_off[p], (int)n, MPI_DOUBLE, peer[p], tag, T->comm, &T->req.back()) != MPI_SUCCESS) return 1;
    }
    if (!T->req.empty() && MPI_Waitall((int)T->req.size(), T->req.data(), MPI_STATUSES_IGNORE) != MPI_SUCCESS) return 1;
    return 0;
  }
  static int allreduce(void *user, double *buf, int count, int op) {
    MpiTransport *T = static_cast<MpiTransport *>(user);
    return MPI_Allreduce(MPI_IN_PLACE, buf, count, MPI_DOUBLE, op == 1 ? MPI_MAX : MPI_SUM, T->comm) == MPI_SUCCESS ? 0 : 1;
  }
};

}  // namespace LAMMPS_NS
#endif  // ISPH_HAVE_MPI
