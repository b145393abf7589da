// halo_lists.h -- what the C ABI needs from a matrix' Epetra_Import: the peers, the owned rows to send to each and the
// number of ghost values to receive from each (isph_mat_set_halo, isph_halo_create, isph_prec_create_overlap), and --
// for Ifpack's "Overlap Level" 1 on more than one rank -- the matrix of the rank's subdomain extended by the rows of
// its ghost columns (what Ifpack_OverlappingRowMatrix imports; ref: precond_ifpack.h:43,60-74).
#pragma once
#include <algorithm>
#include <vector>

#include "isph_compat.h"
#include "isph_hip.h"

namespace LAMMPS_NS {

struct HaloLists {
  std::vector<int> peers, send_ptr, send_idx, recv_ptr;
  int npeers() const { return (int)peers.size(); }
};

// peers = ProcsTo U ProcsFrom (ascending); per peer the owned rows to send (ExportLIDs, grouped by destination) and the
// number of ghost values to receive (ghost columns are stored grouped by source rank, Epetra's column-map order)
inline int halo_lists_from_import(const Epetra_CrsMatrix &E, HaloLists &H) {
#ifdef HAVE_EPETRA
  const Epetra_Import *imp = E.Importer();
  if (!imp) return ISPH_FAILURE;
  const Epetra_MpiDistributor *d = dynamic_cast<const Epetra_MpiDistributor *>(&imp->Distributor());
  if (!d) return ISPH_FAILURE;
  const int nto = d->NumSends(), nfrom = d->NumReceives();
  const int *pto = d->ProcsTo(), *lto = d->LengthsTo(), *pfrom = d->ProcsFrom(), *lfrom = d->LengthsFrom();
  const int *exp = imp->ExportLIDs();
#else
  const Epetra_Import *imp = E.Importer();
  if (!imp) {  // a rank without ghost columns has no importer: empty lists (it still takes part in the collectives)
    H.peers.clear(); H.send_idx.clear();
    H.send_ptr.assign(1, 0); H.recv_ptr.assign(1, 0);
    return E.NumMyCols() == E.NumMyRows() ? ISPH_SUCCESS : ISPH_FAILURE;
  }
  const int nto = imp->NumSends(), nfrom = imp->NumReceives();
  const int *pto = imp->ProcsTo(), *lto = imp->LengthsTo(), *pfrom = imp->ProcsFrom(), *lfrom = imp->LengthsFrom();
  const int *exp = imp->ExportLIDs();
#endif
  H.peers.clear();
  for (int k = 0; k < nto; ++k) H.peers.push_back(pto[k]);
  for (int k = 0; k < nfrom; ++k) H.peers.push_back(pfrom[k]);
  std::sort(H.peers.begin(), H.peers.end());
  H.peers.erase(std::unique(H.peers.begin(), H.peers.end()), H.peers.end());
  const int np = H.npeers();
  H.send_ptr.assign((size_t)np + 1, 0);
  H.recv_ptr.assign((size_t)np + 1, 0);
  H.send_idx.clear();
  std::vector<int> exp_off((size_t)nto + 1, 0);
  for (int k = 0; k < nto; ++k) exp_off[(size_t)k + 1] = exp_off[(size_t)k] + lto[k];
  for (int p = 0; p < np; ++p) {
    for (int k = 0; k < nto; ++k)
      if (pto[k] == H.peers[(size_t)p]) H.send_idx.insert(H.send_idx.end(), exp + exp_off[(size_t)k], exp + exp_off[(size_t)k + 1]);
    H.send_ptr[(size_t)p + 1] = (int)H.send_idx.size();
    int nr = 0;
    for (int k = 0; k < nfrom; ++k)
      if (pfrom[k] == H.peers[(size_t)p]) nr += lfrom[k];
    H.recv_ptr[(size_t)p + 1] = H.recv_ptr[(size_t)p] + nr;
  }
  // ghost columns must be grouped by source rank in ascending rank order (ProcsFrom is sorted by Epetra)
  for (int k = 1; k < nfrom; ++k)
    if (pfrom[k] < pfrom[k - 1]) return ISPH_FAILURE;
  return ISPH_SUCCESS;
}

// CSR of the extended subdomain: rows [0, n) = the local rows as they are, row n + g = the row of ghost column g, sent by
// its owner with global column ids and restricted here to the rank's extended column set (owned + ghost columns; a
// global id that is both -- the self-peer importer of the tests, where a ghost is an image of an owned row -- goes to
// the ghost column).  Rows travel as (length, global ids, values) per peer: MPI_Sendrecv under ISPH_HAVE_MPI, a local
// copy when the only peer is this rank.  Returns ISPH_FAILURE when rows of other ranks are needed and MPI is not there.
inline int extend_rows_one_layer(const Epetra_CrsMatrix &E, const Epetra_MpiComm &comm, const HaloLists &H,
                                 std::vector<int> &rp, std::vector<int> &ci, std::vector<double> &v) {
  int *erp = nullptr, *eci = nullptr;
  double *ev = nullptr;
  E.ExtractCrsDataPointers(erp, eci, ev);
  const int n = E.NumMyRows(), ncol = E.NumMyCols(), me = comm.MyPID(), np = H.npeers();
  const int nghost = ncol - n;
  if (H.recv_ptr[(size_t)np] != nghost) return ISPH_FAILURE;
  // global row offsets of every rank
  std::vector<long long> off((size_t)comm.NumProc() + 1, 0);
#ifdef ISPH_HAVE_MPI
  {
    std::vector<int> all((size_t)comm.NumProc());
    MPI_Allgather(&n, 1, MPI_INT, all.data(), 1, MPI_INT, comm.Comm());
    for (int r = 0; r < comm.NumProc(); ++r) off[(size_t)r + 1] = off[(size_t)r] + all[(size_t)r];
  }
#else
  off[1] = n;
#endif
  // outgoing batches per peer: row gid, length, then (gid, value) pairs of the row -- columns in MY numbering need the
  // gids of my ghost columns, which arrive with the incoming batches (their row gids): two phases
  std::vector<long long> ghost_gid((size_t)(nghost > 0 ? nghost : 1), -1);
  std::vector<std::vector<long long>> in_rowgid((size_t)np);
  for (int p = 0; p < np; ++p) {
    const int peer = H.peers[(size_t)p], s0 = H.send_ptr[(size_t)p], s1 = H.send_ptr[(size_t)p + 1];
    const int r0 = H.recv_ptr[(size_t)p], r1 = H.recv_ptr[(size_t)p + 1];
    std::vector<long long> mine((size_t)(s1 - s0));
    for (int k = s0; k < s1; ++k) mine[(size_t)(k - s0)] = off[(size_t)me] + H.send_idx[(size_t)k];
    in_rowgid[(size_t)p].assign((size_t)(r1 - r0), -1);
    if (peer == me) {
      if (s1 - s0 != r1 - r0) return ISPH_FAILURE;
      in_rowgid[(size_t)p] = mine;
    } else {
#ifdef ISPH_HAVE_MPI
      MPI_Sendrecv(mine.data(), s1 - s0, MPI_LONG_LONG, peer, 71, in_rowgid[(size_t)p].data(), r1 - r0, MPI_LONG_LONG, peer, 71,
                   comm.Comm(), MPI_STATUS_IGNORE);
#else
      return ISPH_FAILURE;
#endif
    }
    for (int g = r0; g < r1; ++g) ghost_gid[(size_t)g] = in_rowgid[(size_t)p][(size_t)(g - r0)];
  }
  auto gid_of_col = [&](int c) { return c < n ? off[(size_t)me] + c : ghost_gid[(size_t)(c - n)]; };
  // sorted ghost gids for the look-up (ghost first, then owned)
  std::vector<std::pair<long long, int>> gs((size_t)nghost);
  for (int g = 0; g < nghost; ++g) gs[(size_t)g] = {ghost_gid[(size_t)g], g};
  std::sort(gs.begin(), gs.end());
  auto ext_of_gid = [&](long long gid) -> int {
    auto it = std::lower_bound(gs.begin(), gs.end(), std::make_pair(gid, -1));
    if (it != gs.end() && it->first == gid) return n + it->second;
    if (gid >= off[(size_t)me] && gid < off[(size_t)me] + n) return (int)(gid - off[(size_t)me]);
    return -1;
  };
  rp.assign(erp, erp + n + 1);
  ci.assign(eci, eci + erp[n]);
  v.assign(ev, ev + erp[n]);
  for (int p = 0; p < np; ++p) {
    const int peer = H.peers[(size_t)p], s0 = H.send_ptr[(size_t)p], s1 = H.send_ptr[(size_t)p + 1];
    const int r0 = H.recv_ptr[(size_t)p], r1 = H.recv_ptr[(size_t)p + 1];
    std::vector<int> out_len((size_t)(s1 - s0)), in_len((size_t)(r1 - r0));
    std::vector<long long> out_gid, in_gid;
    std::vector<double> out_val, in_val;
    for (int k = s0; k < s1; ++k) {
      const int row = H.send_idx[(size_t)k];
      out_len[(size_t)(k - s0)] = erp[row + 1] - erp[row];
      for (int q = erp[row]; q < erp[row + 1]; ++q) { out_gid.push_back(gid_of_col(eci[q])); out_val.push_back(ev[q]); }
    }
    if (peer == me) {
      in_len = out_len; in_gid = out_gid; in_val = out_val;
    } else {
#ifdef ISPH_HAVE_MPI
      MPI_Sendrecv(out_len.data(), s1 - s0, MPI_INT, peer, 72, in_len.data(), r1 - r0, MPI_INT, peer, 72, comm.Comm(), MPI_STATUS_IGNORE);
      long long tin = 0;
      for (int x : in_len) tin += x;
      in_gid.resize((size_t)tin); in_val.resize((size_t)tin);
      MPI_Sendrecv(out_gid.data(), (int)out_gid.size(), MPI_LONG_LONG, peer, 73, in_gid.data(), (int)tin, MPI_LONG_LONG, peer, 73,
                   comm.Comm(), MPI_STATUS_IGNORE);
      MPI_Sendrecv(out_val.data(), (int)out_val.size(), MPI_DOUBLE, peer, 74, in_val.data(), (int)tin, MPI_DOUBLE, peer, 74,
                   comm.Comm(), MPI_STATUS_IGNORE);
#else
      return ISPH_FAILURE;
#endif
    }
    size_t pos = 0;
    std::vector<std::pair<int, double>> row;
    for (int g = 0; g < r1 - r0; ++g) {
      row.clear();
      for (int q = 0; q < in_len[(size_t)g]; ++q, ++pos) {
        const int e = ext_of_gid(in_gid[pos]);
        if (e >= 0) row.push_back({e, in_val[pos]});
      }
      std::stable_sort(row.begin(), row.end(), [](const std::pair<int, double> &a, const std::pair<int, double> &b) { return a.first < b.first; });
      for (auto &e : row) { ci.push_back(e.first); v.push_back(e.second); }
      rp.push_back((int)ci.size());
    }
  }
  return ISPH_SUCCESS;
}

}  // namespace LAMMPS_NS
