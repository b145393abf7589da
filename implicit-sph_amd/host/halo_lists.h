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

#ifdef ISPH_HAVE_MPI
namespace LAMMPS_NS {

// "Overlap Level" L on more than one rank (precond_ifpack.h:43; Ifpack_OverlappingRowMatrix with OverlapLevel = L): the
// rank's rows, then L layers of imported rows -- layer 1 = the rows of the matrix' ghost columns, layer l + 1 = the rows of
// the columns the layer-l rows reference outside everything gathered so far -- each layer in ascending global row number
// (rank-concatenated numbering), entries that leave the extended set dropped.  Rows of a layer > 1 may belong to ranks the
// matrix' own importer does not talk to, so every layer is one MPI_Alltoall(v) round over the communicator: wanted local
// indices out, (length, global column ids, values) back.  XH receives the halo lists of the imported rows in the form
// isph_prec_create_overlap takes: one (peer, send range, receive range) triple per layer and owner -- a rank can appear
// once per layer -- with the receive ranges in extended row order.  Same result as dist.extend_rows_levels (the Python
// plumbing), which tests/test_dist_cpu.py checks entry for entry.
inline int extend_rows_levels(const Epetra_CrsMatrix &E, const Epetra_MpiComm &comm, const HaloLists &H, int levels,
                              std::vector<int> &rp, std::vector<int> &ci, std::vector<double> &v, HaloLists &XH) {
  int *erp = nullptr, *eci = nullptr;
  double *ev = nullptr;
  E.ExtractCrsDataPointers(erp, eci, ev);
  const int n = E.NumMyRows(), ncol = E.NumMyCols(), me = comm.MyPID(), P = comm.NumProc(), np0 = H.npeers();
  const int nghost = ncol - n;
  if (H.recv_ptr[(size_t)np0] != nghost) return ISPH_FAILURE;
  MPI_Comm mc = comm.Comm();
  std::vector<long long> off((size_t)P + 1, 0);
  {
    std::vector<int> all((size_t)P);
    MPI_Allgather(&n, 1, MPI_INT, all.data(), 1, MPI_INT, mc);
    for (int r = 0; r < P; ++r) off[(size_t)r + 1] = off[(size_t)r] + all[(size_t)r];
  }
  // global ids of my ghost columns: the owners' local indices arrive over the matrix' own lists
  std::vector<long long> gcol((size_t)ncol);
  for (int c = 0; c < n; ++c) gcol[(size_t)c] = off[(size_t)me] + c;
  {
    std::vector<MPI_Request> rq;
    std::vector<std::vector<long long>> mine((size_t)np0);
    for (int p = 0; p < np0; ++p) {
      const int r0 = H.recv_ptr[(size_t)p], r1 = H.recv_ptr[(size_t)p + 1];
      if (r1 > r0) { rq.push_back(MPI_REQUEST_NULL); MPI_Irecv(gcol.data() + n + r0, r1 - r0, MPI_LONG_LONG, H.peers[(size_t)p], 81, mc, &rq.back()); }
    }
    for (int p = 0; p < np0; ++p) {
      const int s0 = H.send_ptr[(size_t)p], s1 = H.send_ptr[(size_t)p + 1];
      mine[(size_t)p].resize((size_t)(s1 - s0));
      for (int k = s0; k < s1; ++k) mine[(size_t)p][(size_t)(k - s0)] = off[(size_t)me] + H.send_idx[(size_t)k];
      if (s1 > s0) { rq.push_back(MPI_REQUEST_NULL); MPI_Isend(mine[(size_t)p].data(), s1 - s0, MPI_LONG_LONG, H.peers[(size_t)p], 81, mc, &rq.back()); }
    }
    if (!rq.empty()) MPI_Waitall((int)rq.size(), rq.data(), MPI_STATUSES_IGNORE);
  }
  auto owner_of = [&](long long g) { return (int)(std::upper_bound(off.begin(), off.end(), g) - off.begin()) - 1; };
  std::vector<long long> have;  // sorted global ids of the imported rows gathered so far (the owned ones are a range)
  auto has = [&](long long g) {
    return (g >= off[(size_t)me] && g < off[(size_t)me] + n) || std::binary_search(have.begin(), have.end(), g);
  };
  std::vector<long long> ext_gid;
  std::vector<int> ext_len;
  std::vector<long long> ext_cols;
  std::vector<double> ext_vals;
  XH.peers.clear(); XH.send_idx.clear();
  XH.send_ptr.assign(1, 0); XH.recv_ptr.assign(1, 0);
  std::vector<long long> frontier;
  for (int q = 0; q < erp[n]; ++q) frontier.push_back(gcol[(size_t)eci[q]]);
  for (int level = 0; level < levels; ++level) {
    std::sort(frontier.begin(), frontier.end());
    frontier.erase(std::unique(frontier.begin(), frontier.end()), frontier.end());
    std::vector<std::vector<int>> want((size_t)P);
    for (long long g : frontier)
      if (!has(g)) { const int r = owner_of(g); want[(size_t)r].push_back((int)(g - off[(size_t)r])); }
    // round 1: wanted local indices to their owners
    std::vector<int> scnt((size_t)P), rcnt((size_t)P), sdis((size_t)P + 1, 0), rdis((size_t)P + 1, 0);
    for (int r = 0; r < P; ++r) scnt[(size_t)r] = (int)want[(size_t)r].size();
    MPI_Alltoall(scnt.data(), 1, MPI_INT, rcnt.data(), 1, MPI_INT, mc);
    for (int r = 0; r < P; ++r) { sdis[(size_t)r + 1] = sdis[(size_t)r] + scnt[(size_t)r]; rdis[(size_t)r + 1] = rdis[(size_t)r] + rcnt[(size_t)r]; }
    std::vector<int> sbuf((size_t)sdis[(size_t)P] + 1), asked((size_t)rdis[(size_t)P] + 1);
    for (int r = 0; r < P; ++r) std::copy(want[(size_t)r].begin(), want[(size_t)r].end(), sbuf.begin() + sdis[(size_t)r]);
    MPI_Alltoallv(sbuf.data(), scnt.data(), sdis.data(), MPI_INT, asked.data(), rcnt.data(), rdis.data(), MPI_INT, mc);
    // round 2: the rows back -- lengths, then global column ids and values
    std::vector<int> olen((size_t)rdis[(size_t)P] + 1), ocnt((size_t)P, 0), odis((size_t)P + 1, 0);
    std::vector<long long> ogid;
    std::vector<double> oval;
    for (int r = 0; r < P; ++r) {
      for (int k = rdis[(size_t)r]; k < rdis[(size_t)r + 1]; ++k) {
        const int row = asked[(size_t)k];
        if (row < 0 || row >= n) return ISPH_FAILURE;
        olen[(size_t)k] = erp[row + 1] - erp[row];
        for (int q = erp[row]; q < erp[row + 1]; ++q) { ogid.push_back(gcol[(size_t)eci[q]]); oval.push_back(ev[q]); }
        ocnt[(size_t)r] += olen[(size_t)k];
      }
      odis[(size_t)r + 1] = odis[(size_t)r] + ocnt[(size_t)r];
    }
    std::vector<int> ilen((size_t)sdis[(size_t)P] + 1), icnt((size_t)P, 0), idis((size_t)P + 1, 0);
    MPI_Alltoallv(olen.data(), rcnt.data(), rdis.data(), MPI_INT, ilen.data(), scnt.data(), sdis.data(), MPI_INT, mc);
    for (int r = 0; r < P; ++r) {
      for (int k = sdis[(size_t)r]; k < sdis[(size_t)r + 1]; ++k) icnt[(size_t)r] += ilen[(size_t)k];
      idis[(size_t)r + 1] = idis[(size_t)r] + icnt[(size_t)r];
    }
    std::vector<long long> igid((size_t)idis[(size_t)P] + 1);
    std::vector<double> ival((size_t)idis[(size_t)P] + 1);
    ogid.push_back(0); oval.push_back(0.0);  // never empty buffers
    MPI_Alltoallv(ogid.data(), ocnt.data(), odis.data(), MPI_LONG_LONG, igid.data(), icnt.data(), idis.data(), MPI_LONG_LONG, mc);
    MPI_Alltoallv(oval.data(), ocnt.data(), odis.data(), MPI_DOUBLE, ival.data(), icnt.data(), idis.data(), MPI_DOUBLE, mc);
    // one triple per rank that owns rows of this layer or asked for some of mine, ascending rank = ascending global id
    frontier.clear();
    for (int r = 0; r < P; ++r) {
      if (scnt[(size_t)r] == 0 && rcnt[(size_t)r] == 0) continue;
      XH.peers.push_back(r);
      for (int k = rdis[(size_t)r]; k < rdis[(size_t)r + 1]; ++k) XH.send_idx.push_back(asked[(size_t)k]);
      XH.send_ptr.push_back((int)XH.send_idx.size());
      XH.recv_ptr.push_back(XH.recv_ptr.back() + scnt[(size_t)r]);
      size_t pos = (size_t)idis[(size_t)r];
      for (int k = sdis[(size_t)r]; k < sdis[(size_t)r + 1]; ++k) {
        ext_gid.push_back(off[(size_t)r] + sbuf[(size_t)k]);
        ext_len.push_back(ilen[(size_t)k]);
        for (int q = 0; q < ilen[(size_t)k]; ++q, ++pos) { ext_cols.push_back(igid[pos]); ext_vals.push_back(ival[pos]); frontier.push_back(igid[pos]); }
      }
    }
    have.assign(ext_gid.begin(), ext_gid.end());
    std::sort(have.begin(), have.end());
  }
  // local index of a global id inside the extended set
  std::vector<std::pair<long long, int>> gs(ext_gid.size());
  for (size_t k = 0; k < ext_gid.size(); ++k) gs[k] = {ext_gid[k], (int)k};
  std::sort(gs.begin(), gs.end());
  auto ext_of = [&](long long g) -> int {
    if (g >= off[(size_t)me] && g < off[(size_t)me] + n) return (int)(g - off[(size_t)me]);
    auto it = std::lower_bound(gs.begin(), gs.end(), std::make_pair(g, -1));
    return (it != gs.end() && it->first == g) ? n + it->second : -1;
  };
  rp.assign(1, 0); ci.clear(); v.clear();
  std::vector<std::pair<int, double>> row;
  auto emit = [&](const long long *g, const double *a, int len) {
    row.clear();
    for (int q = 0; q < len; ++q) { const int e = ext_of(g[q]); if (e >= 0) row.push_back({e, a[q]}); }
    std::stable_sort(row.begin(), row.end(), [](const std::pair<int, double> &x, const std::pair<int, double> &y) { return x.first < y.first; });
    for (auto &e : row) { ci.push_back(e.first); v.push_back(e.second); }
    rp.push_back((int)ci.size());
  };
  std::vector<long long> tmpg;
  for (int i = 0; i < n; ++i) {
    tmpg.clear();
    for (int q = erp[i]; q < erp[i + 1]; ++q) tmpg.push_back(gcol[(size_t)eci[q]]);
    emit(tmpg.data(), ev + erp[i], erp[i + 1] - erp[i]);
  }
  size_t pos = 0;
  for (size_t k = 0; k < ext_gid.size(); ++k) { emit(ext_cols.data() + pos, ext_vals.data() + pos, ext_len[k]); pos += (size_t)ext_len[k]; }
  return ISPH_SUCCESS;
}

}  // namespace LAMMPS_NS
#endif  // ISPH_HAVE_MPI
