// Host-only parts of the multi-rank C++ mirror under real MPI (no device is touched): the row import of Ifpack's
// "Overlap Level" 1 (host/halo_lists.h: MPI_Allgather + MPI_Sendrecv, ref: precond_ifpack.h:43,60-74) and
// SolverLin::createNullVector with its GLOBAL norm (ref: solver_lin.cpp:59-77).  Run as
//     mpiexec -n 2 test_mpi_host <dir>
// Rank r reads <dir>/rank<r>.bin = n, ncol, nnz | rp, ci, val | nto, procs_to, lengths_to | nexp, export_lids |
// nfrom, procs_from, lengths_from | mask[n]   and writes <dir>/ext<r>.bin (next, nnz, rp, ci, val) and <dir>/nv<r>.bin.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "halo_lists.h"
#include "solver_lin.h"

using namespace LAMMPS_NS;

template <class T>
static bool rd(FILE *f, std::vector<T> &v, size_t n) {
  v.resize(n);
  return n == 0 || std::fread(v.data(), sizeof(T), n, f) == n;
}

int main(int argc, char **argv) {
  MPI_Init(&argc, &argv);
  MPI_Comm world = MPI_COMM_WORLD;
  Epetra_MpiComm comm(world);
  int rc = 0;
  {
    const std::string dir = argc > 1 ? argv[1] : ".";
    FILE *f = std::fopen((dir + "/rank" + std::to_string(comm.MyPID()) + ".bin").c_str(), "rb");
    int hdr[3] = {0, 0, 0}, nto = 0, nexp = 0, nfrom = 0;
    std::vector<int> rp, ci, pto, lto, exp, pfrom, lfrom, mask, gid;
    std::vector<double> val;
    bool ok = f && std::fread(hdr, 4, 3, f) == 3 && rd(f, rp, (size_t)hdr[0] + 1) && rd(f, ci, (size_t)hdr[2]) && rd(f, val, (size_t)hdr[2]) &&
              std::fread(&nto, 4, 1, f) == 1 && rd(f, pto, (size_t)nto) && rd(f, lto, (size_t)nto) && std::fread(&nexp, 4, 1, f) == 1 &&
              rd(f, exp, (size_t)nexp) && std::fread(&nfrom, 4, 1, f) == 1 && rd(f, pfrom, (size_t)nfrom) && rd(f, lfrom, (size_t)nfrom) &&
              rd(f, mask, (size_t)hdr[0]);
    if (f) std::fclose(f);
    if (!ok) { std::fprintf(stderr, "rank %d: cannot read its input\n", comm.MyPID()); MPI_Abort(world, 2); }
    const int n = hdr[0], ncol = hdr[1];
    // a rank without ghost columns carries no importer, like a filled Epetra matrix whose column map is its row map
    Epetra_Import importer(nto, pto.data(), lto.data(), exp.data(), nfrom, pfrom.data(), lfrom.data());
    Epetra_CrsMatrix A(n, ncol, rp.data(), ci.data(), val.data(), ncol > n ? &importer : nullptr);
    HaloLists H;
    std::vector<int> erp, eci;
    std::vector<double> ev;
    int bad = halo_lists_from_import(A, H) != ISPH_SUCCESS ? 1 : 0;
    if (comm.MaxAll(bad) == 0) bad = extend_rows_one_layer(A, comm, H, erp, eci, ev) != ISPH_SUCCESS ? 1 : 0;
    if (comm.MaxAll(bad)) rc = 1;
    if (!rc) {
      const int next = (int)erp.size() - 1, ennz = (int)eci.size();
      f = std::fopen((dir + "/ext" + std::to_string(comm.MyPID()) + ".bin").c_str(), "wb");
      std::fwrite(&next, 4, 1, f); std::fwrite(&ennz, 4, 1, f);
      std::fwrite(erp.data(), 4, erp.size(), f); std::fwrite(eci.data(), 4, eci.size(), f); std::fwrite(ev.data(), 8, ev.size(), f);
      std::fclose(f);
    }
    // argv[2] = L: "Overlap Level" L through the all-to-all rounds of extend_rows_levels -> <dir>/extL<r>.bin =
    // next, nnz, rp, ci, val | ntriples, peers, send_ptr, nsend, send_idx, recv_ptr
    const int levels = argc > 2 ? std::atoi(argv[2]) : 0;
    if (levels > 0 && !rc) {
      HaloLists XH;
      std::vector<int> lrp, lci;
      std::vector<double> lv;
      int badl = extend_rows_levels(A, comm, H, levels, lrp, lci, lv, XH) != ISPH_SUCCESS ? 1 : 0;
      if (comm.MaxAll(badl)) rc = 1;
      if (!rc) {
        const int next = (int)lrp.size() - 1, ennz = (int)lci.size(), nt = XH.npeers(), ns = (int)XH.send_idx.size();
        f = std::fopen((dir + "/extL" + std::to_string(comm.MyPID()) + ".bin").c_str(), "wb");
        std::fwrite(&next, 4, 1, f); std::fwrite(&ennz, 4, 1, f);
        std::fwrite(lrp.data(), 4, lrp.size(), f); std::fwrite(lci.data(), 4, lci.size(), f); std::fwrite(lv.data(), 8, lv.size(), f);
        std::fwrite(&nt, 4, 1, f); std::fwrite(XH.peers.data(), 4, (size_t)nt, f); std::fwrite(XH.send_ptr.data(), 4, (size_t)nt + 1, f);
        std::fwrite(&ns, 4, 1, f); std::fwrite(XH.send_idx.data(), 4, (size_t)ns, f); std::fwrite(XH.recv_ptr.data(), 4, (size_t)nt + 1, f);
        std::fclose(f);
      }
    }
    // SolverLin::createNullVector over the communicator, with a mask of unequal local counts
    gid.resize((size_t)n);
    for (int i = 0; i < n; ++i) gid[(size_t)i] = i + 1;
    SolverLin solver(world);
    solver.createNodalMap(n, gid.data());
    Epetra_IntSerialDenseVector m(View, mask.data(), n);
    solver.setNullVectorMask(&m);
    solver.createNullVector();
    f = std::fopen((dir + "/nv" + std::to_string(comm.MyPID()) + ".bin").c_str(), "wb");
    std::fwrite(solver.getNullVector()->Values(), 8, (size_t)n, f);
    std::fclose(f);
    std::printf("rank %d of %d: n %d ncol %d peers %d extended rows %d\n", comm.MyPID(), comm.NumProc(), n, ncol, H.npeers(), (int)erp.size() - 1);
  }
  MPI_Finalize();
  return rc;
}
