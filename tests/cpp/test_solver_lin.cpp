// Drives the C++ mirror exactly like the reference's second consumer does
// (ref: USER-REAXC-T/fix_qeq_reax.cpp:671-693): SolverLin_Belos li_solver(world);
// setParameters(); setNodalMap; setMatrix; prec.setMatrix; create*MultiVector;
// solveProblem(&prec, "...").  Reads a CSR system from a binary file, writes x.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "precond_ifpack.h"
#include "precond_ml.h"
#include "solver_lin_hip.h"

using namespace LAMMPS_NS;

#ifdef ISPH_HAVE_MPI
static const MPI_Comm kWorld = MPI_COMM_WORLD;  // built with -DISPH_HAVE_MPI: the real communicator (one rank under mpiexec -n 1 or singleton init)
#else
static const MPI_Comm kWorld = 0;
#endif

// "block": in.bin = n, nnz, H (rp, ci, val) | dim | for each block (i,j) row-major: flag [, nnz, rp, ci, val] |
// b [dim][n].  Drives createBlockMatrix / setBlock / solveBlockProblem exactly like pair_isph.cpp:944-972, with
// PrecondWrapper_ML built from H (prec->setMatrix(A.crs), pair_isph.cpp:926).
static int run_block(const char *fin, const char *fout) {
  FILE *f = std::fopen(fin, "rb");
  if (!f) return 2;
  int n = 0, nnz = 0, dim = 0;
  if (std::fread(&n, 4, 1, f) != 1 || std::fread(&nnz, 4, 1, f) != 1) return 2;
  std::vector<int> rp((size_t)n + 1), ci((size_t)nnz), gid((size_t)n);
  std::vector<double> val((size_t)nnz);
  if (std::fread(rp.data(), 4, rp.size(), f) != rp.size() || std::fread(ci.data(), 4, ci.size(), f) != ci.size() ||
      std::fread(val.data(), 8, val.size(), f) != val.size() || std::fread(&dim, 4, 1, f) != 1) return 2;
  std::vector<std::vector<int> > brp(9), bci(9);
  std::vector<std::vector<double> > bval(9);
  std::vector<Epetra_CrsMatrix *> blk(9, (Epetra_CrsMatrix *)NULL);
  for (int k = 0; k < dim * dim; ++k) {
    int flag = 0, bn = 0;
    if (std::fread(&flag, 4, 1, f) != 1) return 2;
    if (!flag) continue;
    if (std::fread(&bn, 4, 1, f) != 1) return 2;
    brp[k].resize((size_t)n + 1); bci[k].resize((size_t)bn); bval[k].resize((size_t)bn);
    if (std::fread(brp[k].data(), 4, brp[k].size(), f) != brp[k].size() ||
        std::fread(bci[k].data(), 4, bci[k].size(), f) != bci[k].size() ||
        std::fread(bval[k].data(), 8, bval[k].size(), f) != bval[k].size()) return 2;
    blk[k] = new Epetra_CrsMatrix(n, n, brp[k].data(), bci[k].data(), bval[k].data());
  }
  std::vector<double> b((size_t)n * dim), x((size_t)n * dim, 0.0);
  if (std::fread(b.data(), 8, b.size(), f) != b.size()) return 2;
  std::fclose(f);
  for (int i = 0; i < n; ++i) gid[(size_t)i] = i + 1;
  MPI_Comm world = kWorld;
  Epetra_Map nodalmap(-1, n, gid.data(), 1, Epetra_MpiComm(world));
  Epetra_CrsMatrix H(n, n, rp.data(), ci.data(), val.data());
  PrecondWrapper_ML prec(world);
  Teuchos::ParameterList *pp = prec.setParameters();
  pp->set("coarse: max size", 64);
  pp->set("aggregation: threshold", 0.02);
  pp->set("isph: block rows", 256);
  SolverLin_Belos li_solver(world);
  li_solver.setParameters();
  li_solver.setNodalMap(&nodalmap);
  li_solver.setMatrix(&H);
  prec.setMatrix(&H);
  li_solver.createBlockMatrix(dim, "Block Helmholtz");
  li_solver.createSolutionMultiVector(x.data(), n, dim);
  li_solver.createLoadMultiVector(b.data(), n, dim);
  li_solver.setBlockBegin();
  for (int k2 = 0; k2 < dim; ++k2)
    for (int k1 = 0; k1 < dim; ++k1) li_solver.setBlock(k1, k2, blk[(size_t)(k1 * dim + k2)]);
  li_solver.setBlockEnd();
  li_solver.setMatrixIsBlocked(true);
  const int rc = li_solver.solveBlockProblem(&prec, "Block 3x3 Helmholtz");
  li_solver.freeBlockMatrix();
  if (rc != LAMMPS_SUCCESS) return 1;
  const isph_solve_info &info = li_solver.lastSolveInfo();
  std::printf("converged=%d iters=%d rel=%.3e\n", info.converged, info.iters, info.rel_res_implicit);
  f = std::fopen(fout, "wb");
  std::fwrite(x.data(), 8, x.size(), f);
  std::fclose(f);
  for (size_t k = 0; k < blk.size(); ++k) delete blk[k];
  return info.converged ? 0 : 3;
}

// "selfhalo" / "selfhalo-overlap": in.bin = n, ncol, nnz, rp, ci (ghost columns >= n), val, b[n], nsend, send_idx[nsend].  The matrix carries
// an Epetra_Import whose only peer is this rank (periodic images routed through the halo plan): drives
// isph_ctx_create_dist + isph_mat_set_halo from the C++ surface, i.e. the N > 1 code path of SolverLin(MPI_Comm&).
static int run_selfhalo(const char *fin, const char *fout, bool overlap, bool extend_only = false) {
  FILE *f = std::fopen(fin, "rb");
  if (!f) return 2;
  int n = 0, ncol = 0, nnz = 0, nsend = 0;
  if (std::fread(&n, 4, 1, f) != 1 || std::fread(&ncol, 4, 1, f) != 1 || std::fread(&nnz, 4, 1, f) != 1) return 2;
  std::vector<int> rp((size_t)n + 1), ci((size_t)nnz), gid((size_t)n);
  std::vector<double> val((size_t)nnz), b((size_t)n), x((size_t)n, 0.0);
  if (std::fread(rp.data(), 4, rp.size(), f) != rp.size() || std::fread(ci.data(), 4, ci.size(), f) != ci.size() ||
      std::fread(val.data(), 8, val.size(), f) != val.size() || std::fread(b.data(), 8, b.size(), f) != b.size() ||
      std::fread(&nsend, 4, 1, f) != 1) return 2;
  std::vector<int> send_idx((size_t)nsend);
  if (std::fread(send_idx.data(), 4, send_idx.size(), f) != send_idx.size()) return 2;
  std::fclose(f);
  for (int i = 0; i < n; ++i) gid[(size_t)i] = i + 1;
  MPI_Comm world = kWorld;
  Epetra_Map nodalmap(-1, n, gid.data(), 1, Epetra_MpiComm(world));
  const int me = 0, nrecv = ncol - n;
  Epetra_Import importer(1, &me, &nsend, send_idx.data(), 1, &me, &nrecv);
  Epetra_CrsMatrix AA(n, ncol, rp.data(), ci.data(), val.data(), &importer);
  if (extend_only) {   // no device: out.bin = next, nnz, rp, ci, val of the overlapped subdomain (host/halo_lists.h)
    HaloLists H;
    std::vector<int> erp, eci;
    std::vector<double> ev;
    if (halo_lists_from_import(AA, H) != ISPH_SUCCESS || extend_rows_one_layer(AA, Epetra_MpiComm(world), H, erp, eci, ev) != ISPH_SUCCESS) return 1;
    const int next = (int)erp.size() - 1, ennz = (int)eci.size();
    f = std::fopen(fout, "wb");
    std::fwrite(&next, 4, 1, f); std::fwrite(&ennz, 4, 1, f);
    std::fwrite(erp.data(), 4, erp.size(), f); std::fwrite(eci.data(), 4, eci.size(), f); std::fwrite(ev.data(), 8, ev.size(), f);
    std::fclose(f);
    return 0;
  }
  PrecondWrapper_Ifpack prec(world);
  Teuchos::ParameterList *pp = prec.setParameters();
  pp->set("fact: level-of-fill", 0);
  // "selfhalo-overlap": the reference's decomposition -- one subdomain per rank, Overlap Level 1: the rows of the ghost
  // columns come through the matrix' importer (host/halo_lists.h) and isph_prec_create_overlap factors the extension
  pp->set("Overlap Level", overlap ? 1 : 0);
  pp->set("isph: block rows", overlap ? 0 : 256);
  SolverLin_Belos li_solver(world);
  li_solver.setParameters();
  li_solver.setNodalMap(&nodalmap);
  li_solver.setMatrix(&AA);
  prec.setMatrix(&AA);
  li_solver.createSolutionMultiVector(x.data(), n, 1);
  li_solver.createLoadMultiVector(b.data(), n, 1);
  Epetra_IntSerialDenseVector null_mask(n);
  for (int i = 0; i < n; ++i) null_mask[i] = 1;
  li_solver.setNullVectorMask(&null_mask);
  li_solver.setMatrixIsSingular(true);
  li_solver.setInitialSolution(SolverLin::Zero);
  if (li_solver.solveProblem(&prec, "self-halo") != LAMMPS_SUCCESS) return 1;
  const isph_solve_info &info = li_solver.lastSolveInfo();
  std::printf("converged=%d iters=%d rel=%.3e\n", info.converged, info.iters, info.rel_res_implicit);
  f = std::fopen(fout, "wb");
  std::fwrite(x.data(), 8, x.size(), f);
  std::fclose(f);
  return info.converged ? 0 : 3;
}

#ifdef ISPH_HAVE_MPI
// "ranks": mpiexec -n N <exe> <dir> <unused> <singular> ranks <bjacobi|overlap|ml>.  Rank r reads <dir>/rank<r>.bin =
// nl, ncol, nnz, rp, ci (ghost columns >= nl), val, b[nl] | nto, procs_to, lengths_to | nexp, export_lids | nfrom,
// procs_from, lengths_from -- the local matrix of one rank of a decomposed run with the lists of its Epetra_Import --
// and drives SolverLin(MPI_Comm&) exactly like PairISPH does on every rank (pair_isph.cpp:988-1011).  All ranks of the
// test share device 0, so SolverLin_HIP picks the MPI transport (host/mpi_transport.h).  Writes <dir>/x<r>.bin.
static int run_ranks(const char *dir, bool singular, const std::string &mode) {
  int me = 0, np = 1;
  MPI_Comm_rank(MPI_COMM_WORLD, &me);
  MPI_Comm_size(MPI_COMM_WORLD, &np);
  const std::string fin = std::string(dir) + "/rank" + std::to_string(me) + ".bin";
  FILE *f = std::fopen(fin.c_str(), "rb");
  if (!f) return 2;
  int nl = 0, ncol = 0, nnz = 0;
  if (std::fread(&nl, 4, 1, f) != 1 || std::fread(&ncol, 4, 1, f) != 1 || std::fread(&nnz, 4, 1, f) != 1) return 2;
  std::vector<int> rp((size_t)nl + 1), ci((size_t)nnz), gid((size_t)nl);
  std::vector<double> val((size_t)nnz), b((size_t)nl), x((size_t)nl, 0.0);
  if (std::fread(rp.data(), 4, rp.size(), f) != rp.size() || std::fread(ci.data(), 4, ci.size(), f) != ci.size() ||
      std::fread(val.data(), 8, val.size(), f) != val.size() || std::fread(b.data(), 8, b.size(), f) != b.size()) return 2;
  auto read_list = [&](std::vector<int> &v) {
    int n = 0;
    if (std::fread(&n, 4, 1, f) != 1) return false;
    v.resize((size_t)n);
    return n == 0 || std::fread(v.data(), 4, v.size(), f) == v.size();
  };
  std::vector<int> pto, lto, exp, pfrom, lfrom;
  int nto = 0, nfrom = 0;
  if (std::fread(&nto, 4, 1, f) != 1) return 2;
  pto.resize((size_t)nto); lto.resize((size_t)nto);
  if (nto && (std::fread(pto.data(), 4, pto.size(), f) != pto.size() || std::fread(lto.data(), 4, lto.size(), f) != lto.size())) return 2;
  if (!read_list(exp)) return 2;
  if (std::fread(&nfrom, 4, 1, f) != 1) return 2;
  pfrom.resize((size_t)nfrom); lfrom.resize((size_t)nfrom);
  if (nfrom && (std::fread(pfrom.data(), 4, pfrom.size(), f) != pfrom.size() || std::fread(lfrom.data(), 4, lfrom.size(), f) != lfrom.size())) return 2;
  std::fclose(f);
  int off = 0;
  MPI_Exscan(&nl, &off, 1, MPI_INT, MPI_SUM, MPI_COMM_WORLD);
  if (me == 0) off = 0;
  for (int i = 0; i < nl; ++i) gid[(size_t)i] = off + i + 1;
  MPI_Comm world = MPI_COMM_WORLD;
  Epetra_Map nodalmap(-1, nl, gid.data(), 1, Epetra_MpiComm(world));
  Epetra_Import importer(nto, pto.data(), lto.data(), exp.data(), nfrom, pfrom.data(), lfrom.data());
  Epetra_CrsMatrix AA(nl, ncol, rp.data(), ci.data(), val.data(), ncol > nl ? &importer : nullptr);
  PrecondWrapper_Ifpack prec_ifpack(world);
  PrecondWrapper_ML prec_ml(world);
  PrecondWrapper &prec = mode == "ml" ? static_cast<PrecondWrapper &>(prec_ml) : static_cast<PrecondWrapper &>(prec_ifpack);
  Teuchos::ParameterList *pp = prec.setParameters();
  if (mode == "ml") {
    pp->set("coarse: max size", 64);
    pp->set("aggregation: threshold", 0.02);
    pp->set("isph: block rows", 256);
  } else {
    pp->set("fact: level-of-fill", 0);
    const bool ov = mode.rfind("overlap", 0) == 0;      // "overlap" = level 1, "overlap2" / "overlap3" = more layers
    pp->set("Overlap Level", ov ? (mode.size() > 7 ? std::atoi(mode.c_str() + 7) : 1) : 0);
    pp->set("isph: block rows", ov ? 0 : 256);
  }
  SolverLin_Belos li_solver(world);
  li_solver.setParameters();
  li_solver.setNodalMap(&nodalmap);
  li_solver.setMatrix(&AA);
  prec.setMatrix(&AA);
  li_solver.createSolutionMultiVector(x.data(), nl, 1);
  li_solver.createLoadMultiVector(b.data(), nl, 1);
  Epetra_IntSerialDenseVector null_mask(nl);
  if (singular) {
    for (int i = 0; i < nl; ++i) null_mask[i] = 1;
    li_solver.setNullVectorMask(&null_mask);
    li_solver.setMatrixIsSingular(true);
  }
  li_solver.setInitialSolution(SolverLin::Zero);
  if (li_solver.solveProblem(&prec, "ranks") != LAMMPS_SUCCESS) return 1;
  const isph_solve_info &info = li_solver.lastSolveInfo();
  std::printf("rank %d of %d: converged=%d iters=%d rel=%.3e\n", me, np, info.converged, info.iters, info.rel_res_implicit);
  const std::string fout = std::string(dir) + "/x" + std::to_string(me) + ".bin";
  f = std::fopen(fout.c_str(), "wb");
  const int hdr[2] = {info.converged, info.iters};
  std::fwrite(hdr, 4, 2, f);
  std::fwrite(x.data(), 8, x.size(), f);
  std::fclose(f);
  return info.converged ? 0 : 3;
}
#endif

static int run(int argc, char **argv);
int main(int argc, char **argv) {
#ifdef ISPH_HAVE_MPI
  MPI_Init(&argc, &argv);
  const int rc = run(argc, argv);
  MPI_Finalize();
  return rc;
#else
  return run(argc, argv);
#endif
}

static int run(int argc, char **argv) {
#ifdef ISPH_HAVE_MPI
  if (argc > 5 && std::string(argv[4]) == "ranks") return run_ranks(argv[1], std::atoi(argv[3]) != 0, argv[5]);
#endif
  if (argc > 4 && std::string(argv[4]) == "block") return run_block(argv[1], argv[2]);
  if (argc > 4 && std::string(argv[4]) == "selfhalo") return run_selfhalo(argv[1], argv[2], false);
  if (argc > 4 && std::string(argv[4]) == "selfhalo-overlap") return run_selfhalo(argv[1], argv[2], true);
  if (argc > 4 && std::string(argv[4]) == "selfhalo-extend") return run_selfhalo(argv[1], argv[2], true, true);
  if (argc < 4) { std::fprintf(stderr, "usage: %s in.bin out.bin singular(0/1) [cg|ml]\n", argv[0]); return 2; }
  FILE *f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int n = 0, nnz = 0;
  if (std::fread(&n, 4, 1, f) != 1 || std::fread(&nnz, 4, 1, f) != 1) return 2;
  std::vector<int> rp((size_t)n + 1), ci((size_t)nnz), gid((size_t)n);
  std::vector<double> val((size_t)nnz), b((size_t)n), x((size_t)n, 0.0);
  if (std::fread(rp.data(), 4, rp.size(), f) != rp.size() || std::fread(ci.data(), 4, ci.size(), f) != ci.size() ||
      std::fread(val.data(), 8, val.size(), f) != val.size() || std::fread(b.data(), 8, b.size(), f) != b.size()) return 2;
  std::fclose(f);
  for (int i = 0; i < n; ++i) gid[(size_t)i] = i + 1;
  const bool singular = std::atoi(argv[3]) != 0;

  MPI_Comm world = kWorld;
  Epetra_Map nodalmap(-1, n, gid.data(), 1, Epetra_MpiComm(world));
  Epetra_CrsMatrix AA(n, n, rp.data(), ci.data(), val.data());

  // "ml-xml": PrecondWrapper_ML with the keys of the benchmark protocol's ml.xml (bench-script/hopper/tgv/1728/ml.xml)
  const bool ml_xml = argc > 4 && std::string(argv[4]) == "ml-xml";
  const bool use_ml = (argc > 4 && std::string(argv[4]) == "ml") || ml_xml;
  const bool use_cg = argc > 4 && std::string(argv[4]) == "cg";
  const bool ifpack_defaults = argc > 4 && std::string(argv[4]) == "ifpack-defaults";  // level-of-fill 1, overlap 1
  const bool ifpack_reference = argc > 4 && std::string(argv[4]) == "ifpack-reference";  // + one subdomain = the whole matrix
  const bool recycling = argc > 4 && std::string(argv[4]) == "recycling";  // "Solver Type" = "Recycling GMRES"
  const bool timed = argc > 4 && std::string(argv[4]) == "timed";  // repeat the solve, report wall time per phase
  PrecondWrapper_Ifpack prec_ifpack(world);
  PrecondWrapper_ML prec_ml(world);
  PrecondWrapper &prec = use_ml ? static_cast<PrecondWrapper &>(prec_ml) : static_cast<PrecondWrapper &>(prec_ifpack);
  Teuchos::ParameterList *pp = prec.setParameters();
  if (use_ml) {  // the keys of precond_ml.h:44-55 are already set; shrink the hierarchy to the test size
    pp->set("coarse: max size", 64);
    if (ml_xml) {
      pp->set("max levels", 10);
      pp->set("smoother: type", "ML Gauss-Seidel");
      pp->set("smoother: Gauss-Seidel efficient symmetric", true);
      pp->set("smoother: sweeps", 4);
    } else {
      pp->set("aggregation: threshold", 0.02);
    }
  } else if (!ifpack_defaults && !ifpack_reference) {
    pp->set("fact: level-of-fill", 0);
    pp->set("Overlap Level", 0);
  }
  pp->set("isph: block rows", ifpack_reference ? 0 : timed ? 512 : 256);
  // argv[6] = rows per subdomain of a caller-defined table (PrecondWrapper_Ifpack::setSubdomains): consecutive ranges of
  // that many rows, e.g. the 500-row bricks of bench.py's particle numbering
  std::vector<int> subptr;
  if (argc > 6 && std::atoi(argv[6]) > 0 && !use_ml) {
    const int sub = std::atoi(argv[6]);
    for (int at = 0; at < n; at += sub) subptr.push_back(at);
    subptr.push_back(n);
    prec_ifpack.setSubdomains((int)subptr.size() - 1, subptr.data());
  }

  // argv[7] = file with the coordinates of the rows, three arrays of n doubles (x, y, z) one after the other: the adapter's
  // three-line call of INTEGRATION.md, PrecondWrapper_Ifpack::setCoordinates / PrecondWrapper_ML::setCoordinates
  // (pair_isph.cpp:1290-1303 makes it for ML) -- the library then numbers the rows itself
  std::vector<double> coords;
  if (argc > 7 && std::string(argv[7]) != "-") {
    FILE *fc = std::fopen(argv[7], "rb");
    coords.resize((size_t)3 * n);
    if (!fc || std::fread(coords.data(), 8, coords.size(), fc) != coords.size()) { std::fprintf(stderr, "cannot read the coordinates\n"); return 2; }
    std::fclose(fc);
    if (use_ml) prec_ml.setCoordinates(3, coords.data(), coords.data() + n, coords.data() + 2 * (size_t)n);
    else prec_ifpack.setCoordinates(3, coords.data(), coords.data() + n, coords.data() + 2 * (size_t)n);
  }

  SolverLin_Belos li_solver(world);
  li_solver.setParameters();
  Teuchos::ParameterList cgp;
  if (use_cg) {  // USER-REAXC-T defaults: Block CG (USER-REAXC-T/solver_lin_belos.h:236-245)
    cgp.set("Solver Type", "Block CG");
    cgp.set("Convergence Tolerance", 1.0e-8);
    cgp.set("Maximum Iterations", 500);
    li_solver.setParameters(&cgp);
  }
  Teuchos::ParameterList rcp_;
  if (recycling) {  // GCRO-DR(20, 5): the keys of solver_lin_belos.h:224-264 with a valid recycle size
    rcp_.set("Solver Type", "Recycling GMRES");
    rcp_.set("Num Blocks", 20);
    rcp_.set("Num Recycled Blocks", 5);
    rcp_.set("Convergence Tolerance", 1.0e-8);
    li_solver.setParameters(&rcp_);
  }
  li_solver.setNodalMap(&nodalmap);
  li_solver.setMatrix(&AA);
  prec.setMatrix(&AA);
  li_solver.createSolutionMultiVector(x.data(), n, 1);
  li_solver.createLoadMultiVector(b.data(), n, 1);
  Epetra_IntSerialDenseVector null_mask(n);
  if (singular) {
    for (int i = 0; i < n; ++i) null_mask[i] = 1;
    li_solver.setNullVectorMask(&null_mask);
    li_solver.setMatrixIsSingular(true);
  }
  li_solver.setInitialSolution(SolverLin::Zero);
  if (timed) {
    // the per-step sequence of PairISPH::computePoisson's caller (pair_isph.cpp:988-1011) repeated on one host matrix:
    // setMatrix / setInitialSolution / solveProblem, wall time per call split into ingress / set-up / Krylov
    const int repeat = argc > 5 ? std::atoi(argv[5]) : 5;
    const std::vector<double> b0 = b;  // solveProblem projects b in place
    li_solver.setTiming(true);
    std::vector<double> t[5];
    int iters = 0, conv = 1;
    double gpu_solve_ms = 0.0;
    for (int k = 0; k < repeat + 1; ++k) {
      b = b0;
      li_solver.setMatrix(&AA);
      prec.setMatrix(&AA);
      li_solver.setInitialSolution(SolverLin::Zero);
      const std::chrono::steady_clock::time_point w0 = std::chrono::steady_clock::now();
      if (li_solver.solveProblem(&prec, NULL) != LAMMPS_SUCCESS) return 1;
      const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
      if (k == 0) continue;  // first call: context, pinned ring and device pool are created
      const double *ms = li_solver.lastTimingsMs();
      for (int q = 0; q < 4; ++q) t[q].push_back(ms[q]);
      t[4].push_back(wall);
      iters = li_solver.lastSolveInfo().iters;
      gpu_solve_ms = li_solver.lastSolveInfo().solve_ms;
      conv = conv && li_solver.lastSolveInfo().converged;
    }
    for (int q = 0; q < 5; ++q) std::sort(t[q].begin(), t[q].end());
    const size_t mid = t[4].size() / 2;
    std::printf("{\"dropin\": {\"rows\": %d, \"entries\": %d, \"repeat\": %d, \"iterations\": %d, \"converged\": %d, "
                "\"ms_per_solve\": %.3f, \"ingress_ms\": %.3f, \"setup_ms\": %.3f, \"krylov_ms\": %.3f, \"release_ms\": %.3f, "
                "\"ingress_GBps\": %.2f, \"krylov_gpu_event_ms_last_call\": %.3f}}\n",
                n, nnz, repeat, iters, conv, t[4][mid], t[0][mid], t[1][mid], t[2][mid], t[3][mid],
                (12.0 * nnz + 4.0 * (n + 1)) / t[0][mid] * 1e-6, gpu_solve_ms);
    double gi[8];
    if (li_solver.lastIngressInfo(gi) == ISPH_SUCCESS)
      std::printf("{\"ingress_last_call\": {\"staged_ms\": %.3f, \"queued_ms\": %.3f, \"copied_ms\": %.3f, \"device_done_ms\": %.3f, "
                  "\"end_ms\": %.3f, \"waited_for_staging_ms\": %.3f, \"link_bytes\": %.0f, \"link_GBps_while_copying\": %.2f, \"threads\": %.0f}}\n",
                  gi[0], gi[1], gi[2], gi[3], gi[4], gi[5], gi[6], gi[6] / gi[2] * 1e-6, gi[7]);
  }
  const int rc = timed ? LAMMPS_SUCCESS : li_solver.solveProblem(&prec, "test_solver_lin");
  if (rc != LAMMPS_SUCCESS) return 1;
  const isph_solve_info &info = li_solver.lastSolveInfo();
  std::printf("converged=%d iters=%d rel=%.3e\n", info.converged, info.iters, info.rel_res_implicit);
  f = std::fopen(argv[2], "wb");
  std::fwrite(x.data(), 8, x.size(), f);
  std::fwrite(b.data(), 8, b.size(), f);
  std::fclose(f);
  return info.converged ? 0 : 3;
}
