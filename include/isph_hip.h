/*
 * isph_hip.h -- C ABI of libisph_hip.so: the MI355X (gfx950) implementation of
 * implicit-sph's pressure-Poisson / Helmholtz inner loop.
 *
 * This is the drop-in boundary.  Everything above it (the SolverLin /
 * PrecondWrapper C++ classes in implicit-sph_amd/host/, a LAMMPS adapter, the
 * Python ctypes binding) talks to the GPU only through these entry points:
 * plain pointers and sizes, opaque handles, int return codes
 * (0 = LAMMPS_SUCCESS, -1 = LAMMPS_FAILURE, ref: macrodef.h:20-24), no
 * exceptions, no torch types.  The caller keeps ownership of every pointer it
 * passes; the library owns the device mirrors it creates.
 *
 * "ref:" citations are relative to /root/reference/IMPLICIT-SPH/ and name the
 * reference interface each entry point replaces.
 *
 * Pointer arguments marked [h|d] may be host or device pointers; the `on_device`
 * flag of the call says which (0 = host, the library stages the copy).
 * There is NO CPU fallback: every compute entry point fails with -1 if no
 * HIP device is usable.
 */
#ifndef ISPH_HIP_H
#define ISPH_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct isph_ctx isph_ctx;   /* device, stream, workspaces, RCCL communicator */
typedef struct isph_mat isph_mat;   /* device matrix: sliced-ELL (+ CSR view) + halo plan */
typedef struct isph_prec isph_prec; /* device preconditioner */

#define ISPH_SUCCESS 0
#define ISPH_FAILURE (-1)
#define ISPH_UID_BYTES 128

/* ---- context ---------------------------------------------------------- */

/* Replaces the Epetra_MpiComm every reference object is built on
 * (ref: solver_lin.cpp:30-31, precond.h:26).  `stream` = an existing
 * hipStream_t to launch on (NULL: the library creates its own). */
int isph_ctx_create(int device, void *stream, isph_ctx **ctx);
/* Multi-GPU: one process per GPU; the communicator is RCCL.  `uid` is the
 * ncclUniqueId (ISPH_UID_BYTES) produced on rank 0 by isph_comm_unique_id
 * and broadcast by the launcher (torch.distributed / MPI). */
int isph_comm_unique_id(char *uid);
int isph_ctx_create_dist(int device, void *stream, int rank, int nranks,
                         const char *uid, isph_ctx **ctx);
/* Multi-rank WITHOUT RCCL: ranks that share a device (several MPI ranks of a LAMMPS run on one GPU -- RCCL refuses two
 * ranks of a communicator on one device), or a launcher that wants its own MPI communicator to carry the traffic as
 * the reference does (Epetra_MpiComm, ref: solver_lin.cpp:30-31).  The library stages the device buffers through pinned
 * host memory and calls back; kernels, streams and their order are those of the RCCL path, only the host waits.
 *   exchange : for p in [0,npeers): send[send_off[p] .. send_off[p+1]) -> rank peer[p], recv[recv_off[p] .. recv_off[p+1])
 *              <- rank peer[p]; all messages posted together (MPI_Irecv/MPI_Isend/MPI_Waitall); a peer may be the rank
 *              itself.  Every rank calls it in the same order; a rank without peers does not call it.
 *   allreduce: in place over all ranks, op 0 = sum, 1 = max; called by every rank.
 * Both return 0 on success.  host/mpi_transport.h is the MPI implementation the C++ mirror uses. */
typedef struct {
  void *user;
  int (*exchange)(void *user, int npeers, const int *peer, const double *send, const long long *send_off,
                  double *recv, const long long *recv_off);
  int (*allreduce)(void *user, double *buf, int count, int op);
} isph_host_transport;
int isph_ctx_create_hostcomm(int device, void *stream, int rank, int nranks, const isph_host_transport *transport,
                             isph_ctx **ctx);
/* Physical identity of HIP device `device` as this process sees it: its PCI bus id ("0000:c1:00.0"), zero-padded to
 * ISPH_DEVICE_ID_BYTES.  Two ranks share a GPU exactly when their strings are equal -- their ordinals say nothing when
 * every rank runs under its own ROCR_VISIBLE_DEVICES mask (host/solver_lin_hip.h picks the transport by it). */
#define ISPH_DEVICE_ID_BYTES 64
int isph_device_identity(int device, char id[ISPH_DEVICE_ID_BYTES]);
int isph_ctx_sync(isph_ctx *ctx);
void isph_ctx_destroy(isph_ctx *ctx);
/* Device buffers released by the library are kept for the next set-up (the reference rebuilds matrix and preconditioner
 * every time step, pair_isph.cpp:1257-1287,1363) up to 80 % of the device's memory (a failed allocation trims them first).  isph_pool_trim() synchronises the
 * device and returns them to the driver; isph_pool_cached_bytes() reports how much is held.  No reference counterpart
 * (the reference has no device memory). */
int isph_pool_trim(void);
/* Upper limit of the cache in bytes (default: 80 % of the device memory that was free when the
 * library first gave a block back; bytes <= 0 restores the default).  A process that shares the
 * device with another allocator (torch, a second library) sets this to what it can spare, or calls
 * isph_pool_trim() before the other allocator needs the memory. */
int isph_pool_set_cap(long long bytes);
/* The triangular-solve stream of the block ILU / Gauss-Seidel set-up is reserved at twice a proven bound of its size
 * (one pass, no counting) unless that reservation exceeds this many bytes; above it the schedule first counts the
 * chunks of every block and the stream is allocated exactly (one more pass over the factor pattern, half the memory:
 * the smoother of the 4 M x 749 operator of BASELINE configs[4] takes 55 GB instead of 110).  Default 4 GiB;
 * bytes < 0 restores it, 0 sizes every stream exactly.  Process-wide.  No reference counterpart. */
int isph_set_exact_stream_threshold(long long bytes);
long long isph_pool_cached_bytes(void);
/* [0] bytes cached, [1] bytes handed out to live objects, [2] high-water mark of [1] (reset to [1] when reset_peak != 0),
 * [3] the cache limit in force (0: not yet determined). */
int isph_pool_info(long long info[4], int reset_peak);
const char *isph_last_error(void);

/* ---- matrix ----------------------------------------------------------- */

/* Ingress of an assembled local matrix: the three arrays
 * Epetra_CrsMatrix::ExtractCrsDataPointers returns after
 * FillComplete+OptimizeStorage (ref: pair_isph.cpp:1266-1270,
 * solver_lin.h:52 setMatrix).  Rows = locally owned particles; columns
 * [0,nrow) = owned, [nrow,ncol) = ghost columns filled by the halo exchange.
 * Replaces SolverLin::setMatrix / PrecondWrapper::setMatrix. */
int isph_mat_create_csr(isph_ctx *ctx, int nrow, int ncol, const int *rowptr /*[h|d]*/,
                        const int *colidx /*[h|d]*/, const double *val /*[h|d]*/,
                        int on_device, isph_mat **A);
/* The same ingress from HOST arrays, fused with the set-up of the block-Jacobi ILU(0)
 * preconditioner ("bjacobi-ilu0", block_size rows per subdomain): the matrix crosses PCIe in
 * chunks, and the set-up of the blocks whose rows have arrived runs while the rest is still on
 * the link.  What SolverLin_Belos::solveProblem does between receiving the host matrix and
 * starting Belos -- prec->create() = Ifpack Initialize + Compute (ref: solver_lin_belos.h:147-156,
 * precond_ifpack.h:60-74) -- with the result of isph_mat_create_csr(on_device = 0) followed by
 * isph_prec_create(A, "bjacobi-ilu0", block_size), bit for bit. */
int isph_mat_create_csr_bjacobi(isph_ctx *ctx, int nrow, int ncol, const int *rowptr /*[h]*/,
                                const int *colidx /*[h]*/, const double *val /*[h]*/, int block_size,
                                isph_mat **A, isph_prec **M);
/* The same with the caller's subdomains (see isph_prec_create_blocks): the result of isph_mat_create_csr(on_device = 0)
 * followed by isph_prec_create_blocks(A, nblocks, block_ptr), bit for bit. */
int isph_mat_create_csr_blocks(isph_ctx *ctx, int nrow, int ncol, const int *rowptr /*[h]*/, const int *colidx /*[h]*/,
                               const double *val /*[h]*/, int nblocks, const int *block_ptr /*[h]*/, isph_mat **A,
                               isph_prec **M);
/* The same ingress INTO THE LIBRARY'S OWN ROW NUMBERING (isph_ctx_set_ordering), for the drop-in path, where the matrix
 * arrives assembled in LAMMPS' atom order and no particle array comes with it: the caller hands over the coordinates of
 * its rows the way PrecondWrapper_ML::setCoordinates receives them (ref: precond_ml.h:63-94 -- three host arrays of nrow
 * doubles cut from atom->x, pair_isph.cpp:1290-1303; z may be NULL when dim == 2).  The rows are sorted into the bricks
 * of order.hpp from the coordinates, the matrix crosses the link as it is and is permuted on the device in one pass
 * (rows, owned columns, column order inside the rows).  The result behaves like a matrix from isph_assemble_poisson:
 * every vector at this boundary stays in the caller's numbering, isph_prec_create(.., block_size 0) uses the bricks. */
int isph_mat_create_csr_coords(isph_ctx *ctx, int nrow, int ncol, const int *rowptr /*[h]*/, const int *colidx /*[h]*/,
                               const double *val /*[h]*/, int dim, const double *x /*[h]*/, const double *y /*[h]*/,
                               const double *z /*[h]*/, isph_mat **A);
/* The same fused with the set-up of "bjacobi-ilu0" on the library's bricks: the staging threads gather the caller's rows
 * in the NEW order (the permutation is known before the first entry leaves the host), the conversion kernel renames the
 * columns and a ranged row sort follows it, so that -- as in isph_mat_create_csr_blocks -- the 16-bit column windows and
 * the ILU(0) extraction / schedule / factorisation of the bricks whose rows have arrived run while the rest of the matrix
 * is still on the link.  Result: the matrix of isph_mat_create_csr_coords and isph_prec_create(A, "bjacobi-ilu0", 0). */
int isph_mat_create_csr_coords_bjacobi(isph_ctx *ctx, int nrow, int ncol, const int *rowptr /*[h]*/, const int *colidx /*[h]*/,
                                       const double *val /*[h]*/, int dim, const double *x /*[h]*/, const double *y /*[h]*/,
                                       const double *z /*[h]*/, isph_mat **A, isph_prec **M);
/* Diagnostics of the last host-side ingress on this context (milliseconds since its start):
 * [0] staging threads started, device buffers reserved  [1] all chunks queued on the copy stream
 * [2] copy stream drained  [3] compute stream drained (conversion + fused set-up)  [4] end
 * [5] time the queueing thread waited for staged chunks  [6] bytes that crossed the link (16-bit
 * column differences where the rows allow them: 10 instead of 12 per entry)  [7] staging threads.
 * No reference counterpart. */
int isph_ingress_info(const isph_ctx *ctx, double info[8]);
/* Halo plan = what Epetra builds inside FillComplete (column map + Import,
 * ref: functor_graph.h:97).  For peer p: send x[send_idx[send_ptr[p]..send_ptr[p+1])]
 * and receive the ghost columns nrow+recv_ptr[p] .. nrow+recv_ptr[p+1]. */
int isph_mat_set_halo(isph_ctx *ctx, isph_mat *A, int npeers, const int *peer_rank,
                      const int *send_ptr, const int *send_idx, const int *recv_ptr);
/* The same plan as a stand-alone object, for per-atom fields that are not matrix columns yet: LAMMPS'
 * comm->forward_comm_pair(this) with PairISPH::pack_forward_comm / unpack_forward_comm (ref: pair_isph.cpp:1924-2110;
 * call sites functor_volume.h:78-80 for Vfrac, pair_isph.cpp:977-979 Vstar, :1017-1019 DeltaP, :1123-1125 Pressure,
 * :1161-1163 Velocity).  isph_halo_forward sends ncomp doubles per listed owned atom (x is [nlocal][ncomp], the layout
 * of atom->v / the pair's per-atom arrays) and returns the ghost values in ghost-column order: ghosts is
 * [recv_ptr[npeers]][ncomp].  Runs on the context's stream over its RCCL communicator (grouped ncclSend/ncclRecv). */
typedef struct isph_halo_plan isph_halo_plan;
int isph_halo_create(isph_ctx *ctx, int nlocal, int npeers, const int *peer_rank, const int *send_ptr,
                     const int *send_idx, const int *recv_ptr, isph_halo_plan **plan);
int isph_halo_forward(isph_ctx *ctx, const isph_halo_plan *plan, const double *x /*[h|d]*/, double *ghosts /*[h|d]*/,
                      int ncomp, int on_device);
void isph_halo_destroy(isph_halo_plan *plan);
/* sizes: [0]=nrow [1]=ncol [2]=nnz [3]=number of 64-row slices
 *        [4]=stored (padded) entries [5]=bytes of the sliced-ELL arrays */
int isph_mat_info(const isph_mat *A, long long info[6]);
/* Export as CSR with sorted columns (device->host); caller sizes from info. */
int isph_mat_export_csr(isph_ctx *ctx, const isph_mat *A, int *rowptr, int *colidx, double *val);
/* Rows [row_begin, row_begin + nrows) as CSR, row pointers relative to the range (64-bit), columns ascending; fails
 * when the rows hold more than `capacity` entries.  For host-side checks of matrices beyond a 32-bit CSR (the
 * 3*10^9-entry operator of BASELINE configs[4]): a test exports a few thousand sampled rows and multiplies them itself.
 * No reference counterpart (Epetra_CrsMatrix::ExtractMyRowView is the closest). */
int isph_mat_export_rows(isph_ctx *ctx, const isph_mat *A, int row_begin, int nrows, long long *rowptr, int *colidx,
                         double *val, long long capacity);
void isph_mat_destroy(isph_mat *A);

/* y = A x  (Epetra_CrsMatrix::Apply incl. the ghost Import; ref: solver_lin.h:133).
 * x: nrow owned entries -- the ghost columns are fetched through the halo plan; a matrix WITH ghost columns (ncol >
 * nrow) but WITHOUT a plan (isph_mat_set_halo not called) reads all ncol entries from the caller.  y: nrow entries. */
int isph_spmv(isph_ctx *ctx, const isph_mat *A, const double *x /*[h|d]*/,
              double *y /*[h|d]*/, int on_device);
/* Time `reps` back-to-back SpMV launches with HIP events on the library
 * stream; returns the average kernel time in milliseconds.  variant 0 = the production kernel; 1..5 = experimental
 * instantiations of the 32-bit-column kernel (unroll / cache policy), for scripts/spmv_variants.py only. */
int isph_spmv_time(isph_ctx *ctx, const isph_mat *A, const double *x_dev, double *y_dev,
                   int reps, int variant, double *avg_ms);

/* ---- preconditioner --------------------------------------------------- */

/* Replaces PrecondWrapper_Ifpack::create() = Ifpack factory + Initialize +
 * Compute (ref: precond_ifpack.h:52-75).  type:
 *   "none"          identity
 *   "jacobi"        point Jacobi (debug)
 *   "bjacobi-ilu<k>"  k = 0..8: block-Jacobi, ILU(k) per block == Ifpack
 *                   AdditiveSchwarz<ILU>, "Overlap Level"=0,
 *                   "fact: level-of-fill"=k (precond_ifpack.h:35; the reference's
 *                   default is k = 1), one block per `block_size` rows
 *                   (k > 0: block_size <= 1024 and the symbolic phase runs on the device)
 *   "ilu<k>"        ILU(k) of the whole local matrix: Ifpack on one MPI rank (precond_ifpack.h:60-74 with
 *                   Comm.NumProc() == 1); see isph_prec_create_schwarz
 *   "sa-amg"        PrecondWrapper_ML::create() with its default parameters and no null vector
 *                   (isph_prec_create_amg takes the parameters and the null vector of a singular system)
 * Rebuilt every solve in the reference (solver_lin_belos.h:153,190). */
/* block_size 0 ("bjacobi-ilu<k>" only): the matrix' own subdomains, i.e. the bricks the assembly sorted the particles
 * into (isph_ctx_set_ordering); fails for a matrix in the caller's numbering. */
int isph_prec_create(isph_ctx *ctx, const isph_mat *A, const char *type, int block_size,
                     isph_prec **M);
/* "bjacobi-ilu0" on the CALLER'S subdomains: block b = rows block_ptr[b] .. block_ptr[b+1] (host array of nblocks + 1
 * ascending offsets from 0 to nrow, at most 1024 rows each; the rows of a subdomain are consecutive in the matrix).  The
 * reference's subdomains are the bricks of LAMMPS' spatial decomposition (one per MPI rank, ref: precond_ifpack.h:60-74
 * with pair_isph.cpp:1258-1259); a caller that numbers its particles brick by brick hands the brick boundaries over
 * here instead of accepting a cut every block_size rows -- a thin remainder brick at the edge of the box is then a
 * subdomain of its own shape and not the two halves of its neighbours.  Same kernels and data layout as
 * isph_prec_create("bjacobi-ilu0"). */
int isph_prec_create_blocks(isph_ctx *ctx, const isph_mat *A, int nblocks, const int *block_ptr, isph_prec **M);
/* The same with "fact: level-of-fill" = level_of_fill in 0..8 (ref: precond_ifpack.h:35, the reference's default is 1): the
 * level-of-fill pattern of every subdomain (Ifpack_IlukGraph's rule) is formed on the device by level_of_fill sweeps of
 * the merge kernel, whatever the lengths of the subdomains. */
int isph_prec_create_blocks_fill(isph_ctx *ctx, const isph_mat *A, int nblocks, const int *block_ptr, int level_of_fill,
                                 isph_prec **M);
/* Ifpack_AdditiveSchwarz<Ifpack_ILU> with the parameters PrecondWrapper_Ifpack sets (ref: precond_ifpack.h:30-45,
 * 60-74): "fact: level-of-fill" (default 1), "Overlap Level" (default 1), "schwarz: combine mode" (default "Add" = 0;
 * 1 = "Zero", restricted additive Schwarz).  block_size = 0: one subdomain = the whole local matrix, which is what
 * the reference factors on one MPI rank (Ifpack ignores the overlap there); block_size = B > 0: consecutive
 * subdomains of B rows (any B), each extended by `overlap` layers of the rows its columns reference, like the ranks
 * of a parallel run.  isph_prec_create(type = "ilu<k>") is the block_size = 0 case.  Level-scheduled on the device:
 * the fidelity path; "bjacobi-ilu<k>" is the throughput path.
 * info: [0] extended rows [1] factor entries [2] subdomains [3] L levels [4] U levels [5] longest factor row
 * [6] 2 = one workgroup per subdomain does both sweeps (many small subdomains), 1 = persistent launches, 0 = one launch
 * per dependency level (see level_launches).
 * export: rows[nloc] (global row of every local row), loc_ptr[nsub+1], factor CSR in local numbering. */
typedef struct {
  int level_of_fill, overlap, combine, block_size;
  int level_launches; /* 0 (default): the form is picked by the shape of the problem -- >= 32 subdomains of <= 4096 rows:
                         ONE launch per application, a workgroup per subdomain with its part of the vector in LDS and a
                         barrier per level; narrow levels (whole-matrix factors): the factorisation and every triangular
                         sweep are ONE persistent launch whose rows wait for the rows they depend on; otherwise one launch
                         per level.  1: always one launch per level (the cross-check of the other forms: same factor bit
                         for bit, application equal to rounding) */
} isph_schwarz_params;
void isph_schwarz_params_default(isph_schwarz_params *p);
int isph_prec_create_schwarz(isph_ctx *ctx, const isph_mat *A, const isph_schwarz_params *prm, isph_prec **M);
int isph_prec_schwarz_info(const isph_prec *M, long long info[7]);
/* wall time of the create call, ms: [0] matrix to the host [1] subdomains + local matrices [2] level-of-fill pattern
 * [3] dependency levels, orders, combine lists [4] upload [5] numeric factorisation */
int isph_prec_schwarz_timing(const isph_prec *M, double ms[6]);
int isph_prec_schwarz_export(isph_ctx *ctx, const isph_prec *M, int *rows, int *loc_ptr, long long *rowptr,
                             int *colidx, double *val);
/* Ifpack_AdditiveSchwarz<ILU(k)> with "Overlap Level" 1 across ranks (ref: precond_ifpack.h:43,60-74; Ifpack builds an
 * Ifpack_OverlappingRowMatrix from the matrix' Epetra_Import).  Aext is the square matrix of this rank's extended
 * subdomain: rows/columns [0,nlocal) = owned, [nlocal, nlocal + nghost) = the rows of the ghost columns in ghost-column
 * order, entries outside the extended column set dropped (the adapter imports those rows with the matrix' importer;
 * dist.extend_rows does it for the Python plumbing).  The halo lists are the ones of isph_mat_set_halo for the
 * un-extended matrix.  apply: gather the ghost part of r from its owners, ILU(k) solve on the extended vector (one
 * subdomain, level-scheduled path), and with combine 0 = "Add" (the wrapper's default, precond_ifpack.h:37) send the
 * ghost part of the result back to the owners, who add it; 1 = "Zero" keeps the owned part (restricted Schwarz). */
int isph_prec_create_overlap(isph_ctx *ctx, const isph_mat *Aext, int nlocal, int level_of_fill, int combine, int npeers,
                             const int *peer_rank, const int *send_ptr, const int *send_idx, const int *recv_ptr,
                             isph_prec **M);

/* z = M^-1 r (Belos::EpetraPrecOp::Apply -> Ifpack ApplyInverse). */
int isph_prec_apply(isph_ctx *ctx, const isph_prec *M, const double *r /*[h|d]*/,
                    double *z /*[h|d]*/, int on_device);
/* Export the factor as CSR (strict L, D, strict U in one pattern == A's
 * in-block pattern, columns sorted) for parity tests. */
int isph_prec_export_ilu(isph_ctx *ctx, const isph_prec *M, int *rowptr, int *colidx, double *val);
long long isph_prec_nnz(const isph_prec *M);
/* sizes: [0]=factor entries [1]=triangular-solve stream chunks in use (64 entries each)
 *        [2]=stream capacity in chunks [3]=number of blocks */
int isph_prec_info(isph_ctx *ctx, const isph_prec *M, long long info[4]);
void isph_prec_destroy(isph_prec *M);

/* ---- solve ------------------------------------------------------------ */

/* Same keys and defaults as SolverLin_Belos::setParameters
 * (ref: solver_lin_belos.h:224-264). */
typedef struct {
  int solver_type;   /* 0 "Block GMRES" (default), 1 "Block CG", 2 "Recycling GMRES" = GCRO-DR(num_blocks,
                        num_recycled) (solver_lin_belos.h:173-181)           */
  int flexible;      /* "Flexible Gmres" (default 1)                        */
  int num_blocks;    /* "Num Blocks" (50)                                   */
  int max_iters;     /* "Maximum Iterations" (500)                          */
  int max_restarts;  /* "Maximum Restarts" (15)                             */
  double tol;        /* "Convergence Tolerance" (1e-8)                      */
  int ortho;         /* "Orthogonalization": 0 DGKS (default), 1 ICGS, 2 IMGS */
  int verbose;       /* rank-0 status lines like Belos "Verbosity"          */
  int num_recycled;  /* "Num Recycled Blocks" (50, solver_lin_belos.h:240); solver_type 2 needs
                        0 < num_recycled < num_blocks, as Belos::GCRODRSolMgr does */
} isph_solver_params;
void isph_solver_params_default(isph_solver_params *p);

typedef struct {
  int converged, iters, restarts;
  double rel_res_implicit;  /* recurrence residual / ||r0||                 */
  double rel_res_explicit;  /* ||b - A x|| / ||b|| (solver_lin_belos.h:201-212) */
  double prec_setup_ms;     /* filled by callers that build M around the solve */
  double solve_ms;          /* HIP-event time of the whole call on the stream */
  double spmv_ms;           /* sum of SpMV kernel times (profile mode only) */
  int spmv_calls;
  int reorth;               /* Gram-Schmidt steps whose second pass was applied (DGKS: when the norm dropped below
                               1/sqrt(2) of its value before the first pass; ICGS: every step) */
} isph_solve_info;

/* Replaces SolverLin_Belos::solveProblem (ref: solver_lin_belos.h:130-222):
 * singular => n = mask/||mask||, b -= (b.n)n, operator y = Ax - (Ax.n)n,
 * x -= (x.n)n; right preconditioning; non-convergence is reported in `info`,
 * never an error.  b and x are column-major [lda x nvec] exactly as
 * createLoadMultiVector / createSolutionMultiVector receive them
 * (ref: solver_lin.cpp:45-58); columns are solved one after another
 * (Belos block size 1).  x holds the initial guess on entry.
 * b is overwritten by its projection when singular, as in the reference. */
int isph_solve(isph_ctx *ctx, const isph_mat *A, const isph_prec *M,
               double *b /*[h|d]*/, double *x /*[h|d]*/, int nvec, int lda,
               int is_singular, const int *null_mask /*[h] or NULL*/,
               const isph_solver_params *prm, isph_solve_info *info, int on_device);
/* profile mode: HIP events on the library's stream around the launches of the hot kernels, by class:
 *   [0] SpMV (k_sell_spmv16 incl. the halo exchange when there is one)   [1] preconditioner application (block ILU:
 *   k_ilu_solve_stream)   [2] k_multi_dot   [3] k_multi_axpy_dot   [4] k_multi_axpy_norm (the three sweeps of one
 *   DGKS step)   [5] k_ilu_extract   [6] k_ilu_schedule   [7] k_ilu_factor (set-up of the block ILU).
 * isph_ctx_set_profile switches the mode and starts a new collection; isph_ctx_profile_read synchronises the stream,
 * returns milliseconds and launch counts per class since the collection started, and starts the next one.
 * isph_solve_info::spmv_ms / spmv_calls are class 0 of that one solve.  No reference counterpart (the reference's
 * Teuchos timers stop at "ISPH: solvePoisson", utils.cpp:37-38). */
int isph_ctx_set_profile(isph_ctx *ctx, int on);
/* Between isph_ctx_hold_neighbours(ctx, 1) and isph_ctx_hold_neighbours(ctx, 0) the caller guarantees that the neighbour
 * list it passes in isph_particles (the same neigh_ptr / neigh_idx arrays, on the device) does not change: the layout the
 * row kernels read -- slice-transposed, ordered by matrix column for the assemblies -- is then built by the first operator
 * call and reused by the following ones, instead of once per call.  What LAMMPS' neighbour list is to the reference's
 * functors between two re-neighbourings (pair_isph.cpp: every functor of a time step walks the same list->firstneigh).
 * Either call drops what was kept.  Without it every call is self-contained. */
int isph_ctx_hold_neighbours(isph_ctx *ctx, int on);
int isph_ctx_profile_read(isph_ctx *ctx, double ms[8], int calls[8]);
/* profile mode, products of a matrix with a halo plan (more than one rank): per product the time from "boundary values
 * packed" to "ghost values landed" on the halo stream (ms[0], summed over `calls` products), the time the interior slices
 * took on the compute stream from the same instant (ms[1]), and the part of the exchange that was not hidden behind them,
 * max(0, exchange - interior) (ms[2]).  Starts the next collection.  No reference counterpart (Epetra's Import is
 * synchronous inside Epetra_CrsMatrix::Apply, solver_lin.h:133). */
int isph_ctx_halo_profile_read(isph_ctx *ctx, double ms[3], int *calls);
/* info: [0] transport of the context: 0 none (one rank), 1 RCCL, 2 host-staged  [1] ranks -- ncclCommCount of the
 * communicator for transport 1  [2] this rank (ncclCommUserRank)  [3] device (ncclCommCuDevice) */
int isph_ctx_comm_info(const isph_ctx *ctx, long long info[4]);

/* ---- row numbering ----------------------------------------------------- */

/* The reference's matrix rows follow LAMMPS' atom order (the Epetra map is atom->tag[0..nlocal), ref:
 * pair_isph.cpp:1258-1259) and Ifpack's subdomains are the bricks of the MPI decomposition (ref: precond_ifpack.h:60-74).
 * ISPH_ORDER_BRICKS (the default): every assembly entry point (isph_assemble_poisson / _helmholtz / _block_helmholtz and
 * the scalar callers) sorts the rank's owned particles into bricks of about 500 particles by their coordinates
 * (10 x 10 x 5 mean spacings in 3-D, 22 x 22 in 2-D; x fastest inside a brick; over-full bricks split) and builds the
 * matrix in that numbering -- whatever order the caller's atoms are in.  Nothing of it is visible at this boundary: b, x
 * and the null mask of isph_solve, x / y of isph_spmv, r / z of isph_prec_apply, the null vector of isph_prec_create_amg,
 * the send list of isph_mat_set_halo and the exports isph_mat_export_csr / _rows are in the caller's numbering (ghost
 * columns are never renumbered).  The bricks are the subdomains of "bjacobi-ilu<k>" when isph_prec_create is called with
 * block_size 0.  While the neighbour list is held (isph_ctx_hold_neighbours) the order of the first assembly serves the
 * following ones.  Internals exported for tests (isph_prec_export_ilu, isph_prec_amg_*) are in the matrix' numbering;
 * isph_mat_ordering gives the permutation.
 * ISPH_ORDER_CALLER: rows stay in the caller's atom order (rounds 1-4; subdomains then are `block_size` consecutive rows
 * or the caller's table, isph_prec_create_blocks).  Matrices from isph_mat_create_csr are always in the caller's order. */
#define ISPH_ORDER_CALLER 0
#define ISPH_ORDER_BRICKS 1
int isph_ctx_set_ordering(isph_ctx *ctx, int mode);
/* Optional, for ISPH_ORDER_BRICKS: the caller's periodic box (LAMMPS: domain->boxlo / boxhi / periodicity; on more than one
 * rank set periodic[a] only for axes the rank's sub-domain spans alone).  Atoms that have just wrapped around a periodic
 * box end sit at the other end of the coordinate range -- half a lattice plane at x = L - eps, the other half at +eps --,
 * which stretches the bounding box the bricks are cut from by a spacing and leaves bricks of 9 ... 11 planes per axis
 * instead of 10 (all subdomains then pay for the largest one's LDS).  With the box the sort takes coordinates relative to a
 * point in the widest empty stretch of each periodic axis, modulo the period.  NULL arguments forget the box.  Without the
 * call everything works, with uneven bricks in that situation.  No reference counterpart. */
int isph_ctx_set_periodic_box(isph_ctx *ctx, const double lo[3], const double hi[3], const int periodic[3]);
/* what the sort was made with: cell(a) = number of faces of axis a that are <= x_a (isph_mat_ordering_faces: ncell[a] - 1
 * ascending doubles), brick(a) = cell(a) / cells_per_brick[a]; key = ((brick_z * nbrick[1] + brick_y) * nbrick[0] +
 * brick_x) * (cells per brick) + (cz' * cpb[1] + cy') * cpb[0] + cx' with c' = cell mod cells_per_brick; rows ascend by
 * key, ties in the caller's order.  The faces are quantiles of the owned particles' coordinates: face k of an axis is the
 * upper edge lo + (j + 1) / inv_bin of the first bin j of the histogram bin(x) = clamp(floor((x - lo) * inv_bin), 0,
 * nbins - 1) at which the cumulative count reaches ceil(k n / ncell). */
typedef struct {
  int dim;
  double lo[3], inv_bin[3];
  int nbins[3], ncell[3], cells_per_brick[3], nbrick[3];
  double shift[3], period[3]; /* every coordinate enters as t = x - shift, + period when t < 0 (isph_ctx_set_periodic_box; period 0: as is) */
} isph_order_geometry;
/* info: [0] 1 when A carries the library's numbering (0: the caller's) [1] rows [2] subdomains; geom may be NULL */
int isph_mat_ordering_info(const isph_mat *A, long long info[3], isph_order_geometry *geom);
/* perm[r] = the caller's row held by internal row r ([nrow], may be NULL); block_ptr [subdomains + 1] (may be NULL).
 * Fails for a matrix in the caller's numbering. */
int isph_mat_ordering(isph_ctx *ctx, const isph_mat *A, int *perm, int *block_ptr);
/* faces[0 .. ncell[axis] - 2] of one axis (host array of the caller) */
int isph_mat_ordering_faces(const isph_mat *A, int axis, double *faces);

/* ---- assembly --------------------------------------------------------- */

/* Particle view = the LAMMPS arrays the reference functors capture
 * (ref: functor.h:86-104) flattened; all pointers [h|d] per `on_device`. */
typedef struct {
  int dim, nlocal, nall, ntypes;
  int kernel;              /* 0 Wendland, 1 Quintic, 2 Cubic (kernel_*.h)     */
  const double *x;         /* [nall][3]  atom->x                              */
  const int *type;         /* [nall]     atom->type (1-based)                 */
  const int *kind;         /* [ntypes+1] getParticleKind(type)  (host always) */
  const double *h;         /* [(ntypes+1)^2] pair->h            (host always) */
  const double *cutsq;     /* [(ntypes+1)^2] pair->cutsq        (host always) */
  const int *neigh_ptr;    /* [nlocal+1] flattened list->firstneigh           */
  const int *neigh_idx;    /* neighbour indices into [0,nall)                 */
  const int *colmap;       /* [nall] matrix column of particle j              */
  const double *vfrac;     /* [nall] atom->vfrac (NULL: computed on device)   */
  const double *Gc;        /* [nlocal][dim*dim] or NULL (AntiSymmetric family): only the rows of owned particles are
                            * read (the reference's Gc[nmax] holds nothing else: computePre fills local rows and
                            * does no forward comm of them, pair_isph_corrected.cpp:302-369)                        */
  const double *Lc;        /* [nlocal][dimL]    or NULL                         */
  /* ns.boundary == MorrisHolmes (pair_isph.h:125-132): fluid-solid pairs are weighted by
   * MirrorMorrisHolmes::computeMirrorCoefficient (mirror_morris_holmes.h:39-52) -- in the
   * divergence of the Poisson RHS and in the Laplacian of the Helmholtz matrix */
  int morris_holmes;       /* 0 = MirrorNothing                               */
  const double *pnd;       /* [nall] particle number density (pair->pnd) or NULL */
  double morris_safe_coeff;/* pair->morris_safe_coeff (default 0.43301)       */
  /* wall normals (pair->normal, [nall][3]) or NULL.  With a singular-Poisson mode other than
   * NotSingular the Solid rows receive the homogeneous-Neumann operator -dt n.grad
   * (ref: functor_incomp_navier_stokes_poisson.h:98-107, functor_gradient_dot_operator_matrix.h:39-79);
   * needs Gc.  solid_normal_diag = what A.diagonal[i] holds for such a row when the Poisson functor
   * runs: the functor does not assign it (:137-147); it is 1 after the scalar Helmholtz pass of the
   * same step, 0 if only block matrices were built. */
  const double *normal;
  double solid_normal_diag;
  /* 64-bit offsets of the flattened neighbour list, used instead of neigh_ptr when non-NULL: the reference's
   * largest configuration (bcc lattice, Quintic cut 3h: 748 neighbours x 4 M particles = 3e9 list entries,
   * sph-script/pore-scale-flow-3d.lmp) does not fit 32-bit offsets.  neigh_ptr may be NULL then. */
  const long long *neigh_ptr64;
} isph_particles;

/* Replaces PairISPH_Corrected::computePoisson -> FunctorOuterIncompNavierStokesPoisson
 * (ref: pair_isph_corrected.cpp:969-1015, functor_incomp_navier_stokes_poisson.h:52-181)
 * including FunctorOuterGraph (functor_graph.h:38-99), the Laplacian rows
 * (functor_laplacian_matrix.h:73-316) and the divergence RHS
 * (functor_divergence.h:54-124).  antisym=1 is the momentum-preserving family
 * (default, pair_isph.cpp:1779).  singular_mode: 0 NotSingular, 1 NullSpace,
 * 2 PinZero, 3 DoubleDiag (pair_isph.h:134-138).  ncol = number of matrix
 * columns (owned + ghost).  b_out: nlocal entries. */
int isph_assemble_poisson(isph_ctx *ctx, const isph_particles *P, int antisym, double dt,
                          const double *rho /*[nall]*/, const double *vstar /*[nall][3]*/,
                          int singular_mode, int is_rank0, int ncol,
                          isph_mat **A_out, double *b_out /*[h|d]*/, int on_device);
/* Replaces PairISPH_Corrected::computeHelmholtz -> FunctorOuterIncompNavierStokesHelmholtz
 * (ref: pair_isph_corrected.cpp:868-925, functor_incomp_navier_stokes_helmholtz.h:52-159) for the
 * NoBoundaryCond/HomogeneousNeumann family: A = I - theta dt (1/rho) div(nu rho grad .),
 * b = v + (1-theta) dt (1/rho) div(nu rho grad v) + dt (f/rho + g) - dt/rho grad p.
 * v, force: [nall][3] (atom->v, atom->f, ghosts included); nu, rho, pres: [nall];
 * g: 3 doubles (host).  b_out: column-major [lda x dim] like the reference's
 * b multivector (pair_isph.cpp:936-946); the result feeds isph_solve(nvec=dim).
 * A_out == NULL: only b is formed (theta = 0, where the reference copies b into x, pair_isph.cpp:964-966). */
int isph_assemble_helmholtz(isph_ctx *ctx, const isph_particles *P, int antisym, double dt, double theta,
                            const double *nu, const double *rho, const double *pres, const double *force,
                            const double *g, int incremental_pressure, const double *v, int ncol,
                            isph_mat **A_out, double *b_out /*[h|d]*/, int lda, int on_device);
/* The two scalar callers of the same solver objects, built on the same Laplacian rows:
 * PairISPH_Corrected::computeSoluteTransportSpecies -> FunctorOuterSoluteTransport (ref: pair_isph_corrected.cpp:844-861,
 * functor_solute_transport.h:47-138; solved at pair_isph.cpp:811-835):  (I - theta dt D lap) c_new = c + (1-theta) dt D lap c
 * on the rows of kind Fluid, unit rows for Solid / BufferDirichlet / BufferNeumann particles (b = c there); the Laplacian
 * couples a Fluid row to Fluid and buffer neighbours, not to Solid ones (FilterMatchBinary(Fluid, Fluid - BufferNeumann):
 * the neighbour mask is only consulted for Solid neighbours, functor_laplacian_matrix.h:143-146).  conc: [nall];
 * b_out: [nlocal].  Kinds in P->kind may be 99 Fluid, 12 Solid, 32 BufferDirichlet, 64 BufferNeumann (pair_isph.h:113-124). */
int isph_assemble_solute_transport(isph_ctx *ctx, const isph_particles *P, int antisym, double dt, double theta,
                                   double dcoeff, const double *conc /*[h|d]*/, int ncol, isph_mat **A_out,
                                   double *b_out /*[h|d]*/, int on_device);
/* PairISPH_Corrected::computeAppliedElectricPotential -> FunctorOuterAppliedElectricPotential (ref:
 * pair_isph_corrected.cpp:569-620, functor_applied_electric_potential.h:36-98; solved at pair_isph.cpp:635-657):
 * -div(sigma grad phi) = 0 on the Fluid rows (FilterMatchBinary(Fluid, Fluid)), unit rows with b = phi on the buffer
 * particles (the Dirichlet data), unit rows with b = 0 on Solid ones.  sigma: [nall] conductivity or NULL (= 1);
 * phi: [nall]; b_out: [nlocal]. */
int isph_assemble_applied_potential(isph_ctx *ctx, const isph_particles *P, int antisym, const double *sigma /*[h|d]*/,
                                    const double *phi /*[h|d]*/, int ncol, isph_mat **A_out, double *b_out /*[h|d]*/,
                                    int on_device);
/* Replaces PairISPH_Corrected::computeBlockHelmholtz -> FunctorOuterIncompNavierStokesBlockHelmholtz
 * (ref: pair_isph_corrected.cpp:941-964, functor_incomp_navier_stokes_block_helmholtz.h:57-187) with the block branch
 * of the Laplacian functor (functor_laplacian_matrix.h:269-314) and FunctorOuterBoundaryNavierSlip
 * (functor_boundary_navier_slip.h:53-184): the dim x dim blocks A_blk(ib,jb) of the velocity Helmholtz system with
 * wall rows distributed by the wall normals, exactly as the reference writes them (csrc/block_helmholtz.hpp lists
 * the steps).  normal: [nall][3] wall normals (pair->normal) or NULL; beta: ns.beta (slip length factor).
 * blocks_out[ib*dim+jb] receives block (ib,jb) on the scalar pattern; without normals the off-diagonal blocks are
 * zero and returned as NULL.  b_out: column-major [lda x dim].  The result feeds isph_solve_block. */
int isph_assemble_block_helmholtz(isph_ctx *ctx, const isph_particles *P, int antisym, double dt, double theta,
                                  double beta, const double *nu, const double *rho, const double *pres,
                                  const double *force, const double *g, int incremental_pressure, const double *v,
                                  const double *normal, int ncol, isph_mat **blocks_out, double *b_out, int lda,
                                  int on_device);
/* FunctorOuterVolume (ref: functor_volume.h:40-80); vfrac_out [nlocal]. */
int isph_compute_volumes(isph_ctx *ctx, const isph_particles *P, double *vfrac_out, int on_device);
/* Particle number density of the MorrisHolmes mirror, the `pnd` member of isph_particles: W(0) + sum of W(r_ij) over the
 * neighbours that are not of the opposite phase (FunctorOuterNormal, ref: functor_normal.h:57-133, called by
 * PairISPH_Corrected::computeNormals, pair_isph_corrected.cpp:396-419); pnd_out [nlocal], ghosts by forward comm. */
int isph_compute_pnd(isph_ctx *ctx, const isph_particles *P, double *pnd_out, int on_device);

/* FunctorOuterGradientCorrection + FunctorOuterLaplacianCorrection
 * (ref: functor_gradient_correction.h:23-71, functor_laplacian_correction.h:24-153,
 * pair_isph_corrected.cpp:333-369): G_i [nlocal][dim*dim] column-major and L_i
 * [nlocal][dimL] packed upper for the owned particles; P->vfrac must hold the
 * ghosts' volumes already (forward comm). */
int isph_compute_corrections(isph_ctx *ctx, const isph_particles *P, double *Gc_out, double *Lc_out, int on_device);

/* ---- streaming operators either side of the solve -------- */

/* Corrected::FunctorOuterGradient<.,AntiSymmetric?> (ref: functor_gradient.h:78-170): f [nall] ->
 * grad_out [nlocal][3].  use_filter/filt_i/filt_j = FilterBinary particle-kind masks (filter.h:37-54). */
int isph_gradient(isph_ctx *ctx, const isph_particles *P, int antisym, const double *f, double alpha,
                  int use_filter, int filt_i, int filt_j, double *grad_out, int on_device);
/* Corrected::FunctorOuterDivergence (ref: functor_divergence.h:54-124): f [nall][3] -> div_out [nlocal]. */
int isph_divergence(isph_ctx *ctx, const isph_particles *P, int antisym, const double *f, double alpha,
                    int use_filter, int filt_i, int filt_j, double *div_out, int on_device);
/* PairISPH_Corrected::correctVelocity + correctPressure (ref: pair_isph_corrected.cpp:1020-1050,
 * functor_correct_velocity.h, functor_correct_pressure.h): vstar [nall][3] (owned rows updated),
 * p [nall] (all updated), dp [nall] with ghost values. */
int isph_correct_velocity_pressure(isph_ctx *ctx, const isph_particles *P, int antisym, double dt,
                                   const double *rho, const double *dp, double *vstar, double *p,
                                   int incremental_pressure, int on_device);
/* PairISPH_Corrected::advanceTime (ref: pair_isph_corrected.cpp:1172-1199): begin computes
 * dp_out[nlocal] = grad p . dt/2 (vnp1+v); the caller forward-communicates it; end applies
 * p += dp, x += dt/2 (vnp1+v), v = vnp1 to the first `count` atoms. */
int isph_advance_begin(isph_ctx *ctx, const isph_particles *P, int antisym, double dt, const double *p,
                       const double *v, const double *vnp1, double *dp_out, int on_device);
int isph_advance_end(isph_ctx *ctx, int count, int dim, double dt, const double *dp, const double *vnp1,
                     double *p, double *x, double *v, int on_device);

/* SolverLin_Belos::solveBlockProblem (solver_lin_belos.h:53-128) over the blocks SolverLin::setBlock collects
 * (solver_lin.cpp:127-138): blocks[i*dim+j] is block (i,j) of the dim x dim operator (NULL = zero block, diagonal
 * blocks required), b and x are column-major [lda x dim] like the reference's multivectors, M (may be NULL) is
 * applied to every component -- the block-diagonal preconditioner PrecondWrapper_ML::create(dim) builds
 * (precond_ml.h:138-155).  Singular systems are not supported, as in the reference (:60-61). */
int isph_solve_block(isph_ctx *ctx, int dim, const isph_mat *const *blocks, const isph_prec *M, double *b, double *x,
                     int lda, const isph_solver_params *prm, isph_solve_info *info, int on_device);

/* ---- smoothed-aggregation AMG in place of PrecondWrapper_ML (precond_ml.h:40-171) ----
 * Parameters mirror the keys the wrapper sets (precond_ml.h:44-55): "max levels" 5, "aggregation: type" Uncoupled,
 * "smoother: type" symmetric Gauss-Seidel with "smoother: sweeps" 1 pre and post, direct coarse solve; plus ML's
 * defaults "coarse: max size" 128, "aggregation: damping factor" 4/3, "aggregation: threshold" 0.  `block` is the
 * row-block the Gauss-Seidel sweeps are local to (ML: the processor).  nullvec != NULL restates
 * PrecondWrapper_ML::setNullVector (precond_ml.h:97-127): one pre-computed null-space vector, smoother as the
 * coarse solver.  The hierarchy is rebuilt on every create, like the reference does on every solve.
 * More than one rank (a context with a communicator, A with its halo plan set): the call is COLLECTIVE.  Aggregates and
 * prolongator stay on the rank (Uncoupled), the coarse operators are P^T A P with the whole A -- ghost columns and a halo
 * plan per level, derived from A's -- and the coarsest systems of all ranks are solved as one (a dense inverse on every
 * rank; the smoother when nullvec is given).  Every rank must call it, with the same parameters. */
typedef struct {
  int max_levels, coarse_max;
  double omega;
  int block, sweeps;
  double theta;
  /* "smoother: type": 0 = symmetric Gauss-Seidel, `sweeps` symmetric sweeps before and after the coarse correction
   * (the wrapper's default, precond_ml.h:50-52); 1 = "ML Gauss-Seidel" / "Gauss-Seidel" with "smoother: Gauss-Seidel
   * efficient symmetric" (the ml.xml of the reference's benchmark protocol, bench-script/hopper/tgv/1728/ml.xml):
   * `sweeps` FORWARD sweeps before the coarse correction, `sweeps` BACKWARD sweeps after it.  (With a null vector the
   * coarsest level is still solved by symmetric sweeps.) */
  int smoother;
} isph_amg_params;
void isph_amg_params_default(isph_amg_params *p);
int isph_prec_create_amg(isph_ctx *ctx, const isph_mat *A, const isph_amg_params *prm, const double *nullvec,
                         int on_device, isph_prec **M);
/* test/diagnostic access: info = {rows, nnz(A_l), nnz(P_l)}; what: 0 = A_l, 1 = P_l (CSR, columns ascending) */
int isph_prec_amg_levels(const isph_prec *M);
int isph_prec_amg_info(isph_ctx *ctx, const isph_prec *M, int level, long long info[3]);
int isph_prec_amg_export(isph_ctx *ctx, const isph_prec *M, int level, int what, int *rowptr, int *colidx, double *val);
int isph_prec_amg_aggregates(isph_ctx *ctx, const isph_prec *M, int level, int *agg);

/* Particle shifting (fix isph/shift -> PairISPH_Corrected::shiftParticles, pair_isph_corrected.cpp:1203-1262).
 * isph_compute_shift replaces FunctorOuterComputeShift (functor_compute_shift.h:48-113): dr[nlocal][3] for the fluid
 * particles, pairs inside min(cutsq, shiftcut^2), alpha = shift*dt*vmax.
 * isph_apply_shift replaces FunctorOuterApplyShift (functor_apply_shift.h:76-108): p += grad p . dr,
 * v_k += grad v_k . dr, x += dr on the nlocal rows of x, v [nall][3], p [nall]; types with fixed[type] != 0
 * (PairISPH::isParticleFixed; fixed may be NULL) do not move.  All rows read the pre-shift state (the reference's
 * serial loop lets row i see rows < i already shifted).
 * isph_shift_particles is the whole shiftParticles(): vmax = max fluid |v| over all ranks (returned in *vmax_out
 * when not NULL), dr from shift*dt*vmax, then the apply. */
int isph_compute_shift(isph_ctx *ctx, const isph_particles *P, double alpha, double shiftcut, double nonfluidweight,
                       double *dr, int on_device);
int isph_apply_shift(isph_ctx *ctx, const isph_particles *P, int antisym, const int *fixed, const double *dr,
                     double *x, double *v, double *p, int on_device);
int isph_shift_particles(isph_ctx *ctx, const isph_particles *P, int antisym, const int *fixed, double shift,
                         double shiftcut, double nonfluidweight, double dt, double *x, double *v, double *p,
                         double *vmax_out, int on_device);

#ifdef __cplusplus
}
#endif
#endif
