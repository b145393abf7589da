"""-m gpu: SolverLin_Belos::solveBlockProblem -- GMRES over the dim x dim blocked operator with
the block-diagonal preconditioner (one ILU(0) or one AMG operator applied to every component), against the
oracle solving the same system assembled as one CSR."""
import numpy as np
import pytest
import scipy.sparse as sps

from isph_amd import hip, workload
import oracle as orc
from problems import Problem, tgv_spec

pytestmark = pytest.mark.gpu


def _blocks(dim, n):
    """diagonal blocks: the scalar Helmholtz operator (theta = 0.5); off-diagonal blocks: a weak unsymmetric
    coupling on the same pattern (what the wall terms of the block Helmholtz functor produce); block (0,dim-1)
    is left empty to exercise NULL blocks."""
    pr = Problem(tgv_spec(dim=dim, n=n, mode=workload.JITTER, brick=4))
    p = pr.parts
    nall = p["nall"]
    vel = np.ascontiguousarray(p["v"])
    rp, ci, val, b = pr.P.helmholtz(pr.spec.dt, 0.5, p["nu"], p["rho"], np.zeros(nall), np.zeros((nall, 3)),
                                    np.zeros(3), vel)
    H = sps.csr_matrix((val, ci, rp), shape=(pr.n, pr.n))
    off = H - sps.diags(H.diagonal())
    blocks = [[None] * dim for _ in range(dim)]
    for i in range(dim):
        for j in range(dim):
            if i == j:
                blocks[i][j] = (H * (1.0 + 0.1 * i)).tocsr()
            elif not (i == 0 and j == dim - 1):
                blocks[i][j] = (off * (0.15 + 0.05 * (i - j))).tocsr()
    rhs = np.stack([np.cos(p["x"][:pr.n, 0] + k) + 0.2 * k for k in range(dim)])
    return pr, H.tocsr(), blocks, rhs


@pytest.mark.parametrize("dim,n,prec", [(2, 24, "ilu"), (3, 12, "ilu"), (3, 12, "amg"), (3, 10, "none")])
def test_block_solve_matches_oracle(gpu_ctx, dim, n, prec):
    pr, H, blocks, rhs = _blocks(dim, n)
    nl = pr.n
    big = sps.bmat([[blocks[i][j] for j in range(dim)] for i in range(dim)], format="csr")
    big.sort_indices()
    bs = 256
    ilu = amg = M = None
    A00 = hip.Matrix.from_csr(gpu_ctx, H.indptr, H.indices, H.data)
    if prec == "ilu":
        bp = np.arange(0, nl + bs, bs).clip(0, nl).astype(np.int32)
        ilu = orc.ILU(H.indptr, H.indices, H.data, 0, bp)
        M = hip.Precond(gpu_ctx, A00, "bjacobi-ilu0", bs)
    elif prec == "amg":
        kw = dict(theta=0.02, block=bs, coarse_max=64)
        amg = orc.AMG(H.indptr, H.indices, H.data, **kw)
        M = hip.PrecondAMG(gpu_ctx, A00, params=hip.AmgParams(**kw))
    xo, io = orc.solve_block(big.indptr, big.indices, big.data, rhs.ravel(), dim, prec=prec, ilu=ilu, amg=amg)
    mats = [[None if blocks[i][j] is None else
             hip.Matrix.from_csr(gpu_ctx, blocks[i][j].indptr, blocks[i][j].indices, blocks[i][j].data)
             for j in range(dim)] for i in range(dim)]
    lda = nl + 7                                             # padded leading dimension like a strided multivector
    b = np.zeros((dim, lda)); b[:, :nl] = rhs
    x = np.zeros((dim, lda))
    info = hip.solve_block(gpu_ctx, mats, b, x, prec=M, lda=lda)
    assert info.converged == 1 and io.converged == 1 and abs(info.iters - io.iters) <= 1
    xg = x[:, :nl].ravel()
    assert np.linalg.norm(xg - xo) <= 1e-6 * np.linalg.norm(xo)
    assert np.linalg.norm(big @ xg - rhs.ravel()) <= 1e-7 * np.linalg.norm(rhs)
    assert not x[:, nl:].any()                               # padding rows of the multivector untouched


def test_block_solve_rejects_missing_diagonal(gpu_ctx):
    pr, H, blocks, rhs = _blocks(2, 16)
    A = hip.Matrix.from_csr(gpu_ctx, H.indptr, H.indices, H.data)
    with pytest.raises(hip.IsphError):
        hip.solve_block(gpu_ctx, [[A, A], [A, None]], rhs.copy(), np.zeros_like(rhs))


def test_cpp_solve_block_problem_mirror(tmp_path):
    """SolverLin_Belos::createBlockMatrix / setBlock / solveBlockProblem + PrecondWrapper_ML::create(dim)
    (implicit-sph_amd/host/*.h) driven like pair_isph.cpp:944-972 drives the reference."""
    import subprocess
    from isph_amd import build
    exe = build.build_cpp_test()
    dim = 3
    pr, H, blocks, rhs = _blocks(dim, 12)
    nl = pr.n
    fin, fout = tmp_path / "blk.bin", tmp_path / "x.bin"
    with open(fin, "wb") as f:
        np.array([nl, H.nnz], np.int32).tofile(f)
        H.indptr.astype(np.int32).tofile(f); H.indices.astype(np.int32).tofile(f); H.data.tofile(f)
        np.array([dim], np.int32).tofile(f)
        for i in range(dim):
            for j in range(dim):
                Bm = blocks[i][j]
                if Bm is None:
                    np.array([0], np.int32).tofile(f)
                    continue
                Bm.sort_indices()
                np.array([1, Bm.nnz], np.int32).tofile(f)
                Bm.indptr.astype(np.int32).tofile(f); Bm.indices.astype(np.int32).tofile(f); Bm.data.tofile(f)
        rhs.tofile(f)
    r = subprocess.run([exe, str(fin), str(fout), "0", "block"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert ">> Belos(Block)::Label - Block 3x3 Helmholtz" in r.stdout and ">> Belos::Status - Passed!" in r.stdout
    x = np.fromfile(fout)
    big = sps.bmat([[blocks[i][j] for j in range(dim)] for i in range(dim)], format="csr")
    big.sort_indices()
    amg = orc.AMG(H.indptr, H.indices, H.data, theta=0.02, block=256, coarse_max=64)
    xo, io = orc.solve_block(big.indptr, big.indices, big.data, rhs.ravel(), dim, prec="amg", amg=amg)
    assert np.linalg.norm(x - xo) <= 1e-6 * np.linalg.norm(xo)
