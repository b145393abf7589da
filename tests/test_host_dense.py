"""CPU: the host-side dense kernels of the Recycling GMRES path (csrc/dense_small.hpp: complex Hessenberg-QR
eigen-solver, Householder QR, least squares) against numpy/LAPACK, on random matrices and on the kind of matrix
GCRO-DR feeds it (an upper Hessenberg matrix with a rank-one update)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("dense") / "test_dense_small")
    src = os.path.join(ROOT, "tests", "cpp", "test_dense_small.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "implicit-sph_amd", "csrc"), "-o", out, src],
                   check=True)
    return out


def _run(exe, A):
    n = A.shape[0]
    txt = "%d\n" % n + "\n".join(" ".join("%.17g" % v for v in row) for row in A) + "\n"
    r = subprocess.run([exe], input=txt, capture_output=True, text=True, check=True)
    lines = r.stdout.split("\n")
    worst = float(lines[0])
    lam = np.array([complex(*map(float, l.split())) for l in lines[1:1 + n]])
    qerr, rerr = map(float, lines[1 + n].split())
    y = np.array([float(l) for l in lines[2 + n:2 + n + n - 1]])
    return worst, lam, qerr, rerr, y


@pytest.mark.parametrize("n,kind", [(6, "random"), (25, "random"), (50, "hessenberg"), (40, "symmetric"), (12, "defective")])
def test_dense_small_against_numpy(exe, n, kind):
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n))
    if kind == "hessenberg":                               # GCRO-DR's first harmonic Ritz problem
        H = np.triu(A, -1) + 3 * np.eye(n)
        f = np.linalg.solve(H.T, np.eye(n)[:, -1])
        A = H + 0.3 ** 2 * np.outer(f, np.eye(n)[:, -1])
    elif kind == "symmetric":
        A = A + A.T
    elif kind == "defective":
        A = np.triu(A)
        A[np.arange(n), np.arange(n)] = np.repeat(np.arange(1, n // 2 + 1), 2)[:n]     # repeated eigenvalues
    worst, lam, qerr, rerr, y = _run(exe, A)
    scale = np.abs(A).max()
    assert worst < 1e-9 * scale * n
    ref = np.linalg.eigvals(A)
    for l in lam:                                          # same spectrum (matched greedily)
        assert np.min(np.abs(ref - l)) < 1e-6 * max(1.0, scale)
    assert qerr < 1e-12 and rerr < 1e-12 * scale
    yo, *_ = np.linalg.lstsq(A[:, :n - 1], 1.0 + np.arange(n), rcond=None)
    assert np.linalg.norm(y - yo) < 1e-8 * max(np.linalg.norm(yo), 1.0)
