"""GCRO-DR(m, k): the oracle's restatement of Belos::GCRODRSolMgr -- TEST INFRASTRUCTURE, numpy on top of the C oracle.

The reference selects it with "Solver Type" = "Recycling GMRES" (solver_lin_belos.h:178-179) and the keys "Num Blocks"
(m) and "Num Recycled Blocks" (k) of setParameters (:224-264).  A new solver manager is created for every solve
(:161-184), so nothing is recycled BETWEEN solves: what remains is GCRO-DR inside one solve -- a GMRES(m) cycle, then
cycles of m - k Arnoldi steps on (I - C C^T) A M^-1 with the k harmonic Ritz vectors of smallest magnitude carried as the
recycle space (U, C = A M^-1 U, C^T C = I).  Belos (Trilinos, not vendored, no version pinned) implements Parks, de
Sturler, Mackey, Johnson, Maiti, "Recycling Krylov subspaces for sequences of linear systems", SIAM J. Sci. Comput. 28
(2006), Algorithm GCRO-DR; that published algorithm is what is restated:

  r0 = b - Op x0,  Op = P_n A M^-1 (right preconditioning, PoissonProjection as in solveProblem, :130-222)
  cycle 1: m Arnoldi steps V_{m+1}, Hbar; y = argmin ||beta e1 - Hbar y||; t += V_m y
           harmonic Ritz pairs of H_m:  (H_m + h_{m+1,m}^2 H_m^{-T} e_m e_m^T) z = theta z,
           P = the k of smallest |theta| (a complex pair enters as Re z, Im z; when the k-th slot would split a pair
           k+1 vectors are kept, like GCRODRSolMgr::getHarmonicVecs1),  [Q,R] = qr(Hbar P), C = V_{m+1} Q, U = V_m P R^-1
  cycle > 1: v1 = r/||r||; p = m - k Arnoldi steps with w <- w - C (C^T w), B = C^T A M^-1 V_p
           D = diag(1/||u_i||), Ut = U D, G = [[D, B], [0, Hbar]], W = [C V_{p+1}], Vh = [Ut V_p]  (A M^-1 Vh = W G)
           y = argmin ||W^T r - G y||; t += Vh y; r -= W G y
           harmonic Ritz pairs:  G^T G z = theta G^T W^T Vh z, k smallest |theta| -> P;  [Q,R] = qr(G P), C = W Q,
           U = Vh P R^-1
  x = x0 + M^-1 t.  One iteration = one Arnoldi step; convergence = implicit residual / ||r0|| <= tol, tested every
  step (the least-squares residual of the Hbar part: the recycle block can always be zeroed by its own unknowns).
"""
import numpy as np
import scipy.linalg as sla

import oracle as orc


def _real_basis(theta, Z, k):
    """k (or k+1) real vectors spanning the eigenvectors of smallest |theta|; conjugate pairs stay together."""
    order = np.argsort(np.abs(theta), kind="stable")
    used = np.zeros(len(theta), dtype=bool)
    cols = []
    for idx in order:
        if used[idx] or len(cols) >= k:
            continue
        used[idx] = True
        z = Z[:, idx]
        lam = theta[idx]
        if abs(lam.imag) <= 1e-12 * max(abs(lam), 1e-300):
            j = np.argmax(np.abs(z))
            cols.append((z / z[j]).real)
        else:
            cand = np.nonzero(~used)[0]
            if len(cand):
                partner = cand[np.argmin(np.abs(theta[cand] - np.conj(lam)))]
                used[partner] = True
            cols.append(z.real.copy())
            cols.append(z.imag.copy())          # k+1 vectors when the pair straddles the k-th slot
    P = np.stack(cols, axis=1)
    return P / np.linalg.norm(P, axis=0)


def solve(rowptr, colidx, val, b, x0=None, singular=False, null_mask=None, prec=None, num_blocks=50, num_recycled=20,
          tol=1e-8, max_iters=500, max_restarts=15):
    """prec: callable r -> M^-1 r (or None).  Returns (x, dict(converged, iters, restarts, rel_res))."""
    n = len(rowptr) - 1
    m, k = int(num_blocks), int(num_recycled)
    assert 0 < k < m
    b = np.array(b, dtype=np.float64)
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    nvec = None
    if singular:
        nvec = np.ones(n) if null_mask is None else np.asarray(null_mask, dtype=np.float64).copy()
        nvec /= np.linalg.norm(nvec)
        b -= (b @ nvec) * nvec
    Minv = (lambda r: r) if prec is None else prec

    def A(v):
        y = orc.spmv(rowptr, colidx, val, v)
        return y - (y @ nvec) * nvec if singular else y

    def op(v):
        return A(Minv(v))

    r = b - A(x)
    beta0 = np.linalg.norm(r)
    scale = beta0 if beta0 > 0 else 1.0
    info = dict(converged=beta0 / scale <= tol, iters=0, restarts=0, rel_res=beta0 / scale)
    t = np.zeros(n)
    U = C = None

    def arnoldi(r, steps, C):
        """returns V (n x (j+1)), Hbar ((j+1) x j), B (k x j), j, converged"""
        beta = np.linalg.norm(r)
        V = np.zeros((n, steps + 1))
        H = np.zeros((steps + 1, steps))
        B = np.zeros((0 if C is None else C.shape[1], steps))
        V[:, 0] = r / beta
        j = 0
        conv = False
        while j < steps:
            w = op(V[:, j])
            if C is not None:
                B[:, j] = C.T @ w
                w = w - C @ B[:, j]
            for _ in range(2):                                 # two classical Gram-Schmidt passes
                c = V[:, :j + 1].T @ w
                w = w - V[:, :j + 1] @ c
                H[:j + 1, j] += c
            H[j + 1, j] = np.linalg.norm(w)
            V[:, j + 1] = w / H[j + 1, j]
            j += 1
            info["iters"] += 1
            e1 = np.zeros(j + 1)
            e1[0] = beta
            y, *_ = np.linalg.lstsq(H[:j + 1, :j], e1, rcond=None)
            info["rel_res"] = np.linalg.norm(e1 - H[:j + 1, :j] @ y) / scale
            if info["rel_res"] <= tol:
                conv = True
                break
            if info["iters"] >= max_iters:
                break
        return V[:, :j + 1], H[:j + 1, :j], B[:, :j], j, conv, beta

    while not info["converged"] and info["iters"] < max_iters:
        if U is None:
            V, H, _, j, conv, beta = arnoldi(r, m, None)
            e1 = np.zeros(j + 1)
            e1[0] = beta
            y, *_ = np.linalg.lstsq(H, e1, rcond=None)
            t += V[:, :j] @ y
            r = V @ (e1 - H @ y)
            if conv:
                info["converged"] = True
                break
            if info["iters"] >= max_iters or info["restarts"] >= max_restarts:
                break
            Hm = H[:j, :j]
            em = np.zeros(j)
            em[-1] = 1.0
            f = np.linalg.solve(Hm.T, em)
            theta, Z = np.linalg.eig(Hm + H[j, j - 1] ** 2 * np.outer(f, em))
            P = _real_basis(theta, Z, k)
            Q, R = np.linalg.qr(H @ P)
            C = V @ Q
            U = np.linalg.solve(R.T, (V[:, :j] @ P).T).T
        else:
            kk = U.shape[1]
            V, H, B, j, conv, beta = arnoldi(r, m - kk, C)
            d = 1.0 / np.linalg.norm(U, axis=0)
            Ut = U * d
            G = np.zeros((kk + j + 1, kk + j))
            G[:kk, :kk] = np.diag(d)
            G[:kk, kk:] = B
            G[kk:, kk:] = H
            W = np.concatenate([C, V], axis=1)
            Vh = np.concatenate([Ut, V[:, :j]], axis=1)
            rhs = np.concatenate([C.T @ r, np.eye(j + 1)[:, 0] * beta])
            y, *_ = np.linalg.lstsq(G, rhs, rcond=None)
            t += Vh @ y
            r = r - W @ (G @ y)
            if conv:
                info["converged"] = True
                break
            if info["iters"] >= max_iters or info["restarts"] >= max_restarts:
                break
            WtV = np.zeros((kk + j + 1, kk + j))
            WtV[:kk, :kk] = C.T @ Ut
            WtV[kk:, :kk] = V.T @ Ut
            WtV[kk:kk + j, kk:] = np.eye(j)
            theta, Z = sla.eig(G.T @ G, G.T @ WtV)
            P = _real_basis(theta, Z, k)
            Q, R = np.linalg.qr(G @ P)
            C = W @ Q
            U = np.linalg.solve(R.T, (Vh @ P).T).T
        info["restarts"] += 1
    x = x + Minv(t)
    if singular:
        x -= (x @ nvec) * nvec
    return x, info
