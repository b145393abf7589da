"""TEST INFRASTRUCTURE (oracle side).  Third pin against numbers the reference holds:
sph-script/conv-channel-edl-potential-2d-morrisholmes-rev722.txt ("Wendland Kernel h = 1.2dx, cut over h = 2.0;
MorrisHolmes"), printed by fix isph/error (fix_isph_error.cpp:188-345) for sph-script/channel-edl-potential-2d.lmp:
the electric double layer in a channel |y| < 1 between two charged walls of solid particles (psi = 1 on the wall,
`set group solid isph_electric_potential_on_wall 1.0`), periodic in x, linearised Poisson-Boltzmann
-lap_h psi + kappa^2 psi = 0 with kappa^2 = 2 ezcb / psiref = 100 (channel-edl-potential.xml), analytic solution
psi = cosh(kappa y) / cosh(kappa) (the xml's "Linear version"; sol.psi.norm2 of the table is that function on the
lattice to all 16 digits).  Fluid rows: the corrected (Symmetric-family) Laplacian with the MorrisHolmes mirror
coefficient on fluid-solid pairs (functor_boundary_morris_holmes.h:49-67, mirror_morris_holmes.h:39-52), whose
distances come from the particle number density pnd (functor_normal.h:57-133); solid rows: psi = psi0
(functor_poisson_boltzmann_f.h:66-72).  On this lattice sum_j a_ij e_ij = 0, so the operator form the reference
evaluates F with and the matrix functor of the pressure path (functor_laplacian_matrix.h:144-146) give the
same rows; the table pins mirror coefficient + pnd + Dirichlet solid columns of that functor.  The file's second
section, "ConstExtension", is the same problem without the mirror (the wall value extended into the solid).

    total # of particles = fluid particles,  total volume = sum of V_i over them
    err.psi.norm2 = sqrt( sum_fluid (psi_i - psi_exact)^2 / n_fluid )

A fourth, weaker pin: conv-channel-edl-potential-2d-morrisholmes-rev406.txt, the same channel at an earlier revision with
"h = 1.02 dx".  With that h the total volume of its rows N = 32 and 64 comes out to 15 digits (kernel + volume functor in
another h/dx regime) and err.psi.norm2 to 4 (4.2e-5 and 2.6e-4 relative: that revision's solve is not the one rev722
records to 10 digits); from N = 128 on the table's volumes leave the exact scale invariance of the lattice in the 8th
digit and earlier, i.e. its h was no longer exactly 1.02 dx, and those rows are not used.

usage: python oracle/pb_channel.py [N ...]"""
import os
import sys

import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spla

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [_HERE, os.path.join(_HERE, "..")]
import oracle as orc  # noqa: E402

KAPPA = 10.0                      # sqrt(2 ezcb / psiref / eps), ezcb = 50
KINDS = [orc.FLUID, orc.SOLID, orc.FLUID]     # type 1 flow, 2 wall ("solid:fixed"), 3 near-wall fluid


def known_answers(boundary="MorrisHolmes"):
    """rows of the table's "MorrisHolmes" or "ConstExtension" section (data); "rev406": the earlier table with h = 1.02 dx"""
    import json
    g = json.load(open(os.path.join(_HERE, "..", "tests", "golden", "reference_known_answers.json")))
    if boundary == "rev406":
        return {int(k): v for k, v in g["conv_channel_edl_potential_2d_morrisholmes_rev406"]["rows"].items()}
    key = "rows" if boundary == "MorrisHolmes" else "rows_const_extension"
    return {int(k): v for k, v in g["conv_channel_edl_potential_2d_morrisholmes_rev722"][key].items()}


def channel(N, h_over_dx=1.2):
    """atoms of channel-edl-potential-2d.lmp: lattice sq dx origin 0.5 0.5 in [-len, len) x [-(1 + 6 dx), 1 + 6 dx),
    dx = 2 / N, len = round(0.2 N) / N; periodic images as ghosts; full neighbour list, ascending atom index per row."""
    from scipy.spatial import cKDTree
    Nx = int(round(N * 0.2))
    length, dx = Nx / N, 2.0 / N
    wall = 6 * dx
    h = h_over_dx * dx
    cut = 2.0 * h
    s = (np.arange(-4 * N, 4 * N) + 0.5) * dx
    xs = s[(s >= -length - 1e-12) & (s < length - 1e-12)]
    ys = s[(s >= -(1 + wall) - 1e-12) & (s < (1 + wall) - 1e-12)]
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    x = np.stack([X.ravel(), Y.ravel(), np.zeros(X.size)], axis=1)
    n = len(x)
    Lx, Ly = 2 * length, 2 * (1 + wall)
    xall, own = [x], [np.arange(n)]
    for sx in (-1, 0, 1):
        for sy in (-1, 0, 1):
            if sx == 0 and sy == 0:
                continue
            xi = x + np.array([sx * Lx, sy * Ly, 0.0])
            keep = (np.abs(xi[:, 0]) < length + cut) & (np.abs(xi[:, 1]) < 1 + wall + cut)
            xall.append(xi[keep])
            own.append(np.nonzero(keep)[0])
    xall, own = np.concatenate(xall), np.concatenate(own)
    nall = len(xall)
    tree = cKDTree(xall[:, :2])
    D = tree.sparse_distance_matrix(cKDTree(xall[:n, :2]), cut * (1 + 1e-9), output_type="coo_matrix")
    i, j = D.col, D.row                                     # i owned, j any
    d = xall[i, :2] - xall[j, :2]
    keep = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] < cut * cut) & (i != j)
    i, j = i[keep], j[keep]
    order = np.lexsort((j, i))
    i, j = i[order], j[order]
    ptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(ptr, i + 1, 1)
    ptr = np.cumsum(ptr)
    ay = np.abs(xall[:, 1])
    typ = np.where(ay > 1.0, 2, np.where(ay > 1.0 - cut, 3, 1)).astype(np.int32)

    class _Spec:
        rank = 0
    parts = dict(spec=_Spec(), dim=2, nlocal=n, nall=nall, x=np.ascontiguousarray(xall), type=typ,
                 tag=(own + 1).astype(np.int32), owner_rank=np.zeros(nall, np.int32), owner_index=own.astype(np.int32),
                 neigh_ptr=ptr.astype(np.int32), neigh_idx=j.astype(np.int32), h=h, cut=cut, kinds=KINDS)
    return parts, own


def exact(y):
    return np.cosh(KAPPA * y) / np.cosh(KAPPA)


def solve_rows(rp, ci, val, typ_local):
    """psi of  (A + kappa^2 I) psi = 0 on the fluid rows, psi = 1 on the solid rows (host direct solve)"""
    n = len(rp) - 1
    A = sps.csr_matrix((val, ci, rp), shape=(n, n))
    fl, so = np.nonzero(typ_local != 2)[0], np.nonzero(typ_local == 2)[0]
    Aff, Afs = A[fl][:, fl], A[fl][:, so]
    psi = np.ones(n)
    psi[fl] = spla.spsolve((Aff + KAPPA ** 2 * sps.eye(len(fl))).tocsc(), -(Afs @ np.ones(len(so))))
    return psi, fl


def run(N, boundary="MorrisHolmes", h_over_dx=1.2):
    """boundary "ConstExtension": the wall particles carry psi0 and enter the rows like any neighbour (no mirror,
    pair_isph_corrected.cpp:451-460).  h_over_dx = 1.02: the setting of conv-channel-edl-potential-2d-morrisholmes-rev406.txt"""
    parts, own = channel(N, h_over_dx)
    n, nall = parts["nlocal"], parts["nall"]
    P0 = orc.Particles(parts, own, kernel="wendland", kinds=KINDS)
    pnd = P0.compute_pnd()
    P = orc.Particles(parts, own, kernel="wendland", kinds=KINDS, pnd=pnd, morris_safe_coeff=0.0)   # MorrisSafeCoeff of the xml
    P.precompute(corrections=True)
    rp, ci = P.graph()
    val = P.laplacian_matrix(rp, ci, antisym=False, alpha=-1.0, material=np.ones(nall), filt=(orc.FLUID, orc.ALL),
                             morris=1 if boundary == "MorrisHolmes" else 0)
    psi, fl = solve_rows(rp, ci, val, parts["type"][:n])
    ex = exact(parts["x"][fl, 1])
    return dict(N=N, particles=len(fl), volume=float(P.vfrac[fl].sum()), sol_psi=float(np.sqrt(np.mean(ex ** 2))),
                err_psi=float(np.sqrt(np.mean((psi[fl] - ex) ** 2))), xi_min=float((pnd[:n] * P.vfrac[:n])[fl].min()))


if __name__ == "__main__":
    for boundary in ("MorrisHolmes", "ConstExtension"):
        ref = known_answers(boundary)
        print(boundary)
        for N in [int(a) for a in sys.argv[1:]] or [32, 64, 128, 256]:
            r = run(N, boundary)
            print("  N = %d   %d fluid particles (table %d), smallest own-phase fraction of a fluid particle %.3f" % (N, r["particles"], ref[N]["particles"], r["xi_min"]))
            for name, key in (("total volume", "volume"), ("sol.psi.norm2", "sol_psi"), ("err.psi.norm2", "err_psi")):
                print("      %-16s oracle %.15e   reference %.15e   rel. diff %.2e" % (name, r[key], ref[N][key], abs(r[key] - ref[N][key]) / ref[N][key]))
