"""-m gpu: BASELINE configs[4] -- pore-scale flow through a bead pack in a cylinder (sph-script/pore-scale-flow-3d.lmp,
compute_isph_cylinder_porous.cpp:195-224): bcc lattice, Quintic kernel cut 3h (749 entries per row), MorrisHolmes
boundary, NotSingular Poisson, SA-AMG preconditioner.  Oracle parity on a 43 904-particle cylinder, size-independent
properties on 1.02 M particles (767 M matrix entries), on 2.96 M particles (2.22e9 entries: 64-bit offsets) and at the
configuration's own size, 4.0 M particles (3.0e9 entries, 36 GB of sliced-ELL on one GPU)."""
import numpy as np
import pytest
import scipy.sparse as sps

from isph_amd import hip, workload
import oracle as orc

from sampled_check import assert_spmv_matches_host_on_sampled_rows

pytestmark = pytest.mark.gpu


def test_porous_small_matches_oracle(gpu_ctx):
    p = workload.make_porous_cylinder(28, brick=(4, 4, 4), jitter=0.02)
    n, nall = p["nlocal"], p["nall"]
    colmap = workload.single_rank_colmap(p)
    assert np.diff(p["neigh_ptr"]).max() >= 700                       # bcc + cut 4.5 dx
    assert all((p["type"][:n] == t).sum() > 0 for t in (1, 2, 3, 4))
    P0 = orc.Particles(p, colmap, kernel="quintic", kinds=p["kinds"])
    P0.precompute(corrections=False)
    pnd = np.ascontiguousarray(1.0 / P0.vfrac)                        # particle number density of the mirror formula
    P = orc.Particles(p, colmap, kernel="quintic", kinds=p["kinds"], pnd=pnd, morris_safe_coeff=0.43301)
    P.precompute(corrections=False)
    vstar = np.zeros((nall, 3))
    xw = p["x"]
    vstar[:, 1] = 1e-3 * np.cos(xw[:, 0]) * (p["type"] <= 2)          # some divergence-free-ish fluid motion along the axis
    vstar[:, 0] = 1e-3 * np.sin(xw[:, 1]) * (p["type"] <= 2)
    rp, ci, val, b = P.poisson(p["dt"], p["rho"], vstar, singular=orc.NOT_SINGULAR, morris=1)
    vf = hip.compute_volumes(gpu_ctx, p, colmap, kernel="quintic")
    assert np.max(np.abs(vf - P.vfrac[:n])) < 1e-13 * np.abs(P.vfrac).max()
    A, bg = hip.assemble_poisson(gpu_ctx, p, colmap, p["dt"], p["rho"], vstar, singular=hip.NOT_SINGULAR, vfrac=P.vfrac,
                                 kernel="quintic", kinds=p["kinds"], pnd=pnd)
    rg, cg, vg = A.export_csr()
    assert np.array_equal(rg, rp) and np.array_equal(cg, ci)
    assert np.max(np.abs(vg - val)) <= 1e-12 * np.abs(val).max()
    assert np.max(np.abs(bg - b)) <= 1e-12 * np.abs(b).max()
    solid = p["type"][:n] >= 3
    d = sps.csr_matrix((vg, cg, rg)).diagonal()
    assert np.all(d[solid] == 1.0) and np.all(bg[solid] == 0.0)
    # SA-AMG (PrecondWrapper_ML defaults, no null vector: the system is not singular) vs the oracle's hierarchy
    prm = hip.AmgParams(block=512, coarse_max=128)
    M = hip.PrecondAMG(gpu_ctx, A, params=prm)
    G = orc.AMG(rp, ci, val, block=512, coarse_max=128)
    assert M.levels == G.levels
    x = np.zeros(n)
    info = hip.solve(gpu_ctx, A, bg.copy(), x, prec=M, singular=False)
    xo, io, _ = orc.solve(rp, ci, val, b, singular=False, prec="amg", amg=G)
    assert info.converged == 1 and io.converged == 1 and abs(info.iters - io.iters) <= 1
    assert np.linalg.norm(x - xo) <= 1e-6 * np.linalg.norm(xo)


@pytest.mark.parametrize("nc,reference_beads", [(80, False), (114, False), (112, True)])
def test_porous_config4_properties_at_size(gpu_ctx, nc, reference_beads):
    """nc = 80: 2 x 80^3 = 1 024 000 particles, 749 entries per row (767 M entries, 9 GB of sliced-ELL).
    nc = 114: 2 x 114^3 = 2 963 088 particles, 2.22e9 matrix entries -- beyond 2^31: the neighbour list goes in through
    isph_particles::neigh_ptr64 and the AMG set-up / Gauss-Seidel stream run on 64-bit offsets.
    nc = 112 with the reference's own bead pack (tests/golden/pore_scale_flow_bead_centeroids_3d.npz = the script's
    pore-scale-flow-bead-centeroids-3d.dat: 3807 beads in the middle half of a cylinder of aspect ratio 1.634):
    2 x 112 x 168 x 112 = 4 214 784 particles, 3.16e9 entries -- BASELINE configs[4] at its own size on ONE GPU.
    Assembled on the device from torch-resident arrays and solved with FGMRES + SA-AMG.  Properties: row length of the
    bcc/Quintic stencil, solid rows are identity rows, the solve converges, residual <= 2e-8 re-computed with an
    independent SpMV, zero pressure on the solid rows."""
    import torch
    dev = torch.device("cuda", 0)
    if reference_beads:
        import os
        bp = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pore_scale_flow_bead_centeroids_3d.npz"))
        p = workload.make_porous_cylinder(nc, bead_pack=bp)
        assert p["spec"].ncell == (112, 168, 112) and len(np.unique(p["part"][p["type"] == 3])) == 3807
    else:
        p = workload.make_porous_cylinder(nc, nbeads=40, rbead_cells=6.0)
    n, nall = p["nlocal"], p["nall"]
    assert n == 2 * int(np.prod(p["spec"].ncell))
    assert (p["neigh_ptr"].dtype == np.int64) == (n * 748 >= 2 ** 31)      # 64-bit list offsets exactly when needed
    torch.cuda.empty_cache()
    colmap_h = workload.single_rank_colmap(p)
    dp = dict(p)
    for k in ("x", "type", "neigh_ptr", "neigh_idx"):
        dp[k] = torch.from_numpy(np.ascontiguousarray(p[k])).to(dev)
    colmap = torch.from_numpy(colmap_h).to(dev)
    own = torch.from_numpy(p["owner_index"].astype(np.int64)).to(dev)
    rho = torch.from_numpy(p["rho"]).to(dev)
    vf = hip.compute_volumes(gpu_ctx, dp, colmap, kernel="quintic")
    vfrac = vf[own].contiguous()
    pnd = (1.0 / vfrac).contiguous()
    x = dp["x"]
    fluid = (dp["type"] <= 2).to(torch.float64)
    vstar = torch.zeros((nall, 3), dtype=torch.float64, device=dev)
    vstar[:, 1] = 1e-3 * torch.cos(x[:, 0]) * fluid
    vstar[:, 0] = 1e-3 * torch.sin(x[:, 1]) * fluid
    A, b = hip.assemble_poisson(gpu_ctx, dp, colmap, p["dt"], rho, vstar, singular=hip.NOT_SINGULAR, vfrac=vfrac,
                                kernel="quintic", kinds=p["kinds"], pnd=pnd)
    im = A.info()
    assert im["nrow"] == n and 745 < im["nnz"] / n <= 749
    del dp["neigh_idx"], dp["neigh_ptr"]                       # the list (12 GB at 4 M particles) is not needed any more
    p.pop("neigh_idx")
    torch.cuda.empty_cache()
    solid = dp["type"][:n] >= 3
    e = torch.zeros(n, dtype=torch.float64, device=dev)
    e[solid] = 1.0
    y = A.spmv(e)                                              # identity rows: (A e)_i = 1 on the solid rows themselves
    assert float((y[solid] - 1.0).abs().max()) == 0.0 and float(b[solid].abs().max()) == 0.0
    M = hip.PrecondAMG(gpu_ctx, A, params=hip.AmgParams(block=512))
    assert M.levels >= 2
    xs = torch.zeros(n, dtype=torch.float64, device=dev)
    bw = b.clone()
    info = hip.solve(gpu_ctx, A, bw, xs, prec=M, singular=False)
    assert info.converged == 1
    ax = A.spmv(xs)
    r = b - ax
    assert float(r.norm() / b.norm()) < 2e-8
    # the residual above trusts the kernel under test: 4 096 rows of the operator are exported and multiplied on the
    # host (nothing shared with the SpMV kernels) -- they must give the device's product, and a small residual
    rows, axh = assert_spmv_matches_host_on_sampled_rows(A, xs, ax)
    bh = b.cpu().numpy()
    assert np.linalg.norm(bh[rows] - axh) <= 2e-8 * float(b.norm()) * np.sqrt(len(rows) / n) * 20
    assert float(xs[solid].abs().max()) <= 1e-12 * float(xs.abs().max())
    del M, A
    torch.cuda.empty_cache()
    hip.pool_trim()                                            # hand the tens of GB back before the next test module
    assert hip.pool_cached_bytes() == 0
