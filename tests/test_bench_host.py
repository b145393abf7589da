"""-m "not gpu": the host-side logic of bench.py that needs no device -- the brick grids of the weak-scaling runs, SURVEY's
algorithmic byte count, the command line, and the self-launch of --gpus N (children started before anything touches a
GPU; a failing rank ends the run with its exit code instead of leaving the others in a collective)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_brick_grids_and_algorithmic_bytes():
    assert bench.pgrid_for(1) == (1, 1, 1) and bench.pgrid_for(2) == (2, 1, 1)
    assert bench.pgrid_for(4) == (2, 2, 1) and bench.pgrid_for(8) == (2, 2, 2)
    # 10^6 rows x 93 -> 1.136 GB
    assert bench.spmv_algorithmic_bytes(10 ** 6, 93 * 10 ** 6 + 10 ** 6) == 12 * 94 * 10 ** 6 + 16 * 10 ** 6 + 4 * (10 ** 6 + 1)
    assert abs(bench.spmv_algorithmic_bytes(10 ** 6, 93 * 10 ** 6) - 1.136e9) < 1e6


def test_every_weak_scaling_brick_has_the_same_lattice():
    """(2N, N, N) and (2N, 2N, N) boxes keep the spacing and the stencil of the single brick (they were stretched once)"""
    import numpy as np
    import isph_amd  # noqa: F401
    from isph_amd import workload
    ref = None
    for world in (1, 2, 4, 8):
        pg = bench.pgrid_for(world)
        spec = workload.TGVSpec(dim=3, ncell=(16 * pg[0], 16 * pg[1], 16 * pg[2]), pgrid=pg, rank=world - 1, brick=(8, 8, 8),
                                mode=workload.LATTICE)
        p = workload.make_tgv(spec)
        n = p["nlocal"]
        assert n == 16 ** 3
        x = p["x"][:n]
        d = [np.diff(np.unique(np.round(x[:, a] / spec.dx, 6))) for a in range(3)]
        assert all(np.allclose(k, 1.0) for k in d)                    # one spacing in every direction
        per_row = np.diff(p["neigh_ptr"]).astype(np.int64)           # (the 30 sites exactly on the cut radius round either way)
        ref = per_row.mean() if ref is None else ref
        assert per_row.min() >= 90 and per_row.max() <= 122 and abs(per_row.mean() - ref) <= 0.08 * ref


def test_self_launch_propagates_a_failing_rank():
    """`python bench.py --gpus 2` here (no GPU): both children stop at the no-GPU assertion; the parent must come back
    promptly with a non-zero code and must not have touched a GPU itself."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "bench.py needs a GPU" in r.stderr
    assert r.stdout.strip() == ""                                     # no JSON line from a failed run


def test_cloud_builder_reproduces_the_generators_neighbourhoods():
    """workload.make_cloud (ghost images + cell-list neighbour search of ARBITRARY positions in a periodic box: the LAMMPS
    side of bench.py --workload step) against workload.make_tgv on the advected lattice: same neighbour geometry per row
    (pairs exactly on the cut radius, where W = 0, round either way and are left out of the comparison)."""
    import numpy as np
    import isph_amd  # noqa: F401
    from isph_amd import workload
    for dim, n in ((3, 16), (2, 24)):
        spec = workload.TGVSpec(dim=dim, ncell=(n,) * dim, brick=(8,) * dim, origin=(0.5,) * dim if dim == 2 else (0.0,) * 3,
                                mode=workload.ADVECT)
        p = workload.make_tgv(spec)
        nl = p["nlocal"]
        c = workload.make_cloud(p["x"][:nl], (2 * np.pi,) * dim, spec.h, spec.cut, dim=dim, like=p)
        assert c["nlocal"] == nl and np.array_equal(c["owner_index"][:nl], np.arange(nl))
        assert np.all(c["tag"] == c["owner_index"] + 1)

        def rows(q):
            out = []
            for i in range(0, nl, 17):
                j = q["neigh_idx"][q["neigh_ptr"][i]:q["neigh_ptr"][i + 1]]
                d = q["x"][j] - q["x"][i]
                d = d[np.abs(np.linalg.norm(d, axis=1) - spec.cut) > 1e-9]
                out.append(sorted(map(tuple, np.round(d, 8))))
            return out
        assert rows(c) == rows(p)
        # ghosts are exact periodic images of their owners
        L = 2 * np.pi
        dxo = c["x"][nl:] - c["x"][c["owner_index"][nl:]]
        assert np.allclose(dxo / L, np.round(dxo / L), atol=1e-12) and np.all(np.abs(dxo).max(axis=1) > 1.0)
