"""Seeded TGV test problems shared by the CPU and GPU tests.  The oracle builds
the expected matrices / right-hand sides; the product path never sees this."""
import numpy as np

import isph_amd  # noqa: F401
from isph_amd import workload
import oracle as orc


def tgv_spec(dim=3, n=12, mode=workload.ADVECT, brick=8, kernel="wendland", cut_over_h=2.0, **kw):
    if dim == 3:
        return workload.TGVSpec(dim=3, ncell=(n, n, n), brick=(brick,) * 3, mode=mode, cut_over_h=cut_over_h,
                                kernel=kernel, **kw)
    return workload.TGVSpec(dim=2, ncell=(n, n), brick=(brick, brick), origin=(0.5, 0.5), mode=mode,
                            cut_over_h=cut_over_h, kernel=kernel, **kw)


class Problem:
    """particles + oracle precompute + oracle Poisson system for one rank."""

    def __init__(self, spec, antisym=True, singular=orc.NULLSPACE, kinds=None, types=None):
        self.spec = spec
        self.parts = workload.make_tgv(spec)
        if types is not None:
            self.parts["type"] = np.ascontiguousarray(types(self.parts), dtype=np.int32)
        self.colmap = workload.single_rank_colmap(self.parts)
        self.antisym, self.singular, self.kinds = antisym, singular, kinds
        self.P = orc.Particles(self.parts, self.colmap, kernel=spec.kernel, kinds=kinds)
        self.P.precompute(corrections=not antisym)
        if not antisym:  # ghosts need the owner's G_i / L_i only through row i: no comm needed
            pass
        self.n = self.parts["nlocal"]

    def poisson(self):
        p = self.parts
        return self.P.poisson(self.spec.dt, p["rho"], p["v"], antisym=self.antisym, singular=self.singular)
