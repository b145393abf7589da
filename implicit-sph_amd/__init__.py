"""implicit-sph_amd: MI355X-native pressure-Poisson / Helmholtz hot path of
sandialabs/implicit-sph (assembly + Krylov solve + preconditioner) behind the
reference's SolverLin / PrecondWrapper surface.  See DESIGN.md.

The directory name carries a hyphen (it mirrors the reference repo name), so it
is imported through ``isph_amd`` (repo root) or ``importlib.import_module``."""
from . import build  # noqa: F401
from . import workload  # noqa: F401
from . import dist  # noqa: F401
from . import hip  # noqa: F401
