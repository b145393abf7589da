"""N ranks on ONE GPU for the -m gpu tests: every rank is a thread of this process with its own isph_ctx made by
isph_ctx_create_hostcomm over the test transport tests/cpp/rank_threads.cpp (MPI-like mailboxes between the threads).
RCCL cannot place two ranks of a communicator on one device and the GPU box admits six processes on its card; this is
how the 2-, 4- and 8-rank decompositions run through the library's real multi-rank code -- pack kernels, the exchange on
the halo stream, ghost-column SpMV, all-reduced dots, rank-0-only branches -- with DIFFERENT peers per rank.
ctypes releases the GIL for the duration of a library call, so the ranks really run concurrently."""
import ctypes as C
import threading
import traceback

import numpy as np

from isph_amd import build, hip


class RankGroup:
    def __init__(self, nranks, timeout_s=240.0):
        self.n = int(nranks)
        self.lib = C.CDLL(build.build_rank_threads())
        self.lib.rt_group_create.restype = C.c_void_p
        self.lib.rt_group_create.argtypes = [C.c_int, C.c_double]
        self.lib.rt_rank_create.restype = C.c_void_p
        self.lib.rt_rank_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        for f in (self.lib.rt_group_destroy, self.lib.rt_group_abort, self.lib.rt_rank_destroy):
            f.argtypes = [C.c_void_p]
            f.restype = None
        self.lib.rt_group_aborted.argtypes = [C.c_void_p]
        self.lib.rt_group_counts.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.rt_group_counts.restype = None
        self.g = C.c_void_p(self.lib.rt_group_create(self.n, float(timeout_s)))
        self._ranks = []
        self._barrier = threading.Barrier(self.n)
        self._slots = [None] * self.n

    def transport(self, rank):
        """hip.HostTransport of `rank`; valid until close()"""
        t = hip.HostTransport()
        h = C.c_void_p(self.lib.rt_rank_create(self.g, int(rank), C.byref(t)))
        self._ranks.append((h, t))
        return t

    def context(self, rank, device=0, ordering=None):
        return hip.Context(device, rank=rank, nranks=self.n, transport=self.transport(rank), ordering=ordering)

    def td(self, rank):
        """the two torch.distributed calls isph_amd.dist uses at plan time, between the rank threads"""
        return _ThreadTD(self, rank)

    def counts(self):
        a = (C.c_longlong * 2)()
        self.lib.rt_group_counts(self.g, a)
        return dict(exchanges=int(a[0]), allreduces=int(a[1]))

    def run(self, fn, *args):
        """fn(rank, group, *args) on every rank thread; returns the list of results.  A rank that raises aborts the
        group (the others' transport calls fail instead of waiting) and the first exception is re-raised here."""
        out, err = [None] * self.n, [None] * self.n

        def body(r):
            try:
                out[r] = fn(r, self, *args)
            except BaseException as e:  # noqa: BLE001
                err[r] = (e, traceback.format_exc())
                self.lib.rt_group_abort(self.g)
                self._barrier.abort()

        th = [threading.Thread(target=body, args=(r,), name="rank%d" % r) for r in range(self.n)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        first = [e for e in err if e is not None and not isinstance(e[0], threading.BrokenBarrierError)] or [e for e in err if e]
        if first:
            raise RuntimeError("rank thread failed:\n" + first[0][1]) from first[0][0]
        return out

    def close(self):
        for h, _ in self._ranks:
            self.lib.rt_rank_destroy(h)
        self._ranks = []
        if self.g:
            self.lib.rt_group_destroy(self.g)
            self.g = C.c_void_p()


class _ThreadTD:
    def __init__(self, group, rank):
        self.G, self.rank = group, rank

    def get_world_size(self):
        return self.G.n

    def get_rank(self):
        return self.rank

    def all_gather_object(self, out, obj):
        G = self.G
        G._slots[self.rank] = obj
        G._barrier.wait()
        for r in range(G.n):
            out[r] = G._slots[r]
        G._barrier.wait()


def empty_parts(spec_like, rank):
    """a rank that owns no particles (LAMMPS allows empty subdomains): the dict workload.make_tgv would return"""
    import copy
    sp = copy.copy(spec_like)
    sp.rank = rank
    z3 = np.zeros((0, 3))
    zi = np.zeros(0, dtype=np.int32)
    return dict(spec=sp, dim=sp.dim, nlocal=0, nall=0, x=z3, v=z3.copy(), tag=zi, type=zi.copy(), owner_rank=zi.copy(),
                owner_index=zi.copy(), neigh_ptr=np.zeros(1, dtype=np.int32), neigh_idx=zi.copy(), rho=np.zeros(0),
                nu=np.zeros(0), h=sp.h, cut=sp.cut, dt=sp.dt)
