"""Domain-decomposition plumbing for the multi-GPU path (one process per GPU).

What Epetra builds inside FillComplete for the reference (column map +
Epetra_Import, ref: functor_graph.h:97, pair_isph.cpp:1258-1270) and what
LAMMPS' forward_comm_pair does for per-atom scalars: from each rank's ghost
atoms (owner rank, owner local index) derive
  * colmap   matrix column of every local+ghost particle (owned columns first,
             ghost columns grouped by owning rank),
  * the send/recv lists of the per-SpMV halo exchange.
Index lists are exchanged once at plan time with torch.distributed object
collectives (works on gloo and nccl); the per-SpMV exchange itself runs inside
libisph_hip on RCCL (csrc/solver.hpp halo_exchange).
"""
import sys
from dataclasses import dataclass, field

import numpy as np


@dataclass
class HaloPlan:
    rank: int
    nranks: int
    nlocal: int
    ncol: int
    colmap: np.ndarray
    peers: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    send_ptr: np.ndarray = field(default_factory=lambda: np.zeros(1, np.int32))
    send_idx: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    recv_ptr: np.ndarray = field(default_factory=lambda: np.zeros(1, np.int32))
    ghost_col_of: np.ndarray = None   # [nall-nlocal] column of each ghost particle
    recv_idx: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))   # owner-local row of every ghost column

    @property
    def npeers(self):
        return len(self.peers)


def make_self_halo_plan(parts):
    """Single-rank plan that routes the periodic images through the halo machinery
    (ghost columns + send/recv to self) instead of folding them onto the owner's
    column.  Used to exercise the RCCL halo path on one GPU."""
    me = int(parts["spec"].rank)
    nlocal, nall = int(parts["nlocal"]), int(parts["nall"])
    assert np.all(parts["owner_rank"] == me)
    oidx = parts["owner_index"].astype(np.int64)
    colmap = np.empty(nall, dtype=np.int32)
    colmap[:nlocal] = np.arange(nlocal, dtype=np.int32)
    uidx, inv = np.unique(oidx[nlocal:], return_inverse=True)
    colmap[nlocal:] = (nlocal + inv).astype(np.int32)
    plan = HaloPlan(me, 1, nlocal, nlocal + len(uidx), colmap, np.asarray([me], np.int32),
                    np.asarray([0, len(uidx)], np.int32), uidx.astype(np.int32), np.asarray([0, len(uidx)], np.int32))
    plan.ghost_col_of = colmap[nlocal:].copy()
    plan.recv_idx = uidx.astype(np.int32)
    return plan


def prune_ghosts(parts):
    """Drops the ghost particles no owned particle lists as a neighbour.  The generator (like LAMMPS) creates every
    ghost inside the box of the rank's brick grown by the cut-off; the corners of that box lie outside every
    neighbourhood sphere.  Epetra's column map only holds columns that were inserted (FunctorOuterGraph inserts tag[j]
    for r^2 < cutsq, functor_graph.h:62-80; FillComplete builds the map and the Import from them), so those ghosts are
    neither matrix columns nor halo traffic in the reference -- and not rows of Ifpack's overlapped subdomain
    (precond_ifpack.h:43).  Returns a new dict with nall reduced and neigh_idx renumbered; owned particles are kept."""
    nl, na = int(parts["nlocal"]), int(parts["nall"])
    keep = np.zeros(na, dtype=bool)
    keep[:nl] = True
    keep[parts["neigh_idx"]] = True
    if keep.all():
        return parts
    newidx = np.cumsum(keep, dtype=np.int64) - 1
    out = dict(parts)
    for k, v in parts.items():
        if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == na and k not in ("neigh_idx", "neigh_ptr"):
            out[k] = np.ascontiguousarray(v[keep])
    out["neigh_idx"] = newidx[parts["neigh_idx"]].astype(parts["neigh_idx"].dtype)
    out["nall"] = int(keep.sum())
    return out


def make_plan(parts, td=None):
    """parts: output of workload.make_tgv for this rank.  td: torch.distributed
    (initialised) or None for a single rank."""
    me = int(parts["spec"].rank)
    nlocal, nall = int(parts["nlocal"]), int(parts["nall"])
    orank = parts["owner_rank"].astype(np.int64)
    oidx = parts["owner_index"].astype(np.int64)
    colmap = np.empty(nall, dtype=np.int32)
    colmap[:nlocal] = np.arange(nlocal, dtype=np.int32)
    g = np.arange(nlocal, nall)
    local_img = orank[g] == me
    colmap[g[local_img]] = oidx[g[local_img]]            # periodic image of an owned particle
    remote = g[~local_img]
    nranks = 1 if td is None else td.get_world_size()
    if len(remote) == 0 or td is None:
        assert len(remote) == 0, "remote ghosts need torch.distributed"
        plan = HaloPlan(me, nranks, nlocal, nlocal, colmap)
        plan.ghost_col_of = colmap[nlocal:].copy()
        if td is not None:                                # still take part in the collective
            out = [None] * nranks
            td.all_gather_object(out, {})
        return plan
    key = orank[remote] * (1 << 32) + oidx[remote]
    ukey, inv = np.unique(key, return_inverse=True)       # sorted by (rank, index): ghost column order
    colmap[remote] = (nlocal + inv).astype(np.int32)
    urank = (ukey >> 32).astype(np.int32)
    uidx = (ukey & 0xFFFFFFFF).astype(np.int32)
    want = {int(r): uidx[urank == r] for r in np.unique(urank)}
    allwant = [None] * nranks
    td.all_gather_object(allwant, want)
    sends = {p: np.asarray(allwant[p][me], dtype=np.int32) for p in range(nranks) if p != me and me in allwant[p]}
    peers = sorted(set(want) | set(sends))
    send_ptr, recv_ptr, send_idx = [0], [0], []
    for p in peers:
        s = sends.get(p, np.zeros(0, np.int32))
        send_idx.append(s)
        send_ptr.append(send_ptr[-1] + len(s))
        recv_ptr.append(recv_ptr[-1] + len(want.get(p, ())))
    plan = HaloPlan(me, nranks, nlocal, nlocal + len(ukey), colmap,
                    np.asarray(peers, np.int32), np.asarray(send_ptr, np.int32),
                    np.concatenate(send_idx).astype(np.int32) if send_idx else np.zeros(0, np.int32),
                    np.asarray(recv_ptr, np.int32))
    plan.ghost_col_of = colmap[nlocal:].copy()
    plan.recv_idx = np.concatenate([np.asarray(want.get(p, np.zeros(0, np.int32)), dtype=np.int32) for p in peers]) \
        if peers else np.zeros(0, np.int32)
    return plan


def exchange(plan, owned, td, device=None):
    """Halo exchange of one owned vector with torch.distributed p2p (plan-time
    utility and CPU/gloo test path).  Returns the ncol-long extended vector."""
    import torch
    owned_t = owned if isinstance(owned, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(owned))
    if device is not None:
        owned_t = owned_t.to(device)
    ext = torch.empty(plan.ncol, dtype=owned_t.dtype, device=owned_t.device)
    ext[:plan.nlocal] = owned_t[:plan.nlocal]
    if plan.npeers == 0:
        return ext
    sidx = torch.from_numpy(plan.send_idx.astype(np.int64)).to(owned_t.device)
    sbuf = owned_t[sidx].contiguous()
    ops, keep = [], []
    for k, p in enumerate(plan.peers):
        s0, s1 = int(plan.send_ptr[k]), int(plan.send_ptr[k + 1])
        r0, r1 = int(plan.recv_ptr[k]), int(plan.recv_ptr[k + 1])
        if s1 > s0:
            ops.append(td.P2POp(td.isend, sbuf[s0:s1], int(p)))
        if r1 > r0:
            rb = torch.empty(r1 - r0, dtype=owned_t.dtype, device=owned_t.device)
            keep.append((r0, r1, rb))
            ops.append(td.P2POp(td.irecv, rb, int(p)))
    for w in td.batch_isend_irecv(ops):
        w.wait()
    for r0, r1, rb in keep:
        ext[plan.nlocal + r0:plan.nlocal + r1] = rb
    return ext


def forward_scalar(plan, owned, td, device=None):
    """forward_comm_pair of a per-atom scalar (e.g. Vfrac, functor_volume.h:76-80):
    returns the [nall] array with every ghost holding its owner's value."""
    import torch
    ext = exchange(plan, owned, td, device)
    cm = torch.from_numpy(plan.colmap.astype(np.int64)).to(ext.device)
    return ext[cm].contiguous()


def forward_scalar_rccl(fwd, plan, owned):
    """forward_scalar through the library's own communicator (hip.HaloForward made from this plan): the [nall] array
    with every ghost holding its owner's value.  owned: torch tensor on the device or numpy array, [>= nlocal]."""
    import torch
    ghosts = fwd.forward(owned[:plan.nlocal].contiguous() if isinstance(owned, torch.Tensor) else np.ascontiguousarray(owned[:plan.nlocal]))
    if isinstance(owned, torch.Tensor):
        ext = torch.cat([owned[:plan.nlocal], ghosts])
        return ext[torch.from_numpy(plan.colmap.astype(np.int64)).to(ext.device)].contiguous()
    return np.concatenate([owned[:plan.nlocal], ghosts])[plan.colmap]


def extend_rows(plan, rowptr, colidx, val, td=None):
    """The matrix of this rank's subdomain extended by one layer ("Overlap Level" 1 across ranks,
    precond_ifpack.h:43,63: Ifpack_OverlappingRowMatrix): rows [0, nlocal) are the local rows as they are, row
    nlocal + g is the row of ghost column g, received from its owner and restricted to this rank's extended column set
    (owned + ghost columns); entries that point further out are dropped.  Returns CSR (rowptr, colidx, val) of the
    square (nlocal + nghost) matrix, columns ascending.  td: torch.distributed, or None for a single-rank plan whose
    only peer is the rank itself (make_self_halo_plan: the ghost rows are then copies of owned rows; they couple to
    the image columns where an image exists and to the owned columns otherwise).  The numeric factorisation and the application live in the library (isph_prec_create_overlap)."""
    n, ncol = int(plan.nlocal), int(plan.ncol)
    rowptr, colidx, val = np.asarray(rowptr, dtype=np.int64), np.asarray(colidx, dtype=np.int64), np.asarray(val, dtype=np.float64)
    me = int(plan.rank)
    if td is None:
        assert all(int(p) == me for p in plan.peers), "remote peers need torch.distributed"
        off = {me: 0}
    else:
        sizes = [None] * td.get_world_size()
        td.all_gather_object(sizes, n)
        off = {r: int(sum(sizes[:r])) for r in range(len(sizes))}
    gcol = np.empty(ncol, dtype=np.int64)                      # global id of every local column
    gcol[:n] = off[me] + np.arange(n)
    for k, p in enumerate(plan.peers):
        r0, r1 = int(plan.recv_ptr[k]), int(plan.recv_ptr[k + 1])
        gcol[n + r0:n + r1] = off[int(p)] + plan.recv_idx[r0:r1].astype(np.int64)
    out = {}
    for k, p in enumerate(plan.peers):                         # the rows each peer holds as ghost columns
        rows = plan.send_idx[int(plan.send_ptr[k]):int(plan.send_ptr[k + 1])].astype(np.int64)
        lens = rowptr[rows + 1] - rowptr[rows]
        take = np.concatenate([np.arange(rowptr[r], rowptr[r + 1]) for r in rows]) if len(rows) else np.zeros(0, np.int64)
        out[int(p)] = (lens, gcol[colidx[take]], val[take])
    if td is None:
        incoming = {me: out.get(me)}
    else:
        allout = [None] * td.get_world_size()
        td.all_gather_object(allout, out)
        incoming = {int(p): allout[int(p)].get(me) for p in plan.peers}
    # global id -> extended index: owned columns by arithmetic, ghost columns by a sorted look-up
    gg = gcol[n:]
    order = np.argsort(gg, kind="stable")
    gsorted = gg[order]

    def ext_index(g):
        # ghost columns first, then owned ones: between different ranks a global id is one or the other; with the
        # self-peer plan a ghost is an image of an owned row, and the image rows couple to the images where there are any
        e = np.full(len(g), -1, dtype=np.int64)
        if len(g) and len(gsorted):
            pos = np.searchsorted(gsorted, g)
            pos[pos >= len(gsorted)] = len(gsorted) - 1
            hit = gsorted[pos] == g
            e[hit] = n + order[pos[hit]]
        own = (e < 0) & (g >= off[me]) & (g < off[me] + n)
        e[own] = g[own] - off[me]
        return e

    grp, gci, gv = [int(rowptr[n])], [], []
    for k, p in enumerate(plan.peers):
        lens, gids, vals = incoming[int(p)]
        assert len(lens) == int(plan.recv_ptr[k + 1]) - int(plan.recv_ptr[k]), "peer sent a different number of rows"
        e = ext_index(np.asarray(gids, dtype=np.int64))
        start = 0
        for ln in lens:
            ee, vv = e[start:start + ln], vals[start:start + ln]
            keep = ee >= 0
            ee, vv = ee[keep], vv[keep]
            o = np.argsort(ee, kind="stable")
            gci.append(ee[o])
            gv.append(vv[o])
            grp.append(grp[-1] + int(keep.sum()))
            start += int(ln)
    rp_ext = np.concatenate([rowptr[:n + 1], np.asarray(grp[1:], dtype=np.int64)])
    ci_ext = np.concatenate([colidx[:rowptr[n]]] + gci) if gci else colidx[:rowptr[n]]
    v_ext = np.concatenate([val[:rowptr[n]]] + gv) if gv else val[:rowptr[n]]
    assert rp_ext[-1] < 2 ** 31
    return rp_ext.astype(np.int32), ci_ext.astype(np.int32), v_ext


def extend_rows_levels(plan, rowptr, colidx, val, td, levels=1):
    """"Overlap Level" L across ranks (precond_ifpack.h:43; Ifpack_OverlappingRowMatrix with OverlapLevel = L): the rank's
    rows, then L layers of imported rows -- layer 1 = the rows of the matrix' ghost columns, layer l + 1 = the rows of
    the columns the layer-l rows reference outside everything gathered so far -- each layer in ascending global row
    number (rank-concatenated numbering: rows of rank r are off[r] .. off[r] + n_r), entries that leave the extended set
    dropped (Ifpack_LocalFilter).  Rows may come from ranks that are not neighbours of the matrix' own halo.

    Returns (rowptr, colidx, val, xplan): the CSR of the square extended matrix and the halo lists of its imported rows in
    the form isph_prec_create_overlap takes -- one (peer, send range, receive range) triple per layer and owner, so a rank
    can appear once per layer; receive ranges follow the extended row order.  xplan.nlocal / xplan.ncol = owned /
    extended rows.  td: torch.distributed-like (get_world_size, all_gather_object)."""
    n = int(plan.nlocal)
    me, world = int(plan.rank), td.get_world_size()
    rowptr, colidx, val = np.asarray(rowptr, dtype=np.int64), np.asarray(colidx, dtype=np.int64), np.asarray(val, dtype=np.float64)
    sizes = [None] * world
    td.all_gather_object(sizes, n)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    # global id of every local column of my own rows
    gcol = np.empty(int(plan.ncol), dtype=np.int64)
    gcol[:n] = off[me] + np.arange(n)
    for k, p in enumerate(plan.peers):
        r0, r1 = int(plan.recv_ptr[k]), int(plan.recv_ptr[k + 1])
        gcol[n + r0:n + r1] = off[int(p)] + plan.recv_idx[r0:r1].astype(np.int64)
    have = set(range(int(off[me]), int(off[me]) + n))            # global ids of the extended rows gathered so far
    ext_gids = []                                                   # imported rows, in extended order
    ext_rows = []                                                   # (global column ids, values) of every imported row
    peers, send_ptr, send_idx, recv_ptr = [], [0], [], [0]
    frontier = np.unique(gcol[colidx[:rowptr[n]]])                  # columns referenced by the rows of the previous layer
    for _ in range(int(levels)):
        want = np.asarray(sorted(g for g in frontier.tolist() if g not in have), dtype=np.int64)
        owner = np.searchsorted(off, want, side="right") - 1
        mine = {int(r): (want[owner == r] - off[r]).astype(np.int64) for r in np.unique(owner)}
        allwant = [None] * world
        td.all_gather_object(allwant, mine)
        # rows the others want from me, with global column ids
        out = {}
        for p in range(world):
            idx = allwant[p].get(me) if p != me else None
            if idx is None or len(idx) == 0:
                continue
            lens = rowptr[idx + 1] - rowptr[idx]
            take = np.concatenate([np.arange(rowptr[r], rowptr[r + 1]) for r in idx])
            out[p] = (lens, gcol[colidx[take]], val[take])
        allout = [None] * world
        td.all_gather_object(allout, out)
        nxt = []
        for r in sorted(set(mine) | set(out)):                      # one triple per owner / requester of this layer
            idx_r = mine.get(r, np.zeros(0, np.int64))
            snd = np.asarray(allwant[r].get(me, np.zeros(0, np.int64)), dtype=np.int64) if r != me else np.zeros(0, np.int64)
            peers.append(int(r))
            send_idx.append(snd.astype(np.int32))
            send_ptr.append(send_ptr[-1] + len(snd))
            recv_ptr.append(recv_ptr[-1] + len(idx_r))
            if len(idx_r):
                lens, gids, vals = allout[r][me]
                assert len(lens) == len(idx_r)
                start = 0
                for k, ln in enumerate(lens):
                    g = int(off[r] + idx_r[k])
                    ext_gids.append(g)
                    have.add(g)
                    ext_rows.append((np.asarray(gids[start:start + ln], dtype=np.int64), np.asarray(vals[start:start + ln])))
                    nxt.append(ext_rows[-1][0])
                    start += int(ln)
        frontier = np.unique(np.concatenate(nxt)) if nxt else np.zeros(0, np.int64)
    # local index of every global id of the extended set
    ext_gids = np.asarray(ext_gids, dtype=np.int64)
    order = np.argsort(ext_gids, kind="stable")
    gs = ext_gids[order]

    def ext_index(g):
        e = np.full(len(g), -1, dtype=np.int64)
        own = (g >= off[me]) & (g < off[me] + n)
        e[own] = g[own] - off[me]
        if len(gs):
            pos = np.clip(np.searchsorted(gs, g), 0, len(gs) - 1)
            hit = (gs[pos] == g) & ~own
            e[hit] = n + order[pos[hit]]
        return e

    rp_out, ci_out, v_out = [0], [], []
    eo = ext_index(gcol[colidx[:rowptr[n]]])
    for i in range(n):
        a, b = int(rowptr[i]), int(rowptr[i + 1])
        e, vv = eo[a:b], val[a:b]
        keep = e >= 0
        o = np.argsort(e[keep], kind="stable")
        ci_out.append(e[keep][o]); v_out.append(vv[keep][o]); rp_out.append(rp_out[-1] + int(keep.sum()))
    for gids, vals in ext_rows:
        e = ext_index(gids)
        keep = e >= 0
        o = np.argsort(e[keep], kind="stable")
        ci_out.append(e[keep][o]); v_out.append(vals[keep][o]); rp_out.append(rp_out[-1] + int(keep.sum()))
    xplan = HaloPlan(me, world, n, n + len(ext_gids), plan.colmap, np.asarray(peers, np.int32), np.asarray(send_ptr, np.int32),
                     np.concatenate(send_idx).astype(np.int32) if send_idx else np.zeros(0, np.int32), np.asarray(recv_ptr, np.int32))
    xplan.ext_gids = ext_gids
    return (np.asarray(rp_out, dtype=np.int32), np.concatenate(ci_out).astype(np.int32) if ci_out else np.zeros(0, np.int32),
            np.concatenate(v_out) if v_out else np.zeros(0), xplan)


def td_host_transport(td):
    """hip.HostTransport (isph_host_transport) over a torch.distributed process group with CPU tensors (gloo): the
    host-staged transport for process ranks that SHARE one device -- RCCL refuses to form a communicator there.  Used by
    `bench.py --share-gpu` to rehearse the N-rank run (decomposition, plan, halo exchange, all-reduces, timing) on a
    one-GPU box; the callbacks must outlive the context: keep the returned object."""
    import ctypes as C
    import torch
    from . import hip
    EX = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_longlong),
                     C.POINTER(C.c_double), C.POINTER(C.c_longlong))
    AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int)
    me = td.get_rank()

    def exchange(user, npeers, peer, send, so, recv, ro):
        try:
            ns_all, nr_all = int(so[npeers]), int(ro[npeers])
            sbuf = torch.from_numpy(np.ctypeslib.as_array(send, shape=(max(ns_all, 1),)))
            rbuf = torch.from_numpy(np.ctypeslib.as_array(recv, shape=(max(nr_all, 1),)))
            reqs = []
            for p in range(npeers):
                s0, s1, r0, r1 = int(so[p]), int(so[p + 1]), int(ro[p]), int(ro[p + 1])
                if int(peer[p]) == me:                    # periodic wrap onto the same rank
                    assert s1 - s0 == r1 - r0
                    rbuf[r0:r1] = sbuf[s0:s1]
                    continue
                if r1 > r0:
                    reqs.append(td.irecv(rbuf[r0:r1], src=int(peer[p])))
                if s1 > s0:
                    reqs.append(td.isend(sbuf[s0:s1].clone(), dst=int(peer[p])))
            for q in reqs:
                q.wait()
            return 0
        except Exception as e:                            # never let an exception cross the C boundary
            sys.stderr.write("td_host_transport.exchange: %r\n" % (e,))
            return 1

    def allreduce(user, buf, count, op):
        try:
            t = torch.from_numpy(np.ctypeslib.as_array(buf, shape=(max(int(count), 1),)))[:int(count)]
            td.all_reduce(t, op=td.ReduceOp.MAX if op == 1 else td.ReduceOp.SUM)
            return 0
        except Exception as e:
            sys.stderr.write("td_host_transport.allreduce: %r\n" % (e,))
            return 1

    tr = hip.HostTransport()
    tr._keep = (EX(exchange), AR(allreduce))              # the CFUNCTYPE objects own the thunks
    tr.exchange = C.cast(tr._keep[0], C.c_void_p)
    tr.allreduce = C.cast(tr._keep[1], C.c_void_p)
    return tr
