"""Graph partitions and the once-per-sweep halo exchange.

The reference has no distributed path at all; this is new functionality built on the same C ABI.
A rank's local graph holds the variables it owns, every factor touching them, and — for factors cut by the
partition — the remote variable as a degree-1 *ghost* whose variable→factor message is imported each sweep.
Per sweep and per neighbouring rank exactly one message per cut factor travels in each direction
(16 bytes, natural form): grid strips exchange n_cols messages per boundary.  No collective is involved:
point-to-point send/recv between partition neighbours (RCCL over xGMI through torch.distributed on GPUs,
gloo in the CPU tests).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple

import os
import numpy as np

from . import _lib as L
from . import synth


@dataclass
class Peer:
    rank: int
    send: slice      # rows of the send buffer that go to this peer
    recv: slice      # rows of the recv buffer filled by this peer


@dataclass
class Partition:
    model: synth.Model
    rank: int
    world: int
    send_var: np.ndarray     # (variable_id, factor_id) of exported variable→factor messages, grouped by peer
    send_fac: np.ndarray
    recv_var: np.ndarray     # (ghost variable_id, factor_id) of imported messages, grouped by peer
    recv_fac: np.ndarray
    peers: List[Peer] = field(default_factory=list)
    depth: int = 0                 # 0: per-sweep message halo; w > 0: state ("deep") halo exchanged every w sweeps
    owned_x: np.ndarray = None     # deep halo: the latent variables this rank owns (model.x_ids also lists the redundant ones)
    layer_var: np.ndarray = None   # deep halo: ids of the redundant variables and stand-ins ...
    layer: np.ndarray = None       # ... and their distance from the owned set (1 .. depth, stand-ins depth + 1): cx_halo_set_layers


def _row_bounds(n_rows: int, world: int) -> np.ndarray:
    """near-equal contiguous row blocks of ONE n_rows-row grid: rank r owns rows [b[r], b[r+1])"""
    return (np.arange(world + 1, dtype=np.int64) * n_rows) // world


def _grid_cut(total: int, n_cols: int, bounds, rank: int, seed: int) -> Partition:
    world = len(bounds) - 1
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    model = synth.gaussian_grid(total, n_cols, seed=seed, row0=r0, row1=r1)
    V, H = total * n_cols, total * (n_cols - 1)
    jj = np.arange(n_cols, dtype=np.int64)
    vid = lambda i: 1 + i * n_cols + jj                  # noqa: E731
    vfac = lambda i: 2 * V + H + 1 + i * n_cols + jj     # noqa: E731  factor between rows i and i+1
    sv, sf, rv, rf, peers = [], [], [], [], []
    pos_s = pos_r = 0
    if rank > 0:       # boundary with the strip above: cut factors between rows r0-1 | r0
        sv.append(vid(r0)); sf.append(vfac(r0 - 1)); rv.append(vid(r0 - 1)); rf.append(vfac(r0 - 1))
        peers.append(Peer(rank - 1, slice(pos_s, pos_s + n_cols), slice(pos_r, pos_r + n_cols)))
        pos_s += n_cols; pos_r += n_cols
    if rank < world - 1:  # boundary with the strip below: cut factors between rows r1-1 | r1
        sv.append(vid(r1 - 1)); sf.append(vfac(r1 - 1)); rv.append(vid(r1)); rf.append(vfac(r1 - 1))
        peers.append(Peer(rank + 1, slice(pos_s, pos_s + n_cols), slice(pos_r, pos_r + n_cols)))
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.int64)  # noqa: E731
    return Partition(model=model, rank=rank, world=world, send_var=cat(sv), send_fac=cat(sf), recv_var=cat(rv),
                     recv_fac=cat(rf), peers=peers)


def grid_strip(rows_per_rank: int, n_cols: int, rank: int, world: int, seed: int = 1234) -> Partition:
    """Row-strip partition of the (rows_per_rank*world) x n_cols Gaussian grid (weak scaling): rank r owns rows
    [r*rows_per_rank, (r+1)*rows_per_rank).  Cut = the vertical factors between neighbouring strips
    (n_cols factors per boundary).  METIS is not available in this image; for a regular grid strips minimise the
    number of neighbours (2) and give perfectly balanced parts."""
    return _grid_cut(rows_per_rank * world, n_cols, np.arange(world + 1, dtype=np.int64) * rows_per_rank, rank, seed)


def grid_rows(n_rows: int, n_cols: int, rank: int, world: int, seed: int = 1234) -> Partition:
    """ONE n_rows x n_cols grid cut into `world` near-equal row blocks (strong scaling; BASELINE config 4 is the
    1415 x 1415 grid cut 8 ways), one message halo per sweep."""
    return _grid_cut(n_rows, n_cols, _row_bounds(n_rows, world), rank, seed)


def _grid_cut_deep(total: int, n_cols: int, bounds, rank: int, depth: int, seed: int) -> Partition:
    world = len(bounds) - 1
    if depth < 1 or depth > int(np.min(np.diff(bounds))):
        raise ValueError("depth must be in [1, rows of the smallest block]: redundant rows come from the direct neighbours only")
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    lo, hi = max(r0 - depth, 0), min(r1 + depth, total)
    model = synth.gaussian_grid(total, n_cols, seed=seed, row0=lo, row1=hi)
    ev, ef = np.asarray(model.edge_var, np.int64), np.asarray(model.edge_fac, np.int64)
    row = (ev - 1) // n_cols                              # global row of the edge's variable (ghost rows: lo-1, hi)

    def edges_of_rows(a, b):
        m = (row >= a) & (row < b)
        v, f = ev[m], ef[m]
        o = np.lexsort((f, v))
        return v[o], f[o]

    sv, sf, rv, rf, peers = [], [], [], [], []
    ps = pr = 0
    for peer, send_rows, recv_rows in ((rank - 1, (r0, min(r0 + depth, r1)), (lo, r0)), (rank + 1, (max(r1 - depth, r0), r1), (r1, hi))):
        if peer < 0 or peer >= world:
            continue
        a, b = edges_of_rows(*send_rows)
        c, d = edges_of_rows(*recv_rows)
        sv.append(a); sf.append(b); rv.append(c); rf.append(d)
        peers.append(Peer(peer, slice(ps, ps + len(a)), slice(pr, pr + len(c))))
        ps += len(a); pr += len(c)
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.int64)  # noqa: E731
    jj = np.arange(n_cols, dtype=np.int64)
    owned = (1 + np.arange(r0, r1, dtype=np.int64)[:, None] * n_cols + jj[None, :]).ravel()
    vids = np.unique(ev)
    vrow = (vids - 1) // n_cols
    lay = np.where(vrow < r0, r0 - vrow, np.where(vrow >= r1, vrow - r1 + 1, 0))
    red = lay > 0
    return Partition(model=model, rank=rank, world=world, send_var=cat(sv), send_fac=cat(sf), recv_var=cat(rv), recv_fac=cat(rf),
                     peers=peers, depth=depth, owned_x=owned, layer_var=vids[red], layer=lay[red].astype(np.int32))


def grid_strip_deep(rows_per_rank: int, n_cols: int, rank: int, world: int, depth: int, seed: int = 1234) -> Partition:
    """Row strips with a deep halo (weak scaling): besides its own rows [r0, r1) the rank holds `depth` redundant rows of
    each neighbour (all their factors included; beyond them the usual degree-1 stand-ins of the next row).  Halo lists =
    every factor→variable message of the redundant rows, ordered by (variable id, factor id) — both sides of a cut build
    the same list.  One exchange per `depth` plain sweeps keeps every message of an owned variable bit-identical to the
    un-partitioned sweep (the error of the frozen outer edge advances one row per sweep)."""
    return _grid_cut_deep(rows_per_rank * world, n_cols, np.arange(world + 1, dtype=np.int64) * rows_per_rank, rank, depth, seed)


def grid_rows_deep(n_rows: int, n_cols: int, rank: int, world: int, depth: int, seed: int = 1234) -> Partition:
    """ONE n_rows x n_cols grid cut into `world` near-equal row blocks, each with `depth` redundant rows per side
    (strong scaling: BASELINE config 4 = the 1415 x 1415 grid, 8 blocks of 176 / 177 rows)."""
    return _grid_cut_deep(n_rows, n_cols, _row_bounds(n_rows, world), rank, depth, seed)


def by_assignment(model: synth.Model, owner_of_variable, rank: int, world: int, one_stand_in_per_cut_factor: bool = False) -> Partition:
    """Generic vertex partition: `owner_of_variable(ids) -> ranks` assigns every variable to a rank.
    one_stand_in_per_cut_factor: a far variable that touches several cut factors of this rank appears once PER FACTOR, under fresh ids
    above every id of the model — each copy a degree-1 stand-in, as the library's halo lists want them (on a graph where the far
    variable would otherwise sit between two of this rank's pieces as a free variable of degree 2: TreeRegionExchange).
    Rank r keeps its variables, every factor touching one of them and, for pairwise factors whose other variable lives
    elsewhere, that variable as a degree-1 ghost.  Exports / imports are ordered by (peer, factor id), which both sides
    of a cut compute identically, so the k-th exported message of rank a towards rank b is the k-th imported message of
    b from a.  Scalar models with unary and pairwise factors."""
    ev, ef = np.asarray(model.edge_var, np.int64), np.asarray(model.edge_fac, np.int64)
    own_e = np.asarray(owner_of_variable(ev), np.int64)
    # pairwise factors: the two edges of each factor
    order = np.lexsort((ev, ef))
    fs, vs, os_ = ef[order], ev[order], own_e[order]
    first = np.flatnonzero(np.r_[True, fs[1:] != fs[:-1]])
    counts = np.diff(np.r_[first, len(fs)])
    if np.any(counts > 2):
        raise ValueError("by_assignment handles unary and pairwise factors")
    two = first[counts == 2]
    a_v, b_v, a_o, b_o, f2 = vs[two], vs[two + 1], os_[two], os_[two + 1], fs[two]
    one = first[counts == 1]
    keep_unary = one[os_[one] == rank]
    mine_a, mine_b = a_o == rank, b_o == rank
    inner = mine_a & mine_b
    cut_a = mine_a & ~mine_b          # I own a, b is remote
    cut_b = mine_b & ~mine_a
    # local edges: unary + inner pairs (both edges) + cut pairs (own edge + ghost edge)
    loc_var = np.concatenate([vs[keep_unary], a_v[inner], b_v[inner], a_v[cut_a], b_v[cut_a], a_v[cut_b], b_v[cut_b]])
    loc_fac = np.concatenate([fs[keep_unary], f2[inner], f2[inner], f2[cut_a], f2[cut_a], f2[cut_b], f2[cut_b]])
    loc_role = None
    if model.edge_role is not None:      # directed factors (x_out = A x_in + noise): the role travels with the edge
        ro = np.asarray(model.edge_role)[order]
        ra, rb = ro[two], ro[two + 1]
        loc_role = np.concatenate([ro[keep_unary], ra[inner], rb[inner], ra[cut_a], rb[cut_a], ra[cut_b], rb[cut_b]]).astype(np.int32)
    fid = np.asarray(model.factor_ids, np.int64)
    keep_f = np.isin(fid, np.unique(loc_fac))
    x_own = np.asarray(model.x_ids, np.int64)
    x_own = x_own[np.asarray(owner_of_variable(x_own)) == rank]

    def sel(var_arr, *cols):
        var_arr = np.asarray(var_arr, np.int64)
        m = np.asarray(owner_of_variable(var_arr)) == rank if len(var_arr) else np.zeros(0, bool)
        return [np.asarray(c)[m] for c in (var_arr,) + cols]

    dv, df, dy = sel(model.data_var, model.data_fac, model.data_y) if len(model.data_var) else (np.zeros(0, np.int64),) * 2 + (np.zeros(0),)
    if len(model.prior_var):
        pv, pf, pm, pvv = sel(model.prior_var, model.prior_fac, model.prior_mean, model.prior_variance)
    else:
        pv = pf = np.zeros(0, np.int64); pm = pvv = np.zeros(0)
    local = synth.Model(edge_var=loc_var, edge_fac=loc_fac, factor_ids=fid[keep_f], factor_kind=np.asarray(model.factor_kind)[keep_f],
                        factor_var=np.asarray(model.factor_var)[keep_f], x_ids=x_own, data_var=dv, data_fac=df, data_y=dy,
                        prior_var=pv, prior_fac=pf, prior_mean=pm, prior_variance=pvv, meta=dict(model.meta),
                        dim=model.dim, edge_role=loc_role, psets=dict(model.psets))
    if one_stand_in_per_cut_factor:
        base = int(max(ev.max(), ef.max(), fid.max())) + 1
        n_a, n_b = int(cut_a.sum()), int(cut_b.sum())
        fresh_a, fresh_b = base + np.arange(n_a, dtype=np.int64), base + n_a + np.arange(n_b, dtype=np.int64)
        # loc_var is [unary, inner a, inner b, cut_a own, cut_a ghost, cut_b ghost, cut_b own]: rename the two ghost runs
        n0 = len(keep_unary) + 2 * int(inner.sum())
        loc_var = loc_var.copy()
        loc_var[n0 + n_a:n0 + 2 * n_a] = fresh_a
        loc_var[n0 + 2 * n_a:n0 + 2 * n_a + n_b] = fresh_b
        local.edge_var = loc_var
        gh_of_cut_a, gh_of_cut_b = fresh_a, fresh_b
    else:
        gh_of_cut_a, gh_of_cut_b = b_v[cut_a], a_v[cut_b]
    # halo lists grouped by peer, ordered by factor id within a peer
    my_v = np.concatenate([a_v[cut_a], b_v[cut_b]]); gh_v = np.concatenate([gh_of_cut_a, gh_of_cut_b])
    cf = np.concatenate([f2[cut_a], f2[cut_b]]); peer = np.concatenate([b_o[cut_a], a_o[cut_b]])
    o = np.lexsort((cf, peer))
    my_v, gh_v, cf, peer = my_v[o], gh_v[o], cf[o], peer[o]
    peers, pos = [], 0
    for pr in np.unique(peer):
        n = int((peer == pr).sum())
        peers.append(Peer(int(pr), slice(pos, pos + n), slice(pos, pos + n)))
        pos += n
    return Partition(model=local, rank=rank, world=world, send_var=my_v, send_fac=cf, recv_var=gh_v, recv_fac=cf, peers=peers)


def by_assignment_deep(model: synth.Model, owner_of_variable, rank: int, world: int, depth: int) -> Partition:
    """Generic vertex partition with a deep halo (the scheme of grid_strip_deep for any unary + pairwise model): rank r
    keeps its variables, `depth` breadth-first layers of redundant variables around them with ALL their factors, and the
    next layer as degree-1 stand-ins.  Halo lists: every factor→variable message of a redundant variable, imported from its
    owner; ordered by (variable id, factor id), which both sides compute identically from the global model."""
    import scipy.sparse as sp

    if depth < 1:
        raise ValueError("depth >= 1")
    ev, ef = np.asarray(model.edge_var, np.int64), np.asarray(model.edge_fac, np.int64)
    vids = np.unique(ev)
    vix = np.searchsorted(vids, ev)
    nv = len(vids)
    own = np.asarray(owner_of_variable(vids), np.int64)
    # variable adjacency through the pairwise factors
    order = np.lexsort((ev, ef))
    fs, vx = ef[order], vix[order]
    first = np.flatnonzero(np.r_[True, fs[1:] != fs[:-1]])
    counts = np.diff(np.r_[first, len(fs)])
    # variables are adjacent when they share a factor: the two ends of a pairwise factor, every pair of a factor of more variables
    # (round 5: CX_FACTOR_GAUSS_LINEAR_N — a cut factor then keeps ALL its variables on every rank that holds one of them within the
    # halo, the ones beyond it as stand-ins)
    a_list, b_list = [], []
    for k in np.unique(counts[counts >= 2]):
        st = first[counts == k]
        for i in range(int(k)):
            for j in range(i + 1, int(k)):
                a_list.append(vx[st + i]); b_list.append(vx[st + j])
    a = np.concatenate(a_list) if a_list else np.zeros(0, np.int64)
    b = np.concatenate(b_list) if b_list else np.zeros(0, np.int64)
    A = sp.coo_matrix((np.ones(2 * len(a), np.int8), (np.r_[a, b], np.r_[b, a])), shape=(nv, nv)).tocsr()

    def layers(r):
        """distance (0 = owned by r) of every variable within depth + 1 of rank r's variables, -1 beyond"""
        dist = np.full(nv, -1, np.int64)
        seen = own == r
        dist[seen] = 0
        frontier = seen.copy()
        for d in range(1, depth + 2):
            reach = (A @ frontier.astype(np.int8)) > 0
            frontier = reach & ~seen
            dist[frontier] = d
            seen |= frontier
        return dist

    dist_me = layers(rank)
    full = (dist_me >= 0) & (dist_me <= depth)           # owned + redundant: all their factors are kept
    local_v = dist_me >= 0                                # plus the stand-in layer
    # factors: unary ones of `full` variables, pairwise ones with at least one `full` end (the other end is local by construction)
    fac_of_edge_full = full[vix]
    keep_fac = np.unique(ef[fac_of_edge_full])
    e_keep = np.isin(ef, keep_fac) & local_v[vix]
    loc_var, loc_fac = ev[e_keep], ef[e_keep]
    loc_role = None if model.edge_role is None else np.asarray(model.edge_role)[e_keep].astype(np.int32)
    fid = np.asarray(model.factor_ids, np.int64)
    keep_f = np.isin(fid, keep_fac)
    x_all = np.asarray(model.x_ids, np.int64)
    x_loc = x_all[full[np.searchsorted(vids, x_all)]]
    x_own = x_all[dist_me[np.searchsorted(vids, x_all)] == 0]
    local_ids = vids[local_v]

    def sel(var_arr, *cols):
        var_arr = np.asarray(var_arr, np.int64)
        m = np.isin(var_arr, local_ids) if len(var_arr) else np.zeros(0, bool)
        return [np.asarray(c)[m] for c in (var_arr,) + cols]

    dv, df, dy = sel(model.data_var, model.data_fac, model.data_y) if len(model.data_var) else (np.zeros(0, np.int64),) * 2 + (np.zeros(0),)
    if len(dv):     # a datum enters through an edge that must exist locally
        m = np.isin(df, keep_fac)
        dv, df, dy = dv[m], df[m], dy[m]
    if len(model.prior_var):
        pv, pf, pm, pvv = sel(model.prior_var, model.prior_fac, model.prior_mean, model.prior_variance)
        m = np.isin(pf, keep_fac)
        pv, pf, pm, pvv = pv[m], pf[m], pm[m], pvv[m]
    else:
        pv = pf = np.zeros(0, np.int64); pm = pvv = np.zeros(0)
    local = synth.Model(edge_var=loc_var, edge_fac=loc_fac, factor_ids=fid[keep_f], factor_kind=np.asarray(model.factor_kind)[keep_f],
                        factor_var=np.asarray(model.factor_var)[keep_f], x_ids=x_loc, data_var=dv, data_fac=df, data_y=dy,
                        prior_var=pv, prior_fac=pf, prior_mean=pm, prior_variance=pvv, meta=dict(model.meta),
                        dim=model.dim, edge_role=loc_role, psets=dict(model.psets))

    if "coef_var" in local.meta:      # per-edge coefficients of factors of more than two variables (synth.kary_model): the local edges' only
        cvv, cff, caa = (np.asarray(local.meta[k]) for k in ("coef_var", "coef_fac", "coef"))
        have = set(zip(loc_var.tolist(), loc_fac.tolist()))
        m = np.array([(int(v), int(f)) in have for v, f in zip(cvv, cff)], dtype=bool) if len(cvv) else np.zeros(0, bool)
        local.meta["coef_var"], local.meta["coef_fac"], local.meta["coef"] = cvv[m], cff[m], caa[m]
        if "kary_ids" in local.meta:      # the per-factor rows of synth.kary_model's meta, for the factors this rank holds
            kk = np.isin(np.asarray(local.meta["kary_ids"]), keep_fac)
            for key in ("kary_ids", "q", "b", "out_var"):
                local.meta[key] = np.asarray(local.meta[key])[kk]
            local.meta["fac_vars"] = [fv for fv, keep in zip(local.meta["fac_vars"], kk) if keep]

    def edges_of(mask_v):
        m = mask_v[vix]
        v, f = ev[m], ef[m]
        o = np.lexsort((f, v))
        return v[o], f[o]

    sv, sf, rv, rf, peers = [], [], [], [], []
    ps = pr = 0
    for q in range(world):
        if q == rank:
            continue
        recv_mask = (own == q) & full                       # q's variables that are redundant here
        dist_q = layers(q)
        send_mask = (own == rank) & (dist_q >= 1) & (dist_q <= depth)   # my variables that are redundant on q
        if not recv_mask.any() and not send_mask.any():
            continue
        a_, b_ = edges_of(send_mask)
        c_, d_ = edges_of(recv_mask)
        sv.append(a_); sf.append(b_); rv.append(c_); rf.append(d_)
        peers.append(Peer(q, slice(ps, ps + len(a_)), slice(pr, pr + len(c_))))
        ps += len(a_); pr += len(c_)
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.int64)  # noqa: E731
    red = dist_me > 0
    return Partition(model=local, rank=rank, world=world, send_var=cat(sv), send_fac=cat(sf), recv_var=cat(rv), recv_fac=cat(rf),
                     peers=peers, depth=depth, owned_x=x_own, layer_var=vids[red], layer=dist_me[red].astype(np.int32))


def metis_assignment(model: synth.Model, world: int):
    """Variable → rank map from METIS_PartGraphKway, if a libmetis with 32-bit idx_t / 32-bit real_t can be loaded
    (BASELINE's config names a METIS 8-way cut; neither libmetis nor pymetis is in this image, so this path is untested
    here and every caller falls back to contiguous blocks when it returns None).  Graph = latent variables and the other
    variables, adjacent through the pairwise factors; returns (sorted variable ids, owner per id) or None."""
    import ctypes
    import ctypes.util

    name = ctypes.util.find_library("metis")
    if not name:
        return None
    try:
        lib = ctypes.CDLL(name)
        fn = lib.METIS_PartGraphKway
    except (OSError, AttributeError):
        return None
    import scipy.sparse as sp

    ev, ef = np.asarray(model.edge_var, np.int64), np.asarray(model.edge_fac, np.int64)
    vids = np.unique(ev)
    vix = np.searchsorted(vids, ev)
    order = np.lexsort((ev, ef))
    fs, vx = ef[order], vix[order]
    first = np.flatnonzero(np.r_[True, fs[1:] != fs[:-1]])
    two = first[np.diff(np.r_[first, len(fs)]) == 2]
    a, b = vx[two], vx[two + 1]
    nv = len(vids)
    A = sp.coo_matrix((np.ones(2 * len(a), np.int8), (np.r_[a, b], np.r_[b, a])), shape=(nv, nv)).tocsr()
    xadj, adjncy = A.indptr.astype(np.int32), A.indices.astype(np.int32)
    i32 = ctypes.c_int32
    n, ncon, nparts, objval = i32(nv), i32(1), i32(world), i32(0)
    part = np.zeros(nv, np.int32)
    p32 = ctypes.POINTER(ctypes.c_int32)
    rc = fn(ctypes.byref(n), ctypes.byref(ncon), xadj.ctypes.data_as(p32), adjncy.ctypes.data_as(p32), None, None, None,
            ctypes.byref(nparts), None, None, None, ctypes.byref(objval), part.ctypes.data_as(p32))
    if rc != 1 or part.min() < 0 or part.max() >= world:     # METIS_OK == 1
        return None
    return vids, part.astype(np.int64)


def auto_partition(model: synth.Model, rank: int, world: int, depth: int = 0) -> Partition:
    """METIS k-way cut when libmetis is present, contiguous blocks otherwise; deep halo when depth > 0."""
    got = metis_assignment(model, world) if world > 1 else None
    if got is None:
        return contiguous_blocks(model, rank, world, depth=depth)
    vids, part = got
    owner = lambda ids: part[np.searchsorted(vids, np.asarray(ids, np.int64))]  # noqa: E731
    return by_assignment_deep(model, owner, rank, world, depth) if depth else by_assignment(model, owner, rank, world)


def by_components(model: synth.Model, rank: int, world: int) -> synth.Model:
    """Independent objects ("shards naturally"): the connected components of the NON-OBSERVED part of a model — the trees
    of a forest, the chains of a batch of state-space models — dealt to the ranks, largest first onto the least loaded; no exchange of
    any kind follows: every rank runs its exact schedule (CX_SCHED_TREE, CX_SCHED_CHAIN_SCAN) on the sub-model this returns.  An
    observed variable belongs to no component: it travels with every factor that touches it (each rank sets its datum on its own copy).
    The sub-model keeps the model's ids."""
    ev, ef = np.asarray(model.edge_var, np.int64), np.asarray(model.edge_fac, np.int64)
    obs = np.isin(ev, np.asarray(model.data_var, np.int64))
    # union-find over factors: two factors that share a non-observed variable are one component
    fids, finv = np.unique(ef, return_inverse=True)
    parent = np.arange(len(fids))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    order = np.argsort(ev, kind="stable")
    evs, fis, obs_s = ev[order], finv[order], obs[order]
    for i in range(1, len(evs)):
        if evs[i] == evs[i - 1] and not obs_s[i]:
            a, b = find(fis[i]), find(fis[i - 1])
            if a != b:
                parent[a] = b
    root = np.array([find(i) for i in range(len(fids))])
    comp_ids, comp_of_fac = np.unique(root, return_inverse=True)
    load = np.bincount(comp_of_fac[finv], minlength=len(comp_ids))          # edges per component
    owner_of_comp = np.zeros(len(comp_ids), np.int64)
    totals = np.zeros(world, np.int64)
    for c in np.argsort(-load, kind="stable"):
        r = int(np.argmin(totals))
        owner_of_comp[c] = r
        totals[r] += load[c]
    fac_mine = owner_of_comp[comp_of_fac] == rank                            # per unique factor id
    keep_e = fac_mine[finv]
    my_f = set(fids[fac_mine].tolist())
    fmask = np.array([int(f) in my_f for f in np.asarray(model.factor_ids, np.int64)], dtype=bool)
    my_v = set(ev[keep_e].tolist())

    def pick(mask_src, *arrs):
        return tuple(np.asarray(a)[mask_src] for a in arrs)
    dm = np.array([int(v) in my_v and int(f) in my_f for v, f in zip(model.data_var, model.data_fac)], dtype=bool) if len(model.data_var) else np.zeros(0, bool)
    pm = np.array([int(f) in my_f for f in model.prior_fac], dtype=bool) if len(model.prior_var) else np.zeros(0, bool)
    meta = dict(model.meta)
    if "kary_ids" in meta:                                                   # synth.kary_model / tree_model: per-factor and per-edge tables follow
        km = np.array([int(f) in my_f for f in meta["kary_ids"]], dtype=bool)
        meta["kary_ids"] = np.asarray(meta["kary_ids"])[km]
        meta["fac_vars"] = [fv for fv, k in zip(meta["fac_vars"], km) if k]
        for key in ("q", "b", "out_var"):
            meta[key] = np.asarray(meta[key])[km]
        if "is_kary" in meta:
            meta["is_kary"] = np.asarray(meta["is_kary"])[km]
        for pre in ("coef", "all_coef"):
            if pre + "_fac" in meta:
                cm = np.array([int(f) in my_f for f in meta[pre + "_fac"]], dtype=bool)
                meta[pre + "_var"], meta[pre + "_fac"], meta[pre] = np.asarray(meta[pre + "_var"])[cm], np.asarray(meta[pre + "_fac"])[cm], np.asarray(meta[pre])[cm]
        meta["used"] = np.array(sorted(my_v), dtype=np.int64)
    return synth.Model(edge_var=ev[keep_e], edge_fac=ef[keep_e], factor_ids=np.asarray(model.factor_ids)[fmask], factor_kind=np.asarray(model.factor_kind)[fmask],
                       factor_var=np.asarray(model.factor_var)[fmask], x_ids=np.array([v for v in model.x_ids if int(v) in my_v], dtype=np.int64),
                       data_var=np.asarray(model.data_var)[dm], data_fac=np.asarray(model.data_fac)[dm], data_y=np.asarray(model.data_y)[dm],
                       prior_var=np.asarray(model.prior_var)[pm], prior_fac=np.asarray(model.prior_fac)[pm], prior_mean=np.asarray(model.prior_mean)[pm],
                       prior_variance=np.asarray(model.prior_variance)[pm], meta=meta, dim=model.dim,
                       edge_role=None if model.edge_role is None else np.asarray(model.edge_role)[keep_e], psets=model.psets)


def contiguous_blocks(model: synth.Model, rank: int, world: int, depth: int = 0) -> Partition:
    """Equal contiguous id-blocks of the latent variables (time blocks of a chain, row blocks of a grid); every other
    variable (observations) goes with its first neighbour among the latent variables."""
    x = np.sort(np.asarray(model.x_ids, np.int64))
    bounds = x[(np.arange(1, world) * len(x)) // world] if world > 1 else np.zeros(0, np.int64)
    ev, ef = np.asarray(model.edge_var, np.int64), np.asarray(model.edge_fac, np.int64)
    is_x = np.isin(ev, x)
    # an observation's owner = owner of the latent variable sharing its factor
    order = np.argsort(ef, kind="stable")
    fs, vs, xs = ef[order], ev[order], is_x[order]
    fac_latent = {}
    lat_f, lat_v = fs[xs], vs[xs]
    first = np.flatnonzero(np.r_[True, lat_f[1:] != lat_f[:-1]]) if len(lat_f) else np.zeros(0, np.int64)
    fac_latent = dict(zip(lat_f[first].tolist(), lat_v[first].tolist()))
    obs_owner_var = {int(v): fac_latent[int(f)] for v, f in zip(ev[~is_x], ef[~is_x]) if int(f) in fac_latent}

    def owner(ids):
        ids = np.asarray(ids, np.int64)
        rep = np.array([obs_owner_var.get(int(i), int(i)) for i in ids], dtype=np.int64) if len(ids) else ids
        return np.searchsorted(bounds, rep, side="right")

    return by_assignment_deep(model, owner, rank, world, depth) if depth else by_assignment(model, owner, rank, world)


def cylinder_self(n_rows: int, n_cols: int, seed: int = 1234):
    """A grid whose last row is also coupled to its first row (a cylinder), held by ONE rank that is its own halo
    neighbour: every wrap-around factor is cut, each side sees the other through a ghost variable, and the exported
    messages travel rank 0 -> rank 0.  Exercises the whole exchange machinery (pack, RCCL send/recv to self, events,
    unpack, ghost push) on a single GPU.  Returns (Partition, wrap) where wrap = (top variable ids, bottom variable ids,
    variances) describes the wrap factors for building the un-partitioned cylinder."""
    base = synth.gaussian_grid(n_rows, n_cols, seed=seed)
    rng = np.random.default_rng([seed, 7])
    qw = rng.uniform(0.5, 2.0, n_cols)
    top = 1 + np.arange(n_cols, dtype=np.int64)                          # row 0
    bot = 1 + (n_rows - 1) * n_cols + np.arange(n_cols, dtype=np.int64)  # row n_rows-1
    nid = int(max(base.factor_ids.max(), base.edge_var.max()))
    f_bot = nid + 1 + np.arange(n_cols, dtype=np.int64)                  # wrap factor as seen from the bottom row
    f_top = f_bot + n_cols                                               # ... and from the top row
    g_for_top = f_top + n_cols                                           # ghost of top[j], attached to f_bot[j]
    g_for_bot = g_for_top + n_cols                                       # ghost of bot[j], attached to f_top[j]
    m = synth.Model(
        edge_var=np.concatenate([base.edge_var, bot, g_for_top, top, g_for_bot]),
        edge_fac=np.concatenate([base.edge_fac, f_bot, f_bot, f_top, f_top]),
        factor_ids=np.concatenate([base.factor_ids, f_bot, f_top]),
        factor_kind=np.concatenate([base.factor_kind, np.full(2 * n_cols, 1, np.int32)]),
        factor_var=np.concatenate([base.factor_var, qw, qw]),
        x_ids=base.x_ids, prior_var=base.prior_var, prior_fac=base.prior_fac, prior_mean=base.prior_mean,
        prior_variance=base.prior_variance, meta=dict(base.meta))
    part = Partition(model=m, rank=0, world=1,
                     send_var=np.concatenate([top, bot]), send_fac=np.concatenate([f_top, f_bot]),
                     recv_var=np.concatenate([g_for_top, g_for_bot]), recv_fac=np.concatenate([f_bot, f_top]),
                     peers=[Peer(0, slice(0, 2 * n_cols), slice(0, 2 * n_cols))])
    return part, (top, bot, qw)


def deep_self(n_rows: int, n_cols: int, depth: int, seed: int = 1234) -> Partition:
    """One rank that is its own deep-halo neighbour: the whole grid, whose first and last `depth` rows are both exported
    and imported (send list == recv list, peer = rank 0).  The exchange then rewrites those messages with their own
    values: the sweeps must equal the un-partitioned ones bit for bit, while pack, RCCL send/recv and unpack move exactly
    the volume a middle rank of a strip partition moves.  For exercising and timing the deep-halo path on one GPU."""
    model = synth.gaussian_grid(n_rows, n_cols, seed=seed)
    ev, ef = np.asarray(model.edge_var, np.int64), np.asarray(model.edge_fac, np.int64)
    row = (ev - 1) // n_cols
    m = (row < depth) | (row >= n_rows - depth)
    v, f = ev[m], ef[m]
    o = np.lexsort((f, v))
    v, f = v[o], f[o]
    # the exchanged rows count as layer 1 — "rewritten by an exchange" — whatever their distance from the edge, so that every
    # sweep of a batch still runs them (they are real rows here: trimming must not drop them) while cx_halo_exchange_sweep sees
    # which slices hold no exchanged variable and can run beside the exchange
    xs = np.asarray(model.x_ids, np.int64)
    xrow = (xs - 1) // n_cols
    lv = xs[(xrow < depth) | (xrow >= n_rows - depth)]
    return Partition(model=model, rank=0, world=1, send_var=v, send_fac=f, recv_var=v, recv_fac=f,
                     peers=[Peer(0, slice(0, len(v)), slice(0, len(v)))], depth=depth, owned_x=np.asarray(model.x_ids),
                     layer_var=lv, layer=np.ones(len(lv), np.int32))


class RcclExchange:
    """The partitioned sweep with the exchange issued by the library on RCCL (cx_sweep_exchange): Python only hands
    over the peer table and the ncclUniqueId (broadcast through torch.distributed when there is more than one rank)."""

    def __init__(self, dev, part: Partition, dist=None, torch=None, device=None):
        self.dev = dev
        dev.halo_configure(part.send_var, part.send_fac, part.recv_var, part.recv_fac)
        dev.halo_peers([(p.rank, p.send.start, p.send.stop - p.send.start, p.recv.start, p.recv.stop - p.recv.start)
                        for p in part.peers])
        self.send = self.recv = None
        if torch is not None and device is not None:
            # caller-visible halo buffers (the library packs into / unpacks from them), so that a driver can audit what
            # the library-issued exchange moved (verify_last_exchange)
            self.send = torch.zeros((max(len(part.send_var), 1), 2), dtype=torch.float64, device=device)
            self.recv = torch.zeros((max(len(part.recv_var), 1), 2), dtype=torch.float64, device=device)
            dev.halo_set_buffers(self.send.data_ptr(), self.recv.data_ptr())
        if part.world > 1:
            idt = torch.zeros(128, dtype=torch.uint8, device=device)
            if part.rank == 0:
                idt.copy_(torch.frombuffer(bytearray(dev.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, 0)
            uid = bytes(idt.cpu().numpy().tobytes())
        else:
            uid = dev.comm_unique_id()
        dev.comm_init(part.world, part.rank, uid)

    def sweep(self, n: int = 1):
        self.dev.sweep_exchange(n)


class DeepHaloRccl:
    """Deep-halo partition driven by the library: one cx_halo_state_exchange (pack, grouped RCCL send/recv, unpack, all on
    the handle's stream) before every `depth` plain sweeps."""

    def __init__(self, dev, part: Partition, dist=None, torch=None, device=None, overlap: bool = False):
        """overlap: cx_halo_exchange_sweep (the exchange on a second stream beside the owned part of the batch's first sweep) instead
        of cx_halo_state_exchange + cx_sweep.  Bit-identical; measured SLOWER on one MI355X (13.2 against 11.1-11.5 us per sweep for
        a 1/8 strip of C4 at depth 16: the two cross-stream hand-offs and the split sweep cost more than the 9 us of sweep the
        exchange hides behind), hence off by default."""
        self.dev, self.depth, self.k, self.overlap = dev, part.depth, 0, overlap
        dev.halo_configure_state(part.send_var, part.send_fac, part.recv_var, part.recv_fac)
        if part.layer_var is not None and part.depth and dev.dim == 1:
            dev.halo_set_layers(part.layer_var, part.layer, part.depth)    # trimmed sweeps between exchanges
        dev.halo_peers([(p.rank, p.send.start, p.send.stop - p.send.start, p.recv.start, p.recv.stop - p.recv.start)
                        for p in part.peers])
        self.send = self.recv = None
        if torch is not None and device is not None:
            w = getattr(dev, "halo_doubles", 2)
            self.send = torch.zeros((max(len(part.send_var), 1), w), dtype=torch.float64, device=device)
            self.recv = torch.zeros((max(len(part.recv_var), 1), w), dtype=torch.float64, device=device)
            dev.halo_set_buffers(self.send.data_ptr(), self.recv.data_ptr())
        if part.world > 1:
            idt = torch.zeros(128, dtype=torch.uint8, device=device)
            if part.rank == 0:
                idt.copy_(torch.frombuffer(bytearray(dev.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, 0)
            uid = bytes(idt.cpu().numpy().tobytes())
        else:
            uid = dev.comm_unique_id()
        dev.comm_init(part.world, part.rank, uid)

    def sweep(self, n: int = 1):
        """n sweeps; the sweeps between two exchanges go to the library as ONE cx_sweep call (at 1/8 of the C4 grid a sweep is
        a few microseconds of kernel: a Python call per sweep would be the bound)"""
        while n > 0:
            run = min(n, self.depth - self.k % self.depth)
            if self.k % self.depth == 0 and self.overlap:
                self.dev.halo_exchange_sweep(run)       # the exchange rides beside the owned part of the batch's first sweep
            else:
                if self.k % self.depth == 0:
                    self.dev.halo_state_exchange()
                self.dev.sweep(run)
            self.k += run
            n -= run


class DeepHaloIpc:
    """Deep-halo partition with the exchange done by the library WITHOUT a collective: every rank pushes its boundary state
    straight into its neighbours' IPC-mapped receive areas and raises an epoch flag there (cx_api_ipc.hip) — ONE launch per
    exchange on the handle's stream.  Bit-identical to DeepHaloRccl.  `dist` is only used once, to carry the 64-byte memory
    handles between the ranks (all_gather_object) — and by `audit`; a rank that is its own neighbour, and handles of the same
    process (`connect=False`, then `connect({rank: info})`), connect by device address.  A rank whose allocation or connection
    fails raises on EVERY rank after the handles have been gathered, so that all ranks can fall back together."""

    def __init__(self, dev, part: Partition, dist=None, torch=None, device=None, connect=True, overlap: bool = False, early_push: bool = False,
                 peers_on_other_devices: bool = False):
        """overlap: cx_halo_ipc_exchange_sweep — the owned part of a batch's first sweep between the push and the unpack (the
        neighbours' pushes travel meanwhile).  early_push: cx_halo_ipc_batch — additionally the NEXT exchange is pushed inside the
        last sweep of every full batch, as soon as the slices that write the boundary state have run (two partial sweeps of compute
        between a push and the wait for it).  Both bit-identical; both cost launches where nothing travels (one GPU).
        peers_on_other_devices: the caller's assertion that every neighbour pushes from another GPU — only then do push and unpack of
        an exchange share one launch (cx_halo_ipc_set_fused)."""
        self.dev, self.depth, self.k, self.part, self.fresh, self.overlap, self.early_push = dev, part.depth, 0, part, False, overlap, early_push
        dev.halo_configure_state(part.send_var, part.send_fac, part.recv_var, part.recv_fac)
        if part.layer_var is not None and part.depth and dev.dim == 1:
            dev.halo_set_layers(part.layer_var, part.layer, part.depth)    # trimmed sweeps between exchanges
        dev.halo_peers([(p.rank, p.send.start, p.send.stop - p.send.start, p.recv.start, p.recv.stop - p.recv.start)
                        for p in part.peers])
        self.send = self.recv = None
        self.info, err = None, None
        try:
            self.handle, self.base, self.area_bytes = dev.halo_ipc_alloc()
            dev.halo_ipc_set_fused(bool(peers_on_other_devices))
            # what a neighbour needs to push to me: my peer entries (index, sending rank, recv offset, count)
            self.info = {"rank": part.rank, "handle": self.handle, "base": self.base, "area_bytes": self.area_bytes, "pid": os.getpid(),
                         "entries": [(i, p.rank, p.recv.start, p.recv.stop - p.recv.start) for i, p in enumerate(part.peers)]}
        except Exception as e:
            if not (connect and part.world > 1):
                raise
            err = e
        if connect:
            if part.world > 1:
                infos = [None] * part.world
                dist.all_gather_object(infos, self.info)
                bad = [r for r, x in enumerate(infos) if x is None]
                if bad:
                    raise RuntimeError(f"IPC halo: rank(s) {bad} could not allocate / export their receive area" + (f" ({err})" if err else ""))
            else:
                infos = {part.rank: self.info}
            self.connect(infos)

    def connect(self, infos):
        """infos[rank] = the `info` of that rank's DeepHaloIpc.  The k-th peer entry of mine for rank q pairs with the k-th peer
        entry of q for me (the order in which a grouped send/recv would match them)."""
        part, seen = self.part, {}
        for i, p in enumerate(part.peers):
            k = seen.get(p.rank, 0)
            seen[p.rank] = k + 1
            theirs = infos[p.rank]
            match = [e for e in theirs["entries"] if e[1] == part.rank]
            if k >= len(match):
                raise ValueError(f"rank {p.rank} has no peer entry {k} for rank {part.rank}")
            entry, _, recv_off, recv_count = match[k]
            if recv_count != p.send.stop - p.send.start:
                raise ValueError(f"rank {part.rank} sends {p.send.stop - p.send.start} messages to rank {p.rank}, which expects {recv_count}")
            same = theirs["pid"] == os.getpid()
            self.dev.halo_ipc_connect(i, entry, recv_off, theirs["area_bytes"], handle=None if same else theirs["handle"],
                                      same_process_base=theirs["base"] if same else None)

    def sweep(self, n: int = 1):
        while n > 0:
            run = min(n, self.depth - self.k % self.depth)
            if self.k % self.depth == 0 and not self.fresh and self.early_push and run == self.depth and run >= 2:
                self.dev.halo_ipc_batch(run)         # a full batch: its last sweep carries the push of the next exchange
            elif self.k % self.depth == 0 and not self.fresh and (self.overlap or self.early_push):
                self.dev.halo_ipc_exchange_sweep(run)
            else:
                if self.k % self.depth == 0:
                    if self.fresh:
                        self.fresh = False          # `audit` has just made this exchange
                    else:
                        self.dev.halo_ipc_exchange()
                self.dev.sweep(run)
            self.k += run
            n -= run

    def check(self):
        """synchronises; raises when an unpack gave up waiting for a neighbour"""
        timed_out, n = self.dev.halo_ipc_status()
        if timed_out:
            raise RuntimeError(f"rank {self.part.rank}: a neighbour did not arrive at one of {n} IPC halo exchanges (stale redundant rows)")
        return n

    def audit(self, dist, torch, device) -> bool:
        """An exchange NOW (only between batches: the sweeps so far a multiple of the depth), then the comparison a caller-owned
        transport gets from verify_last_exchange: what this rank's redundant rows hold == what their owners hold, the owners'
        values carried a second time over torch.distributed.  Collective; False on any rank means False on all."""
        from . import _lib as L
        if self.k % self.depth != 0:
            raise RuntimeError("DeepHaloIpc.audit: between batches only")
        ok = True
        try:
            if not self.fresh:
                self.dev.halo_ipc_exchange()
            self.check()
        except Exception:
            ok = False
        self.fresh = True
        part = self.part
        send = torch.from_numpy(self.dev.get_messages(part.send_var, part.send_fac, L.TO_VARIABLE, L.FORM_NATURAL)).to(device)
        recv = torch.from_numpy(self.dev.get_messages(part.recv_var, part.recv_fac, L.TO_VARIABLE, L.FORM_NATURAL)).to(device)
        same = verify_last_exchange(part, send, recv, dist, torch)
        if dist is not None and part.world > 1:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item() == 1)
        return bool(same and ok)


class DeviceStateSweeper:
    """Adapter for a caller-owned transport of the deep halo: pack / unpack around torch tensors."""

    def __init__(self, dev, part: Partition, torch, device):
        self.dev = dev
        dev.halo_configure_state(part.send_var, part.send_fac, part.recv_var, part.recv_fac)
        if part.layer_var is not None and part.depth and dev.dim == 1:
            dev.halo_set_layers(part.layer_var, part.layer, part.depth)    # trimmed sweeps between exchanges
        w = getattr(dev, "halo_doubles", 2)      # doubles per message: the storage form of the handle's dim
        self.send = torch.zeros((max(len(part.send_var), 1), w), dtype=torch.float64, device=device)
        self.recv = torch.zeros((max(len(part.recv_var), 1), w), dtype=torch.float64, device=device)
        dev.halo_set_buffers(self.send.data_ptr(), self.recv.data_ptr())

    def pack(self):
        self.dev.halo_state_pack()
        self.dev.sync()

    def unpack(self):
        self.dev.halo_state_unpack()

    def sweep(self, n: int = 1):
        self.dev.sweep(n)


class HostStagedStateSweeper(DeviceStateSweeper):
    """Rehearsal transport (gloo cannot move device memory): the halo tensors are staged through the host."""

    def __init__(self, dev, part: Partition, torch, device):
        super().__init__(dev, part, torch, device)
        self.dsend, self.drecv = self.send, self.recv
        self.send = torch.zeros_like(self.dsend, device="cpu")
        self.recv = torch.zeros_like(self.drecv, device="cpu")

    def pack(self):
        super().pack()
        self.send.copy_(self.dsend)

    def unpack(self):
        self.drecv.copy_(self.recv)
        super().unpack()


class DeepHaloExchange:
    """Deep halo over torch.distributed: before every `depth` sweeps pack → isend/irecv with the partition neighbours →
    unpack.  `sweeper` provides pack / unpack / sweep and the `send` / `recv` tensors."""

    def __init__(self, sweeper, part: Partition, dist):
        self.sw, self.part, self.dist, self.k = sweeper, part, dist, 0

    def sweep(self, n: int = 1):
        dist, sw, depth = self.dist, self.sw, self.part.depth
        while n > 0:
            if self.k % depth == 0:
                sw.pack()
                ops = []
                for p in self.part.peers:
                    ops.append(dist.P2POp(dist.isend, sw.send[p.send], p.rank))
                    ops.append(dist.P2POp(dist.irecv, sw.recv[p.recv], p.rank))
                for w in (dist.batch_isend_irecv(ops) if ops else []):
                    w.wait()
                sw.unpack()
            run = min(n, depth - self.k % depth)
            sw.sweep(run)
            self.k += run
            n -= run


class DeviceSweeper:
    """Adapter: a DeviceGraph whose halo buffers are torch tensors (so torch.distributed can move them)."""

    def __init__(self, dev, part: Partition, torch, device):
        self.dev = dev
        dev.halo_configure(part.send_var, part.send_fac, part.recv_var, part.recv_fac)
        self.send = torch.zeros((max(len(part.send_var), 1), 2), dtype=torch.float64, device=device)
        self.recv = torch.zeros((max(len(part.recv_var), 1), 2), dtype=torch.float64, device=device)
        dev.halo_set_buffers(self.send.data_ptr(), self.recv.data_ptr())

    def sweep_begin(self):
        self.dev.sweep_begin()

    def sweep_main(self):
        self.dev.sweep_main()

    def sweep_end(self):
        self.dev.sweep_end()


class HostStagedSweeper(DeviceSweeper):
    """Rehearsal transport: the halo tensors live on the host (gloo cannot move device memory); every sweep copies the
    packed messages device -> host before the exchange and host -> device after it.  Only for exercising the multi-rank
    control flow on a box with fewer GPUs than ranks — never a measured configuration."""

    def __init__(self, dev, part: Partition, torch, device):
        super().__init__(dev, part, torch, device)
        self.dsend, self.drecv = self.send, self.recv
        self.send = torch.zeros_like(self.dsend, device="cpu")
        self.recv = torch.zeros_like(self.drecv, device="cpu")

    def sweep_begin(self):
        self.dev.sweep_begin()
        self.dev.sync()
        self.send.copy_(self.dsend)

    def sweep_end(self):
        self.drecv.copy_(self.recv)
        self.dev.sweep_end()


class HaloExchange:
    """One partitioned sweep = begin (pack the exported messages) → start send/recv with the partition neighbours →
    main sweep (overlaps the exchange) → wait → end (unpack + push the imported messages through the cut factors).
    `sweeper` provides sweep_begin/main/end and the `send` / `recv` tensors; `dist` is torch.distributed."""

    def __init__(self, sweeper, part: Partition, dist):
        self.sw, self.part, self.dist = sweeper, part, dist

    def sweep(self, n: int = 1):
        dist, sw = self.dist, self.sw
        for _ in range(n):
            sw.sweep_begin()
            ops = []
            for p in self.part.peers:
                ops.append(dist.P2POp(dist.isend, sw.send[p.send], p.rank))
                ops.append(dist.P2POp(dist.irecv, sw.recv[p.recv], p.rank))
            works = dist.batch_isend_irecv(ops) if ops else []
            sw.sweep_main()
            for w in works:
                w.wait()
            sw.sweep_end()


# ---- chain-scan partition (contiguous time blocks + one composed map per block) ------------------------------
class TreeRegionExchange:
    """A FOREST cut at its variables, every rank under an EXACT local schedule (CX_SCHED_TREE; round 5).

    `by_assignment` gives every rank its variables, every factor that touches one of them and, for a cut factor, the far variable as a
    degree-1 stand-in.  A stand-in's message into the cut factor has no dependencies on this rank (a variable of degree 1,
    src/dependencies.jl:48-55): it is whatever was stored — here, what the owning rank computed.  One round = one exact local sweep
    (every message of the piece from final local inputs), then the boundary variables' messages into the cut factors travel and become
    the stand-ins' messages.  After r rounds every message that depends on at most r regions is exact; on a forest the rounds stop
    changing anything after (regions on the longest path of the region tree) rounds, at the exact posterior — which is what the
    un-partitioned tree schedule computes in one sweep.  The boundary is one message per cut factor and direction, staged through the
    host (a few messages per round).  `dist`: torch.distributed (gloo / nccl with CPU tensors) or anything with its P2P surface."""

    def __init__(self, dev, part: Partition, dist, torch):
        self.dev, self.part, self.dist, self.torch = dev, part, dist, torch
        # the far variables of the cut factors are stand-ins (cx_halo_configure flags them): constants hanging off their factors for the tree
        # plan — level by level or over heavy paths — exactly like observed variables; what they send is what this class stores there
        dev.halo_configure(part.send_var, part.send_fac, part.recv_var, part.recv_fac)
        self.width = int(dev.lib.cx_payload_doubles(dev.dim, L.FORM_NATURAL))
        self.send = torch.zeros((max(len(part.send_var), 1), self.width), dtype=torch.float64)
        self.recv = torch.full((max(len(part.recv_var), 1), self.width), float("nan"), dtype=torch.float64)
        self.rounds = 0

    def round(self) -> bool:
        """one exact local sweep + one exchange; True when an imported message changed"""
        dev, part, dist, torch = self.dev, self.part, self.dist, self.torch
        dev.sweep(1)
        n_s, n_r = len(part.send_var), len(part.recv_var)
        if n_s:
            # the boundary variables' messages INTO the cut factors: nobody on this rank listens to them (the factor's other variable is a
            # stand-in), so the lazy tree plan does not compute them — the exchange asks for them, as items over the sweep's final messages
            dev.update_batch([L.ITEM_MESSAGE_TO_FACTOR] * n_s, part.send_var, part.send_fac)
            self.send[:n_s] = torch.from_numpy(dev.get_messages(part.send_var, part.send_fac, L.TO_FACTOR, L.FORM_NATURAL)[:, :self.width])
        before = self.recv.clone()
        ops = []
        for p in part.peers:
            ops.append(dist.P2POp(dist.isend, self.send[p.send], p.rank))
            ops.append(dist.P2POp(dist.irecv, self.recv[p.recv], p.rank))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        changed = not bool(torch.equal(torch.nan_to_num(before, nan=0.12345), torch.nan_to_num(self.recv, nan=0.12345)))
        if n_r and changed:
            dev.set_messages(part.recv_var, part.recv_fac, L.TO_FACTOR, L.FORM_NATURAL, self.recv[:n_r].numpy())
        self.rounds += 1
        return changed

    def solve(self, max_rounds: int = 64) -> int:
        """rounds until no rank imports anything new (agreed by an all-gather of one flag per rank), then one last local sweep"""
        torch, dist = self.torch, self.dist
        for _ in range(max_rounds):
            mine = torch.tensor([1.0 if self.round() else 0.0], dtype=torch.float64)
            flags = [torch.zeros(1, dtype=torch.float64) for _ in range(self.part.world)]
            dist.all_gather(flags, mine)
            if not any(float(f[0]) for f in flags):
                return self.rounds
        raise RuntimeError(f"TreeRegionExchange: the boundary messages still moved after {max_rounds} rounds (is the graph a forest?)")


def _lin_apply(M, m):
    """the projective-linear map (e f g A B C; D = 1) of csrc/cx_chain.hip applied to a natural-form message (xi, w)"""
    e, f, g, A, B, Cc = M
    den = Cc * m[1] + 1.0
    return np.array([(e * m[0] + f * m[1] + g) / den, (A * m[1] + B) / den])


def _rule_additive(q, m):
    """factor→variable rule of an additive factor on a natural-form message (test/inference_engine_tests.jl:426-427)"""
    s = 1.0 / (1.0 + q * m[1])
    return np.array([m[0] * s, m[1] * s])


# ---- d-dimensional blocks (dim 2..4): the maps of csrc/cx_mvchain.hip on the host (what the all-gathered rows are put together with)
def _mv_sym(p, d):
    """packed upper triangle (row by row) -> symmetric d x d"""
    S = np.zeros((d, d))
    S[np.triu_indices(d)] = p
    return S + np.triu(S, 1).T


def _mv_map_apply(M, d, eta, lam):
    """f(eta, Lambda) = (c + B (Lambda + P)^-1 (eta + h), C - B (Lambda + P)^-1 B') for the map M = P | B | C | h | c"""
    nt = d * (d + 1) // 2
    P, B, Cm = _mv_sym(M[:nt], d), M[nt:nt + d * d].reshape(d, d), _mv_sym(M[nt + d * d:2 * nt + d * d], d)
    h, c = M[2 * nt + d * d:2 * nt + d * d + d], M[2 * nt + d * d + d:]
    BW = np.linalg.solve(lam + P, B.T).T          # B (Lambda + P)^-1 (the matrix is symmetric)
    return c + BW @ (eta + h), Cm - BW @ B.T


_MV_TABLES = {}


def _mv_rule_tables(A, Q, forward):
    """(P, B, C) of x_out = A x_in + N(0, Q) for the receiving end (csrc/cx_mv_core.h: mv_rule_tables), cached per matrix pair.
    The key is the matrices' CONTENT (a digest of their bytes): an address can be recycled by a temporary that holds another matrix with
    the same corner entries, and a stale (P, B, C) would be a silently wrong boundary message of a d = 64 block exchange."""
    import hashlib

    A, Q = np.ascontiguousarray(A, dtype=np.float64), np.ascontiguousarray(Q, dtype=np.float64)
    key = (hashlib.blake2b(A.tobytes() + Q.tobytes(), digest_size=16).digest(), A.shape, bool(forward))
    tab = _MV_TABLES.get(key)
    if tab is None:
        Qi = np.linalg.inv(Q)
        tab = (A.T @ Qi @ A, Qi @ A, Qi) if forward else (Qi, A.T @ Qi, A.T @ Qi @ A)
        if len(_MV_TABLES) > 64:
            _MV_TABLES.clear()
        _MV_TABLES[key] = tab
    return tab


def _mv_rule(eta, lam, A, Q, forward):
    """factor→variable rule of x_out = A x_in + N(0, Q) on a natural-form message (cx_mv.hip): forward = the receiver is the OUT edge"""
    P, B, Cm = _mv_rule_tables(A, Q, forward)
    BW = np.linalg.solve(lam + P, B.T).T          # B (Lambda + P)^-1 (the matrix is symmetric)
    return BW @ eta, Cm - BW @ B.T


def _rule_linear(abq, m):
    """factor→variable rule with the RECEIVING edge's effective (a, b, q) on a natural-form message (csrc/cx_kernels.hip:
    factor_rule): s = 1 / (a² + q w);  (xi, w) -> ((a xi + b w) s, w s).  Additive factors are (1, 0, q)."""
    a, b, q = abq
    s = 1.0 / (a * a + q * m[1])
    return np.array([(a * m[0] + b * m[1]) * s, m[1] * s])


def _effective_abq(params, receiver_is_out):
    """x_out = a x_in + b + N(0, q): the parameters the receiving edge applies — forward (receiver = out) {a, b, q}, backward
    (receiver = in) {1/a, -b/a, q/a²} (cx_graph_create)"""
    p = np.atleast_1d(np.asarray(params, float))
    q, a, b = (p[0], 1.0, 0.0) if len(p) < 3 or p[1] == 0.0 else (p[0], p[1], p[2])
    return (a, b, q) if receiver_is_out else (1.0 / a, -b / a, q / (a * a))


class ChainScanExchange:
    """A state-space chain cut into contiguous time blocks, one chain-scan handle per rank (`partition.contiguous_blocks(model,
    rank, world)`: the block's variables, the cut transition factors, the remote end of each as a degree-1 stand-in).

    One `update()` = the exact forward/backward result of the WHOLE chain on every rank's block:
      1. every rank asks its handle for the block's composed forward and backward maps and the side sums of its end variables
         (`cx_chain_block_maps`: the scan's first two kernels), with the cut messages out of the picture;
      2. ONE all-gather of 22 doubles per rank (the only collective);
      3. every rank applies the maps of the blocks before it (after it) to the empty message and obtains the one message that
         enters its block from the left (right): the stand-ins' variable→factor messages;
      4. one local `cx_sweep(1)`.
    `block` provides chain_block_maps / set_messages / sweep (a DeviceGraph, or the CPU stand-in of the tests); `dist` is
    torch.distributed (or None for world 1).

    Scalar chains (additive and — round 3 — linear factors x_out = a x_in + b + noise) and, since round 3, d-dimensional chains (dim 2..4: the maps (P, B, C, h, c) of
    csrc/cx_mvchain.hip, 2 ND + 2 nc + 4 doubles per rank in the all-gather: 120 for d = 4) and — round 4 — dim 64 (the ONE potential of a
    block's two end variables out of the composition tree of csrc/cx_mv64chain.hip: 21,060 doubles per rank)."""

    def __init__(self, block, part: Partition, dist, torch, device="cpu"):
        self.block, self.part, self.dist, self.torch, self.device = block, part, dist, torch, device
        # cx_chain_block_maps composes the maps of a block's LINKS: a block of one latent variable has none (the library refuses it
        # with "must form ONE path").  Say so here, where the cause is visible: more ranks than the chain has pairs of states.
        owned = np.setdiff1d(np.intersect1d(np.asarray(part.model.x_ids), np.asarray(part.model.edge_var)), np.asarray(part.recv_var))
        if len(owned) < 2:
            raise ValueError(f"ChainScanExchange: rank {part.rank} of {part.world} holds {len(owned)} latent variable(s); a time block needs at "
                             "least two (one link) — use fewer ranks for a chain this short")
        fv = dict(zip(np.asarray(part.model.factor_ids).tolist(), np.asarray(part.model.factor_var).tolist()))
        self.dim = int(getattr(part.model, "dim", 1))
        role = {}
        if part.model.edge_role is not None:
            role = {(int(v), int(f)): int(r) for v, f, r in zip(part.model.edge_var, part.model.edge_fac, part.model.edge_role)}
        # (stand-in variable, cut factor, own end variable, factor variance | parameter set, own end variable is the OUT edge)
        self.left = self.right = None
        for p in part.peers:
            assert p.send.stop - p.send.start == 1 and p.recv.stop - p.recv.start == 1, "a chain block has one cut factor per neighbour"
            cut = int(part.recv_fac[p.recv.start])
            own = int(part.send_var[p.send.start])
            rec = (int(part.recv_var[p.recv.start]), cut, own, fv[cut], role.get((own, cut), 0) == 0)      # ROLE_OUT == 0
            if p.rank < part.rank:
                self.left = rec
            else:
                self.right = rec
        if self.dim > 1:      # the library has to know which variables are stand-ins (they are not observed, and not part of the chain)
            block.halo_configure(part.send_var, part.send_fac, part.recv_var, part.recv_fac)

    def update(self):
        if self.dim > 1:
            return self._update_mv()
        from . import _lib as L

        blk, nan2, zero2 = self.block, [float("nan")] * 2, [0.0, 0.0]
        for rec in (self.left, self.right):
            if rec is not None:      # the cut messages out of the picture: nothing enters, nothing is known about the stand-in
                blk.set_messages([rec[2]], [rec[1]], L.TO_VARIABLE, L.FORM_NATURAL, zero2)
                blk.set_messages([rec[0]], [rec[1]], L.TO_FACTOR, L.FORM_NATURAL, nan2)
        fwd, bwd, s_first, s_last, _v0, _v1, _nl = blk.chain_block_maps()
        # the rule a message goes through when it ENTERS this block over a cut factor: the effective (a, b, q) of the own end's edge
        eff = [(1.0, 0.0, 0.0) if r is None else _effective_abq(r[3], r[4]) for r in (self.left, self.right)]
        mine = np.concatenate([fwd, bwd, s_first, s_last, eff[0], eff[1]])
        world, rank = self.part.world, self.part.rank
        if world > 1:
            t = self.torch.from_numpy(mine.copy()).to(self.device)
            got = [self.torch.zeros_like(t) for _ in range(world)]
            self.dist.all_gather(got, t)
            rows = np.stack([g.cpu().numpy() for g in got])
        else:
            rows = mine[None, :]
        # message that leaves block r on its right end towards the cut factor, given what entered it from the left
        out = np.zeros(2)
        left_in = None
        for r in range(world):
            F, sl, abqL = rows[r, 0:6], rows[r, 14:16], rows[r, 16:19]
            if r == rank:
                left_in = out.copy() if r > 0 else None
            enter = _rule_linear(abqL, out) if r > 0 else np.zeros(2)     # through the cut factor; block 0 starts from the empty message
            out = _lin_apply(F, enter) + sl
        out = np.zeros(2)
        right_in = None
        for r in range(world - 1, -1, -1):
            B, sf, abqR = rows[r, 6:12], rows[r, 12:14], rows[r, 19:22]
            if r == rank:
                right_in = out.copy() if r < world - 1 else None
            enter = _rule_linear(abqR, out) if r < world - 1 else np.zeros(2)
            out = _lin_apply(B, enter) + sf
        if self.left is not None:
            blk.set_messages([self.left[0]], [self.left[1]], L.TO_FACTOR, L.FORM_NATURAL, left_in)
        if self.right is not None:
            blk.set_messages([self.right[0]], [self.right[1]], L.TO_FACTOR, L.FORM_NATURAL, right_in)
        blk.sweep(1)


def _chain_scan_exchange_update_mv(self):
    """ChainScanExchange.update for dim 2..4"""
    from . import _lib as L

    blk, d, psets = self.block, self.dim, self.part.model.psets
    nt = d * (d + 1) // 2
    nd, ns = 2 * nt + d * d + 2 * d, d + nt
    zero, nan = np.zeros(d + d * d), np.full(d + d * d, np.nan)
    # dim 64 keeps ONE message buffer and no stored variable→factor messages: what enters a block is handed over as the cut factor's
    # message INTO the block's end variable (the rule through the cut factor applied here, on the host); dim 2..4 hand over the
    # stand-in's message into the cut factor and let the device apply the rule
    after_cut = d == 64
    for rec in (self.left, self.right):
        if rec is not None:      # the cut messages out of the picture: nothing enters, nothing is known about the stand-in
            blk.set_messages([rec[2]], [rec[1]], L.TO_VARIABLE, L.FORM_NATURAL, zero)
            if not after_cut:
                blk.set_messages([rec[0]], [rec[1]], L.TO_FACTOR, L.FORM_NATURAL, nan)
    fwd, bwd, s_first, s_last, _v0, _v1, _nl = blk.chain_block_maps()
    cuts = [(-1.0, 0.0) if r is None else (float(int(r[3])), 1.0 if r[4] else 0.0) for r in (self.left, self.right)]
    mine = np.concatenate([fwd, bwd, s_first, s_last, cuts[0], cuts[1]])
    world, rank = self.part.world, self.part.rank
    if world > 1:
        t = self.torch.from_numpy(mine.copy()).to(self.device)
        got = [self.torch.zeros_like(t) for _ in range(world)]
        self.dist.all_gather(got, t)
        rows = np.stack([g.cpu().numpy() for g in got])
    else:
        rows = mine[None, :]
    side = lambda p: (p[:d].copy(), _mv_sym(p[d:], d))       # noqa: E731
    o = 2 * nd + 2 * ns
    # what leaves block r on its right end towards the cut factor, given what entered it from the left
    out, left_in = None, None
    for r in range(world):
        if r == rank:
            left_in = out
        if r > 0:         # through the cut factor into block r's first variable (receiver: that variable)
            A, Q = psets[int(rows[r, o])]
            enter = _mv_rule(out[0], out[1], np.asarray(A, float), np.asarray(Q, float), rows[r, o + 1] == 1.0)
        else:
            enter = (np.zeros(d), np.zeros((d, d)))          # block 0 starts from the empty message
        e, lam = _mv_map_apply(rows[r, :nd], d, *enter)
        se, sl = side(rows[r, 2 * nd + ns:2 * nd + 2 * ns])
        out = (e + se, lam + sl)
    out, right_in = None, None
    for r in range(world - 1, -1, -1):
        if r == rank:
            right_in = out
        if r < world - 1:
            A, Q = psets[int(rows[r, o + 2])]
            enter = _mv_rule(out[0], out[1], np.asarray(A, float), np.asarray(Q, float), rows[r, o + 3] == 1.0)
        else:
            enter = (np.zeros(d), np.zeros((d, d)))
        e, lam = _mv_map_apply(rows[r, nd:2 * nd], d, *enter)
        se, sl = side(rows[r, 2 * nd:2 * nd + ns])
        out = (e + se, lam + sl)
    for rec, m in ((self.left, left_in), (self.right, right_in)):
        if rec is None:
            continue
        if after_cut:
            A, Q = psets[int(rec[3])]
            m = _mv_rule(m[0], m[1], np.asarray(A, float), np.asarray(Q, float), bool(rec[4]))
            blk.set_messages([rec[2]], [rec[1]], L.TO_VARIABLE, L.FORM_NATURAL, np.concatenate([m[0], m[1].ravel()]))
        else:
            blk.set_messages([rec[0]], [rec[1]], L.TO_FACTOR, L.FORM_NATURAL, np.concatenate([m[0], m[1].ravel()]))
    blk.sweep(1)


ChainScanExchange._update_mv = _chain_scan_exchange_update_mv


def verify_last_exchange(part: Partition, send, recv, dist, torch) -> bool:
    """Audit of the most recent halo exchange, whatever transport made it: every rank re-sends the slices of its packed
    send buffer to its partition neighbours over torch.distributed and compares what arrives, bit for bit, with the
    slices its recv buffer holds.  Collective over the partition's ranks; call it only when no sweep is in flight."""
    ok = True
    if part.world == 1:
        for p in part.peers:           # a rank that is its own neighbour (periodic cut)
            ok = ok and bool(torch.equal(recv[p.recv.start:p.recv.stop], send[p.send.start:p.send.stop]))
        return ok
    ops, got = [], []
    for p in part.peers:
        tmp = torch.empty_like(recv[p.recv.start:p.recv.stop])
        got.append((p, tmp))
        ops.append(dist.P2POp(dist.isend, send[p.send.start:p.send.stop].contiguous(), p.rank))
        ops.append(dist.P2POp(dist.irecv, tmp, p.rank))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for p, tmp in got:
        ok = ok and bool(torch.equal(tmp, recv[p.recv.start:p.recv.stop]))
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=send.device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item() == 1)


def converge(exchange, local_residual, dist, torch, tol: float, max_sweeps: int, check_every: int, device="cpu"):
    """Partitioned sweeps until the largest message change over `check_every` sweeps, maximised over all ranks, is <= tol.
    The one collective on this path: an 8-byte MAX all-reduce every `check_every` sweeps; every rank takes the
    same decision, so the halo exchanges stay matched.  `local_residual()` returns this rank's max change since its last call
    (DeviceGraph.residual).  Returns (sweeps run, global residual)."""
    local_residual()                       # snapshot of the starting point
    done, r = 0, float("inf")
    while done < max_sweeps:
        k = min(check_every, max_sweeps - done)
        exchange.sweep(k)
        done += k
        r = float(local_residual())
        if dist is not None:
            t = torch.tensor([r if r == r else float("inf")], dtype=torch.float64, device=device)   # NaN (undefined messages) never converges
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            r = float(t.item())
        if r <= tol:
            break
    return done, r
