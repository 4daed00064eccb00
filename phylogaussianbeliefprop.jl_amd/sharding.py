"""Site sharding across ranks: independent sites are partitioned contiguously,
each rank owns one engine with its share as `n_sites`, calibration needs NO communication, and the only
collective is the final all-gather of the per-site log-likelihoods (a few KB: latency-bound, one call)."""
import numpy as np


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous, balanced partition: the first n_total % world ranks get one extra site."""
    base, extra = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_sites(local, n_total: int, dist=None, device="cpu"):
    """All-gather per-site values (1-D float64 of this rank's shard) into the full [n_total] vector on every
    rank.  dist = torch.distributed (initialised: nccl = RCCL on ROCm, or gloo on CPU) or None (single rank)."""
    local = np.ascontiguousarray(local, dtype=np.float64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        assert local.shape[0] == n_total
        return local.copy()
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    width = -(-n_total // world)  # equal-size slots: one fused collective, padded
    buf = torch.zeros(width, dtype=torch.float64, device=device)
    lo, hi = shard_range(n_total, rank, world)
    assert local.shape[0] == hi - lo
    buf[: hi - lo] = torch.from_numpy(local).to(device)
    out = torch.empty(world * width, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, buf)
    out = out.cpu().numpy().reshape(world, width)
    full = np.empty(n_total)
    for r in range(world):
        a, b = shard_range(n_total, r, world)
        full[a:b] = out[r, : b - a]
    return full


# ---------------------------------------------------------------------------------------------------------------------
# Behind the C ABI (include/pgbp.h, "several GPUs"; csrc/pgbp_dist.hip): what a Julia host binds.  The two classes below
# are thin ctypes mirrors used by bench.py and the tests.

class EngineGroup:
    """pgbp_group: ONE process, one engine per listed device over contiguous site ranges; every call fans out on one
    host thread per device.  `devices` may repeat a device (rehearsal on a one-GPU box)."""

    def __init__(self, dims, sepset_clusters, scope_off, scope_idx, n_sites, devices):
        import ctypes as C
        from . import _lib as L
        self._C, self._L = C, L
        self.lib = L.load()
        desc, self._keep = L.make_desc(dims, sepset_clusters, scope_off, scope_idx, n_sites, 0)
        self.n_sites = int(n_sites)
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        self._g = C.c_void_p()
        code = self.lib.pgbp_group_create(C.byref(desc), len(dev), L.i32p(dev), C.byref(self._g))
        if code != 0:
            raise L.PgbpError(code, self.lib.pgbp_group_last_error(None).decode())
        self.packed_size = int(self.lib.pgbp_packed_size(self.lib.pgbp_group_engine(self._g, 0)))

    def _check(self, code):
        if code != 0:
            raise self._L.PgbpError(code, self.lib.pgbp_group_last_error(self._g).decode())

    def close(self):
        if self._g:
            self.lib.pgbp_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self):
        return int(self.lib.pgbp_group_size(self._g))

    def range(self, shard):
        a, n = self._C.c_int32(), self._C.c_int32()
        self._check(self.lib.pgbp_group_range(self._g, shard, self._C.byref(a), self._C.byref(n)))
        return a.value, n.value

    def engine(self, shard):
        return self.lib.pgbp_group_engine(self._g, shard)

    def set_schedule(self, schedule):
        L = self._L
        off = np.zeros(len(schedule) + 1, np.int32)
        for i, s in enumerate(schedule):
            off[i + 1] = off[i] + len(s[0])
        pa = np.ascontiguousarray(np.concatenate([np.asarray(s[0]) for s in schedule]).astype(np.int32))
        ch = np.ascontiguousarray(np.concatenate([np.asarray(s[1]) for s in schedule]).astype(np.int32))
        self._check(self.lib.pgbp_group_set_schedule(self._g, len(schedule), L.i32p(off), L.i32p(pa), L.i32p(ch)))

    def set_beliefs(self, packed, snapshot_factors=True):
        packed = np.ascontiguousarray(packed, dtype=np.float64)
        assert packed.shape == (self.n_sites, self.packed_size)
        self._check(self.lib.pgbp_group_set_beliefs(self._g, self._L.f64p(packed), 1 if snapshot_factors else 0))

    def get_beliefs(self):
        out = np.zeros((self.n_sites, self.packed_size))
        self._check(self.lib.pgbp_group_get_beliefs(self._g, self._L.f64p(out)))
        return out

    def calibrate(self, niter=1, opts=None):
        L = self._L
        res = (L.Result * self.n_sites)()
        o = opts if opts is not None else L.Opts(0, 1, 0, 0, 1e-5)
        self._check(self.lib.pgbp_group_calibrate(self._g, int(niter), self._C.byref(o), res))
        return res

    def integrate(self, belief, dim):
        mu = np.zeros((self.n_sites, max(1, dim)))
        norm = np.zeros(self.n_sites)
        info = np.zeros(self.n_sites, np.int32)
        self._check(self.lib.pgbp_group_integrate(self._g, int(belief), self._L.f64p(mu), self._L.f64p(norm), self._L.i32p(info)))
        return mu, norm, info

    def enqueue_calibrate(self, reps, reset_each=0, opts=None):
        o = opts if opts is not None else self._L.Opts(0, 1, 0, 0, 1e-5)
        self._check(self.lib.pgbp_group_enqueue_calibrate(self._g, int(reps), int(reset_each), self._C.byref(o)))

    def enqueue_loglik(self, reps=1, opts=None, lg=False):
        o = opts if opts is not None else self._L.Opts(0, 1, 0, 0, 1e-5)
        fn = self.lib.pgbp_group_enqueue_loglik_lg if lg else self.lib.pgbp_group_enqueue_loglik
        self._check(fn(self._g, int(reps), self._C.byref(o)))

    def fetch_loglik(self):
        norm = np.zeros(self.n_sites)
        info = np.zeros(self.n_sites, np.int32)
        self._check(self.lib.pgbp_group_fetch_loglik(self._g, self._L.f64p(norm), self._L.i32p(info)))
        return norm, info

    def sync(self):
        self._check(self.lib.pgbp_group_sync(self._g))


class PatternGroup:
    """pgbp_patterns: sites with different missing-data patterns -- different scopes, hence different belief dimensions
    (allocatebeliefs, src/beliefs.jl:551-559) -- behind one handle: one engine per pattern, the sites of a pattern batched
    inside it, per-site results in the caller's site order.
    patterns: list of (dims, sepset_clusters, scope_off, scope_idx, sites) -- the description arrays of include/pgbp.h for
    the pattern's scopes and the (global) indices of the sites that have it.  self.beliefs[k]: a ClusterGraphBelief over
    pattern k's engine (pgbp_patterns_engine) for everything pattern-specific: beliefs, lg_setup, assignfactors_lg_."""

    def __init__(self, patterns, device=0):
        import ctypes as C
        from . import _lib as L
        from .clustergraphbeliefs import ClusterGraphBelief
        self._C, self._L = C, L
        self.lib = L.load()
        descs, self._keep = [], []
        sites = []
        for dims, sc, so, si, st in patterns:
            d, k = L.make_desc(dims, sc, so, si, len(st), device)
            descs.append(d)
            self._keep.append(k)
            sites.extend(int(x) for x in st)
        self.n_sites = len(sites)
        self.sites = np.ascontiguousarray(sites, dtype=np.int32)
        arr = (C.POINTER(L.Desc) * len(descs))(*[C.pointer(d) for d in descs])
        self._descs = descs
        self._g = C.c_void_p()
        code = self.lib.pgbp_patterns_create(len(descs), arr, L.i32p(self.sites), C.byref(self._g))
        if code != 0:
            raise L.PgbpError(code, self.lib.pgbp_patterns_last_error(None).decode())
        self.beliefs = [ClusterGraphBelief.from_arrays(dims, sc, so, si, None, n_sites=len(st), device=device,
                                                       engine=self.lib.pgbp_patterns_engine(self._g, k))
                        for k, (dims, sc, so, si, st) in enumerate(patterns)]

    def _check(self, code):
        if code != 0:
            raise self._L.PgbpError(code, self.lib.pgbp_patterns_last_error(self._g).decode())

    def close(self):
        if self._g:
            for b in self.beliefs:
                b._eng = None
            self.lib.pgbp_patterns_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_schedule(self, schedule):
        L = self._L
        trees = [(np.asarray(t[-2], np.int32), np.asarray(t[-1], np.int32)) for t in schedule]
        off = np.zeros(len(trees) + 1, np.int32)
        for i, (pa, _) in enumerate(trees):
            off[i + 1] = off[i] + len(pa)
        pa = np.ascontiguousarray(np.concatenate([t[0] for t in trees]))
        ch = np.ascontiguousarray(np.concatenate([t[1] for t in trees]))
        self._check(self.lib.pgbp_patterns_set_schedule(self._g, len(trees), L.i32p(off), L.i32p(pa), L.i32p(ch)))

    def calibrate(self, niter=1, opts=None):
        """-> pgbp_result per site, in the caller's site order"""
        L = self._L
        res = (L.Result * self.n_sites)()
        o = opts if opts is not None else L.Opts(0, 1, 0, 0, 1e-5)
        self._check(self.lib.pgbp_patterns_calibrate(self._g, int(niter), self._C.byref(o), res))
        return res

    def loglik_lg(self, reps=1, opts=None):
        """device factor fill + postorder + root integrate of every site (score() body) -> (loglik, info) in site order"""
        L = self._L
        o = opts if opts is not None else L.Opts(0, 1, 0, 0, 1e-5)
        self._check(self.lib.pgbp_patterns_enqueue_loglik_lg(self._g, int(reps), self._C.byref(o)))
        norm, info = np.zeros(self.n_sites), np.zeros(self.n_sites, np.int32)
        self._check(self.lib.pgbp_patterns_fetch_loglik(self._g, L.f64p(norm), L.i32p(info)))
        return norm, info

    def integrate(self, belief):
        norm, info = np.zeros(self.n_sites), np.zeros(self.n_sites, np.int32)
        self._check(self.lib.pgbp_patterns_integrate(self._g, int(belief), self._L.f64p(norm), self._L.i32p(info)))
        return norm, info


class Comm:
    """pgbp_comm: one process per GPU; ONE ncclAllGather (RCCL) per gather_loglik call.
    `bcast(bytes_or_None) -> bytes` carries rank 0's message (a status byte + the unique id) to the other ranks and
    `allmin(int) -> int` the minimum of an integer over the ranks (e.g. torch.distributed / MPI collectives of the group
    that launched the ranks); both unused for n_ranks == 1.  Every rank takes part in BOTH collectives whatever happened
    to it locally and all ranks raise together: a rank that gave up alone would leave its peers inside a collective
    (the launcher's broadcast, or ncclCommInitRank, which blocks until every rank has arrived)."""

    ID_BYTES = 128

    def __init__(self, n_ranks, rank, device, bcast=None, allmin=None):
        import ctypes as C
        from . import _lib as L
        self._C, self._L = C, L
        self.lib = L.load()
        self.n_ranks, self.rank = int(n_ranks), int(rank)
        self._c = C.c_void_p()
        if self.n_ranks > 1 and (bcast is None or allmin is None):
            # (without them a rank that fails alone would leave its peers inside the broadcast or the collective create)
            raise ValueError("Comm: n_ranks > 1 needs both bcast and allmin of the group that launched the ranks")
        # 1. what can fail on this rank alone (RCCL not loadable, no such device), agreed on by all ranks
        code = self.lib.pgbp_comm_precheck(int(device))
        why = self.lib.pgbp_comm_last_error(None).decode() if code != 0 else ""
        if n_ranks > 1:
            if allmin(1 if code == 0 else 0) == 0:
                raise L.PgbpError(code or L.ERR_NO_DEVICE, why or "pgbp_comm: another rank cannot open its communicator")
        elif code != 0:
            raise L.PgbpError(code, why)
        # 2. rank 0's unique id: the broadcast always runs; a failure travels in the status byte
        ident = (C.c_uint8 * self.ID_BYTES)()
        status, msg = 0, ""
        if rank == 0:
            status = self.lib.pgbp_comm_unique_id(ident)
            if status != 0:
                msg = self.lib.pgbp_comm_last_error(None).decode()
                ident = (C.c_uint8 * self.ID_BYTES)()
        if n_ranks > 1:
            raw = bcast((bytes([min(255, abs(int(status)))]) + bytes(ident)) if rank == 0 else None)
            status = raw[0]
            ident = (C.c_uint8 * self.ID_BYTES).from_buffer_copy(raw[1:1 + self.ID_BYTES])
        if status != 0:
            raise L.PgbpError(int(status), msg or "pgbp_comm_unique_id failed on rank 0")
        # 3. the collective create, agreed on again: a rank whose ncclCommInitRank (or its allocations) failed must not leave
        # the others to walk into the first all-gather without it
        code = self._create(ident, int(device))
        why = self.lib.pgbp_comm_last_error(None).decode() if code != 0 else ""
        if n_ranks > 1 and allmin(1 if code == 0 else 0) == 0:
            self.close()
            raise L.PgbpError(code or L.ERR_HIP, why or "pgbp_comm: another rank could not create its communicator")
        if code != 0:
            raise L.PgbpError(code, why)

    def _create(self, ident, device):
        return self.lib.pgbp_comm_create(ident, self.n_ranks, self.rank, device, self._C.byref(self._c))

    def close(self):
        if self._c:
            self.lib.pgbp_comm_destroy(self._c)
            self._c = None

    def exchange_beliefs(self, engine, lists, site=0, include_self=False):
        """pgbp_comm_exchange_beliefs: lists[r] = the belief indices rank r contributes (the same lists on every rank);
        one ncclAllGather, device to device; afterwards this rank's engine holds the other ranks' records"""
        L = self._L
        off = np.zeros(self.n_ranks + 1, np.int32)
        for r in range(self.n_ranks):
            off[r + 1] = off[r] + len(lists[r])
        flat = np.ascontiguousarray(np.concatenate([np.asarray(x, np.int32) for x in lists]) if off[-1] else np.zeros(1, np.int32))
        code = self.lib.pgbp_comm_exchange_beliefs(self._c, engine, int(site), L.i32p(off), L.i32p(flat), int(include_self))
        if code != 0:
            raise L.PgbpError(code, self.lib.pgbp_comm_last_error(self._c).decode())

    def gather_loglik(self, engine, slot_sites):
        """-> (norm [n_ranks, slot_sites], info [n_ranks, slot_sites], all_succ, all_iscal) on every rank"""
        C, L = self._C, self._L
        norm = np.zeros((self.n_ranks, slot_sites))
        info = np.zeros((self.n_ranks, slot_sites), np.int32)
        succ, iscal = C.c_int32(), C.c_int32()
        code = self.lib.pgbp_comm_gather_loglik(self._c, engine, int(slot_sites), L.f64p(norm), L.i32p(info),
                                                C.byref(succ), C.byref(iscal))
        if code != 0:
            raise L.PgbpError(code, self.lib.pgbp_comm_last_error(self._c).decode())
        return norm, info, bool(succ.value), bool(iscal.value)


def unpack_slots(recv, n_ranks, slot_sites):
    """pgbp_comm_unpack_slots: the host half of pgbp_comm_gather_loglik on a gathered buffer
    -> (norm [n_ranks, slot_sites], info, all_succ, all_iscal)"""
    import ctypes as C
    from . import _lib as L
    lib = L.load()
    recv = np.ascontiguousarray(recv, dtype=np.float64)
    assert recv.size == n_ranks * (2 * slot_sites + 2)
    norm = np.zeros((n_ranks, slot_sites))
    info = np.zeros((n_ranks, slot_sites), np.int32)
    succ, iscal = C.c_int32(), C.c_int32()
    code = lib.pgbp_comm_unpack_slots(L.f64p(recv), int(n_ranks), int(slot_sites), L.f64p(norm), L.i32p(info),
                                      C.byref(succ), C.byref(iscal))
    if code != 0:
        raise L.PgbpError(code, "pgbp_comm_unpack_slots")
    return norm, info, bool(succ.value), bool(iscal.value)


# ---------------------------------------------------------------------------------------------------------------------
# A cluster graph CUT across devices (third bullet): the loop of
# src/calibration.jl:35-60 with each traversal (src/calibration.jl:111-161) cut by spanning-tree subtrees.

def cut_spanning_tree(pa, ch, n_ranks, min_subtrees=4):
    """Cut one spanning tree -- its preorder edge list (pa[i], ch[i]), as spanningtree_clusterlist returns it
    (src/clustergraph.jl:885-894) -- into a TOP (the clusters less than `depth` edges from the root, every rank runs it)
    and the SUBTREES hanging below it, dealt to the ranks largest first, each to the rank with the least edges so far.
    depth: the smallest that leaves at least min_subtrees * n_ranks subtrees with an edge of their own, none of them larger
    than half a rank's share (or the whole tree in the top when it is too small to cut).
    -> dict(top=(pa, ch) edge arrays of the top INCLUDING the boundary edges top -> subtree root,
            sub=[(rank, root cluster, (pa, ch) of the edges inside the subtree)], boundary=[edge index of each boundary edge])"""
    pa = np.asarray(pa, np.int64)
    ch = np.asarray(ch, np.int64)
    n = len(pa)
    if n == 0:
        return {"top": (pa.astype(np.int32), ch.astype(np.int32)), "sub": [], "boundary": [], "depth": 0}
    depth = {int(pa[0]): 0}
    for a, c in zip(pa, ch):
        depth[int(c)] = depth[int(a)] + 1
    dch = np.array([depth[int(c)] for c in ch])           # depth of the child end of edge i
    dmax = int(dch.max())
    # size of the subtree below each edge's child (edges strictly inside): children come after parents in preorder
    below = {int(c): 0 for c in ch}
    below[int(pa[0])] = 0
    for a, c in zip(pa[::-1], ch[::-1]):
        below[int(a)] = below.get(int(a), 0) + below[int(c)] + 1
    # the smallest depth with enough subtrees AND none larger than half a rank's share (largest-first dealing then balances
    # the ranks to about a quarter of a share); failing that, the best-balanced depth whose top stays below 5 % of the tree
    pick, best = None, None
    for d in range(1, dmax + 1):
        sizes = [below[int(ch[i])] for i in range(n) if dch[i] == d and below[int(ch[i])] > 0]
        if len(sizes) < min_subtrees * n_ranks:
            continue
        top_size = int((dch <= d).sum())
        if top_size > max(64, n // 20):
            break
        if max(sizes) <= (n - top_size) / (2.0 * n_ranks):
            pick = d
            break
        if best is None or max(sizes) < best[0]:
            best = (max(sizes), d)
    if pick is None and best is not None:
        pick = best[1]
    if pick is None or n_ranks <= 1:
        return {"top": (pa.astype(np.int32), ch.astype(np.int32)), "sub": [], "boundary": [], "depth": dmax + 1}
    top_edges = [i for i in range(n) if dch[i] <= pick]            # edges with the parent in the top (depth < pick)
    boundary = [i for i in range(n) if dch[i] == pick]
    # the subtree of each boundary edge's child: a contiguous run of the preorder list?  Not necessarily (the list is A
    # preorder, children of a node need not be adjacent), so collect by root
    root_of = {}
    for i in boundary:
        root_of[int(ch[i])] = int(ch[i])
    sub_edges = {int(ch[i]): [] for i in boundary}
    for i in range(n):
        if dch[i] > pick:
            r = root_of[int(pa[i])]
            root_of[int(ch[i])] = r
            sub_edges[r].append(i)
    load = [0] * n_ranks
    sub = []
    for r, edges in sorted(sub_edges.items(), key=lambda kv: (-len(kv[1]), kv[0])):
        k = int(np.argmin(load))
        load[k] += len(edges)
        sub.append((k, r, (pa[edges].astype(np.int32), ch[edges].astype(np.int32))))
    return {"top": (pa[top_edges].astype(np.int32), ch[top_edges].astype(np.int32)), "sub": sub, "boundary": boundary,
            "depth": pick, "load": load}


class NetworkCut:
    """calibrate!(beliefs, schedule, niter) (src/calibration.jl:35-60) of ONE cluster graph on several engines ("ranks":
    one per GPU; on a one-GPU box the same device several times), every traversal cut by cut_spanning_tree:
      1. postorder inside the subtrees a rank owns (propagate_1traversal_postorder!, src/calibration.jl:111-135);
      2. EXCHANGE A: the root cluster of every subtree (one record each: the boundary buffer) goes to every rank;
      3. every rank runs the top, postorder then preorder, boundary edges included -- the same arithmetic on the same
         operands, so the top, the boundary sepsets and the subtree roots stay bit-identical on all ranks;
      4. preorder inside the owned subtrees (src/calibration.jl:137-161);
      5. EXCHANGE B: what a rank's subtrees hold now (clusters, sepsets) goes to every rank -- the next spanning tree of a
         loopy graph is cut elsewhere.  (One spanning tree: nothing to do, the beliefs are read from their owners.)
    Messages that do not depend on each other run in another order than on one engine, nothing else changes: the beliefs
    are those of the single-engine run bit for bit.  The exchanges are pgbp_pack_beliefs / pgbp_unpack_beliefs through the
    host here (one process, engines side by side); between processes the packed buffer is what an all-gather carries.
    `beliefs`: one ClusterGraphBelief per rank over the same graph, same state.  A subtree is a tree of its rank's schedule
    of its own (the planner takes trees, not forests): its launches are narrower than the uncut traversal's.

    BETWEEN PROCESSES (one process per GPU, the driver's launch): `beliefs` = [this rank's ClusterGraphBelief], `rank`,
    `n_ranks`, and two callables of the group that launched the ranks -- `exchange(lists)`: lists[r] = the belief indices
    rank r contributes; afterwards this rank's engine holds the other ranks' records (CommExchange: ONE ncclAllGather behind
    the C ABI, pgbp_comm_exchange_beliefs, device to device; HostExchange: pgbp_pack_beliefs -> an all-gather of the
    launcher's group -> pgbp_unpack_beliefs, the one-GPU rehearsal) -- and `allmin(int)`: the minimum over the ranks (the
    AND of the ranks' calibration flags, and of their traversals' success).  Every rank computes the same cut."""

    def __init__(self, beliefs, schedule, min_subtrees=4, rank=None, n_ranks=None, exchange=None, allmin=None):
        self.rank = rank
        if rank is None:
            self.ranks = dict(enumerate(beliefs))
            self.K = len(self.ranks)
        else:
            assert len(beliefs) == 1 and n_ranks is not None and exchange is not None and allmin is not None
            self.ranks = {int(rank): beliefs[0]}
            self.K = int(n_ranks)
        self._exchange_fn, self._allmin = exchange, allmin
        self.schedule = [(np.asarray(t[-2], np.int32), np.asarray(t[-1], np.int32)) for t in schedule]
        self.cuts = [cut_spanning_tree(pa, ch, self.K, min_subtrees) for pa, ch in self.schedule]
        b0 = next(iter(self.ranks.values()))
        sep_of = {}
        for k, (a, c) in enumerate(b0._sepcl):
            sep_of[(min(int(a), int(c)), max(int(a), int(c)))] = b0.nclusters + k
        self._sep_of = sep_of
        # per rank: its schedule = for every spanning tree, the top, then its subtrees (those with an edge)
        self.tree_index = []        # [t] -> dict(top={rank: idx}, sub={rank: [idx, ...]}): indices into the rank's own schedule
        per_rank = [[] for _ in range(self.K)]
        for cut in self.cuts:
            entry = {"sub": {r: [] for r in range(self.K)}, "top": {}}
            for r in range(self.K):
                entry["top"][r] = len(per_rank[r])
                per_rank[r].append(cut["top"])
            for r in range(self.K):
                for (k, _root, edges) in cut["sub"]:
                    if k == r and len(edges[0]) > 0:   # (a boundary child that is a leaf has no edge of its own: nothing to
                        entry["sub"][r].append(len(per_rank[r]))   # traverse, only its root to exchange)
                        per_rank[r].append(edges)
            self.tree_index.append(entry)
        for r, b in self.ranks.items():
            b.set_schedule(per_rank[r])
        # exchange lists
        self.roots = []        # [t][rank] -> belief indices of the roots of its subtrees
        self.owned = []        # [t][rank] -> belief indices of everything inside its subtrees (clusters incl. roots, sepsets)
        for cut in self.cuts:
            roots = [[] for _ in range(self.K)]
            owned = [[] for _ in range(self.K)]
            for (k, root, (pa, ch)) in cut["sub"]:
                roots[k].append(root)
                owned[k].append(root)
                for a, c in zip(pa, ch):
                    owned[k].append(int(c))
                    owned[k].append(sep_of[(min(int(a), int(c)), max(int(a), int(c)))])
            self.roots.append([np.asarray(x, np.int32) for x in roots])
            self.owned.append([np.asarray(x, np.int32) for x in owned])
        self.last_writer = {}      # (receiver, sender) -> rank that sent the message last
        self.exchanged_doubles = 0

    # -- the exchange: `idx[r]` from rank r to every other rank
    def _exchange(self, idx):
        import ctypes as C
        from . import _lib as L
        if self._exchange_fn is not None:      # between processes: one all-gather
            self._exchange_fn(idx)
            b = self.ranks[self.rank]
            self.exchanged_doubles += sum(int(b._poff[i + 1] - b._poff[i]) for x in idx for i in x)
            return
        for r, b in self.ranks.items():
            if len(idx[r]) == 0:
                continue
            lst = np.ascontiguousarray(idx[r], np.int32)
            n = int(b._lib.pgbp_packed_beliefs_size(b._eng, len(lst), L.i32p(lst)))
            buf = np.zeros(n)
            code = b._lib.pgbp_pack_beliefs(b._eng, 0, len(lst), L.i32p(lst), L.f64p(buf))
            if code != 0:
                raise L.PgbpError(code, b._lib.pgbp_last_error(b._eng).decode())
            self.exchanged_doubles += n
            for q, o in self.ranks.items():
                if q == r:
                    continue
                code = o._lib.pgbp_unpack_beliefs(o._eng, 0, len(lst), L.i32p(lst), L.f64p(buf))
                if code != 0:
                    raise L.PgbpError(code, o._lib.pgbp_last_error(o._eng).decode())

    def _traverse(self, r, tree, direction, opts):
        import ctypes as C
        from . import _lib as L
        b = self.ranks[r]
        res = (L.Result * b.n_sites)()
        code = b._lib.pgbp_traverse(b._eng, int(tree), int(direction), C.byref(opts), res)
        if code != 0:
            raise L.PgbpError(code, b._lib.pgbp_last_error(b._eng).decode())
        return bool(res[0].succ)

    def _note(self, r, edges, direction):
        pa, ch = edges
        for a, c in zip(pa, ch):
            key = (int(a), int(c)) if direction == 0 else (int(c), int(a))   # (receiver, sender)
            self.last_writer[key] = r

    def traversal(self, t, opts=None):
        """one spanning tree, postorder then preorder, cut -> True if every message went through"""
        from . import _lib as L
        o = opts if opts is not None else L.Opts(0, 1, 0, 0, 1e-5)
        cut, ix = self.cuts[t], self.tree_index[t]
        ok = True
        subs = {r: [e for (k, _root, e) in cut["sub"] if k == r and len(e[0]) > 0] for r in range(self.K)}
        mine = sorted(self.ranks)                                 # (one process: every rank; between processes: this one)
        for r in range(self.K):                                   # 1
            for j, tree in enumerate(ix["sub"][r]):
                if r in self.ranks:
                    ok &= self._traverse(r, tree, 0, o)
                self._note(r, subs[r][j], 0)
        self._exchange(self.roots[t])                             # 2
        for r in mine:                                            # 3 (every rank; rank 0 speaks for the top's flags)
            ok &= self._traverse(r, ix["top"][r], 0, o)
            ok &= self._traverse(r, ix["top"][r], 1, o)
        self._note(0, cut["top"], 0)
        self._note(0, cut["top"], 1)
        for r in range(self.K):                                   # 4
            for j, tree in enumerate(ix["sub"][r]):
                if r in self.ranks:
                    ok &= self._traverse(r, tree, 1, o)
                self._note(r, subs[r][j], 1)
        if len(self.schedule) > 1:                                # 5
            self._exchange(self.owned[t])
        if self._allmin is not None:                              # a message that failed on any rank fails the traversal on all
            ok = bool(self._allmin(1 if ok else 0))
        return ok

    def iscalibrated_residnorm(self):
        """iscalibrated_residnorm over every message residual (src/beliefs.jl:994-1003), each read from the rank that
        sent the message last; a message never sent is not calibrated"""
        from . import _lib as L
        b0 = next(iter(self.ranks.values()))
        nm = 2 * b0.nsepsets
        flags = {}
        for r, b in self.ranks.items():
            f = np.zeros(b.n_sites * nm, np.int32)
            code = b._lib.pgbp_get_residuals(b._eng, None, L.i32p(f), None, None)
            if code != 0:
                raise L.PgbpError(code, b._lib.pgbp_last_error(b._eng).decode())
            flags[r] = f[:nm]
        ok = True
        for k, (a, c) in enumerate(b0._sepcl):
            for mid, key in ((2 * k, (int(a), int(c))), (2 * k + 1, (int(c), int(a)))):   # (receiver, sender)
                r = self.last_writer.get(key)
                if r is None or (r in flags and not flags[r][mid]):   # (between processes: each rank answers for what it sent)
                    ok = False
        if self._allmin is not None:
            ok = bool(self._allmin(1 if ok else 0))
        return ok

    def calibrate(self, niter=1, auto=False, opts=None):
        """-> (succ, iscal, (iteration, tree) reached if auto stopped there else None)"""
        for it in range(niter):
            for t in range(len(self.schedule)):
                if not self.traversal(t, opts):
                    return False, False, None
                if auto and self.iscalibrated_residnorm():
                    return True, True, (it + 1, t + 1)
        return True, self.iscalibrated_residnorm(), None

    def gather(self):
        """the calibrated beliefs [packed_size], every record from the rank that owns it after the last traversal (one
        spanning tree: exchange B was skipped; several: any rank holds everything)"""
        if self.rank is not None and len(self.schedule) == 1:
            self._exchange(self.owned[0])      # between processes: what the ranks own goes to every rank once, at the end
        b0 = self.ranks[min(self.ranks)]
        from . import _lib as L
        out = np.zeros(int(b0._lib.pgbp_packed_size(b0._eng)))
        full = np.zeros((1, len(out)))
        code = b0._lib.pgbp_get_beliefs(b0._eng, L.f64p(full))
        if code != 0:
            raise L.PgbpError(code, b0._lib.pgbp_last_error(b0._eng).decode())
        out[:] = full[0]
        if len(self.schedule) == 1 and self.rank is None:
            for r in range(1, self.K):
                lst = self.owned[0][r]
                if len(lst) == 0:
                    continue
                b = self.ranks[r]
                n = int(b._lib.pgbp_packed_beliefs_size(b._eng, len(lst), L.i32p(lst)))
                buf = np.zeros(n)
                code = b._lib.pgbp_pack_beliefs(b._eng, 0, len(lst), L.i32p(lst), L.f64p(buf))
                if code != 0:
                    raise L.PgbpError(code, b._lib.pgbp_last_error(b._eng).decode())
                at = 0
                for i in lst:
                    ln = int(b0._poff[i + 1] - b0._poff[i])
                    out[b0._poff[i]:b0._poff[i] + ln] = buf[at:at + ln]
                    at += ln
        return out


class CommExchange:
    """NetworkCut's exchange between processes through pgbp_comm: ONE ncclAllGather per exchange, device to device."""

    def __init__(self, comm, belief):
        self.comm, self.b = comm, belief

    def __call__(self, lists):
        self.comm.exchange_beliefs(self.b._eng, lists)


class HostExchange:
    """The same through the launcher's own group (torch.distributed, e.g. gloo): pgbp_pack_beliefs -> all_gather of the padded
    host buffers -> pgbp_unpack_beliefs.  What the one-GPU rehearsal of the cut uses (two RCCL ranks cannot share a device)."""

    def __init__(self, dist, belief, rank, n_ranks):
        self.dist, self.b, self.rank, self.K = dist, belief, int(rank), int(n_ranks)

    def __call__(self, lists):
        import torch
        from . import _lib as L
        b = self.b
        size = [int(b._lib.pgbp_packed_beliefs_size(b._eng, len(x), L.i32p(np.ascontiguousarray(x, np.int32)))) if len(x) else 0
                for x in lists]
        slot = max(size)
        if slot == 0:
            return
        mine = np.ascontiguousarray(lists[self.rank], np.int32)
        buf = np.zeros(slot)
        if len(mine):
            code = b._lib.pgbp_pack_beliefs(b._eng, 0, len(mine), L.i32p(mine), L.f64p(buf))
            if code != 0:
                raise L.PgbpError(code, b._lib.pgbp_last_error(b._eng).decode())
        got = [torch.zeros(slot, dtype=torch.float64) for _ in range(self.K)]
        self.dist.all_gather(got, torch.from_numpy(buf))
        for r in range(self.K):
            if r == self.rank or size[r] == 0:
                continue
            lst = np.ascontiguousarray(lists[r], np.int32)
            code = b._lib.pgbp_unpack_beliefs(b._eng, 0, len(lst), L.i32p(lst), L.f64p(np.ascontiguousarray(got[r].numpy())))
            if code != 0:
                raise L.PgbpError(code, b._lib.pgbp_last_error(b._eng).decode())
