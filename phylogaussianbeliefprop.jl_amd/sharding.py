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
        # 1. what can fail on this rank alone (RCCL not loadable, no such device), agreed on by all ranks
        code = self.lib.pgbp_comm_precheck(int(device))
        why = self.lib.pgbp_comm_last_error(None).decode() if code != 0 else ""
        if n_ranks > 1 and allmin is not None:
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
        # 3. the collective create
        code = self.lib.pgbp_comm_create(ident, self.n_ranks, self.rank, int(device), C.byref(self._c))
        if code != 0:
            raise L.PgbpError(code, self.lib.pgbp_comm_last_error(None).decode())

    def close(self):
        if self._c:
            self.lib.pgbp_comm_destroy(self._c)
            self._c = None

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
