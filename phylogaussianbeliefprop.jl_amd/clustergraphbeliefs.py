"""ClusterGraphBelief (src/clustergraphbeliefs.jl:26-109) backed by the device engine."""
import ctypes as C

import numpy as np

from . import _lib as L
from .beliefs import CanonicalBelief, MessageResidual, bclustertype, bsepsettype, scopeindex
from .beliefupdates import BPPosDefException


def _check(code, eng=None, lib=None):
    if code != L.PGBP_OK:
        lib = lib or L.load()
        msg = lib.pgbp_last_error(eng).decode() if lib else "?"
        raise L.PgbpError(code, msg)


class _BeliefList:
    """belief vector: CanonicalBelief objects whose h/J/g are views of the packed host mirror."""

    def __init__(self, owner):
        self._o = owner

    def __len__(self):
        return self._o.nbeliefs

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        return self._o._belief_view(i)

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class ClusterGraphBelief:
    """Device-resident ClusterGraphBelief.

    ClusterGraphBelief(beliefs, node2cluster, node2family, node2fixed, cluster2nodes)
    mirrors src/clustergraphbeliefs.jl:89-109: clusters first, sepsets last; builds cdict,
    sdict, the message residuals and the factors (copies of the initial cluster beliefs),
    precomputes scopeindex(sepset, cluster) for both ends of every sepset, and uploads.
    `ClusterGraphBelief.from_arrays` is the bulk constructor for large synthetic graphs."""

    def __init__(self, beliefs, node2cluster=None, node2family=None, node2fixed=None, cluster2nodes=None,
                 device=0):
        types = [b.type for b in beliefs]
        nc = types.index(bsepsettype) if bsepsettype in types else len(beliefs)
        if not all(t == bclustertype for t in types[:nc]):
            raise ValueError("clusters are not consecutive")
        if not all(t == bsepsettype for t in types[nc:]):
            raise ValueError("sepsets are not consecutive")
        cdict = {beliefs[j].metadata: j for j in range(nc)}
        dims = np.array([b.dimension for b in beliefs], dtype=np.int32)
        sepcl, off, idx = [], [0], []
        for j in range(nc, len(beliefs)):
            l1, l2 = beliefs[j].metadata
            a, b = cdict[l1], cdict[l2]
            sepcl += [a, b]
            for c in (a, b):
                ind = scopeindex(beliefs[j], beliefs[c])
                idx.append(ind)
                off.append(off[-1] + len(ind))
        idx = np.concatenate(idx) if idx else np.zeros(0, np.int32)
        self._init_common(dims, np.array(sepcl, np.int32), np.array(off, np.int64), idx, 1, device)
        self._objs = list(beliefs)
        self.cdict = cdict
        self.sdict = {frozenset(beliefs[j].metadata): j for j in range(nc, len(beliefs))}
        self.node2cluster, self.node2family = node2cluster, node2family
        self.node2fixed, self.cluster2nodes = node2fixed, cluster2nodes
        # copy h,J,g into the packed mirror and re-bind the objects' arrays to views of it
        for i, b in enumerate(self._objs):
            J, h, g = self._views(0, i)
            J[...] = b.J
            h[...] = b.h
            g[...] = b.g
            b.J, b.h, b.g = J, h, g
            b._owner, b._index = self, i
        self._upload(snapshot_factors=True)

    @classmethod
    def from_arrays(cls, dims, sepset_clusters, scope_off, scope_idx, packed, n_sites=1, device=0,
                    labels=None, engine=None):
        """Bulk constructor: description arrays of include/pgbp.h + packed (J,h,g) beliefs
        [n_sites, packed_size]; the cluster part is snapshot as the factors.  packed=None: nothing is uploaded and no
        host mirror is kept (large site batches whose factors are assigned on the device)."""
        self = cls.__new__(cls)
        self._init_common(np.asarray(dims, np.int32), np.asarray(sepset_clusters, np.int32).reshape(-1),
                          np.asarray(scope_off, np.int64), np.asarray(scope_idx, np.int32), n_sites, device, engine=engine)
        self._objs = None
        self._labels = labels
        self.cdict = self.sdict = None
        if packed is not None:   # None: all beliefs start as the constant 1 on the device (a device factor fill follows);
            self._packed[...] = np.asarray(packed, dtype=np.float64).reshape(self._packed.shape)   # no host mirror until a pull
            self._upload(snapshot_factors=True)
        return self

    # ------------------------------------------------------------------ internals
    def _init_common(self, dims, sepcl, scope_off, scope_idx, n_sites, device, engine=None):
        """engine: an engine created elsewhere for the same description (PatternGroup: pgbp_patterns_engine); it is
        borrowed, not destroyed with this object."""
        self._lib = L.load()
        self._eng = None
        self._borrowed = engine is not None
        self.n_sites = int(n_sites)
        self._dims = dims
        self.nsepsets = (len(scope_off) - 1) // 2
        self.nclusters = len(dims) - self.nsepsets
        self.nbeliefs = len(dims)
        self._sepcl = sepcl.reshape(-1, 2) if sepcl.size else np.zeros((0, 2), np.int32)
        self._scope_off, self._scope_idx = scope_off, scope_idx
        desc, self._keep = L.make_desc(dims, sepcl, scope_off, scope_idx, n_sites, device)
        eng = C.c_void_p()
        if engine is not None:
            eng = engine if isinstance(engine, C.c_void_p) else C.c_void_p(engine)
        else:
            code = self._lib.pgbp_create(C.byref(desc), C.byref(eng))
            if code != L.PGBP_OK:
                raise L.PgbpError(code, self._lib.pgbp_last_error(None).decode())
        self._eng = eng
        m = dims.astype(np.int64)
        self._poff = np.concatenate([[0], np.cumsum(m * m + m + 1)])
        s = np.repeat(dims[self.nclusters:].astype(np.int64), 2)
        self._roff = np.concatenate([[0], np.cumsum(s * s + s)])
        assert self._poff[-1] == self._lib.pgbp_packed_size(eng)
        assert self._roff[-1] == self._lib.pgbp_residual_size(eng)
        self._packed_arr = None   # host mirror [n_sites, packed_size], allocated at first use
        self._res = None
        self._flg = None
        self._kl = None
        self._klflg = None
        # LAZY write-back (one site): after a device call the mirror is only marked stale; a belief's record is fetched at
        # its first read (pgbp_get_belief), a residual's at its first read (pgbp_get_residual), the flag vectors on demand.
        # _stale: None = the mirror is current, else one bool per belief; _res_have: residual records fetched since the
        # last device call.  Engines with several sites keep the eager pull.
        self._stale = None
        self._n_single = 0
        self._res_have = None
        self.lazy = True
        self._schedule = None
        self.site = 0  # which site the belief views / residual views show
        self.belief = _BeliefList(self)
        self.messageresidual = _ResidualDict(self)
        self.last_results = None

    @property
    def _packed_raw(self):
        if self._packed_arr is None:
            self._packed_arr = np.zeros((self.n_sites, int(self._poff[-1])))
        return self._packed_arr

    @property
    def _packed(self):
        """the host mirror as a whole: made current first (what is still stale is fetched in one transfer)"""
        self._refresh_all()
        return self._packed_raw

    def _invalidate(self):
        """the device state changed: drop the host copies.  One site: lazily (nothing moves until something is read);
        several sites: the eager pull."""
        if self.n_sites == 1 and self.lazy:
            self._stale = np.ones(self.nbeliefs, dtype=bool)
            self._n_single = 0
            self._res = self._flg = self._kl = self._klflg = None
            self._res_have = None
        else:
            self.pull()

    def _refresh(self, i):
        if self._stale is None or not self._stale[i]:
            return
        self._n_single += 1
        if self._n_single > 64:   # a caller walking over the beliefs: one transfer of what is left beats thousands of small ones
            self._refresh_all()
            return
        n = int(self._poff[i + 1] - self._poff[i])
        rec = np.zeros(max(1, n))
        _check(self._lib.pgbp_get_belief(self._eng, 0, int(i), L.f64p(rec)), self._eng)
        self._packed_raw[0, self._poff[i]: self._poff[i + 1]] = rec[:n]
        self._stale[i] = False

    def _refresh_all(self):
        if self._stale is None:
            return
        idx = np.nonzero(self._stale)[0].astype(np.int32)
        if len(idx) == self.nbeliefs:
            _check(self._lib.pgbp_get_beliefs(self._eng, L.f64p(self._packed_raw)), self._eng)
        elif len(idx):
            # the stale records only, gathered on the device into one buffer (records edited on the host stay as they are)
            n = int(self._lib.pgbp_packed_beliefs_size(self._eng, len(idx), L.i32p(idx)))
            buf = np.zeros(max(1, n))
            _check(self._lib.pgbp_pack_beliefs(self._eng, 0, len(idx), L.i32p(idx), L.f64p(buf)), self._eng)
            at = 0
            for i in idx:
                m = int(self._poff[i + 1] - self._poff[i])
                self._packed_raw[0, self._poff[i]: self._poff[i + 1]] = buf[at: at + m]
                at += m
        self._stale = None

    def __del__(self):
        try:
            if getattr(self, "_eng", None):
                if not getattr(self, "_borrowed", False):
                    self._lib.pgbp_destroy(self._eng)
                self._eng = None
        except Exception:
            pass

    def _views(self, site, i):
        m = int(self._dims[i])
        self._refresh(i)
        rec = self._packed_raw[site, self._poff[i]: self._poff[i + 1]]
        return rec[: m * m].reshape(m, m, order="F"), rec[m * m: m * m + m], rec[m * m + m:]

    def _belief_view(self, i):
        if self._objs is not None and self.site == 0:
            return self._objs[i]
        J, h, g = self._views(self.site, i)
        site_is_lazy = self.n_sites == 1
        b = CanonicalBelief.__new__(CanonicalBelief)
        b.nodelabel, b.ntraits, b.inscope = None, None, None
        b.J, b.h, b.g, b.mu = J, h, g, np.zeros(len(h))
        if site_is_lazy:
            b._owner, b._index = self, i
        b.type = bclustertype if i < self.nclusters else bsepsettype
        b.metadata = self._labels[i] if getattr(self, "_labels", None) is not None else i
        return b

    def _upload(self, snapshot_factors=False):
        # (self._packed: whatever of the mirror is stale is fetched first, so an upload never writes old values back)
        _check(self._lib.pgbp_set_beliefs(self._eng, L.f64p(self._packed), int(snapshot_factors)), self._eng)

    def push(self):
        """host mirror -> device (after editing belief arrays on the host)."""
        self._upload(False)

    def pull(self):
        """device -> host mirror, everything at once: beliefs, residuals, flags (the eager write-back)."""
        self._stale = None
        _check(self._lib.pgbp_get_beliefs(self._eng, L.f64p(self._packed_raw)), self._eng)
        self._res_have = None
        nm = 2 * self.nsepsets
        self._res = np.zeros((self.n_sites, max(1, int(self._roff[-1]))))
        self._flg = np.zeros((self.n_sites, max(1, nm)), dtype=np.int32)
        self._kl = np.zeros((self.n_sites, max(1, nm)))
        self._klflg = np.zeros((self.n_sites, max(1, nm)), dtype=np.int32)
        _check(self._lib.pgbp_get_residuals(self._eng, L.f64p(self._res), L.i32p(self._flg), L.f64p(self._kl),
                                            L.i32p(self._klflg)), self._eng)

    def _fetch_words(self):
        """flags, kldiv, iscalibrated_kl of every message (no residual record moves)"""
        nm = 2 * self.nsepsets
        self._flg = np.zeros((self.n_sites, max(1, nm)), dtype=np.int32)
        self._kl = np.zeros((self.n_sites, max(1, nm)))
        self._klflg = np.zeros((self.n_sites, max(1, nm)), dtype=np.int32)
        _check(self._lib.pgbp_get_residuals(self._eng, None, L.i32p(self._flg), L.f64p(self._kl), L.i32p(self._klflg)), self._eng)

    def _residual_record(self, d):
        if self._res is None and not (self.n_sites == 1 and self.lazy):
            self.pull()
        if self._res is None:   # lazy: one record at a time
            self._res = np.zeros((1, max(1, int(self._roff[-1]))))
            self._res_have = np.zeros(2 * self.nsepsets, dtype=bool)
        if self._res_have is not None and not self._res_have[d] and int(self._res_have.sum()) >= 64:
            # a caller walking over the residuals: the rest in one transfer
            _check(self._lib.pgbp_get_residuals(self._eng, L.f64p(self._res), None, None, None), self._eng)
            self._res_have = None
        if self._res_have is not None and not self._res_have[d]:
            n = int(self._roff[d + 1] - self._roff[d])
            rec = np.zeros(max(1, n))
            _check(self._lib.pgbp_get_residual(self._eng, 0, int(d), L.f64p(rec), None, None, None), self._eng)
            self._res[0, self._roff[d]: self._roff[d + 1]] = rec[:n]
            self._res_have[d] = True
        return self._res[self.site, self._roff[d]: self._roff[d + 1]]

    def _residual_words(self, d):
        """(iscalibrated_resid, kldiv, iscalibrated_kl) of message d"""
        if self._flg is None:
            self._fetch_words()
        return self._flg[self.site][d], self._kl[self.site][d], self._klflg[self.site][d]

    def _flags(self):
        if self._flg is None:
            self._fetch_words()
        return self._flg[self.site]

    def _kldiv(self):
        if self._kl is None:
            self._fetch_words()
        return self._kl[self.site]

    def _klflags(self):
        if getattr(self, "_klflg", None) is None:
            self._fetch_words()
        return self._klflg[self.site]

    def residual_kldiv_(self, cluster_to, sepset, cluster_from, atol=1e-5):
        """residual_kldiv!(messageresidual[(to, from)], sepset) (src/beliefs.jl:1060-1075) on belief indices:
        updates kldiv / iscalibrated_kl of that residual on the device and returns the flag."""
        out = np.zeros(self.n_sites, dtype=np.int32)
        o = self._opts(atol=atol)
        _check(self._lib.pgbp_residual_kldiv(self._eng, int(cluster_to), int(sepset), int(cluster_from), C.byref(o),
                                             L.i32p(out)), self._eng)
        self._flg = self._kl = self._klflg = None   # (the beliefs and the residual records are untouched)
        if int(self._dims[sepset]) == 0:
            return True
        return bool(out[self.site])

    def _msg_id(self, receiver, sender):
        for k in range(self.nsepsets):
            a, b = self._sepcl[k]
            if a == receiver and b == sender:
                return 2 * k
            if b == receiver and a == sender:
                return 2 * k + 1
        raise KeyError((receiver, sender))

    def _opts(self, auto=False, update_residualnorm=True, update_residualkldiv=False, atol=1e-5):
        return L.Opts(int(auto), int(update_residualnorm), int(update_residualkldiv), 0, float(atol))

    def _integrate_index_1based(self, sender, sepset_k, side):
        o0, o1 = self._scope_off[2 * sepset_k + side], self._scope_off[2 * sepset_k + side + 1]
        keep = set(self._scope_idx[o0:o1].tolist())
        return [i + 1 for i in range(int(self._dims[sender])) if i not in keep]

    def _exception_for(self, sender, sepset_k, info):
        """"belief $metadata, integrating $(integrate_index)" (src/beliefupdates.jl:71)."""
        side = 0 if self._sepcl[sepset_k][0] == sender else 1
        meta = self._belief_label(sender)
        idx = self._integrate_index_1based(sender, sepset_k, side)
        return BPPosDefException(f"belief {meta}, integrating {idx}", info)

    def _belief_label(self, i):
        if self._objs is not None:
            return self._objs[i].metadata
        if getattr(self, "_labels", None) is not None:
            return self._labels[i]
        return i

    def _propagate(self, cluster_to, sepset, cluster_from, sync=True):
        info = np.zeros(self.n_sites, dtype=np.int32)
        o = self._opts()
        _check(self._lib.pgbp_propagate(self._eng, int(cluster_to), int(sepset), int(cluster_from), C.byref(o),
                                        L.i32p(info)), self._eng)
        if sync:
            self._invalidate()
        if info[self.site] != 0:
            return self._exception_for(cluster_from, sepset - self.nclusters, int(info[self.site]))
        return None

    # ------------------------------------------------------------------ reference API
    def clusterindex(self, label):
        return self.cdict[label]

    def sepsetindex(self, l1, l2):
        return self.sdict[frozenset((l1, l2))]

    def set_schedule(self, schedule):
        """schedule: list of spanning trees (pa_lab, ch_lab, pa_j, ch_j) as spanningtree_clusterlist
        returns them (src/clustergraph.jl:885-894), or just (pa_j, ch_j); 0-based cluster indices."""
        trees = [(np.asarray(t[-2], np.int32), np.asarray(t[-1], np.int32)) for t in schedule]
        off = np.zeros(len(trees) + 1, dtype=np.int32)
        for i, (pa, _) in enumerate(trees):
            off[i + 1] = off[i] + len(pa)
        pa = np.concatenate([t[0] for t in trees]) if trees else np.zeros(0, np.int32)
        ch = np.concatenate([t[1] for t in trees]) if trees else np.zeros(0, np.int32)
        pa = np.ascontiguousarray(pa if pa.size else np.zeros(1, np.int32))
        ch = np.ascontiguousarray(ch if ch.size else np.zeros(1, np.int32))
        _check(self._lib.pgbp_set_schedule(self._eng, len(trees), L.i32p(off), L.i32p(pa), L.i32p(ch)), self._eng)
        self._schedule = [(t[0].copy(), t[1].copy()) for t in trees]

    def _ensure_schedule(self, schedule):
        trees = [(np.asarray(t[-2], np.int32), np.asarray(t[-1], np.int32)) for t in schedule]
        same = self._schedule is not None and len(trees) == len(self._schedule) and all(
            np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(trees, self._schedule))
        if not same:
            self.set_schedule(schedule)

    def init_beliefs_reset_fromfactors_(self, sync=True):
        """init_beliefs_reset_fromfactors! (src/clustergraphbeliefs.jl:126-139)."""
        _check(self._lib.pgbp_reset_from_factors(self._eng), self._eng)
        if sync:
            self._invalidate()

    def init_factors_frombeliefs_(self):
        """init_factors_frombeliefs! (src/beliefs.jl:746-761) on the device state."""
        _check(self._lib.pgbp_init_factors_frombeliefs(self._eng), self._eng)

    def init_messagecalibrationflags_reset_(self, reset_kl=True):
        """init_messagecalibrationflags_reset! (src/clustergraphbeliefs.jl:146-150)."""
        _check(self._lib.pgbp_reset_flags(self._eng, int(reset_kl)), self._eng)
        self._flg = self._kl = self._klflg = None

    def iscalibrated_residnorm(self):
        """iscalibrated_residnorm(beliefs) (src/clustergraphbeliefs.jl:168-169)."""
        self._fetch_words()   # (the flag vectors alone: no belief, no residual record moves)
        return bool(np.all(self._flg[self.site][: 2 * self.nsepsets] != 0))

    def integratebelief_(self, j, all_sites=False):
        """integratebelief!(obj, beliefindex) (src/clustergraphbeliefs.jl:194): (mu, norm)."""
        m = int(self._dims[j])
        mu = np.zeros((self.n_sites, max(1, m)))
        norm = np.zeros(self.n_sites)
        info = np.zeros(self.n_sites, dtype=np.int32)
        _check(self._lib.pgbp_integrate(self._eng, int(j), L.f64p(mu), L.f64p(norm), L.i32p(info)), self._eng)
        mu = mu.reshape(-1)[: self.n_sites * m].reshape(self.n_sites, m)
        if all_sites:
            return mu, norm, info
        if info[self.site] != 0:
            raise np.linalg.LinAlgError(
                f"PosDefException: matrix is not positive definite; Cholesky factorization failed (info={info[self.site]}).")
        if self._objs is not None and self.site == 0:
            self._objs[j].mu = mu[0].copy()
        return mu[self.site].copy(), float(norm[self.site])

    def default_sepset1(self):
        """default_sepset1 (src/clustergraphbeliefs.jl:197-202)."""
        for j in range(self.nclusters, self.nbeliefs):
            if len(self._objs[j].nodelabel) == 1:
                return j
        raise ValueError("no sepset with a single node")

    # ------------------------------------------------------------------ scores (src/score.jl)
    def free_energy(self, all_sites=False):
        """free_energy(beliefs) (src/score.jl:162-182): (average energy, approximate entropy, free energy)."""
        out = np.zeros((self.n_sites, 3))
        info = np.zeros(self.n_sites, dtype=np.int32)
        _check(self._lib.pgbp_free_energy(self._eng, L.f64p(out), L.i32p(info)), self._eng)
        if all_sites:
            return out, info
        if info[self.site]:
            raise np.linalg.LinAlgError(f"PosDefException: belief {info[self.site] - 1} is not positive definite")
        return tuple(float(x) for x in out[self.site])

    def factored_energy(self):
        """factored_energy(beliefs) (src/score.jl:151-154): third value = -free energy."""
        a, e, f = self.free_energy()
        return (a, e, -f)

    # ------------------------------------------------------------------ device factor assignment
    def bm_tree_setup(self, kind, length, data_row, data):
        """Static part of assignfactors! for a homogeneous BM on a tree (include/pgbp.h: pgbp_bm_tree);
        data: [n_sites, n_rows, p] (or [n_rows, p] for one site) tip data indexed by data_row."""
        data = np.ascontiguousarray(np.asarray(data, np.float64))
        if data.ndim == 2:
            data = data[None]
        assert data.shape[0] == self.n_sites
        self._bm = dict(kind=np.ascontiguousarray(kind, np.int32), length=np.ascontiguousarray(length, np.float64),
                        row=np.ascontiguousarray(data_row, np.int32), data=data)
        t = L.BmTree(int(data.shape[2]), int(data.shape[1]), L.i32p(self._bm["kind"]), L.f64p(self._bm["length"]),
                     L.i32p(self._bm["row"]), L.f64p(data))
        _check(self._lib.pgbp_bm_tree_setup(self._eng, C.byref(t)), self._eng)
        self._bm_p = int(data.shape[2])

    def assignfactors_bm_(self, R, mu, sync=False):
        """assignfactors!(beliefs, MvFullBrownianMotion(R, mu), ...) (src/beliefs.jl:786-861) on the device:
        only R^-1, log det R and mu cross the bus.  R: [p, p] or [n_sites, p, p]; mu: [p] or [n_sites, p]."""
        R = np.asarray(R, np.float64)
        mu = np.asarray(mu, np.float64)
        per_site = R.ndim == 3
        Rs = R if per_site else R[None]
        Rinv = np.ascontiguousarray(np.stack([np.linalg.inv((r + r.T) / 2) for r in Rs]))
        Rinv = np.ascontiguousarray((Rinv + np.transpose(Rinv, (0, 2, 1))) / 2)
        logdet = np.ascontiguousarray([np.linalg.slogdet(r)[1] for r in Rs], np.float64)
        mus = np.ascontiguousarray(mu if mu.ndim == 2 else np.broadcast_to(mu, (len(Rs), self._bm_p)).copy())
        _check(self._lib.pgbp_bm_tree_assignfactors(self._eng, L.f64p(Rinv), L.f64p(logdet), L.f64p(mus),
                                                    int(per_site)), self._eng)
        if sync:
            self._invalidate()

    def lg_setup(self, fam, data):
        """Static part of assignfactors! for any linear-Gaussian model (include/pgbp.h: pgbp_lg_families).
        fam: the dictionary factors.lg_families(...) returns; data: [n_sites, n_rows, p] (or [n_rows, p]) tip data."""
        data = np.ascontiguousarray(np.asarray(data, np.float64))
        if data.ndim == 2:
            data = data[None]
        assert data.shape[0] == self.n_sites and data.shape[2] == fam["p"]
        if fam.get("child_mask") is not None:
            assert fam["parent_mask"].size == len(fam["cluster"]) * max(1, int(fam["max_parents"]))
        k = {n: np.ascontiguousarray(fam[n], np.int32) for n in ("cluster", "n_parents", "child_pos", "data_row",
                                                                 "parent_pos", "color")}
        k.update({n: np.ascontiguousarray(fam[n], np.float64) for n in ("length", "gamma")})
        k["data"] = data
        for n in ("child_mask", "parent_mask"):
            k[n] = np.ascontiguousarray(fam[n], np.uint64) if fam.get(n) is not None else None
        self._lg = k  # keep alive
        nf = len(k["cluster"])
        K = max(1, int(fam["max_parents"]))
        for n in ("parent_pos", "color", "length", "gamma"):
            assert k[n].size == nf * K, n
        t = L.LgFamilies(int(fam["p"]), nf, K, int(fam["n_rates"]), int(data.shape[1]), L.i32p(k["cluster"]),
                         L.i32p(k["n_parents"]), L.i32p(k["child_pos"]), L.i32p(k["data_row"]), L.i32p(k["parent_pos"]),
                         L.f64p(k["length"]), L.f64p(k["gamma"]), L.i32p(k["color"]), L.f64p(data),
                         k["child_mask"].ctypes.data_as(C.POINTER(C.c_uint64)) if k["child_mask"] is not None else None,
                         k["parent_mask"].ctypes.data_as(C.POINTER(C.c_uint64)) if k["parent_mask"] is not None else None)
        _check(self._lib.pgbp_lg_setup(self._eng, C.byref(t)), self._eng)
        self._lg_p, self._lg_nrates = int(fam["p"]), int(fam["n_rates"])

    def assignfactors_lg_(self, R, mu, model="bm", alpha=None, theta=None, sync=False):
        """assignfactors!(beliefs, model, ...) (src/beliefs.jl:786-861) on the device for a Brownian motion
        (homogeneous / heterogeneous: R = the variance rate(s)) or an Ornstein-Uhlenbeck process (R = stationary
        variance), on the families given to lg_setup; only the model parameters cross the bus.
        R: [n_rates, p, p] or, per site, [n_sites, n_rates, p, p]; mu / theta: [p] or [n_sites, p]; alpha: scalar or
        [n_sites].  A random root's prior variance is one more entry of R (the root family's colour)."""
        p, nr = self._lg_p, self._lg_nrates
        R = np.asarray(R, np.float64)
        per_site = R.ndim == 4
        n = self.n_sites if per_site else 1
        R = np.ascontiguousarray(R.reshape(n, nr, p, p).transpose(0, 1, 3, 2))  # column-major blocks
        mu = np.ascontiguousarray(np.broadcast_to(np.asarray(mu, np.float64).reshape(-1, p), (n, p)))
        ou = model == "ou"
        al = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha if ou else 0.0, np.float64).reshape(-1), (n,)))
        th = np.ascontiguousarray(np.broadcast_to(np.asarray(theta if ou else np.zeros(p), np.float64).reshape(-1, p), (n, p)))
        m = L.LgParams(L.LG_OU if ou else L.LG_BM, int(per_site), L.f64p(R), L.f64p(al) if ou else None,
                       L.f64p(th) if ou else None, L.f64p(mu))
        _check(self._lib.pgbp_lg_assignfactors(self._eng, C.byref(m)), self._eng)
        if sync:
            self._invalidate()

    def loglik_lg(self, reps=1):
        """The body of score(theta) (src/calibration.jl:195-221) on the device with the parameters of the last
        assignfactors_lg_: factor fill, postorder of schedule tree 0, root integrate.  Returns (loglik[n_sites], info)."""
        o = self._opts()
        _check(self._lib.pgbp_enqueue_loglik_lg(self._eng, int(reps), C.byref(o)), self._eng)
        norm = np.zeros(self.n_sites)
        info = np.zeros(self.n_sites, dtype=np.int32)
        _check(self._lib.pgbp_fetch_loglik(self._eng, L.f64p(norm), L.i32p(info)), self._eng)
        return norm, info

    def traffic_model(self):
        b = C.c_double()
        n = C.c_int64()
        _check(self._lib.pgbp_traffic_model(self._eng, C.byref(b), C.byref(n)), self._eng)
        return b.value, n.value


class _ResidualDict:
    """messageresidual: (label_to, label_from) -> MessageResidual (src/clustergraphbeliefs.jl:11-20).
    Keys may also be (index_to, index_from)."""

    def __init__(self, owner):
        self._o = owner

    def __getitem__(self, key):
        o = self._o
        to, frm = key
        if o.cdict is not None and to in o.cdict:
            to, frm = o.cdict[to], o.cdict[frm]
        d = o._msg_id(to, frm)
        return MessageResidual(o, d, int(o._dims[o.nclusters + d // 2]))

    def __len__(self):
        return 2 * self._o.nsepsets
