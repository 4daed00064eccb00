"""The three regularisers of src/clustergraphbeliefs.jl:235-403 for a device-resident ClusterGraphBelief.

All three add positive values to diagonal entries of cluster and sepset precisions so that every belief
becomes non-degenerate while the graphical model (product of cluster beliefs / product of sepset beliefs)
is unchanged.  `regularizebeliefs_bycluster_` is the one inside the optimisation loop of
calibrate_optimize_clustergraph! (src/calibration.jl:335-343) and runs entirely on the device; the other two
are graph walks on the host that edit single beliefs through pgbp_get_belief / pgbp_set_belief and send the
real messages of `regularizebeliefs_onschedule!` with pgbp_propagate."""
import numpy as np

from . import _lib as L
from .clustergraphbeliefs import _check

_EPS = float(np.finfo(np.float64).eps)


def regularizebeliefs_bycluster_(beliefs, clustergraph=None, sync=True):
    """regularizebeliefs_bycluster!(beliefs, clustergraph) (src/clustergraphbeliefs.jl:235-249).
    The cluster graph is the one the ClusterGraphBelief was built on; the argument is accepted for
    signature parity."""
    _check(beliefs._lib.pgbp_regularize_bycluster(beliefs._eng), beliefs._eng)
    if sync:
        beliefs.pull()


def _neighbors(beliefs):
    nb = [[] for _ in range(beliefs.nclusters)]
    for k in range(beliefs.nsepsets):
        a, b = (int(x) for x in beliefs._sepcl[k])
        nb[a].append((b, k, 0))
        nb[b].append((a, k, 1))
    return nb


def _scope(beliefs, k, side):
    return beliefs._scope_idx[beliefs._scope_off[2 * k + side]: beliefs._scope_off[2 * k + side + 1]]


def regularizebeliefs_bynodesubtree_(beliefs, clustergraph=None):
    """regularizebeliefs_bynodesubtree!(beliefs, clustergraph) (src/clustergraphbeliefs.jl:306-340): for every
    node, over the tree of clusters and sepsets that hold it (rooted at the cluster with the largest first
    preorder index): eps = max(eps(T), max|J| over those clusters); +eps on the node's shared traits in every
    non-root cluster and in the sepset to its parent.  Needs beliefs built from CanonicalBelief objects
    (node labels and scopes)."""
    if beliefs._objs is None:
        raise ValueError("regularizebeliefs_bynodesubtree_ needs node labels: build the ClusterGraphBelief from beliefs")
    nc = beliefs.nclusters
    b = beliefs._objs
    beliefs.pull()
    nodes = sorted({v for i in range(nc) for v in b[i].nodelabel})
    for v in nodes:
        cl = [i for i in range(nc) if v in b[i].nodelabel]
        if len(cl) <= 1:
            continue
        ed = [(int(beliefs._sepcl[k][0]), int(beliefs._sepcl[k][1]), nc + k) for k in range(beliefs.nsepsets)
              if v in b[nc + k].nodelabel]
        root = max(cl, key=lambda i: b[i].nodelabel[0])
        nbr = {i: [] for i in cl}
        for (a, c, j) in ed:
            nbr[a].append((c, j))
            nbr[c].append((a, j))
        order, seen, stack = [], {root}, [root]
        while stack:
            p = stack.pop()
            for (c, j) in nbr[p]:
                if c not in seen:
                    seen.add(c)
                    stack.append(c)
                    order.append((c, j))
        if len(ed) != len(cl) - 1 or len(seen) != len(cl):
            raise ValueError(f"running intersection violated for node / variable {v}")
        for site in range(beliefs.n_sites):
            eps = _EPS
            for i in cl:
                J = beliefs._views(site, i)[0]
                if J.size:   # maximum(abs, J) of an empty J is 0 in Julia
                    eps = max(eps, float(np.max(np.abs(J))))
            for (c, j) in order:
                s_ind, c_ind = _scopeindex_node(v, b[j], b[c])
                Jc, Js = beliefs._views(site, c)[0], beliefs._views(site, j)[0]
                Jc[c_ind, c_ind] += eps
                Js[s_ind, s_ind] += eps
    beliefs.push()


def regularizebeliefs_bynodesubtree_arrays_(beliefs, cluster_nodes, edges, sepset_nodes, scopes):
    """regularizebeliefs_bynodesubtree! (src/clustergraphbeliefs.jl:306-340) for cluster graphs built on plain arrays
    (networks.py:allocate_scopes: `scopes.clusters[i]` = node labels + in-scope mask of cluster i), in time linear in
    the total cluster membership: the per-node cluster and sepset lists are indexed once instead of searched per node
    (50 000-node networks).  Nodes in increasing label order; every node's eps sees the edits of the nodes before it,
    as in the reference's sequential loop."""
    nc = beliefs.nclusters
    beliefs.pull()
    p_traits = scopes.clusters[0].inscope.shape[0] if nc else 0
    # position of every node's in-scope block inside its clusters / sepsets
    ndim = {}
    cpos = []
    for i, sc in enumerate(scopes.clusters):
        acc, d = 0, {}
        for j, lab in enumerate(sc.nodelabel):
            k = int(sc.inscope[:, j].sum())
            d[lab] = (acc, k)
            ndim[lab] = k
            acc += k
        cpos.append(d)
    spos = []
    for nodes in sepset_nodes:
        acc, d = 0, {}
        for lab in nodes:
            k = ndim.get(lab, 0)
            d[lab] = (acc, k)
            acc += k
        spos.append(d)
    holders, node_seps = {}, {}
    for i, sc in enumerate(scopes.clusters):
        for lab in sc.nodelabel:
            holders.setdefault(lab, []).append(i)
    for k, nodes in enumerate(sepset_nodes):
        for lab in nodes:
            node_seps.setdefault(lab, []).append(k)
    for site in range(beliefs.n_sites):
        Jc = [beliefs._views(site, i)[0] for i in range(nc)]
        Js = [beliefs._views(site, nc + k)[0] for k in range(beliefs.nsepsets)]
        maxabs = np.array([float(np.max(np.abs(J))) if J.size else 0.0 for J in Jc])
        for v in sorted(holders):
            cl = holders[v]
            if len(cl) <= 1 or ndim.get(v, 0) == 0:
                continue
            root = max(cl, key=lambda i: scopes.clusters[i].nodelabel[0])
            nbr = {i: [] for i in cl}
            ne = 0
            for k in node_seps.get(v, []):
                a, c = int(edges[k][0]), int(edges[k][1])
                nbr[a].append((c, k))
                nbr[c].append((a, k))
                ne += 1
            order, seen, stack = [], {root}, [root]
            while stack:
                q = stack.pop()
                for (c, k) in nbr[q]:
                    if c not in seen:
                        seen.add(c)
                        stack.append(c)
                        order.append((c, k))
            if ne != len(cl) - 1 or len(seen) != len(cl):
                raise ValueError(f"running intersection violated for node / variable {v}")
            eps = max(_EPS, float(maxabs[cl].max()))
            for (c, k) in order:
                o, d = cpos[c][v]
                idx = np.arange(o, o + d)
                Jc[c][idx, idx] += eps
                maxabs[c] = max(maxabs[c], float(np.max(np.abs(Jc[c][idx, idx]))))
                o, d = spos[k][v]
                idx = np.arange(o, o + d)
                Js[k][idx, idx] += eps
    beliefs.push()


def _scopeindex_node(node_lab, sep, clu):
    """scopeindex(node_label, sepset, cluster) (src/beliefs.jl:418-436)."""
    if node_lab not in sep.nodelabel:
        raise ValueError(f"{node_lab} not in sepset")
    if node_lab not in clu.nodelabel:
        raise ValueError(f"{node_lab} not in cluster")
    s_j, c_j = sep.nodelabel.index(node_lab), clu.nodelabel.index(node_lab)
    s_node = sep.inscope[:, s_j]
    if np.any(s_node & ~clu.inscope[:, c_j]):
        raise ValueError(f"some traits are in sepset's but not in cluster's scope for node {node_lab}")
    s_insc = np.zeros_like(sep.inscope)
    s_insc[:, s_j] = s_node
    c_insc = np.zeros_like(clu.inscope)
    c_insc[:, c_j] = s_node
    return (np.nonzero(s_insc.T.reshape(-1)[sep.inscope.T.reshape(-1)])[0],
            np.nonzero(c_insc.T.reshape(-1)[clu.inscope.T.reshape(-1)])[0])


def regularizebeliefs_onschedule_(beliefs, clustergraph=None):
    """regularizebeliefs_onschedule!(beliefs, clustergraph) (src/clustergraphbeliefs.jl:376-403): clusters in
    label order; each first receives the default message eps*I (eps = max(max|J|, sqrt(eps(T)))) on every
    edge it has not heard from, then sends its real message (propagate_belief! on the device) on every edge
    it has not yet used.  One site (beliefs.site)."""
    nb = _neighbors(beliefs)
    nc = beliefs.nclusters
    lib, eng, site = beliefs._lib, beliefs._eng, beliefs.site
    sent = set()
    eps0 = float(np.sqrt(_EPS))

    def fetch(i):
        rec = beliefs._packed[site, beliefs._poff[i]: beliefs._poff[i + 1]]
        if rec.size:
            _check(lib.pgbp_get_belief(eng, site, i, L.f64p(rec)), eng)

    def store(i):
        rec = beliefs._packed[site, beliefs._poff[i]: beliefs._poff[i + 1]]
        _check(lib.pgbp_set_belief(eng, site, i, L.f64p(np.ascontiguousarray(rec))), eng)

    for ci in range(nc):
        fetch(ci)
        J = beliefs._views(site, ci)[0]
        eps = max(float(np.max(np.abs(J))) if J.size else 0.0, eps0)
        tosend, edited = [], False
        for (ni, k, side) in nb[ci]:
            if (ni, ci) not in sent:
                up = _scope(beliefs, k, side)
                if len(up):
                    fetch(nc + k)
                    Js = beliefs._views(site, nc + k)[0]
                    J[up, up] += eps
                    Js[np.diag_indices_from(Js)] += eps
                    store(nc + k)
                    edited = True
                sent.add((ni, ci))
            if (ci, ni) not in sent:
                tosend.append((ni, k))
                sent.add((ci, ni))
        if edited:
            store(ci)
        for (ni, k) in tosend:
            flag = beliefs._propagate(ni, nc + k, ci, sync=False)
            if flag is not None:
                raise flag
    beliefs.pull()
