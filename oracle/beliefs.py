"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restatement of belief storage, scope index maps, factor assignment and message
residuals: src/beliefs.jl and src/clustergraphbeliefs.jl.

Node labels are 1-based preorder indices exactly as in the reference
(`nodelabel`); positions inside h/J are 0-based here.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import beliefupdates as bu

CLUSTER, SEPSET = "cluster", "sepset"


class CanonicalBelief:
    """src/beliefs.jl:72-132.  inscope is (ntraits, nnodes) bool."""

    def __init__(self, nodelabel, ntraits, inscope, btype, metadata):
        self.nodelabel = [int(x) for x in nodelabel]
        self.ntraits = int(ntraits)
        self.inscope = np.asarray(inscope, dtype=bool).reshape(self.ntraits, len(self.nodelabel))
        m = int(self.inscope.sum())
        self.mu = np.zeros(m)
        self.h = np.zeros(m)
        self.J = np.zeros((m, m))
        self.g = np.zeros(1)
        self.type = btype
        self.metadata = metadata

    @property
    def dimension(self):
        return self.h.shape[0]


def scopeindex_nodes(node_labels: Sequence[int], belief: CanonicalBelief) -> np.ndarray:
    """src/beliefs.jl:354-375: positions of all in-scope traits of the listed nodes."""
    node_dims = belief.inscope.sum(axis=0)
    cums = np.concatenate([[0], np.cumsum(node_dims)])
    res = []
    for lab in node_labels:
        if lab not in belief.nodelabel:
            raise ValueError("some label is not in the belief's node labels")
        jj = belief.nodelabel.index(lab)
        res.extend(range(int(cums[jj]), int(cums[jj + 1])))
    return np.array(res, dtype=int)


def scopeindex_raw(sub_labels, sub_inscope, bel_labels, bel_inscope) -> np.ndarray:
    """src/beliefs.jl:391-405."""
    node_index = []
    for lab in sub_labels:
        if lab not in bel_labels:
            raise ValueError("subset_labels not a subset of belief_labels")
        node_index.append(list(bel_labels).index(lab))
    if node_index != sorted(node_index):
        raise ValueError("subset labels come in a different order in the belief")
    sub_inscope = np.asarray(sub_inscope, dtype=bool)
    bel_inscope = np.asarray(bel_inscope, dtype=bool)
    if np.any(sub_inscope & ~bel_inscope[:, node_index]):
        raise ValueError("some variable(s) in subset's scope yet not in full belief's scope")
    sub_in_cluster = np.zeros_like(bel_inscope)
    sub_in_cluster[:, node_index] = sub_inscope
    # findall(subset_inclusterscope[belief_inscope]): column-major vectorisation
    flat_sub = sub_in_cluster.T.reshape(-1)   # node-major, traits contiguous
    flat_bel = bel_inscope.T.reshape(-1)
    return np.nonzero(flat_sub[flat_bel])[0]


def scopeindex(sep: CanonicalBelief, clu: CanonicalBelief) -> np.ndarray:
    """src/beliefs.jl:389-390."""
    return scopeindex_raw(sep.nodelabel, sep.inscope, clu.nodelabel, clu.inscope)


class ClusterGraph:
    """Plain stand-in for the MetaGraph cluster graph (src/clustergraph.jl:842-860):
    clusters[i] = (label, [node preorder indices, decreasing]); edges[k] = (i, j, [sepset node indices])."""

    def __init__(self, clusters, edges, method="?"):
        self.clusters = [(str(l), [int(x) for x in n]) for l, n in clusters]
        self.edges = [(int(i), int(j), [int(x) for x in n]) for i, j, n in edges]
        self.method = method

    @property
    def labels(self):
        return [c[0] for c in self.clusters]

    def neighbors(self, i):
        out = []
        for (a, b, _) in self.edges:
            if a == i:
                out.append(b)
            elif b == i:
                out.append(a)
        return sorted(out)


def allocatebeliefs(tbl, taxa, net, cg: ClusterGraph, model):
    """src/beliefs.jl:478-594.  tbl: list of columns (one per trait), each a list over
    `taxa` rows with None for missing.  Returns (beliefs, (node2cluster, node2family,
    node2fixed, node2degen, cluster2nodes)); all node indices 1-based, cluster indices 0-based."""
    prenodes = net.vec_node
    numtraits = len(tbl)
    nnodes = len(prenodes)
    fixedroot = model.isrootfixed()
    pos = {id(n): i for i, n in enumerate(prenodes)}
    node2cluster = [-1] * nnodes
    node2family: List[List[int]] = [None] * nnodes
    node2fixed = [False] * nnodes
    node2degen = [False] * nnodes
    cluster2nodes: List[List[int]] = [[] for _ in cg.clusters]
    cluster2degen = [False] * len(cg.clusters)
    hasdata = np.zeros((numtraits, nnodes), dtype=bool)
    for ni in reversed(range(nnodes)):
        node = prenodes[ni]
        if node.leaf:
            if node.name not in taxa:
                raise ValueError(f"tip {node.name} in network without any data")
            i_row = list(taxa).index(node.name)
            for v in range(numtraits):
                hasdata[v, ni] = tbl[v][i_row] is not None
        i_parents = []
        degen = True
        for e in node.edges:
            if e.child is node:
                if e.length > 0:
                    degen = False
                i_parents.append(pos[id(e.parent)] + 1)
            else:
                hasdata[:, ni] |= hasdata[:, pos[id(e.child)]]
        i_parents.sort(reverse=True)
        nf = [ni + 1] + i_parents
        ci = None
        for k, (_, nodes) in enumerate(cg.clusters):
            if set(nf) <= set(nodes):
                ci = k
                break
        if ci is None:
            raise ValueError(f"no cluster containing the node family for {node.name}")
        node2cluster[ni] = ci
        node2family[ni] = nf
        if node.leaf or (ni == 0 and fixedroot):
            node2fixed[ni] = True
        node2degen[ni] = degen
        cluster2nodes[ci].append(ni + 1)
        if ni > 0 and degen:
            cluster2degen[ci] = True
    if any(cluster2degen):
        raise ValueError("degenerate node family: GeneralizedBelief path is out of scope")

    def build_inscope(nodeindices):
        insc = np.zeros((numtraits, len(nodeindices)), dtype=bool)
        for i, n1 in enumerate(nodeindices):
            node = prenodes[n1 - 1]
            if node.leaf or (n1 == 1 and fixedroot):
                continue
            insc[:, i] = hasdata[:, n1 - 1]
        return insc

    beliefs: List[CanonicalBelief] = []
    for (lab, nodes) in cg.clusters:
        beliefs.append(CanonicalBelief(nodes, numtraits, build_inscope(nodes), CLUSTER, lab))
    for (i, j, nodes) in cg.edges:
        beliefs.append(CanonicalBelief(nodes, numtraits, build_inscope(nodes), SEPSET,
                                       (cg.clusters[i][0], cg.clusters[j][0])))
    return beliefs, (node2cluster, node2family, node2fixed, node2degen, cluster2nodes)


def init_beliefs_reset(beliefs):
    """src/beliefs.jl:706-717."""
    for be in beliefs:
        be.h[:] = 0.0
        be.J[:] = 0.0
        be.g[0] = 0.0


def assignfactors(beliefs, model, tbl, taxa, net, node2cluster, node2family, node2fixed):
    """src/beliefs.jl:786-861."""
    prenodes = net.vec_node
    init_beliefs_reset(beliefs)
    numtraits = model.dimension()
    for ni, ci in enumerate(node2cluster):
        be = beliefs[ci]
        nf = node2family[ni]
        ch = prenodes[ni]
        if len(nf) == 1:
            if ni != 0:
                raise ValueError("only the root node can belong to a family of size 1")
            if node2fixed[0]:
                continue
            phi = model.factor_root()
        else:
            if len(nf) == 2:
                pe = net.parent_edges(ch)[0]
                phi = model.factor_treeedge(pe)
            else:
                pae = []
                for p1 in nf[1:]:
                    pnode = prenodes[p1 - 1]
                    for e in pnode.edges:
                        if e.child is ch:
                            pae.append(e)
                            break
                phi = model.factor_hybridnode(pae)
            if node2fixed[ni]:  # leaf
                i_row = list(taxa).index(ch.name)
                phi = bu.absorbleaf(*phi, [col[i_row] for col in tbl], rowlabel=i_row + 1)
            if any(node2fixed[p1 - 1] for p1 in nf[1:]):  # parent is the fixed root
                n = phi[0].shape[0]
                rootindex = range(n - numtraits, n)
                phi, _ = bu.absorbevidence(*phi, rootindex, list(model.rootpriormeanvector()))
        i_inscope = [x for x in nf if not node2fixed[x - 1]]
        factorind = scopeindex_nodes(i_inscope, be)
        if len(factorind) != numtraits * len(i_inscope):
            cols = [be.nodelabel.index(x) for x in i_inscope]
            var_inscope = be.inscope[:, cols]
            keep_index = np.nonzero(var_inscope.T.reshape(-1))[0]  # column-major LinearIndices
            if not node2fixed[ni]:
                # :840-852 integrate non-inscope traits of the child first, then of the parents
                kch = keep_index[keep_index < numtraits]
                integrate_ch = [i for i in range(numtraits) if i not in set(kch.tolist())]
                keep_ch = [i for i in range(phi[0].shape[0]) if i not in set(integrate_ch)]
                phi = bu.marginalize(*phi, keep_ch, None, be.metadata)
                if any(not node2fixed[p1 - 1] for p1 in nf[1:]):
                    keep_pa = keep_index[keep_index >= numtraits]
                    all_pa = range(numtraits, numtraits * len(i_inscope))
                    integrate_pa = np.array([i for i in all_pa if i not in set(keep_pa.tolist())], dtype=int)
                    shift = numtraits - len(kch)
                    keep_pa = keep_pa - shift
                    integrate_pa = integrate_pa - shift
                    phi = bu.marginalize(*phi, np.concatenate([np.arange(len(kch)), keep_pa]).astype(int),
                                         integrate_pa, be.metadata)
            else:
                phi = bu.marginalize(*phi, keep_index, None, be.metadata)
        bu.mult_inplace(be.h, be.J, be.g, factorind, *phi)


class MessageResidual:
    """src/beliefs.jl:895-924."""

    def __init__(self, s: int):
        self.dh = np.zeros(s)
        self.dJ = np.zeros((s, s))
        if s == 0:
            self.kldiv, self.iscalibrated_resid, self.iscalibrated_kl = 0.0, True, True
        else:
            self.kldiv, self.iscalibrated_resid, self.iscalibrated_kl = -1.0, False, False


def iscalibrated_residnorm_update(res: MessageResidual, atol=1e-5):
    """src/beliefs.jl:994-1003 with p = Inf."""
    def nrm(x):
        x = np.asarray(x).reshape(-1)
        if x.size == 0:
            return 0.0
        return float(np.max(np.abs(x / np.sqrt(x.size))))
    res.iscalibrated_resid = (nrm(res.dh) <= atol) and (nrm(res.dJ) <= atol)
    return res.iscalibrated_resid


class ClusterGraphBelief:
    """src/clustergraphbeliefs.jl:26-109."""

    def __init__(self, beliefs, node2cluster, node2family, node2fixed, cluster2nodes):
        types = [b.type for b in beliefs]
        nc = types.index(SEPSET) if SEPSET in types else len(beliefs)
        if not all(t == CLUSTER for t in types[:nc]):
            raise ValueError("clusters are not consecutive")
        if not all(t == SEPSET for t in types[nc:]):
            raise ValueError("sepsets are not consecutive")
        self.belief = beliefs
        self.nclusters = nc
        self.cdict = {beliefs[j].metadata: j for j in range(nc)}
        self.sdict = {frozenset(beliefs[j].metadata): j for j in range(nc, len(beliefs))}
        self.messageresidual: Dict[Tuple[str, str], MessageResidual] = {}
        for j in range(nc, len(beliefs)):
            l1, l2 = beliefs[j].metadata
            s = beliefs[j].dimension
            self.messageresidual[(l1, l2)] = MessageResidual(s)
            self.messageresidual[(l2, l1)] = MessageResidual(s)
        # factors: copy of the initial cluster beliefs (src/beliefs.jl:604-637)
        self.factor = [(b.h.copy(), b.J.copy(), b.g.copy()) for b in beliefs[:nc]]
        self.node2cluster, self.node2family = node2cluster, node2family
        self.node2fixed, self.cluster2nodes = node2fixed, cluster2nodes

    def sepsetindex(self, l1, l2):
        return self.sdict[frozenset((l1, l2))]

    def clusterindex(self, lab):
        return self.cdict[lab]

    def init_beliefs_reset_fromfactors(self):
        """src/clustergraphbeliefs.jl:126-139."""
        for i in range(self.nclusters):
            h, J, g = self.factor[i]
            self.belief[i].h[:] = h
            self.belief[i].J[:] = J
            self.belief[i].g[0] = g[0]
        for i in range(self.nclusters, len(self.belief)):
            self.belief[i].h[:] = 0.0
            self.belief[i].J[:] = 0.0
            self.belief[i].g[0] = 0.0

    def init_messagecalibrationflags_reset(self, reset_kl=True):
        """src/beliefs.jl:973-979, src/clustergraphbeliefs.jl:146-150."""
        for mr in self.messageresidual.values():
            if mr.dh.size == 0:
                continue
            if reset_kl:
                mr.kldiv = -1.0
            mr.iscalibrated_resid = False
            mr.iscalibrated_kl = False

    def iscalibrated_residnorm(self):
        """src/clustergraphbeliefs.jl:168-169."""
        return all(mr.iscalibrated_resid for mr in self.messageresidual.values())

    def integratebelief(self, j):
        """src/clustergraphbeliefs.jl:194 -> src/beliefupdates.jl:168-172."""
        b = self.belief[j]
        mu, norm = bu.integratebelief(b.h, b.J, b.g[0])
        b.mu[:] = mu
        return mu, norm

    def default_sepset1(self):
        """src/clustergraphbeliefs.jl:197-202 (0-based result)."""
        for j in range(self.nclusters, len(self.belief)):
            if len(self.belief[j].nodelabel) == 1:
                return j
        raise ValueError("no sepset with a single node")


def propagate_belief(cluster_to: CanonicalBelief, sepset: CanonicalBelief,
                     cluster_from: CanonicalBelief, residual: MessageResidual = None):
    """src/beliefupdates.jl:634-665.  With `residual`: returns None or the
    BPPosDefException (returned, not raised).  Without: returns (dh, dJ, dg), raises."""
    def core():
        keep = scopeindex(sepset, cluster_from)
        h, J, g = bu.marginalize(cluster_from.h, cluster_from.J, cluster_from.g[0], keep, None,
                                 cluster_from.metadata)
        dh, dJ, dg = bu.divide(sepset.h, sepset.J, sepset.g[0], h, J, g)
        sepset.h[:] = h
        sepset.J[:] = J
        sepset.g[0] = g
        bu.mult_inplace(cluster_to.h, cluster_to.J, cluster_to.g, scopeindex(sepset, cluster_to), dh, dJ, dg)
        return dh, dJ, dg
    if residual is None:
        return core()
    try:
        dh, dJ, _ = core()
    except bu.BPPosDefException as ex:
        return ex
    residual.dh[:] = dh
    residual.dJ[:] = dJ
    return None


# ---------------------------------------------------------------------------
# scores: src/score.jl (the loopy-BP objective; exact on a calibrated clique tree)
# ---------------------------------------------------------------------------

def entropy(J):
    """src/score.jl:58-66: entropy of N(., J^-1); 0 for an empty J."""
    J = np.atleast_2d(np.asarray(J, dtype=float))
    n = J.shape[1]
    if n == 0 or J.size == 0:
        return 0.0
    S = np.triu(J) + np.triu(J, 1).T  # Symmetric(J): upper triangle
    sign, ld = np.linalg.slogdet(S)
    if sign <= 0:
        raise np.linalg.LinAlgError("entropy: J is not positive definite")
    return (n * (bu.LOG2PI + 1.0) - ld) / 2.0


def average_energy(Jr, hr, Jt, ht, gt):
    """src/score.jl:105-117: E_r[-log C(x; Jt, ht, gt)] with r = N(Jr^-1 hr, Jr^-1)."""
    Jt = np.atleast_2d(np.asarray(Jt, dtype=float))
    if Jt.size == 0:
        return -float(gt)
    Jr = np.asarray(Jr, dtype=float)
    S = np.triu(Jr) + np.triu(Jr, 1).T
    L = np.linalg.cholesky(S)  # raises if not PD
    mu = np.linalg.solve(S, np.asarray(hr, dtype=float))
    return (float(np.trace(np.linalg.solve(S, Jt))) + float(mu @ Jt @ mu)) / 2.0 - float(np.asarray(ht) @ mu) - float(gt)


def free_energy(cgb: "ClusterGraphBelief"):
    """src/score.jl:162-182: (average energy, approximate entropy, free energy)."""
    b = cgb.belief
    nclu = cgb.nclusters
    ave, ent = 0.0, 0.0
    for i in range(nclu):
        fh, fJ, fg = cgb.factor[i]
        if fJ.size == 0:
            ave -= float(fg[0])
        else:
            ave += average_energy(b[i].J, b[i].h, fJ, fh, fg[0])
            ent += entropy(b[i].J)
    for i in range(nclu, len(b)):
        ent -= entropy(b[i].J)
    return ave, ent, ave - ent


def factored_energy(cgb: "ClusterGraphBelief"):
    """src/score.jl:151-154."""
    a, e, f = free_energy(cgb)
    return a, e, -f


def residual_kldiv(res: MessageResidual, sep: CanonicalBelief, atol=1e-5):
    """src/beliefs.jl:1060-1075: KL(message || previous sepset belief); updates res.kldiv / iscalibrated_kl."""
    if sep.J.size == 0:
        return True
    try:
        J0 = np.triu(sep.J) + np.triu(sep.J, 1).T
        np.linalg.cholesky(J0)
        mu0 = np.linalg.solve(J0, sep.h)
        sep.mu = mu0  # side product of getcholesky_μ! (src/score.jl:32-36)
        J1f = sep.J - res.dJ
        J1 = np.triu(J1f) + np.triu(J1f, 1).T
        np.linalg.cholesky(J1)
        mu1 = np.linalg.solve(J1, sep.h - res.dh)
    except np.linalg.LinAlgError:
        return False
    d = mu1 - mu0
    res.kldiv = (-float(np.trace(np.linalg.solve(J0, res.dJ))) + float(d @ J1 @ d)
                 + np.linalg.slogdet(J0)[1] - np.linalg.slogdet(J1)[1]) / 2.0
    res.iscalibrated_kl = abs(res.kldiv) <= atol
    return res.iscalibrated_kl


# ---------------------------------------------------------------------------
# regularisation: src/clustergraphbeliefs.jl:235-403 (graph walks on the host + diagonal updates)
# ---------------------------------------------------------------------------

def _neighbors(cgb: "ClusterGraphBelief"):
    """cluster index -> [(neighbor cluster index, sepset belief index)] in sepset order"""
    nb = {i: [] for i in range(cgb.nclusters)}
    for j in range(cgb.nclusters, len(cgb.belief)):
        a, c = (cgb.cdict[l] for l in cgb.belief[j].metadata)
        nb[a].append((c, j))
        nb[c].append((a, j))
    return nb


def regularizebeliefs_1clustersepset(cluster: CanonicalBelief, sepset: CanonicalBelief, eps):
    """src/clustergraphbeliefs.jl:264-275."""
    upind = scopeindex(sepset, cluster)
    if upind.size == 0:
        return
    cluster.J[upind, upind] += eps
    sepset.J[np.diag_indices_from(sepset.J)] += eps


def regularizebeliefs_bycluster(cgb: "ClusterGraphBelief"):
    """src/clustergraphbeliefs.jl:235-249."""
    nb = _neighbors(cgb)
    for ci in range(cgb.nclusters):
        cl = cgb.belief[ci]
        eps = max(bu.EPS, float(np.max(np.abs(cl.J))) if cl.J.size else 0.0)
        for (_, sj) in nb[ci]:
            regularizebeliefs_1clustersepset(cl, cgb.belief[sj], eps)


def regularizebeliefs_onschedule(cgb: "ClusterGraphBelief"):
    """src/clustergraphbeliefs.jl:376-403: default messages eps*I, then real messages, cluster by cluster."""
    nb = _neighbors(cgb)
    sent = set()
    eps0 = float(np.sqrt(bu.EPS))
    labels = [b.metadata for b in cgb.belief[:cgb.nclusters]]
    for ci in range(cgb.nclusters):
        cl = cgb.belief[ci]
        eps = max(float(np.max(np.abs(cl.J))) if cl.J.size else 0.0, eps0)
        tosend = []
        for (ni, sj) in nb[ci]:
            if (ni, ci) not in sent:
                regularizebeliefs_1clustersepset(cl, cgb.belief[sj], eps)
                sent.add((ni, ci))
            if (ci, ni) not in sent:
                tosend.append((ni, sj))
                sent.add((ci, ni))
        for (ni, sj) in tosend:
            propagate_belief(cgb.belief[ni], cgb.belief[sj], cl, cgb.messageresidual[(labels[ni], labels[ci])])


def scopeindex_node(node_lab: int, sep: CanonicalBelief, clu: CanonicalBelief):
    """src/beliefs.jl:418-436: (ind_in_sepset, ind_in_cluster) of the node's traits shared by both scopes."""
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
    ind_sep = np.nonzero(s_insc.T.reshape(-1)[sep.inscope.T.reshape(-1)])[0]
    ind_clu = np.nonzero(c_insc.T.reshape(-1)[clu.inscope.T.reshape(-1)])[0]
    return ind_sep, ind_clu


def regularizebeliefs_bynodesubtree(cgb: "ClusterGraphBelief"):
    """src/clustergraphbeliefs.jl:306-340.  For each node: the clusters and sepsets holding it form a tree
    (running intersection); rooted at the cluster of largest first preorder index, every non-root cluster
    and the sepset to its parent get +eps on the node's shared traits."""
    nc = cgb.nclusters
    b = cgb.belief
    sep_ends = [tuple(cgb.cdict[l] for l in b[j].metadata) for j in range(nc, len(b))]
    nodes = sorted({v for i in range(nc) for v in b[i].nodelabel})
    for v in nodes:
        cl = [i for i in range(nc) if v in b[i].nodelabel]
        if len(cl) <= 1:
            continue
        ed = [(a, c, nc + k) for k, (a, c) in enumerate(sep_ends) if v in b[nc + k].nodelabel]
        if len(ed) != len(cl) - 1:
            raise ValueError(f"running intersection violated for node / variable {v}")
        root = max(cl, key=lambda i: b[i].nodelabel[0])
        eps = bu.EPS
        for i in cl:
            if b[i].J.size:   # maximum(abs, J) of an empty J is 0 in Julia (Base.mapreduce_empty(abs, max, T))
                eps = max(eps, float(np.max(np.abs(b[i].J))))
        nbr = {i: [] for i in cl}
        for (a, c, j) in ed:
            nbr[a].append((c, j))
            nbr[c].append((a, j))
        seen, stack = {root}, [root]
        while stack:
            p = stack.pop()
            for (c, j) in nbr[p]:
                if c in seen:
                    continue
                seen.add(c)
                stack.append(c)
                s_ind, c_ind = scopeindex_node(v, b[j], b[c])
                b[c].J[c_ind, c_ind] += eps
                b[j].J[s_ind, s_ind] += eps
        if len(seen) != len(cl):
            raise ValueError(f"running intersection violated for node / variable {v}")
