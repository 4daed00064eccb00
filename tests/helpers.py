"""Shared test helpers (tests only): golden loading, model construction from the
fixture dictionaries, building oracle cluster-graph beliefs."""
import json
import os

import numpy as np

from oracle import beliefs as OB
from oracle import calibration as OC
from oracle import clustergraph as OCG
from oracle import models as OM
from oracle import network as ON

HERE = os.path.dirname(os.path.abspath(__file__))


def goldens():
    with open(os.path.join(HERE, "golden", "reference_goldens.json")) as f:
        return json.load(f)


def _num(x):
    if isinstance(x, str):
        return float(x)
    if isinstance(x, list):
        return [_num(v) for v in x]
    return x


def make_model(d):
    k = d["kind"]
    v = _num(d.get("v")) if "v" in d else None
    if k == "UnivariateBM":
        return OM.UnivariateBrownianMotion(d["sigma2"], d["mu"], v)
    if k == "MvDiagBM":
        return OM.MvDiagBrownianMotion(d["R"], d["mu"], v)
    if k == "MvFullBM":
        return OM.MvFullBrownianMotion(d["R"], d["mu"], v)
    if k == "UnivariateOU":
        return OM.UnivariateOrnsteinUhlenbeck(d["sigma2"], d["alpha"], d["theta"], d["mu"], v)
    if k == "HeteroBM":
        return OM.HeterogeneousBrownianMotion(d["rates"], {int(a): b for a, b in d["colors"].items()}, d["mu"], v)
    raise KeyError(k)


def oracle_setup(net, cg, model, tbl, taxa):
    """allocatebeliefs + assignfactors + ClusterGraphBelief with the oracle."""
    b, (n2c, n2f, n2fix, n2d, c2n) = OB.allocatebeliefs(tbl, taxa, net, cg, model)
    OB.assignfactors(b, model, tbl, taxa, net, n2c, n2f, n2fix)
    return OB.ClusterGraphBelief(b, n2c, n2f, n2fix, c2n)


# ---------------------------------------------------------------------------
# bridges between the oracle's objects and the product's host mirror
# ---------------------------------------------------------------------------

def product_beliefs_from_oracle(obeliefs):
    """Oracle CanonicalBelief list -> product CanonicalBelief list (same scopes, same h,J,g)."""
    import pgbp_amd
    out = []
    for b in obeliefs:
        pb = pgbp_amd.CanonicalBelief(b.nodelabel, b.ntraits, b.inscope, b.type, b.metadata)
        pb.h[:] = b.h
        pb.J[:] = b.J
        pb.g[:] = b.g
        out.append(pb)
    return out


def oracle_cgb_from_problem(prob, packed, p):
    """Oracle ClusterGraphBelief with the scopes / values of a pgbp_amd.synth clique-tree Problem."""
    nb = len(prob.dims)
    nc = prob.nclusters
    beliefs = []
    for i in range(nb):
        m = int(prob.dims[i])
        if i < nc:
            labs = [int(x) for x in prob.cluster_nodes[i]]
            # which of (child, parent) is in scope: recover from dims and the scope maps
            insc = np.zeros((p, 2), dtype=bool)
            child_dim = prob.meta["child_dim"][i]
            insc[:, 0] = child_dim > 0
            insc[:, 1] = (m - child_dim) > 0
            b = OB.CanonicalBelief(labs, p, insc, OB.CLUSTER, i)
        else:
            k = i - nc
            a, c = prob.sepset_clusters[k]
            b = OB.CanonicalBelief([int(prob.sepset_nodes[k])], p, np.full((p, 1), m > 0), OB.SEPSET, (int(a), int(c)))
        o = prob.packed_off[i]
        b.J[:] = packed[o:o + m * m].reshape(m, m, order="F")
        b.h[:] = packed[o + m * m:o + m * m + m]
        b.g[0] = packed[o + m * m + m]
        beliefs.append(b)
    return OB.ClusterGraphBelief(beliefs, None, None, None, None)


def oracle_schedule(prob):
    pa, ch = prob.schedule[0]
    return ([int(x) for x in pa], [int(x) for x in ch], [int(x) for x in pa], [int(x) for x in ch])


def pack_oracle(cgb, prob):
    out = np.zeros(int(prob.packed_off[-1]))
    for i, b in enumerate(cgb.belief):
        m = b.dimension
        o = prob.packed_off[i]
        out[o:o + m * m] = b.J.reshape(-1, order="F")
        out[o + m * m:o + m * m + m] = b.h
        out[o + m * m + m] = b.g[0]
    return out


def lg_inputs_from_oracle(P, net, ocgb, model, tbl, taxa):
    """Inputs of the device factor fill (pgbp_amd.lg_families + assignfactors_lg_ keyword arguments) for the
    oracle's network / model objects.  Returns (families, data[n_rows, p], kwargs)."""
    prenodes = net.vec_node
    p = model.dimension()
    hetero = isinstance(model, OM.HeterogeneousBrownianMotion)
    parent_edges, data_row = [], []
    for ni, nf in enumerate(ocgb.node2family):
        ch = prenodes[ni]
        row = []
        for p1 in nf[1:]:
            pnode = prenodes[p1 - 1]
            e = next(e for e in pnode.edges if e.child is ch)   # src/beliefs.jl:813-820
            row.append((e.length, e.gamma if len(nf) > 2 else 1.0, model._c(e) if hetero else 0))
        parent_edges.append(row)
        data_row.append(list(taxa).index(ch.name) if ch.leaf else -1)
    if hetero:
        rates = [np.asarray(r, float) for r in model.rates]
    elif isinstance(model, OM.UnivariateOrnsteinUhlenbeck):
        rates = [np.array([[model.gamma2]])]
    else:
        rates = [np.asarray(model.R, float)]
    root_color = None
    v = np.atleast_2d(np.asarray(model.rootpriorvariance(), float))
    if not model.isrootfixed() and not np.any(np.isinf(np.diag(v))):
        root_color = len(rates)
        rates = rates + [v]
    data = np.array([[np.nan if tbl[v][r] is None else float(tbl[v][r]) for v in range(p)] for r in range(len(taxa))])
    fam = P.lg_families(ocgb.belief[:ocgb.nclusters], ocgb.node2cluster, ocgb.node2family, ocgb.node2fixed,
                        parent_edges, data_row, p, n_rates=len(rates), root_prior_color=root_color, data=data)
    kw = dict(R=np.stack(rates), mu=model.rootpriormeanvector())
    if isinstance(model, OM.UnivariateOrnsteinUhlenbeck):
        kw.update(model="ou", alpha=model.alpha, theta=[model.theta])
    return fam, data, kw


def network_from_newick_file(P, path):
    """(product NetArrays, names, oracle Network with the same preorder and names, tip names in file order =
    PhyloNetworks.tiplabels) for a network file of tests/golden/."""
    import re
    with open(path) as f:
        s = f.read()
    net, names = P.read_newick(s)
    nodes = [ON.Node(name=names[i], leaf=bool(net.is_leaf[i]), hybrid=len(net.node2family[i]) > 2) for i in range(net.nnodes)]
    edges = []
    for i, nf in enumerate(net.node2family):
        for k, pl in enumerate(nf[1:]):
            e = ON.Edge(number=len(edges) + 1, parent=nodes[pl - 1], child=nodes[i], length=net.length[i][k],
                        gamma=net.gamma[i][k], hybrid=len(nf) > 2)
            edges.append(e)
            nodes[pl - 1].edges.append(e)
            nodes[i].edges.append(e)
    onet = ON.Network(nodes[0], nodes, edges)
    onet.set_preorder(names)
    tipset = {n for n, leaf in zip(names, net.is_leaf) if leaf}
    tips = [t for t in re.findall(r"[(,]([A-Za-z_][A-Za-z0-9_.|]*)", s) if t in tipset]
    return net, names, onet, tips


def oracle_cluster_factor(model, net, st, X, ci, families_of_cluster=None):
    """The factor the ORACLE's assignfactors! restatement (oracle/beliefs.py:assignfactors, src/beliefs.jl:786-861) gives
    cluster `ci` of a plain-array problem (pgbp_amd.networks: NetArrays `net`, ScopeTables `st`), complete data, node
    values X[node] (tips: the observed data): the loop body of assignfactors! replayed for the node families assigned to
    this one cluster, with the oracle's own factor formulas (oracle/models.py: factor_treeedge / factor_hybridnode /
    factor_root) and absorbleaf / absorbevidence / mult! (oracle/beliefupdates.py) -- no network object needed, so a
    sample of clusters of a 50 000-node network costs milliseconds.  Returns (h, J, g) in the cluster's variable order."""
    import types
    import numpy as np
    from oracle import beliefupdates as bu
    from oracle.beliefs import scopeindex_nodes
    p = model.dimension()
    cl = st.clusters[ci]
    be = types.SimpleNamespace(nodelabel=list(cl.nodelabel), inscope=np.asarray(cl.inscope, bool), metadata=f"cluster {ci}")
    m = int(be.inscope.sum())
    be.h, be.J, be.g = np.zeros(m), np.zeros((m, m)), np.zeros(1)
    fams = families_of_cluster if families_of_cluster is not None else [ni for ni, c in enumerate(st.node2cluster) if c == ci]
    for ni in fams:                                   # increasing node index = the reference's loop order
        nf = net.node2family[ni]
        if len(nf) == 1:
            if st.node2fixed[0]:
                continue
            phi = model.factor_root()
        else:
            pae = [types.SimpleNamespace(length=net.length[ni][k], gamma=net.gamma[ni][k], number=(ni, k))
                   for k in range(len(nf) - 1)]
            phi = model.factor_treeedge(pae[0]) if len(nf) == 2 else model.factor_hybridnode(pae)
            if st.node2fixed[ni]:                     # leaf: absorb its data
                phi = bu.absorbleaf(*phi, list(X[ni]), rowlabel=ni + 1)
            if any(st.node2fixed[p1 - 1] for p1 in nf[1:]):   # parent is the fixed root
                n = phi[0].shape[0]
                phi, _ = bu.absorbevidence(*phi, range(n - p, n), list(model.rootpriormeanvector()))
        i_inscope = [x for x in nf if not st.node2fixed[x - 1]]
        factorind = scopeindex_nodes(i_inscope, be)
        assert len(factorind) == p * len(i_inscope)   # complete data
        bu.mult_inplace(be.h, be.J, be.g, factorind, *phi)
    return be.h, be.J, float(be.g[0])
