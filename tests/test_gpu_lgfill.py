"""
GPU parity tests of the device factor assignment for every linear-Gaussian model of the reference
(pgbp_lg_setup / pgbp_lg_assignfactors / pgbp_enqueue_loglik_lg of include/pgbp.h) against the oracle's
restatement of assignfactors! (src/beliefs.jl:786-861) and the reference's golden likelihoods.
Tolerance: factors within 1e-10 * max(1, |.|_inf) per belief, log-likelihoods 1e-8 relative.
"""
import numpy as np
import pytest

from helpers import goldens, lg_inputs_from_oracle, make_model, oracle_setup, product_beliefs_from_oracle
from oracle import beliefs as OB
from oracle import calibration as OC
from oracle import clustergraph as OCG
from oracle import densemvn as OD
from oracle import models as OM
from oracle import network as ON

pytestmark = pytest.mark.gpu
G = goldens()


@pytest.fixture(scope="module")
def P():
    import pgbp_amd
    pgbp_amd.load()
    return pgbp_amd


def _zeroed_product(P, ocgb):
    """Product ClusterGraphBelief with the oracle's scopes and all-zero beliefs (what allocatebeliefs returns)."""
    pb = product_beliefs_from_oracle(ocgb.belief)
    for b in pb:
        b.J[...] = 0.0
        b.h[...] = 0.0
        b.g[...] = 0.0
    return P.ClusterGraphBelief(pb, ocgb.node2cluster, ocgb.node2family, ocgb.node2fixed, ocgb.cluster2nodes)


def _assert_factors_equal(pcgb, ocgb, rtol=1e-10):
    for i in range(ocgb.nclusters):
        ob, pb = ocgb.belief[i], pcgb.belief[i]
        for name, x, y in (("J", pb.J, ob.J), ("h", pb.h, ob.h), ("g", pb.g, ob.g)):
            x, y = np.asarray(x), np.asarray(y)
            if x.size:
                scale = max(1.0, float(np.max(np.abs(y))))
                assert float(np.max(np.abs(x - y))) <= rtol * scale, (i, name, x, y)
    for i in range(ocgb.nclusters, len(ocgb.belief)):
        pb = pcgb.belief[i]
        assert not np.any(pb.J) and not np.any(pb.h) and pb.g[0] == 0.0


def _device_fill(P, net, cg, model, tbl, taxa):
    ocgb = oracle_setup(net, cg, model, tbl, taxa)
    pcgb = _zeroed_product(P, ocgb)
    fam, data, kw = lg_inputs_from_oracle(P, net, ocgb, model, tbl, taxa)
    pcgb.lg_setup(fam, data)
    pcgb.assignfactors_lg_(sync=True, **kw)
    return ocgb, pcgb


@pytest.mark.parametrize("case", G["evomodels_postorder"]["cases"], ids=lambda c: c["name"])
def test_lgfill_evomodels_goldens(P, case):
    """test/test_evomodels.jl:74-264 (every case, the ones with missing tip values included): factors filled on the
    device, postorder, integratebelief! at the root cluster = the golden log-likelihood."""
    g = G["evomodels_postorder"]
    net = ON.read_newick(g["net"])
    tbl = [g[t] for t in case["traits"]]
    ct = OCG.cliquetree(net)
    spt = OCG.spanningtree_clusterlist(ct, OCG.default_rootcluster(ct, net))
    ocgb, pcgb = _device_fill(P, net, ct, make_model(case["model"]), tbl, g["taxa"])
    _assert_factors_equal(pcgb, ocgb)
    pcgb.set_schedule([spt])
    ll, info = pcgb.loglik_lg()
    assert not info.any()
    assert abs(ll[0] - case["ll"]) <= 1e-8 * max(1.0, abs(case["ll"])), (ll, case["ll"])


def _models(p, rng, net, which):
    A = rng.normal(size=(p, p))
    R = A @ A.T / p + np.eye(p)
    if which == "bm_fixed":
        return OM.MvFullBrownianMotion(R, rng.normal(size=p))
    if which == "bm_random_root":
        B = rng.normal(size=(p, p))
        return OM.MvFullBrownianMotion(R, rng.normal(size=p), B @ B.T / p + 0.5 * np.eye(p))
    if which == "bm_improper_root":
        return OM.MvDiagBrownianMotion(np.diag(R), rng.normal(size=p), np.full(p, np.inf))
    if which == "hetero":
        rates = [R * s for s in (0.5, 1.0, 2.5)]
        colors = {e.number: 1 + int(rng.integers(3)) for e in net.edges}
        return OM.HeterogeneousBrownianMotion(rates, colors, rng.normal(size=p))
    if which == "hetero_random_root":
        rates = [R * s for s in (0.5, 2.0)]
        colors = {e.number: 1 + int(rng.integers(2)) for e in net.edges}
        return OM.HeterogeneousBrownianMotion(rates, colors, rng.normal(size=p), np.eye(p) * 0.7)
    if which == "ou_fixed":
        return OM.UnivariateOrnsteinUhlenbeck(rng.uniform(0.5, 2), rng.uniform(0.1, 1), rng.normal(), rng.normal(), 0.0)
    if which == "ou_random_root":
        return OM.UnivariateOrnsteinUhlenbeck(rng.uniform(0.5, 2), rng.uniform(0.1, 1), rng.normal(), rng.normal(), 0.8)
    raise KeyError(which)


@pytest.mark.parametrize("graph", ["cliquetree", "bethe"])
@pytest.mark.parametrize("which,p", [("bm_fixed", 3), ("bm_random_root", 2), ("bm_improper_root", 2), ("hetero", 4),
                                     ("hetero_random_root", 2), ("ou_fixed", 1), ("ou_random_root", 1), ("hetero", 16)])
def test_lgfill_random_networks(P, graph, which, p):
    """Random networks with hybrid nodes (2-parent families, several families per cluster in the clique tree): the
    device fill against the oracle's assignfactors!; on the clique tree also the likelihood (device fill + postorder
    + root integrate in one enqueue) against the oracle's traversal and the dense multivariate-normal likelihood."""
    import zlib
    rng = np.random.default_rng(zlib.crc32(f"{graph}-{which}-{p}".encode()))
    ntips, nhyb = (14, 3) if p == 16 else (24, 6)
    net = ON.random_network(ntips, nhyb, rng)
    model = _models(p, rng, net, which)
    taxa = net.tip_names
    tbl = [list(rng.normal(size=len(taxa))) for _ in range(p)]
    cg = OCG.cliquetree(net) if graph == "cliquetree" else OCG.bethe(net)
    if max(len(nodes) for _, nodes in cg.clusters) * p > 64:
        pytest.skip("cluster dimension above PGBP_MAX_DIM")
    ocgb, pcgb = _device_fill(P, net, cg, model, tbl, taxa)
    _assert_factors_equal(pcgb, ocgb)
    if graph == "cliquetree":
        spt = OCG.spanningtree_clusterlist(cg, OCG.default_rootcluster(cg, net))
        pcgb.set_schedule([spt])
        ll, info = pcgb.loglik_lg()
        assert not info.any()
        assert OC.propagate_1traversal_postorder(ocgb, *spt)
        oll = ocgb.integratebelief(spt[2][0])[1]
        assert abs(ll[0] - oll) <= 1e-8 * max(1.0, abs(oll)), (ll, oll)
        if which != "bm_improper_root":
            dense = OD.loglik(net, model, tbl, taxa)
            assert abs(ll[0] - dense) <= 1e-8 * max(1.0, abs(dense)), (ll, dense)
        # a second evaluation with other parameters on the same engine: only the parameters move
        model2 = _models(p, np.random.default_rng(7), net, which)
        if hasattr(model, "colors"):
            model2.colors = model.colors
        ocgb2 = oracle_setup(net, cg, model2, tbl, taxa)
        _, _, kw2 = lg_inputs_from_oracle(P, net, ocgb2, model2, tbl, taxa)
        pcgb.assignfactors_lg_(**kw2)
        ll2, _ = pcgb.loglik_lg()
        assert OC.propagate_1traversal_postorder(ocgb2, *spt)
        oll2 = ocgb2.integratebelief(spt[2][0])[1]
        assert abs(ll2[0] - oll2) <= 1e-8 * max(1.0, abs(oll2)), (ll2, oll2)


def _tree_as_network(S, tr):
    names = [f"n{i}" for i in range(tr.nnodes)]
    net = ON.read_newick(tr.newick(names))
    net.set_preorder(names)
    taxa = [names[i] for i in range(tr.nnodes) if tr.is_leaf[i]]
    return net, names, taxa


def test_lgfill_ou_sites_site_minor(P):
    """cfg4's likelihood evaluation entirely on the device: univariate OU with one (sigma2, alpha, theta, mu) and one
    data column per site, 70 sites (site-minor layout, lg_fill_uni_sm) and 5 sites (wavefront kernels): fill +
    postorder + root integrate against the dense multivariate-normal likelihood of every site, then a calibration
    from the device-filled factors against the oracle."""
    from pgbp_amd import synth as S
    rng = np.random.default_rng(44)
    tr = S.random_tree(25, rng)
    net, names, taxa = _tree_as_network(S, tr)
    prob = S.cliquetree_of_tree(tr, 1)
    clusters = [(str(i), [int(a), int(b)]) for i, (a, b) in enumerate(prob.cluster_nodes)]
    edges = [(int(a), int(b), [int(prob.sepset_nodes[k])]) for k, (a, b) in enumerate(prob.sepset_clusters)]
    cg = OB.ClusterGraph(clusters, edges, "cliquetree")
    spt = ([str(x) for x in prob.schedule[0][0]], [str(x) for x in prob.schedule[0][1]],
           prob.schedule[0][0].tolist(), prob.schedule[0][1].tolist())
    for ns in (70, 5):
        sig, al = rng.uniform(0.5, 2, ns), rng.uniform(0.1, 1, ns)
        th, mu = rng.normal(size=ns), rng.normal(size=ns)
        X = np.zeros((ns, tr.nnodes, 1))
        X[:, tr.is_leaf, 0] = rng.normal(size=(ns, tr.ntips))
        eng = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx,
                                               np.zeros((ns, int(prob.packed_off[-1]))), n_sites=ns)
        eng.set_schedule(prob.schedule)
        eng.lg_setup(S.lg_tree_table(tr, prob, 1), X)
        assert P.calibrate_(eng, prob.schedule, 1)[0]            # brings the state into the layout of the traversals
        gam2 = sig / (2 * al)
        eng.assignfactors_lg_(gam2.reshape(ns, 1, 1, 1), mu[:, None], model="ou", alpha=al, theta=th[:, None])
        ll, info = eng.loglik_lg(reps=2)
        assert not info.any()
        models = [OM.UnivariateOrnsteinUhlenbeck(sig[s], al[s], th[s], mu[s], 0.0) for s in range(ns)]
        ys = [[float(X[s, i, 0]) for i in range(tr.nnodes) if tr.is_leaf[i]] for s in range(ns)]
        for s in range(ns):
            dense = OD.loglik(net, models[s], [ys[s]], taxa)
            assert abs(ll[s] - dense) <= 1e-8 * max(1.0, abs(dense)), (ns, s, ll[s], dense)
        eng.assignfactors_lg_(gam2.reshape(ns, 1, 1, 1), mu[:, None], model="ou", alpha=al, theta=th[:, None])
        assert P.calibrate_(eng, prob.schedule, 2) == (True, True)
        for s in (0, ns - 1):
            ocgb = oracle_setup(net, cg, models[s], [ys[s]], taxa)
            assert OC.calibrate(ocgb, [spt], 2) == (True, True)
            from helpers import pack_oracle
            assert np.allclose(eng._packed[s], pack_oracle(ocgb, prob), rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize("graph,p,ns", [("cliquetree", 16, 1), ("bethe", 8, 2), ("cliquetree", 5, 3), ("cliquetree", 4, 1),
                                        ("cliquetree", 32, 1), ("bethe", 21, 2)])
def test_lgfill_hetero_tree_layouts(P, graph, p, ns):
    """Heterogeneous BM (3 painted rates) on a tree, factors filled on the device while the state is in the layout of
    the register-resident kernel (symmetric block-packed for even p): likelihood against the oracle's traversal on
    oracle-filled factors; per-site rates when ns > 1."""
    from pgbp_amd import synth as S
    rng = np.random.default_rng(900 + p)
    tr = S.random_tree(40, rng)
    net, names, taxa = _tree_as_network(S, tr)
    prob = S.cliquetree_of_tree(tr, p) if graph == "cliquetree" else S.bethe_of_tree(tr, p)
    col = rng.integers(0, 3, size=tr.nnodes)
    X = rng.normal(size=(ns, tr.nnodes, p))
    eng = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx,
                                           np.zeros((ns, int(prob.packed_off[-1]))), n_sites=ns)
    eng.set_schedule(prob.schedule)
    fam = S.lg_tree_table(tr, prob, p, colors=col)
    fam["n_rates"] = 3
    eng.lg_setup(fam, X)
    rates = np.zeros((ns, 3, p, p))
    for s in range(ns):
        R = S.random_rate_matrix(p, rng)
        R = (R + R.T) / 2
        rates[s] = [R * f for f in (0.5, 1.0, 2.0)]
    mus = rng.normal(size=(ns, p))
    for rep in range(2):   # rep 1: the state already sits in the traversal layout
        eng.assignfactors_lg_(rates if ns > 1 else rates[0], mus if ns > 1 else mus[0])
        ll, info = eng.loglik_lg()
        assert not info.any()
        assert P.calibrate_(eng, prob.schedule, 1)[0]
        ll_cal = eng.integratebelief_(prob.root_cluster, all_sites=True)[1]
    ct = OCG.cliquetree(net)
    spt = OCG.spanningtree_clusterlist(ct, OCG.default_rootcluster(ct, net))
    ecol = {e.number: 1 + int(col[int(e.child.name[1:])]) for e in net.edges}
    for s in range(ns):
        model = OM.HeterogeneousBrownianMotion(list(rates[s]), ecol, mus[s])
        tbl = [[float(X[s, int(t[1:]), v]) for t in taxa] for v in range(p)]
        ocgb = oracle_setup(net, ct, model, tbl, taxa)
        assert OC.propagate_1traversal_postorder(ocgb, *spt)
        oll = ocgb.integratebelief(spt[2][0])[1]
        assert abs(ll[s] - oll) <= 1e-8 * max(1.0, abs(oll)), (s, ll[s], oll)
        assert abs(ll_cal[s] - oll) <= 1e-8 * max(1.0, abs(oll))


def test_lgfill_refusals(P):
    """Malformed family tables and out-of-order calls are refused with the reference's kind of error, nothing runs."""
    from pgbp_amd import synth as S
    rng = np.random.default_rng(3)
    tr = S.random_tree(8, rng)
    prob = S.cliquetree_of_tree(tr, 2)
    eng = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx,
                                           np.zeros(int(prob.packed_off[-1])))
    X = rng.normal(size=(tr.nnodes, 2))
    with pytest.raises(AttributeError):
        eng.assignfactors_lg_(np.eye(2)[None], np.zeros(2))          # no lg_setup yet
    good = S.lg_tree_table(tr, prob, 2)
    for key, val, msg in (("child_pos", 3, "overlap or leave"), ("cluster", 10 ** 6, "cluster out of range"),
                          ("length", 0.0, "positive"), ("color", 5, "rate index")):
        bad = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in good.items()}
        i = int(np.nonzero(good["child_pos"] >= 0)[0][-1])
        bad[key][i] = val
        with pytest.raises(P.PgbpError, match=msg):
            eng.lg_setup(bad, X)
    Xnan = X.copy()
    Xnan[int(np.nonzero(tr.is_leaf)[0][0]), 1] = np.nan
    with pytest.raises(P.PgbpError, match="missing"):
        eng.lg_setup(good, Xnan)                                      # NaN without scope masks
    masked = dict(good, child_mask=np.full(len(good["cluster"]), 3, np.uint64),
                  parent_mask=np.full(len(good["cluster"]), 3, np.uint64))
    with pytest.raises(P.PgbpError, match="missing"):
        eng.lg_setup(masked, Xnan)                                    # NaN where the mask says observed
    masked["parent_mask"][int(np.nonzero(good["parent_pos"] >= 0)[0][0])] = 1
    with pytest.raises(P.PgbpError, match="out of the parent's scope|overlap or leave"):
        eng.lg_setup(masked, X)
    eng.lg_setup(good, X)
    eng.set_schedule(prob.schedule)
    with pytest.raises(P.PgbpError, match="pgbp_lg_assignfactors first"):
        eng.loglik_lg()                                               # schedule set? parameters missing


@pytest.mark.parametrize("graph,ntips,nblobs,p", [("bethe", 150, 12, 4), ("joingraph", 150, 12, 4), ("bethe", 60, 5, 2),
                                                  ("joingraph", 40, 3, 16), ("cliquetree", 120, 9, 4), ("ltrip", 100, 8, 3)])
def test_cfg5_pipeline_on_arrays(P, graph, ntips, nblobs, p):
    """The cfg5 pipeline of bench.py --workload network at test size, without any oracle object on the product side:
    level-3 network on plain arrays, Bethe / join-graph cluster graph, scope allocation, device factor fill for a
    heterogeneous BM with hybrid nodes, regularizebeliefs_bycluster!, calibrate!(auto) -- against the plain-C sequential
    engine from the same start (iteration / tree of convergence, every belief to 1e-8) and, for the exact (tree-shaped)
    join graph, the likelihood against the dense multivariate-normal value."""
    from oracle import cengine
    rng = np.random.default_rng(77 + ntips + p)
    net = P.random_level3_network(ntips, nblobs, rng, n_colors=3)
    assert net.nhybrids == 3 * nblobs
    cn, ed, sn = {"joingraph": lambda f: P.joingraph(f, 3), "bethe": P.bethe, "cliquetree": P.cliquetree,
                  "ltrip": P.ltrip}[graph](net.node2family)
    st = P.allocate_scopes(cn, ed, sn, net, p)
    base = np.eye(p) + 0.3
    rates = np.stack([base * f for f in (0.5, 1.0, 2.0)])
    mu = rng.normal(size=p)
    X = P.simulate_bm_network(net, rates, mu, rng)
    pe = [list(zip(net.length[i], net.gamma[i], net.color[i])) for i in range(net.nnodes)]
    fam = P.lg_families(st.clusters, st.node2cluster, net.node2family, st.node2fixed, pe, list(range(net.nnodes)), p, n_rates=3)
    cgb = P.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None)
    cgb.lg_setup(fam, X)
    cgb.assignfactors_lg_(rates, mu)
    loopy = len(ed) > len(cn) - 1
    assert loopy == (graph in ("bethe", "ltrip"))
    if loopy:
        assert P.load().pgbp_regularize_bycluster(cgb._eng) == 0
    cgb.pull()
    start = cgb._packed[0].copy()
    sched = P.spanningtrees_clusterlist(len(cn), ed, cn, net.is_leaf)
    assert P.calibrate_(cgb, sched, 100, auto=True) == (True, True)
    r = cgb.last_results[0]
    ce = cengine.Engine(st.dims, st.sepset_clusters.reshape(-1), st.scope_off, st.scope_idx, start)
    reached = None
    for it in range(1, 101):
        for j, spt in enumerate(sched, start=1):
            succ, iscal = ce.calibrate(spt[2], spt[3], 1, return_iscal=True)
            assert succ
            if iscal:
                reached = (it, j)
                break
        if reached:
            break
    assert reached == (r.iter_reached, r.tree_reached)
    ref = ce.packed()
    assert np.max(np.abs(cgb._packed[0] - ref)) <= 1e-8 * max(1.0, np.max(np.abs(ref)))
    if not loopy and p <= 4:
        # exact graph: the calibrated root cluster integrates to the likelihood
        nodes = [ON.Node(name=f"n{i + 1}", leaf=bool(net.is_leaf[i]), hybrid=len(net.node2family[i]) > 2) for i in range(net.nnodes)]
        edges = []
        for i, nf in enumerate(net.node2family):
            for k, pl in enumerate(nf[1:]):
                e = ON.Edge(number=len(edges) + 1, parent=nodes[pl - 1], child=nodes[i], length=net.length[i][k],
                            gamma=net.gamma[i][k], hybrid=len(nf) > 2)
                edges.append(e); nodes[pl - 1].edges.append(e); nodes[i].edges.append(e)
        onet = ON.Network(nodes[0], nodes, edges)
        onet.set_preorder([n.name for n in nodes])
        colors = {}
        for e in edges:
            ci = int(e.child.name[1:]) - 1
            colors[e.number] = 1 + net.color[ci][net.node2family[ci][1:].index(int(e.parent.name[1:]))]
        model = OM.HeterogeneousBrownianMotion(list(rates), colors, mu)
        taxa = onet.tip_names
        tbl = [[float(X[int(t[1:]) - 1, v]) for t in taxa] for v in range(p)]
        dense = OD.loglik(onet, model, tbl, taxa)
        ll = cgb.integratebelief_(sched[0][2][0])[1]
        assert abs(ll - dense) <= 1e-8 * max(1.0, abs(dense)), (ll, dense)


@pytest.mark.parametrize("graph", ["cliquetree", "bethe"])
@pytest.mark.parametrize("which,p", [("bm_fixed", 3), ("bm_random_root", 2), ("hetero", 4), ("bm_improper_root", 3),
                                     ("hetero", 16)])
def test_lgfill_missing_data_random_networks(P, graph, which, p):
    """Missing tip values on random networks (scope masks of pgbp_lg_families): device fill == the oracle's assignfactors!,
    and on the clique tree the likelihood == the oracle's traversal."""
    import zlib
    from test_lg_families_cpu import missing_pattern
    rng = np.random.default_rng(zlib.crc32(f"gpu-miss-{graph}-{which}-{p}".encode()))
    net = ON.random_network(14 if p == 16 else 20, 3 if p == 16 else 4, rng)
    model = _models(p, rng, net, which)
    tbl, taxa = missing_pattern(net, p, rng, which)
    cg = OCG.cliquetree(net) if graph == "cliquetree" else OCG.bethe(net)
    if max(len(nodes) for _, nodes in cg.clusters) * p > 64:
        pytest.skip("cluster dimension above PGBP_MAX_DIM")
    ocgb, pcgb = _device_fill(P, net, cg, model, tbl, taxa)
    _assert_factors_equal(pcgb, ocgb)
    if graph == "cliquetree":
        spt = OCG.spanningtree_clusterlist(cg, OCG.default_rootcluster(cg, net))
        pcgb.set_schedule([spt])
        ll, info = pcgb.loglik_lg()
        assert not info.any()
        assert OC.propagate_1traversal_postorder(ocgb, *spt)
        oll = ocgb.integratebelief(spt[2][0])[1]
        assert abs(ll[0] - oll) <= 1e-8 * max(1.0, abs(oll)), (ll, oll)


@pytest.mark.parametrize("variant", ["improper", "fixed"])
def test_lgfill_partial_internal_scopes_level3_golden(P, variant):
    """test/test_calibration.jl:131-185 with the factors assigned on the device: y2 missing at B leaves hybrid nodes with
    one of two traits in scope; device fill == oracle fill, and the calibrated clique tree gives the golden
    normalisation constant at every belief."""
    g = G["calibration_level3_joingraph"]
    net = ON.read_newick(g["net"])
    net.set_preorder(g["preorder"])
    ct = OCG.cliquetree(net)
    spt = OCG.spanningtree_clusterlist(ct, OCG.default_rootcluster(ct, net))
    ocgb, pcgb = _device_fill(P, net, ct, make_model(g["model_" + variant]), [g["y1"], g["y2"]], g["taxa"])
    assert any(b.dimension % 2 for b in ocgb.belief[:ocgb.nclusters])
    _assert_factors_equal(pcgb, ocgb)
    assert P.calibrate_(pcgb, [spt])[0]
    for i, be in enumerate(ocgb.belief):
        if be.dimension:
            norm = pcgb.integratebelief_(i)[1]
            assert abs(norm - g["norm_" + variant]) <= 1e-9 * abs(g["norm_" + variant])


def test_getting_started_pipeline_product_only(P):
    """docs/src/man/getting_started.md:30-292 with the product's own host side from the Newick string to the likelihood --
    no oracle object anywhere: read_newick, clique tree (17 clusters, 16 sepsets as the doctest prints), scope allocation,
    spanning-tree schedule, factors assigned on the device (UnivariateBrownianMotion(1, 0)), calibrate!, and
    integratebelief! at every belief = the doctest's log-likelihood; factored_energy = the same value."""
    g = G["doctest_lazaridis"]
    net, names = P.read_newick(g["net"])
    assert sorted(n for n, leaf in zip(names, net.is_leaf) if leaf) == sorted(g["taxa"])
    cn, ed, sn = P.cliquetree(net.node2family)
    assert (len(cn), len(ed)) == (g["nclusters"], g["nsepsets"])
    st = P.allocate_scopes(cn, ed, sn, net, 1)
    row = {t: r for r, t in enumerate(g["taxa"])}
    data_row = [row.get(names[i], -1) for i in range(net.nnodes)]
    pe = [list(zip(net.length[i], net.gamma[i], net.color[i])) for i in range(net.nnodes)]
    fam = P.lg_families(st.clusters, st.node2cluster, net.node2family, st.node2fixed, pe, data_row, 1)
    cgb = P.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None)
    cgb.lg_setup(fam, np.array(g["x"], float)[:, None])
    cgb.assignfactors_lg_(np.array([[[g["model"]["sigma2"]]]], float), [g["model"]["mu"]])
    sched = [P.spanningtree_clusterlist(len(cn), ed, P.default_rootcluster(cn, net.is_leaf))]
    assert P.calibrate_(cgb, sched) == (True, False)      # one iteration: exact, not yet flagged as calibrated
    assert P.calibrate_(cgb, sched, 5, auto=True) == (True, True)
    for i in range(len(st.dims)):
        if st.dims[i]:
            assert abs(cgb.integratebelief_(i, all_sites=True)[1][0] - g["ll"]) <= 1e-9 * abs(g["ll"])
    assert abs(cgb.factored_energy()[2] - g["factored_energy"]) <= 1e-9 * abs(g["factored_energy"])


def test_lgfill_differential_fuzz(P):
    """A slice of tests/fuzz_lgfill_vs_oracle.py in the suite: 120 random (network, cluster graph, model, trait count,
    missing-value pattern) cases, device factor fill against the oracle's assignfactors! (1e-10) and, on exact graphs,
    the likelihood against the oracle's traversal (1e-8)."""
    import fuzz_lgfill_vs_oracle as F
    n_missing, worst = F.run(120, 2025)
    assert n_missing >= 20 and worst <= 1e-10


def _mateescu_on_device(P, graph):
    g = G["optimization_mateescu"]
    net, names = P.read_newick(G["joingraph_mateescu"]["net"])
    cn, ed, sn = P.cliquetree(net.node2family) if graph == "cliquetree" else P.bethe(net.node2family)
    st = P.allocate_scopes(cn, ed, sn, net, 1)
    row = {t: r for r, t in enumerate(g["taxa"])}
    fam = P.lg_families(st.clusters, st.node2cluster, net.node2family, st.node2fixed,
                        [list(zip(net.length[i], net.gamma[i], net.color[i])) for i in range(net.nnodes)],
                        [row.get(names[i], -1) for i in range(net.nnodes)], 1)
    cgb = P.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None)
    cgb.lg_setup(fam, np.array(g["y"], float)[:, None])
    return g, net, (cn, ed, sn), cgb


def test_calibrate_optimize_cliquetree_golden(P):
    """test/test_optimization.jl:5-26: maximum-likelihood (sigma2, mu) of a univariate BM on the Mateescu network (level 4,
    hybrid ladder, a hybrid with a hybrid parent), every likelihood evaluation on the device: the reference's estimates
    and maximised log-likelihood."""
    g, net, (cn, ed, sn), cgb = _mateescu_on_device(P, "cliquetree")
    spt = P.spanningtree_clusterlist(len(cn), ed, P.default_rootcluster(cn, net.is_leaf))
    R, mu, ll, opt = P.calibrate_optimize_cliquetree_(cgb, spt, [[g["start"]["sigma2"]]], [g["start"]["mu"]])
    assert abs(ll - g["ref_ll"]) <= 1e-12 * abs(g["ref_ll"])                  # measured 3e-16
    assert abs(R[0, 0] - g["ref_sigma2"]) <= 1e-7 * g["ref_sigma2"]           # measured 2e-10 (the reference's autodiff
    assert abs(mu[0] - g["ref_mu"]) <= 1e-7 * abs(g["ref_mu"])                # cross-check allows 3e-11 / 4e-10)


def test_calibrate_optimize_clustergraph_golden(P):
    """test/test_optimization.jl:39-48: the same estimation through the Bethe cluster graph (factored energy maximised,
    regularizebeliefs_bycluster! + calibrate!(auto) per evaluation, all on the device), within the reference's tolerances
    of the clique-tree estimates."""
    g, net, (cn, ed, sn), cgb = _mateescu_on_device(P, "bethe")
    sched = P.spanningtrees_clusterlist(len(cn), ed, cn, net.is_leaf)
    R, mu, fe, opt = P.calibrate_optimize_clustergraph_(cgb, sched, [[g["start"]["sigma2"]]], [g["start"]["mu"]])
    tol = g["bethe_rtol"]
    assert abs(mu[0] - g["ref_mu"]) <= tol["mu"] * abs(g["ref_mu"])           # the reference's own tolerances (:46-48)
    assert abs(R[0, 0] - g["ref_sigma2"]) <= tol["sigma2"] * g["ref_sigma2"]
    assert abs(fe - g["ref_ll"]) <= tol["fenergy"] * abs(g["ref_ll"])


def _on_device_from_newick(P, netstr, taxa, columns, graph):
    """columns: list of per-trait value lists (None = missing), rows ordered as `taxa`."""
    net, names = P.read_newick(netstr)
    p = len(columns)
    data = np.array([[np.nan if columns[v][r] is None else float(columns[v][r]) for v in range(p)] for r in range(len(taxa))])
    build = {"cliquetree": P.cliquetree, "bethe": P.bethe}[graph]
    cn, ed, sn = build(net.node2family)
    # scopes with missing data: a node keeps a trait iff some tip below it has it (src/beliefs.jl:509-520, 551-559)
    row = {t: r for r, t in enumerate(taxa)}
    has = np.zeros((net.nnodes, p), bool)
    for i in range(net.nnodes - 1, -1, -1):
        if net.is_leaf[i]:
            has[i] = np.isfinite(data[row[names[i]]])
        for pa in net.node2family[i][1:]:
            has[pa - 1] |= has[i]
    assert has[~net.is_leaf].all()          # these cases keep every internal scope full
    st = P.allocate_scopes(cn, ed, sn, net, p)
    fam = P.lg_families(st.clusters, st.node2cluster, net.node2family, st.node2fixed,
                        [list(zip(net.length[i], net.gamma[i], net.color[i])) for i in range(net.nnodes)],
                        [row.get(names[i], -1) for i in range(net.nnodes)], p, data=data)
    cgb = P.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None)
    cgb.lg_setup(fam, data)
    return net, (cn, ed, sn), cgb


def test_calibrate_optimize_level1_goldens(P):
    """test/test_calibration.jl:187-305: (1) univariate BM through the Bethe graph of a level-1 network, factored energy
    maximised: the values the reference checks against RxInfer (rtol 1e-4); (2) clique tree, one trait: the analytic ML
    values; (3) two independent traits (MvDiagBrownianMotion), one value missing: log-likelihood, means and rates."""
    g = G["optimization_level1"]
    b = g["bethe"]
    net, (cn, ed, sn), cgb = _on_device_from_newick(P, b["net"], b["taxa"], [b["y"]], "bethe")
    sched = P.spanningtrees_clusterlist(len(cn), ed, cn, net.is_leaf)
    R, mu, fe, _ = P.calibrate_optimize_clustergraph_(cgb, sched, [[b["start"]["sigma2"]]], [b["start"]["mu"]])
    assert abs(fe - b["fenergy"]) <= b["rtol"] * abs(b["fenergy"])
    assert abs(mu[0] - b["mu"]) <= b["rtol"] * abs(b["mu"]) and abs(R[0, 0] - b["sigma2"]) <= b["rtol"] * b["sigma2"]
    c = g["cliquetree"]
    net, (cn, ed, sn), cgb = _on_device_from_newick(P, c["net"], c["taxa"], [c["y"]], "cliquetree")
    spt = P.spanningtree_clusterlist(len(cn), ed, P.default_rootcluster(cn, net.is_leaf))
    R, mu, ll, _ = P.calibrate_optimize_cliquetree_(cgb, spt, [[c["start_y"]["sigma2"]]], [c["start_y"]["mu"]])
    assert abs(ll - c["ll_y"]) <= 1e-10 * abs(c["ll_y"])
    assert abs(mu[0] - c["mu_y"]) <= 1e-6 * abs(c["mu_y"]) and abs(R[0, 0] - c["sigma2_y"]) <= 1e-6 * c["sigma2_y"]
    net, (cn, ed, sn), cgb = _on_device_from_newick(P, c["net"], c["taxa"], [c["x"], c["y"]], "cliquetree")
    spt = P.spanningtree_clusterlist(len(cn), ed, P.default_rootcluster(cn, net.is_leaf))
    R, mu, ll, _ = P.calibrate_optimize_cliquetree_(cgb, spt, np.diag(c["start_xy"]["R"]), c["start_xy"]["mu"], diagonal=True)
    assert abs(ll - c["ll_xy"]) <= 1e-9 * abs(c["ll_xy"])
    assert np.allclose(mu, c["mu_xy"], rtol=1e-5, atol=0) and np.allclose(np.diag(R), c["sigma2_xy"], rtol=1e-5, atol=0)


def test_calibrate_optimize_sun2023_bivariate_improper_root(P):
    """test/test_optimization.jl:52-100: full 2 x 2 rate matrix of a bivariate BM with an improper root prior on the
    level-6 network of Sun et al. (clique tree, clusters of up to 5 nodes = 10 variables): the maximised log-likelihood the
    reference records (its L-BFGS ran 1000 iterations, 3180 evaluations, 248 s), and its rate matrix up to the rescaling of
    the file's edge lengths noted in the golden."""
    g = G["optimization_sun2023"]
    net, names = P.read_newick(g["net"])
    assert (net.nnodes, int(net.is_leaf.sum()), net.nhybrids) == (42, 10, 6)
    assert [n for n, leaf in sorted(zip(names, net.is_leaf), key=lambda t: g["taxa_in_file_order"].index(t[0]) if t[1] else -1)
            if leaf] == g["taxa_in_file_order"]
    cn, ed, sn = P.cliquetree(net.node2family)
    assert (min(len(c) for c in cn), max(len(c) for c in cn)) == (2, 5)
    st = P.allocate_scopes(cn, ed, sn, net, 2, fixedroot=False)
    row = {t: r for r, t in enumerate(g["taxa_in_file_order"])}
    fam = P.lg_families(st.clusters, st.node2cluster, net.node2family, st.node2fixed,
                        [list(zip(net.length[i], net.gamma[i], net.color[i])) for i in range(net.nnodes)],
                        [row.get(names[i], -1) for i in range(net.nnodes)], 2)
    cgb = P.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None)
    cgb.lg_setup(fam, np.stack([g["y1"], g["y2"]], axis=1))
    spt = P.spanningtree_clusterlist(len(cn), ed, P.default_rootcluster(cn, net.is_leaf))
    R, mu, ll, opt = P.calibrate_optimize_cliquetree_(cgb, spt, g["start_R"], [0.0, 0.0], maxiter=500)
    assert abs(ll - g["ll_max"]) <= 1e-9 * abs(g["ll_max"]), (ll, opt.nfev)
    assert np.allclose(R * g["R_scale"], g["R_recorded"], rtol=1e-4, atol=0), R


@pytest.mark.parametrize("traits", ["uni", "bi"])
def test_exact_reml_estimates_through_the_device_objective(P, traits):
    """test/test_exactBM.jl:185-226: the REML estimates calibrate_exact_cliquetree! gets in closed form -- the rate (matrix)
    that maximises the likelihood under the improper root prior, the root's posterior mean there, and (one trait) the
    maximised value -- obtained here by maximising the device-computed likelihood."""
    g = G["exact_reml_level1"]
    cols = [g["y"]] if traits == "uni" else [g["x"], g["y"]]
    p = len(cols)
    net, names = P.read_newick(g["net"])
    cn, ed, sn = P.cliquetree(net.node2family)
    st = P.allocate_scopes(cn, ed, sn, net, p, fixedroot=False)
    row = {t: r for r, t in enumerate(g["taxa"])}
    fam = P.lg_families(st.clusters, st.node2cluster, net.node2family, st.node2fixed,
                        [list(zip(net.length[i], net.gamma[i], net.color[i])) for i in range(net.nnodes)],
                        [row.get(names[i], -1) for i in range(net.nnodes)], p)
    cgb = P.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None)
    cgb.lg_setup(fam, np.array(cols, float).T.copy())
    spt = P.spanningtree_clusterlist(len(cn), ed, P.default_rootcluster(cn, net.is_leaf))
    R, _, ll, _ = P.calibrate_optimize_cliquetree_(cgb, spt, np.eye(p), np.zeros(p))
    want = g[traits]
    if traits == "uni":
        restricted = G["calibration_cliquetree_level1"]["ll_every_belief"]      # phylolm's restricted likelihood (:41-46)
        assert abs(ll - restricted) <= 1e-10 * abs(restricted)
        assert abs(R[0, 0] - want["sigma2"]) <= 1e-6 * want["sigma2"]
    else:
        assert np.allclose(R, want["R"], rtol=1e-5, atol=0)
    # the root's posterior mean at the estimate: calibrate, integrate a cluster that holds the root (label 1: listed last)
    cgb.assignfactors_lg_(np.stack([R]), np.zeros(p))
    assert P.calibrate_(cgb, [spt])[0]
    ci = next(i for i, c in enumerate(cn) if 1 in c)
    mu_root = cgb.integratebelief_(ci)[0][-p:]
    assert np.allclose(mu_root, want["mu"], rtol=1e-8, atol=0)
    if traits == "uni":
        # the score calibrate_exact_cliquetree! returns: the likelihood of the model it returns, root FIXED at the estimated
        # mean with the REML rate
        st2 = P.allocate_scopes(cn, ed, sn, net, p, fixedroot=True)
        fam2 = P.lg_families(st2.clusters, st2.node2cluster, net.node2family, st2.node2fixed,
                             [list(zip(net.length[i], net.gamma[i], net.color[i])) for i in range(net.nnodes)],
                             [row.get(names[i], -1) for i in range(net.nnodes)], p)
        cgb2 = P.ClusterGraphBelief.from_arrays(st2.dims, st2.sepset_clusters, st2.scope_off, st2.scope_idx, None)
        cgb2.lg_setup(fam2, np.array(cols, float).T.copy())
        cgb2.set_schedule([spt])
        cgb2.assignfactors_lg_(np.array([[[want["sigma2"]]]]), [want["mu"]])
        ll2, info = cgb2.loglik_lg()
        assert not info.any() and abs(ll2[0] - want["ll"]) <= 1e-10 * abs(want["ll"])


def test_exact_reml_with_fully_missing_sisters(P):
    """test/test_exactBM.jl:228-251: the trait is missing at two sister tips, so their parent has nothing in scope (a
    zero-dimensional block, families that integrate to 1): scopes from allocate_scopes(data=...), factors from the masked
    device fill; REML rate, root posterior mean, and the score of the returned fixed-root model."""
    g = G["exact_reml_missing"]
    net, names = P.read_newick(g["net"])
    row = {t: r for r, t in enumerate(g["taxa"])}
    data_row = [row.get(names[i], -1) for i in range(net.nnodes)]
    data = np.array([[np.nan if v is None else float(v)] for v in g["x"]])
    cn, ed, sn = P.cliquetree(net.node2family)
    pe = [list(zip(net.length[i], net.gamma[i], net.color[i])) for i in range(net.nnodes)]
    engines = {}
    for fixedroot in (False, True):
        st = P.allocate_scopes(cn, ed, sn, net, 1, fixedroot=fixedroot, data=data, data_row=data_row)
        fam = P.lg_families(st.clusters, st.node2cluster, net.node2family, st.node2fixed, pe, data_row, 1, data=data)
        cgb = P.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None)
        cgb.lg_setup(fam, data)
        engines[fixedroot] = (st, cgb)
    st, cgb = engines[False]
    assert 0 in st.dims[:len(cn)].tolist() or any(d == 1 for d in st.dims[:len(cn)])      # reduced scopes are there
    spt = P.spanningtree_clusterlist(len(cn), ed, P.default_rootcluster(cn, net.is_leaf))
    R, _, ll, _ = P.calibrate_optimize_cliquetree_(cgb, spt, [[1.0]], [0.0])
    assert abs(R[0, 0] - g["sigma2"]) <= 1e-6 * g["sigma2"]
    cgb.assignfactors_lg_(np.stack([R]), [0.0])
    assert P.calibrate_(cgb, [spt])[0]
    ci = next(i for i, c in enumerate(cn) if 1 in c and st.dims[i] > 0)
    assert abs(cgb.integratebelief_(ci)[0][-1] - g["mu"]) <= 1e-8 * abs(g["mu"])
    st2, cgb2 = engines[True]
    cgb2.set_schedule([spt])
    cgb2.assignfactors_lg_(np.array([[[g["sigma2"]]]]), [g["mu"]])
    ll2, info = cgb2.loglik_lg()
    assert not info.any() and abs(ll2[0] - g["ll"]) <= 1e-10 * abs(g["ll"])
