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


COMPLETE = [c for c in G["evomodels_postorder"]["cases"]
            if all(v is not None for t in c["traits"] for v in G["evomodels_postorder"][t])]


@pytest.mark.parametrize("case", COMPLETE, ids=lambda c: c["name"])
def test_lgfill_evomodels_goldens(P, case):
    """test/test_evomodels.jl:74-264 (the cases without missing data): factors filled on the device, postorder,
    integratebelief! at the root cluster = the golden log-likelihood."""
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
