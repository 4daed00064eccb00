"""
GPU parity tests (pytest -m gpu): the HIP engine, called through the C ABI
(include/pgbp.h via the ctypes host mirror), against the CPU oracle on the same
inputs and against the reference's golden values.

Tolerance (BASELINE.json north_star: "log-likelihood matching to 1e-8"): relative,
RTOL = 1e-8 on log-likelihoods; calibrated (J, h, g) within 1e-8 * max(1, |.|_inf)
per belief (SURVEY.md section 8(d) parity gate).
"""
import logging

import numpy as np
import pytest

from helpers import (goldens, make_model, oracle_cgb_from_problem, oracle_schedule, oracle_setup, pack_oracle,
                     product_beliefs_from_oracle)
from oracle import beliefs as OB
from oracle import calibration as OC
from oracle import clustergraph as OCG
from oracle import models as OM
from oracle import network as ON

pytestmark = pytest.mark.gpu

RTOL = 1e-8
G = goldens()


@pytest.fixture(scope="module")
def P():
    import pgbp_amd
    pgbp_amd.load()
    return pgbp_amd


def rel_close(a, b, rtol=RTOL):
    return abs(a - b) <= rtol * max(1.0, abs(a), abs(b))


def assert_beliefs_close(prod, orac, rtol=RTOL):
    """prod: product ClusterGraphBelief (pulled); orac: oracle ClusterGraphBelief."""
    for i, ob in enumerate(orac.belief):
        pb = prod.belief[i]
        for name, x, y in (("J", pb.J, ob.J), ("h", pb.h, ob.h), ("g", pb.g, ob.g)):
            x, y = np.asarray(x), np.asarray(y)
            if x.size == 0:
                continue
            scale = max(1.0, float(np.max(np.abs(y))))
            err = float(np.max(np.abs(x - y)))
            assert err <= rtol * scale, f"belief {i} {name}: err {err:.3e} scale {scale:.3e}"


def build_both(P, net, cg, model, tbl, taxa):
    ocgb = oracle_setup(net, cg, model, tbl, taxa)
    pb = product_beliefs_from_oracle(ocgb.belief)
    pcgb = P.ClusterGraphBelief(pb, ocgb.node2cluster, ocgb.node2family, ocgb.node2fixed, ocgb.cluster2nodes)
    return ocgb, pcgb


# ----------------------------------------------------------------------------- goldens

def test_canonicalform_six_messages(P):
    """test/test_canonicalform.jl:100-109 through pgbp_propagate."""
    g = G["canonicalform_six_messages"]
    net = ON.read_newick(g["net"])
    net.set_preorder(g["preorder"])
    names = [n.name for n in net.vec_node]
    clusters = [("".join(names[i - 1] for i in nl), nl) for nl in g["cluster_nodelabels"]]
    cg = OB.ClusterGraph(clusters, [tuple(e) for e in g["sepsets"]], "cliquetree")
    ocgb, pcgb = build_both(P, net, cg, make_model(g["model"]), [g["y"]], g["taxa"])
    for (to, sep, frm) in g["messages_1based"]:
        assert P.propagate_belief_(pcgb, to - 1, sep - 1, frm - 1) is None
        ob = ocgb.belief
        assert OB.propagate_belief(ob[to - 1], ob[sep - 1], ob[frm - 1], OB.MessageResidual(ob[sep - 1].dimension)) is None
        assert_beliefs_close(pcgb, ocgb)
    mu, ll = P.integratebelief_(pcgb, g["root_belief_1based"] - 1)
    assert rel_close(ll, g["ll"])
    omu, oll = ocgb.integratebelief(g["root_belief_1based"] - 1)
    assert np.allclose(mu, omu, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("case", G["evomodels_postorder"]["cases"], ids=lambda c: c["name"])
def test_evomodels_postorder_ll(P, case):
    """test/test_evomodels.jl:74-264: postorder traversal + integratebelief! at the root cluster."""
    g = G["evomodels_postorder"]
    net = ON.read_newick(g["net"])
    tbl = [g[t] for t in case["traits"]]
    ct = OCG.cliquetree(net)
    spt = OCG.spanningtree_clusterlist(ct, OCG.default_rootcluster(ct, net))
    ocgb, pcgb = build_both(P, net, ct, make_model(case["model"]), tbl, g["taxa"])
    assert P.propagate_1traversal_postorder_(pcgb, *spt)
    _, ll = P.integratebelief_(pcgb, spt[2][0])
    assert rel_close(ll, case["ll"]), (ll, case["ll"])
    assert OC.propagate_1traversal_postorder(ocgb, *spt)
    assert_beliefs_close(pcgb, ocgb)


def _calibrate_every_belief(P, g, tbl, ll, atol=None):
    net = ON.read_newick(g["net"])
    ct = OCG.cliquetree(net)
    spt = OCG.spanningtree_clusterlist(ct, OCG.default_rootcluster(ct, net))
    ocgb, pcgb = build_both(P, net, ct, make_model(g["model"]), tbl, g["taxa"])
    succ, iscal = P.calibrate_(pcgb, [spt])
    osucc, oiscal = OC.calibrate(ocgb, [spt])
    assert succ and osucc and iscal == oiscal
    assert_beliefs_close(pcgb, ocgb)
    for i in range(len(ocgb.belief)):
        mu, n = pcgb.integratebelief_(i)
        omu, on = ocgb.integratebelief(i)
        if atol is None:
            assert rel_close(n, ll), (i, n, ll)
        else:
            assert abs(n - ll) <= atol
        assert np.allclose(mu, omu, rtol=1e-8, atol=1e-8, equal_nan=True)
    # residuals and calibration flags of every directed message
    for key, omr in ocgb.messageresidual.items():
        pmr = pcgb.messageresidual[key]
        assert np.allclose(pmr.dJ, omr.dJ, rtol=1e-8, atol=1e-9)
        assert np.allclose(pmr.dh, omr.dh, rtol=1e-8, atol=1e-9)
        assert pmr.iscalibrated_resid == omr.iscalibrated_resid, key
    return net, ct, spt, ocgb, pcgb


def test_exactBM_tree_calibrate(P):
    g = G["exactBM_tree_calibrate"]
    _calibrate_every_belief(P, g, [g["y"]], g["ll"], atol=g["atol"])


def test_calibration_cliquetree_level1(P):
    g = G["calibration_cliquetree_level1"]
    net, ct, spt, ocgb, pcgb = _calibrate_every_belief(P, g, [g["y"]], g["ll_every_belief"])
    root_ind = next(i for i, be in enumerate(ocgb.belief) if 1 in be.nodelabel)
    mu, _ = pcgb.integratebelief_(root_ind)
    assert abs(mu[-1] - g["posterior_root_mean"]) <= g["rtol_posterior"] * abs(g["posterior_root_mean"])
    # init_beliefs_reset_fromfactors! then calibrate again (test/test_calibration.jl:65-76)
    pcgb.init_beliefs_reset_fromfactors_()
    ocgb.init_beliefs_reset_fromfactors()
    assert_beliefs_close(pcgb, ocgb, rtol=0.0)
    assert P.calibrate_(pcgb, [spt])[0]
    assert rel_close(pcgb.integratebelief_(0)[1], g["ll_every_belief"])


def test_calibration_tree_2traits_missing(P):
    """ragged scopes, a dimension-0 sepset, and the all-zero ("fake") marginalisation exit."""
    g = G["calibration_tree_2traits_missing"]
    _calibrate_every_belief(P, g, [g["y1"], g["y2"]], g["ll_every_belief"])


def test_doctest_lazaridis(P):
    g = G["doctest_lazaridis"]
    _calibrate_every_belief(P, g, [g["x"]], g["ll"])


# ----------------------------------------------------------------------------- failure semantics

def test_bpposdef_returned_not_thrown(P, caplog):
    """src/beliefupdates.jl:640-644, src/calibration.jl:129-132: a non-PD block is reported with the
    reference's message text, nothing of that message is applied, calibrate! returns (false, false)."""
    frm = P.CanonicalBelief([2, 1], 1, np.ones((1, 2), bool), P.bclustertype, "c21")
    to = P.CanonicalBelief([3, 2], 1, np.ones((1, 2), bool), P.bclustertype, "c32")
    sep = P.CanonicalBelief([2], 1, np.ones((1, 1), bool), P.bsepsettype, ("c32", "c21"))
    frm.J[:] = [[1.0, 0.2], [0.2, -1.0]]
    to.J[:] = [[2.0, 0.0], [0.0, 2.0]]
    cgb = P.ClusterGraphBelief([frm, to, sep])
    flag = P.propagate_belief_(cgb, 1, 2, 0)
    assert isinstance(flag, P.BPPosDefException) and flag.info == 1
    assert flag.msg == "belief c21, integrating [2]"
    assert flag.showerror() == "BPPosDefException: belief c21, integrating [2]\nmatrix is not positive definite."
    assert np.array_equal(cgb.belief[1].J, [[2.0, 0.0], [0.0, 2.0]]) and not cgb.belief[2].J.any()
    with pytest.raises(P.BPPosDefException):
        P.propagate_belief_(cgb, 1, 2, 0, withresidual=False)
    spt = (["c32"], ["c21"], [1], [0])
    with caplog.at_level(logging.INFO, logger="PhyloGaussianBeliefProp"):
        assert P.calibrate_(cgb, [spt], 3, info=True) == (False, False)
    assert "belief c21, integrating [2]" in caplog.text
    assert "propagation failed: iteration 1, schedule tree 1" in caplog.text
    r = cgb.last_results[0]
    assert (r.fail_iter, r.fail_tree, r.fail_dir, r.fail_edge, r.fail_info) == (1, 1, 0, 0, 1)
    # the other direction works: marginalising c32 (J = 2I) onto node 2
    assert P.propagate_belief_(cgb, 0, 2, 1) is None


def test_first_failure_is_the_reference_order_first(P):
    """Two failing leaf messages in the same level: the reported one is the first in the reference's
    sequential order (postorder = decreasing edge index, src/calibration.jl:121)."""
    from pgbp_amd import synth as S
    rng = np.random.default_rng(11)
    tr = S.random_tree(40, rng)
    p = 2
    prob = S.cliquetree_of_tree(tr, p)
    R = S.random_rate_matrix(p, rng)
    mu = np.zeros(p)
    X = S.simulate_bm(tr, R, mu, rng)
    packed = S.bm_factors_cliquetree(tr, prob, R, mu, X)
    pa, ch = prob.schedule[0]
    # break two internal cliques whose messages integrate something (dim 2p senders)
    senders = [i for i in range(len(pa)) if prob.dims[ch[i]] == 2 * p]
    bad = [senders[2], senders[-3]]
    for i in bad:
        o = prob.packed_off[ch[i]]
        packed[o] = -1.0e6   # J[0,0] << 0 even after the children's messages: first pivot fails, info = 1
    cgb = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packed)
    assert P.calibrate_(cgb, prob.schedule, verbose=False) == (False, False)
    r = cgb.last_results[0]
    assert r.fail_dir == 0 and r.fail_info == 1 and r.fail_edge == max(bad)


def test_a_failed_run_leaves_no_marks_behind(P):
    """The downstream-of-a-failure marks (one word per site and cluster) are cleared at the next enqueue call only where a
    fail word holds a failure (round 4: a 1.3 GB memset per call at cfg4's size otherwise).  One engine, two sites: a good
    run, a run with a non-positive-definite block in site 1 (its failure reported, site 0 untouched), then the good state
    again -- both sites must calibrate to the plain-C engine's beliefs, i.e. no mark of the failed run survives --, and a
    log-likelihood style postorder after that as well."""
    from oracle import cengine
    from pgbp_amd import synth as S
    rng = np.random.default_rng(2027)
    tr = S.random_tree(70, rng)
    p = 4
    prob = S.cliquetree_of_tree(tr, p)
    R = S.random_rate_matrix(p, rng)
    mu = np.zeros(p)
    packs = [S.bm_factors_cliquetree(tr, prob, R, mu, S.simulate_bm(tr, R, mu, rng)) for _ in range(2)]
    good = np.stack(packs)
    pa, ch = prob.schedule[0]
    cgb = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, good.copy(), n_sites=2)

    def check_good():
        assert P.calibrate_(cgb, prob.schedule, 1, verbose=False)[0]
        for s_ in range(2):
            ref = cengine.Engine(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packs[s_])
            assert ref.calibrate(pa, ch, 1, return_iscal=True)[0] and cgb.last_results[s_].succ == 1
            want = ref.packed()
            assert np.max(np.abs(cgb._packed[s_] - want)) <= 1e-8 * max(1.0, np.max(np.abs(want))), s_
    check_good()
    bad = good.copy()
    snd = int(next(c for c in ch if prob.dims[c] == 2 * p))
    bad[1][prob.packed_off[snd]] = -1.0e6
    cgb._packed[...] = bad
    cgb.push()
    cgb.init_messagecalibrationflags_reset_()
    P.calibrate_(cgb, prob.schedule, 1, verbose=False)
    assert (cgb.last_results[0].succ, cgb.last_results[1].succ) == (1, 0)
    for _ in range(2):   # (the second time nothing failed in the run before: the marks are left alone, and must be clean)
        cgb._packed[...] = good
        cgb.push()
        cgb.init_messagecalibrationflags_reset_()
        check_good()


def test_auto_stop_and_info_log(P, caplog):
    """calibrate!(...; auto=true, info=true): on a clique tree calibration is reached at iteration 2
    (the second pass changes nothing), and the log line is the reference's (src/calibration.jl:54)."""
    g = G["calibration_cliquetree_level1"]
    net = ON.read_newick(g["net"])
    ct = OCG.cliquetree(net)
    spt = OCG.spanningtree_clusterlist(ct, OCG.default_rootcluster(ct, net))
    ocgb, pcgb = build_both(P, net, ct, make_model(g["model"]), [g["y"]], g["taxa"])
    olog = []
    ores = OC.calibrate(ocgb, [spt], 5, auto=True, info=True, log=olog)
    with caplog.at_level(logging.INFO, logger="PhyloGaussianBeliefProp"):
        pres = P.calibrate_(pcgb, [spt], 5, auto=True, info=True)
    assert pres == ores == (True, True)
    assert olog[-1][1] in caplog.text
    assert_beliefs_close(pcgb, ocgb)


# ----------------------------------------------------------------------------- random trees vs the oracle

@pytest.mark.parametrize("ntips,p,kind", [(2, 1, "random"), (3, 1, "random"), (9, 1, "random"), (33, 2, "random"),
                                          (64, 3, "random"), (100, 8, "random"), (60, 16, "random"),
                                          (24, 16, "caterpillar"), (17, 32, "random"),
                                          (2, 16, "random"), (3, 16, "random"), (150, 16, "random"),
                                          (80, 16, "poly3"), (120, 16, "poly4"), (90, 16, "poly7"), (40, 3, "poly5"),
                                          (130, 8, "random"), (70, 8, "poly4"), (90, 4, "random"), (2, 8, "random"),
                                          (3, 4, "random"), (40, 8, "caterpillar"),
                                          # every even trait count <= 16 has a register-resident instance
                                          (50, 2, "random"), (45, 6, "random"), (35, 10, "poly3"), (40, 12, "random"),
                                          (30, 14, "caterpillar"), (2, 6, "random"), (25, 5, "random"), (20, 7, "poly4"),
                                          # ... and every odd one the next even instance with a phantom variable
                                          (60, 3, "random"), (45, 9, "random"), (40, 11, "poly3"), (35, 13, "caterpillar"),
                                          (50, 15, "random"), (2, 15, "random"), (3, 9, "random"), (30, 15, "poly4")])
def test_random_tree_cliquetree_vs_oracle(P, ntips, p, kind):
    from pgbp_amd import synth as S
    rng = np.random.default_rng(1000 * p + ntips)
    if kind == "random":
        tr = S.random_tree(ntips, rng)
    elif kind == "caterpillar":
        tr = S.caterpillar_tree(ntips, rng)
    else:
        tr = S.random_multifurcating_tree(ntips, int(kind[4:]), rng)
    R = S.random_rate_matrix(p, rng)
    mu = rng.standard_normal(p)
    X = S.simulate_bm(tr, R, mu, rng)
    prob = S.cliquetree_of_tree(tr, p)
    packed = S.bm_factors_cliquetree(tr, prob, R, mu, X)
    pcgb = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packed)
    ocgb = oracle_cgb_from_problem(prob, packed, p)
    spt = oracle_schedule(prob)
    # postorder only -> log-likelihood (src/calibration.jl:205-212)
    assert P.propagate_1traversal_postorder_(pcgb, *spt)
    assert OC.propagate_1traversal_postorder(ocgb, *spt)
    ll = pcgb.integratebelief_(prob.root_cluster)[1]
    assert rel_close(ll, ocgb.integratebelief(prob.root_cluster)[1])
    assert rel_close(ll, S.bm_loglik_pruning(tr, R, mu, X))
    # full calibration
    pcgb.init_beliefs_reset_fromfactors_()
    pcgb.init_messagecalibrationflags_reset_()
    ocgb.init_beliefs_reset_fromfactors()
    ocgb.init_messagecalibrationflags_reset()
    pres = P.calibrate_(pcgb, prob.schedule, 2)
    ores = OC.calibrate(ocgb, [spt], 2)
    assert pres == ores == (True, True)
    assert_beliefs_close(pcgb, ocgb)
    ref = pack_oracle(ocgb, prob)
    # every belief integrates to the same log-likelihood
    for i in range(0, len(prob.dims), max(1, len(prob.dims) // 25)):
        assert rel_close(pcgb.integratebelief_(i)[1], ll)


def test_multi_site_batch(P):
    """n_sites independent replicas (different data) in one engine == one engine per site."""
    from pgbp_amd import synth as S
    rng = np.random.default_rng(77)
    tr = S.random_tree(30, rng)
    p, ns = 2, 5
    prob = S.cliquetree_of_tree(tr, p)
    R = S.random_rate_matrix(p, rng)
    mu = np.zeros(p)
    packs, lls = [], []
    for s in range(ns):
        X = S.simulate_bm(tr, R, mu, rng)
        packs.append(S.bm_factors_cliquetree(tr, prob, R, mu, X))
        lls.append(S.bm_loglik_pruning(tr, R, mu, X))
    big = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx,
                                           np.stack(packs), n_sites=ns)
    assert P.calibrate_(big, prob.schedule, 2) == (True, True)
    mu_all, norm, info = big.integratebelief_(prob.root_cluster, all_sites=True)
    assert not info.any()
    for s in range(ns):
        assert rel_close(norm[s], lls[s])
        single = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off,
                                                  prob.scope_idx, packs[s])
        assert P.calibrate_(single, prob.schedule, 2) == (True, True)
        assert np.array_equal(single._packed[0], big._packed[s])   # bit-identical


def test_bethe_tree_vs_cliquetree(P):
    """cfg2 shape: the Bethe cluster graph of a tree is a tree, so one calibration is exact."""
    from pgbp_amd import synth as S
    rng = np.random.default_rng(5)
    tr = S.random_tree(50, rng)
    p = 3
    R = S.random_rate_matrix(p, rng)
    mu = rng.standard_normal(p)
    X = S.simulate_bm(tr, R, mu, rng)
    prob = S.bethe_of_tree(tr, p)
    n = tr.ntips
    assert prob.nclusters == 3 * n - 3 and len(prob.dims) - prob.nclusters == 3 * n - 4  # test_clustergraph.jl:41-52
    packed = S.bm_factors_bethe(tr, prob, R, mu, X)
    cgb = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packed)
    assert P.calibrate_(cgb, prob.schedule, 2) == (True, True)
    ll = S.bm_loglik_pruning(tr, R, mu, X)
    for i in range(0, len(prob.dims), 7):
        assert rel_close(cgb.integratebelief_(i)[1], ll)


def test_run_to_run_bitwise_deterministic(P):
    from pgbp_amd import synth as S
    rng = np.random.default_rng(9)
    tr = S.random_tree(200, rng)
    p = 4
    R = S.random_rate_matrix(p, rng)
    X = S.simulate_bm(tr, R, np.zeros(p), rng)
    prob = S.cliquetree_of_tree(tr, p)
    packed = S.bm_factors_cliquetree(tr, prob, R, np.zeros(p), X)
    outs = []
    for _ in range(3):
        cgb = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packed)
        assert P.calibrate_(cgb, prob.schedule, 2) == (True, True)
        outs.append(cgb._packed.copy())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_bs16_layout_roundtrip_and_asymmetric_fallback(P):
    """The symmetric block-packed device layout (pgbp_bs16.hpp) is an internal detail: (i) beliefs uploaded,
    calibrated (register-resident kernel in BS16) and downloaded agree with the oracle -- covered by every
    p = 16 test above; (ii) a download straight after an upload returns the input bit-for-bit for symmetric
    input; (iii) input that is NOT symmetric keeps the plain layout (the reference reads upper(J_I) and J_SI,
    src/beliefupdates.jl:59,68) and still matches the oracle, which follows the reference's reads exactly."""
    from pgbp_amd import synth as S
    rng = np.random.default_rng(123)
    tr = S.random_tree(40, rng)
    p = 16
    R = S.random_rate_matrix(p, rng)
    R = (R + R.T) / 2
    X = S.simulate_bm(tr, R, np.zeros(p), rng)
    prob = S.cliquetree_of_tree(tr, p)
    packed = S.bm_factors_cliquetree(tr, prob, R, np.zeros(p), X)
    # make every J exactly symmetric (np.linalg.inv does not guarantee it)
    for i in range(prob.nclusters):
        m = int(prob.dims[i]); o = prob.packed_off[i]
        J = packed[o:o + m * m].reshape(m, m, order="F")
        packed[o:o + m * m] = ((J + J.T) / 2).reshape(-1, order="F")
    cgb = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packed)
    cgb.set_schedule(prob.schedule)
    assert P.propagate_1traversal_postorder_(cgb, None, None, *prob.schedule[0], sync=False)   # device now in BS16
    cgb.init_beliefs_reset_fromfactors_(sync=True)                                               # back to plain on pull
    assert np.array_equal(cgb._packed[0], packed)                                                # (ii) bit-exact
    # (iii) asymmetric input: perturb the LOWER triangle of one internal clique's integrated block
    pa, ch = prob.schedule[0]
    snd = next(int(c) for c in ch if prob.dims[c] == 2 * p)
    o = prob.packed_off[snd]
    packed2 = packed.copy()
    J = packed2[o:o + 4 * p * p].reshape(2 * p, 2 * p, order="F")
    J[3, 1] += 0.37          # below the diagonal of J_I: the reference never reads it
    cgb2 = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packed2)
    ocgb = oracle_cgb_from_problem(prob, packed2, p)
    assert P.calibrate_(cgb2, prob.schedule, 2)[0] and OC.calibrate(ocgb, [oracle_schedule(prob)], 2)[0]
    assert_beliefs_close(cgb2, ocgb)
    ll = cgb2.integratebelief_(prob.root_cluster)[1]
    assert rel_close(ll, S.bm_loglik_pruning(tr, R, np.zeros(p), X))   # the perturbation is invisible to the path


def test_loopy_bethe_multi_tree_schedule(P, caplog):
    """cfg5-shaped case at test size: loopy BP on a Bethe cluster graph of a network (test_calibration.jl:79-106),
    schedule = 2 spanning trees, auto stop; every iteration goes through the generic kernel (ragged dims 0..3)."""
    g = G["calibration_bethe_level1"]
    net = ON.read_newick(g["net"])
    cg = OCG.bethe(net)
    sched = OCG.spanningtrees_clusterlist(cg, net)
    ocgb, pcgb = build_both(P, net, cg, make_model(g["model"]), [g["y"]], g["taxa"])
    olog = []
    ores = OC.calibrate(ocgb, sched, g["niter"], auto=True, info=True, log=olog)
    with caplog.at_level(logging.INFO, logger="PhyloGaussianBeliefProp"):
        pres = P.calibrate_(pcgb, sched, g["niter"], auto=True, info=True)
    assert pres == ores == (True, True)
    assert "calibration reached: iteration 5, schedule tree 1" in caplog.text
    r = pcgb.last_results[0]
    assert (r.iter_reached, r.tree_reached) == (5, 1)
    assert_beliefs_close(pcgb, ocgb)
    i3 = next(i for i, n in enumerate(net.vec_node) if not n.leaf and net.root in net.parents(n))
    ind = pcgb.clusterindex(net.vec_node[i3].name)
    mu, _ = pcgb.integratebelief_(ind)
    assert abs(mu[-1] - g["posterior_mean_I3"]) <= g["rtol"] * abs(g["posterior_mean_I3"])
    # without auto: all 20 iterations run, still calibrated, same fixed point
    ocgb2, pcgb2 = build_both(P, net, cg, make_model(g["model"]), [g["y"]], g["taxa"])
    assert P.calibrate_(pcgb2, sched, g["niter"]) == OC.calibrate(ocgb2, sched, g["niter"]) == (True, True)
    assert_beliefs_close(pcgb2, ocgb2)


def test_cfg4_univariate_ou_sites(P):
    """cfg4-shaped case at test size: independent univariate OU problems (different alpha, sigma2, theta and
    data per site) on one tree = one engine with n_sites replicas (thread-per-site kernel bp_level_uni when
    n_sites >= 8, wavefront kernel below); each site's log-likelihood against the oracle's dense-MVN likelihood
    (recipe of test/test_evomodels.jl:121-167), calibrated beliefs and flags against the oracle, and a
    non-positive-definite failure confined to its site."""
    from pgbp_amd import synth as S
    from oracle import densemvn as OD
    from oracle import models as OM
    rng = np.random.default_rng(4)
    tr = S.random_tree(25, rng)
    names = [f"n{i}" for i in range(tr.nnodes)]
    net = ON.read_newick(tr.newick(names))
    net.set_preorder(names)
    taxa = [names[i] for i in range(tr.nnodes) if tr.is_leaf[i]]
    prob = S.cliquetree_of_tree(tr, 1)
    clusters = [(str(i), [int(a), int(b)]) for i, (a, b) in enumerate(prob.cluster_nodes)]
    edges = [(int(a), int(b), [int(prob.sepset_nodes[k])]) for k, (a, b) in enumerate(prob.sepset_clusters)]
    cg = OB.ClusterGraph(clusters, edges, "cliquetree")
    spt = ([str(x) for x in prob.schedule[0][0]], [str(x) for x in prob.schedule[0][1]],
           prob.schedule[0][0].tolist(), prob.schedule[0][1].tolist())
    for ns in (3, 11, 70):   # 3: wavefront-per-message kernel; 11: thread-per-site kernel; 70: + site-minor layout
        packs, dense, ocgbs = [], [], []
        for s in range(ns):
            model = OM.UnivariateOrnsteinUhlenbeck(rng.uniform(0.5, 2), rng.uniform(0.1, 1), rng.normal(), rng.normal(), 0.0)
            y = [float(v) for v in rng.normal(size=len(taxa))]
            ocgb = oracle_setup(net, cg, model, [y], taxa)
            assert [b.dimension for b in ocgb.belief] == prob.dims.tolist()
            packs.append(pack_oracle(ocgb, prob))
            dense.append(OD.loglik(net, model, [y], taxa))
            ocgbs.append(ocgb)
        eng = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx,
                                               np.stack(packs), n_sites=ns)
        assert P.propagate_1traversal_postorder_(eng, None, None, *prob.schedule[0])
        mu, norm, info = eng.integratebelief_(prob.root_cluster, all_sites=True)
        assert not info.any()
        for s in range(ns):
            assert rel_close(norm[s], dense[s]), (ns, s, norm[s], dense[s])
        # full calibration of every site against the oracle
        eng.init_beliefs_reset_fromfactors_()
        eng.init_messagecalibrationflags_reset_()
        assert P.calibrate_(eng, prob.schedule, 2) == (True, True)
        for s in (0, ns - 1):
            ocgbs[s].init_beliefs_reset_fromfactors()
            assert OC.calibrate(ocgbs[s], [spt], 2) == (True, True)
            ref = pack_oracle(ocgbs[s], prob)
            assert np.allclose(eng._packed[s], ref, rtol=1e-8, atol=1e-8)
        # failure confined to one site
        bad = np.stack(packs).copy()
        pa, ch = prob.schedule[0]
        snd = next(int(c) for c in ch if prob.dims[c] == 2)
        bad[ns - 2, prob.packed_off[snd]] = -5.0
        eng2 = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, bad,
                                                n_sites=ns)
        eng2.site = ns - 2
        assert P.calibrate_(eng2, prob.schedule, 1, verbose=False) == (False, False)
        res = eng2.last_results
        assert [res[s].succ for s in range(ns)] == [int(s != ns - 2) for s in range(ns)]
        assert res[ns - 2].fail_info == 1 and res[ns - 2].fail_dir == 0
    # site-minor state (ns = 70): entry points that need the plain layout bring it back transparently
    fe, feinfo = eng.free_energy(all_sites=True)
    assert not feinfo.any() and all(rel_close(a, b) for a, b in zip(fe[ns - 1], OB.free_energy(ocgbs[ns - 1])))
    assert P.calibrate_(eng, prob.schedule, 1) == (True, True)          # back to site-minor, state intact
    assert np.allclose(eng._packed[ns - 1], pack_oracle(ocgbs[ns - 1], prob), rtol=1e-8, atol=1e-8)
    # device factor fill straight into the site-minor state: univariate BM with one (sigma2, mu) per site
    sig = rng.uniform(0.5, 2.0, size=ns)
    mus = rng.normal(size=ns)
    X = S.simulate_bm_uni_sites(tr, sig, mus, rng)
    eng.bm_tree_setup(*S.bm_tree_table(tr, prob), X[:, :, None])
    eng.assignfactors_bm_(sig[:, None, None], mus[:, None])
    assert P.calibrate_(eng, prob.schedule, 1)[0]
    ll = eng.integratebelief_(prob.root_cluster, all_sites=True)[1]
    assert np.allclose(ll, S.bm_loglik_pruning_uni_sites(tr, sig, mus, X), rtol=1e-9, atol=0)
    import ctypes as C
    from pgbp_amd import _lib as L
    o = eng._opts()
    assert eng._lib.pgbp_enqueue_loglik_bm(eng._eng, 2, C.byref(o)) == 0      # fill + postorder + integrate, twice
    norm, info = np.zeros(ns), np.zeros(ns, np.int32)
    assert eng._lib.pgbp_fetch_loglik(eng._eng, L.f64p(norm), L.i32p(info)) == 0 and not info.any()
    assert np.allclose(norm, ll, rtol=1e-12, atol=0)


def test_multi_site_p16_bs16(P):
    """n_sites > 1 with the register-resident kernel and the BS16 layout (grid.y = site): every site equals
    its single-site run bit for bit, and its log-likelihood equals the independent pruning value."""
    from pgbp_amd import synth as S
    rng = np.random.default_rng(31)
    tr = S.random_tree(45, rng)
    p, ns = 16, 3
    prob = S.cliquetree_of_tree(tr, p)
    R = S.random_rate_matrix(p, rng)
    R = (R + R.T) / 2
    packs, lls = [], []
    for s in range(ns):
        mu = rng.standard_normal(p)
        X = S.simulate_bm(tr, R, mu, rng)
        packs.append(S.bm_factors_cliquetree(tr, prob, R, mu, X))
        lls.append(S.bm_loglik_pruning(tr, R, mu, X))
    big = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx,
                                           np.stack(packs), n_sites=ns)
    assert P.calibrate_(big, prob.schedule, 2) == (True, True)
    _, norm, info = big.integratebelief_(prob.root_cluster, all_sites=True)
    assert not info.any()
    for s in range(ns):
        assert rel_close(norm[s], lls[s])
        single = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off,
                                                  prob.scope_idx, packs[s])
        assert P.calibrate_(single, prob.schedule, 2) == (True, True)
        assert np.array_equal(single._packed[0], big._packed[s])
    # per-site results of a failure in ONE site: the others are unaffected
    pa, ch = prob.schedule[0]
    snd = next(int(c) for c in ch if prob.dims[c] == 2 * p)
    bad = np.stack(packs).copy()
    bad[1, prob.packed_off[snd]] = -1e9
    eng = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, bad,
                                           n_sites=ns)
    eng.site = 1
    assert P.calibrate_(eng, prob.schedule, 2, verbose=False) == (False, False)
    res = eng.last_results
    assert [res[s].succ for s in range(ns)] == [1, 0, 1] and res[1].fail_info == 1
    assert res[0].iscal == 1 and res[2].iscal == 1
    assert np.array_equal(eng._packed[0], big._packed[0]) and np.array_equal(eng._packed[2], big._packed[2])


def test_single_belief_access_and_set(P):
    """pgbp_get_belief / pgbp_set_belief (reading b[i] after calibration, editing one factor) across the
    internal layout switches."""
    import ctypes as C
    from pgbp_amd import _lib as L
    from pgbp_amd import synth as S
    rng = np.random.default_rng(8)
    tr = S.random_tree(20, rng)
    p = 16
    prob = S.cliquetree_of_tree(tr, p)
    R = S.random_rate_matrix(p, rng); R = (R + R.T) / 2
    X = S.simulate_bm(tr, R, np.zeros(p), rng)
    packed = S.bm_factors_cliquetree(tr, prob, R, np.zeros(p), X)
    cgb = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packed)
    assert P.calibrate_(cgb, prob.schedule, 1, sync=False)[0]          # device in BS16 now
    lib = P.load()
    i = int(np.argmax(prob.dims[:prob.nclusters]))
    m = int(prob.dims[i])
    rec = np.zeros(m * m + m + 1)
    assert lib.pgbp_get_belief(cgb._eng, 0, i, L.f64p(rec)) == 0        # converts back to the ABI layout
    cgb.pull()
    assert np.array_equal(rec, cgb._packed[0, prob.packed_off[i]:prob.packed_off[i + 1]])
    J = rec[:m * m].reshape(m, m, order="F")
    assert np.allclose(J, J.T, rtol=0, atol=1e-9 * np.abs(J).max())
    rec2 = rec.copy(); rec2[-1] += 1.5                                   # g += 1.5 on one belief
    assert lib.pgbp_set_belief(cgb._eng, 0, i, L.f64p(rec2)) == 0
    n0 = cgb.integratebelief_(i)[1]
    cgb.pull()
    assert cgb._packed[0, prob.packed_off[i + 1] - 1] == rec2[-1]


@pytest.mark.parametrize("graph,ntips,p", [("cliquetree", 30, 16), ("cliquetree", 25, 3), ("bethe", 20, 4),
                                           ("cliquetree", 2, 16), ("cliquetree", 3, 1), ("bethe", 35, 8),
                                           ("bethe", 30, 16), ("cliquetree", 40, 8), ("cliquetree", 30, 6),
                                           ("bethe", 25, 12), ("cliquetree", 20, 2), ("bethe", 20, 5),
                                           ("cliquetree", 18, 15), ("cliquetree", 22, 9)])
def test_device_factor_fill_bm_tree(P, graph, ntips, p):
    """pgbp_bm_tree_assignfactors (assignfactors! on the device) == the host fill that
    tests/test_plan_cpu.py pins against the oracle's assignfactors! restatement; a second parameter set is
    evaluated without re-uploading anything but (R^-1, log det R, mu)."""
    import ctypes as C
    from pgbp_amd import _lib as L
    from pgbp_amd import synth as S
    rng = np.random.default_rng(500 + ntips + p)
    tr = S.random_tree(ntips, rng)
    R = S.random_rate_matrix(p, rng); R = (R + R.T) / 2
    mu = rng.standard_normal(p)
    X = S.simulate_bm(tr, R, mu, rng)
    prob = S.cliquetree_of_tree(tr, p) if graph == "cliquetree" else S.bethe_of_tree(tr, p)
    fill = S.bm_factors_cliquetree if graph == "cliquetree" else S.bm_factors_bethe
    cgb = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx,
                                           np.zeros(int(prob.packed_off[-1])))
    cgb.set_schedule(prob.schedule)
    cgb.bm_tree_setup(*S.bm_tree_table(tr, prob), X)
    for (R_, mu_) in ((R, mu), (2.0 * R + 0.3 * np.eye(p), mu + 0.5)):
        cgb.assignfactors_bm_(R_, mu_, sync=True)
        ref = fill(tr, prob, R_, mu_, X)
        scale = max(1.0, np.abs(ref).max())
        assert np.max(np.abs(cgb._packed[0] - ref)) <= 1e-12 * scale
        # the whole score(theta) body on the device: fill + postorder + root integrate
        lib = P.load()
        o = cgb._opts()
        assert lib.pgbp_enqueue_loglik_bm(cgb._eng, 2, C.byref(o)) == 0
        norm = np.zeros(1); info = np.zeros(1, np.int32)
        assert lib.pgbp_fetch_loglik(cgb._eng, L.f64p(norm), L.i32p(info)) == 0
        assert info[0] == 0 and rel_close(norm[0], S.bm_loglik_pruning(tr, R_, mu_, X))
        # and a full calibration from the device-filled factors
        cgb.assignfactors_bm_(R_, mu_)
        assert P.calibrate_(cgb, prob.schedule, 2) == (True, True)
        assert rel_close(cgb.integratebelief_(prob.root_cluster)[1], S.bm_loglik_pruning(tr, R_, mu_, X))


def test_free_energy_goldens_and_oracle(P):
    """factored_energy / free_energy (src/score.jl:151-182) on the device: test/test_calibration.jl:59
    (factored energy == log-likelihood on the calibrated clique tree), docs getting_started.md:288-291,
    and the oracle's restatement on a loopy (approximate) Bethe graph and on a BS-layout 16-trait tree."""
    g = G["calibration_cliquetree_level1"]
    net = ON.read_newick(g["net"])
    ct = OCG.cliquetree(net)
    spt = OCG.spanningtree_clusterlist(ct, OCG.default_rootcluster(ct, net))
    ocgb, pcgb = build_both(P, net, ct, make_model(g["model"]), [g["y"]], g["taxa"])
    assert P.calibrate_(pcgb, [spt])[0] and OC.calibrate(ocgb, [spt])[0]
    fe = pcgb.factored_energy()
    assert rel_close(fe[2], g["ll_every_belief"])
    for x, y in zip(fe, OB.factored_energy(ocgb)):
        assert rel_close(x, y)
    g = G["doctest_lazaridis"]
    net = ON.read_newick(g["net"])
    ct = OCG.cliquetree(net)
    spt = OCG.spanningtree_clusterlist(ct, OCG.default_rootcluster(ct, net))
    ocgb, pcgb = build_both(P, net, ct, make_model(g["model"]), [g["x"]], g["taxa"])
    assert P.calibrate_(pcgb, [spt])[0]
    assert abs(pcgb.factored_energy()[2] - g["factored_energy"]) <= 1e-10 * abs(g["factored_energy"])
    # loopy Bethe: an approximation, compared with the oracle
    g = G["calibration_bethe_level1"]
    net = ON.read_newick(g["net"])
    cg = OCG.bethe(net)
    sched = OCG.spanningtrees_clusterlist(cg, net)
    ocgb, pcgb = build_both(P, net, cg, make_model(g["model"]), [g["y"]], g["taxa"])
    assert P.calibrate_(pcgb, sched, 20, auto=True)[0] and OC.calibrate(ocgb, sched, 20, auto=True)[0]
    for x, y in zip(pcgb.free_energy(), OB.free_energy(ocgb)):
        assert rel_close(x, y)
    # 16-trait tree (packed layout on the device)
    from pgbp_amd import synth as S
    rng = np.random.default_rng(77)
    tr = S.random_tree(30, rng)
    p = 16
    R = S.random_rate_matrix(p, rng); R = (R + R.T) / 2
    X = S.simulate_bm(tr, R, np.zeros(p), rng)
    prob = S.cliquetree_of_tree(tr, p)
    packed = S.bm_factors_cliquetree(tr, prob, R, np.zeros(p), X)
    cgb = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packed)
    assert P.calibrate_(cgb, prob.schedule, 1, sync=False)[0]
    assert rel_close(cgb.factored_energy()[2], S.bm_loglik_pruning(tr, R, np.zeros(p), X))
    # before calibration the sepset beliefs are not normalisable: reported, not crashed
    cgb.init_beliefs_reset_fromfactors_(sync=False)
    out, info = cgb.free_energy(all_sites=True)
    assert info[0] > 0


def test_api_edge_cases(P):
    """niter = 0 returns (false, false) like the reference's empty loop (src/calibration.jl:45-59); a bad
    schedule is refused and the previous one stays in force; calls out of order fail with a message."""
    import ctypes as C
    from pgbp_amd import _lib as L
    from pgbp_amd import synth as S
    rng = np.random.default_rng(2)
    tr = S.random_tree(12, rng)
    p = 2
    prob = S.cliquetree_of_tree(tr, p)
    R = S.random_rate_matrix(p, rng)
    X = S.simulate_bm(tr, R, np.zeros(p), rng)
    packed = S.bm_factors_cliquetree(tr, prob, R, np.zeros(p), X)
    cgb = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packed)
    lib = P.load()
    res = (L.Result * 1)()
    o = cgb._opts()
    assert lib.pgbp_calibrate(cgb._eng, 1, C.byref(o), res) == L.ERR_STATE       # no schedule yet
    assert b"pgbp_set_schedule" in lib.pgbp_last_error(cgb._eng)
    assert P.calibrate_(cgb, prob.schedule, 0) == (False, False)
    assert np.array_equal(cgb._packed[0], packed)                                  # nothing ran
    pa, ch = prob.schedule[0]
    bad = (pa.copy(), ch.copy()); bad[1][-1] = bad[1][0]
    with pytest.raises(P.PgbpError) as ei:
        cgb.set_schedule([bad])
    assert ei.value.code == L.ERR_NOT_TREE
    cgb._schedule = None
    assert P.calibrate_(cgb, prob.schedule, 2) == (True, True)
    assert rel_close(cgb.integratebelief_(prob.root_cluster)[1], S.bm_loglik_pruning(tr, R, np.zeros(p), X))
    o2 = cgb._opts(atol=-1.0)
    assert lib.pgbp_calibrate(cgb._eng, 1, C.byref(o2), res) == L.ERR_INVALID      # refused explicitly


@pytest.mark.parametrize("variant", ["improper", "fixed"])
def test_calibration_level3_network(P, variant):
    """test/test_calibration.jl:131-185 on the device (generic kernel: hybrid-node clusters of 3 nodes, ragged
    scopes from a missing value, improper or fixed root): normalisation constant at every belief, posterior means."""
    g = G["calibration_level3_joingraph"]
    net = ON.read_newick(g["net"])
    ct = OCG.cliquetree(net)
    spt = OCG.spanningtree_clusterlist(ct, OCG.default_rootcluster(ct, net))
    ocgb, pcgb = build_both(P, net, ct, make_model(g["model_" + variant]), [g["y1"], g["y2"]], g["taxa"])
    assert P.calibrate_(pcgb, [spt])[0] and OC.calibrate(ocgb, [spt])[0]
    assert_beliefs_close(pcgb, ocgb)
    seen = {}
    for i, be in enumerate(ocgb.belief):
        if be.dimension == 0:
            continue
        mu, norm = pcgb.integratebelief_(i)
        assert abs(norm - g["norm_" + variant]) <= 1e-9 * abs(g["norm_" + variant])
        k = 0
        for col, lab in enumerate(be.nodelabel):
            d = int(be.inscope[:, col].sum())
            if d == be.ntraits:
                seen.setdefault(net.vec_node[lab - 1].name, mu[k:k + d])
            k += d
    for name, m in g["posterior_means_" + variant].items():
        assert np.allclose(seen[name], m, rtol=1.5e-8, atol=0), name


# ----------------------------------------------------------------------------- KL residuals, regularisers

def _kl_state(cgb, keys):
    return (np.array([cgb.messageresidual[k].kldiv for k in keys]),
            np.array([cgb.messageresidual[k].iscalibrated_kl for k in keys]))


def test_residual_kldiv_golden(P):
    """test/test_calibration.jl:13-33 on the device: a message with nothing to integrate turns the sepset
    (J, h) = ((1/3)[2 -1; -1 2], (1/3)[2, -1]) into (I, [0, 1]); residual_kldiv! = 1.215973 (R: rags2ridges::KLdiv)."""
    g = G["residual_kldiv"]
    dims = np.array([2, 2, 2], np.int32)
    off = np.array([0, 7, 14, 21])
    packed = np.zeros(21)
    packed[7:11] = np.eye(2).reshape(-1)
    packed[11:13] = [0.0, 1.0]
    packed[14:18] = (np.array([[2.0, -1.0], [-1.0, 2.0]]) / 3).reshape(-1)
    packed[18:20] = np.array([2.0, -1.0]) / 3
    cgb = P.ClusterGraphBelief.from_arrays(dims, [0, 1], [0, 2, 4], [0, 1, 0, 1], packed)
    assert cgb.messageresidual[(0, 1)].kldiv == -1.0 and not cgb.messageresidual[(0, 1)].iscalibrated_kl
    assert P.propagate_belief_(cgb, 0, 2, 1) is None
    mr = cgb.messageresidual[(0, 1)]
    assert np.allclose(mr.dJ, np.ones((2, 2)) / 3, rtol=1e-14) and np.allclose(mr.dh, np.array([-2.0, 4.0]) / 3, rtol=1e-14)
    assert cgb.residual_kldiv_(0, 2, 1) is False
    assert abs(mr.kldiv - g["kldiv"]) <= g["rtol"] * g["kldiv"]
    assert not mr.iscalibrated_kl
    assert cgb.messageresidual[(1, 0)].kldiv == -1.0          # the other direction was not touched
    # exact value: ( tr(J1 J0^-1) - p + (mu1-mu0)'J1(mu1-mu0) + log det J0 - log det J1 ) / 2
    J1 = np.array([[2.0, -1.0], [-1.0, 2.0]]) / 3
    d = np.array([1.0, 0.0]) - np.array([0.0, 1.0])
    exact = 0.5 * (np.trace(J1) - 2 + d @ J1 @ d - np.log(np.linalg.det(J1)))
    assert abs(mr.kldiv - exact) <= 1e-13 * abs(exact)
    # resending the same message: KL = 0, flag true
    assert P.propagate_belief_(cgb, 0, 2, 1) is None
    assert cgb.residual_kldiv_(0, 2, 1) is True and abs(mr.kldiv) <= 1e-14 and mr.iscalibrated_kl
    # a sepset that is not positive definite: nothing is updated, false (src/beliefs.jl:1063-1066)
    cgb.belief[2].J[0, 0] = -1.0
    cgb.push()
    cgb.init_messagecalibrationflags_reset_()
    assert cgb.residual_kldiv_(0, 2, 1) is False and cgb.messageresidual[(0, 1)].kldiv == -1.0


@pytest.mark.parametrize("ntips,p,nsites", [(40, 3, 1), (60, 16, 1), (50, 8, 3), (30, 1, 1), (25, 32, 1)])
def test_residual_kldiv_during_calibration(P, ntips, p, nsites):
    """calibrate!(...; update_residualkldiv=true) (src/calibration.jl:128,154): kldiv and iscalibrated_kl of every
    directed message against the oracle after 1 and after 2 iterations (p = 16: BS16 layout of sepsets and
    residuals; p = 32: generic kernel).  Tolerance: 1e-8 relative (1e-8 absolute once the KL is ~0)."""
    from pgbp_amd import synth as S
    rng = np.random.default_rng(77 * p + ntips)
    tr = S.random_tree(ntips, rng)
    R = S.random_rate_matrix(p, rng)
    mu = rng.standard_normal(p)
    prob = S.cliquetree_of_tree(tr, p)
    sites = [S.bm_factors_cliquetree(tr, prob, R, mu, S.simulate_bm(tr, R, mu, rng)) for _ in range(nsites)]
    pcgb = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx,
                                            np.stack(sites), n_sites=nsites)
    spt = oracle_schedule(prob)
    keys = [(int(a), int(c)) for a, c in prob.sepset_clusters] + [(int(c), int(a)) for a, c in prob.sepset_clusters]
    ocgbs = [oracle_cgb_from_problem(prob, sites[s], p) for s in range(nsites)]
    for it in range(2):
        assert P.calibrate_(pcgb, prob.schedule, 1, update_residualkldiv=True)[0]
        for s in range(nsites):
            assert OC.calibrate(ocgbs[s], [spt], 1, update_residualkldiv=True)[0]
            pcgb.site = s
            pk, pf = _kl_state(pcgb, keys)
            ok, of = _kl_state(ocgbs[s], keys)
            assert np.all(np.abs(pk - ok) <= 1e-8 * np.maximum(1.0, np.abs(ok))), (it, s, np.abs(pk - ok).max())
            # flags agree wherever the KL is not within rounding of the 1e-5 threshold
            clear = np.abs(np.abs(ok) - 1e-5) > 1e-7
            assert np.array_equal(pf[clear], of[clear])
            if it == 0:
                assert np.any(ok == -1.0) and np.any(ok > 1e-5)   # postorder: sepsets were improper before
            else:
                assert np.all(np.abs(pk) <= 1e-8) and pf.all()
        pcgb.site = 0
    # reset_kl semantics (src/beliefs.jl:973-979)
    pcgb.init_messagecalibrationflags_reset_(reset_kl=False)
    pk, pf = _kl_state(pcgb, keys)
    nonempty = np.array([prob.dims[prob.nclusters + k] > 0 for k in range(len(prob.sepset_clusters))] * 2)
    assert not pf[nonempty].any() and pf[~nonempty].all() and np.all(np.abs(pk) <= 1e-8)
    pcgb.init_messagecalibrationflags_reset_(reset_kl=True)
    pk = _kl_state(pcgb, keys)[0]
    assert np.all(pk[nonempty] == -1.0) and np.all(pk[~nonempty] == 0.0)


def test_regularize_bycluster_device(P):
    """regularizebeliefs_bycluster! (src/clustergraphbeliefs.jl:235-275) on the device: the edited beliefs are
    bit-identical to the oracle's (same eps, same order of additions); test/test_calibration.jl:72-76: the
    clique-tree log-likelihood is unchanged ("graph invariant was preserved")."""
    g = G["calibration_cliquetree_level1"]
    net = ON.read_newick(g["net"])
    ct = OCG.cliquetree(net)
    spt = OCG.spanningtree_clusterlist(ct, OCG.default_rootcluster(ct, net))
    ocgb, pcgb = build_both(P, net, ct, make_model(g["model"]), [g["y"]], g["taxa"])
    P.regularizebeliefs_bycluster_(pcgb, ct)
    OB.regularizebeliefs_bycluster(ocgb)
    for pb, ob in zip(pcgb.belief, ocgb.belief):
        assert np.array_equal(pb.J, ob.J) and np.array_equal(pb.h, ob.h) and pb.g[0] == ob.g[0]
    assert P.calibrate_(pcgb, [spt])[0] and OC.calibrate(ocgb, [spt])[0]
    assert_beliefs_close(pcgb, ocgb)
    for i in range(len(ocgb.belief)):
        assert rel_close(pcgb.integratebelief_(i)[1], g["ll_every_belief"])


@pytest.mark.parametrize("p,nsites", [(16, 1), (8, 2), (3, 1)])
def test_regularize_bycluster_tree_layouts(P, p, nsites):
    """Device regulariser on clique trees of random trees, including the BS16-resident state (p = 16 / 8: the
    engine converts to the plain layout, edits, and the next calibration converts back): bit-identical edits,
    log-likelihood unchanged."""
    from pgbp_amd import synth as S
    rng = np.random.default_rng(4242 + p)
    tr = S.random_tree(70, rng)
    R = S.random_rate_matrix(p, rng)
    mu = rng.standard_normal(p)
    prob = S.cliquetree_of_tree(tr, p)
    Xs = [S.simulate_bm(tr, R, mu, rng) for _ in range(nsites)]
    sites = [S.bm_factors_cliquetree(tr, prob, R, mu, X) for X in Xs]
    pcgb = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx,
                                            np.stack(sites), n_sites=nsites)
    assert P.calibrate_(pcgb, prob.schedule, 1)[0]           # puts the state in its fast layout
    pcgb.init_beliefs_reset_fromfactors_()
    P.regularizebeliefs_bycluster_(pcgb)
    for s in range(nsites):
        ocgb = oracle_cgb_from_problem(prob, sites[s], p)
        OB.regularizebeliefs_bycluster(ocgb)
        ref = pack_oracle(ocgb, prob)
        if p == 3:
            assert np.array_equal(pcgb._packed[s], ref)
        else:
            # the symmetric block-packed layout keeps one representative of every (i,j)/(j,i) pair: the rounding
            # asymmetry of the input precision (~1e-17 relative) does not survive the round trip
            assert np.allclose(pcgb._packed[s], ref, rtol=1e-13, atol=0.0)
            d = prob.packed_off
            for i in range(0, len(prob.dims), 7):      # diagonals (where eps lands) are exact
                m = int(prob.dims[i])
                Jp = pcgb._packed[s][d[i]:d[i] + m * m].reshape(m, m)
                assert np.array_equal(np.diag(Jp), np.diag(ref[d[i]:d[i] + m * m].reshape(m, m)))
    assert P.calibrate_(pcgb, prob.schedule, 1)[0]
    for s in range(nsites):
        pcgb.site = s
        ll = pcgb.integratebelief_(prob.root_cluster, all_sites=False)[1]
        assert rel_close(ll, S.bm_loglik_pruning(tr, R, mu, Xs[s]))
    pcgb.site = 0


def test_regularize_onschedule_bethe_pipeline(P, caplog):
    """test/test_calibration.jl:94-105 end to end on the device: regularizebeliefs_onschedule! (default messages
    on the host, real messages through pgbp_propagate), then calibrate!(cgb, sched, 20; auto=true):
    "calibration reached: iteration 5, schedule tree 1", posterior mean at I3; beliefs against the oracle."""
    g = G["calibration_bethe_level1"]
    net = ON.read_newick(g["net"])
    cg = OCG.bethe(net)
    sched = OCG.spanningtrees_clusterlist(cg, net)
    ocgb, pcgb = build_both(P, net, cg, make_model(g["model"]), [g["y"]], g["taxa"])
    P.regularizebeliefs_onschedule_(pcgb, cg)
    OB.regularizebeliefs_onschedule(ocgb)
    assert_beliefs_close(pcgb, ocgb)
    with caplog.at_level(logging.INFO, logger="PhyloGaussianBeliefProp"):
        assert P.calibrate_(pcgb, sched, g["niter"], auto=True, info=True) == (True, True)
    assert "calibration reached: iteration 5, schedule tree 1" in caplog.text
    assert OC.calibrate(ocgb, sched, g["niter"], auto=True) == (True, True)
    assert_beliefs_close(pcgb, ocgb)
    i3 = next(i for i, n in enumerate(net.vec_node) if not n.leaf and net.root in net.parents(n))
    mu, _ = pcgb.integratebelief_(pcgb.clusterindex(net.vec_node[i3].name))
    assert abs(mu[-1] - g["posterior_mean_I3"]) <= g["rtol"] * abs(g["posterior_mean_I3"])
    # the regularised loopy graph has proper beliefs: the free energy is finite and matches the oracle
    fe = pcgb.free_energy()
    ofe = OB.free_energy(ocgb)
    assert all(rel_close(a, b) for a, b in zip(fe, ofe))


def test_regularize_bynodesubtree(P):
    """regularizebeliefs_bynodesubtree! (src/clustergraphbeliefs.jl:306-340) against the oracle on the level-1
    clique tree (test/test_calibration.jl:67-71) and on its Bethe cluster graph; the clique-tree likelihood
    is unchanged."""
    g = G["calibration_cliquetree_level1"]
    net = ON.read_newick(g["net"])
    for cg in (OCG.cliquetree(net), OCG.bethe(net)):
        ocgb, pcgb = build_both(P, net, cg, make_model(g["model"]), [g["y"]], g["taxa"])
        P.regularizebeliefs_bynodesubtree_(pcgb, cg)
        OB.regularizebeliefs_bynodesubtree(ocgb)
        for pb, ob in zip(pcgb.belief, ocgb.belief):
            assert np.array_equal(pb.J, ob.J)
        if cg.method != "Bethe":
            spt = OCG.spanningtree_clusterlist(cg, OCG.default_rootcluster(cg, net))
            assert P.calibrate_(pcgb, [spt])[0]
            for i in range(len(ocgb.belief)):
                assert rel_close(pcgb.integratebelief_(i)[1], g["ll_every_belief"])


# ----------------------------------------------------------------------------- BASELINE.json configurations at full size

@pytest.mark.parametrize("ntips,p,graph", [(50000, 16, "cliquetree"), (10000, 8, "bethe")],
                         ids=["cfg3_50k_p16_cliquetree", "cfg2_10k_p8_bethe"])
def test_full_size_config_against_c_oracle(P, ntips, p, graph):
    """BASELINE.json configs[2] (the headline) and configs[1] at their full sizes: every calibrated belief, every
    residual flag and the log-likelihood against the plain-C sequential engine of the oracle (reference message order;
    itself pinned to the numpy restatement in tests/test_oracle_c.py), plus the properties that do not depend on size:
    the likelihood equals the independent pruning algorithm's, every belief integrates to it, and a further
    calibration leaves the calibrated state where it is.  Tolerances: 1e-8 relative (log-likelihood), 1e-8 * max|.| per
    belief."""
    from oracle import cengine
    from pgbp_amd import synth as S
    rng = np.random.default_rng(3)
    tr = S.random_tree(ntips, rng)
    R = S.random_rate_matrix(p, rng)
    mu = np.zeros(p)
    X = S.simulate_bm(tr, R, mu, rng)
    if graph == "cliquetree":
        prob = S.cliquetree_of_tree(tr, p)
        packed = S.bm_factors_cliquetree(tr, prob, R, mu, X)
    else:
        prob = S.bethe_of_tree(tr, p)
        packed = S.bm_factors_bethe(tr, prob, R, mu, X)
    ll_ref = S.bm_loglik_pruning(tr, R, mu, X)
    pcgb = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packed)
    eng = cengine.Engine(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packed)
    pa, ch = prob.schedule[0]
    assert P.calibrate_(pcgb, prob.schedule, 2) == (True, True)
    assert eng.calibrate(pa, ch, 2, return_iscal=True) == (True, True)
    got, ref = pcgb._packed[0], eng.packed()
    off = prob.packed_off
    worst = 0.0
    for i in range(len(prob.dims)):
        a, b = got[off[i]:off[i + 1]], ref[off[i]:off[i + 1]]
        if a.size:
            worst = max(worst, float(np.max(np.abs(a - b))) / max(1.0, float(np.max(np.abs(b)))))
    assert worst <= 1e-8, worst
    _, flags = eng.residuals()
    assert np.array_equal(pcgb._flags().astype(bool), flags.astype(bool))
    ll = pcgb.integratebelief_(prob.root_cluster)[1]
    assert rel_close(ll, ll_ref) and rel_close(ll, eng.integrate(prob.root_cluster)[1])
    for i in np.random.default_rng(1).choice(len(prob.dims), size=64, replace=False):
        if prob.dims[i] > 0:
            assert rel_close(pcgb.integratebelief_(int(i))[1], ll_ref)
    before = got.copy()
    assert P.calibrate_(pcgb, prob.schedule, 1) == (True, True)     # idempotent on a calibrated clique tree
    assert np.max(np.abs(pcgb._packed[0] - before)) <= 1e-9 * max(1.0, float(np.max(np.abs(before))))


def test_full_size_cfg4_sites_against_pruning_and_c_oracle(P):
    """BASELINE.json configs[3] at full size: 1000 sites x 8 traits = 8 000 independent univariate OU problems on a
    20 000-tip tree (site-minor layout, thread-per-site kernel, OU factors assigned on the device).  Every problem's
    log-likelihood against the independent pruning recursion (1e-8 relative); 32 sampled problems belief by belief,
    flag by flag against the plain-C sequential engine on factors pulled from the device, and 64 sampled (problem, cluster)
    factors of the device fill against the oracle's assignfactors! formulas (1e-10); size-independent properties:
    a second calibration leaves the beliefs where they are, every flag is set, no message failed."""
    from oracle import cengine
    from pgbp_amd import synth as S
    ntips, nprob = 20000, 8000
    rng = np.random.default_rng(4)
    tr = S.random_tree(ntips, rng)
    sigma2 = rng.uniform(0.5, 2.0, size=nprob)
    alpha = rng.uniform(0.1, 1.0, size=nprob)
    theta = rng.normal(size=nprob)
    mu = rng.normal(size=nprob)
    X = S.simulate_ou_uni_sites(tr, sigma2, alpha, theta, mu, rng)
    ll_ref = S.ou_loglik_pruning_uni_sites(tr, sigma2, alpha, theta, mu, X)
    prob = S.cliquetree_of_tree(tr, 1)
    cgb = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, None, n_sites=nprob)
    cgb.set_schedule(prob.schedule)
    cgb.lg_setup(S.lg_tree_table(tr, prob, 1), X[:, :, None])
    cgb.assignfactors_lg_((sigma2 / (2.0 * alpha)).reshape(nprob, 1, 1, 1), mu[:, None], model="ou", alpha=alpha, theta=theta[:, None])
    lib = P.load()
    from pgbp_amd import _lib as L
    import ctypes as C
    opts = cgb._opts()
    norm, info = np.zeros(nprob), np.zeros(nprob, np.int32)
    assert lib.pgbp_enqueue_loglik_lg(cgb._eng, 1, C.byref(opts)) == 0
    assert lib.pgbp_fetch_loglik(cgb._eng, L.f64p(norm), L.i32p(info)) == 0
    assert not info.any()
    assert float(np.max(np.abs(norm - ll_ref) / np.maximum(1.0, np.abs(ll_ref)))) <= 1e-8
    # factors of a sample of 32 problems -> the sequential C engine (one download of the whole state, sliced on the host)
    sample = np.random.default_rng(0).choice(nprob, size=32, replace=False)
    assert lib.pgbp_reset_from_factors(cgb._eng) == 0
    psz = int(lib.pgbp_packed_size(cgb._eng))
    nb = len(prob.dims)
    factors = {}
    for s_ in sample:
        buf = np.zeros(psz)
        assert lib.pgbp_get_site_beliefs(cgb._eng, int(s_), L.f64p(buf)) == 0    # one site of the batch, packed
        factors[int(s_)] = buf
    # the device factor fill at full size against the ORACLE's assignfactors! formulas (oracle/models.py: the OU
    # factor_treeedge, absorbleaf, absorbevidence at the fixed root): 64 sampled (problem, cluster) pairs
    import types
    from oracle import beliefupdates as OBU
    from oracle import models as OM
    frng = np.random.default_rng(1)
    for _ in range(64):
        s_ = int(sample[frng.integers(len(sample))])
        c = int(frng.integers(1, tr.nnodes))                      # cluster c - 1 = {node c, its parent}
        model = OM.UnivariateOrnsteinUhlenbeck(sigma2[s_], alpha[s_], theta[s_], mu[s_], 0.0)
        phi = model.factor_treeedge(types.SimpleNamespace(length=float(tr.length[c]), gamma=1.0, number=c))
        if tr.is_leaf[c]:
            phi = OBU.absorbleaf(*phi, [float(X[s_, c])], rowlabel=c)
        if tr.parent[c] == 0:
            n = phi[0].shape[0]
            phi, _ = OBU.absorbevidence(*phi, range(n - 1, n), [float(mu[s_])])
        h_o, J_o, g_o = np.asarray(phi[0], float), np.asarray(phi[1], float), float(phi[2])
        m = int(prob.dims[c - 1])
        assert h_o.shape == (m,)
        rec_ = factors[s_][prob.packed_off[c - 1]:prob.packed_off[c]]
        ref_ = np.concatenate([J_o.reshape(-1, order="F"), h_o, [g_o]])
        assert float(np.max(np.abs(rec_ - ref_))) <= 1e-10 * max(1.0, float(np.max(np.abs(ref_)))), (s_, c, rec_, ref_)
    rec = np.zeros(psz)
    res = (L.Result * nprob)()
    assert lib.pgbp_calibrate(cgb._eng, 2, C.byref(opts), res) == 0
    assert all(res[i].succ == 1 and res[i].iscal == 1 for i in range(nprob))
    pa, ch = prob.schedule[0]
    for s_ in sample:
        eng = cengine.Engine(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, factors[int(s_)])
        assert eng.calibrate(pa, ch, 2, return_iscal=True) == (True, True)
        ref = eng.packed()
        got = np.zeros(psz)
        assert lib.pgbp_get_site_beliefs(cgb._eng, int(s_), L.f64p(got)) == 0
        if s_ == sample[0]:     # the single-belief read-back agrees with the per-site one
            for b in np.random.default_rng(2).choice(nb, size=50, replace=False):
                ln = int(prob.packed_off[b + 1] - prob.packed_off[b])
                if ln:
                    assert lib.pgbp_get_belief(cgb._eng, int(s_), int(b), L.f64p(rec)) == 0
                    assert np.array_equal(got[prob.packed_off[b]:prob.packed_off[b + 1]], rec[:ln])
        err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
        assert float(err.max()) <= 1e-8, (int(s_), float(err.max()))
        assert rel_close(eng.integrate(prob.root_cluster)[1], ll_ref[s_])
    # idempotence on calibrated clique trees, all 8 000 problems: the root's integral does not move
    mu0, n0, i0 = cgb.integratebelief_(prob.root_cluster, all_sites=True)
    assert lib.pgbp_calibrate(cgb._eng, 1, C.byref(opts), res) == 0
    mu1, n1, i1 = cgb.integratebelief_(prob.root_cluster, all_sites=True)
    assert not i0.any() and not i1.any()
    assert float(np.max(np.abs(n1 - n0) / np.maximum(1.0, np.abs(n0)))) <= 1e-10
    assert float(np.max(np.abs(n0 - ll_ref) / np.maximum(1.0, np.abs(ll_ref)))) <= 1e-8


@pytest.mark.parametrize("graph", ["joingraph", "bethe"])
def test_full_size_cfg5_network_against_c_oracle(P, graph):
    """BASELINE.json configs[4] at full size: heterogeneous BM (3 painted rates, 4 traits) on a level-3 network with
    20 000 tips and 5 000 reticulations in varied blobs (networks.py:random_level3_network_varied), loopy cluster graph
    (join-graph structuring with maxclustersize 3 -- the largest bound under which a level-3 network's join graph is
    loopy: its moral graph has cliques of at most 4 nodes -- or Bethe), regularised, calibrate!(beliefs, schedule, 100;
    auto=true) over the spanningtrees_clusterlist schedule.  The factors come from the device fill (checked at this size
    on 96 sampled clusters, 32 of them holding a hybrid family, against the oracle's assignfactors! restatement to 1e-10).
    Against the plain-C sequential engine from the same start: the same (iteration, schedule tree) at which calibration is
    detected and EVERY belief to 1e-8 * max|.|."""
    from oracle import cengine
    from pgbp_amd.regularization import regularizebeliefs_onschedule_
    rng = np.random.default_rng(5)                           # the recorded seed of cfg5
    p = 4
    net = P.random_level3_network_varied(20000, 5001, rng, n_colors=3)
    assert net.nhybrids >= 4900
    cn, ed, sn = P.joingraph(net.node2family, 3) if graph == "joingraph" else P.bethe(net.node2family)
    assert len(ed) > len(cn) - 1 + 500                      # hundreds of independent cycles
    st = P.allocate_scopes(cn, ed, sn, net, p)
    base = P.synth.random_rate_matrix(p, rng)
    base = (base + base.T) / 2
    rates = np.stack([base * f for f in (0.5, 1.0, 2.0)])
    mu = np.zeros(p)
    X = P.simulate_bm_network(net, rates, mu, rng)
    pe = [list(zip(net.length[i], net.gamma[i], net.color[i])) for i in range(net.nnodes)]
    fam = P.lg_families(st.clusters, st.node2cluster, net.node2family, st.node2fixed, pe, list(range(net.nnodes)), p, n_rates=3)
    sched = P.spanningtrees_clusterlist(len(cn), ed, cn, net.is_leaf)
    cgb = P.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None)
    cgb.lg_setup(fam, X)
    cgb.assignfactors_lg_(rates, mu)
    # the device factor fill at this size (50 003 node families) against the ORACLE's assignfactors! restatement, cluster by
    # cluster on a random sample of 96 clusters (helpers.oracle_cluster_factor: the oracle's factor_treeedge /
    # factor_hybridnode, absorbleaf, absorbevidence and mult! replayed for the families of one cluster)
    from helpers import oracle_cluster_factor
    from oracle import models as OM
    cgb.pull()
    filled = cgb._packed[0].copy()
    colors = {(ni, k): int(c) + 1 for ni in range(net.nnodes) for k, c in enumerate(net.color[ni])}
    omodel = OM.HeterogeneousBrownianMotion([rates[0], rates[1], rates[2]], colors, mu)
    fam_of = {}
    for ni, c in enumerate(st.node2cluster):
        fam_of.setdefault(c, []).append(ni)
    srng = np.random.default_rng(11)
    with_hybrid = [c for c, nis in fam_of.items() if any(len(net.node2family[ni]) > 2 for ni in nis)]
    sample_c = list(srng.choice(len(cn), size=64, replace=False)) + list(srng.choice(with_hybrid, size=32, replace=False))
    for c in sample_c:
        c = int(c)
        h_o, J_o, g_o = oracle_cluster_factor(omodel, net, st, X, c, fam_of.get(c, []))
        ref_ = np.concatenate([J_o.reshape(-1, order="F"), h_o, [g_o]])
        got_ = filled[cgb._poff[c]:cgb._poff[c + 1]]
        assert got_.shape == ref_.shape
        assert float(np.max(np.abs(got_ - ref_))) <= 1e-10 * max(1.0, float(np.max(np.abs(ref_)))), (c, cn[c])
    if graph == "joingraph":
        regularizebeliefs_onschedule_(cgb)
    else:
        assert P.load().pgbp_regularize_bycluster(cgb._eng) == 0
    cgb.pull()
    start = cgb._packed[0].copy()
    cgb.init_messagecalibrationflags_reset_()
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
    assert reached == (r.iter_reached, r.tree_reached) and reached[0] >= 5
    got, ref = cgb._packed[0], ce.packed()
    o = cgb._poff[:-1][np.diff(cgb._poff) > 0]
    err = float(np.max(np.maximum.reduceat(np.abs(got - ref), o) / np.maximum(1.0, np.maximum.reduceat(np.abs(ref), o))))
    assert err <= 1e-8, err


@pytest.mark.parametrize("ntips,nhyb,python_bp", [(400, 80, True), (1500, 300, False)], ids=["400tips_80hybrids", "1500tips_300hybrids"])
def test_cfg5_shaped_loopy_network(P, caplog, ntips, nhyb, python_bp):
    """BASELINE.json configs[4] at test size: heterogeneous BM (3 painted rates, 4 traits) on a random network with
    reticulations (triangles and 4-cycles, gamma ~ U(0.1, 0.5), no zero-length edges), loopy cluster graph (Bethe:
    hybrid families of dimension 12 on the generic kernel, tree edges and variable clusters on the P = 4 fast
    instance), regularizebeliefs_bycluster! on the device, calibrate!(beliefs, schedule, 100; auto=true) over the
    spanning-tree schedule of spanningtrees_clusterlist.  Checked against the plain-C sequential engine (and, at the
    small size, the numpy restatement): same iteration / schedule tree at which calibration is detected, calibrated
    beliefs to 1e-8 * max|.|, and the free energy of the fixed point."""
    from oracle import cengine
    rng = np.random.default_rng(5)
    net = ON.random_network(ntips, nhyb, rng)
    p = 4
    rates = [np.eye(p) * s + 0.3 * s for s in (0.5, 1.0, 2.0)]
    colors = {e.number: 1 + int(rng.integers(3)) for e in net.edges}
    model = OM.HeterogeneousBrownianMotion(rates, colors, np.zeros(p))
    taxa = net.tip_names
    tbl = [list(rng.normal(size=len(taxa))) for _ in range(p)]
    cg = OCG.bethe(net)
    sched = OCG.spanningtrees_clusterlist(cg, net)
    assert len(sched) >= 2                                   # loopy
    # the product's own schedule builder (host side, src/clustergraph.jl:908-937) gives the same trees
    psched = P.spanningtrees_clusterlist(len(cg.clusters), [(a, b) for (a, b, _) in cg.edges], [c[1] for c in cg.clusters],
                                         [n.leaf for n in net.vec_node], cg.labels)
    assert [tuple(map(list, t)) for t in psched] == [tuple(map(list, t)) for t in sched]
    ocgb, pcgb = build_both(P, net, cg, model, tbl, taxa)
    P.regularizebeliefs_bycluster_(pcgb, cg)
    OB.regularizebeliefs_bycluster(ocgb)
    for pb, ob in zip(pcgb.belief, ocgb.belief):
        assert np.array_equal(pb.J, ob.J)
    eng = cengine.Engine(pcgb._dims, pcgb._sepcl.reshape(-1), pcgb._scope_off, pcgb._scope_idx, pcgb._packed[0].copy())
    reached = None
    for it in range(1, 101):
        for j, spt in enumerate(sched, start=1):
            succ, iscal = eng.calibrate(spt[2], spt[3], 1, return_iscal=True)
            assert succ
            if iscal:
                reached = (it, j)
                break
        if reached:
            break
    assert reached is not None and reached[0] > 2            # genuinely iterative
    with caplog.at_level(logging.INFO, logger="PhyloGaussianBeliefProp"):
        assert P.calibrate_(pcgb, psched, 100, auto=True, info=True) == (True, True)
    r = pcgb.last_results[0]
    assert (r.iter_reached, r.tree_reached) == reached
    assert f"calibration reached: iteration {reached[0]}, schedule tree {reached[1]}" in caplog.text
    got, ref = pcgb._packed[0], eng.packed()
    off = pcgb._poff
    worst = 0.0
    for i in range(len(pcgb._dims)):
        a, b = got[off[i]:off[i + 1]], ref[off[i]:off[i + 1]]
        if a.size:
            worst = max(worst, float(np.max(np.abs(a - b))) / max(1.0, float(np.max(np.abs(b)))))
    assert worst <= 1e-8, worst
    if python_bp:
        olog = []
        assert OC.calibrate(ocgb, sched, 100, auto=True, info=True, log=olog) == (True, True)
        assert olog[-1] == ("info", f"calibration reached: iteration {reached[0]}, schedule tree {reached[1]}")
        assert_beliefs_close(pcgb, ocgb)
        fe, ofe = pcgb.free_energy(), OB.free_energy(ocgb)
        assert all(rel_close(a, b) for a, b in zip(fe, ofe))


def test_c_abi_example_runs(P):
    """examples/c_abi_example.c: a plain-C host over include/pgbp.h (what any FFI binding does): two clusters, one
    calibration, the log-likelihood of a two-node Gaussian chain against its closed form (exit code 0)."""
    import os
    import shutil
    import subprocess
    import tempfile
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(P.LIB_PATH)
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "ex")
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(root, "include"),
                               os.path.join(root, "examples", "c_abi_example.c"), "-L", libdir, "-lpgbp", "-lm", "-o", exe])
        env = dict(os.environ, LD_LIBRARY_PATH=libdir + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
        out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, (out.stdout, out.stderr)
        assert "succ 1 iscal 1" in out.stdout


def test_differential_fuzz_against_c_oracle(P):
    """A slice of tests/fuzz_gpu_vs_c_oracle.py in the suite: 150 random (tree shape, trait count 1..16, cluster graph,
    sites, iterations) cases, a quarter of them with a non-positive-definite block injected into one site, against the
    plain-C sequential engine: beliefs to 1e-8 * max|.|, flags, (succ, iscal), the first failure's (edge, dir, info)."""
    import fuzz_gpu_vs_c_oracle as F
    n_fail, worst, _ = F.run(150, 2024)
    assert n_fail >= 10 and worst <= 1e-8


@pytest.mark.parametrize("variant", ["improper", "fixed"])
def test_calibration_level3_joingraph_loopy_run_on_device(P, caplog, variant):
    """test/test_calibration.jl:138-176 on the device, cluster graph and schedule from the product's own host builders
    (joingraph, nodesubtree_clusterlist on plain arrays): JoinGraphStructuring(3), regularizebeliefs_bynodesubtree!,
    one schedule tree per node, calibrate!(cgb, sch, 10; auto=true, info=true) -> "calibration reached: iteration 4,
    schedule tree 1"; normalisation constants at clusters 6 (I1I2I3) and 2 (H1H2I1), posterior means."""
    g = G["calibration_level3_joingraph"]
    net = ON.read_newick(g["net"])
    net.set_preorder(g["preorder"])
    cn, ed, sn = P.joingraph(OCG.nodefamilies(net), g["maxclustersize"])
    names = [n.name for n in net.vec_node]
    cg = OB.ClusterGraph([("".join(names[v - 1] for v in n), n) for n in cn], [(a, b, s) for (a, b), s in zip(ed, sn)], "joingraph")
    for lab, i1 in g["cluster_index_1based"].items():
        assert cg.labels[i1 - 1] == lab
    ocgb, pcgb = build_both(P, net, cg, make_model(g["model_" + variant]), [g["y1"], g["y2"]], g["taxa"])
    P.regularizebeliefs_bynodesubtree_(pcgb, cg)
    OB.regularizebeliefs_bynodesubtree(ocgb)
    assert_beliefs_close(pcgb, ocgb)
    sch = []
    for v in range(1, len(names) + 1):
        st = P.nodesubtree_clusterlist(cn, ed, sn, v, labels=cg.labels)
        if st[0]:
            sch.append(st)
    with caplog.at_level(logging.INFO):
        assert P.calibrate_(pcgb, sch, g["niter"], auto=True, info=True) == (True, True)
    if variant == "improper":
        assert g["info_line"] in caplog.text
    assert OC.calibrate(ocgb, sch, g["niter"], auto=True) == (True, True)
    assert_beliefs_close(pcgb, ocgb)
    for lab, i1 in g["cluster_index_1based"].items():
        mu, norm = pcgb.integratebelief_(i1 - 1)
        assert abs(norm - g["norm_" + variant]) <= 1.5e-8 * abs(g["norm_" + variant])
    mu6 = pcgb.integratebelief_(g["cluster_index_1based"]["I1I2I3"] - 1)[0]
    want = [x for n in ("I1", "I2", "I3") for x in g["posterior_means_" + variant].get(n, [])]
    assert np.allclose(mu6[:len(want)], want, rtol=1.5e-8, atol=0)


def test_chain_fusion_opt_in_differential_fuzz(P):
    """PGBP_TUNING=chain_fusion (opt-in, read when an engine is created; a child process keeps the variable out of this one): unary clusters of a schedule tree are
    passed through inside one task of the generic kernel (pgbp_plan.cpp build_traversals).  The differential fuzz
    against the plain-C sequential engine (beliefs 1e-8, flags, first failure) must hold unchanged."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, PGBP_TUNING="chain_fusion")
    out = subprocess.run([sys.executable, os.path.join(here, "fuzz_gpu_vs_c_oracle.py"), "80", "77"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "80 cases ok" in out.stdout, (out.stdout[-1500:], out.stderr[-1500:])


@pytest.mark.parametrize("env", [
    {"PGBP_TUNING": "no_tail"},
    {"PGBP_TUNING": "no_prologue"},
    {"PGBP_TUNING": "no_prologue,no_tail"},
    {"PGBP_TUNING": "no_chunks"},
    {"PGBP_TUNING": "loop=0"},
    {"PGBP_TUNING": "loop=0,no_prologue"},
    {"PGBP_TUNING": "chunk_bins=3"},
    {"PGBP_TUNING": "chunk_bins=2,no_prologue"},
    {"PGBP_TUNING": "chunk_bins=0"},
], ids=["levels_only", "no_prologues", "no_prologues_levels_only", "no_chunks",
        "one_wave_per_record_loops", "one_wave_per_record_loops_no_prologues", "chunks_packed_into_3_workgroups",
        "chunks_packed_into_2_workgroups_no_prologues", "chunks_one_workgroup_per_tree"])
def test_launch_modes_differential_fuzz(P, env):
    """The launch modes of the register-resident kernel (pgbp_fast.hip: one group per workgroup; chunks of fused levels;
    the single-workgroup tail) run the same message body, with or without PROLOGUES (a Bethe graph's variable-to-factor
    messages riding in the record of the factor's own message).  PGBP_TUNING is read when an engine is created (child
    processes keep it out of this one): the level launches alone (no tail, no chunks), without the chunks, the two-level
    schedule without prologues, the tail and chunks on pgbp_fast.hip's own loop mode (one wavefront per record, loop=0)
    instead of pgbp_loop.hip (the default: two wavefronts per record, sender operands requested half a pass early, chains
    through LDS), and -- round 4 -- the trees of every chunk's forest PACKED into 3 or 2 workgroups (the default packs only
    above 256 trees, which these small cases never have; chunk_bins=0: one workgroup per tree, the round-3 launches): same
    messages, same order inside every task, so the packed launches must pass the same fuzz.  Same differential fuzz against the
    plain-C sequential engine as for the defaults: random trees (polytomies: tasks of 3 and 4 messages beside tasks of 1
    and 2 in one group; caterpillars: the whole traversal in the tail), clique trees and Bethe graphs, 1-16 traits (packed
    and plain layouts, odd instances), 1-3 sites, injected non-positive-definite blocks (first failure of the reference's
    order)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "fuzz_gpu_vs_c_oracle.py"), "120", "91"],
                         env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "120 cases ok" in out.stdout, (out.stdout[-1500:], out.stderr[-1500:])


def test_loop_launch_modes_are_bitwise_identical(P):
    """The loop launches of the packed layout (pgbp_loop.hip: two wavefronts per record, chains through LDS), pgbp_fast.hip's
    own loop mode (`loop=0`: one wavefront per record) and the chunks packed into two workgroups run the same operations in
    the same order on every entry: the calibrated states and the flags of the fuzz's 120 cases (caterpillars: whole
    traversals in the tail; polytomies; Bethe graphs with prologues; 1-16 traits; injected failures) are BIT FOR BIT the
    same -- one sha256 over all of them per run (tests/fuzz_gpu_vs_c_oracle.py prints it).  (Round 4 used the same check for
    two experiments that are not in the library: tools/rejected/README.md, "helpers" and "strands".)"""
    import os
    import re
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    digests = {}
    for tuning in ("", "loop=0", "chunk_bins=2"):
        env = dict(os.environ)
        env.pop("PGBP_TUNING", None)
        if tuning:
            env["PGBP_TUNING"] = tuning
        out = subprocess.run([sys.executable, os.path.join(here, "fuzz_gpu_vs_c_oracle.py"), "120", "4242"], env=env,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "120 cases ok" in out.stdout, (tuning, out.stdout[-1500:], out.stderr[-1500:])
        digests[tuning] = re.search(r"digest ([0-9a-f]{64})", out.stdout).group(1)
    assert len(set(digests.values())) == 1, digests


def test_auto_stop_is_per_site(P):
    """calibrate!(...; auto=true) on a batch of sites: every site stops at ITS first calibrated schedule tree, as the
    reference run on that site alone would (src/calibration.jl:53-56) -- the device halts a site's later traversals by
    itself, the other sites go on.  Three sites of one loopy Bethe graph with data on different scales: per site the
    (iteration, tree) and the beliefs of the plain-C engine run on that site alone; one site damaged: its failure report,
    while the others still reach calibration."""
    from oracle import cengine
    rng = np.random.default_rng(321)
    net = P.random_level3_network(60, 5, rng, n_colors=2)
    cn, ed, sn = P.bethe(net.node2family)
    p = 2
    st = P.allocate_scopes(cn, ed, sn, net, p)
    pe = [list(zip(net.length[i], net.gamma[i], net.color[i])) for i in range(net.nnodes)]
    fam = P.lg_families(st.clusters, st.node2cluster, net.node2family, st.node2fixed, pe, list(range(net.nnodes)), p, n_rates=2)
    sched = P.spanningtrees_clusterlist(len(cn), ed, cn, net.is_leaf)
    starts = []
    for scale in (1.0, 300.0, 0.004):
        rates = scale * np.stack([np.eye(p) + 0.2, 2 * np.eye(p) + 0.4])
        X = P.simulate_bm_network(net, rates, np.zeros(p), rng)
        one = P.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None)
        one.lg_setup(fam, X)
        one.assignfactors_lg_(rates, np.zeros(p))
        assert P.load().pgbp_regularize_bycluster(one._eng) == 0
        one.pull()
        starts.append(one._packed[0].copy())
        del one
    ns = len(starts)

    def alone(start, niter):
        ce = cengine.Engine(st.dims, st.sepset_clusters.reshape(-1), st.scope_off, st.scope_idx, start)
        for it in range(1, niter + 1):
            for j, spt in enumerate(sched, start=1):
                succ, iscal = ce.calibrate(spt[2], spt[3], 1, return_iscal=True)
                if not succ:
                    return ("failed", it, j), ce
                if iscal:
                    return ("reached", it, j), ce
        return ("no", 0, 0), ce

    want = [alone(s0, 80) for s0 in starts]
    assert all(w[0][0] == "reached" for w in want)
    assert len({w[0][1:] for w in want}) > 1, [w[0] for w in want]   # the sites do stop at different trees
    cgb = P.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, np.stack(starts), n_sites=ns)
    P.calibrate_(cgb, sched, 80, auto=True)
    for s in range(ns):
        r = cgb.last_results[s]
        assert (r.succ, r.iscal) == (1, 1) and (r.iter_reached, r.tree_reached) == want[s][0][1:], (s, want[s][0])
        ref = want[s][1].packed()
        assert np.max(np.abs(cgb._packed[s] - ref)) <= 1e-8 * max(1.0, np.max(np.abs(ref))), s
    # too few iterations for the slowest site: the others are where they stopped, it is (true, false)
    slow = max(range(ns), key=lambda s: want[s][0][1:])
    few = want[slow][0][1] - 1
    if few >= 1:
        cgb._packed[...] = np.stack(starts)
        cgb.push()
        cgb.init_messagecalibrationflags_reset_()
        P.calibrate_(cgb, sched, few, auto=True)
        for s in range(ns):
            r = cgb.last_results[s]
            w = alone(starts[s], few)[0]
            assert (r.succ, bool(r.iscal), r.iter_reached, r.tree_reached) == (1, w[0] == "reached", w[1], w[2]), (s, w)
    # one site damaged: its own first failure, the other sites unaffected
    bad = [s0.copy() for s0 in starts]
    snd = int(next(c for c in sched[0][3] if st.dims[c] >= 2 * p))
    bad[1][cgb._poff[snd]] = -1.0e6
    cgb._packed[...] = np.stack(bad)
    cgb.push()
    cgb.init_messagecalibrationflags_reset_()
    P.calibrate_(cgb, sched, 80, auto=True, verbose=False)
    w1, ce1 = alone(bad[1], 80)
    assert w1[0] == "failed"
    r = cgb.last_results[1]
    assert (r.succ, r.iscal, r.fail_iter, r.fail_tree) == (0, 0, w1[1], w1[2])
    assert (r.fail_edge, r.fail_dir, r.fail_info) == ce1.last_failure()
    for s in (0, 2):
        r = cgb.last_results[s]
        assert (r.succ, r.iscal) == (1, 1) and (r.iter_reached, r.tree_reached) == want[s][0][1:]
        ref = want[s][1].packed()
        assert np.max(np.abs(cgb._packed[s] - ref)) <= 1e-8 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("env", [{}, {"PGBP_TUNING": "no_chunks"}, {"PGBP_TUNING": "mixed_fast_min=0"},
                                 {"PGBP_TUNING": "small4_min=0"}, {"PGBP_TUNING": "small4_min=0,no_chunks"},
                                 {"PGBP_TUNING": "chunk_bins=3"}, {"PGBP_TUNING": "pair=0"},
                                 {"PGBP_TUNING": "pair=0,chunk_bins=3"}],
                         ids=["default", "level_launches_only", "mixed_levels_always_split",
                              "four_tasks_per_wavefront", "four_tasks_per_wavefront_level_launches_only",
                              "chunks_packed_into_3_workgroups", "chunks_one_wavefront_per_task",
                              "chunks_one_wavefront_per_task_packed_into_3_workgroups"])
def test_network_differential_fuzz(P, env):
    """tests/fuzz_gpu_vs_c_oracle_networks.py: random level-3 networks, clique tree / Bethe / join graphs, every spanning
    tree of the schedule, 1 - 9 traits (and 18 - 22), damaged clusters: the wave-per-task kernels (both message bodies,
    level and chunk launches, the large-belief kernel) against the plain-C sequential engine -- beliefs, flags, (succ,
    iscal) and the first failure of the reference's order; once more with every level as its own launch, and with every
    mixed level split into a register-resident and a wave-per-task launch however few fast-class tasks it has (such a level
    must not enter a generic chunk: its fast-class tasks have no message records); and with every level launch of small
    messages, however narrow, on bp_level_small4 (four tasks per wavefront, one per row of 16 lanes: the default only from
    kSmall4MinTasksDefault tasks on, which these small networks never reach), with and without the chunks; and with the trees of
    every chunk's forest packed into 3 workgroups (round 4: the default packs only above 256 trees); and with the chunks of
    small messages on bp_chunk_generic (pair=0: one wavefront per task) instead of the default bp_chunk_pair (round 4: a
    provider and a consumer wavefront per task, hand-over through LDS sequence numbers)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "fuzz_gpu_vs_c_oracle_networks.py"), "60", "17"],
                         env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "60 cases ok" in out.stdout, (out.stdout[-1500:], out.stderr[-1500:])


@pytest.mark.parametrize("argv", [
    ["--ntips", "400", "--traits", "16", "--steps", "2", "--warmup", "1", "--cpu-budget", "0.5"],
    ["--ntips", "300", "--traits", "8", "--graph", "bethe", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
    ["--workload", "sites", "--sites", "40", "--site-traits", "2", "--ntips", "150", "--steps", "2", "--warmup", "1"],
    ["--workload", "sites", "--site-model", "bm", "--sites", "70", "--site-traits", "1", "--ntips", "120", "--steps", "2", "--warmup", "1"],
    ["--workload", "network", "--ntips", "240", "--steps", "2", "--warmup", "1"],
    ["--workload", "network", "--graph", "joingraph", "--ntips", "240", "--steps", "2", "--warmup", "1"],
], ids=["tree", "tree_bethe", "sites_ou", "sites_bm", "network_bethe", "network_joingraph"])
def test_bench_workloads_at_test_size(P, argv):
    """bench.py end to end at test size for every workload: its parity gates (log-likelihood against the independent
    pruning value; network: beliefs and convergence against the C engine) pass and one JSON line with the contract's
    keys comes out."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    line = json.loads(out.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert line["value"] > 0 and line["dtype"] == "f64" and "workload" in line["config"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])


def test_auto_stop_speculative_batches(P):
    """calibrate!(...; auto=true) on one site enqueues several schedule trees per host round trip and lets the device halt
    itself at the first calibrated tree: (1) a loopy Bethe run stops at the tree the C engine stops at for every niter
    around it, with the same beliefs (nothing runs behind the halt); (2) a failure inside a batch is reported as without
    `auto`; (3) niter too small to calibrate -> (true, false), no tree reached."""
    from oracle import cengine
    rng = np.random.default_rng(123)
    net = P.random_level3_network(60, 5, rng, n_colors=2)
    cn, ed, sn = P.bethe(net.node2family)
    p = 2
    st = P.allocate_scopes(cn, ed, sn, net, p)
    rates = np.stack([np.eye(p) + 0.2, 2 * np.eye(p) + 0.4])
    X = P.simulate_bm_network(net, rates, np.zeros(p), rng)
    pe = [list(zip(net.length[i], net.gamma[i], net.color[i])) for i in range(net.nnodes)]
    fam = P.lg_families(st.clusters, st.node2cluster, net.node2family, st.node2fixed, pe, list(range(net.nnodes)), p, n_rates=2)
    cgb = P.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None)
    cgb.lg_setup(fam, X)
    cgb.assignfactors_lg_(rates, np.zeros(p))
    assert P.load().pgbp_regularize_bycluster(cgb._eng) == 0
    cgb.pull()
    start = cgb._packed[0].copy()
    sched = P.spanningtrees_clusterlist(len(cn), ed, cn, net.is_leaf)
    ce = cengine.Engine(st.dims, st.sepset_clusters.reshape(-1), st.scope_off, st.scope_idx, start)
    reached = None
    for it in range(1, 60):
        for j, spt in enumerate(sched, start=1):
            if ce.calibrate(spt[2], spt[3], 1, return_iscal=True)[1]:
                reached = (it, j)
                break
        if reached:
            break
    assert reached and reached[0] >= 3
    ref = ce.packed()
    for niter in (reached[0], reached[0] + 1, reached[0] + 7, 50):
        cgb._packed[0][:] = start
        cgb.push()
        cgb.init_messagecalibrationflags_reset_()
        assert P.calibrate_(cgb, sched, niter, auto=True) == (True, True)
        r = cgb.last_results[0]
        assert (r.iter_reached, r.tree_reached) == reached
        assert np.max(np.abs(cgb._packed[0] - ref)) <= 1e-8 * max(1.0, np.max(np.abs(ref)))
    # (3) not enough iterations
    cgb._packed[0][:] = start
    cgb.push()
    cgb.init_messagecalibrationflags_reset_()
    assert P.calibrate_(cgb, sched, reached[0] - 1, auto=True) == (True, False)
    assert cgb.last_results[0].iter_reached == 0
    # (2) a failure in the third tree of a batch: same report with and without auto
    bad = start.copy()
    snd = int(next(c for c in sched[0][3] if st.dims[c] >= 2 * p))
    bad[cgb._poff[snd]] = -1.0e6
    reports = []
    for auto in (False, True):
        cgb._packed[0][:] = bad
        cgb.push()
        cgb.init_messagecalibrationflags_reset_()
        assert P.calibrate_(cgb, sched, 6, auto=auto, verbose=False) == (False, False)
        r = cgb.last_results[0]
        reports.append((r.fail_iter, r.fail_tree, r.fail_dir, r.fail_edge, r.fail_info))
    assert reports[0] == reports[1] and reports[0][4] > 0


@pytest.mark.parametrize("argv", [
    ["--ntips", "300", "--traits", "16", "--steps", "2", "--warmup", "1"],
    ["--workload", "sites", "--sites", "37", "--site-traits", "3", "--ntips", "120", "--steps", "2", "--warmup", "1"],
    ["--workload", "network", "--ntips", "200", "--steps", "2", "--warmup", "1"],
], ids=["tree_replicas", "sites_sharded", "network_replicas"])
def test_bench_two_ranks_rehearsal(P, argv):
    """bench.py's N > 1 path end to end on the one GPU of this box: 2 ranks launched as the driver launches them
    (torch.distributed.run), both on device 0, gloo in place of RCCL (PGBP_BENCH_REHEARSAL=1): per-rank engines, the
    barrier / MAX timing contract, the site shards and their one all-gather, rank 0's single JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PGBP_BENCH_REHEARSAL="1")
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2"] + argv
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                  # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0
    if argv[:2] == ["--workload", "sites"]:
        assert line["scaling"] == "strong" and line["config"]["problems_per_rank"] in (55, 56)   # 111 problems over 2 ranks
        assert line["loglik_max_rel_err_vs_pruning"] <= 1e-8
    else:
        assert line["scaling"] == "weak"


def test_regularization_and_schedule_doctests_on_the_lipson_network_device(P, caplog):
    """docs/src/man/regularization.md:133-200 and message_schedules.md:55-75 on the device (Lipson et al. network, 44
    nodes, 11 hybrids, Bethe graph; cluster graph and schedule from the product's own host builders): the two error
    lines of the unregularised iteration, none after either regulariser, and "calibration reached: iteration 1, schedule
    tree 2" for beliefs without factors."""
    import os
    from helpers import network_from_newick_file
    g = G["doctests_lipson2020b"]
    net, names, onet, tips = network_from_newick_file(P, os.path.join(os.path.dirname(__file__), "golden", "lipson_2020b.phy"))
    cn, ed, sn = P.bethe(net.node2family)
    labels = ["".join(names[v - 1] for v in c) for c in cn]
    cg = OB.ClusterGraph(list(zip(labels, cn)), [(a, b, s) for (a, b), s in zip(ed, sn)], "Bethe")
    sched = P.spanningtrees_clusterlist(len(cn), ed, cn, net.is_leaf, labels)
    model = make_model(g["model"])
    ocgb, pcgb = build_both(P, onet, cg, model, [g["x_in_tiplabels_order"]], tips)
    with caplog.at_level(logging.ERROR):
        assert P.calibrate_(pcgb, sched) == (False, False)
    errors = [r.getMessage() for r in caplog.records if r.levelno >= logging.ERROR]
    # The reference logs two lines: the postorder of schedule tree 1 stops at its first ill-defined message, then the
    # preorder of the same tree still runs on the partially updated beliefs and stops at another one
    # (src/calibration.jl:80-82).  The engine reports the FIRST failure of the reference's order and stops (the state
    # after a failure is unspecified: a level-synchronous pass cannot stop "mid-level"): the first line, exactly.
    assert errors == g["errors_without_regularization"][:1]
    r = pcgb.last_results[0]
    assert (r.fail_iter, r.fail_tree, r.fail_dir) == (1, 1, 0)
    for regul in (P.regularizebeliefs_bynodesubtree_, P.regularizebeliefs_onschedule_):
        pcgb.init_beliefs_reset_fromfactors_()
        pcgb.init_messagecalibrationflags_reset_()
        caplog.clear()
        regul(pcgb, cg)
        with caplog.at_level(logging.ERROR):
            assert P.calibrate_(pcgb, sched)[0]
        assert not [r for r in caplog.records if r.levelno >= logging.ERROR]
    # beliefs without factors (the setup of the message_schedules.md doctest)
    b, (n2c, n2f, n2fix, n2d, c2n) = OB.allocatebeliefs([g["x_in_tiplabels_order"]], tips, onet, cg, model)
    from helpers import product_beliefs_from_oracle
    p0 = P.ClusterGraphBelief(product_beliefs_from_oracle(b), n2c, n2f, n2fix, c2n)
    P.regularizebeliefs_bynodesubtree_(p0, cg)
    caplog.clear()
    with caplog.at_level(logging.INFO):
        assert P.calibrate_(p0, sched, 100, auto=True, info=True) == (True, True)
    assert g["info_line_without_factors"] in caplog.text


@pytest.mark.parametrize("kind,p", [("tree", 4), ("tree", 3), ("tree", 16), ("bethe", 2), ("tree1", 1)])
def test_residual_norm_flags_at_the_tolerance_boundary(P, kind, p):
    """iscalibrated_residnorm! (src/beliefs.jl:994-1003) on the device is a comparison with thresholds the host derives
    from atol (no division in the kernels): with atol set to a quotient max|dh|/sqrt(s) (or max|dJ|/s) that some message
    attains exactly, and to its floating-point neighbour below, every message's flag equals the reference's expression
    evaluated in numpy on the device's own residuals -- register-resident, wave-per-task and (many sites) univariate
    kernels."""
    import ctypes as C
    from pgbp_amd import _lib as L
    from pgbp_amd import synth as S
    rng = np.random.default_rng(50 + p)
    ns = 9 if kind == "tree1" else 1
    tr = S.random_tree(40, rng)
    R = S.random_rate_matrix(p, rng)
    prob = S.bethe_of_tree(tr, p) if kind == "bethe" else S.cliquetree_of_tree(tr, p)
    packs = []
    for _ in range(ns):
        X = S.simulate_bm(tr, R, np.zeros(p), rng)
        packs.append(S.bm_factors_bethe(tr, prob, R, np.zeros(p), X) if kind == "bethe" else
                     S.bm_factors_cliquetree(tr, prob, R, np.zeros(p), X))
    cgb = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx,
                                           np.stack(packs) if ns > 1 else packs[0], n_sites=ns)
    lib = P.load()
    cgb.set_schedule(prob.schedule)
    res = (L.Result * ns)()
    nm = 2 * cgb.nsepsets
    sdim = np.repeat(np.asarray(prob.dims[cgb.nclusters:], dtype=np.int64), 2)

    def run(atol):
        cgb.init_beliefs_reset_fromfactors_()
        cgb.init_messagecalibrationflags_reset_()
        o = cgb._opts(atol=atol)
        assert lib.pgbp_calibrate(cgb._eng, 1, C.byref(o), res) == 0
        cgb.pull()
        qh, qJ = np.zeros((ns, nm)), np.zeros((ns, nm))
        for site in range(ns):
            for d in range(nm):
                s = int(sdim[d])
                if s == 0:
                    continue
                rec = cgb._res[site, cgb._roff[d]: cgb._roff[d + 1]]
                qJ[site, d] = np.abs(rec[:s * s]).max() / np.sqrt(float(s * s))
                qh[site, d] = np.abs(rec[s * s: s * s + s]).max() / np.sqrt(float(s))
        return qh, qJ, cgb._flg.copy()
    qh, qJ, flg = run(1e-5)
    want = (qh <= 1e-5) & (qJ <= 1e-5)
    assert np.array_equal(flg != 0, want)
    cand = np.unique(np.concatenate([qh[qh > 0], qJ[qJ > 0]]))
    assert len(cand) > 10
    for q in cand[[0, len(cand) // 3, len(cand) // 2, -2]]:
        for atol in (q, np.nextafter(q, 0.0), np.nextafter(q, np.inf)):
            qh2, qJ2, flg2 = run(float(atol))
            assert np.array_equal(qh2, qh) and np.array_equal(qJ2, qJ)            # same messages, same residuals
            assert np.array_equal(flg2 != 0, (qh <= atol) & (qJ <= atol)), atol
    qh0, qJ0, flg0 = run(0.0)
    assert np.array_equal(flg0 != 0, (qh == 0) & (qJ == 0))


@pytest.mark.parametrize("p,where", [(4, "root_end"), (4, "both"), (3, "both"), (20, "both")])
def test_first_failure_in_the_wave_per_task_kernels(P, p, where):
    """A clique tree of a network (generic-class tasks: the register-resident small-message body for p = 3, 4, the in-LDS
    body for p = 20; level launches at the leaf end, chunks of fused levels at the root end): clusters whose J is made
    negative definite fail at their first pivot.  The engine reports the failure the reference's sequential postorder
    meets first (the largest edge index, src/calibration.jl:121) with its PosDefException.info, from whichever launch
    mode ran the message; without the damage the same engine calibrates."""
    import ctypes as C
    from pgbp_amd import _lib as L
    rng = np.random.default_rng(300 + p)
    net = P.random_level3_network_varied(260 if p < 10 else 60, 60 if p < 10 else 12, rng, n_colors=2)
    cn, ed, sn = P.cliquetree(net.node2family)
    st = P.allocate_scopes(cn, ed, sn, net, p)
    base = P.synth.random_rate_matrix(p, rng)
    base = (base + base.T) / 2
    rates = np.stack([base, 2.0 * base])
    X = P.simulate_bm_network(net, rates, np.zeros(p), rng)
    parent_edges = [list(zip(net.length[i], net.gamma[i], net.color[i])) for i in range(net.nnodes)]
    fam = P.lg_families(st.clusters, st.node2cluster, net.node2family, st.node2fixed, parent_edges,
                        list(range(net.nnodes)), p, n_rates=2)
    sched = P.spanningtrees_clusterlist(len(cn), ed, cn, net.is_leaf)
    assert len(sched) == 1
    pa, ch = np.asarray(sched[0][2]), np.asarray(sched[0][3])
    cgb = P.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None)
    cgb.lg_setup(fam, X)
    cgb.assignfactors_lg_(rates, np.zeros(p))
    cgb.pull()
    good = cgb._packed[0].copy()
    cgb.set_schedule(sched)
    assert P.calibrate_(cgb, sched, 2) == (True, True)
    # heights in the schedule tree; senders that integrate something (their cluster is larger than the sepset above it)
    nc = len(cn)
    h = np.zeros(nc, np.int64)
    for a, c in zip(pa[::-1].tolist(), ch[::-1].tolist()):
        h[a] = max(h[a], h[c] + 1)
    dims = np.asarray(st.dims)
    sep_of = {}
    for k, (a, b) in enumerate(np.asarray(st.sepset_clusters).reshape(-1, 2).tolist()):
        sep_of[(a, b)] = sep_of[(b, a)] = k
    cand = [i for i in range(len(pa)) if dims[ch[i]] > dims[nc + sep_of[(int(pa[i]), int(ch[i]))]] > 0]
    near_root = max(cand, key=lambda i: (h[ch[i]], -i))
    hmin = min(h[ch[i]] for i in cand)
    leaf_end = max(i for i in cand if h[ch[i]] == hmin)
    bad = [near_root] if where == "root_end" else [near_root, leaf_end]
    assert p >= 10 or h[ch[near_root]] >= 6, "the damaged cluster near the root sits in the narrow levels"
    poff = cgb._poff
    packed = good.copy()
    for i in bad:
        c = int(ch[i]); m = int(dims[c])
        J = packed[poff[c]: poff[c] + m * m].reshape(m, m)
        J[np.arange(m), np.arange(m)] = -1.0e6           # every pivot candidate negative: info = 1 whatever is integrated
    cgb._packed[0][:] = packed
    cgb._upload(True)                                   # (also the factors: calibrate resets nothing here, but keep both alike)
    cgb.init_messagecalibrationflags_reset_()
    assert P.calibrate_(cgb, sched, 1, verbose=False) == (False, False)
    r = cgb.last_results[0]
    assert (r.fail_dir, r.fail_info, r.fail_edge) == (0, 1, max(bad))
    # and the engine recovers: the undamaged beliefs calibrate again
    cgb._packed[0][:] = good
    cgb._upload(True)
    cgb.init_messagecalibrationflags_reset_()
    assert P.calibrate_(cgb, sched, 2) == (True, True)


def test_lazy_write_back_equals_the_eager_pull_and_keeps_aliases(P):
    """calibrate! updates the beliefs IN PLACE (src/calibration.jl:35-84; SURVEY.md section 8(b): callers hold aliases, e.g.
    test_calibration.jl:141-161 reads b[6] after calibrating).  The host mirror moves nothing at the call: a belief's record
    comes over at its first read (pgbp_get_belief), a residual's at its first read (pgbp_get_residual), the flag vectors on
    demand.  Here: an alias taken BEFORE the call, a fresh index, a residual, a flag -- each equal, bit for bit, to what
    the eager pull of a second engine on the same inputs holds; then an edit on the host followed by push() uploads the
    edited record and nothing stale; a walk over every belief falls back to one transfer."""
    from pgbp_amd import synth as S
    rng = np.random.default_rng(11)
    p = 3
    tr = S.random_tree(60, rng)
    R = S.random_rate_matrix(p, rng)
    X = S.simulate_bm(tr, R, np.zeros(p), rng)
    prob = S.cliquetree_of_tree(tr, p)
    packed = S.bm_factors_cliquetree(tr, prob, R, np.zeros(p), X)
    mk = lambda: P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packed)
    lazy, eager = mk(), mk()
    eager.lazy = False
    alias = lazy.belief[5]                       # held across the call
    before = np.array(alias.J, copy=True)
    assert P.calibrate_(lazy, prob.schedule, 1)[0] and P.calibrate_(eager, prob.schedule, 1)[0]
    assert lazy._stale is not None and lazy._stale.all() and eager._stale is None
    E = eager._packed_raw[0]
    o = lazy._poff
    m = int(prob.dims[5])
    assert np.array_equal(np.asarray(alias.J).reshape(-1, order="F"), E[o[5]: o[5] + m * m]) and not np.array_equal(alias.J, before)
    assert lazy._stale.sum() == lazy.nbeliefs - 1          # one record moved, nothing else
    b9 = lazy.belief[9]
    assert np.array_equal(b9.h, eager.belief[9].h) and np.array_equal(b9.g, eager.belief[9].g)
    sc0 = np.asarray(prob.sepset_clusters).reshape(-1, 2)[0]
    key = (int(sc0[0]), int(sc0[1]))
    mr, mre = lazy.messageresidual[key], eager.messageresidual[key]
    assert np.array_equal(mr.dJ, mre.dJ) and np.array_equal(mr.dh, mre.dh) and mr.iscalibrated_resid == mre.iscalibrated_resid
    assert lazy._res_have.sum() == 1
    assert lazy.iscalibrated_residnorm() == eager.iscalibrated_residnorm()
    _, ll = lazy.integratebelief_(prob.root_cluster)
    assert rel_close(ll, S.bm_loglik_pruning(tr, R, np.zeros(p), X))
    # an edit on the host, then push(): the stale records are fetched first, the edit goes up, nothing old overwrites the device
    lazy.belief[9].h[0] += 1.0
    lazy.push()
    assert lazy._stale is None
    lazy.pull()
    want = eager._packed_raw[0].copy()
    want[o[9] + int(prob.dims[9]) ** 2] += 1.0
    assert np.array_equal(lazy._packed_raw[0], want)
    # a caller that walks over every belief: the mirror switches to one transfer
    assert P.calibrate_(lazy, prob.schedule, 1)[0]
    for i in range(lazy.nbeliefs):
        lazy.belief[i].g
    assert lazy._stale is None


def test_fused_calibration_flag_of_site_minor_batches(P):
    """iscalibrated_residnorm(beliefs) (src/clustergraphbeliefs.jl:168-169) of a univariate site batch in the site-minor layout
    (>= 64 sites): after a postorder + preorder pair over a tree that holds every sepset the engine takes it from the marks
    the message kernels set (DevState::notcal) instead of reducing the 2 n_sepsets x n_sites flag array.  It must equal the
    AND of the flags per site -- some sites calibrated, others not (their data moved between two calibrations), one site
    with a damaged cluster (failed: (false, false)) -- and calibrate!(...; auto = true) must stop every site where the C
    engine run on that site alone stops it."""
    from pgbp_amd import synth as S
    from pgbp_amd import _lib as L
    import ctypes as C
    rng = np.random.default_rng(77)
    ns = 96
    tr = S.random_tree(40, rng)
    prob = S.cliquetree_of_tree(tr, 1)
    sig = rng.uniform(0.5, 2.0, size=ns)
    mu = rng.normal(size=ns)
    X = S.simulate_bm_uni_sites(tr, sig, mu, rng)
    cgb = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, None, n_sites=ns)
    cgb.set_schedule(prob.schedule)
    cgb.bm_tree_setup(*S.bm_tree_table(tr, prob), X[:, :, None])
    cgb.assignfactors_bm_(sig[:, None, None], mu[:, None])
    lib = P.load()
    res = (L.Result * ns)()
    o = cgb._opts()
    assert lib.pgbp_calibrate(cgb._eng, 1, C.byref(o), res) == 0
    nm = 2 * cgb.nsepsets

    def flags_and():
        f = np.zeros((ns, nm), np.int32)
        assert lib.pgbp_get_residuals(cgb._eng, None, L.i32p(f), None, None) == 0
        return f.all(axis=1)
    assert all(r.succ for r in res) and [bool(r.iscal) for r in res] == list(flags_and()) and not any(r.iscal for r in res)
    assert lib.pgbp_calibrate(cgb._eng, 1, C.byref(o), res) == 0      # a tree: calibrated after the second pair
    assert [bool(r.iscal) for r in res] == list(flags_and()) and all(r.iscal for r in res)
    # move the beliefs of every third site (new factors for them would do the same: here, one cluster's h), calibrate once:
    # those sites are not calibrated in that pair, the others stay so
    cgb.pull()
    moved = np.arange(ns) % 3 == 0
    i0 = int(np.argmax(prob.dims[:cgb.nclusters]))
    for s_ in np.nonzero(moved)[0]:
        J, h, g = cgb._views(int(s_), i0)
        h[0] += 0.5
    cgb.push()
    assert lib.pgbp_calibrate(cgb._eng, 1, C.byref(o), res) == 0
    got = np.array([bool(r.iscal) for r in res])
    assert np.array_equal(got, flags_and()) and not got[moved].any() and got[~moved].all()
    # a damaged site fails alone; auto stops every other site at its own first calibrated tree
    cgb.pull()
    J, h, g = cgb._views(5, i0)
    J[0, 0] = -abs(J[0, 0]) - 1.0
    cgb.push()
    oa = cgb._opts(auto=True)
    assert lib.pgbp_calibrate(cgb._eng, 6, C.byref(oa), res) == 0
    assert not res[5].succ and not res[5].iscal and res[5].fail_info > 0
    ok = [s_ for s_ in range(ns) if s_ != 5]
    assert all(res[s_].succ and res[s_].iscal for s_ in ok)
    assert {(res[s_].iter_reached, res[s_].tree_reached) for s_ in ok if not moved[s_]} == {(1, 1)}
    assert {(res[s_].iter_reached, res[s_].tree_reached) for s_ in ok if moved[s_]} <= {(1, 1), (2, 1)}
    assert np.array_equal(flags_and()[ok], np.ones(len(ok), bool))
