"""CPU tests: the plain-C sequential oracle (oracle/c) against the numpy restatement, the golden
values that pass through it, and the independent pruning / dense-MVN log-likelihoods."""
import numpy as np
import pytest

import pgbp_amd  # host-side synthetic generators only (no GPU needed)
from pgbp_amd import synth as S

from helpers import goldens, make_model, oracle_cgb_from_problem, oracle_schedule, oracle_setup, pack_oracle
from oracle import beliefs as OB
from oracle import calibration as OC
from oracle import cengine
from oracle import clustergraph as OCG
from oracle import network as ON

G = goldens()


def engine_from_oracle(ocgb):
    """C engine with the scopes/values of an oracle ClusterGraphBelief."""
    b = ocgb.belief
    nc = ocgb.nclusters
    dims = [x.dimension for x in b]
    sepcl, off, idx = [], [0], []
    for j in range(nc, len(b)):
        a, c = (ocgb.cdict[l] for l in b[j].metadata)
        sepcl += [a, c]
        for cl in (a, c):
            ind = OB.scopeindex(b[j], b[cl])
            idx += ind.tolist()
            off.append(off[-1] + len(ind))
    packed = np.concatenate([np.concatenate([x.J.reshape(-1, order="F"), x.h, x.g]) for x in b])
    return cengine.Engine(dims, sepcl, off, idx, packed)


@pytest.mark.parametrize("key,traits,llkey", [
    ("calibration_cliquetree_level1", ["y"], "ll_every_belief"),
    ("calibration_tree_2traits_missing", ["y1", "y2"], "ll_every_belief"),
    ("doctest_lazaridis", ["x"], "ll"),
    ("exactBM_tree_calibrate", ["y"], "ll"),
])
def test_c_oracle_goldens(key, traits, llkey):
    g = G[key]
    net = ON.read_newick(g["net"])
    ct = OCG.cliquetree(net)
    spt = OCG.spanningtree_clusterlist(ct, OCG.default_rootcluster(ct, net))
    ocgb = oracle_setup(net, ct, make_model(g["model"]), [g[t] for t in traits], g["taxa"])
    eng = engine_from_oracle(ocgb)
    succ, iscal = eng.calibrate(spt[2], spt[3], 1, return_iscal=True)
    osucc, oiscal = OC.calibrate(ocgb, [spt])
    assert succ and osucc and iscal == oiscal
    tol = g.get("atol", None)
    for i in range(len(ocgb.belief)):
        mu, n = eng.integrate(i)
        if tol:
            assert abs(n - g[llkey]) <= tol
        else:
            assert abs(n - g[llkey]) <= 1.5e-8 * abs(g[llkey])
    ref = np.concatenate([np.concatenate([x.J.reshape(-1, order="F"), x.h, x.g]) for x in ocgb.belief])
    assert np.allclose(eng.packed(), ref, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("ntips,p", [(2, 1), (7, 1), (30, 2), (64, 4), (40, 16)])
def test_c_oracle_vs_numpy_random_trees(ntips, p):
    rng = np.random.default_rng(7 * ntips + p)
    tr = S.random_tree(ntips, rng)
    R = S.random_rate_matrix(p, rng)
    mu = rng.standard_normal(p)
    X = S.simulate_bm(tr, R, mu, rng)
    prob = S.cliquetree_of_tree(tr, p)
    packed = S.bm_factors_cliquetree(tr, prob, R, mu, X)
    eng = cengine.Engine(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packed)
    ocgb = oracle_cgb_from_problem(prob, packed, p)
    pa, ch = prob.schedule[0]
    assert eng.calibrate(pa, ch, 2, return_iscal=True) == OC.calibrate(ocgb, [oracle_schedule(prob)], 2) == (True, True)
    assert np.allclose(eng.packed(), pack_oracle(ocgb, prob), rtol=1e-9, atol=1e-9)
    ll = eng.integrate(prob.root_cluster)[1]
    assert abs(ll - S.bm_loglik_pruning(tr, R, mu, X)) <= 1e-9 * max(1, abs(ll))
    res, flags = eng.residuals()
    oflags = []
    for k, (a, c) in enumerate(prob.sepset_clusters):
        oflags += [ocgb.messageresidual[(int(a), int(c))].iscalibrated_resid,
                   ocgb.messageresidual[(int(c), int(a))].iscalibrated_resid]
    assert flags.astype(bool).tolist() == oflags


def test_c_oracle_failure_semantics():
    """non-PD block: stops at the first failing message in sequential order, nothing applied."""
    rng = np.random.default_rng(3)
    tr = S.random_tree(12, rng)
    p = 2
    prob = S.cliquetree_of_tree(tr, p)
    R = S.random_rate_matrix(p, rng)
    X = S.simulate_bm(tr, R, np.zeros(p), rng)
    packed = S.bm_factors_cliquetree(tr, prob, R, np.zeros(p), X)
    pa, ch = prob.schedule[0]
    senders = [i for i in range(len(pa)) if prob.dims[ch[i]] == 2 * p]
    i_bad = senders[1]
    packed[prob.packed_off[ch[i_bad]]] = -1e6
    eng = cengine.Engine(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packed)
    assert eng.calibrate(pa, ch, 1) is False
    assert eng.last_failure() == (i_bad, 0, 1)
