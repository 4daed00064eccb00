"""Several GPUs behind the C ABI (include/pgbp.h "several GPUs", csrc/pgbp_dist.hip), rehearsed on ONE GPU:
pgbp_group with the same device listed twice / three times (one engine + stream + host thread per shard), pgbp_comm
with a single rank (ncclCommInitRank + ncclAllGather of one rank through the dlopen'd RCCL)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgbp_amd  # noqa: E402
from pgbp_amd import _lib as L  # noqa: E402
from pgbp_amd import synth as S  # noqa: E402
from pgbp_amd.sharding import Comm, EngineGroup, shard_range  # noqa: E402


def _problem(ntips, p, ns, seed):
    rng = np.random.default_rng(seed)
    tr = S.random_tree(ntips, rng)
    prob = S.cliquetree_of_tree(tr, p)
    packs, lls = [], []
    for s in range(ns):
        R = S.random_rate_matrix(p, rng)
        mu = rng.standard_normal(p)
        X = S.simulate_bm(tr, R, mu, rng)
        packs.append(S.bm_factors_cliquetree(tr, prob, R, mu, X))
        lls.append(S.bm_loglik_pruning(tr, R, mu, X))
    return tr, prob, np.stack(packs), np.array(lls)


def test_group_needs_a_device_and_valid_arguments():
    """No CPU fallback behind the group either; bad arguments are refused with a message."""
    lib = pgbp_amd.load()
    tr, prob, packs, _ = _problem(6, 2, 3, 0)
    desc, keep = L.make_desc(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, 3, 0)
    g = C.c_void_p()
    dev = np.zeros(4, np.int32)
    assert lib.pgbp_group_create(C.byref(desc), 4, L.i32p(dev), C.byref(g)) == 1      # more shards than sites
    assert b"n_devices" in lib.pgbp_group_last_error(None)
    import torch
    if not torch.cuda.is_available():
        code = lib.pgbp_group_create(C.byref(desc), 2, L.i32p(dev), C.byref(g))
        assert code == 5 and not g.value, code                                          # PGBP_ERR_NO_DEVICE
        assert b"no CPU path" in lib.pgbp_group_last_error(None)


def test_patterns_handle_checks_its_arguments():
    """pgbp_patterns_create: the site list must be a permutation of the caller's site indices, every pattern the same
    cluster graph; no CPU fallback behind this handle either."""
    lib = pgbp_amd.load()
    tr, prob, packs, _ = _problem(6, 2, 3, 0)
    d0, k0 = L.make_desc(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, 2, 0)
    d1, k1 = L.make_desc(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, 1, 0)
    arr = (C.POINTER(L.Desc) * 2)(C.pointer(d0), C.pointer(d1))
    h = C.c_void_p()
    bad = np.array([0, 1, 1], np.int32)
    assert lib.pgbp_patterns_create(2, arr, L.i32p(bad), C.byref(h)) == 1 and not h.value
    assert b"permutation" in lib.pgbp_patterns_last_error(None)
    tr2, prob2, _, _ = _problem(7, 2, 1, 1)                       # another cluster graph
    d2, k2 = L.make_desc(prob2.dims, prob2.sepset_clusters, prob2.scope_off, prob2.scope_idx, 1, 0)
    arr2 = (C.POINTER(L.Desc) * 2)(C.pointer(d0), C.pointer(d2))
    assert lib.pgbp_patterns_create(2, arr2, L.i32p(np.array([2, 0, 1], np.int32)), C.byref(h)) == 1
    assert b"same cluster graph" in lib.pgbp_patterns_last_error(None)
    import torch
    if not torch.cuda.is_available():
        code = lib.pgbp_patterns_create(2, arr, L.i32p(np.array([2, 0, 1], np.int32)), C.byref(h))
        assert code == 5 and not h.value                                                # PGBP_ERR_NO_DEVICE


@pytest.mark.gpu
@pytest.mark.parametrize("ntips,p,ns,nshards", [(40, 16, 5, 2), (25, 3, 7, 3), (60, 1, 70, 2)])
def test_group_equals_single_engine(ntips, p, ns, nshards):
    """A group of shards on device 0 = the single engine over all sites: same calibrated beliefs bit for bit, same
    (succ, iscal), same log-likelihoods, shard ranges = shard_range()."""
    tr, prob, packs, lls = _problem(ntips, p, ns, 100 + ntips)
    one = pgbp_amd.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx,
                                                  packs, n_sites=ns)
    assert pgbp_amd.calibrate_(one, prob.schedule, 2) == (True, True)
    grp = EngineGroup(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, ns, [0] * nshards)
    assert grp.size == nshards
    for i in range(nshards):
        lo, hi = shard_range(ns, i, nshards)
        assert grp.range(i) == (lo, hi - lo)
    grp.set_schedule(prob.schedule)
    grp.set_beliefs(packs)
    res = grp.calibrate(2)
    assert all(r.succ == 1 and r.iscal == 1 for r in res)
    assert np.array_equal(grp.get_beliefs(), np.stack(one._packed))
    mu, norm, info = grp.integrate(prob.root_cluster, int(prob.dims[prob.root_cluster]))
    assert not info.any()
    assert np.all(np.abs(norm - lls) <= 1e-8 * np.maximum(1.0, np.abs(lls)))
    # the zero-copy path: reset + postorder + root integrate on every shard, one fetch
    grp.enqueue_loglik(2)
    norm2, info2 = grp.fetch_loglik()
    assert not info2.any() and np.allclose(norm2, norm, rtol=1e-12, atol=0)   # (root after the postorder alone)
    grp.enqueue_calibrate(1, reset_each=1)
    grp.sync()
    one2 = pgbp_amd.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx,
                                                   packs, n_sites=ns)
    pgbp_amd.calibrate_(one2, prob.schedule, 1)
    assert np.array_equal(grp.get_beliefs(), np.stack(one2._packed))
    grp.close()


@pytest.mark.gpu
def test_group_reports_the_failing_shard_sites():
    """A non-positive-definite block in one site of the second shard: that site's result carries the failure, every
    other site (both shards) calibrates."""
    tr, prob, packs, _ = _problem(30, 4, 5, 7)
    big = [i for i in range(prob.nclusters) if prob.dims[i] > 0]
    packs[3, prob.packed_off[big[2]]] = -50.0
    grp = EngineGroup(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, 5, [0, 0])
    grp.set_schedule(prob.schedule)
    grp.set_beliefs(packs)
    res = grp.calibrate(2)
    assert [r.succ for r in res] == [1, 1, 1, 0, 1]
    assert res[3].fail_info > 0 and res[3].fail_tree == 1
    one = pgbp_amd.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx,
                                                  packs, n_sites=5)
    pgbp_amd.calibrate_(one, prob.schedule, 2, verbose=False)
    r1 = one.last_results[3]
    assert (res[3].fail_iter, res[3].fail_dir, res[3].fail_edge, res[3].fail_info) == (r1.fail_iter, r1.fail_dir, r1.fail_edge, r1.fail_info)
    grp.close()


@pytest.mark.gpu
def test_comm_single_rank_gather():
    """pgbp_comm through the dlopen'd RCCL with one rank: unique id, ncclCommInitRank, ONE ncclAllGather carrying
    log-likelihoods, info words and the (succ, iscal) minimum."""
    tr, prob, packs, lls = _problem(35, 8, 6, 11)
    cgb = pgbp_amd.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx,
                                                  packs, n_sites=6)
    cgb.set_schedule(prob.schedule)
    lib = pgbp_amd.load()
    opts = cgb._opts()
    comm = Comm(1, 0, 0)
    assert lib.pgbp_enqueue_calibrate(cgb._eng, 2, 0, C.byref(opts)) == 0
    assert lib.pgbp_enqueue_loglik(cgb._eng, 1, C.byref(opts)) == 0
    norm, info, succ, iscal = comm.gather_loglik(cgb._eng, 8)      # slot larger than the rank's 6 sites
    assert norm.shape == (1, 8) and not info.any() and succ
    assert np.all(np.abs(norm[0, :6] - lls) <= 1e-8 * np.maximum(1.0, np.abs(lls))) and not norm[0, 6:].any()
    ref = np.zeros(6)
    assert lib.pgbp_fetch_loglik(cgb._eng, L.f64p(ref), None) == 0
    assert np.array_equal(ref, norm[0, :6])
    with pytest.raises(L.PgbpError):
        comm.gather_loglik(cgb._eng, 3)                            # slot smaller than the rank's sites
    comm.close()


@pytest.mark.gpu
def test_nodesubtree_regulariser_on_plain_arrays_equals_the_object_walk():
    """regularization.py:regularizebeliefs_bynodesubtree_arrays_ (indexed once, linear time: what the 50 000-node network
    of BASELINE configs[4] needs) edits exactly what the per-node search of regularizebeliefs_bynodesubtree_ edits
    (src/clustergraphbeliefs.jl:306-340), on a loopy join graph of a network with varied level-3 blobs."""
    import pgbp_amd as P
    from pgbp_amd.regularization import regularizebeliefs_bynodesubtree_, regularizebeliefs_bynodesubtree_arrays_
    rng = np.random.default_rng(9)
    net = P.random_level3_network_varied(120, 40, rng, n_colors=2)
    cn, ed, sn = P.joingraph(net.node2family, 3)
    assert len(ed) > len(cn) - 1                     # genuinely loopy
    p = 2
    st = P.allocate_scopes(cn, ed, sn, net, p)
    rates = np.stack([np.eye(p) + 0.2, 2 * np.eye(p) + 0.4])
    X = P.simulate_bm_network(net, rates, np.zeros(p), rng)
    pe = [list(zip(net.length[i], net.gamma[i], net.color[i])) for i in range(net.nnodes)]
    fam = P.lg_families(st.clusters, st.node2cluster, net.node2family, st.node2fixed, pe, list(range(net.nnodes)), p, n_rates=2)
    out = []
    for which in (0, 1):
        cgb = P.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None)
        cgb.lg_setup(fam, X)
        cgb.assignfactors_lg_(rates, np.zeros(p))
        cgb.pull()
        if which == 0:
            regularizebeliefs_bynodesubtree_arrays_(cgb, cn, ed, sn, st)
        else:
            class Obj:
                pass
            objs = []
            for sc in st.clusters:
                o = Obj(); o.nodelabel = list(sc.nodelabel); o.inscope = sc.inscope; objs.append(o)
            insc = {lab: sc.inscope[:, j] for sc in st.clusters for j, lab in enumerate(sc.nodelabel)}
            for nodes in sn:
                o = Obj(); o.nodelabel = list(nodes); o.inscope = np.stack([insc[v] for v in nodes], axis=1); objs.append(o)
            cgb._objs = objs
            regularizebeliefs_bynodesubtree_(cgb)
            cgb._objs = None
        cgb.pull()
        out.append(cgb._packed[0].copy())
    assert np.array_equal(out[0], out[1])
    start = P.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None)
    start.lg_setup(fam, X); start.assignfactors_lg_(rates, np.zeros(p)); start.pull()
    assert not np.array_equal(out[0], start._packed[0])


@pytest.mark.gpu
@pytest.mark.parametrize("p", [2, 1, 3, 4, 6])
def test_muller_clique_tree_with_beliefs_beyond_64_dimensions(p):
    """The reference's documented clique tree of the Mueller et al. network (docs/src/man/clustergraphs.md:40-89: 664
    cliques, the largest holds 54 nodes) with p = 2 traits: beliefs of up to 108 variables, sepsets of up to 106 --
    beyond the 64 the wave-per-task kernel's lane grids hold, on bp_level_big (a workgroup per task, the sender in up to
    132 KB of LDS).  calibrate!() against the plain-C sequential engine: every belief to 1e-8 * max|.|, every residual
    flag; the log-likelihood is the same at every belief (exact on a clique tree); free_energy (blocked right-hand
    sides above dimension 96) equals minus the log-likelihood; a single pgbp_propagate of the largest message; p = 1
    (54 dimensions) runs the same graph on the wave-per-task kernel; p = 3 and 4: beliefs of 162 and 216 variables, beyond
    the 128 a CU's LDS holds: the working matrix of bp_level_big, of integratebelief!, of free_energy and the accumulator of
    the device factor fill then live in global memory (the workspace variants); p = 6: 324 variables, the reference's
    documented graph at six traits (PGBP_MAX_DIM = 384; receivers of more than 254 variables run on bp_level_big whatever
    their senders).  No size the reference accepts is refused: free_energy equals minus the log-likelihood at every p."""
    import pgbp_amd as P
    from oracle import cengine
    path = os.path.join(ROOT, "tests", "golden", "muller_2022.phy")
    net, names = P.read_newick(open(path).read())
    cn, ed, sn = P.cliquetree(net.node2family)
    assert len(cn) == 664 and max(len(c) for c in cn) == 54
    st = P.allocate_scopes(cn, ed, sn, net, p)
    assert st.dims.max() >= 50 * p           # (tips and the fixed root are out of scope)
    rng = np.random.default_rng(2)
    rates = np.stack([np.eye(p) + 0.3])
    X = P.simulate_bm_network(net, rates, np.zeros(p), rng)
    pe = [list(zip(net.length[i], net.gamma[i], net.color[i])) for i in range(net.nnodes)]
    fam = P.lg_families(st.clusters, st.node2cluster, net.node2family, st.node2fixed, pe, list(range(net.nnodes)), p, n_rates=1)
    cgb = P.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None)
    cgb.lg_setup(fam, X)
    cgb.assignfactors_lg_(rates, np.zeros(p))
    cgb.pull()
    start = cgb._packed[0].copy()
    root = P.default_rootcluster(cn, net.is_leaf)
    spt = P.spanningtree_clusterlist(len(cn), ed, root)
    sched = [(np.asarray(spt[2]), np.asarray(spt[3]))]
    assert P.calibrate_(cgb, sched, 2) == (True, True)
    ce = cengine.Engine(st.dims, st.sepset_clusters.reshape(-1), st.scope_off, st.scope_idx, start)
    assert ce.calibrate(sched[0][0], sched[0][1], 2, return_iscal=True) == (True, True)
    got, ref = cgb._packed[0], ce.packed()
    off = cgb._poff
    worst = 0.0
    for i in range(len(st.dims)):
        a, b = got[off[i]:off[i + 1]], ref[off[i]:off[i + 1]]
        if a.size:
            worst = max(worst, float(np.max(np.abs(a - b))) / max(1.0, float(np.max(np.abs(b)))))
    assert worst <= 1e-8, worst
    assert np.array_equal(cgb._flags().astype(bool), ce.residuals()[1].astype(bool))
    ll = ce.integrate(root)[1]
    big = int(np.argmax(st.dims[:len(cn)]))
    for i in (root, big, 0, len(cn) - 1, len(cn) + 5):
        if st.dims[i] > 0:
            v = cgb.integratebelief_(int(i))[1]
            assert abs(v - ll) <= 1e-8 * max(1.0, abs(ll)), (i, v, ll)
    fe = cgb.free_energy()      # (beliefs above 96 variables: the workspace instance of the kernel)
    assert abs(fe[2] + ll) <= 1e-8 * max(1.0, abs(ll)), (fe, ll)
    # one message on its own, from the start state: the largest sender towards one of its neighbours
    cgb._packed[0][:] = start
    cgb.push()
    k = next(k for k, (a, b) in enumerate(ed) if big in (a, b))
    a, b = ed[k]
    to = b if a == big else a
    assert cgb._propagate(to, len(cn) + k, big, sync=True) is None
    ce2 = cengine.Engine(st.dims, st.sepset_clusters.reshape(-1), st.scope_off, st.scope_idx, start)
    assert ce2.propagate(to, k, big) == 0
    r2 = ce2.packed()
    for i in (to, len(cn) + k):
        x, y = cgb._packed[0][off[i]:off[i + 1]], r2[off[i]:off[i + 1]]
        assert float(np.max(np.abs(x - y))) <= 1e-8 * max(1.0, float(np.max(np.abs(y))))


@pytest.mark.gpu
def test_three_missing_data_patterns_twelve_sites_behind_one_handle():
    """pgbp_patterns (include/pgbp.h): sites whose tips miss different traits have different scopes (allocatebeliefs,
    src/beliefs.jl:551-559) -- one engine per pattern behind one handle, results in the caller's site order.  A network
    with hybrid nodes, one trait (test/test_exactBM.jl:228-251 is the reference's own case of this kind; with several
    correlated traits the reference's assignfactors! itself fails its Cholesky on the numerically-zero precision a fully
    missing subtree leaves, src/beliefs.jl:842 "fixit"), 12 sites in 3 missing-data patterns -- complete; missing at three
    scattered tips; missing at both tips of a cherry, so that their parent drops out of scope -- interleaved.  Every
    site against the ORACLE run on that site alone (allocatebeliefs + assignfactors! + calibrate! on its own scopes): the
    postorder log-likelihood (device factor fill with scope masks, 1e-8), the dense multivariate-normal likelihood, and
    after calibrate!() the integral of the root cluster; the per-site (succ, iscal)."""
    import zlib
    import pgbp_amd as P
    from helpers import lg_inputs_from_oracle, oracle_setup, product_beliefs_from_oracle
    from oracle import calibration as OC
    from oracle import clustergraph as OCG
    from oracle import densemvn as OD
    from oracle import models as OM
    from oracle import network as ON
    from pgbp_amd.sharding import PatternGroup
    rng = np.random.default_rng(zlib.crc32(b"patterns"))
    p, ntips = 1, 12
    net = ON.random_network(ntips, 3, rng)
    taxa = net.tip_names
    model = OM.UnivariateBrownianMotion(1.7, 0.4)
    ct = OCG.cliquetree(net)
    spt = OCG.spanningtree_clusterlist(ct, OCG.default_rootcluster(ct, net))
    # a cherry (two tips with the same parent) for pattern 2
    par = {}
    for e in net.edges:
        if e.child.leaf and not e.child.hybrid:
            par.setdefault(id(e.parent), []).append(taxa.index(e.child.name))
    cherry = next((v for v in par.values() if len(v) == 2), [0, 1])
    others = [i for i in range(ntips) if i not in cherry]

    def mask_of(k):
        m = np.zeros((ntips, p), bool)         # True = missing
        if k == 1:
            m[others[0], 0] = m[others[1], 0] = m[others[2], 0] = True
        elif k == 2:
            m[cherry[0], :] = True
            m[cherry[1], :] = True
        return m
    n_sites = 12
    pattern_of = [0, 1, 2, 2, 1, 0, 1, 1, 2, 0, 2, 0]        # interleaved
    X = rng.normal(size=(n_sites, ntips, p))
    tbls = []
    for s in range(n_sites):
        m = mask_of(pattern_of[s])
        tbls.append([[None if m[r, t] else float(X[s, r, t]) for r in range(ntips)] for t in range(p)])
    # the oracle, site by site
    want_ll, want_root = [], []
    ocgbs = {}
    for s in range(n_sites):
        ocgb = oracle_setup(net, ct, model, tbls[s], taxa)
        ocgbs.setdefault(pattern_of[s], (s, ocgb))
        import copy
        o2 = copy.deepcopy(ocgb)
        assert OC.propagate_1traversal_postorder(o2, *spt)
        ll = o2.integratebelief(spt[2][0])[1]
        dense = OD.loglik(net, model, tbls[s], taxa)
        assert abs(ll - dense) <= 1e-8 * max(1.0, abs(dense))
        want_ll.append(ll)
    # the product: one description per pattern, from the oracle's scopes of one of its sites
    patterns, fams = [], []
    for k in range(3):
        s0, ocgb = ocgbs[k]
        pb = product_beliefs_from_oracle(ocgb.belief)
        host = P.ClusterGraphBelief(pb, ocgb.node2cluster, ocgb.node2family, ocgb.node2fixed, ocgb.cluster2nodes)
        sites = [s for s in range(n_sites) if pattern_of[s] == k]
        patterns.append((host._dims.copy(), host._sepcl.reshape(-1).copy(), host._scope_off.copy(), host._scope_idx.copy(), sites))
        fam, _, kw = lg_inputs_from_oracle(P, net, ocgb, model, tbls[s0], taxa)
        data = np.stack([np.where(mask_of(k), np.nan, X[s]) for s in sites])
        fams.append((fam, data, kw))
        del host
    dims_by_pattern = [tuple(int(x) for x in pt[0]) for pt in patterns]
    assert len(set(dims_by_pattern)) >= 2                     # the cherry pattern has other scopes than the complete one
    grp = PatternGroup(patterns)
    grp.set_schedule([spt])
    for k in range(3):
        fam, data, kw = fams[k]
        grp.beliefs[k].lg_setup(fam, data)
        grp.beliefs[k].assignfactors_lg_(**kw)
    ll, info = grp.loglik_lg()
    assert not info.any()
    assert np.max(np.abs(ll - np.array(want_ll)) / np.maximum(1.0, np.abs(want_ll))) <= 1e-8
    # calibrate!() of every site from its factors: (succ, iscal) per site, the root cluster integrates to the likelihood
    for k in range(3):
        assert grp.lib.pgbp_reset_from_factors(grp.beliefs[k]._eng) == 0
    res = grp.calibrate(2)
    assert all(res[s].succ == 1 and res[s].iscal == 1 for s in range(n_sites))
    norm, info = grp.integrate(spt[2][0])
    assert not info.any()
    assert np.max(np.abs(norm - np.array(want_ll)) / np.maximum(1.0, np.abs(want_ll))) <= 1e-8
    grp.close()


def _small_network(graph, ntips=600, seed=3, p=2):
    import argparse
    import bench as B
    args = argparse.Namespace(seed=seed, traits=p, blob_style="varied", ntips=ntips, blobs=ntips // 12, graph=graph,
                              maxclustersize=3)
    return B.build_network_workload(args, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("graph,n_ranks", [("bethe", 2), ("cliquetree", 3), ("joingraph", 2)])
def test_cluster_graph_cut_across_two_engines_equals_one_engine(graph, n_ranks):
    """cfg5 across devices, rehearsed on one (third bullet): the engines of a
    pgbp_group with the device listed n_ranks times hold the same level-3 network's cluster graph; NetworkCut runs
    calibrate! with every traversal cut by spanning-tree subtrees -- postorder in the owned subtrees, the subtree roots
    exchanged (pgbp_pack_beliefs / pgbp_unpack_beliefs), the top on every rank, preorder in the owned subtrees, and, on
    a loopy graph, the owned records exchanged for the next spanning tree.  Every belief equals the single-engine run's BIT
    FOR BIT (independent messages in another order, nothing else), so do the calibration flags and the iteration the
    automatic stop is reached at."""
    from pgbp_amd.sharding import NetworkCut
    net, (cn, ed, sn), st, fam, X, rates, mu, sched = _small_network(graph)
    loopy = len(ed) > len(cn) - 1
    lib = pgbp_amd.load()

    def prepare(cgb):
        cgb.lg_setup(fam, X)
        cgb.assignfactors_lg_(rates, mu)
        if loopy and graph == "joingraph":
            from pgbp_amd.regularization import regularizebeliefs_onschedule_
            regularizebeliefs_onschedule_(cgb)
        elif loopy:
            assert lib.pgbp_regularize_bycluster(cgb._eng) == 0
        cgb.pull()
        return cgb._packed[0].copy()

    one = pgbp_amd.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None)
    start = prepare(one)
    niter = 3 if loopy else 1
    one.set_schedule(sched)
    res = (L.Result * 1)()
    o = one._opts()
    assert lib.pgbp_calibrate(one._eng, niter, C.byref(o), res) == 0 and res[0].succ
    want = np.zeros((1, len(start)))
    assert lib.pgbp_get_beliefs(one._eng, L.f64p(want)) == 0
    flags_one = np.zeros(2 * one.nsepsets, np.int32)
    assert lib.pgbp_get_residuals(one._eng, None, L.i32p(flags_one), None, None) == 0

    grp = EngineGroup(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, n_ranks, [0] * n_ranks)
    ranks = [pgbp_amd.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None,
                                                     n_sites=1, engine=grp.engine(r)) for r in range(n_ranks)]
    for b in ranks:   # the same start on every rank (the regularisation walk is deterministic; upload the one state)
        b._packed[0, :] = start
        b._upload(snapshot_factors=True)
    cut = NetworkCut(ranks, sched)
    assert all(len(c["sub"]) >= 4 * n_ranks for c in cut.cuts), [len(c["sub"]) for c in cut.cuts]
    succ, iscal, _ = cut.calibrate(niter)
    assert succ
    got = cut.gather()
    scale = np.maximum(1.0, np.abs(want[0]))
    assert np.max(np.abs(got - want[0]) / scale) <= 1e-11, float(np.max(np.abs(got - want[0]) / scale))
    assert iscal == bool(flags_one.all())
    # what crossed between the ranks: per traversal the subtree roots (the boundary buffer), plus, on a loopy graph, the
    # records inside the owned subtrees
    n_trav = niter * len(sched)
    root_doubles = sum(int(ranks[0]._poff[i + 1] - ranks[0]._poff[i]) for t in range(len(sched)) for x in cut.roots[t] for i in x)
    owned_doubles = sum(int(ranks[0]._poff[i + 1] - ranks[0]._poff[i]) for t in range(len(sched)) for x in cut.owned[t] for i in x)
    assert cut.exchanged_doubles == niter * (root_doubles + (owned_doubles if len(sched) > 1 else 0))
    assert n_trav > 0 and root_doubles > 0
    if loopy:
        # the automatic stop (src/calibration.jl:51-57): the cut run reaches calibration at the iteration and schedule tree the
        # single engine reaches it at, each message's flag read from the rank that sent it last
        one._packed[0, :] = start
        one._upload(snapshot_factors=True)
        assert lib.pgbp_reset_flags(one._eng, 1) == 0
        oa = one._opts(auto=True)
        res2 = (L.Result * 1)()
        assert lib.pgbp_calibrate(one._eng, 60, C.byref(oa), res2) == 0 and res2[0].succ
        for b in ranks:
            b._packed[0, :] = start
            b._upload(snapshot_factors=True)
            assert lib.pgbp_reset_flags(b._eng, 1) == 0
        cut.last_writer = {}
        succ, iscal, reached = cut.calibrate(60, auto=True)
        assert succ and iscal == bool(res2[0].iscal)
        if res2[0].iscal:
            assert reached == (res2[0].iter_reached, res2[0].tree_reached), (reached, res2[0].iter_reached, res2[0].tree_reached)
    for b in ranks:
        b._eng = None
    grp.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ntips,p", [(30, 16), (25, 3)])
def test_exchange_buffer_round_trip_and_argument_checks(ntips, p):
    """pgbp_pack_beliefs / pgbp_unpack_beliefs (the exchange buffer of a cut cluster graph): the packed buffer holds the
    listed records back to back in the order of the list, exactly as pgbp_get_belief returns each (in the block-packed device
    layout too: 16 traits); unpacked into another engine they overwrite those beliefs and nothing else; a bad index is a size
    of -1 and PGBP_ERR_INVALID, an empty list is a no-op."""
    tr, prob, packs, lls = _problem(ntips, p, 2, 400 + ntips)
    lib = pgbp_amd.load()
    a = pgbp_amd.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packs, n_sites=2)
    assert pgbp_amd.calibrate_(a, prob.schedule, 1)[0]      # (the state now lives in the layout the traversal chose)
    rng = np.random.default_rng(5)
    nb = len(prob.dims)
    lst = np.ascontiguousarray(rng.permutation(nb)[: max(3, nb // 3)], np.int32)
    n = int(lib.pgbp_packed_beliefs_size(a._eng, len(lst), L.i32p(lst)))
    assert n == sum(int(a._poff[i + 1] - a._poff[i]) for i in lst)
    for site in (0, 1):
        buf = np.zeros(n)
        assert lib.pgbp_pack_beliefs(a._eng, site, len(lst), L.i32p(lst), L.f64p(buf)) == 0
        at = 0
        for i in lst:
            ln = int(a._poff[i + 1] - a._poff[i])
            rec = np.zeros(ln)
            assert lib.pgbp_get_belief(a._eng, site, int(i), L.f64p(rec)) == 0
            assert np.array_equal(buf[at:at + ln], rec), int(i)
            at += ln
        b = pgbp_amd.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packs, n_sites=2)
        before = np.zeros((2, len(packs[0])))
        assert lib.pgbp_get_beliefs(b._eng, L.f64p(before)) == 0
        assert lib.pgbp_unpack_beliefs(b._eng, site, len(lst), L.i32p(lst), L.f64p(buf)) == 0
        after = np.zeros_like(before)
        assert lib.pgbp_get_beliefs(b._eng, L.f64p(after)) == 0
        want = before.copy()
        at = 0
        for i in lst:
            ln = int(a._poff[i + 1] - a._poff[i])
            want[site, a._poff[i]:a._poff[i] + ln] = buf[at:at + ln]
            at += ln
        assert np.array_equal(after, want)
    bad = np.array([0, nb], np.int32)
    assert lib.pgbp_packed_beliefs_size(a._eng, 2, L.i32p(bad)) == -1
    assert lib.pgbp_pack_beliefs(a._eng, 0, 2, L.i32p(bad), L.f64p(np.zeros(4096))) == 1
    assert lib.pgbp_pack_beliefs(a._eng, 2, 1, L.i32p(lst), L.f64p(np.zeros(4096))) == 1        # site out of range
    assert lib.pgbp_pack_beliefs(a._eng, 0, 0, None, None) == 0


def _kldiv_longdouble(J0f, h0f, dJf, dhf):
    """residual_kldiv! (src/beliefs.jl:1060-1075) in 80-bit arithmetic: Cholesky factors and triangular solves written out"""
    LD = np.longdouble
    J0 = np.triu(J0f).astype(LD)
    J0 = J0 + np.triu(J0, 1).T
    J1f = (np.asarray(J0f, LD) - np.asarray(dJf, LD))
    J1 = np.triu(J1f) + np.triu(J1f, 1).T
    h0, h1 = np.asarray(h0f, LD), np.asarray(h0f, LD) - np.asarray(dhf, LD)
    dJ = np.asarray(dJf, LD)

    def chol(A):
        n = len(A)
        Lm = np.zeros((n, n), LD)
        for j in range(n):
            Lm[j, j] = np.sqrt(A[j, j] - np.dot(Lm[j, :j], Lm[j, :j]))
            for i in range(j + 1, n):
                Lm[i, j] = (A[i, j] - np.dot(Lm[i, :j], Lm[j, :j])) / Lm[j, j]
        return Lm

    def solve(Lm, B):
        n = len(Lm)
        Y = np.array(B, LD, copy=True)
        for i in range(n):
            Y[i] = (Y[i] - np.dot(Lm[i, :i], Y[:i])) / Lm[i, i]
        for i in range(n - 1, -1, -1):
            Y[i] = (Y[i] - np.dot(Lm[i + 1:, i], Y[i + 1:])) / Lm[i, i]
        return Y
    L0, L1 = chol(J0), chol(J1)
    mu0, mu1 = solve(L0, h0), solve(L1, h1)
    tr = np.trace(solve(L0, dJ))
    dd = mu1 - mu0
    ld0, ld1 = 2 * np.sum(np.log(np.diag(L0))), 2 * np.sum(np.log(np.diag(L1)))
    return float((-tr + dd @ (J1 @ dd) + ld0 - ld1) / 2)


@pytest.mark.gpu
@pytest.mark.parametrize("p", [3, 2])
def test_residual_kldiv_on_the_muller_clique_tree_with_sepsets_beyond_the_lds(p):
    """calibrate!(...; update_residualkldiv = true) (src/calibration.jl:128,154; residual_kldiv!, src/beliefs.jl:1060-1075, has
    no size bound) on the reference's documented clique tree of the Mueller et al. network at p = 3: sepsets of up to 126
    variables, beyond the 96 whose two systems fit a CU's LDS -- those messages run on the workspace instance of the kernel,
    the others of the same level on the LDS instance sized by that level's largest sepset that fits (p = 2: up to 84, the LDS
    instance alone).  After one calibrate!()
    from the factors every preorder message is the last one sent over its edge, so its sepset still holds it: its kldiv must
    equal the oracle's residual_kldiv! of (that sepset, that residual); where J0 - dJ is not positive definite the reference
    leaves kldiv alone (-1) and so does the device.  The standalone pgbp_residual_kldiv of the largest one gives the same."""
    import pgbp_amd as P
    from oracle import beliefs as OB
    path = os.path.join(ROOT, "tests", "golden", "muller_2022.phy")
    net, names = P.read_newick(open(path).read())
    cn, ed, sn = P.cliquetree(net.node2family)
    st = P.allocate_scopes(cn, ed, sn, net, p)
    assert (int(st.dims[len(cn):].max()) > 96) == (p == 3)
    rng = np.random.default_rng(3)
    rates = np.stack([np.eye(p) + 0.3])
    X = P.simulate_bm_network(net, rates, np.zeros(p), rng)
    pe = [list(zip(net.length[i], net.gamma[i], net.color[i])) for i in range(net.nnodes)]
    fam = P.lg_families(st.clusters, st.node2cluster, net.node2family, st.node2fixed, pe, list(range(net.nnodes)), p, n_rates=1)
    cgb = P.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None)
    cgb.lg_setup(fam, X)
    cgb.assignfactors_lg_(rates, np.zeros(p))
    root = P.default_rootcluster(cn, net.is_leaf)
    spt = P.spanningtree_clusterlist(len(cn), ed, root)
    sched = [(np.asarray(spt[2]), np.asarray(spt[3]))]
    # calibrate once; then disturb a few large cliques and calibrate once more with the KL residuals on: in that second
    # iteration every sepset starts from a calibrated marginal (positive definite) and moves by a visible amount -- after a
    # first calibrate!() from the factors most "beliefs before" are 0 or singular and residual_kldiv! leaves nearly every
    # message alone (563 of 663 here)
    assert P.calibrate_(cgb, sched, 1)[0]
    order = np.argsort(-st.dims[:len(cn)])
    for i in order[:12]:
        m = int(st.dims[i])
        rec = np.zeros(m * m + m + 1)
        assert cgb._lib.pgbp_get_belief(cgb._eng, 0, int(i), L.f64p(rec)) == 0
        Jm = rec[:m * m].reshape(m, m)
        Jm[np.diag_indices(m)] *= 1.05
        rec[m * m: m * m + m] *= 1.02
        assert cgb._lib.pgbp_set_belief(cgb._eng, 0, int(i), L.f64p(rec)) == 0
    assert P.calibrate_(cgb, sched, 1, update_residualkldiv=True)[0]
    cgb.pull()
    nc = len(cn)
    checked = big = skipped = illcond = 0
    biggest = None
    for a, c in zip(*sched[0]):                       # the preorder message a -> c
        d = cgb._msg_id(int(c), int(a))
        k = d // 2
        s_ = int(st.dims[nc + k])
        if s_ == 0:
            continue
        J, h, _ = cgb._views(0, nc + k)
        sep = OB.CanonicalBelief.__new__(OB.CanonicalBelief)
        sep.J, sep.h, sep.mu = np.array(J), np.array(h), np.zeros(s_)
        rec = cgb._residual_record(d)
        res = OB.MessageResidual(s_)
        res.dJ, res.dh = rec[: s_ * s_].reshape(s_, s_, order="F").copy(), rec[s_ * s_: s_ * s_ + s_].copy()
        res.kldiv = -1.0
        ok = OB.residual_kldiv(res, sep)
        got, flag = float(cgb._kldiv()[d]), bool(cgb._klflags()[d])
        if res.kldiv == -1.0 and got == -1.0:      # not positive definite: left alone by both
            skipped += 1
            continue
        if res.kldiv == -1.0 or got == -1.0 or abs(got - res.kldiv) > 1e-8 * max(1.0, abs(res.kldiv)):
            # A first calibrate!() from the factors: KL values of 1e2 between beliefs whose precisions differ by orders of
            # magnitude -- tr(J0^-1 dJ), two log-determinants and a quadratic form of ill-conditioned, sometimes nearly singular
            # matrices.  Where the device (Gauss-Jordan without pivoting) and the oracle (LAPACK, float64) part ways -- a
            # different value, or one of them meeting a non-positive pivot where the other does not -- an 80-bit restatement
            # says how far each is from the truth.  Not positive definite in 80 bits, or so close to it that float64 LAPACK is
            # itself off by more than 1e-9 (or gave up): either answer is within rounding.  Otherwise the device must be within
            # 1e-8 of the 80-bit value, or no further from it than a hundred times the float64 oracle's own distance (observed:
            # 18 times, on a KL of 105.9 -- five orders of magnitude above the 1e-5 at which the value decides anything:
            # iscalibrated_kl!, src/beliefs.jl:1014-1016).
            ref = _kldiv_longdouble(sep.J, sep.h, res.dJ, res.dh)
            scale = max(1.0, abs(ref)) if np.isfinite(ref) else 1.0
            if (not np.isfinite(ref)) or res.kldiv == -1.0 or (got == -1.0 and abs(res.kldiv - ref) > 1e-9 * scale):
                skipped += 1
                continue
            assert got != -1.0, (d, s_, got, res.kldiv, ref)
            assert abs(got - ref) <= max(1e-8 * scale, 100.0 * abs(res.kldiv - ref)), (d, s_, got, res.kldiv, ref)
            illcond += 1
            checked += 1
            if s_ > 96:
                big += 1
            continue
        assert flag == bool(ok)
        checked += 1
        if s_ > 96:
            big += 1
            if biggest is None or s_ > biggest[0]:
                biggest = (s_, int(c), nc + k, int(a), got)
    assert checked >= 400 and (big >= 1 or p == 2) and illcond <= checked // 4, (checked, big, skipped, illcond)
    if biggest is None:
        return
    # the standalone call on the largest sepset (workspace instance, one entry)
    s_, to, sepset, frm, want = biggest
    cgb.residual_kldiv_(to, sepset, frm)
    assert float(cgb._kldiv()[cgb._msg_id(to, frm)]) == want


@pytest.mark.gpu
@pytest.mark.parametrize("graph", ["bethe", "cliquetree"])
def test_cluster_graph_cut_between_two_processes(graph):
    """The cut of a cluster graph across ranks BETWEEN PROCESSES (tests/run_cut_rehearsal.py): two
    ranks launched as the driver launches the bench, each with its own engine on this box's one GPU, the exchange buffers
    carried by an all-gather of the launcher's group (gloo here; pgbp_comm_exchange_beliefs = the same all-gather over RCCL,
    device to device, where every rank has a GPU of its own), the calibration flags and successes ANDed over the ranks.
    Every belief equals the single-engine run's (1e-11 relative), so does iscal, and on the loopy graph the iteration and
    schedule tree at which calibrate!(...; auto = true) stops."""
    import json
    import subprocess
    import sys
    port = 29700 + (os.getpid() % 250)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "run_cut_rehearsal.py"), graph]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["succ"] and r["n_ranks"] == 2 and all(n >= 8 for n in r["subtrees"])
    assert r["max_rel_belief_diff"] <= 1e-11, r
    assert r["iscal"] == r["iscal_one_engine"] and r["exchanged_doubles"] > 0
    if graph == "bethe":
        assert r["auto_reached"] == r["auto_reached_one_engine"], r


@pytest.mark.gpu
def test_comm_exchange_beliefs_single_rank_round_trip():
    """pgbp_comm_exchange_beliefs on ONE rank (two RCCL ranks cannot share this box's GPU): the listed records are gathered
    on the device into the send slot, travel through ncclAllGather and -- include_self -- are scattered back from the receive
    buffer into the engine.  The call must leave every belief bit for bit as it was (a wrong offset table or slot size would
    scramble the listed records or their neighbours), and a bad index is refused."""
    from pgbp_amd import synth as S
    from pgbp_amd.sharding import Comm
    rng = np.random.default_rng(5)
    tr = S.random_tree(40, rng)
    p = 3
    R = S.random_rate_matrix(p, rng)
    X = S.simulate_bm(tr, R, np.zeros(p), rng)
    prob = S.cliquetree_of_tree(tr, p)
    packed = S.bm_factors_cliquetree(tr, prob, R, np.zeros(p), X)
    cgb = pgbp_amd.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packed)
    assert pgbp_amd.calibrate_(cgb, prob.schedule, 1)[0]
    cgb.pull()
    before = cgb._packed[0].copy()
    try:
        comm = Comm(1, 0, 0)
    except L.PgbpError as ex:
        pytest.skip(f"RCCL not loadable here: {ex}")
    lst = np.array([3, 0, 17, len(prob.dims) - 1, 5], np.int32)
    comm.exchange_beliefs(cgb._eng, [lst], include_self=True)
    cgb.pull()
    assert np.array_equal(cgb._packed[0], before)
    with pytest.raises(L.PgbpError):
        comm.exchange_beliefs(cgb._eng, [np.array([len(prob.dims)], np.int32)], include_self=True)
    comm.close()
