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
