"""CPU tests (no GPU): the C-ABI library loads and exports every symbol include/pgbp.h
declares; the host-only planner (layout, message table, level schedule) is correct;
the product has no CPU fallback; the vectorised synthetic factor fill agrees with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import pgbp_amd
from pgbp_amd import _lib as L
from pgbp_amd import synth as S

from helpers import oracle_cgb_from_problem, oracle_schedule, pack_oracle
from oracle import beliefs as OB
from oracle import calibration as OC
from oracle import clustergraph as OCG
from oracle import models as OM
from oracle import network as ON

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "pgbp.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pgbp_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = C.CDLL(L.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"libpgbp.so does not export {name}"
    assert declared == set(L.SYMBOLS.keys())
    pgbp_amd.load()


def _plan(prob, n_sites=1):
    lib = pgbp_amd.load()
    desc, keep = L.make_desc(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, n_sites, 0)
    pl = C.c_void_p()
    code = lib.pgbp_plan_create(C.byref(desc), C.byref(pl))
    return lib, pl, code, keep


def _set_sched(lib, pl, schedule):
    off = np.zeros(len(schedule) + 1, np.int32)
    for i, (pa, ch) in enumerate(schedule):
        off[i + 1] = off[i] + len(pa)
    pa = np.ascontiguousarray(np.concatenate([s[0] for s in schedule]).astype(np.int32))
    ch = np.ascontiguousarray(np.concatenate([s[1] for s in schedule]).astype(np.int32))
    return lib.pgbp_plan_set_schedule(pl, len(schedule), L.i32p(off), L.i32p(pa), L.i32p(ch))


def _traversal(lib, pl, tree, d):
    nl, nt, ne = C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.pgbp_plan_traversal_sizes(pl, tree, d, C.byref(nl), C.byref(nt), C.byref(ne)) == 0
    lo = np.zeros(nl.value + 1, np.int32)
    to = np.zeros(nt.value + 1, np.int32)
    em, ee, er = (np.zeros(max(1, ne.value), np.int32) for _ in range(3))
    assert lib.pgbp_plan_traversal(pl, tree, d, L.i32p(lo), L.i32p(to), L.i32p(em), L.i32p(ee), L.i32p(er)) == 0
    return lo, to, em[:ne.value], ee[:ne.value], er[:ne.value]


@pytest.mark.parametrize("ntips,p,kind", [(2, 1, "random"), (3, 2, "random"), (40, 3, "random"), (500, 2, "random"),
                                          (30, 2, "caterpillar"), (60, 16, "random"), (70, 16, "poly6"),
                                          (50, 8, "random"), (45, 4, "random")])
def test_level_schedule_invariants(ntips, p, kind):
    rng = np.random.default_rng(ntips)
    if kind == "random":
        tr = S.random_tree(ntips, rng)
    elif kind == "caterpillar":
        tr = S.caterpillar_tree(ntips, rng)
    else:
        tr = S.random_multifurcating_tree(ntips, int(kind[4:]), rng)
    prob = S.cliquetree_of_tree(tr, p)
    lib, pl, code, keep = _plan(prob)
    assert code == 0, lib.pgbp_plan_last_error(pl)
    assert lib.pgbp_plan_packed_size(pl) == prob.packed_off[-1]
    assert lib.pgbp_plan_n_messages(pl) == 2 * (len(prob.dims) - prob.nclusters)
    assert _set_sched(lib, pl, prob.schedule) == 0, lib.pgbp_plan_last_error(pl)
    pa, ch = prob.schedule[0]
    n = len(pa)
    sepcl = prob.sepset_clusters
    for d in (0, 1):
        lo, to, em, ee, er = _traversal(lib, pl, 0, d)
        assert sorted(ee.tolist()) == list(range(n))          # every edge exactly once
        level_of_edge = np.zeros(n, int)
        for Lv in range(len(lo) - 1):
            targets, senders = set(), set()
            for t in range(lo[Lv], lo[Lv + 1]):
                ents = list(range(to[t], to[t + 1]))
                assert ents
                for e in ents:
                    i = ee[e]
                    level_of_edge[i] = Lv
                    snd, rcv = (ch[i], pa[i]) if d == 0 else (pa[i], ch[i])
                    k = em[e] // 2
                    assert {int(sepcl[k][0]), int(sepcl[k][1])} == {int(pa[i]), int(ch[i])}
                    assert int(sepcl[k][em[e] % 2]) == int(rcv)   # dir 0: received by a
                    senders.add(int(snd))
                    targets.add(int(rcv))
                edges = [ee[e] for e in ents]
                if d == 0:   # one task per target, entries in the reference's order (decreasing edge index)
                    assert len({int(pa[i]) for i in edges}) == 1, (Lv, edges)
                    assert edges == sorted(edges, reverse=True)
                else:        # one task per sender, increasing edge index
                    assert len({int(pa[i]) for i in edges}) == 1
                    assert edges == sorted(edges)
                    assert er[to[t]] == 0
            if d == 0:
                tl = [int(pa[ee[to[t]]]) for t in range(lo[Lv], lo[Lv + 1])]
                assert len(tl) == len(set(tl))                  # distinct targets across tasks
            nf = np.zeros(len(lo) - 1, np.int32)
            assert lib.pgbp_plan_level_nfast(pl, 0, d, L.i32p(nf)) == 0
            assert 0 <= nf[Lv] <= lo[Lv + 1] - lo[Lv]
            if p in (4, 8, 16) and kind == "random" and ntips > 3:
                assert nf[Lv] == lo[Lv + 1] - lo[Lv]            # a 4/8/16-trait bifurcating tree is all fast-class
            assert not (targets & senders)                      # no cluster both read and written in a level
        # dependencies: postorder: a child's incoming messages precede its outgoing one
        child_edge = {int(c): i for i, c in enumerate(ch)}
        for i in range(n):
            if int(pa[i]) in child_edge:
                j = child_edge[int(pa[i])]   # edge above pa[i]
                if d == 0:
                    assert level_of_edge[i] < level_of_edge[j]
                else:
                    assert level_of_edge[j] < level_of_edge[i]
        if kind == "caterpillar" and d == 1:
            assert len(lo) - 1 >= ntips - 2
    lib.pgbp_plan_destroy(pl)


def test_preorder_marginal_reuse_flag():
    rng = np.random.default_rng(1)
    tr = S.random_tree(50, rng)
    prob = S.cliquetree_of_tree(tr, 2)
    lib, pl, code, keep = _plan(prob)
    assert _set_sched(lib, pl, prob.schedule) == 0
    lo, to, em, ee, er = _traversal(lib, pl, 0, 1)
    assert er.sum() > 0          # internal cliques send the same marginal to both children
    lo, to, em, ee, er = _traversal(lib, pl, 0, 0)
    assert er.sum() == 0
    lib.pgbp_plan_destroy(pl)


def test_plan_rejects_bad_input():
    rng = np.random.default_rng(5)
    tr = S.random_tree(6, rng)
    prob = S.cliquetree_of_tree(tr, 2)
    lib, pl, code, keep = _plan(prob)
    pa, ch = prob.schedule[0]
    # child seen twice / parent unseen -> not a tree
    bad = (pa.copy(), ch.copy())
    bad[1][-1] = bad[1][0]
    assert _set_sched(lib, pl, [bad]) == L.ERR_NOT_TREE
    bad = (pa[::-1].copy(), ch[::-1].copy())
    assert _set_sched(lib, pl, [bad]) in (L.ERR_NOT_TREE, L.ERR_INVALID)
    assert b"schedule tree 0" in lib.pgbp_plan_last_error(pl)
    lib.pgbp_plan_destroy(pl)
    # scope indices not increasing (labels out of order: src/beliefs.jl:398)
    p2 = S.cliquetree_of_tree(tr, 2)
    k = np.nonzero(np.diff(p2.scope_off) == 2)[0][0]
    p2.scope_idx[p2.scope_off[k]:p2.scope_off[k] + 2] = p2.scope_idx[p2.scope_off[k]:p2.scope_off[k] + 2][::-1]
    lib, pl, code, keep = _plan(p2)
    assert code == L.ERR_INVALID and b"strictly increasing" in lib.pgbp_plan_last_error(pl)
    lib.pgbp_plan_destroy(pl)
    # an unknown token of PGBP_TUNING is an error of the create call, not a silently ignored typo
    os.environ["PGBP_TUNING"] = "no_tail,chunk_binz=3"
    try:
        lib, pl, code, keep = _plan(S.cliquetree_of_tree(tr, 2))
        assert code == L.ERR_INVALID and b"chunk_binz=3" in lib.pgbp_plan_last_error(pl)
        lib.pgbp_plan_destroy(pl)
    finally:
        del os.environ["PGBP_TUNING"]
    # dimension above PGBP_MAX_DIM is refused, not silently mishandled
    p3 = S.cliquetree_of_tree(tr, 200)         # internal cliques of dimension 400 > 384
    lib, pl, code, keep = _plan(p3)
    assert code == L.ERR_TOO_LARGE and b"PGBP_MAX_DIM=384" in lib.pgbp_plan_last_error(pl)
    lib.pgbp_plan_destroy(pl)
    lib, pl, code, keep = _plan(S.cliquetree_of_tree(tr, 40))   # dimension 80: the large-belief kernel's range
    assert code == 0
    lib.pgbp_plan_destroy(pl)
    lib, pl, code, keep = _plan(S.cliquetree_of_tree(tr, 100))  # dimension 200: its workspace variant's
    assert code == 0
    # more sites than grid.y holds: refused for the wavefront-per-message kernels, fine for univariate batches
    lib, pl, code, keep = _plan(S.cliquetree_of_tree(tr, 2), n_sites=70000)
    assert code == L.ERR_TOO_LARGE and b"65535" in lib.pgbp_plan_last_error(pl)
    lib, pl, code, keep = _plan(S.cliquetree_of_tree(tr, 1), n_sites=70000)
    assert code == 0
    lib.pgbp_plan_destroy(pl)


def test_no_cpu_fallback():
    """Without a GPU the product must fail loudly (skipped on a GPU box)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    tr = S.random_tree(4, np.random.default_rng(0))
    prob = S.cliquetree_of_tree(tr, 1)
    with pytest.raises(pgbp_amd.PgbpError) as ei:
        pgbp_amd.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx,
                                                np.zeros(int(prob.packed_off[-1])))
    assert ei.value.code == L.ERR_NO_DEVICE


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "phylogaussianbeliefprop.jl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f
                assert "oracle/" not in txt, f


@pytest.mark.parametrize("ntips,p", [(2, 1), (5, 1), (12, 2), (40, 3)])
def test_synth_factors_match_oracle_assignfactors(ntips, p):
    """pgbp_amd.synth's vectorised BM factor fill == the oracle's assignfactors! restatement, and the
    oracle's calibrated log-likelihood == dense MVN == the pruning check."""
    from oracle import densemvn as OD
    rng = np.random.default_rng(100 + ntips)
    tr = S.random_tree(ntips, rng)
    R = S.random_rate_matrix(p, rng)
    mu = rng.standard_normal(p)
    X = S.simulate_bm(tr, R, mu, rng)
    prob = S.cliquetree_of_tree(tr, p)
    packed = S.bm_factors_cliquetree(tr, prob, R, mu, X)
    # oracle route: Newick -> network -> our explicit clique tree -> allocate + assignfactors
    names = [f"n{i}" for i in range(tr.nnodes)]
    net = ON.read_newick(tr.newick(names))
    net.set_preorder(names)
    taxa = [names[i] for i in range(tr.nnodes) if tr.is_leaf[i]]
    tbl = [[float(X[i, t]) for i in range(tr.nnodes) if tr.is_leaf[i]] for t in range(p)]
    clusters = [(str(i), [int(a), int(b)]) for i, (a, b) in enumerate(prob.cluster_nodes)]
    edges = [(int(a), int(b), [int(prob.sepset_nodes[k])]) for k, (a, b) in enumerate(prob.sepset_clusters)]
    cg = OB.ClusterGraph(clusters, edges, "cliquetree")
    model = OM.MvFullBrownianMotion(R, mu)
    b, (n2c, n2f, n2fix, n2d, c2n) = OB.allocatebeliefs(tbl, taxa, net, cg, model)
    assert [x.dimension for x in b] == prob.dims.tolist()
    OB.assignfactors(b, model, tbl, taxa, net, n2c, n2f, n2fix)
    cgb = OB.ClusterGraphBelief(b, n2c, n2f, n2fix, c2n)
    ref = pack_oracle(cgb, prob)
    assert np.allclose(packed, ref, rtol=1e-12, atol=1e-12)
    # scope maps agree with the reference's scopeindex
    for k, (a, c) in enumerate(prob.sepset_clusters):
        sb = b[prob.nclusters + k]
        for side, cl in enumerate((a, c)):
            ind = OB.scopeindex(sb, b[cl])
            o0, o1 = prob.scope_off[2 * k + side], prob.scope_off[2 * k + side + 1]
            assert ind.tolist() == prob.scope_idx[o0:o1].tolist()
    spt = ([str(x) for x in prob.schedule[0][0]], [str(x) for x in prob.schedule[0][1]],
           prob.schedule[0][0].tolist(), prob.schedule[0][1].tolist())
    assert OC.calibrate(cgb, [spt])[0]
    ll = cgb.integratebelief(prob.root_cluster)[1]
    dense = OD.loglik(net, model, tbl, taxa)
    assert abs(ll - dense) <= 1e-9 * max(1, abs(dense))
    assert abs(S.bm_loglik_pruning(tr, R, mu, X) - dense) <= 1e-9 * max(1, abs(dense))


def test_mixed_level_goes_to_the_generic_kernel_whole():
    """A level that needs the generic kernel anyway (a task with more than 4 messages: a 7-way polytomy) and has
    fewer than 2048 fast-class tasks is planned as ONE generic launch (level_nfast = 0); levels made of fast-class
    tasks only keep them all on the register-resident kernel; even trait counts <= 16 all have an instance."""
    rng = np.random.default_rng(11)
    tr = S.random_multifurcating_tree(300, 7, rng)
    for p in (16, 6, 2):
        prob = S.cliquetree_of_tree(tr, p)
        lib, pl, code, keep = _plan(prob)
        assert code == 0 and _set_sched(lib, pl, prob.schedule) == 0
        saw_mixed = saw_fast = False
        for d in (0, 1):
            nl, nt, ne = C.c_int32(), C.c_int32(), C.c_int32()
            assert lib.pgbp_plan_traversal_sizes(pl, 0, d, C.byref(nl), C.byref(nt), C.byref(ne)) == 0
            lo = np.zeros(nl.value + 1, np.int32); to = np.zeros(nt.value + 1, np.int32)
            em = np.zeros(ne.value, np.int32); ee = em.copy(); er = em.copy()
            assert lib.pgbp_plan_traversal(pl, 0, d, L.i32p(lo), L.i32p(to), L.i32p(em), L.i32p(ee), L.i32p(er)) == 0
            nf = np.zeros(nl.value, np.int32)
            assert lib.pgbp_plan_level_nfast(pl, 0, d, L.i32p(nf)) == 0
            for Lv in range(nl.value):
                sizes = np.diff(to[lo[Lv]: lo[Lv + 1] + 1])
                if (sizes > 4).any():
                    assert nf[Lv] == 0
                    saw_mixed = True
                elif nf[Lv] == len(sizes):   # (a level may still hold a task the fast class does not cover, e.g. several
                    saw_fast = True          # constant messages into one receiver: then it is generic as a whole, too)
                else:
                    assert nf[Lv] == 0
        assert saw_mixed and saw_fast
        lib.pgbp_plan_destroy(pl)


def test_header_is_plain_c_and_example_links():
    """include/pgbp.h is a C header (C99, -pedantic): a foreign-function binding needs nothing else; the C example
    compiles and links against the built library (it runs in tests/test_gpu_parity.py)."""
    import shutil
    import subprocess
    import tempfile
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    with tempfile.TemporaryDirectory() as td:
        subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c",
                               os.path.join(ROOT, "examples", "c_abi_example.c"), "-o", os.path.join(td, "ex.o")])
        libdir = os.path.dirname(pgbp_amd.LIB_PATH)
        subprocess.check_call(["gcc", os.path.join(td, "ex.o"), "-L", libdir, "-lpgbp", "-lm", "-o", os.path.join(td, "ex")])


def test_schedule_builders_match_the_oracle():
    """The product's host-side spanningtree(s)_clusterlist / default_rootcluster (src/clustergraph.jl:885-937,
    1022-1029) against the oracle's restatement (itself behind the golden loopy runs: "iteration 5, schedule tree 1")
    on clique trees and Bethe graphs of random networks and on the reference's level-1 network."""
    from oracle import clustergraph as OCG
    from oracle import network as ON
    from helpers import goldens
    nets = [ON.read_newick(goldens()["calibration_bethe_level1"]["net"])]
    rng = np.random.default_rng(8)
    nets += [ON.random_network(40, 8, rng), ON.random_network(120, 40, rng)]
    for net in nets:
        is_leaf = [n.leaf for n in net.vec_node]
        for cg in (OCG.bethe(net), OCG.cliquetree(net)):
            nodes = [c[1] for c in cg.clusters]
            edges = [(a, b) for (a, b, _) in cg.edges]
            root = pgbp_amd.default_rootcluster(nodes, is_leaf)
            assert root == OCG.default_rootcluster(cg, net)
            got = pgbp_amd.spanningtree_clusterlist(len(nodes), edges, root, cg.labels)
            ref = OCG.spanningtree_clusterlist(cg, root)
            assert tuple(map(list, got)) == tuple(map(list, ref))
            gots = pgbp_amd.spanningtrees_clusterlist(len(nodes), edges, nodes, is_leaf, cg.labels)
            refs = OCG.spanningtrees_clusterlist(cg, net)
            assert len(gots) == len(refs)
            for g, r in zip(gots, refs):
                assert tuple(map(list, g)) == tuple(map(list, r))
            covered = set()
            for g in gots:   # every edge is in some tree; each tree spans all clusters
                assert len(g[2]) == len(nodes) - 1
                covered |= {frozenset(e) for e in zip(g[2], g[3])}
            assert covered == {frozenset(e) for e in edges}


# ----------------------------------------------------------------------------- cluster-graph construction on plain arrays

def _arrays_to_oracle_network(net):
    from oracle import network as ON
    nodes = [ON.Node(name=f"n{i + 1}", leaf=bool(net.is_leaf[i]), hybrid=len(net.node2family[i]) > 2)
             for i in range(net.nnodes)]
    edges = []
    for i, nf in enumerate(net.node2family):
        for k, pl in enumerate(nf[1:]):
            e = ON.Edge(number=len(edges) + 1, parent=nodes[pl - 1], child=nodes[i], length=net.length[i][k],
                        gamma=net.gamma[i][k], hybrid=len(nf) > 2)
            edges.append(e)
            nodes[pl - 1].edges.append(e)
            nodes[i].edges.append(e)
    o = ON.Network(nodes[0], nodes, edges)
    o.set_preorder([n.name for n in nodes])
    return o


def test_joingraph_minfill_equal_the_oracle_on_random_networks():
    """The product's heap-based min-fill order and join-graph structuring (clustergraph.py) against the oracle's
    line-by-line restatement (rescan of every vertex per elimination, src/clustergraph.jl:87-121, 605-697)."""
    import pgbp_amd as P
    from oracle import clustergraph as OCG
    from oracle import network as ON
    for seed in range(12):
        rng = np.random.default_rng(seed)
        net = (ON.random_network(int(rng.integers(5, 50)), int(rng.integers(0, 12)), rng) if seed % 2 else
               ON.random_level3_network(int(rng.integers(6, 40)), int(rng.integers(1, 5)), rng))
        fam = OCG.nodefamilies(net)
        adj_o, adj_p = OCG.moralize(net), P.moralize(fam)
        assert OCG.triangulate_minfill(adj_o) == P.triangulate_minfill(adj_p) and adj_o == adj_p
        for k in (3, 4, 6):
            cg = OCG.joingraph(net, k)
            cn, ed, sn = P.joingraph(fam, k)
            assert [n for _, n in cg.clusters] == cn
            assert sorted((a, b, tuple(s)) for a, b, s in cg.edges) == sorted((a, b, tuple(s)) for (a, b), s in zip(ed, sn))
            assert OCG.isfamilypreserving(cg, net) and OCG.check_runningintersection(cg, net)
            assert max(len(n) for n in cn) <= k
    with pytest.raises(ValueError, match="smaller than the size of largest node family"):
        P.joingraph(fam, 2)


def test_arrays_pipeline_equals_the_oracle():
    """networks.py / clustergraph.py on plain arrays == the oracle's objects: node families, Bethe and join-graph
    cluster graphs, and what allocatebeliefs establishes (dimensions, node2cluster, scopeindex of both sepset ends)."""
    import pgbp_amd as P
    from helpers import oracle_setup
    from oracle import beliefs as OB
    from oracle import clustergraph as OCG
    from oracle import models as OM
    p = 2
    for seed in range(5):
        rng = np.random.default_rng(seed)
        net = P.random_level3_network(int(rng.integers(6, 40)), int(rng.integers(1, 5)), rng, n_colors=3)
        onet = _arrays_to_oracle_network(net)
        assert OCG.nodefamilies(onet) == net.node2family
        for kind in ("bethe", "join"):
            cn, ed, sn = P.bethe(net.node2family) if kind == "bethe" else P.joingraph(net.node2family, 3)
            ocg = OCG.bethe(onet) if kind == "bethe" else OCG.joingraph(onet, 3)
            assert [n for _, n in ocg.clusters] == cn
            norm = lambda a, b, s: (min(a, b), max(a, b), tuple(s))
            assert sorted(norm(a, b, s) for a, b, s in ocg.edges) == sorted(norm(a, b, s) for (a, b), s in zip(ed, sn))
            ocg2 = OB.ClusterGraph([(str(i), n) for i, n in enumerate(cn)], [(a, b, s) for (a, b), s in zip(ed, sn)], kind)
            st = P.allocate_scopes(cn, ed, sn, net, p)
            model = OM.MvFullBrownianMotion(np.eye(p), np.zeros(p))
            taxa = onet.tip_names
            tbl = [list(rng.normal(size=len(taxa))) for _ in range(p)]
            ocgb = oracle_setup(onet, ocg2, model, tbl, taxa)
            assert [b.dimension for b in ocgb.belief] == st.dims.tolist()
            assert ocgb.node2cluster == st.node2cluster
            assert list(ocgb.node2fixed) == st.node2fixed.tolist()
            idx, off = [], [0]
            for j in range(ocgb.nclusters, len(ocgb.belief)):
                for c in ocg2.edges[j - ocgb.nclusters][:2]:
                    idx += OB.scopeindex(ocgb.belief[j], ocgb.belief[c]).tolist()
                    off.append(len(idx))
            assert idx == st.scope_idx.tolist() and off == st.scope_off.tolist()


def test_nodesubtree_clusterlist_equals_the_oracle():
    import pgbp_amd as P
    from oracle import clustergraph as OCG
    from oracle import network as ON
    rng = np.random.default_rng(3)
    net = ON.random_level3_network(30, 4, rng)
    cg = OCG.bethe(net)
    cn = [n for _, n in cg.clusters]
    ed = [(a, b) for a, b, _ in cg.edges]
    sn = [s for _, _, s in cg.edges]
    for v in range(1, len(net.vec_node) + 1):
        o = OCG.nodesubtree_clusterlist(cg, v)
        q = P.nodesubtree_clusterlist(cn, ed, sn, v)
        assert (o[2], o[3]) == (q[2], q[3])


# ----------------------------------------------------------------------------- fused chains (generic-kernel schedules)

def _check_traversal(lo, to, em, ee, pa, ch, sepcl, d):
    """Invariants of one traversal (dir d) that make the level-synchronous, chain-fused execution equal to the
    reference's sequential loop: every edge once; the messages of a task are executed in order by one wave, tasks of a
    level run concurrently, levels in sequence.  (1) a message is sent only after every message into its sender (of this
    traversal) has been delivered: in an earlier level or earlier in the same task; (2) two tasks of one level touch
    disjoint beliefs, except that they may READ a common cluster nobody writes; (3) the messages a task delivers into one
    receiver are applied in the reference's order (postorder: decreasing edge index; siblings of different heights
    arrive in different levels, as without fusion: sums then differ from the sequential loop in the last bits only)."""
    n = len(pa)
    assert sorted(ee.tolist()) == list(range(n))
    when = {}
    for Lv in range(len(lo) - 1):
        written_by, read_by = {}, {}
        for t in range(lo[Lv], lo[Lv + 1]):
            assert to[t + 1] > to[t]
            for pos, e in enumerate(range(to[t], to[t + 1])):
                i = int(ee[e])
                snd, rcv = (int(ch[i]), int(pa[i])) if d == 0 else (int(pa[i]), int(ch[i]))
                k = int(em[e]) // 2
                assert {int(sepcl[k][0]), int(sepcl[k][1])} == {snd, rcv} and int(sepcl[k][em[e] % 2]) == rcv
                when[i] = (Lv, t, pos)
                written_by.setdefault(rcv, set()).add(t)
                read_by.setdefault(snd, set()).add(t)
        for c, ts in written_by.items():
            assert len(ts) == 1, ("two tasks of a level write cluster", c)
            assert read_by.get(c, ts) == ts, ("a task reads a cluster another task of the level writes", c)
    into = {}
    for i in range(n):
        rcv = int(pa[i]) if d == 0 else int(ch[i])
        into.setdefault(rcv, []).append(i)
    for i in range(n):
        snd = int(ch[i]) if d == 0 else int(pa[i])
        for j in into.get(snd, []):
            (Lj, tj, pj), (Li, ti, pi) = when[j], when[i]
            assert Lj < Li or (Lj == Li and tj == ti and pj < pi), (d, j, i)
    if d == 0:
        for rcv, edges in into.items():
            for Lv in {when[i][0] for i in edges}:
                same = [i for i in edges if when[i][0] == Lv]
                assert sorted(same, key=lambda i: when[i]) == sorted(same, reverse=True), rcv


def test_fused_chain_schedule_invariants_subprocess():
    """Chain fusion is opt-in through PGBP_TUNING=chain_fusion (read when a plan is built): the checks run in a child process."""
    import subprocess
    import sys
    env = dict(os.environ, PGBP_TUNING="chain_fusion")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_plan_cpu as T\n"
            "for g, s in [('bethe_tree', 1), ('bethe_net', 2), ('join_net', 3), ('bethe_net', 4), ('nodesubtrees', 5), ('path', 6)]:\n"
            "    T._fused_chain_schedule_invariants(g, s, True)\n"
            "print('fused ok')" % (ROOT, os.path.join(ROOT, "tests")))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "fused ok" in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])


@pytest.mark.parametrize("graph,seed", [("bethe_tree", 1), ("bethe_net", 2), ("join_net", 3), ("nodesubtrees", 5), ("path", 6)])
def test_generic_schedule_invariants(graph, seed):
    """The same invariants on the default (unfused) level schedules of generic-kernel graphs."""
    _fused_chain_schedule_invariants(graph, seed, False)


def _fused_chain_schedule_invariants(graph, seed, fused):
    """Chain fusion (pgbp_plan.cpp build_traversals): unary clusters of the schedule tree are passed through inside one
    task; Bethe graphs lose about half of their levels."""
    import pgbp_amd as P
    rng = np.random.default_rng(seed)
    if graph == "bethe_tree":
        tr = S.random_tree(60, rng)
        prob = S.bethe_of_tree(tr, 1)                    # univariate, one site: generic kernel
        dims, sepcl, soff, sidx, sched = prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, prob.schedule
    else:
        net = P.random_level3_network(80, 6, rng)
        if graph == "join_net":
            cn, ed, sn = P.joingraph(net.node2family, 3)
        else:
            cn, ed, sn = P.bethe(net.node2family)
        st = P.allocate_scopes(cn, ed, sn, net, 1)       # univariate, one site: every task on the generic kernel
        dims, sepcl, soff, sidx = st.dims, st.sepset_clusters, st.scope_off, st.scope_idx
        if graph == "nodesubtrees":
            sched = [x for x in (P.nodesubtree_clusterlist(cn, ed, sn, v) for v in range(1, net.nnodes + 1)) if x[0]]
        elif graph == "path":
            full = P.spanningtrees_clusterlist(len(cn), ed, cn, net.is_leaf)[0]
            # one root-to-leaf path of the spanning tree: every cluster unary -> one level per direction
            pa_l, ch_l = list(full[2]), list(full[3])
            node = ch_l[-1]
            path = []
            parent = dict(zip(ch_l, pa_l))
            while node in parent:
                path.append((parent[node], node))
                node = parent[node]
            path.reverse()
            sched = [(None, None, [a for a, _ in path], [b for _, b in path])]
        else:
            sched = P.spanningtrees_clusterlist(len(cn), ed, cn, net.is_leaf)
        sched = [(np.asarray(t[2], np.int32), np.asarray(t[3], np.int32)) for t in sched]
    desc, keep = L.make_desc(dims, sepcl, soff, sidx)
    lib = L.load()
    pl = C.c_void_p()
    assert lib.pgbp_plan_create(C.byref(desc), C.byref(pl)) == 0
    assert _set_sched(lib, pl, sched) == 0, lib.pgbp_plan_last_error(pl)
    sepcl = np.asarray(sepcl).reshape(-1, 2)
    for t, (pa, ch) in enumerate(sched):
        depth = {int(pa[0]): 0} if len(pa) else {}
        for a, c in zip(pa, ch):
            depth[int(c)] = depth[int(a)] + 1
        for d in (0, 1):
            lo, to, em, ee, er = _traversal(lib, pl, t, d)
            _check_traversal(lo, to, em, ee, pa, ch, sepcl, d)
            nlev = len(lo) - 1
            if fused and graph in ("bethe_tree", "bethe_net") and len(pa) > 20:
                assert nlev <= 0.65 * max(depth.values())
            if graph == "path":
                assert (nlev == 1 and to[1] - to[0] == len(pa)) if fused else nlev == len(pa)
    lib.pgbp_plan_destroy(pl)


def test_planner_under_address_and_undefined_sanitizers(tmp_path):
    """csrc/pgbp_plan.cpp (layout, message table, level schedules, chain fusion: host-only code) built with
    -fsanitize=address,undefined and driven through the pgbp_plan_* API over trees, Bethe and join graphs of networks,
    node-subtree schedules and malformed inputs, once with and once without chain fusion.  GPU sanitizers are not
    available on the pool; this is the host half."""
    import subprocess
    import sys
    so = str(tmp_path / "libpgbp_plan_asan.so")
    src = os.path.join(ROOT, "phylogaussianbeliefprop.jl_amd", "csrc", "pgbp_plan.cpp")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-fsanitize=address,undefined",
                           "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined", "-shared", "-o", so, src])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    for fuse in (False, True):
        env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
                   UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
        env.pop("PGBP_TUNING", None)
        if fuse:
            env["PGBP_TUNING"] = "chain_fusion"
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitize_plan.py"), so], env=env,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "sanitized planner ok" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_cliquetree_on_arrays_is_a_clique_tree_with_the_oracles_cliques():
    """clustergraph.py:cliquetree (elimination-tree construction, linear in the clique sizes) against the oracle's
    (all clique pairs + Kruskal): same maximal cliques, a tree, family-preserving, running intersection, and the same
    total sepset weight (both are maximum-weight spanning trees of the clique graph)."""
    import pgbp_amd as P
    from oracle import beliefs as OB
    from oracle import clustergraph as OCG
    from oracle import network as ON
    for seed in range(10):
        rng = np.random.default_rng(100 + seed)
        net = (ON.random_network(int(rng.integers(5, 60)), int(rng.integers(0, 14)), rng) if seed % 2 else
               ON.random_level3_network(int(rng.integers(6, 40)), int(rng.integers(1, 5)), rng))
        cn, ed, sn = P.cliquetree(OCG.nodefamilies(net))
        oct_ = OCG.cliquetree(net)
        assert sorted(cn) == sorted(n for _, n in oct_.clusters)
        assert len(ed) == len(cn) - 1
        assert sum(len(s) for s in sn) == sum(len(s) for _, _, s in oct_.edges)
        cg = OB.ClusterGraph([(str(i), n) for i, n in enumerate(cn)], [(a, b, s) for (a, b), s in zip(ed, sn)], "cliquetree")
        assert OCG.isfamilypreserving(cg, net) and OCG.check_runningintersection(cg, net)
        for (a, b), s in zip(ed, sn):
            assert s == sorted(set(cn[a]) & set(cn[b]), reverse=True) and s


def test_read_newick_equals_the_oracles_reader_on_the_golden_networks():
    """networks.read_newick (product, plain arrays) against the oracle's independent reader on every network string of
    the goldens: same tips, same number of nodes and hybrids, and the same parent sets / edge lengths / inheritances once
    nodes are identified by the tips below them."""
    import pgbp_amd as P
    from helpers import goldens
    from oracle import network as ON
    Gd = goldens()
    for key in ("doctest_lazaridis", "calibration_level3_joingraph", "joingraph_mateescu", "clustergraph_netstr",
                "exactBM_tree_calibrate", "canonicalform_six_messages"):
        s = Gd[key]["net"]
        net, names = P.read_newick(s)
        o = ON.read_newick(s)
        kids = [[] for _ in range(net.nnodes)]
        for nf in net.node2family:
            for pa in nf[1:]:
                kids[pa - 1].append(nf[0] - 1)
        below = [None] * net.nnodes
        for i in range(net.nnodes - 1, -1, -1):
            below[i] = frozenset([names[i]]) if net.is_leaf[i] else frozenset().union(*[below[c] for c in kids[i]])
        # a hybrid and the tree node right above / below it can share their tip set: add the node's own name if it has one
        tag = lambda nm, b: (tuple(sorted(b)), nm if not (nm[:1] == "I" and nm[1:].isdigit()) else "")
        a = sorted((tag(names[nf[0] - 1], below[nf[0] - 1]), sorted(tuple(sorted(below[x - 1])) for x in nf[1:]),
                    sorted(np.nan_to_num(net.length[i], nan=-1.0).tolist()), sorted(net.gamma[i]))
                   for i, nf in enumerate(net.node2family))
        ob = {}
        for n in reversed(o.vec_node):
            ob[id(n)] = frozenset([n.name]) if n.leaf else frozenset().union(*[ob[id(c)] for c in o.children(n)])
        b = sorted((tag(n.name, ob[id(n)]), sorted(tuple(sorted(ob[id(e.parent)])) for e in o.parent_edges(n)),
                    sorted(e.length for e in o.parent_edges(n)),
                    sorted((e.gamma if len(o.parent_edges(n)) > 1 else 1.0) for e in o.parent_edges(n))) for n in o.vec_node)
        assert len(a) == len(b) and net.nhybrids == sum(n.hybrid for n in o.nodes), key
        for x, y in zip(a, b):
            assert x[0] == y[0] and x[1] == y[1] and np.allclose(x[2], y[2]) and np.allclose(x[3], y[3]), (key, x, y)


def test_allocate_scopes_with_missing_data_equals_the_oracle():
    """networks.allocate_scopes with missing tip values (scattered, whole tips, a trait missing below whole subtrees so that
    internal nodes lose it) == the oracle's allocatebeliefs: dimensions, scopes of every cluster, both scopeindex maps."""
    import pgbp_amd as P
    from oracle import beliefs as OB
    from oracle import clustergraph as OCG
    from oracle import models as OM
    p = 3
    for seed in range(6):
        rng = np.random.default_rng(40 + seed)
        net = P.random_level3_network(int(rng.integers(8, 30)), int(rng.integers(1, 4)), rng)
        onet = _arrays_to_oracle_network(net)
        taxa = onet.tip_names
        row = {t: r for r, t in enumerate(taxa)}
        data = rng.normal(size=(len(taxa), p))
        data[rng.random(data.shape) < 0.35] = np.nan
        data[0] = 0.5                                   # at least one complete tip
        tbl = [[None if np.isnan(data[r, v]) else float(data[r, v]) for r in range(len(taxa))] for v in range(p)]
        data_row = [row.get(f"n{i + 1}", -1) for i in range(net.nnodes)]
        for kind in ("bethe", "cliquetree"):
            cn, ed, sn = P.bethe(net.node2family) if kind == "bethe" else P.cliquetree(net.node2family)
            ocg = OB.ClusterGraph([(str(i), n) for i, n in enumerate(cn)], [(a, b, s) for (a, b), s in zip(ed, sn)], kind)
            st = P.allocate_scopes(cn, ed, sn, net, p, data=data, data_row=data_row)
            ob, (n2c, n2f, n2fix, _, _) = OB.allocatebeliefs(tbl, taxa, onet, ocg, OM.MvDiagBrownianMotion(np.ones(p), np.zeros(p)))
            assert [b.dimension for b in ob] == st.dims.tolist()
            assert n2c == st.node2cluster
            for i in range(len(cn)):
                assert np.array_equal(ob[i].inscope, st.clusters[i].inscope)
            idx, off = [], [0]
            for j in range(len(cn), len(ob)):
                for c in ocg.edges[j - len(cn)][:2]:
                    idx += OB.scopeindex(ob[j], ob[c]).tolist()
                    off.append(len(idx))
            assert idx == st.scope_idx.tolist() and off == st.scope_off.tolist()


def test_clustergraphs_md_doctests_on_the_muller_network():
    """docs/src/man/clustergraphs.md:30-215 on the Müller et al. virus recombination network (801 nodes, 361 hybrids), with
    the product's host side only: reader + PhyloNetworks' preorder, then the four cluster-graph methods.  Every number the
    doctests print: cluster and edge counts, the summary statistics of the cluster sizes, the clique tree's first cluster
    by labels and preorder indices, the join graphs for k* = 10 and 54 (the latter a clique tree), the error for k* = 2,
    LTRIP from the join graph's clusters and from the node families."""
    import pgbp_amd as P
    from helpers import goldens
    g = goldens()["clustergraphs_muller2022"]
    with open(os.path.join(ROOT, "tests", "golden", "muller_2022.phy")) as f:
        net, names = P.read_newick(f.read())
    assert (net.nnodes, sum(len(nf) - 1 for nf in net.node2family), int(net.is_leaf.sum()), net.nhybrids) == \
        (g["nodes"], g["edges"], g["tips"], g["hybrids"])

    def check(cn, ed, want):
        a = np.array([len(c) for c in cn])
        assert (len(cn), len(ed)) == (want["clusters"], want["edges"])
        if "mean" in want:
            assert abs(a.mean() - want["mean"]) < 5e-7 and abs(a.std(ddof=1) - want["std"]) < 5e-7
            assert (a.min(), np.percentile(a, 25), np.median(a), np.percentile(a, 75), a.max()) == \
                (want["min"], want["q1"], want["median"], want["q3"], want["max"])
    fam = net.node2family
    cn, ed, sn = P.cliquetree(fam)
    check(cn, ed, g["cliquetree"])
    first = g["cliquetree"]["first_cluster_preorder"]
    assert first in cn and [names[v - 1] for v in first] == g["cliquetree"]["first_cluster_labels"]
    check(*P.bethe(fam)[:2], g["bethe"])
    jcn, jed, jsn = P.joingraph(fam, 10)
    check(jcn, jed, g["joingraph10"])
    with pytest.raises(ValueError) as ei:
        P.joingraph(fam, 2)
    assert str(ei.value) == g["joingraph2_error"]
    cn54, ed54, _ = P.joingraph(fam, 54)
    check(cn54, ed54, g["joingraph54"])
    lcn, led, _ = P.ltrip(fam, jcn)
    check(lcn, led, g["ltrip_of_joingraph10_clusters"])
    lcn, led, _ = P.ltrip(fam)
    check(lcn, led, g["ltrip"])


def test_read_newick_gives_the_preorders_the_reference_tests_imply():
    """PhyloNetworks.preorder! as restated in networks.read_newick reproduces the node numberings (and the names of unnamed
    internal nodes) that four of the reference's tests imply through cluster labels, cluster indices and the expected
    min-fill order (derived independently in tests/golden/make_reference_goldens.py)."""
    import pgbp_amd as P
    from helpers import goldens
    Gd = goldens()
    for key in ("calibration_level3_joingraph", "joingraph_mateescu"):
        assert P.read_newick(Gd[key]["net"])[1] == Gd[key]["preorder"]
    g = Gd["clustergraph_netstr"]
    net, names = P.read_newick(g["net"])
    inv = {tuple(sorted(v)): k for k, v in g["internal_names"].items()}
    want = [x if isinstance(x, str) else inv[tuple(sorted(x))] for x in g["preorder"]]
    assert names == want
    assert [names[v - 1] for v in P.triangulate_minfill(P.moralize(net.node2family))] == g["minfill_order_names"]


def _groups(lib, pl, tree, d, nlev):
    ng = np.zeros(nlev, np.int32)
    tl = C.c_int32()
    assert lib.pgbp_plan_groups(pl, tree, d, L.i32p(ng), C.byref(tl), None, None) == 0
    rec = np.zeros((max(1, int(ng.sum())), 4, 6), np.int32)
    trec = np.zeros((max(1, tl.value), 8, 6), np.int32)
    assert lib.pgbp_plan_groups(pl, tree, d, None, None, L.i32p(rec), L.i32p(trec)) == 0
    # the records' prologues (pgbp_plan_prologues) ride along as a 7th word: message id or -1
    pro = np.full((max(1, int(ng.sum())), 4), -1, np.int32)
    tpro = np.full((max(1, tl.value), 8), -1, np.int32)
    assert lib.pgbp_plan_prologues(pl, tree, d, L.i32p(pro), L.i32p(tpro), None) == 0
    rec = np.concatenate([rec, pro[:, :, None]], axis=2)
    trec = np.concatenate([trec, tpro[:, :, None]], axis=2)
    assert np.array_equal(rec[:, :, 6] >= 0, (rec[:, :, 5] & 8) != 0) and np.array_equal(trec[:, :, 6] >= 0, (trec[:, :, 5] & 8) != 0)
    return ng, tl.value, rec[:int(ng.sum())], trec[:tl.value]


def _record_msgs(r):
    """messages a record's wavefront sends, in order: its prologue (if any), then its own"""
    return ([int(r[6])] if r[6] >= 0 else []) + [int(r[1])]


@pytest.mark.parametrize("ntips,p,kind", [(2, 3, "random"), (3, 2, "random"), (40, 4, "random"), (500, 16, "random"),
                                          (300, 8, "random"), (30, 16, "caterpillar"), (70, 16, "poly4"),
                                          (120, 5, "random"), (90, 2, "poly6")])
def test_group_packing_and_tail_invariants(ntips, p, kind):
    """Launch form of the fast-class tasks (finalize_traversal in pgbp_plan.cpp): every level's tasks are packed into
    groups of 4 wavefront records, a task's messages consecutive and in schedule order inside ONE group, every message of
    the level exactly once, no group emptier than first-fit allows; the tail repeats the levels at the root end of the
    tree (last of a postorder, first of a preorder) as one group of 8 records per level, in level order."""
    rng = np.random.default_rng(ntips + p)
    if kind == "random":
        tr = S.random_tree(ntips, rng)
    elif kind == "caterpillar":
        tr = S.caterpillar_tree(ntips, rng)
    else:
        tr = S.random_multifurcating_tree(ntips, int(kind[4:]), rng)
    prob = S.cliquetree_of_tree(tr, p)
    lib, pl, code, keep = _plan(prob)
    assert code == 0, lib.pgbp_plan_last_error(pl)
    assert _set_sched(lib, pl, prob.schedule) == 0, lib.pgbp_plan_last_error(pl)

    def check_group(grp, W, level_tasks):
        """grp: [W][6] records; level_tasks: message tuple -> True for the tasks of the level not yet seen"""
        w = 0
        while w < W and grp[w, 0]:
            base, ln = int(grp[w, 2]), int(grp[w, 3])
            assert base == w and 1 <= ln <= 4 and w + ln <= W
            msgs = tuple(m for i in range(ln) for m in _record_msgs(grp[w + i]))
            assert level_tasks.pop(msgs, None) is not None, "a group's task is a task of its level, once"
            for i in range(ln):
                valid, msg, gb, gl, src, mode, pro = (int(x) for x in grp[w + i])
                assert valid == 1 and gb == base and gl == ln
                assert base <= src <= w + i, "the marginal comes from an earlier (or the same) wave of the task"
                mode &= ~8                                                  # (bit 3: the record has a prologue)
                if d == 0 and ln > 1:
                    assert mode & ~4 == (3 if i == 0 else 2) and src == w + i   # accumulate: first wave owns the block
                else:
                    assert mode == 1
                    assert src == w + i or er_of[msg] == 1                 # reuse only where the planner said so
            w += ln
        assert not grp[w:, 0].any(), "invalid records only behind the last task"
        return w

    for d in (0, 1):
        lo, to, em, ee, er = _traversal(lib, pl, 0, d)
        er_of = {int(m): int(r) for m, r in zip(em, er)}
        nlev = len(lo) - 1
        nf = np.zeros(nlev, np.int32)
        assert lib.pgbp_plan_level_nfast(pl, 0, d, L.i32p(nf)) == 0
        ng, tail_levels, rec, trec = _groups(lib, pl, 0, d, nlev)
        g0 = 0
        level_msgs = []
        for lv in range(nlev):
            tasks = {tuple(int(m) for m in em[to[t]:to[t + 1]]): True for t in range(lo[lv], lo[lv] + nf[lv])}
            n_msgs = sum(len(k) for k in tasks) - sum(int(er[e] == 2) for t in range(lo[lv], lo[lv] + nf[lv]) for e in range(to[t], to[t + 1]))
            level_msgs.append(n_msgs)     # wavefront records: a prologue rides in the record of the message behind it
            used = 0
            for g in range(g0, g0 + ng[lv]):
                used += check_group(rec[g], 4, tasks)
            assert not tasks and used == n_msgs
            # first fit, largest first: at most one group of the level is less than half full
            assert sum(1 for g in range(g0, g0 + ng[lv]) if rec[g, :, 0].sum() <= 1) <= 1 or n_msgs <= ng[lv]
            assert ng[lv] * 4 < 2 * n_msgs + 4
            g0 += ng[lv]
        # the tail: the maximal run of all-fast levels with at most 8 messages at the root end
        order = list(range(nlev - 1, -1, -1)) if d == 0 else list(range(nlev))
        want = 0
        for lv in order:
            if nf[lv] == lo[lv + 1] - lo[lv] and nf[lv] > 0 and level_msgs[lv] <= 8:
                want += 1
            else:
                break
        assert tail_levels == want
        for q in range(tail_levels):
            lv = nlev - tail_levels + q if d == 0 else q
            tasks = {tuple(int(m) for m in em[to[t]:to[t + 1]]): True for t in range(lo[lv], lo[lv + 1])}
            check_group(trec[q], 8, tasks)
            assert not tasks
    lib.pgbp_plan_destroy(pl)


def _header_structs_and_functions():
    """(struct name -> [(C type, field)], function name -> (return type, [C argument types])) of include/pgbp.h."""
    hdr = open(os.path.join(ROOT, "include", "pgbp.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    structs = {}
    for name, body in re.findall(r"typedef struct (pgbp_\w+) \{(.*?)\} \1;", hdr, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if decl:
                m = re.fullmatch(r"(.*?)\s*(\w+)", decl)
                fields.append((m.group(1).replace(" *", "*"), m.group(2)))
        structs[name] = fields
    funcs = {}
    for ret, name, args in re.findall(r"^\s*((?:const )?\w+\s*\*?)\s*(pgbp_\w+)\s*\((.*?)\)\s*;", hdr, flags=re.S | re.M):
        argt = []
        for a in args.split(","):
            a = " ".join(a.split())
            if a and a != "void":
                m = re.fullmatch(r"(.*?)\s*(\w+)", a)     # every prototype names its parameters
                argt.append(m.group(1).replace(" *", "*"))
        funcs[name] = (" ".join(ret.split()).replace(" *", "*"), argt)
    return structs, funcs


def _c_kind(ctype):
    """Layout class of a C type as a foreign-function binding sees it."""
    t = ctype.replace("const ", "").strip()
    if t.endswith("*"):
        return "ptr"
    return {"int32_t": "i32", "int": "i32", "int64_t": "i64", "uint64_t": "u64", "double": "f64", "void": "void"}[t]


def test_julia_shim_and_ctypes_mirror_agree_with_the_header():
    """The Julia @ccall shim (julia/PGBPDevice.jl) cannot run here (no Julia), so its declarations are
    checked against include/pgbp.h textually: every struct it mirrors has the header's fields in the header's order
    with the same machine types, and every @ccall names an exported function with the header's argument count, argument
    classes and return class.  The same for the ctypes structures and prototypes of the Python host mirror."""
    structs, funcs = _header_structs_and_functions()
    assert {"pgbp_desc", "pgbp_opts", "pgbp_result", "pgbp_lg_families", "pgbp_lg_params"} <= set(structs)
    jl = open(os.path.join(ROOT, "phylogaussianbeliefprop.jl_amd", "julia", "PGBPDevice.jl")).read()
    jkind = {"Int32": "i32", "Cint": "i32", "Int64": "i64", "UInt64": "u64", "Float64": "f64", "Cdouble": "f64",
             "Cvoid": "void", "Cstring": "ptr", "Ptr": "ptr"}   # ("Ptr": a return type `::Ptr{...}` cut at the brace)

    def jl_kind(t):
        t = t.strip()
        return "ptr" if t.startswith(("Ptr{", "Ref{")) else jkind[t]

    n_structs = 0
    for jname, cname, body in re.findall(r"^struct (\w+)\s*# (pgbp_\w+)\n(.*?)^end", jl, flags=re.S | re.M):
        body = re.sub(r"#.*", "", body)
        jfields = re.findall(r"(\w+)::([\w{}]+)", body)
        cfields = structs[cname]
        assert [f for f, _ in jfields] == [f for _, f in cfields], (jname, cname)
        assert [jl_kind(t) for _, t in jfields] == [_c_kind(t) for t, _ in cfields], (jname, cname)
        n_structs += 1
    assert n_structs >= 5
    calls = []
    for m in re.finditer(r"@ccall\(?\s*LIB\.(pgbp_\w+)\(", jl):
        i, depth = m.end(), 1
        while depth:                      # the matching parenthesis of the argument list
            depth += jl[i] in "([{"
            depth -= jl[i] in ")]}"
            i += 1
        ret = re.match(r"::(\w+)", jl[i:])
        calls.append((m.group(1), jl[m.end():i - 1], ret.group(1)))
    assert len(calls) >= 15
    for name, args, ret in calls:
        assert name in funcs, f"PGBPDevice.jl calls {name}, which include/pgbp.h does not declare"
        cret, cargs = funcs[name]
        # split the arguments at top-level commas; each ends in ::Type
        parts, depth, cur = [], 0, ""
        for ch in args:
            depth += ch in "([{"
            depth -= ch in ")]}"
            if ch == "," and depth == 0:
                parts.append(cur)
                cur = ""
            else:
                cur += ch
        if cur.strip():
            parts.append(cur)
        jtypes = [p.rsplit("::", 1)[1] for p in parts]
        assert len(jtypes) == len(cargs), (name, jtypes, cargs)
        assert [jl_kind(t) for t in jtypes] == [_c_kind(t) for t in cargs], (name, jtypes, cargs)
        assert jl_kind(ret) == _c_kind(cret), (name, ret, cret)
    # the ctypes mirror: structures field by field, prototypes argument by argument
    ckind = {C.c_int32: "i32", C.c_int: "i32", C.c_int64: "i64", C.c_uint64: "u64", C.c_double: "f64", None: "void"}

    def ct_kind(t):
        return ckind[t] if t in ckind else "ptr"

    for cname, cls in (("pgbp_desc", L.Desc), ("pgbp_opts", L.Opts), ("pgbp_result", L.Result),
                       ("pgbp_lg_families", L.LgFamilies), ("pgbp_lg_params", L.LgParams), ("pgbp_bm_tree", L.BmTree)):
        assert [f for f, _ in cls._fields_] == [f for _, f in structs[cname]], cname
        assert [ct_kind(t) for _, t in cls._fields_] == [_c_kind(t) for t, _ in structs[cname]], cname
    for name, (restype, argtypes) in L.SYMBOLS.items():
        cret, cargs = funcs[name]
        assert len(argtypes) == len(cargs), name
        assert [ct_kind(t) for t in argtypes] == [_c_kind(t) for t in cargs], (name, cargs)
        assert ct_kind(restype) == _c_kind(cret), name


def test_struct_layout_matches_the_c_compiler(tmp_path):
    """sizeof and every field offset of the ABI structs as gcc lays them out for include/pgbp.h, against (i) the ctypes
    mirror and (ii) the Julia struct declarations of PGBPDevice.jl laid out by Julia's own rule for isbits structs (C
    layout: each field at the next multiple of its alignment, the struct padded to its largest alignment)."""
    import subprocess
    structs, _ = _header_structs_and_functions()
    names = ["pgbp_desc", "pgbp_opts", "pgbp_result", "pgbp_lg_families", "pgbp_lg_params", "pgbp_bm_tree"]
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "pgbp.h")}"', "int main(void) {"]
    for n in names:
        src.append(f'  printf("{n} %zu", sizeof({n}));')
        for _, f in structs[n]:
            src.append(f'  printf(" %zu", offsetof({n}, {f}));')
        src.append('  printf("\\n");')
    src += ["  return 0;", "}"]
    c = tmp_path / "probe.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=c99", "-o", str(exe), str(c)])
    want = {}
    for line in subprocess.check_output([str(exe)], text=True).splitlines():
        parts = line.split()
        want[parts[0]] = [int(x) for x in parts[1:]]
    mirror = {"pgbp_desc": L.Desc, "pgbp_opts": L.Opts, "pgbp_result": L.Result, "pgbp_lg_families": L.LgFamilies,
              "pgbp_lg_params": L.LgParams, "pgbp_bm_tree": L.BmTree}
    for n in names:
        cls = mirror[n]
        got = [C.sizeof(cls)] + [getattr(cls, f).offset for f, _ in cls._fields_]
        assert got == want[n], (n, got, want[n])
    # Julia: isbits struct layout from the declared field types
    jl = open(os.path.join(ROOT, "phylogaussianbeliefprop.jl_amd", "julia", "PGBPDevice.jl")).read()
    size = {"Int32": 4, "Int64": 8, "UInt64": 8, "Float64": 8}
    seen = 0
    for jname, cname, body in re.findall(r"^struct (\w+)\s*# (pgbp_\w+)\n(.*?)^end", jl, flags=re.S | re.M):
        body = re.sub(r"#.*", "", body)
        off, offs, amax = 0, [], 1
        for _, t in re.findall(r"(\w+)::([\w{}]+)", body):
            sz = 8 if t.startswith(("Ptr{", "Ref{")) else size[t]
            off = (off + sz - 1) // sz * sz
            offs.append(off)
            off += sz
            amax = max(amax, sz)
        total = (off + amax - 1) // amax * amax
        assert [total] + offs == want[cname], (jname, cname, [total] + offs, want[cname])
        seen += 1
    assert seen >= 5


def _chunks(lib, pl, tree, d):
    n = C.c_int32()
    assert lib.pgbp_plan_chunks(pl, tree, d, C.byref(n), None, None, None) == 0
    info = np.zeros((max(1, n.value), 4), np.int32)
    assert lib.pgbp_plan_chunks(pl, tree, d, C.byref(n), L.i32p(info), None, None) == 0
    info = info[:n.value]
    wg = np.zeros(max(1, int((info[:, 2] + 1).sum())), np.int32)
    rec = np.zeros((max(1, int(np.abs(info[:, 3]).sum())), 8, 6), np.int32)
    assert lib.pgbp_plan_chunks(pl, tree, d, C.byref(n), None, L.i32p(wg), L.i32p(rec)) == 0
    cpro = np.full((rec.shape[0], 8), -1, np.int32)          # the records' prologues as a 7th word
    assert lib.pgbp_plan_prologues(pl, tree, d, None, None, L.i32p(cpro)) == 0
    rec = np.concatenate([rec, cpro[:, :, None]], axis=2)
    out, w0, g0 = [], 0, 0
    for (l0, l1, nwg, ng) in info:
        generic = ng < 0          # groups of 8 task ids instead of 8 message records
        ng = abs(int(ng))
        offs = wg[w0:w0 + nwg + 1]
        out.append((int(l0), int(l1), [rec[g0 + offs[b]:g0 + offs[b + 1]] for b in range(nwg)], generic))
        assert offs[0] == 0 and offs[-1] == ng and np.all(np.diff(offs) > 0)
        w0 += nwg + 1
        g0 += ng
    return out


@pytest.mark.parametrize("ntips,p,kind,graph,tuning", [
    (3000, 16, "random", "cliquetree", ""), (800, 4, "random", "cliquetree", ""), (60, 16, "caterpillar", "cliquetree", ""),
    (900, 8, "random", "bethe", ""), (400, 3, "poly4", "cliquetree", ""), (700, 4, "network", "joingraph", ""),
    (500, 2, "network", "bethe", ""), (300, 3, "poly7", "cliquetree", ""), (80, 16, "poly3", "cliquetree", ""),
    (120, 6, "poly7", "cliquetree", ""), (9000, 2, "random", "cliquetree", ""), (6000, 3, "network", "joingraph", ""),
    # the trees of every chunk's forest packed into a handful of workgroups (PGBP_TUNING, read when the plan is built)
    (3000, 16, "random", "cliquetree", "chunk_bins=5"), (900, 8, "random", "bethe", "chunk_bins=2"),
    (700, 4, "network", "joingraph", "chunk_bins=3"), (400, 3, "poly4", "cliquetree", "chunk_bins=1"),
    (800, 4, "random", "cliquetree", "chunk_bins=0")])
def test_fused_chunks_are_dependency_closed(ntips, p, kind, graph, tuning, monkeypatch):
    """Chunks of fused levels (build_chunks in pgbp_plan.cpp): replaying one calibrate iteration launch by launch --
    level launches, chunk launches (their workgroups in ANY order: checked forwards and backwards), the tail -- every
    message finds what it depends on done either by an earlier launch or by an earlier step of its OWN workgroup; what a
    workgroup writes no other workgroup of the launch reads or writes; every message runs exactly once."""
    rng = np.random.default_rng(ntips + p)
    if tuning:
        monkeypatch.setenv("PGBP_TUNING", tuning)
    if kind == "network":
        # loopy cluster graphs of a level-3 network: generic-class tasks (hybrid families, 2-node sepsets), several trees
        import pgbp_amd as P

        class Prob:
            pass
        net = P.random_level3_network_varied(ntips, ntips // 4, rng, n_colors=2)
        cn, ed, sn = P.joingraph(net.node2family, 3) if graph == "joingraph" else P.bethe(net.node2family)
        st = P.allocate_scopes(cn, ed, sn, net, p)
        prob = Prob()
        prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx = st.dims, st.sepset_clusters, st.scope_off, st.scope_idx
        prob.schedule = [(np.asarray(t[2]), np.asarray(t[3])) for t in P.spanningtrees_clusterlist(len(cn), ed, cn, net.is_leaf)][:1]
    else:
        if kind == "random":
            tr = S.random_tree(ntips, rng)
        elif kind == "caterpillar":
            tr = S.caterpillar_tree(ntips, rng)
        else:
            tr = S.random_multifurcating_tree(ntips, int(kind[4:]), rng)
        prob = S.cliquetree_of_tree(tr, p) if graph == "cliquetree" else S.bethe_of_tree(tr, p)
    lib, pl, code, keep = _plan(prob)
    assert code == 0, lib.pgbp_plan_last_error(pl)
    assert _set_sched(lib, pl, prob.schedule) == 0, lib.pgbp_plan_last_error(pl)
    pa, ch = (np.asarray(x) for x in prob.schedule[0])
    sc = np.asarray(prob.sepset_clusters).reshape(-1, 2)
    parent = {int(c): int(a) for a, c in zip(pa, ch)}
    children = {}
    for a, c in zip(pa, ch):
        children.setdefault(int(a), []).append(int(c))
    edge_of = {(int(a), int(c)): i for i, (a, c) in enumerate(zip(pa, ch))}

    def ends(m):
        k, side = divmod(int(m), 2)
        return int(sc[k][1 - side]), int(sc[k][side]), k     # sender, receiver, sepset

    def group_tasks(grp):
        w, tasks = 0, []
        while w < grp.shape[0] and grp[w, 0]:
            ln = int(grp[w, 3])
            tasks.append([m for i in range(ln) for m in _record_msgs(grp[w + i])])
            w += ln
        assert not grp[w:, 0].any()
        return tasks

    n_chunk_launches = packed_launches = 0
    done = set()
    recv_order = {}
    for d in (0, 1):
        lo, to, em, ee, er = _traversal(lib, pl, 0, d)
        nlev = len(lo) - 1
        ng, tl, rec, trec = _groups(lib, pl, 0, d, nlev)
        chunks = {c[0]: c for c in _chunks(lib, pl, 0, d)}
        tail = set(range(nlev - tl, nlev)) if d == 0 else set(range(tl))
        launches = []      # each: list of workgroups; a workgroup: list of steps; a step: list of tasks
        Lv = 0
        while Lv < nlev:
            if Lv in chunks:
                l0, l1, wgs, generic = chunks[Lv]
                assert not (set(range(l0, l1)) & tail), "chunks lie below the tail"

                def tasks_of(g):
                    if not generic:
                        return group_tasks(g)
                    ids = [int(r[1]) for r in g if r[0]]
                    assert all(r[0] for r in g[:len(ids)]) and len(ids) >= 1
                    # eight working matrices share one workgroup's LDS: small senders only (kChunkGenericMaxMf)
                    assert all(int(prob.dims[ends(m)[0]]) <= 24 for t in ids for m in em[to[t]:to[t + 1]])
                    return [[int(m) for m in em[to[t]:to[t + 1]]] for t in ids]
                want = sorted(sorted(int(m) for m in em[to[t]:to[t + 1]]) for t in range(lo[l0], lo[l1]))
                got = sorted(sorted(tk) for w in wgs for g in w for tk in tasks_of(g))
                assert want == got, "a chunk runs exactly the tasks of its levels"
                launches.append([[tasks_of(g) for g in w] for w in wgs])
                n_chunk_launches += 1
                packed_launches += len(wgs) == 256 and min(lo[L + 1] - lo[L] for L in range(l0, l1)) > 256
                if tuning.startswith("chunk_bins=") and int(tuning.split("=")[1]) > 0:
                    assert len(wgs) <= int(tuning.split("=")[1])
                Lv = l1
            else:
                launches.append([[[[int(m) for m in em[to[t]:to[t + 1]]]] for t in range(lo[Lv], lo[Lv + 1])]])
                Lv += 1
        for launch in launches:
            for wg_order in (launch, launch[::-1]):
                touched, read_by = {}, {}
                for wi, wg in enumerate(wg_order):
                    local = set()
                    for step in wg:
                        readers, writers = {}, {}
                        for ti, tk in enumerate(step):
                            mine = set()      # delivered earlier in this very task (a prologue in front of its record's own message)
                            for m in tk:
                                s_, r_, k_ = ends(m)
                                pre = [(c, s_) for c in children.get(s_, []) if c != r_]
                                if parent.get(s_) != r_:                       # a preorder message: parent first, all children
                                    pre = [(c, s_) for c in children.get(s_, [])]
                                    if s_ in parent:
                                        pre.append((parent[s_], s_))
                                for q in pre:
                                    assert q in done or q in local or q in mine, (d, m, q)
                                mine.add((s_, r_))
                                readers.setdefault(s_, set()).add(ti)
                                assert writers.setdefault(r_, ti) == ti
                                # what a workgroup WRITES (the receiver, the sepset) no other workgroup of the launch
                                # touches; a sender that nobody writes may be read by several (preorder: one task
                                # per message of a generic-class sender)
                                for obj in (("c", r_), ("s", k_)):
                                    assert touched.setdefault(obj, wi) == wi, "two workgroups of one launch share a belief"
                                    assert read_by.get(obj, {wi}) == {wi}, "a workgroup writes what another one reads"
                                assert touched.get(("c", s_), wi) == wi, "a workgroup reads what another one writes"
                                read_by.setdefault(("c", s_), set()).add(wi)
                        # a cluster written in a step is read in it only by the task that writes it (a prologue's F)
                        assert all(readers.get(c, {ti}) == {ti} for c, ti in writers.items())
                        for tk in step:
                            for m in tk:
                                local.add(ends(m)[:2])
            for wg in launch:
                for step in wg:
                    for tk in step:
                        for m in tk:
                            s_, r_, _ = ends(m)
                            assert (s_, r_) not in done
                            done.add((s_, r_))
                            recv_order.setdefault((d, r_), []).append(m)
    assert len(done) == 2 * len(pa)
    # messages into one receiver: every one of them once (within a task the planner keeps the reference's order, which
    # test_level_schedule_invariants checks; across levels the level schedule itself reorders: DESIGN.md section 3)
    for (d, r_), msgs in recv_order.items():
        want = len(children.get(r_, [])) if d == 0 else 1
        assert len(msgs) == want and len(set(msgs)) == len(msgs)
    if ntips >= 800:
        assert n_chunk_launches >= 2
    if ntips >= 6000:
        # a forest of more trees than the chip has CUs: its trees share workgroups (kChunkBins), the launch stays closed
        assert packed_launches >= 1
    lib.pgbp_plan_destroy(pl)


def test_level3_networks_have_moral_cliques_of_at_most_4_nodes():
    """Why BASELINE configs[4] runs join-graph structuring with maxclustersize 3: the moral graph of a network of level
    <= 3 (a tree plus at most 3 reticulations per blob, plus one marrying edge per hybrid) triangulates (min-fill) into
    cliques of at most 4 nodes on every blob the varied generator draws, so JoinGraphStructuring(6) never splits a bucket
    (its join graph is a clique tree) while JoinGraphStructuring(3) is loopy as soon as one 4-clique exists."""
    import pgbp_amd as P
    seen4 = 0
    for seed in range(12):
        rng = np.random.default_rng(100 + seed)
        net = P.random_level3_network_varied(300, 120, rng, n_colors=2)
        assert net.nhybrids >= 60
        # level: no node has more than two parents, and the generator's sites hold at most 3 hybrids each
        assert max(len(nf) for nf in net.node2family) <= 3
        cn, ed, sn = P.cliquetree(net.node2family)
        sizes = [len(c) for c in cn]
        assert max(sizes) <= 4
        seen4 += sum(1 for x in sizes if x == 4)
        cn6, ed6, sn6 = P.joingraph(net.node2family, 6)
        assert len(ed6) == len(cn6) - 1                      # a tree
        cn3, ed3, sn3 = P.joingraph(net.node2family, 3)
        assert max(len(c) for c in cn3) <= 3
        if 4 in sizes:
            assert len(ed3) > len(cn3) - 1                   # loopy
    assert seen4 >= 10


@pytest.mark.parametrize("ntips,p,kind,graph", [(300, 4, "network", "joingraph"), (200, 3, "network", "bethe"),
                                                (90, 5, "poly7", "cliquetree"), (40, 24, "random", "cliquetree"),
                                                (25, 30, "poly4", "cliquetree")])
def test_message_records_of_the_wave_per_task_kernels(ntips, p, kind, graph):
    """The self-contained 128-byte records (pgbp_plan_records; struct GRec): for every generic-class task the chain
    first record -> next -> ... -1 lists the task's messages in order; a level's first records are consecutive in task
    order; dimensions and index maps are those of the message (sender / receiver / sepset dimensions from the problem,
    kept and updated positions from the scope maps, integrated variables = the rest, ascending); contiguous maps are
    flagged as such; generic PREORDER tasks hold one message each (one sender, k children = k independent messages)."""
    rng = np.random.default_rng(7 * ntips + p)
    if kind == "network":
        import pgbp_amd as P

        class Prob:
            pass
        net = P.random_level3_network_varied(ntips, ntips // 4, rng, n_colors=2)
        cn, ed, sn = P.joingraph(net.node2family, 3) if graph == "joingraph" else P.bethe(net.node2family)
        st = P.allocate_scopes(cn, ed, sn, net, p)
        prob = Prob()
        prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx = st.dims, st.sepset_clusters, st.scope_off, st.scope_idx
        prob.schedule = [(np.asarray(t[2]), np.asarray(t[3])) for t in P.spanningtrees_clusterlist(len(cn), ed, cn, net.is_leaf)][:1]
    else:
        tr = S.random_tree(ntips, rng) if kind == "random" else S.random_multifurcating_tree(ntips, int(kind[4:]), rng)
        prob = S.cliquetree_of_tree(tr, p)
    lib, pl, code, keep = _plan(prob)
    assert code == 0, lib.pgbp_plan_last_error(pl)
    assert _set_sched(lib, pl, prob.schedule) == 0, lib.pgbp_plan_last_error(pl)
    dims = np.asarray(prob.dims, dtype=np.int64)
    sc = np.asarray(prob.sepset_clusters).reshape(-1, 2)
    nc = len(dims) - len(sc)
    soff, sidx = np.asarray(prob.scope_off), np.asarray(prob.scope_idx)
    rec_t = np.dtype([("from_off", "<i8"), ("to_off", "<i8"), ("sep_off", "<i8"), ("res_off", "<i8"),
                      ("msg", "<i4"), ("seq", "<i4"), ("from_b", "<i4"), ("to_b", "<i4"),
                      ("keep_map", "<i4"), ("up_map", "<i4"), ("int_map", "<i4"), ("next", "<i4"),
                      ("mf", "u1"), ("mt", "u1"), ("s", "u1"), ("ni", "u1"), ("keep0", "u1"), ("up0", "u1"),
                      ("reuse", "u1"), ("inl", "u1"), ("perm", "u1", 40), ("up", "u1", 16)])
    assert rec_t.itemsize == 128
    seen_generic = 0
    for d in (0, 1):
        lo, to, em, ee, er = _traversal(lib, pl, 0, d)
        nlev, ntask = len(lo) - 1, len(to) - 1
        n = C.c_int32()
        assert lib.pgbp_plan_records(pl, 0, d, C.byref(n), None, None, None) == 0
        lf, tf = np.zeros(max(1, nlev), np.int32), np.zeros(max(1, ntask), np.int32)
        raw = np.zeros(max(1, n.value) * 128, np.uint8)
        assert lib.pgbp_plan_records(pl, 0, d, C.byref(n), L.i32p(lf), L.i32p(tf), raw.ctypes.data) == 0
        recs = raw.view(rec_t)[:n.value]
        used = np.zeros(n.value, bool)
        for Lv in range(nlev):
            firsts = [int(tf[t]) for t in range(lo[Lv], lo[Lv + 1]) if tf[t] >= 0]
            if firsts:
                assert firsts == list(range(int(lf[Lv]), int(lf[Lv]) + len(firsts))), "a level's first records: consecutive"
            for t in range(lo[Lv], lo[Lv + 1]):
                if tf[t] < 0:
                    continue
                seen_generic += 1
                msgs = [int(m) for m in em[to[t]:to[t + 1]]]
                if d == 1 and int(dims.max()) > 2:
                    senders = {int(sc[m // 2][1 - m % 2]) for m in msgs}
                    assert len(msgs) == 1 or len(senders) > 1, "a generic preorder task = one message (unless chain-fused)"
                q, chain = int(tf[t]), []
                while q >= 0:
                    assert not used[q]
                    used[q] = True
                    chain.append(q)
                    q = int(recs[q]["next"])
                assert [int(recs[q]["msg"]) for q in chain] == msgs
                for q, m in zip(chain, msgs):
                    r = recs[q]
                    k, side = divmod(m, 2)
                    receiver, sender = int(sc[k][side]), int(sc[k][1 - side])
                    s_, mf, mt = int(dims[nc + k]), int(dims[sender]), int(dims[receiver])
                    assert (int(r["from_b"]), int(r["to_b"])) == (sender, receiver)
                    assert (int(r["mf"]), int(r["mt"]), int(r["s"]), int(r["ni"])) == (mf, mt, s_, mf - s_)
                    # scope maps of the sepset: side 0 = its position in cluster sc[k][0], side 1 = in sc[k][1]
                    keep = sidx[soff[2 * k + (1 - side)]: soff[2 * k + (1 - side) + 1]].tolist()
                    upd = sidx[soff[2 * k + side]: soff[2 * k + side + 1]].tolist()
                    assert len(keep) == len(upd) == s_
                    integ = [v for v in range(mf) if v not in set(keep)]
                    contiguous = lambda v: len(v) > 0 and v == list(range(v[0], v[0] + len(v)))
                    if contiguous(keep) or s_ == 0:
                        assert s_ == 0 or int(r["keep0"]) == keep[0]
                    else:
                        assert int(r["keep0"]) == 255
                        if mf <= 40:
                            assert int(r["inl"]) & 1 and r["perm"][:mf].tolist() == integ + keep
                    if contiguous(upd) or s_ == 0:
                        assert s_ == 0 or int(r["up0"]) == upd[0]
                    else:
                        assert int(r["up0"]) == 255
                        if s_ <= 16:
                            assert int(r["inl"]) & 2 and r["up"][:s_].tolist() == upd
        assert used.all(), "every record belongs to exactly one task"
    assert seen_generic > 0
    lib.pgbp_plan_destroy(pl)


def test_residual_norm_thresholds_are_exact():
    """pgbp_residual_threshold(c, atol) (host): the largest x with fl(x / c) <= atol -- what the kernels compare the
    residual maxima with instead of dividing (iscalibrated_residnorm!, src/beliefs.jl:994-1003).  For divisors sqrt(s)
    and s, tolerances over 300 orders of magnitude (subnormal ones included): x passes, its upper neighbour does not; the
    special cases of the header."""
    lib = L.load()
    rng = np.random.default_rng(5)
    f = lib.pgbp_residual_threshold
    for s in list(range(1, 130)) + [1000, 4096]:
        for c in (float(np.sqrt(float(s))), float(s)):
            tols = np.concatenate([10.0 ** rng.uniform(-300, 300, 40), [1e-5, 1.0, 5e-324, 2.2250738585072014e-308, 1e308, 0.0],
                                   rng.random(10)])
            for atol in tols.tolist():
                x = f(c, float(atol))
                if np.isinf(x):
                    assert atol * c > 1.7e308          # everything finite passes
                    continue
                assert x >= 0.0 and np.float64(x) / np.float64(c) <= atol, (s, c, atol, x)
                with np.errstate(over="ignore"):
                    up = np.nextafter(x, np.inf)
                assert np.isinf(up) or np.float64(up) / np.float64(c) > atol, (s, c, atol, x)
    assert f(0.0, 1e-5) == np.inf and f(3.0, np.inf) == np.inf
    assert f(3.0, -1.0) == -1.0 and f(3.0, float("nan")) == -1.0


@pytest.mark.parametrize("ntips,p", [(300, 8), (57, 5), (200, 16)])
def test_bethe_graph_of_a_tree_is_scheduled_with_prologues(ntips, p):
    """Prologue fusion (build_traversals, fuse = 2): in the Bethe graph of a tree every factor cluster F = {v, pa(v)} has
    one child and one parent in the schedule tree, and the one message it receives in a traversal (from a variable
    cluster: nothing integrated) lands on the block its own message integrates out.  That message is planned as the
    PROLOGUE of F's own -- the entry in front of it in the same task (entry_reuse = 2), the 7th word of its record -- in
    both traversals; no variable-to-factor message is left as a record of its own, so the levels are those of the
    factor-to-variable messages alone; a clique tree has no such pair and keeps its schedule."""
    rng = np.random.default_rng(ntips)
    tr = S.random_tree(ntips, rng)
    prob = S.bethe_of_tree(tr, p)
    lib, pl, code, keep = _plan(prob)
    assert code == 0 and _set_sched(lib, pl, prob.schedule) == 0
    pa, ch = (np.asarray(x) for x in prob.schedule[0])
    sc = np.asarray(prob.sepset_clusters).reshape(-1, 2)
    nchild = {}
    for a in pa:
        nchild[int(a)] = nchild.get(int(a), 0) + 1
    has_parent = {int(c) for c in ch}
    dims = np.asarray(prob.dims)

    def ends(m):
        k, side = divmod(int(m), 2)
        return int(sc[k][1 - side]), int(sc[k][side])     # sender, receiver
    fusable = {c for c in nchild if nchild[c] == 1 and c in has_parent and dims[c] == 2 * p}
    assert len(fusable) >= ntips // 2                       # the factor clusters of the internal edges
    for d in (0, 1):
        lo, to, em, ee, er = _traversal(lib, pl, 0, d)
        n_pro = 0
        for t in range(len(to) - 1):
            for e in range(to[t], to[t + 1]):
                s_, r_ = ends(em[e])
                if er[e] == 2:
                    n_pro += 1
                    assert e + 1 < to[t + 1] and er[e + 1] != 2
                    s2, r2 = ends(em[e + 1])
                    assert r_ == s2 and r_ in fusable and dims[s_] == p    # X -> F in front of F -> Y, X a variable cluster
                else:
                    # a message that is not a prologue never goes INTO a cluster that could have taken it as one
                    assert not (r_ in fusable and dims[s_] == p and (s_, r_) in {(int(c), int(a)) for a, c in zip(pa, ch)} and d == 0)
        assert n_pro == len(fusable)
        nlev = len(lo) - 1
        ng, tl, rec, trec = _groups(lib, pl, 0, d, nlev)
        assert int((rec[:, :, 6] >= 0).sum()) + int((trec[:, :, 6] >= 0).sum()) > 0
    # height of the schedule tree in clusters ~ 2 x (levels of factor-to-variable messages): the fused schedule has
    # about half as many levels as the tree is high
    depth = {int(pa[0]): 0}
    for a, c in zip(pa, ch):
        depth[int(c)] = depth[int(a)] + 1
    lo0 = _traversal(lib, pl, 0, 0)[0]
    assert len(lo0) - 1 <= max(depth.values()) // 2 + 2
    lib.pgbp_plan_destroy(pl)
    # a clique tree: no prologues
    prob = S.cliquetree_of_tree(tr, p)
    lib, pl, code, keep = _plan(prob)
    assert code == 0 and _set_sched(lib, pl, prob.schedule) == 0
    for d in (0, 1):
        assert not (_traversal(lib, pl, 0, d)[4] == 2).any()
    lib.pgbp_plan_destroy(pl)


@pytest.mark.parametrize("ntips,p,kind,graph", [(3000, 16, "random", "cliquetree"), (900, 8, "random", "bethe"),
                                                (60, 16, "caterpillar", "cliquetree"), (400, 4, "poly4", "cliquetree"),
                                                (300, 6, "random", "bethe")])
def test_loop_launches_read_nothing_the_pass_before_wrote_except_through_a_chain(ntips, p, kind, graph):
    """Chains and late groups of the loop launches (link_chains in pgbp_plan.cpp, pgbp_loop.hip): the workgroup of a tail or
    chunk loads a group's operands while the group before it is still being stored.  Replaying every walk group by group:
    whatever a record of a group that is not `late` reads -- its sender, its prologue's X, its sepset(s), its receiver --
    was not written by the group before it, EXCEPT the one block a chain names: then the chain's source record of the
    group before is the owner of that very cluster, and the chain covers exactly what was written (the integrated block
    of a 2P sender, a P-dim cluster whole).  The first group of a walk is late; on a tree the tail has late groups only
    at its start and where the postorder turns into the preorder, so that the chains are what the tail runs on."""
    rng = np.random.default_rng(ntips + p)
    if kind == "random":
        tr = S.random_tree(ntips, rng)
    elif kind == "caterpillar":
        tr = S.caterpillar_tree(ntips, rng)
    else:
        tr = S.random_multifurcating_tree(ntips, int(kind[4:]), rng)
    prob = S.cliquetree_of_tree(tr, p) if graph == "cliquetree" else S.bethe_of_tree(tr, p)
    lib, pl, code, keep = _plan(prob)
    assert code == 0 and _set_sched(lib, pl, prob.schedule) == 0
    sc = np.asarray(prob.sepset_clusters).reshape(-1, 2)
    dims = np.asarray(prob.dims)

    def ends(m):
        k, side = divmod(int(m), 2)
        return int(sc[k][1 - side]), int(sc[k][side]), k     # sender, receiver, sepset

    def check_walk(groups, chains):
        """groups: [n][8][7] records (+ prologue word); chains: [n][8] words"""
        n_chain = n_late = 0
        for gi in range(len(groups)):
            late = [int(c) >> 16 & 1 for c in chains[gi]]
            assert len(set(late)) == 1, "late is a property of the group"
            if gi == 0:
                assert late[0] == 1
            if late[0]:
                n_late += 1
                assert not any(int(c) & 0xff for c in chains[gi])
                continue
            prev = groups[gi - 1]
            wrote_cl, wrote_sep, owner = set(), set(), {}
            for w in range(8):
                if not prev[w, 0]:
                    continue
                for m in _record_msgs(prev[w]):
                    s_, r_, k_ = ends(m)
                    wrote_sep.add(k_)
                    if m != int(prev[w, 1]):
                        wrote_cl.add(r_)                      # a prologue stores into F
                if prev[w, 5] & 1:                            # kFOwn: this record stores the receiver
                    r_ = ends(prev[w, 1])[1]
                    wrote_cl.add(r_)
                    owner[r_] = w
            for w in range(8):
                rec = groups[gi][w]
                if not rec[0]:
                    continue
                kind, src = int(chains[gi][w]) & 0xff, int(chains[gi][w]) >> 8 & 0xff
                s_, r_, k_ = ends(rec[1])
                assert k_ not in wrote_sep
                if rec[5] & 1:
                    assert r_ not in wrote_cl, "a receiver written in the pass before: the group must be late"
                provider = int(rec[4]) == w
                x_ = ends(rec[6])[0] if rec[6] >= 0 else None
                if rec[6] >= 0:
                    assert ends(rec[6])[2] not in wrote_sep
                if provider and s_ in wrote_cl:
                    assert kind in (1, 3) and owner.get(s_) == src
                    assert (kind == 1) == (dims[s_] == 2 * p)
                elif kind in (1, 3):
                    assert False, "a chain without a write to carry"
                if x_ is not None and x_ in wrote_cl:
                    assert kind == 2 and owner.get(x_) == src and dims[x_] == p
                elif kind == 2:
                    assert False, "a chain without a write to carry"
                n_chain += kind != 0
        return n_chain, n_late

    for d in (0, 1):
        lo = _traversal(lib, pl, 0, d)[0]
        nlev = len(lo) - 1
        ng, tl, rec, trec = _groups(lib, pl, 0, d, nlev)
        ch = _chunks(lib, pl, 0, d)
        n_rec_chunks = sum(len(g) for c in ch for w in c[2] for g in [w]) if ch else 0
        tail_chain = np.zeros((max(1, tl), 8), np.int32)
        chunk_chain = np.zeros((max(1, sum(len(w) for c in ch for w in c[2])), 8), np.int32)
        assert lib.pgbp_plan_chains(pl, 0, d, L.i32p(tail_chain), L.i32p(chunk_chain)) == 0
        if d == 0:
            post_tail = (trec, tail_chain[:tl].copy())
        else:
            # the tail launch walks the postorder's tail, then the preorder's, as one sequence
            groups = np.concatenate([post_tail[0], trec]) if tl + len(post_tail[0]) else trec
            chains = np.concatenate([post_tail[1], tail_chain[:tl]])
            if len(groups):
                n_chain, n_late = check_walk(groups, chains)
                if ntips >= 300 and graph == "cliquetree" and kind == "random":
                    assert n_late <= len(groups) // 2 and n_chain >= len(groups) // 2
        o = 0
        for (l0, l1, wgs, generic) in ch:
            for w in wgs:
                if not generic and len(w):
                    check_walk(w, chunk_chain[o:o + len(w)])
                o += len(w)
    lib.pgbp_plan_destroy(pl)


@pytest.mark.parametrize("ntips,p,graph", [(300, 4, "joingraph"), (200, 3, "bethe"), (150, 2, "joingraph")])
def test_row_form_of_small_postorder_levels(ntips, p, graph):
    """pgbp_plan_rows: on a network's cluster graph (small clusters, a few traits) every postorder level of generic-class
    tasks has a row form -- each message of the level in exactly one row, the rows of a task consecutive, in the task's order
    (position 0 .. k - 1, k <= 4), inside ONE wavefront of four rows; empty rows only where a task did not fit; the preorder
    (one message per task) has none."""
    import pgbp_amd as P
    rng = np.random.default_rng(11 * ntips + p)

    class Prob:
        pass
    net = P.random_level3_network_varied(ntips, ntips // 4, rng, n_colors=2)
    cn, ed, sn = P.joingraph(net.node2family, 3) if graph == "joingraph" else P.bethe(net.node2family)
    st = P.allocate_scopes(cn, ed, sn, net, p)
    prob = Prob()
    prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx = st.dims, st.sepset_clusters, st.scope_off, st.scope_idx
    prob.schedule = [(np.asarray(t[2]), np.asarray(t[3])) for t in P.spanningtrees_clusterlist(len(cn), ed, cn, net.is_leaf)][:1]
    lib, pl, code, keep = _plan(prob)
    assert code == 0, lib.pgbp_plan_last_error(pl)
    assert _set_sched(lib, pl, prob.schedule) == 0, lib.pgbp_plan_last_error(pl)
    rec_t = np.dtype([("off", "<i8", 4), ("msg", "<i4"), ("seq", "<i4"), ("from_b", "<i4"), ("to_b", "<i4"),
                      ("maps", "<i4", 3), ("next", "<i4"), ("dims", "u1", 4), ("fl", "u1", 4), ("perm", "u1", 40), ("up", "u1", 16)])
    assert rec_t.itemsize == 128
    levels_with_rows = 0
    for d in (0, 1):
        lo, to, em, ee, er = _traversal(lib, pl, 0, d)
        nlev, ntask = len(lo) - 1, len(to) - 1
        n = C.c_int32()
        assert lib.pgbp_plan_records(pl, 0, d, C.byref(n), None, None, None) == 0
        lf, tf = np.zeros(max(1, nlev), np.int32), np.zeros(max(1, ntask), np.int32)
        raw = np.zeros(max(1, n.value) * 128, np.uint8)
        assert lib.pgbp_plan_records(pl, 0, d, C.byref(n), L.i32p(lf), L.i32p(tf), raw.ctypes.data) == 0
        recs = raw.view(rec_t)[:n.value]
        nr = C.c_int64()
        assert lib.pgbp_plan_rows(pl, 0, d, C.byref(nr), None, None, None) == 0
        r0, rn = np.zeros(max(1, nlev), np.int64), np.zeros(max(1, nlev), np.int32)
        rm = np.zeros(max(1, 2 * nr.value), np.int32)
        assert lib.pgbp_plan_rows(pl, 0, d, C.byref(nr), r0.ctypes.data_as(C.POINTER(C.c_int64)), L.i32p(rn), L.i32p(rm)) == 0
        if d == 1:
            assert nr.value == 0 and not rn.any()
            continue
        for Lv in range(nlev):
            gen = [t for t in range(lo[Lv], lo[Lv + 1]) if tf[t] >= 0]
            if not gen:
                assert rn[Lv] == 0
                continue
            sizes = [to[t + 1] - to[t] for t in gen]
            small = True   # every message of the level: at most 8 integrated and 8 kept variables
            for t in gen:
                q = int(tf[t])
                while q >= 0:
                    small = small and recs[q]["dims"][3] <= 8 and recs[q]["dims"][2] <= 8
                    q = int(recs[q]["next"])
            if max(sizes) > 4 or not small:
                assert rn[Lv] == 0
                continue
            assert rn[Lv] > 0 and rn[Lv] % 4 == 0
            levels_with_rows += 1
            rows = rm[2 * r0[Lv]: 2 * (r0[Lv] + rn[Lv])].reshape(-1, 2)
            want = {}
            for t in gen:
                q, chain = int(tf[t]), []
                while q >= 0:
                    chain.append(q)
                    q = int(recs[q]["next"])
                for c, q in enumerate(chain):
                    want[q] = (c, len(chain), chain)
            seen = set()
            for i, (rec, meta) in enumerate(rows):
                if rec < 0:
                    continue
                rec, pos, k = int(rec), int(meta) & 255, int(meta) >> 8
                assert rec not in seen and want[rec][:2] == (pos, k)
                seen.add(rec)
                first = i - pos
                assert first // 4 == (first + k - 1) // 4, "a task inside one wavefront"
                assert [int(x) for x in rows[first:first + k, 0]] == want[rec][2], "the rows of a task: its messages in order"
            assert seen == set(want)
            assert (rows[:, 0] < 0).sum() < 4 * max(1, (rn[Lv] // 4)) and rn[Lv] <= 4 * len(gen)
    assert levels_with_rows > 0
    lib.pgbp_plan_destroy(pl)
