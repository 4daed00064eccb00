#!/usr/bin/env python3
"""Child process of tests/test_plan_cpu.py::test_planner_under_address_and_undefined_sanitizers: drives the host-only
planning API (pgbp_plan_* of include/pgbp.h) of an ASan + UBSan build of csrc/pgbp_plan.cpp over trees, Bethe graphs,
network graphs (several schedule trees, node-subtree schedules, fused chains) and malformed inputs.  Any sanitizer
report aborts the process (halt_on_error)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pgbp_amd import _lib as L  # noqa: E402  (ctypes structures only: the product library is NOT loaded here)
from pgbp_amd import clustergraph as CG  # noqa: E402
from pgbp_amd import networks as NW  # noqa: E402
from pgbp_amd import synth as S  # noqa: E402

lib = C.CDLL(sys.argv[1])
for name, (res, args) in L.SYMBOLS.items():
    if name.startswith("pgbp_plan_"):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args


def plan(dims, sepcl, soff, sidx, n_sites=1):
    desc, keep = L.make_desc(dims, sepcl, soff, sidx, n_sites)
    pl = C.c_void_p()
    code = lib.pgbp_plan_create(C.byref(desc), C.byref(pl))
    return pl, code, keep


def schedule(pl, sched):
    trees = [(np.ascontiguousarray(t[-2], np.int32), np.ascontiguousarray(t[-1], np.int32)) for t in sched]
    off = np.zeros(len(trees) + 1, np.int32)
    off[1:] = np.cumsum([len(t[0]) for t in trees])
    pa = np.ascontiguousarray(np.concatenate([t[0] for t in trees] + [np.zeros(1, np.int32)]))
    ch = np.ascontiguousarray(np.concatenate([t[1] for t in trees] + [np.zeros(1, np.int32)]))
    return lib.pgbp_plan_set_schedule(pl, len(trees), L.i32p(off), L.i32p(pa), L.i32p(ch)), len(trees)


def walk(pl, ntrees):
    for t in range(ntrees):
        for d in (0, 1):
            nl, nt, ne = C.c_int32(), C.c_int32(), C.c_int32()
            assert lib.pgbp_plan_traversal_sizes(pl, t, d, C.byref(nl), C.byref(nt), C.byref(ne)) == 0
            lo, to = np.zeros(nl.value + 1, np.int32), np.zeros(nt.value + 1, np.int32)
            em, ee, er = (np.zeros(max(1, ne.value), np.int32) for _ in range(3))
            assert lib.pgbp_plan_traversal(pl, t, d, L.i32p(lo), L.i32p(to), L.i32p(em), L.i32p(ee), L.i32p(er)) == 0
            nf = np.zeros(max(1, nl.value), np.int32)
            assert lib.pgbp_plan_level_nfast(pl, t, d, L.i32p(nf)) == 0


rng = np.random.default_rng(0)
count = 0
for ntips, p, kind in [(2, 1, "t"), (3, 16, "t"), (40, 3, "t"), (200, 16, "t"), (60, 8, "b"), (50, 1, "b"), (30, 5, "c"), (25, 16, "m")]:
    tr = {"t": S.random_tree, "b": S.random_tree, "c": S.caterpillar_tree}.get(kind, None)
    tr = S.random_multifurcating_tree(ntips, 6, rng) if kind == "m" else tr(ntips, rng)
    prob = S.bethe_of_tree(tr, p) if kind == "b" else S.cliquetree_of_tree(tr, p)
    for ns in (1, 9):
        pl, code, keep = plan(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, ns)
        assert code == 0
        rc, nt = schedule(pl, prob.schedule)
        assert rc == 0
        walk(pl, nt)
        lib.pgbp_plan_destroy(pl)
        count += 1
for seed in range(4):
    net = NW.random_level3_network(int(rng.integers(10, 120)), int(rng.integers(1, 8)), rng, n_colors=2)
    for graph in ("bethe", "join"):
        cn, ed, sn = CG.bethe(net.node2family) if graph == "bethe" else CG.joingraph(net.node2family, 3)
        for p in (1, 4):
            st = NW.allocate_scopes(cn, ed, sn, net, p)
            pl, code, keep = plan(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx)
            assert code == 0
            for sched in (CG.spanningtrees_clusterlist(len(cn), ed, cn, net.is_leaf),
                          [x for x in (CG.nodesubtree_clusterlist(cn, ed, sn, v) for v in range(1, net.nnodes + 1)) if x[2]]):
                rc, nt = schedule(pl, sched)
                assert rc == 0, lib.pgbp_plan_last_error(pl)
                walk(pl, nt)
                count += 1
            # malformed schedules are refused, not read out of bounds
            good = CG.spanningtrees_clusterlist(len(cn), ed, cn, net.is_leaf)[0]
            bad = (np.array(good[2][::-1], np.int32), np.array(good[3][::-1], np.int32))
            assert schedule(pl, [bad])[0] != 0
            bad = (np.array(good[2], np.int32) + 10 ** 6, np.array(good[3], np.int32))
            assert schedule(pl, [bad])[0] != 0
            lib.pgbp_plan_destroy(pl)
# malformed descriptions
tr = S.random_tree(6, rng)
p2 = S.cliquetree_of_tree(tr, 2)
p2.scope_idx[:2] = p2.scope_idx[:2][::-1]
pl, code, keep = plan(p2.dims, p2.sepset_clusters, p2.scope_off, p2.scope_idx)
assert code != 0
lib.pgbp_plan_destroy(pl)
p3 = S.cliquetree_of_tree(tr, 200)   # dimension 400 > PGBP_MAX_DIM = 384
pl, code, keep = plan(p3.dims, p3.sepset_clusters, p3.scope_off, p3.scope_idx)
assert code != 0
lib.pgbp_plan_destroy(pl)
print(f"sanitized planner ok: {count} schedules")
