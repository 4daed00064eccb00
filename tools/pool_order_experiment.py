#!/usr/bin/env python3
"""Experiment: does the ORDER of the belief records in the pool matter on cfg3?  The engine lays records out in the
caller's belief order (clusters in the preorder of the tree); a level of the schedule then reads records scattered over the
whole 750 MB pool.  Here the same problem is relabelled so that clusters (and sepsets) are numbered by their postorder level
(height), i.e. every level launch reads and writes contiguous runs; both variants are timed.
usage: python tools/pool_order_experiment.py [ntips] [traits]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgbp_amd as P  # noqa: E402
from pgbp_amd import synth as S  # noqa: E402


def relabel(prob, packed, cperm, sperm):
    """cperm[new] = old cluster, sperm[new] = old sepset"""
    nc, ns = prob.nclusters, len(prob.sepset_clusters)
    cinv = np.empty(nc, np.int64); cinv[cperm] = np.arange(nc)
    dims = np.concatenate([prob.dims[:nc][cperm], prob.dims[nc:][sperm]]).astype(np.int32)
    sc = cinv[prob.sepset_clusters[sperm]].astype(np.int32)
    lens = np.diff(prob.scope_off)                       # per (sepset, side)
    order = (2 * sperm[:, None] + np.arange(2)[None, :]).reshape(-1)
    new_lens = lens[order]
    scope_off = np.concatenate([[0], np.cumsum(new_lens)]).astype(np.int64)
    scope_idx = np.concatenate([prob.scope_idx[prob.scope_off[o]:prob.scope_off[o + 1]] for o in order]).astype(np.int32) \
        if new_lens.sum() else np.zeros(0, np.int32)
    pa, ch = prob.schedule[0]
    sched = [(cinv[pa].astype(np.int32), cinv[ch].astype(np.int32))]
    off = prob.packed_off
    border = np.concatenate([cperm, nc + sperm])
    new_off = S._packed_offsets(dims)
    out = np.zeros_like(packed)
    for i, b in enumerate(border.tolist()):
        out[new_off[i]:new_off[i + 1]] = packed[off[b]:off[b + 1]]
    q = S.Problem(dims=dims, sepset_clusters=sc, scope_off=scope_off, scope_idx=scope_idx, schedule=sched,
                  nclusters=nc, root_cluster=int(cinv[prob.root_cluster]))
    q.packed_off = new_off
    return q, out


def timed(prob, packed, label, reps=200):
    cgb = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packed)
    lib = P.load()
    cgb.set_schedule(prob.schedule)
    o = cgb._opts()
    assert lib.pgbp_enqueue_calibrate(cgb._eng, 20, 0, C.byref(o)) == 0 and lib.pgbp_sync(cgb._eng) == 0
    t = time.perf_counter()
    assert lib.pgbp_enqueue_calibrate(cgb._eng, reps, 0, C.byref(o)) == 0 and lib.pgbp_sync(cgb._eng) == 0
    dt = (time.perf_counter() - t) / reps
    ll = cgb.integratebelief_(prob.root_cluster)[1]
    print(f"{label}: {dt * 1e3:.4f} ms per calibrate, loglik {ll:.6f}", flush=True)
    return ll


def main():
    ntips = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    p = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    rng = np.random.default_rng(3)
    tr = S.random_tree(ntips, rng)
    R = S.random_rate_matrix(p, rng)
    X = S.simulate_bm(tr, R, np.zeros(p), rng)
    prob = S.cliquetree_of_tree(tr, p)
    packed = S.bm_factors_cliquetree(tr, prob, R, np.zeros(p), X)
    ll0 = timed(prob, packed, "caller's order (preorder of the tree)")
    # height of every cluster in the schedule tree
    pa, ch = prob.schedule[0]
    nc = prob.nclusters
    h = np.zeros(nc, np.int64)
    for a, c in zip(pa[::-1].tolist(), ch[::-1].tolist()):
        h[a] = max(h[a], h[c] + 1)
    cperm = np.lexsort((np.arange(nc), h))
    sep_child = np.maximum(prob.sepset_clusters[:, 0], prob.sepset_clusters[:, 1])   # (the child has the larger preorder index)
    sperm = np.lexsort((np.arange(len(sep_child)), h[sep_child]))
    q, qp = relabel(prob, packed, cperm, sperm)
    ll1 = timed(q, qp, "numbered by postorder level")
    assert abs(ll0 - ll1) <= 1e-9 * abs(ll0)
    rp = rng.permutation(nc)
    q2, qp2 = relabel(prob, packed, rp, rng.permutation(len(sep_child)))
    ll2 = timed(q2, qp2, "random order")
    assert abs(ll0 - ll2) <= 1e-9 * abs(ll0)


if __name__ == "__main__":
    main()
