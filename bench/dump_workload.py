#!/usr/bin/env python3
"""Writes a clique-tree workload of this repo's generator in the flat binary form bench/reference_calibrate.jl reads
(the reference's own set-up is O(n^2) in the number of nodes and cannot build a 50 000-tip clique tree)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgbp_amd import synth as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--ntips", type=int, default=50000)
    ap.add_argument("--traits", type=int, default=16)
    ap.add_argument("--seed", type=int, default=3)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    tr = S.random_tree(a.ntips, rng)
    R = S.random_rate_matrix(a.traits, rng)
    mu = np.zeros(a.traits)
    X = S.simulate_bm(tr, R, mu, rng)
    prob = S.cliquetree_of_tree(tr, a.traits)
    packed = S.bm_factors_cliquetree(tr, prob, R, mu, X)
    nc = prob.nclusters
    ns = len(prob.dims) - nc
    pa, ch = prob.schedule[0]
    cn = np.zeros((nc, 2), np.int32)
    for i, nodes in enumerate(prob.cluster_nodes):
        cn[i, :len(nodes)] = np.asarray(nodes, np.int32)[:2]
    with open(a.out, "wb") as f:
        np.array([nc, ns, a.traits, len(pa), packed.size], np.int64).tofile(f)
        np.asarray(prob.dims, np.int32).tofile(f)
        cn.tofile(f)
        np.asarray(prob.meta["child_dim"], np.int32).tofile(f)
        np.asarray(prob.sepset_nodes, np.int32).tofile(f)
        np.asarray(prob.sepset_clusters, np.int32).reshape(-1).tofile(f)
        np.asarray(pa, np.int32).tofile(f)
        np.asarray(ch, np.int32).tofile(f)
        np.asarray(packed, np.float64).tofile(f)
    print(f"{a.out}: {nc} clusters, {ns} sepsets, {packed.size} doubles", file=sys.stderr)


if __name__ == "__main__":
    main()
