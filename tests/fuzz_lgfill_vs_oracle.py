#!/usr/bin/env python3
"""Differential fuzz of the device factor fill (pgbp_lg_setup / pgbp_lg_assignfactors) against the oracle's
assignfactors! restatement: random networks (level-1 blobs or the level-3 blob), cluster graphs (clique tree, Bethe,
join graph), models (BM fixed / random / improper root, heterogeneous BM, OU), trait counts 1..6, with and without
missing tip values (patterns the reference itself can process: tests/test_lg_families_cpu.py:missing_pattern).

  python tests/fuzz_lgfill_vs_oracle.py [n_cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pgbp_amd as P  # noqa: E402
from helpers import lg_inputs_from_oracle, oracle_setup, product_beliefs_from_oracle  # noqa: E402
from oracle import calibration as OC  # noqa: E402
from oracle import clustergraph as OCG  # noqa: E402
from oracle import network as ON  # noqa: E402
from test_gpu_lgfill import _models  # noqa: E402
from test_lg_families_cpu import missing_pattern  # noqa: E402


def run(n_cases, seed):
    rng = np.random.default_rng(seed)
    worst = 0.0
    n_missing = 0
    for case in range(n_cases):
        ntips = int(rng.integers(4, 40))
        net = (ON.random_level3_network(ntips, int(rng.integers(1, 4)), rng) if rng.random() < 0.4 else
               ON.random_network(ntips, int(rng.integers(0, max(1, ntips // 4))), rng))
        which = ["bm_fixed", "bm_random_root", "bm_improper_root", "hetero", "hetero_random_root", "ou_fixed",
                 "ou_random_root"][int(rng.integers(7))]
        p = 1 if which.startswith("ou") else int(rng.integers(1, 7))
        model = _models(p, rng, net, which)
        if rng.random() < 0.5 and not which.startswith("ou"):
            tbl, taxa = missing_pattern(net, p, rng, which)
            n_missing += 1
        else:
            taxa = net.tip_names
            tbl = [[float(x) for x in rng.normal(size=len(taxa))] for _ in range(p)]
        graph = ["cliquetree", "bethe", "joingraph"][int(rng.integers(3))]
        cg = OCG.cliquetree(net) if graph == "cliquetree" else OCG.bethe(net) if graph == "bethe" else OCG.joingraph(net, 3)
        if max(len(nodes) for _, nodes in cg.clusters) * p > 64:
            continue
        ocgb = oracle_setup(net, cg, model, tbl, taxa)
        pb = product_beliefs_from_oracle(ocgb.belief)
        for b in pb:
            b.J[...] = 0.0; b.h[...] = 0.0; b.g[...] = 0.0
        pcgb = P.ClusterGraphBelief(pb, ocgb.node2cluster, ocgb.node2family, ocgb.node2fixed, ocgb.cluster2nodes)
        fam, data, kw = lg_inputs_from_oracle(P, net, ocgb, model, tbl, taxa)
        pcgb.lg_setup(fam, data)
        pcgb.assignfactors_lg_(sync=True, **kw)
        for i in range(ocgb.nclusters):
            ob, qb = ocgb.belief[i], pcgb.belief[i]
            for x, y in ((qb.J, ob.J), (qb.h, ob.h), (qb.g, ob.g)):
                x, y = np.asarray(x), np.asarray(y)
                if x.size:
                    err = float(np.max(np.abs(x - y))) / max(1.0, float(np.max(np.abs(y))))
                    worst = max(worst, err)
                    assert err <= 1e-10, (case, graph, which, p, i, x, y)
        if graph != "bethe":   # exact graphs: the likelihood too
            spt = OCG.spanningtree_clusterlist(cg, OCG.default_rootcluster(cg, net))
            pcgb.set_schedule([spt])
            ll, info = pcgb.loglik_lg()
            assert not info.any()
            assert OC.propagate_1traversal_postorder(ocgb, *spt)
            oll = ocgb.integratebelief(spt[2][0])[1]
            assert abs(ll[0] - oll) <= 1e-8 * max(1.0, abs(oll)), (case, graph, which, ll, oll)
        del pcgb
    return n_missing, worst


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    n_missing, worst = run(n_cases, seed)
    print(f"{n_cases} cases ok ({n_missing} with missing tip values), worst relative factor error {worst:.2e}")


if __name__ == "__main__":
    main()
