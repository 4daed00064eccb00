#!/usr/bin/env python3
"""The reference's getting-started pipeline (docs/src/man/getting_started.md:30-292) on the MI355X engine, with the
product's own host side from the Newick string to the likelihood:

    network string -> clique tree -> belief scopes -> schedule -> factors on the device -> calibrate! -> log-likelihood

Network, trait values and the expected numbers are the doctest's (tests/golden/reference_goldens.json: doctest_lazaridis).
Needs a GPU:  python examples/getting_started.py [cliquetree|bethe|joingraph|ltrip]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgbp_amd as P  # noqa: E402


def main():
    method = sys.argv[1] if len(sys.argv) > 1 else "cliquetree"
    with open(os.path.join(ROOT, "tests", "golden", "reference_goldens.json")) as f:
        g = json.load(f)["doctest_lazaridis"]
    net, names = P.read_newick(g["net"])                         # readnewick + preprocessnet!
    build = {"cliquetree": P.cliquetree, "bethe": P.bethe, "ltrip": P.ltrip, "joingraph": lambda f: P.joingraph(f, 3)}[method]
    cn, ed, sn = build(net.node2family)                          # clustergraph!(net, method)
    print(f"{method}: {len(cn)} clusters, {len(ed)} sepsets, largest cluster {max(len(c) for c in cn)} nodes")
    st = P.allocate_scopes(cn, ed, sn, net, 1)                   # allocatebeliefs
    row = {t: r for r, t in enumerate(g["taxa"])}
    fam = P.lg_families(st.clusters, st.node2cluster, net.node2family, st.node2fixed,
                        [list(zip(net.length[i], net.gamma[i], net.color[i])) for i in range(net.nnodes)],
                        [row.get(names[i], -1) for i in range(net.nnodes)], 1)
    cgb = P.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None)
    cgb.lg_setup(fam, np.array(g["x"], float)[:, None])
    cgb.assignfactors_lg_(np.array([[[g["model"]["sigma2"]]]], float), [g["model"]["mu"]])   # UnivariateBrownianMotion(1, 0)
    exact = len(ed) == len(cn) - 1
    if not exact:
        P.load().pgbp_regularize_bycluster(cgb._eng)             # regularizebeliefs_bycluster!
    sched = P.spanningtrees_clusterlist(len(cn), ed, cn, net.is_leaf)
    succ, iscal = P.calibrate_(cgb, sched, 50, auto=True, info=True)
    r = cgb.last_results[0]
    print(f"calibrate!: succ {succ}, calibrated {iscal} (iteration {r.iter_reached}, schedule tree {r.tree_reached})")
    root = sched[0][2][0]
    mu, norm = cgb.integratebelief_(root)
    _, _, fe = cgb.factored_energy()
    print(f"integratebelief! at cluster {root}: {norm:.12f};  factored energy {fe:.12f};  doctest log-likelihood {g['ll']:.12f}"
          + ("" if exact else "  (loopy graph: approximations)"))
    if exact:
        assert abs(norm - g["ll"]) <= 1e-9 * abs(g["ll"])


if __name__ == "__main__":
    main()
