#!/usr/bin/env python3
"""cfg5-shaped measurement (BASELINE.json configs[4]) -- a test-side script, not part of
bench.py: the network, its Bethe cluster graph and the heterogeneous-BM factors are produced by the oracle's
restatement of the reference's host code (test infrastructure), the loopy calibration runs on the device engine.

  python tests/run_cfg5_network.py [ntips] [nhybrids]

Prints one JSON line: iterations to convergence, messages per calibrate!, messages/s on the GPU, and the plain-C
sequential engine's rate on one host core for the same schedule."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pgbp_amd as P  # noqa: E402
from helpers import oracle_setup, product_beliefs_from_oracle  # noqa: E402
from oracle import cengine  # noqa: E402
from oracle import clustergraph as OCG  # noqa: E402
from oracle import models as OM  # noqa: E402
from oracle import network as ON  # noqa: E402


def main():
    ntips = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    nhyb = int(sys.argv[2]) if len(sys.argv) > 2 else ntips // 4
    rng = np.random.default_rng(5)
    t0 = time.time()
    net = ON.random_network(ntips, nhyb, rng)
    p = 4
    rates = [np.eye(p) * s + 0.3 * s for s in (0.5, 1.0, 2.0)]
    colors = {e.number: 1 + int(rng.integers(3)) for e in net.edges}
    model = OM.HeterogeneousBrownianMotion(rates, colors, np.zeros(p))
    taxa = net.tip_names
    tbl = [list(rng.normal(size=len(taxa))) for _ in range(p)]
    cg = OCG.bethe(net)
    sched = OCG.spanningtrees_clusterlist(cg, net)
    ocgb = oracle_setup(net, cg, model, tbl, taxa)
    pcgb = P.ClusterGraphBelief(product_beliefs_from_oracle(ocgb.belief), ocgb.node2cluster, ocgb.node2family,
                                ocgb.node2fixed, ocgb.cluster2nodes)
    t_setup = time.time() - t0
    P.regularizebeliefs_bycluster_(pcgb, cg)
    start = pcgb._packed[0].copy()            # the regularised state (clusters AND sepsets) every run starts from
    nmsg_tree = [2 * len(s[2]) for s in sched]
    lib = P.load()
    # device: calibrate!(beliefs, sched, 100; auto=true), timed end to end (host-driven auto stop included)
    pcgb.set_schedule(sched)
    best = None
    for rep in range(5):
        pcgb._packed[0][:] = start
        pcgb.push()
        pcgb.init_messagecalibrationflags_reset_()
        lib.pgbp_sync(pcgb._eng)
        t = time.perf_counter()
        res = P.calibrate_(pcgb, sched, 100, auto=True, sync=False)
        dt = time.perf_counter() - t
        assert res == (True, True)
        r = pcgb.last_results[0]
        n_pairs = (r.iter_reached - 1) * len(sched) + r.tree_reached
        nmsg = sum(nmsg_tree[q % len(sched)] for q in range(n_pairs))
        best = dt if best is None else min(best, dt)
    # CPU: the plain-C sequential engine, same schedule and stop rule, one core
    eng = cengine.Engine(pcgb._dims, pcgb._sepcl.reshape(-1), pcgb._scope_off, pcgb._scope_idx, start)
    t = time.perf_counter()
    reached = None
    for it in range(1, 101):
        for j, spt in enumerate(sched, start=1):
            succ, iscal = eng.calibrate(spt[2], spt[3], 1, return_iscal=True)
            if iscal:
                reached = (it, j)
                break
        if reached:
            break
    dt_cpu = time.perf_counter() - t
    print(json.dumps({"workload": f"cfg5-shaped: heterogeneous BM (3 rates, 4 traits), {ntips} tips, {nhyb} reticulations, Bethe",
                      "clusters": int(pcgb.nclusters), "sepsets": int(pcgb.nsepsets), "schedule_trees": len(sched),
                      "iterations_to_convergence": [int(r.iter_reached), int(r.tree_reached)], "cpu_reached": reached,
                      "messages": int(nmsg), "gpu_ms": 1e3 * best, "gpu_messages_per_s": nmsg / best,
                      "cpu_ms": 1e3 * dt_cpu, "cpu_messages_per_s": nmsg / dt_cpu, "host_setup_s": t_setup}))


if __name__ == "__main__":
    main()
