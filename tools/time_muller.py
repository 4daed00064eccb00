#!/usr/bin/env python3
"""Timing of the large-belief kernels on the reference's documented clique tree of the Mueller et al. network
(docs/src/man/clustergraphs.md:40-89; tests/golden/muller_2022.phy: 664 cliques, the largest of 54 nodes) for p traits:
p = 2 -> beliefs of up to 108 variables (bp_level_big, working matrix in LDS), p = 3, 4 -> 162, 216 (its workspace variant).
  python tools/time_muller.py [p ...]          one JSON line per p: ms per calibrate!(), messages, largest belief
  rocprofv3 --kernel-trace --stats -- python3 tools/time_muller.py 4     per-kernel launch averages (bp_level_big<...>)"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgbp_amd as P  # noqa: E402


def main():
    ps = [int(x) for x in sys.argv[1:]] or [2, 3, 4, 6]
    net, names = P.read_newick(open(os.path.join(ROOT, "tests", "golden", "muller_2022.phy")).read())
    cn, ed, sn = P.cliquetree(net.node2family)
    lib = P.load()
    for p in ps:
        st = P.allocate_scopes(cn, ed, sn, net, p)
        rng = np.random.default_rng(2)
        rates = np.stack([np.eye(p) + 0.3])
        X = P.simulate_bm_network(net, rates, np.zeros(p), rng)
        pe = [list(zip(net.length[i], net.gamma[i], net.color[i])) for i in range(net.nnodes)]
        fam = P.lg_families(st.clusters, st.node2cluster, net.node2family, st.node2fixed, pe, list(range(net.nnodes)), p, n_rates=1)
        cgb = P.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None)
        cgb.lg_setup(fam, X)
        cgb.assignfactors_lg_(rates, np.zeros(p))
        root = P.default_rootcluster(cn, net.is_leaf)
        spt = P.spanningtree_clusterlist(len(cn), ed, root)
        cgb.set_schedule([spt])
        opts = cgb._opts()
        ms = C.c_float()
        assert lib.pgbp_time_enqueued(cgb._eng, 0, 3, 0, C.byref(opts), C.byref(ms)) == 0      # warm-up
        reps = 20
        assert lib.pgbp_time_enqueued(cgb._eng, 0, reps, 0, C.byref(opts), C.byref(ms)) == 0
        # round 4: the same with residual_kldiv! after every message (sepsets above 96 variables on the workspace instance of
        # the kernel), and free_energy (beliefs above 139 variables on its workspace instance)
        import time
        from pgbp_amd import _lib as L
        res = (L.Result * 1)()

        def wall(o, n):   # pgbp_calibrate (the entry point that honours update_residualkldiv), host wall clock
            assert lib.pgbp_calibrate(cgb._eng, 1, C.byref(o), res) == 0 and res[0].succ
            t0 = time.perf_counter()
            assert lib.pgbp_calibrate(cgb._eng, n, C.byref(o), res) == 0 and res[0].succ
            return (time.perf_counter() - t0) / n * 1e3
        ms_plain = wall(cgb._opts(), 3)
        ms_kl = wall(cgb._opts(update_residualkldiv=True), 3)
        # integratebelief! of the LARGEST calibrated belief (src/beliefupdates.jl:187-200), host wall clock incl. the result's way back
        biggest = int(np.argmax(st.dims[:len(cn)]))
        cgb.integratebelief_(biggest)
        t0 = time.perf_counter()
        ib = cgb.integratebelief_(biggest)
        ms_ib = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        fe = cgb.free_energy()
        ms_fe = (time.perf_counter() - t0) * 1e3
        ll, info = cgb.loglik_lg()
        big = sorted(int(d) for d in st.dims[:len(cn)])[-3:]
        print(json.dumps({"workload": f"Mueller et al. clique tree, {p} traits", "cliques": len(cn), "messages_per_calibrate": 2 * len(ed),
                          "largest_beliefs": big, "beliefs_above_64": int((st.dims[:len(cn)] > 64).sum()),
                          "beliefs_above_128": int((st.dims[:len(cn)] > 128).sum()),
                          "largest_sepset": int(st.dims[len(cn):].max()), "ms_per_calibrate": ms.value / reps,
                          "ms_per_pgbp_calibrate_wall": ms_plain, "ms_per_pgbp_calibrate_wall_with_residual_kldiv": ms_kl,
                          "free_energy_ms": ms_fe, "minus_free_energy": -fe[2],
                          "integratebelief_largest_ms": ms_ib, "integratebelief_largest_norm": float(np.ravel(ib[1])[0]),
                          "loglik": float(ll[0]), "info": int(info[0])}), flush=True)


if __name__ == "__main__":
    main()
