#!/usr/bin/env python3
"""Differential fuzzing on NETWORKS (run by tests/test_gpu_parity.py in a child process): random level-<=3 networks,
1..9 traits (now and then 18..22: senders beyond the register-resident small-message body, some beyond 64 variables),
clique tree / Bethe / join graph (maxclustersize 3 or 4), every spanning tree of `spanningtrees_clusterlist` as the
schedule, heterogeneous BM factors filled on the device, loopy graphs regularised (Bethe: by cluster, join graphs: on the schedule), 1..3 iterations,
1..2 sites; the device engine (wave-per-task kernels: level launches, chunks of fused levels, small-message and in-LDS
bodies, large-belief kernel; register-resident kernel where a level is all fast-class) against the plain-C sequential
engine of the oracle: (succ, iscal) after every schedule tree's worth of messages, beliefs to 1e-8 * max|.|, residual
flags; one case in five with a damaged (negative definite) cluster: the same first failure.

  python tests/fuzz_gpu_vs_c_oracle_networks.py [n_cases] [seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgbp_amd as P  # noqa: E402
from oracle import cengine  # noqa: E402


def run(n_cases, seed):
    rng = np.random.default_rng(seed)
    worst, n_fail, n_loopy = 0.0, 0, 0
    lib = P.load()
    for case in range(n_cases):
        big = rng.random() < 0.12
        p = int(rng.integers(18, 23)) if big else int(rng.integers(1, 10))
        ntips = int(rng.integers(8, 40)) if big else int(rng.integers(8, 160))
        net = P.random_level3_network_varied(ntips, max(1, ntips // int(rng.integers(3, 9))), rng, n_colors=2)
        graph = str(rng.choice(["cliquetree", "bethe", "joingraph3", "joingraph4"]))
        if graph == "cliquetree":
            cn, ed, sn = P.cliquetree(net.node2family)
        elif graph == "bethe":
            cn, ed, sn = P.bethe(net.node2family)
        else:
            cn, ed, sn = P.joingraph(net.node2family, int(graph[-1]))
        st = P.allocate_scopes(cn, ed, sn, net, p)
        if int(np.max(st.dims)) > 128:
            continue
        base = P.synth.random_rate_matrix(p, rng)
        base = (base + base.T) / 2
        rates = np.stack([base, 1.7 * base])
        mu = rng.standard_normal(p)
        X = P.simulate_bm_network(net, rates, mu, rng)
        pe = [list(zip(net.length[i], net.gamma[i], net.color[i])) for i in range(net.nnodes)]
        fam = P.lg_families(st.clusters, st.node2cluster, net.node2family, st.node2fixed, pe, list(range(net.nnodes)), p, n_rates=2)
        sched = P.spanningtrees_clusterlist(len(cn), ed, cn, net.is_leaf)
        loopy = len(ed) > len(cn) - 1
        n_loopy += loopy
        niter = int(rng.integers(1, 4))
        cgb = P.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None)
        cgb.lg_setup(fam, X)
        cgb.assignfactors_lg_(rates, mu)
        if loopy and graph.startswith("joingraph"):
            # (as bench.py does for cfg5: with the by-cluster regulariser the first sweeps of a join graph pass through
            # nearly singular J_I, where the engines' different elimination orders are amplified to 1e-4)
            from pgbp_amd.regularization import regularizebeliefs_onschedule_
            regularizebeliefs_onschedule_(cgb)
        elif loopy:
            assert lib.pgbp_regularize_bycluster(cgb._eng) == 0
        cgb.pull()
        start = cgb._packed[0].copy()
        damaged = -1
        if rng.random() < 0.2:
            cand = [c for c in range(len(cn)) if st.dims[c] > 0]
            damaged = int(rng.choice(cand))
            m = int(st.dims[damaged])
            J = start[cgb._poff[damaged]: cgb._poff[damaged] + m * m].reshape(m, m)
            J[np.arange(m), np.arange(m)] = -abs(rng.normal()) * 1e4 - 1e3
            cgb._packed[0][:] = start
            cgb.push()
        ce = cengine.Engine(st.dims, np.asarray(st.sepset_clusters).reshape(-1), st.scope_off, st.scope_idx, start)
        cgb.set_schedule(sched)
        cgb.init_messagecalibrationflags_reset_()
        got = P.calibrate_(cgb, sched, niter, verbose=False)
        want, failed, where = (True, False), False, None
        for it in range(niter):
            for j, spt in enumerate(sched):
                succ, iscal = ce.calibrate(spt[2], spt[3], 1, return_iscal=True)
                want = (succ, iscal)
                if not succ:
                    failed, where = True, (it + 1, j + 1)
                    break
            if failed:
                break
        desc = (case, p, ntips, graph, len(sched), niter, damaged)
        r = cgb.last_results[0]
        if failed:
            assert got == (False, False), (desc, got)
            fe, fd, fi = ce.last_failure()
            assert (r.fail_edge, r.fail_dir, r.fail_info) == (fe, fd, fi), (desc, (r.fail_edge, r.fail_dir, r.fail_info), (fe, fd, fi))
            assert (r.fail_iter, r.fail_tree) == where, (desc, (r.fail_iter, r.fail_tree), where)
            n_fail += 1
            del cgb
            continue
        assert got == want, (desc, got, want)
        cgb.pull()
        a, b = cgb._packed[0], ce.packed()
        off = cgb._poff
        for i in range(len(st.dims)):
            x, y = a[off[i]:off[i + 1]], b[off[i]:off[i + 1]]
            if x.size:
                err = float(np.max(np.abs(x - y))) / max(1.0, float(np.max(np.abs(y))))
                worst = max(worst, err)
                assert err <= 1e-8, (desc, i, err)
        _, flags = ce.residuals()
        assert np.array_equal(cgb._flags().astype(bool), flags.astype(bool)), desc
        del cgb
        # now and then the same problem as site 0 of a TWO-site engine beside a rescaled copy (1.5 x the canonical
        # parameters: another valid start), each site against the C engine on its own start (blockIdx.y = site in every kernel)
        if case % 4 == 0:
            starts2 = np.stack([start, 1.5 * start])
            cgb2 = P.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, starts2, n_sites=2)
            cgb2.set_schedule(sched)
            cgb2.init_messagecalibrationflags_reset_()
            P.calibrate_(cgb2, sched, niter, verbose=False)
            for sidx in range(2):
                ce2 = cengine.Engine(st.dims, np.asarray(st.sepset_clusters).reshape(-1), st.scope_off, st.scope_idx, starts2[sidx])
                w2 = (True, False)
                for it in range(niter):
                    for spt in sched:
                        if w2[0]:
                            w2 = ce2.calibrate(spt[2], spt[3], 1, return_iscal=True)
                r2 = cgb2.last_results[sidx]
                assert (bool(r2.succ), bool(r2.iscal)) == tuple(bool(x) for x in w2), (desc, sidx)
                if not w2[0]:
                    continue
                a2, b2 = cgb2._packed[sidx], ce2.packed()
                err = float(np.max(np.abs(a2 - b2))) / max(1.0, float(np.max(np.abs(b2))))
                assert err <= 1e-8, (desc, sidx, err)
            del cgb2
    return n_fail, n_loopy, worst


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    n_fail, n_loopy, worst = run(n_cases, seed)
    print(f"{n_cases} cases ok ({n_loopy} loopy, {n_fail} with a damaged cluster reported identically), worst relative belief error {worst:.2e}")


if __name__ == "__main__":
    main()
