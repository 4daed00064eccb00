#!/usr/bin/env python3
"""Differential fuzzing (manual run, not collected by pytest): random trees (bifurcating / polytomies / caterpillars),
random trait counts 1..16, clique tree or Bethe graph, 1..3 sites, 1..3 calibration iterations; the device engine
against the plain-C sequential engine of the oracle: beliefs to 1e-8 * max|.|, residual flags, (succ, iscal).

  python tests/fuzz_gpu_vs_c_oracle.py [n_cases] [seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgbp_amd as P  # noqa: E402
from oracle import cengine  # noqa: E402
from pgbp_amd import synth as S  # noqa: E402


def run(n_cases, seed):
    """Returns (cases with an injected failure, worst relative belief error, sha256 over every bit of the calibrated states
    and flags: equal across launch modes that run the same arithmetic in the same order); asserts on any disagreement."""
    import hashlib
    digest = hashlib.sha256()
    rng = np.random.default_rng(seed)
    worst = 0.0
    n_fail = 0
    for case in range(n_cases):
        p = int(rng.integers(1, 17))
        ntips = int(rng.integers(2, 120))
        kind = rng.choice(["random", "poly", "caterpillar"])
        if kind == "random":
            tr = S.random_tree(ntips, rng)
        elif kind == "poly":
            tr = S.random_multifurcating_tree(max(ntips, 3), int(rng.integers(3, 8)), rng)
        else:
            tr = S.caterpillar_tree(ntips, rng)
        graph = rng.choice(["cliquetree", "bethe"])
        ns = int(rng.integers(1, 4))
        niter = int(rng.integers(1, 4))
        R = S.random_rate_matrix(p, rng)
        mu = rng.standard_normal(p)
        prob = S.cliquetree_of_tree(tr, p) if graph == "cliquetree" else S.bethe_of_tree(tr, p)
        packs = []
        for _ in range(ns):
            X = S.simulate_bm(tr, R, mu, rng)
            packs.append(S.bm_factors_cliquetree(tr, prob, R, mu, X) if graph == "cliquetree"
                         else S.bm_factors_bethe(tr, prob, R, mu, X))
        # one case in four: a non-positive-definite block somewhere in one site -> the first failure of the
        # reference's sequential order must be reported, the other sites must be unaffected
        bad_site = -1
        if rng.random() < 0.25:
            bad_site = int(rng.integers(ns))
            big = [i for i in range(prob.nclusters) if prob.dims[i] > 0]
            if not big:
                bad_site = -1
        if bad_site >= 0:
            c = int(rng.choice(big))
            k = int(rng.integers(prob.dims[c]))
            packs[bad_site] = packs[bad_site].copy()
            packs[bad_site][prob.packed_off[c] + k * (prob.dims[c] + 1)] = -abs(rng.normal()) * 1e3
        eng = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx,
                                               np.stack(packs), n_sites=ns)
        got = P.calibrate_(eng, prob.schedule, niter, verbose=False)
        pa, ch = prob.schedule[0]
        for s in range(ns):
            ref = cengine.Engine(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packs[s])
            want = ref.calibrate(pa, ch, niter, return_iscal=True)
            eng.site = s
            r = eng.last_results[s]
            assert (bool(r.succ), bool(r.iscal)) == want, (case, s, want, (r.succ, r.iscal))
            if not want[0]:
                assert s == bad_site
                assert (r.fail_edge, r.fail_dir, r.fail_info) == ref.last_failure(), (case, s, ref.last_failure())
                n_fail += 1
                continue   # the state after a failed calibration is not compared (later levels may have run)
            a, b = eng._packed[s], ref.packed()
            digest.update(np.ascontiguousarray(a).tobytes())
            digest.update(np.ascontiguousarray(eng._flags().astype(np.uint8)).tobytes())
            off = prob.packed_off
            for i in range(len(prob.dims)):
                x, y = a[off[i]:off[i + 1]], b[off[i]:off[i + 1]]
                if x.size:
                    err = float(np.max(np.abs(x - y))) / max(1.0, float(np.max(np.abs(y))))
                    worst = max(worst, err)
                    assert err <= 1e-8, (case, p, ntips, kind, graph, ns, niter, s, i, err)
            _, flags = ref.residuals()
            assert np.array_equal(eng._flags().astype(bool), flags.astype(bool)), (case, s)
        assert got == (bool(eng.last_results[0].succ), bool(eng.last_results[0].iscal))
        del eng
    return n_fail, worst, digest.hexdigest()


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    n_fail, worst, digest = run(n_cases, seed)
    print(f"{n_cases} cases ok ({n_fail} with an injected failure reported identically), worst relative belief error {worst:.2e}, "
          f"digest {digest}")


if __name__ == "__main__":
    main()
