#!/usr/bin/env python3
"""Measurement (test-side script): log-likelihood evaluations per second when B parameter sets (R, mu) are evaluated
in ONE pass -- the site dimension of the engine carries the B candidate models over the same data (what an optimiser
with finite-difference gradients, or a multi-start / grid search, issues).  cfg3 tree and data.

  python tests/run_param_batch.py [B ...]
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgbp_amd as P  # noqa: E402
from pgbp_amd import _lib as L  # noqa: E402
from pgbp_amd import synth as S  # noqa: E402


def main():
    batches = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]
    ntips, p = 50000, 16
    rng = np.random.default_rng(3)
    tr = S.random_tree(ntips, rng)
    R = S.random_rate_matrix(p, rng)
    mu = np.zeros(p)
    X = S.simulate_bm(tr, R, mu, rng)
    prob = S.cliquetree_of_tree(tr, p)
    lib = P.load()
    out = []
    for B in batches:
        cgb = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx,
                                               np.zeros((B, int(prob.packed_off[-1]))), n_sites=B)
        cgb.set_schedule(prob.schedule)
        kind, length, row = S.bm_tree_table(tr, prob)
        cgb.bm_tree_setup(kind, length, row, np.broadcast_to(X, (B,) + X.shape).copy())
        Rs = np.stack([R * (1.0 + 0.05 * b) for b in range(B)])
        cgb.assignfactors_bm_(Rs, np.broadcast_to(mu, (B, p)).copy())
        opts = cgb._opts()
        ms = C.c_float()
        reps = 20
        assert lib.pgbp_time_enqueued(cgb._eng, 2, 3, 1, C.byref(opts), C.byref(ms)) == 0
        assert lib.pgbp_time_enqueued(cgb._eng, 2, reps, 1, C.byref(opts), C.byref(ms)) == 0
        norm = np.zeros(B)
        info = np.zeros(B, np.int32)
        assert lib.pgbp_fetch_loglik(cgb._eng, L.f64p(norm), L.i32p(info)) == 0
        ref = [S.bm_loglik_pruning(tr, Rs[b], mu, X) for b in range(min(B, 2))]
        err = max(abs(norm[b] - ref[b]) / abs(ref[b]) for b in range(len(ref)))
        out.append({"parameter_sets": B, "ms_per_pass": ms.value / reps, "ll_evals_per_s": B * reps / (ms.value / 1e3),
                    "max_rel_err_vs_pruning": err})
        del cgb
    print(json.dumps(out))


if __name__ == "__main__":
    main()
