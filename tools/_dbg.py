import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, pgbp_amd as P
from pgbp_amd import synth as S
from helpers import oracle_cgb_from_problem, oracle_schedule
from oracle import calibration as OC
for (n, p) in [(3, 8), (6, 8), (6, 4), (6, 16)]:
    rng = np.random.default_rng(n * 100 + p)
    tr = S.random_tree(n, rng); R = S.random_rate_matrix(p, rng); R = (R + R.T) / 2; mu = rng.standard_normal(p)
    X = S.simulate_bm(tr, R, mu, rng); prob = S.cliquetree_of_tree(tr, p)
    packed = S.bm_factors_cliquetree(tr, prob, R, mu, X)
    cgb = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packed)
    ocgb = oracle_cgb_from_problem(prob, packed, p); spt = oracle_schedule(prob)
    assert P.propagate_1traversal_postorder_(cgb, *spt); OC.propagate_1traversal_postorder(ocgb, *spt)
    worst = []
    for i, ob in enumerate(ocgb.belief):
        pb = cgb.belief[i]
        for nm, x, y in (("J", pb.J, ob.J), ("h", pb.h, ob.h), ("g", pb.g, ob.g)):
            if np.size(y):
                e = np.max(np.abs(np.asarray(x) - np.asarray(y)))
                if e > 1e-8 * max(1, np.max(np.abs(y))): worst.append((i, nm, int(prob.dims[i]), float(e)))
    print("n", n, "p", p, "dims", prob.dims.tolist(), "sched", prob.schedule[0][0].tolist(), prob.schedule[0][1].tolist(), "BAD" if worst else "ok", worst[:6])
