#!/usr/bin/env python3
"""Per-launch durations and kernel-to-kernel gaps of one calibrate! on the cfg3 workload, from a rocprofv3 kernel trace.

  step 1 (under the profiler; the program itself after `--`):
     rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lt -- python3 tools/level_times.py run [ntips] [traits]
     (or `run-network [joingraph|bethe] [ntips]`: the cfg5 network workload of bench.py;
      `run-ll [ntips] [traits]`: cfg3's log-likelihood evaluation, for the PMC passes of `ll_eval.roofline.traffic`;
      `run-bethe [ntips] [traits]`: cfg2, the Bethe cluster graph of a tree)
  step 2 (plain): python3 tools/level_times.py parse gpurun_out/lt [out.json]

`run` builds the workload, does 3 warm-up calibrates and 5 more, each bracketed by a device sync so that the
calibrates are separable in the trace by their idle gaps.  `parse` reads the *_kernel_trace.csv, keeps the message-kernel
dispatches of the LAST calibrate and prints, per launch: grid size (workgroups), duration, gap to the previous launch."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(ntips, p, bethe=False):
    import numpy as np
    import pgbp_amd as P
    from pgbp_amd import synth as S
    rng = np.random.default_rng(2 if bethe else 3)
    tr = S.random_tree(ntips, rng)
    R = S.random_rate_matrix(p, rng)
    X = S.simulate_bm(tr, R, np.zeros(p), rng)
    if bethe:   # cfg2: Bethe cluster graph of the tree (a tree itself: one spanning tree, exact in one iteration)
        prob = S.bethe_of_tree(tr, p)
        packed = S.bm_factors_bethe(tr, prob, R, np.zeros(p), X)
    else:
        prob = S.cliquetree_of_tree(tr, p)
        packed = S.bm_factors_cliquetree(tr, prob, R, np.zeros(p), X)
    cgb = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packed)
    import time
    for _ in range(8):
        assert P.calibrate_(cgb, prob.schedule, 1, sync=False)[0]
        time.sleep(0.003)
    print("loglik", cgb.integratebelief_(prob.root_cluster)[1])


def run_ll(ntips, p):
    """the log-likelihood evaluation of cfg3 (bench.py's `ll_eval`): assignfactors! on the device, postorder, root
    integratebelief! -- 3 warm-up evaluations and 8 more, each behind a device sync, and nothing else behind the set-up"""
    import ctypes as C
    import time
    import numpy as np
    import pgbp_amd as P
    from pgbp_amd import synth as S
    rng = np.random.default_rng(3)
    tr = S.random_tree(ntips, rng)
    R = S.random_rate_matrix(p, rng)
    X = S.simulate_bm(tr, R, np.zeros(p), rng)
    prob = S.cliquetree_of_tree(tr, p)
    packed = S.bm_factors_cliquetree(tr, prob, R, np.zeros(p), X)
    cgb = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packed)
    cgb.set_schedule(prob.schedule)
    cgb.bm_tree_setup(*S.bm_tree_table(tr, prob), X)
    cgb.assignfactors_bm_(R, np.zeros(p))
    lib = P.load()
    opts = cgb._opts()
    for _ in range(11):
        assert lib.pgbp_enqueue_loglik_bm(cgb._eng, 1, C.byref(opts)) == 0
        assert lib.pgbp_sync(cgb._eng) == 0
        time.sleep(0.003)
    norm = np.zeros(1)
    info = np.zeros(1, dtype=np.int32)
    from pgbp_amd import _lib as L
    assert lib.pgbp_fetch_loglik(cgb._eng, L.f64p(norm), L.i32p(info)) == 0
    print("loglik", norm[0], "pruning", S.bm_loglik_pruning(tr, R, np.zeros(p), X))


def run_network(graph, ntips):
    """the cfg5 network workload (bench.py --workload network): 8 calibrate iterations from the regularised start"""
    import argparse
    import ctypes as C
    import time
    import pgbp_amd as P
    import bench as B
    args = argparse.Namespace(seed=5, traits=4, blob_style="varied", ntips=ntips, blobs=(ntips + 11) // 12, graph=graph, maxclustersize=3)   # bench.py's defaults
    net, (cn, ed, sn), st, fam, X, rates, mu, sched = B.build_network_workload(args, 0)
    cgb = P.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None)
    cgb.lg_setup(fam, X)
    cgb.assignfactors_lg_(rates, mu)
    lib = P.load()
    if graph == "joingraph":
        from pgbp_amd.regularization import regularizebeliefs_onschedule_
        regularizebeliefs_onschedule_(cgb)
    else:
        assert lib.pgbp_regularize_bycluster(cgb._eng) == 0
    cgb.set_schedule(sched)
    opts = cgb._opts()
    for _ in range(8):
        assert lib.pgbp_enqueue_calibrate(cgb._eng, 1, 0, C.byref(opts)) == 0
        assert lib.pgbp_sync(cgb._eng) == 0
        time.sleep(0.003)
    print("trees", len(sched), "messages per iteration", sum(2 * len(s[2]) for s in sched))


def parse(d, out=None):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    assert files, f"no kernel trace under {d}"
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                             int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)))
    rows.sort()
    msg = [r for r in rows if "pgbp::bp_" in r[2]]
    # the run sleeps 3 ms between its calibrates: the last one = the dispatches after the last idle gap of over 1 ms
    cut = 0
    for i in range(1, len(msg)):
        if msg[i][0] - msg[i - 1][1] > 1_000_000:
            cut = i
    groups = [msg]
    g = msg[cut:]
    t_first, t_last = g[0][0], g[-1][1]
    res = []
    prev_end = None
    for (s, e, name, grid, wg) in g:
        short = name.split("(")[0].replace("void pgbp::", "")
        res.append({"kernel": short, "workgroups": grid // max(1, wg), "wg_size": wg, "us": (e - s) / 1e3,
                    "gap_us": None if prev_end is None else (s - prev_end) / 1e3})
        prev_end = e
    tot = sum(r["us"] for r in res)
    gaps = sum(r["gap_us"] or 0 for r in res)
    summary = {"launches": len(res), "span_us": (t_last - t_first) / 1e3, "sum_kernel_us": tot, "sum_gap_us": gaps,
               "calibrates_seen": len(groups)}
    for i, r in enumerate(res):
        print(f"{i:3d} {r['kernel'][:44]:44s} wgs {r['workgroups']:6d} x{r['wg_size']:4d}  {r['us']:8.2f} us  gap {r['gap_us'] if r['gap_us'] is not None else 0:6.2f}")
    print(json.dumps(summary))
    if out:
        with open(out, "w") as fh:
            json.dump({"summary": summary, "launches": res}, fh, indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "run-network":
        run_network(sys.argv[2] if len(sys.argv) > 2 else "joingraph", int(sys.argv[3]) if len(sys.argv) > 3 else 20000)
    elif sys.argv[1] == "run-bethe":   # cfg2: 10 000 tips, 8 traits
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 10000, int(sys.argv[3]) if len(sys.argv) > 3 else 8, bethe=True)
    elif sys.argv[1] == "run-ll":      # cfg3's log-likelihood evaluation (fill + postorder + root integrate)
        run_ll(int(sys.argv[2]) if len(sys.argv) > 2 else 50000, int(sys.argv[3]) if len(sys.argv) > 3 else 16)
    elif sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 50000, int(sys.argv[3]) if len(sys.argv) > 3 else 16)
    else:
        parse(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
