#!/usr/bin/env python3
"""Experiment: per-phase timestamps (s_memtime / s_memrealtime) of the single-task launches of one calibrate
on the cfg3 workload.  Needs the instrumented build:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DPGBP_TRACE -o build/exp/libpgbp_trace.so csrc/*.cpp csrc/*.hip
  PGBP_LIB=build/exp/libpgbp_trace.so python tools/trace_narrow_levels.py
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgbp_amd as P  # noqa: E402
from pgbp_amd import synth as S  # noqa: E402


def main():
    ntips = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    p = 16
    rng = np.random.default_rng(3)
    tr = S.random_tree(ntips, rng)
    R = S.random_rate_matrix(p, rng)
    X = S.simulate_bm(tr, R, np.zeros(p), rng)
    prob = S.cliquetree_of_tree(tr, p)
    packed = S.bm_factors_cliquetree(tr, prob, R, np.zeros(p), X)
    cgb = P.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packed)
    lib = P.load()
    lib.pgbp_debug_trace.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_int]
    cap = 1 << 19
    out = np.zeros((cap, 10), dtype=np.uint64)
    n = C.c_uint(0)
    full = len(sys.argv) > 2 and sys.argv[2] == "all"
    assert lib.pgbp_debug_trace_mode(1 if full else 0) == 0
    for it in range(3):
        P.calibrate_(cgb, prob.schedule, 1, sync=False)
        assert lib.pgbp_debug_trace(out.ctypes.data, cap, C.byref(n), 1) == 0
    k = min(n.value, cap)
    t = out[:k].astype(np.int64)
    if not full:
        t = t[np.argsort(t[:, 8])]
        print(f"{k} traced single-task launches; shader cycles from start to: record, data-arrived, elim-done, "
              f"handover-done, stores-issued, stores-acked; realtime ticks (100 MHz): kernel length, gap to next start")
        for i in range(k):
            gap = int(t[i + 1, 8] - t[i, 9]) if i + 1 < k else -1
            print(i, [int(x) for x in t[i, 1:7]], "len", int(t[i, 9] - t[i, 8]), "gap", gap)
        return
    np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "trace_all.npy"),
            out[:k])
    grid = (t[:, 7] >> 32)
    t0 = t[:, 8].min()
    print(f"{k} waves traced; per launch (by grid size, in start order):")
    print("tasks   waves  span_us  first_start_us  life_us(p10/p50/p90)  cycles to: record data elim handover issued acked (median)  start_spread_us(p50,p90,max)")
    launches = {}
    for g in np.unique(grid):
        m = grid == g
        launches.setdefault(int(t[m, 8].min()), []).append(int(g))
    for st in sorted(launches):
        for g in launches[st]:
            m = grid == g
            s0, e0 = t[m, 8], t[m, 9]
            life = (e0 - s0) / 100.0
            rel = (s0 - s0.min()) / 100.0
            med = [int(np.median(t[m, j])) for j in range(1, 7)]
            print(f"{g:6d} {m.sum():7d} {(e0.max() - s0.min()) / 100.0:8.1f} {(s0.min() - t0) / 100.0:10.1f}   "
                  f"{np.percentile(life, 10):5.1f}/{np.percentile(life, 50):5.1f}/{np.percentile(life, 90):5.1f}   {med}   "
                  f"{np.percentile(rel, 50):5.1f},{np.percentile(rel, 90):5.1f},{rel.max():5.1f}")


if __name__ == "__main__":
    main()
