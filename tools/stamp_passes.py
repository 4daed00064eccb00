#!/usr/bin/env python3
"""Experiment: where a pass of bp_fast16's level and loop modes (PGBP_TUNING=loop=0: tail / fused chunks on one wavefront per record)
spends its time (the streaming mode of round 2 is gone; bp_loop16 has tools/stamp_loop.py).
Needs the instrumented build (clock stamps of the phases of every pass of the first 64 workgroups of each launch):
  make -C phylogaussianbeliefprop.jl_amd/csrc FASTFLAGS="-DPGBP_STAMP -DPGBP_ONLY_P16" -B ../../build/obj/pgbp_fast.o && make ... ; cp libpgbp.so build/libpgbp_stamp.so
  PGBP_LIB=build/libpgbp_stamp.so PGBP_TUNING=loop=0 python tools/stamp_passes.py
Phases (shader clocks, medians over waves that did an elimination):
  0 top -> 1 operands waited for (streaming: vmcnt(0)) -> 2 loads issued, elimination starts -> 3 elimination done
  -> 4 marginal handed over -> 5 barrier 1 passed -> 6 tiles waited for -> 7 divide done -> 8 barrier 2 passed
  -> 9 next record resident -> 10 mult + stores issued -> 11 stores acknowledged"""
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
    cap = 1 << 16
    out = np.zeros((cap, 16), dtype=np.uint32)
    n = C.c_uint(0)
    lib.pgbp_debug_stamps.argtypes = [C.c_void_p, C.c_uint, C.c_void_p]
    for it in range(3):
        P.calibrate_(cgb, prob.schedule, 1, sync=False)
        assert lib.pgbp_debug_stamps(out.ctypes.data, cap, C.byref(n)) == 0
    k = min(n.value, cap)
    t = out[:k].astype(np.int64)
    tag = t[:, 15]
    for key in np.unique(tag):
        sel = t[tag == key]
        mode, grid = divmod(int(key), 1000000)
        d = (sel[:, 1:12] - sel[:, 0:11]) & 0xFFFFFFFF
        elim = d[:, 2] > 200                     # passes that eliminated
        if elim.sum() < 4:
            continue
        dd = d[elim]
        tot = ((sel[elim, 11] - sel[elim, 0]) & 0xFFFFFFFF)
        print(f"mode {mode} grid {grid:5d} passes {len(dd):5d} | total {np.median(tot):7.0f} clk | " +
              " ".join(f"{i}>{i+1}:{np.median(dd[:, i]):6.0f}" for i in range(11)))


if __name__ == "__main__":
    main()
