#!/usr/bin/env python3
"""Experiment: where a pass of the loop launches of the packed layout (pgbp_loop.hip: tail, chunks) spends its time.
Needs the instrumented build:
  make -C phylogaussianbeliefprop.jl_amd/csrc -B ../../build/obj/pgbp_loop.o ../../build/obj/pgbp_fast.o FASTFLAGS="-DPGBP_STAMP -DPGBP_ONLY_P16"
  make -C phylogaussianbeliefprop.jl_amd/csrc && cp phylogaussianbeliefprop.jl_amd/csrc/libpgbp.so build/libpgbp_stamp.so   (then rebuild the product)
  PGBP_LIB=build/libpgbp_stamp.so python tools/stamp_loop.py
Phases (shader clocks, medians over the passes of waves that eliminated), by launch (grid) and kind of pass (first of
its walk / late / early):
  0 top -> 1 operands in registers, chain patched (early passes: at once) -> 2 symmetrised, elimination starts -> 3 done
  -> 4 own stores of the pass before complete -> 5 marginal handed over, barrier 1 -> 6 next record decoded, its sender
  operands requested -> 7 divide, stores issued -> 8 barrier 2 -> 9 mult, stores issued, chain slot written"""
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
    lib.pgbp_debug_lstamps.argtypes = [C.c_void_p, C.c_uint, C.c_void_p]
    for it in range(3):
        P.calibrate_(cgb, prob.schedule, 1, sync=False)
        assert lib.pgbp_debug_lstamps(out.ctypes.data, cap, C.byref(n)) == 0
    k = min(n.value, cap)
    t = out[:k].astype(np.int64)
    tag = t[:, 15]
    for key in np.unique(tag):
        sel = t[tag == key]
        grid, kind = divmod(int(key), 4)
        d = (sel[:, 1:10] - sel[:, 0:9]) & 0xFFFFFFFF
        elim = (sel[:, 2] != 0) & (sel[:, 3] != 0)          # passes that eliminated
        if elim.sum() < 2:
            continue
        dd = d[elim]
        tot = ((sel[elim, 9] - sel[elim, 0]) & 0xFFFFFFFF)
        kinds = {0: "early", 1: "first", 2: "late", 3: "first"}
        print(f"grid {grid:5d} {kinds[kind]:5s} passes {len(dd):5d} | top -> chain written {np.median(tot):7.0f} clk | " +
              " ".join(f"{i}>{i+1}:{np.median(dd[:, i]):6.0f}" for i in range(9)))
    # whole passes of the tail (grid 1): from the top of one to the top of the next, wave 0
    tail = t[(tag // 4 == 1) & (t[:, 13] == 0)]
    if len(tail) > 4:
        tops = np.sort(tail[:, 0])
        dt = np.diff(tops)
        print("tail, wave 0: top-to-top of consecutive passes, median", int(np.median(dt[dt < 100000])))


if __name__ == "__main__":
    main()
