#!/usr/bin/env python3
"""Experiment: where a pass of the loop launches of the packed layout (pgbp_loop.hip: tail, chunks) spends its time.
Needs the instrumented build:
  make -C phylogaussianbeliefprop.jl_amd/csrc -B ../../build/obj/pgbp_loop.o ../../build/obj/pgbp_fast.o FASTFLAGS="-DPGBP_STAMP -DPGBP_ONLY_P16"
  make -C phylogaussianbeliefprop.jl_amd/csrc && cp phylogaussianbeliefprop.jl_amd/csrc/libpgbp.so build/libpgbp_stamp.so   (then rebuild the product)
  PGBP_LIB=build/libpgbp_stamp.so python tools/stamp_loop.py
Phases (shader clocks, medians over the passes of wavefronts whose record eliminated), by launch (grid), kind of pass
(first of its walk / late / early) and role:
  provider (waves 0-7):  0 top -> 1 operands unpacked, chain patched -> 2 symmetrised, elimination starts -> 3 done
                         -> 4 marginal in LDS -> 5 barrier 1 passed -> 6 next record decoded, its sender operands requested
                         -> 7 barrier 2 passed
  consumer (waves 8-15): 0 top -> 4 its loads (and the stores of the pass before) complete -> 5 barrier 1 passed
                         -> 6 divide!, mult! (the whole task's, in its first consumer) -> 7 chain slot written
                         -> 8 barrier 2 passed -> 9 stores issued, next record decoded"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgbp_amd as P  # noqa: E402
from pgbp_amd import synth as S  # noqa: E402


def main():
    bethe = len(sys.argv) > 1 and sys.argv[1] == "bethe"   # cfg2 (10 000 tips, 8 traits) instead of cfg3
    args = sys.argv[2:] if bethe else sys.argv[1:]
    ntips = int(args[0]) if len(args) > 0 else (10000 if bethe else 50000)
    p = int(args[1]) if len(args) > 1 else (8 if bethe else 16)
    rng = np.random.default_rng(2 if bethe else 3)
    tr = S.random_tree(ntips, rng)
    R = S.random_rate_matrix(p, rng)
    X = S.simulate_bm(tr, R, np.zeros(p), rng)
    if bethe:
        prob = S.bethe_of_tree(tr, p)
        packed = S.bm_factors_bethe(tr, prob, R, np.zeros(p), X)
    else:
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
    roles = {"provider": (t[:, 13] < 8, [0, 1, 2, 3, 4, 5, 6, 7]), "consumer": (t[:, 13] >= 8, [0, 4, 5, 6, 7, 8, 9])}
    # the record slots that eliminated, per (grid, group): the consumer rows of the same slot belong to them too
    elim_slots = {(int(r[12]), int(r[14]), int(r[13])) for r in t if r[13] < 8 and r[2] != 0 and r[3] != 0}
    for key in np.unique(tag):
        grid, kind = divmod(int(key), 4)
        kinds = {0: "early", 1: "first", 2: "late", 3: "first"}
        for role, (mask, idx) in roles.items():
            sel = t[(tag == key) & mask]
            keep = np.array([(int(r[12]), int(r[14]), int(r[13]) % 8) in elim_slots for r in sel], dtype=bool)
            sel = sel[keep] if len(sel) else sel
            if len(sel) < 2:
                continue
            d = [(sel[:, idx[k + 1]] - sel[:, idx[k]]) & 0xFFFFFFFF for k in range(len(idx) - 1)]
            tot = (sel[:, idx[-1]] - sel[:, 0]) & 0xFFFFFFFF
            print(f"grid {grid:5d} {kinds[kind]:5s} {role} passes {len(sel):5d} | top -> last stamp {np.median(tot):7.0f} clk | " +
                  " ".join(f"{idx[k]}>{idx[k + 1]}:{np.median(d[k]):6.0f}" for k in range(len(d))))
    # the tail (grid 1), pass by pass: how many records eliminate, how long an elimination takes beside that many others
    tl = t[(tag // 4 == 1) & (t[:, 13] < 8) & (t[:, 2] != 0) & (t[:, 3] != 0)]
    for gg in np.unique(tl[:, 14]):
        r = tl[tl[:, 14] == gg]
        el = (r[:, 3] - r[:, 2]) & 0xFFFFFFFF
        print(f"tail pass {int(gg):3d}: {len(r) / 3:4.1f} eliminations, slots {sorted(set(int(x) for x in r[:, 13]))}, "
              f"median {np.median(el):6.0f} clk, min {el.min():6d}")
    # whole passes of the tail (grid 1): from the top of one to the top of the next, wave 0
    tail = t[(tag // 4 == 1) & (t[:, 13] == 0)]
    if len(tail) > 4:
        tops = np.sort(tail[:, 0])
        dt = np.diff(tops)
        print("tail, wave 0: top-to-top of consecutive passes, median", int(np.median(dt[dt < 100000])))


if __name__ == "__main__":
    main()
