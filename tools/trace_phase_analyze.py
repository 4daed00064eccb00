#!/usr/bin/env python3
"""Median per-phase shader cycles by task-index ventile, full trace (-DPGBP_TRACE without LIGHT)."""
import sys
import numpy as np
t = np.load(sys.argv[1]).astype(np.int64)
t = t[t[:, 9] > 0]
grid = t[:, 7] >> 32
blk = t[:, 7] & 0xffffffff
for g in np.unique(grid):
    m = grid == g
    tt, b = t[m], blk[m]
    order = np.argsort(b)
    n = len(order)
    print(f"launch grid={g}: cumulative shader cycles at marks [record, data-arrived, elim-done, handover, stores-issued, stores-acked]; life in us")
    for q in range(0, 100, 10):
        sel = order[int(n * q / 100): int(n * (q + 10) / 100)]
        x = tt[sel]
        ok = (x[:, 2] > 0) & (x[:, 2] < 10**7)   # waves that eliminated
        med = [int(np.median(x[ok, j])) if ok.any() else -1 for j in range(1, 7)]
        life = (x[:, 9] - x[:, 8]) / 100.0
        print(f"  tasks {q:3d}-{q+10:3d}%: n_elim {ok.sum():6d} {med}  life p50 {np.median(life):5.1f}  (eliminating waves: {np.median(life[ok]) if ok.any() else 0:5.1f})")
