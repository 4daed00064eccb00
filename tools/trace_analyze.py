#!/usr/bin/env python3
"""Concurrency / lifetime summary of gpurun_out/trace_all.npy (tools/trace_narrow_levels.py ... all)."""
import sys
import numpy as np

t = np.load(sys.argv[1]).astype(np.int64)
t = t[t[:, 9] > 0]
grid = t[:, 7] >> 32
t0 = t[:, 8].min()
print("tasks  waves span_us start_us  life p10/p50/p90 us   max_conc mean_conc  start p50/p90/max us")
first = {}
for g in np.unique(grid):
    first[int(t[grid == g, 8].min())] = int(g)
for st in sorted(first):
    g = first[st]
    m = grid == g
    s, e = t[m, 8], t[m, 9]
    ev = np.concatenate([np.stack([s, np.ones_like(s)], 1), np.stack([e, -np.ones_like(e)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    conc = np.cumsum(ev[:, 1])
    life = (e - s) / 100.0
    rel = (s - s.min()) / 100.0
    span = (e.max() - s.min()) / 100.0
    print(f"{g:6d} {m.sum():6d} {span:7.1f} {(s.min() - t0) / 100.0:8.1f}   {np.percentile(life, 10):5.1f}/{np.percentile(life, 50):5.1f}/"
          f"{np.percentile(life, 90):5.1f}   {conc.max():6d} {(e - s).sum() / (e.max() - s.min()):8.0f}   "
          f"{np.percentile(rel, 50):5.1f}/{np.percentile(rel, 90):5.1f}/{rel.max():5.1f}")
