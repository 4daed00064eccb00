#!/usr/bin/env python3
"""Per-level latency of the message kernels on a PATH cluster graph (every level = one message = one launch):
n clusters of dimension m in a row, sepsets of dimension s on the first s variables of both ends, random positive
definite beliefs.  Prints microseconds per level for a postorder + preorder pass, per (m, s).
  python tools/path_latency.py [n]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgbp_amd as P  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    lib = P.load()
    rng = np.random.default_rng(0)
    for (m, s) in ((4, 4), (8, 4), (12, 4), (16, 4), (12, 8), (24, 8), (32, 16), (3, 1), (13, 5), (40, 8)):
        dims = np.array([m] * n + [s] * (n - 1), np.int32)
        sepcl = np.array([[i, i + 1] for i in range(n - 1)], np.int32)
        off = np.arange(0, 2 * (n - 1) * s + 1, s, dtype=np.int64)
        idx = np.tile(np.arange(s, dtype=np.int32), 2 * (n - 1))
        poff = np.concatenate([[0], np.cumsum(dims.astype(np.int64) ** 2 + dims + 1)])
        packed = np.zeros(int(poff[-1]))
        for i in range(n):
            A = rng.normal(size=(m, m))
            J = A @ A.T / m + np.eye(m) * 2
            packed[poff[i]:poff[i] + m * m] = J.reshape(-1, order="F")
            packed[poff[i] + m * m:poff[i] + m * m + m] = rng.normal(size=m)
        cgb = P.ClusterGraphBelief.from_arrays(dims, sepcl, off, idx, packed)
        sched = [(np.arange(n - 1, dtype=np.int32), np.arange(1, n, dtype=np.int32))]
        cgb.set_schedule(sched)
        o = cgb._opts()
        lib.pgbp_enqueue_calibrate(cgb._eng, 2, 1, C.byref(o))
        lib.pgbp_sync(cgb._eng)
        reps = 5
        t = time.perf_counter()
        lib.pgbp_enqueue_calibrate(cgb._eng, reps, 1, C.byref(o))
        lib.pgbp_sync(cgb._eng)
        dt = time.perf_counter() - t
        res = (C.c_int * 10)()
        print(f"m = {m:2d}  s = {s:2d}  ni = {m - s:2d}: {1e6 * dt / reps / (2 * (n - 1)):6.2f} us per level", flush=True)
        del cgb


if __name__ == "__main__":
    main()
