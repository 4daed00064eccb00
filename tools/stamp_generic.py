#!/usr/bin/env python3
"""Experiment: where a message of the wave-per-task kernels (bp_level_generic / bp_chunk_generic) spends its time on the
narrow launches (<= 512 workgroups) of the cfg5 network workload.  Needs the instrumented build:
  cd phylogaussianbeliefprop.jl_amd/csrc && make KERNFLAGS=-DPGBP_GSTAMP -B ../../build/obj/pgbp_kernels.o && make \
     && cp libpgbp.so ../../build/libpgbp_gstamp.so && make -B ../../build/obj/pgbp_kernels.o && make
  PGBP_LIB=build/libpgbp_gstamp.so python tools/stamp_generic.py [joingraph|bethe] [ntips]
Phases (shader clocks, medians): 0 message starts (record resident) -> 1 operands requested and arrived, sender in LDS
-> 2 synchronised -> 3 fake test + symmetrisation done -> 4 elimination done -> 5 divide starts -> 6 stores issued
-> 7 stores acknowledged."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgbp_amd as P  # noqa: E402
import bench as B  # noqa: E402


def main():
    graph = sys.argv[1] if len(sys.argv) > 1 else "joingraph"
    ntips = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    args = argparse.Namespace(seed=0, traits=4, blob_style="varied", ntips=ntips, blobs=ntips // 12, graph=graph, maxclustersize=3)
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
    cap = 1 << 16
    out = np.zeros((cap, 12), dtype=np.uint32)
    n = C.c_uint(0)
    lib.pgbp_debug_gstamps.argtypes = [C.c_void_p, C.c_uint, C.c_void_p]
    assert lib.pgbp_debug_gstamps(out.ctypes.data, cap, C.byref(n)) == 0      # drop what the set-up recorded
    for _ in range(2):
        assert lib.pgbp_enqueue_calibrate(cgb._eng, 1, 0, C.byref(opts)) == 0
        assert lib.pgbp_debug_gstamps(out.ctypes.data, cap, C.byref(n)) == 0
    k = min(n.value, cap)
    t = out[:k].astype(np.int64)
    print("messages stamped", n.value)
    # PGBP_STAMP_MAX_GRID=n: only the messages of launches of at most n workgroups (the narrow passes: an iteration's critical path)
    max_grid = int(os.environ.get("PGBP_STAMP_MAX_GRID", "0"))
    if max_grid > 0:
        t = t[t[:, 11] <= max_grid]
        print("messages of launches of at most", max_grid, "workgroups:", len(t))
    dims = t[:, 10]
    mode = (dims >> 24) & 1
    for md in (0, 1):
        for key in sorted(set(dims[mode == md] & 0xFFFFFF)):
            sel = t[(mode == md) & ((dims & 0xFFFFFF) == key)]
            if len(sel) < 20:
                continue
            mf, ni, s = key & 255, (key >> 8) & 255, (key >> 16) & 255
            d = []
            prev = sel[:, 0]
            for i in range(1, 8):
                cur = np.where(sel[:, i] == 0, prev, sel[:, i])     # a phase that did not run keeps the previous stamp
                d.append(int(np.median((cur - prev) & 0xFFFFFFFF)))
                prev = cur
            tot = int(np.median((sel[:, 7] - sel[:, 0]) & 0xFFFFFFFF))
            print(f"{'chunk' if md else 'level'} mf {mf:2d} ni {ni:2d} s {s:2d}: n {len(sel):6d}  phases {d}  total {tot} clk")


if __name__ == "__main__":
    main()
