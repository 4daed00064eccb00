#!/usr/bin/env python3
"""Per-phase timing of the generic message kernel on launches with at most 4 tasks (the narrow levels of a loopy
network schedule).  Needs the instrumented build:
  hipcc ... -DPGBP_GTRACE -shared -o build/exp/libpgbp_gtrace.so <csrc sources>      (see tools/README.md)
  PGBP_LIB=build/exp/libpgbp_gtrace.so python tools/trace_generic.py [ntips] [graph]
Phases (s_memtime after s_waitcnt 0): 0 entry, 1 fail word, 2 task_off, 3 entry record, 4 message descriptor,
5 poison word, 6 permutation in LDS, 7 sender gathered, 8 eliminated, 9 stores issued, 10 stores acknowledged, 11 end."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgbp_amd as P  # noqa: E402
from pgbp_amd import _lib as L  # noqa: E402


def main():
    ntips = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    graph = sys.argv[2] if len(sys.argv) > 2 else "bethe"
    rng = np.random.default_rng(3)
    p = 4
    net = P.random_level3_network(ntips, (ntips + 11) // 12, rng, n_colors=3)
    cn, ed, sn = P.bethe(net.node2family) if graph == "bethe" else P.joingraph(net.node2family, 3)
    st = P.allocate_scopes(cn, ed, sn, net, p)
    rates = np.stack([(np.eye(p) + 0.3) * f for f in (0.5, 1.0, 2.0)])
    X = P.simulate_bm_network(net, rates, np.zeros(p), rng)
    pe = [list(zip(net.length[i], net.gamma[i], net.color[i])) for i in range(net.nnodes)]
    fam = P.lg_families(st.clusters, st.node2cluster, net.node2family, st.node2fixed, pe, list(range(net.nnodes)), p, n_rates=3)
    cgb = P.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None)
    cgb.lg_setup(fam, X)
    cgb.assignfactors_lg_(rates, np.zeros(p))
    lib = P.load()
    lib.pgbp_regularize_bycluster(cgb._eng)
    sched = P.spanningtrees_clusterlist(len(cn), ed, cn, net.is_leaf)
    cgb.set_schedule(sched)
    o = cgb._opts()
    lib.pgbp_enqueue_calibrate(cgb._eng, 2, 0, C.byref(o))
    lib.pgbp_sync(cgb._eng)
    f = lib.pgbp_debug_gtrace
    f.argtypes = [C.POINTER(C.c_ulonglong), C.c_uint, C.POINTER(C.c_uint), C.c_int]
    cap = 1 << 16
    buf = np.zeros((cap, 14), dtype=np.uint64)
    n = C.c_uint()
    assert f(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), cap, C.byref(n), 1) == 0
    lib.pgbp_enqueue_calibrate(cgb._eng, 1, 0, C.byref(o))
    lib.pgbp_sync(cgb._eng)
    assert f(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), cap, C.byref(n), 1) == 0
    m = min(n.value, cap)
    t = buf[:m].astype(np.int64)
    real = (t[:, 12] & 0xffffffff)            # 100 MHz ticks, whole wave
    nent = t[:, 12] >> 32
    mf, ni, grid = t[:, 13] & 0xff, (t[:, 13] >> 8) & 0xff, t[:, 13] >> 16
    cyc = (t[:, 11] - t[:, 0]).astype(float)
    ghz = np.median(cyc / (real * 10.0))      # shader clocks per ns
    print(f"{m} traced waves; shader clock ~{ghz:.2f} GHz; wave lifetime median {np.median(real) * 10:.0f} ns")
    names = ["fail word", "task_off", "entry", "descriptor", "poison", "perm->LDS", "gather", "eliminate",
             "stores issued", "stores acked", "rest of task"]
    for sel, lab in ((nent == 1, "single-message tasks"), (nent == 2, "two-message tasks"), (nent >= 3, ">= 3 messages")):
        if not sel.any():
            continue
        print(f"-- {lab}: {int(sel.sum())} waves, mf median {np.median(mf[sel]):.0f}, ni median {np.median(ni[sel]):.0f}, "
              f"lifetime {np.median(real[sel]) * 10:.0f} ns")
        for q in range(11):
            d = (t[sel, q + 1] - t[sel, q]) / ghz
            print(f"   {names[q]:14s} median {np.median(d):7.0f} ns   p90 {np.percentile(d, 90):7.0f} ns")


if __name__ == "__main__":
    main()
