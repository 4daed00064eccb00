#!/usr/bin/env python3
"""Experiment: where the two wavefronts of a task of bp_chunk_pair (pgbp_pair.hip) spend a narrow pass on the cfg5 network.
Needs the instrumented build:
  cd phylogaussianbeliefprop.jl_amd/csrc && make FASTFLAGS=-DPGBP_GSTAMP -B ../../build/obj/pgbp_pair.o && make \
     && cp libpgbp.so ../../build/libpgbp_pstamp.so && make -B ../../build/obj/pgbp_pair.o && make
  PGBP_LIB=build/libpgbp_pstamp.so python tools/stamp_pair.py [joingraph|bethe]
Stamps (s_memtime, one clock for the chip), launches of at most 64 workgroups only.  Provider: 0 message starts -> 1 decoded,
addresses formed -> 2 past the barrier -> 3 operands arrived, frame built -> 4 eliminated -> 5 marginal in LDS -> 6 published.
Consumer: 0 -> 1 decoded -> 2 past the barrier -> 3 operands requested -> 4 publication seen -> 5 marginal read -> 6 stores
issued, acknowledged to the provider."""
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
    args = argparse.Namespace(seed=0, traits=4, blob_style="varied", ntips=20000, blobs=20000 // 12, graph=graph, maxclustersize=3)
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
    lib.pgbp_debug_pstamps.argtypes = [C.c_void_p, C.c_uint, C.c_void_p]
    assert lib.pgbp_debug_pstamps(out.ctypes.data, cap, C.byref(n)) == 0
    for _ in range(2):
        assert lib.pgbp_enqueue_calibrate(cgb._eng, 1, 0, C.byref(opts)) == 0
        assert lib.pgbp_debug_pstamps(out.ctypes.data, cap, C.byref(n)) == 0
    k = min(n.value, cap)
    t = out[:k].astype(np.int64)
    t = t[t[:, 11] != 0]
    print("half-messages stamped (last writer of every slot)", len(t))
    role = t[:, 9]
    for r, name in ((0, "provider"), (1, "consumer")):
        sel = t[role == r]
        d = []
        prev = sel[:, 0]
        for i in range(1, 7):
            cur = np.where(sel[:, i] == 0, prev, sel[:, i])
            d.append(int(np.median((cur - prev) & 0xFFFFFFFF)))
            prev = cur
        print(f"{name}: n {len(sel)} phases {d} total {int(np.median((sel[:, 6] - sel[:, 0]) & 0xFFFFFFFF))}")
    sel = t[(role == 0) & (t[:, 7] != 0)]
    if len(sel):
        print("provider: past the barrier -> sender record staged in LDS (7):", int(np.median(sel[:, 7] - sel[:, 2])),
              " staged -> frame built (3):", int(np.median(sel[:, 3] - sel[:, 7])))
    # pair the two halves of a message: same workgroup, same pair of wavefronts, same sequence number, close in time
    # (a slot keeps its LAST writer: the halves of one message of one launch are at most a pass apart)
    prov, cons = t[role == 0], t[role == 1]
    ckey = {}
    for r in cons:
        ckey[(int(r[8]), int(r[11]))] = ckey.get((int(r[8]), int(r[11])), []) + [r]
    pairs = []
    for r in prov:
        for c in ckey.get((int(r[8]), int(r[11])), []):
            if abs(int(c[4]) - int(r[6])) < 20000:
                pairs.append((r, c))
    print("paired messages", len(pairs))
    if pairs:
        pl = np.array([a for a, _ in pairs]); cl = np.array([b for _, b in pairs])
        print("  provider 2 -> 6 (past the barrier -> published):", int(np.median(pl[:, 6] - pl[:, 2])))
        print("  consumer sees the publication after:", int(np.median(cl[:, 4] - pl[:, 6])))
        print("  consumer 4 -> 6 (seen -> stores issued, acknowledged to the provider):", int(np.median(cl[:, 6] - cl[:, 4])))
        # the next pass of the same workgroup: the first provider stamp 2 behind this consumer's stamp 6
        gaps = []
        for b in np.unique(pl[:, 8]):
            p2 = np.sort(prov[prov[:, 8] == b][:, 2])
            for c in cl[cl[:, 8] == b]:
                k = np.searchsorted(p2, c[6], side="left")
                if k < len(p2) and p2[k] - c[6] < 20000:
                    gaps.append(int(p2[k] - c[6]))
        if gaps:
            print("  consumer's last store -> a provider of the workgroup past the next barrier: median", int(np.median(gaps)),
                  "p10", int(np.percentile(gaps, 10)), "p90", int(np.percentile(gaps, 90)), "n", len(gaps))
        # pass to pass: consecutive provider barrier exits of one workgroup
        pp = []
        for b in np.unique(prov[:, 8]):
            p2 = np.unique(np.sort(prov[prov[:, 8] == b][:, 2]) // 256)   # (the providers of one pass leave the barrier together)
            d = np.diff(p2) * 256
            pp += [int(x) for x in d if 1000 < x < 20000]
        if pp:
            print("  barrier exit to barrier exit (one workgroup): median", int(np.median(pp)), "p10", int(np.percentile(pp, 10)),
                  "p90", int(np.percentile(pp, 90)), "n", len(pp))


if __name__ == "__main__":
    main()
