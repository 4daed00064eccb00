#!/usr/bin/env python3
"""Probe (round 4): does the per-level fixed cost of the wide level launches hide behind ANOTHER stream's work?
Two engines (each owns a HIP stream), each a clique tree of a --ntips tree: K calibrates enqueued on one engine, then on
the other (back to back = no overlap), against K on both at once.  If the concurrent pair takes clearly less than the sum,
independent strands of ONE tree's level schedule on several streams are worth building.
    python tools/two_stream_probe.py [--ntips 25000 --traits 16 --steps 50 --engines 2]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ntips", type=int, default=25000)
    ap.add_argument("--traits", type=int, default=16)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--engines", type=int, default=2)
    ap.add_argument("--graph", default="cliquetree")
    args = ap.parse_args()
    import torch
    import pgbp_amd
    from bench import build_workload
    lib = pgbp_amd.load()
    engines = []
    for e in range(args.engines):
        tr, prob, packed, ll_check, _ = build_workload(args.ntips, args.traits, 3 + e, args.graph)
        cgb = pgbp_amd.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx,
                                                      packed, device=0)
        cgb.set_schedule(prob.schedule)
        engines.append((cgb, cgb._eng, cgb._opts()))

    def check(eng, code):
        if code != 0:
            raise RuntimeError(lib.pgbp_last_error(eng).decode())

    def enqueue(which, k):
        for i in which:
            cgb, eng, opts = engines[i]
            check(eng, lib.pgbp_enqueue_calibrate(eng, k, 0, C.byref(opts)))

    def sync(which):
        for i in which:
            check(engines[i][1], lib.pgbp_sync(engines[i][1]))

    allk = list(range(args.engines))
    enqueue(allk, 5)
    sync(allk)
    torch.cuda.synchronize()
    out = {"ntips": args.ntips, "traits": args.traits, "graph": args.graph, "steps": args.steps, "engines": args.engines}
    alone = []
    for i in allk:
        t = time.perf_counter()
        enqueue([i], args.steps)
        sync([i])
        alone.append((time.perf_counter() - t) / args.steps * 1e3)
    out["alone_ms_per_calibrate"] = alone
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        for _k in range(args.steps // 5):      # interleaved on the host so neither stream runs dry
            enqueue(allk, 5)
        sync(allk)
        ts.append((time.perf_counter() - t) / args.steps * 1e3)
    out["concurrent_ms_per_round_of_all"] = ts
    out["sum_alone_ms"] = sum(alone)
    out["concurrent_over_sum"] = min(ts) / sum(alone)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
