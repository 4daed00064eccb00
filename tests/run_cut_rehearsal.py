#!/usr/bin/env python3
"""A cluster graph CUT between PROCESSES (third bullet), rehearsed on one GPU:
launched like the driver launches the bench (torch.distributed.run, one process per rank), every rank on device 0 with its
own engine over the same level-3 network's cluster graph, gloo in place of RCCL (two RCCL ranks cannot share a device).
sharding.NetworkCut runs calibrate! with every traversal of src/calibration.jl:111-161 cut by spanning-tree subtrees; the
exchanges go through HostExchange (pgbp_pack_beliefs -> all_gather -> pgbp_unpack_beliefs), the AND of the ranks' flags and
successes through an all-reduce(min).  Rank 0 compares EVERY belief, the calibration flag and the auto-stop point with one
engine's run of the same calibrate! and prints one JSON line.
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tests/run_cut_rehearsal.py [bethe|joingraph|cliquetree]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    import pgbp_amd
    from pgbp_amd import _lib as L
    from pgbp_amd.sharding import HostExchange, NetworkCut
    from test_gpu_multidevice import _small_network
    graph = sys.argv[1] if len(sys.argv) > 1 else "bethe"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    lib = pgbp_amd.load()
    net, (cn, ed, sn), st, fam, X, rates, mu, sched = _small_network(graph)     # same seed on every rank
    loopy = len(ed) > len(cn) - 1

    def fresh():
        return pgbp_amd.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None)

    mine = fresh()
    mine.lg_setup(fam, X)
    mine.assignfactors_lg_(rates, mu)
    if loopy and graph == "joingraph":
        from pgbp_amd.regularization import regularizebeliefs_onschedule_
        regularizebeliefs_onschedule_(mine)
    elif loopy:
        assert lib.pgbp_regularize_bycluster(mine._eng) == 0
    mine.pull()
    start = mine._packed[0].copy()
    mine._upload(snapshot_factors=True)

    def allmin(v):
        t = torch.tensor([int(v)], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item())

    cut = NetworkCut([mine], sched, rank=rank, n_ranks=world, exchange=HostExchange(dist, mine, rank, world), allmin=allmin)
    niter = 3 if loopy else 1
    succ, iscal, _ = cut.calibrate(niter)
    got = cut.gather()
    reached = None
    if loopy:   # the automatic stop, each message's flag answered for by the rank that sent it last
        mine._packed[0, :] = start
        mine._upload(snapshot_factors=True)
        assert lib.pgbp_reset_flags(mine._eng, 1) == 0
        cut.last_writer = {}
        s2, i2, reached = cut.calibrate(60, auto=True)
    out = {"rank": rank, "succ": bool(succ)}
    if rank == 0:
        one = fresh()
        one._packed[0, :] = start
        one._upload(snapshot_factors=True)
        one.set_schedule(sched)
        res = (L.Result * 1)()
        o = one._opts()
        assert lib.pgbp_calibrate(one._eng, niter, C.byref(o), res) == 0 and res[0].succ
        want = np.zeros((1, len(start)))
        assert lib.pgbp_get_beliefs(one._eng, L.f64p(want)) == 0
        flags = np.zeros(2 * one.nsepsets, np.int32)
        assert lib.pgbp_get_residuals(one._eng, None, L.i32p(flags), None, None) == 0
        err = float(np.max(np.abs(got - want[0]) / np.maximum(1.0, np.abs(want[0]))))
        out.update(graph=graph, n_ranks=world, clusters=len(cn), sepsets=len(ed), schedule_trees=len(sched),
                   subtrees=[len(c["sub"]) for c in cut.cuts], max_rel_belief_diff=err, iscal=bool(iscal),
                   iscal_one_engine=bool(flags.all()), exchanged_doubles=int(cut.exchanged_doubles))
        if loopy:
            one._packed[0, :] = start
            one._upload(snapshot_factors=True)
            assert lib.pgbp_reset_flags(one._eng, 1) == 0
            oa = one._opts(auto=True)
            res2 = (L.Result * 1)()
            assert lib.pgbp_calibrate(one._eng, 60, C.byref(oa), res2) == 0 and res2[0].succ
            out.update(auto_reached=list(reached) if reached else None,
                       auto_reached_one_engine=[int(res2[0].iter_reached), int(res2[0].tree_reached)] if res2[0].iscal else None)
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
