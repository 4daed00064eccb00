"""CPU test of bench.py's N > 1 path: 2 ranks over gloo run the timing contract (barrier + sync on both
sides, MAX over ranks, whole-job aggregate).  The hot path itself does not shard for one big tree
("replicas only"), so there is no data-path collective to test."""
import os
import sys
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sleep = 0.05 if rank == 0 else 0.25   # uneven ranks: the MAX must win

    def work():
        time.sleep(sleep)
    dt = bench.timed_region(work, dist, lambda: None, reduce_device="cpu")
    value = bench.whole_job_rate(1000, 4, world, dt)
    q.put((rank, dt, value))
    dist.destroy_process_group()


def test_two_rank_timing_contract():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, dt0, v0), (r1, dt1, v1) = out
    assert dt0 == dt1 and 0.25 <= dt0 < 1.0          # both ranks report the slowest rank's time
    assert v0 == v1 == 2 * 1000 * 4 / dt0            # whole-job aggregate, not per-rank


def test_single_rank_path():
    sys.path.insert(0, ROOT)
    import bench
    dt = bench.timed_region(lambda: time.sleep(0.01), None, lambda: None)
    assert 0.01 <= dt < 0.5
    assert bench.whole_job_rate(10, 2, 1, 2.0) == 10.0


def _gather_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import numpy as np
    import pgbp_amd  # noqa: F401  (host-side helpers only)
    from pgbp_amd.sharding import gather_sites, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_total = 11                                  # uneven split: 6 + 5
    lo, hi = shard_range(n_total, rank, world)
    local = np.arange(lo, hi, dtype=np.float64) * 1.5 - 3.0   # stands for this rank's per-site log-likelihoods
    full = gather_sites(local, n_total, dist, device="cpu")
    q.put((rank, (lo, hi), full.tolist()))
    dist.destroy_process_group()


def test_site_sharding_gather_two_ranks():
    """The N > 1 data path of the site-sharded configuration (cfg4): contiguous shards, one all-gather."""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out[0][1] == (0, 6) and out[1][1] == (6, 11)
    expect = (np.arange(11) * 1.5 - 3.0).tolist()
    assert out[0][2] == expect and out[1][2] == expect


def test_shard_range_properties():
    sys.path.insert(0, ROOT)
    import pgbp_amd  # noqa: F401
    from pgbp_amd.sharding import shard_range
    for n in (0, 1, 7, 1000):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def test_gpus_flag_never_prints_a_figure_for_another_world_size():
    """`--gpus N` must mean N ranks: with a launcher environment of another size the run exits non-zero before touching
    the GPU (without a launcher environment it starts torch.distributed.run itself; not exercised here: no GPU)."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "--gpus 2 but WORLD_SIZE=1" in (out.stderr + out.stdout)
    assert '"metric"' not in out.stdout


def test_unpack_of_two_fabricated_rank_slots():
    """The host half of pgbp_comm_gather_loglik (pgbp_comm_unpack_slots, no GPU / RCCL needed) on the buffer two ranks
    would have gathered: 11 sites split 6 + 5 in slots of 6, a failed site on rank 1 (info word, succ = 0)."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import pgbp_amd  # noqa: F401
    from pgbp_amd.sharding import shard_range, unpack_slots
    n_total, world = 11, 2
    slot = -(-n_total // world)
    ll = np.arange(n_total) * -2.5 - 100.0
    recv = np.zeros((world, 2 * slot + 2))
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        recv[r, : hi - lo] = ll[lo:hi]
        recv[r, 2 * slot] = 1.0          # succ
        recv[r, 2 * slot + 1] = 1.0      # iscal
    recv[1, slot + 3] = 7.0              # rank 1, its 4th site: PosDefException.info = 7
    recv[1, 2 * slot] = 0.0              # ... so that rank's calibration did not succeed
    recv[0, 2 * slot + 1] = 0.0          # rank 0 not calibrated
    norm, info, succ, iscal = unpack_slots(recv.reshape(-1), world, slot)
    full = np.concatenate([norm[r, : shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0]] for r in range(world)])
    assert full.tolist() == ll.tolist()
    assert norm[1, 5] == 0.0                        # the unused tail of the shorter rank's slot
    assert info[1, 3] == 7 and int(np.count_nonzero(info)) == 1
    assert succ is False and iscal is False
    recv[1, 2 * slot] = 1.0
    recv[0, 2 * slot + 1] = 1.0
    _, _, succ, iscal = unpack_slots(recv.reshape(-1), world, slot)
    assert succ is True and iscal is True


def test_rccl_not_found_is_an_error_code_not_a_crash():
    """the RCCL-not-found path read dlerror() twice (the second call returns NULL: a segfault).  A child
    process points the loader at a library that does not exist and must get PGBP_ERR_NO_DEVICE plus a message."""
    import subprocess
    code = ("import sys, ctypes as C; sys.path.insert(0, %r); import pgbp_amd; from pgbp_amd import _lib as L; lib = L.load();"
            "ident = (C.c_uint8 * 128)(); rc = lib.pgbp_comm_unique_id(ident); msg = lib.pgbp_comm_last_error(None).decode();"
            "h = C.c_void_p(); rc2 = lib.pgbp_comm_create(ident, 1, 0, 0, C.byref(h)); rc3 = lib.pgbp_comm_precheck(0);"
            "print(rc, rc2, rc3, msg)") % ROOT
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PGBP_RCCL_LIB="/nonexistent/librccl.so.1"),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout, out.stderr[-800:])
    rc, rc2, rc3, msg = out.stdout.strip().split(" ", 3)
    assert (rc, rc2, rc3) == ("5", "5", "5")          # PGBP_ERR_NO_DEVICE
    assert "RCCL not found" in msg and "/nonexistent/librccl.so.1" in msg


def _comm_fail_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    if rank == 1:
        os.environ["PGBP_RCCL_LIB"] = "/nonexistent/librccl.so.1"   # only this rank cannot open its communicator
    import pgbp_amd  # noqa: F401
    from pgbp_amd import _lib as L
    from pgbp_amd.sharding import Comm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def bcast(raw):
        t = torch.zeros(1 + Comm.ID_BYTES, dtype=torch.uint8)
        if raw is not None:
            t.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
        dist.broadcast(t, src=0)
        return bytes(t.numpy().tobytes())

    def allmin(v):
        t = torch.tensor([int(v)], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item())
    try:
        Comm(world, rank, 0, bcast, allmin)
        q.put((rank, "created"))
    except L.PgbpError as ex:
        q.put((rank, f"PgbpError {ex.code}"))
    dist.barrier()   # every rank is still in step with the launcher's group
    dist.destroy_process_group()


def test_comm_failure_on_one_rank_is_raised_by_every_rank():
    """a rank that fails before the launcher's broadcast (or inside ncclCommInitRank) must not leave its
    peers in a collective.  Rank 1 cannot load RCCL: both ranks raise the same error after the agreed minimum and go on
    to the next collective of the launcher's group together (here, without a GPU, rank 0's own precheck fails as well;
    what is tested is that nobody hangs and nobody proceeds alone)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_comm_fail_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out[0][1].startswith("PgbpError") and out[1][1].startswith("PgbpError")


def test_cut_of_a_spanning_tree_partitions_its_edges():
    """sharding.cut_spanning_tree (the cfg5-across-GPUs cut): top + subtrees hold every edge exactly
    once; every subtree edge list is itself a preorder list of a tree whose root is the child end of a boundary edge;
    boundary edges belong to the top; the loads of the ranks are balanced to within the largest subtree; a tree too small
    to cut stays whole."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from pgbp_amd.sharding import cut_spanning_tree
    rng = np.random.default_rng(11)
    for n_nodes, K in ((2000, 2), (5000, 4), (777, 8)):
        # a random recursive tree, preorder edge list by DFS
        parent = np.zeros(n_nodes, np.int64)
        for v in range(1, n_nodes):
            parent[v] = rng.integers(0, v)
        kids = [[] for _ in range(n_nodes)]
        for v in range(1, n_nodes):
            kids[parent[v]].append(v)
        pa, ch, stack = [], [], [0]
        while stack:
            u = stack.pop()
            for v in kids[u]:
                pa.append(u)
                ch.append(v)
                stack.append(v)
        # (a stack DFS emits all children of u before descending: still parent-before-child, as the planner requires)
        cut = cut_spanning_tree(pa, ch, K)
        all_edges = set(zip(pa, ch))
        top = set(zip(cut["top"][0].tolist(), cut["top"][1].tolist()))
        seen = set(top)
        assert len(top) == len(cut["top"][0])
        roots = {int(ch[i]) for i in cut["boundary"]}
        assert {(int(pa[i]), int(ch[i])) for i in cut["boundary"]} <= top
        load = [0] * K
        for (k, root, (spa, sch)) in cut["sub"]:
            assert root in roots and 0 <= k < K
            placed = {root}
            for a, c in zip(spa.tolist(), sch.tolist()):
                assert a in placed and c not in placed        # preorder list of a tree rooted at `root`
                placed.add(c)
                assert (a, c) not in seen
                seen.add((a, c))
            load[k] += len(spa)
        assert seen == all_edges
        assert len(cut["sub"]) >= 4 * K or not cut["sub"]
        if cut["sub"]:
            biggest = max(len(s[2][0]) for s in cut["sub"])
            assert max(load) - min(load) <= biggest
    whole = cut_spanning_tree([0, 1], [1, 2], 2)
    assert whole["sub"] == [] and len(whole["top"][0]) == 2


def _comm_create_fails_on_one_rank(rank, world, port, q):
    sys.path.insert(0, ROOT)
    """2 gloo ranks: the precheck passes everywhere, then rank 1's pgbp_comm_create fails -- BOTH ranks must raise (the second
    agreement of Comm.__init__), nobody may walk on into a collective."""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pgbp_amd import _lib as L
        from pgbp_amd.sharding import Comm

        def bcast(raw):
            t = torch.zeros(1 + Comm.ID_BYTES, dtype=torch.uint8)
            if raw is not None:
                t.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
            dist.broadcast(t, src=0)
            return bytes(t.numpy().tobytes())

        def allmin(v):
            t = torch.tensor([int(v)], dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t.item())

        class FakeLib:   # the precheck and rank 0's unique id succeed without a GPU; the create fails on rank 1 only
            def __init__(self, real):
                self._real = real

            def pgbp_comm_precheck(self, device):
                return 0

            def pgbp_comm_unique_id(self, ident):
                return 0

            def pgbp_comm_last_error(self, c):
                return b"ncclCommInitRank: injected failure" if rank == 1 else b""

            def pgbp_comm_create(self, ident, n, r, device, out):
                return L.ERR_HIP if rank == 1 else 0

            def pgbp_comm_destroy(self, c):
                return None

        orig = L.load
        L.load = lambda: FakeLib(None)
        try:
            try:
                Comm(world, rank, 0, bcast, allmin)
                q.put((rank, "no error"))
            except L.PgbpError as ex:
                q.put((rank, "raised: " + ex.msg))
            try:
                Comm(world, rank, 0, bcast, None)
                q.put((rank, "no ValueError"))
            except ValueError:
                q.put((rank, "ValueError"))
        finally:
            L.load = orig
    finally:
        dist.destroy_process_group()


def test_comm_create_failure_on_one_rank_is_raised_by_every_rank():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 90)
    procs = [ctx.Process(target=_comm_create_fails_on_one_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=10) for _ in range(4))
    assert [g for g in got if g[0] == 0] == [(0, "ValueError"), (0, "raised: pgbp_comm: another rank could not create its communicator")]
    assert [g for g in got if g[0] == 1] == [(1, "ValueError"), (1, "raised: ncclCommInitRank: injected failure")]
