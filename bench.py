#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the MI355X belief-propagation engine.

Metric (BASELINE.json): clique-tree messages/sec (+ log-likelihood evals/sec), homogeneous BM,
16 traits, 50k-tip random bifurcating tree, clique tree (SURVEY.md section 8(d) "cfg3").
One "step" = one calibrate!() (postorder + preorder over the whole clique tree = 2 * n_sepsets
directed canonical messages, src/calibration.jl:72-84) on factors already resident in HBM.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, each calibrating its own replica of the workload (one big tree does
not shard -- "replicas only"); no data-path collective; timing = barrier +
synchronize on both sides, MAX over ranks; value = all ranks' messages / that time.
The same line carries, for every N, the block `site_sharded_cfg4`: BASELINE.json configs[3] (8000
independent univariate OU problems on a 20k-tip tree) sharded over the N ranks -- the configuration
that does shard (strong scaling), its one all-gather (RCCL, behind the C ABI) inside the timed region.
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # before torch initialises HIP (see pgbp_amd/_lib.py)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def build_workload(ntips, p, seed, graph):
    from pgbp_amd import synth as S
    rng = np.random.default_rng(seed)
    tr = S.random_tree(ntips, rng)
    R = S.random_rate_matrix(p, rng)
    mu = np.zeros(p)
    X = S.simulate_bm(tr, R, mu, rng)
    if graph == "cliquetree":
        prob = S.cliquetree_of_tree(tr, p)
        packed = S.bm_factors_cliquetree(tr, prob, R, mu, X)
    else:
        prob = S.bethe_of_tree(tr, p)
        packed = S.bm_factors_bethe(tr, prob, R, mu, X)
    ll_check = S.bm_loglik_pruning(tr, R, mu, X)
    return tr, prob, packed, ll_check, (R, mu, X)


def host_cores():
    """Cores this process may really use: the scheduler affinity, cut down to the cgroup's CPU quota where there is one
    (a GPU box hands a 1-GPU job 16 of its 256 logical CPUs), and never more than 16 per GPU without a quota to read."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            return max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return min(n, 16)


def cpu_baseline(prob, packed, budget_s=20.0):
    """The oracle's plain-C sequential engine (reference message order, 1 core) on a bounded sample:
    whole calibrate!() passes over the same workload until ~budget_s of CPU time is spent."""
    try:
        from oracle import cengine
    except Exception as ex:  # oracle not built: report, never fake
        return {"value": None, "unit": "messages/s", "cores": 1, "kind": "port", "sample": f"unavailable: {ex}"}
    cengine.use_native_build()  # -O3 -march=native on this host
    eng = cengine.Engine(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx, packed)
    pa, ch = prob.schedule[0]
    eng.prepare(pa, ch)
    nmsg = 2 * len(pa)
    t_total, passes = 0.0, 0
    ll = None
    while t_total < budget_s and passes < 50:
        eng.reset()
        t0 = time.perf_counter()
        ok = eng.calibrate()
        t_total += time.perf_counter() - t0
        passes += 1
        assert ok
    ll = eng.integrate(prob.root_cluster)[1]
    out = {"value": nmsg * passes / t_total, "unit": "messages/s", "cores": 1, "kind": "port",
           "sample": f"{passes} full calibrate!() passes of the same workload ({nmsg} messages each), "
                     f"oracle/c sequential engine (gcc -O3 -march=native, 1 thread of {os.cpu_count()} host cpus), "
                     f"{t_total:.1f} s",
           "loglik": ll}
    # secondary: the same engine level by level on every host core (OpenMP); the reference
    # itself is single-threaded, so this is the socket-vs-GPU view, not the headline baseline
    try:
        ncores = host_cores()
        levels = eng.levels_of_tree(pa, ch)
        eng.reset()
        assert eng.calibrate_levels(levels, 1, ncores)      # warm-up: thread pool, caches
        t_all, passes_all = 0.0, 0
        while t_all < max(2.0, budget_s / 4) and passes_all < 50:
            eng.reset()
            t0 = time.perf_counter()
            ok = eng.calibrate_levels(levels, 1, ncores)
            t_all += time.perf_counter() - t0
            passes_all += 1
            assert ok
        ll_all = eng.integrate(prob.root_cluster)[1]
        out["all_cores"] = {"value": nmsg * passes_all / t_all, "unit": "messages/s", "cores": ncores, "kind": "port",
                            "sample": f"{passes_all} level-synchronous calibrate!() passes (OpenMP, {ncores} threads: tasks of a "
                                      f"level in parallel, the reference's order inside a task), {t_all:.1f} s",
                            "loglik_rel_diff_vs_1_core": abs(ll_all - ll) / max(1.0, abs(ll))}
    except Exception as ex:  # an oracle build without OpenMP: say so
        out["all_cores"] = {"value": None, "sample": f"unavailable: {ex}"}
    # opportunistic: the real Julia calibrate!() where julia + the package already exist
    try:
        import subprocess
        raw = subprocess.run(["sh", os.path.join(ROOT, "bench", "run_reference_calibrate.sh")], capture_output=True,
                             text=True, timeout=600).stdout.strip().splitlines()
        out["reference_julia"] = json.loads(raw[-1]) if raw else {"available": False, "reason": "no output"}
    except Exception as ex:
        out["reference_julia"] = {"available": False, "reason": str(ex)}
    return out


def timed_region(enqueue_and_wait, dist, device_sync, reduce_device=None):
    """The timing contract: barrier + device sync on both sides of the timed work, MAX over ranks.
    `enqueue_and_wait()` runs exactly the K timed steps of THIS rank.  Returns seconds (same on all ranks).
    Factored out so that the N > 1 path is covered by a 2-rank gloo test on CPU (tests/test_bench_dist.py)."""
    def barrier():
        if dist is not None:
            dist.barrier()
        device_sync()
    barrier()
    t0 = time.perf_counter()
    enqueue_and_wait()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        if reduce_device is None:
            reduce_device = "cpu" if dist.get_backend() == "gloo" else "cuda"
        t = torch.tensor([dt], device=reduce_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def make_comm(dist, torch, rank, world, local_rank):
    """pgbp_comm (C ABI, RCCL bound inside libpgbp.so) for the one exchange of the sharded paths; rank 0's unique id
    travels over the launcher's torch.distributed group, as does the agreement that every rank can open its communicator.
    None for a single rank or the one-GPU gloo rehearsal (two RCCL ranks cannot share a device).  A communicator that does
    not come up is an ERROR on every rank (the run exits non-zero): there is no fallback to another collective."""
    if world == 1 or dist is None or os.environ.get("PGBP_BENCH_REHEARSAL") == "1":
        return None
    from pgbp_amd.sharding import Comm

    def bcast(raw):
        t = torch.zeros(1 + Comm.ID_BYTES, dtype=torch.uint8, device=f"cuda:{local_rank}")
        if raw is not None:
            t.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
        dist.broadcast(t, src=0)
        return bytes(t.cpu().numpy().tobytes())

    def allmin(v):
        t = torch.tensor([int(v)], dtype=torch.int32, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item())
    return Comm(world, rank, local_rank, bcast, allmin)


def whole_job_rate(units_per_rank_step, steps, world, dt):
    """value = the units ALL ranks processed / the max-over-ranks time (weak scaling: per-rank work fixed)."""
    return world * units_per_rank_step * steps / dt


def measured_copy_bandwidth(torch, local_rank):
    """GB/s (read + write) of a 1 GiB device-to-device copy, best of 5, timed with events on torch's current stream."""
    try:
        dev = f"cuda:{local_rank}"
        x = torch.empty(1 << 27, dtype=torch.float64, device=dev)
        y = torch.empty_like(x)
        x.zero_()
        best = None
        for _ in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            y.copy_(x)
            b.record()
            b.synchronize()
            ms = a.elapsed_time(b)
            best = ms if best is None else min(best, ms)
        del x, y
        return 2.0 * (1 << 30) / (best * 1e-3) / 1e9
    except Exception:
        return None


def csrc_sha16():
    """Stamp of the kernel sources of the running tree: sha256 over csrc/*.{hip,hpp,cpp} in name order, first 16 hex digits.
    tools/pmc_traffic.py writes the same stamp into the PMC reduction it produces."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "phylogaussianbeliefprop.jl_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.hpp")) + glob.glob(os.path.join(d, "*.cpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_pmc_traffic(name="pmc_traffic_latest.json"):
    """HBM bytes per launch of the message kernel from the committed rocprofv3 PMC passes
    (tools/pmc_traffic.py -> profiles/pmc_traffic_latest.json; the sites workload: pmc_traffic_sites_latest.json);
    None if absent.  `stale` is set when the file's source stamp is not the running tree's (or it has none): the figure
    then belongs to other kernels than the ones being timed."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            d = json.load(f)
    except Exception:
        return None
    d["stale"] = d.get("csrc_sha16") != csrc_sha16()
    return d


def run_sites(args, torch, dist, rank, world, local_rank, emit=True):
    """cfg4 (BASELINE.json configs[3]: "OU model, 8 traits x 1000 independent sites, 20k-tip tree, sites sharded"):
    the reference's only OU is UnivariateOrnsteinUhlenbeck, so 8 traits x 1000 sites = 8000 independent univariate OU
    problems (one sigma2, alpha, theta, mu and one data column each) on one tree.  Problems are
    sharded contiguously across ranks (STRONG scaling: the 8000 problems are fixed), no communication during calibration,
    ONE all-gather (RCCL, behind the C ABI: pgbp_comm) of the per-problem log-likelihoods per step.  Factors are
    assigned on the device (pgbp_lg_assignfactors, OU), so only the four parameters per problem and the tip data cross the
    bus.  --site-model bm: univariate BM instead.

    Two timed regions, both under the timing contract (barrier + sync on both sides, MAX over ranks):
      * the sharded step -- calibrate!() of every local problem, then the score() body (device factor fill + postorder +
        root integratebelief!) and THE ALL-GATHER that puts every problem's log-likelihood on every rank, K times:
        `value` and `sharded_step` (the collective is inside the timed region);
      * calibrate!() alone, K times back to back (`calibrate_only`: the hot path without its exchange step).
    Returns the result dict on rank 0 (None elsewhere); emit: print it as the JSON line."""
    import pgbp_amd
    from pgbp_amd import _lib as L
    from pgbp_amd import synth as S
    from pgbp_amd.sharding import gather_sites, shard_range
    lib = pgbp_amd.load()
    nprob = args.sites * args.site_traits
    lo, hi = shard_range(nprob, rank, world)
    ns = hi - lo
    rng = np.random.default_rng(args.seed)               # same tree and parameters on every rank
    tr = S.random_tree(args.ntips, rng)
    sigma2_all = rng.uniform(0.5, 2.0, size=nprob)
    alpha_all = rng.uniform(0.1, 1.0, size=nprob)
    theta_all = rng.normal(size=nprob)
    mu_all = rng.normal(size=nprob)
    sigma2, alpha, theta, mu = sigma2_all[lo:hi], alpha_all[lo:hi], theta_all[lo:hi], mu_all[lo:hi]
    rng_x = np.random.default_rng(args.seed + 1000 + rank)
    prob = S.cliquetree_of_tree(tr, 1)
    cgb = pgbp_amd.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx,
                                                  None, n_sites=ns, device=local_rank)
    cgb.set_schedule(prob.schedule)
    ou = args.site_model == "ou"
    if ou:
        X = S.simulate_ou_uni_sites(tr, sigma2, alpha, theta, mu, rng_x)
        ll_check = S.ou_loglik_pruning_uni_sites(tr, sigma2, alpha, theta, mu, X)
        cgb.lg_setup(S.lg_tree_table(tr, prob, 1), X[:, :, None])
        cgb.assignfactors_lg_((sigma2 / (2.0 * alpha)).reshape(ns, 1, 1, 1), mu[:, None], model="ou", alpha=alpha,
                              theta=theta[:, None])
        enqueue_ll, ll_kind = lib.pgbp_enqueue_loglik_lg, 3
    else:
        X = S.simulate_bm_uni_sites(tr, sigma2, mu, rng_x)
        ll_check = S.bm_loglik_pruning_uni_sites(tr, sigma2, mu, X)
        cgb.bm_tree_setup(*S.bm_tree_table(tr, prob), X[:, :, None])
        cgb.assignfactors_bm_(sigma2[:, None, None], mu[:, None])
        enqueue_ll, ll_kind = lib.pgbp_enqueue_loglik_bm, 2
    del X
    eng, opts = cgb._eng, cgb._opts()

    def check(code):
        if code != 0:
            raise RuntimeError(lib.pgbp_last_error(eng).decode())
    norm = np.zeros(ns)
    info = np.zeros(ns, dtype=np.int32)
    check(enqueue_ll(eng, 1, C.byref(opts)))
    check(lib.pgbp_fetch_loglik(eng, L.f64p(norm), L.i32p(info)))
    rel = float(np.max(np.abs(norm - ll_check) / np.maximum(1.0, np.abs(ll_check))))
    if info.any() or rel > 1e-8:
        raise SystemExit(f"parity gate failed (sites): max rel err {rel:.3e}, failures {int(info.astype(bool).sum())}")
    bytes_per_cal, msgs_per_cal = cgb.traffic_model()     # all local problems, one calibrate

    # the one collective of this configuration: all ranks get every problem's log-likelihood.  Behind the C ABI: ONE
    # ncclAllGather carrying every rank's log-likelihoods, info words and (succ, iscal).  A communicator that does not come
    # up raises on every rank (exit code != 0): no silent fallback.
    rehearsal = os.environ.get("PGBP_BENCH_REHEARSAL") == "1"
    comm = make_comm(dist, torch, rank, world, local_rank)
    slot = -(-nprob // world)
    gather_via = ("single rank (pgbp_fetch_loglik)" if world == 1 else
                  "torch.distributed all_gather over gloo (one-GPU rehearsal)" if comm is None else
                  "pgbp_comm (one ncclAllGather behind the C ABI), inside the timed region")

    def gather_all():
        if comm is not None:
            g_norm, g_info, all_succ, _ = comm.gather_loglik(eng, slot)
            if g_info.any() or not all_succ:
                raise SystemExit("sites workload: a rank reported a failed message")
            return np.concatenate([g_norm[r, :shard_range(nprob, r, world)[1] - shard_range(nprob, r, world)[0]] for r in range(world)])
        check(lib.pgbp_fetch_loglik(eng, L.f64p(norm), L.i32p(info)))
        if info.any():
            raise SystemExit("sites workload: a failed message")
        return gather_sites(norm, nprob, dist if world > 1 else None, device="cpu")
    full = None

    def one_sharded_step():
        nonlocal full
        check(lib.pgbp_enqueue_calibrate(eng, 1, 0, C.byref(opts)))
        check(enqueue_ll(eng, 1, C.byref(opts)))
        full = gather_all()
    for _ in range(max(1, args.warmup)):
        one_sharded_step()

    def k_sharded_steps():
        for _ in range(args.steps):
            one_sharded_step()
    dt_sh = timed_region(k_sharded_steps, dist, torch.cuda.synchronize)
    rel_all = float(np.max(np.abs(full[lo:hi] - ll_check) / np.maximum(1.0, np.abs(ll_check))))
    if rel_all > 1e-8:
        raise SystemExit(f"parity gate failed (sites, gathered): max rel err {rel_all:.3e}")
    # calibrate!() alone
    check(lib.pgbp_enqueue_calibrate(eng, args.warmup, 0, C.byref(opts)))

    def k_steps():
        check(lib.pgbp_enqueue_calibrate(eng, args.steps, 0, C.byref(opts)))
        check(lib.pgbp_sync(eng))
    dt = timed_region(k_steps, dist, torch.cuda.synchronize)
    if comm is not None:
        comm.close()
    # log-likelihood evaluations (device factor fill + postorder + root integrate) of all local problems per second
    ms_ll = C.c_float()
    check(lib.pgbp_time_enqueued(eng, ll_kind, 3, 0, C.byref(opts), C.byref(ms_ll)))
    # launches per calibrate (one per level and direction) for the per-launch roofline figures
    ms_k, nl_k = C.c_float(), C.c_int32()
    check(lib.pgbp_time_message_kernels(eng, 1, C.byref(opts), C.byref(ms_k), C.byref(nl_k)))
    n_launches_per_cal = int(nl_k.value)
    sites_traffic = load_pmc_traffic("pmc_traffic_sites_latest.json")
    default_size = (args.sites == 1000 and args.site_traits == 8 and args.ntips == 20000 and ou and args.seed == 4)
    total_msgs = msgs_per_cal / max(1, ns) * nprob        # same per-problem count on every rank
    if rank != 0:
        return None
    ms_step, ms_cal = dt_sh / args.steps * 1e3, dt / args.steps * 1e3
    out = {
        "metric": "clique-tree messages/sec (calibrate!), independent univariate sites sharded across GPUs",
        "value": total_msgs * args.steps / dt_sh, "unit": "messages/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"cfg4: {args.sites} sites x {args.site_traits} traits = {nprob} independent univariate "
                               f"{'OU' if ou else 'BM'} problems, {args.ntips}-tip tree (seed {args.seed}), clique tree, sharded over "
                               f"{world} rank(s); step = calibrate! + score() body (device factor fill, postorder, root "
                               f"integrate) + one all-gather of the per-problem log-likelihoods",
                   "problems_per_rank": ns, "messages_per_problem_per_step": int(msgs_per_cal // max(1, ns))},
        "sharded_step": {"ms_per_step": ms_step, "problem_logliks_per_s": nprob * args.steps / dt_sh,
                         "calibrate_messages_per_s": total_msgs * args.steps / dt_sh, "collective_in_timed_region": world > 1,
                         "loglik_gather": gather_via},
        "calibrate_only": {"ms_per_step": ms_cal, "messages_per_s": total_msgs * args.steps / dt,
                           "note": "calibrate!() alone, K times back to back: the hot path without its exchange step"},
        "roofline": {"bound": "hbm", "achieved": bytes_per_cal / (ms_cal * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": bytes_per_cal / (ms_cal * 1e-3) / 1e9 / 8000.0,
                     # PMC passes of the default-size run on ONE rank (8000 problems, 20 000 tips); not valid for other sizes
                     "traffic": (sites_traffic or {}).get("hbm_bytes_per_launch") if (default_size and world == 1) else None,
                     "traffic_stale": (sites_traffic or {}).get("stale") if (default_size and world == 1) else None,
                     "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes; "
                                     "profiles/pmc_traffic_sites_latest.json)",
                     "algorithmic_bytes_per_launch": bytes_per_cal / max(1, n_launches_per_cal),
                     "kernel": "bp_level_uni1 (level launches) + bp_chunk_uni1 (fused narrow levels)", "note": "rank 0's algorithmic bytes (168 B per univariate message) / wall time of its calibrate (calibrate_only)"},
        "ll_evals_per_s": world * ns * 3 / (ms_ll.value * 1e-3),
        "ll_eval_note": "problem log-likelihoods per second: device factor fill + postorder + root integrate (score() body)",
        "loglik_sum": float(full.sum()), "loglik_max_rel_err_vs_pruning": max(rel, rel_all), "loglik_gather": gather_via}
    if emit:
        print(json.dumps(out))
    return out


def build_network_workload(args, rank):
    """cfg5 (BASELINE.json configs[4]): heterogeneous BM (3 painted rates, 4 traits by default) on a random level-3
    admixture network (20 000 tips, 1 667 level-3 blobs = 5 001 reticulations by default), cluster graph built on the
    host (Bethe: loopy; joingraph: join-graph structuring with --maxclustersize), scopes allocated on plain arrays."""
    import pgbp_amd as P
    rng = np.random.default_rng(args.seed + rank)
    p = args.traits
    if args.blob_style == "template":   # round-1 generator: copies of the reference's own level-3 test blob (treewidth 2)
        net = P.random_level3_network(args.ntips, args.blobs, rng, n_colors=3)
    else:                               # varied blobs of level <= 3, some with 4-node cliques (treewidth 3)
        net = P.random_level3_network_varied(args.ntips, 3 * args.blobs, rng, n_colors=3)
    if args.graph == "joingraph":
        cn, ed, sn = P.joingraph(net.node2family, args.maxclustersize)
    elif args.graph == "cliquetree":
        cn, ed, sn = P.cliquetree(net.node2family)
    elif args.graph == "ltrip":
        cn, ed, sn = P.ltrip(net.node2family)
    else:
        cn, ed, sn = P.bethe(net.node2family)
    st = P.allocate_scopes(cn, ed, sn, net, p)
    base = P.synth.random_rate_matrix(p, rng)
    base = (base + base.T) / 2
    rates = np.stack([base * f for f in (0.5, 1.0, 2.0)])
    mu = np.zeros(p)
    X = P.simulate_bm_network(net, rates, mu, rng)
    parent_edges = [list(zip(net.length[i], net.gamma[i], net.color[i])) for i in range(net.nnodes)]
    fam = P.lg_families(st.clusters, st.node2cluster, net.node2family, st.node2fixed, parent_edges,
                        list(range(net.nnodes)), p, n_rates=3)
    sched = P.spanningtrees_clusterlist(len(cn), ed, cn, net.is_leaf)
    return net, (cn, ed, sn), st, fam, X, rates, mu, sched


def run_network(args, torch, dist, rank, world, local_rank, emit=True):
    """cfg5: loopy belief propagation on a level-3 network.  A step = one calibrate! iteration (every schedule tree,
    postorder + preorder) from the regularised start; N > 1: one independent replica (its own network) per GPU.
    Returns the result dict on rank 0 (None elsewhere); emit: print it as the JSON line."""
    import pgbp_amd as P
    from pgbp_amd import _lib as L
    lib = P.load()
    t0 = time.time()
    # (as a block of the default line every rank runs the SAME replica -- rank 0's network, the one the parity gate and the
    # committed PMC passes are about; the standalone workload gives every rank a network of its own)
    net, (cn, ed, sn), st, fam, X, rates, mu, sched = build_network_workload(args, rank if emit else 0)
    cgb = P.ClusterGraphBelief.from_arrays(st.dims, st.sepset_clusters, st.scope_off, st.scope_idx, None, device=local_rank)
    cgb.lg_setup(fam, X)
    cgb.assignfactors_lg_(rates, mu)                      # assignfactors! on the device (heterogeneous BM, hybrid nodes)
    eng, opts = cgb._eng, cgb._opts()

    def check(code):
        if code != 0:
            raise RuntimeError(lib.pgbp_last_error(eng).decode())
    loopy = len(ed) > len(cn) - 1
    if loopy and args.graph == "joingraph":
        # join graphs: regularizebeliefs_onschedule! (src/clustergraphbeliefs.jl:376-403) -- with the by-cluster or the
        # by-node-subtree regulariser the first postorder over a spanning tree of this graph meets an ill-defined message
        # (the plain-C engine fails at the same message with the same PosDefException.info); one-off host walk, the
        # messages it sends are pgbp_propagate calls
        from pgbp_amd.regularization import regularizebeliefs_onschedule_
        regularizebeliefs_onschedule_(cgb)
    elif loopy:
        check(lib.pgbp_regularize_bycluster(eng))         # regularizebeliefs_bycluster! (src/calibration.jl:335-343)
    cgb.pull()
    t_host = time.time() - t0                             # network, cluster graph, scopes, factors, regularisation
    start = cgb._packed[0].copy()                         # the state every run starts from (clusters and sepsets)
    cgb.set_schedule(sched)
    # calibrate!(beliefs, sched, 100; auto=true): iterations to convergence, end to end with the host-driven auto stop
    cgb.init_messagecalibrationflags_reset_()
    check(lib.pgbp_sync(eng))
    t = time.perf_counter()
    res = P.calibrate_(cgb, sched, 100, auto=True, sync=False)
    t_auto = time.perf_counter() - t
    r = cgb.last_results[0]
    if res != (True, True):
        raise SystemExit(f"network workload: calibrate!(auto) returned {res}")
    nmsg_tree = [2 * len(s[2]) for s in sched]
    n_pairs = (r.iter_reached - 1) * len(sched) + r.tree_reached
    nmsg_auto = sum(nmsg_tree[q % len(sched)] for q in range(n_pairs))
    cgb.pull()
    converged = cgb._packed[0].copy()
    bytes_per_cal, msgs_per_cal = cgb.traffic_model()

    def restart():
        cgb._packed[0][:] = start
        cgb.push()
        cgb.init_messagecalibrationflags_reset_()
    restart()
    check(lib.pgbp_enqueue_calibrate(eng, args.warmup, 0, C.byref(opts)))
    restart()
    check(lib.pgbp_sync(eng))

    def k_steps():
        check(lib.pgbp_enqueue_calibrate(eng, args.steps, 0, C.byref(opts)))
        check(lib.pgbp_sync(eng))
    dt = timed_region(k_steps, dist, torch.cuda.synchronize)
    if rank != 0:
        return None
    ms_step = dt / args.steps * 1e3
    net_traffic = load_pmc_traffic(f"pmc_traffic_network_{args.graph}_latest.json")
    default_net = (args.ntips == 20000 and args.traits == 4 and args.seed == 5 and args.blob_style == "varied"
                   and args.graph in ("joingraph", "bethe") and args.maxclustersize == 3 and args.blobs == (20000 + 11) // 12)
    out = {
        "metric": "cluster-graph messages/sec (calibrate!: every schedule tree, postorder+preorder), loopy BP on a level-3 network",
        "value": whole_job_rate(msgs_per_cal, args.steps, world, dt), "unit": "messages/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"cfg5: heterogeneous BM (3 rates), {args.traits} traits, level-3 network with {args.ntips} tips "
                               f"and {net.nhybrids} reticulations ({net.nnodes} nodes), "
                               + (f"join-graph structuring (maxclustersize {args.maxclustersize})" if args.graph == "joingraph"
                                  else "clique tree" if args.graph == "cliquetree" else "LTRIP cluster graph (node families)"
                                  if args.graph == "ltrip" else "Bethe cluster graph")
                               + ((", regularizebeliefs_onschedule!" if args.graph == "joingraph" else ", regularizebeliefs_bycluster!")
                                  if loopy else " (a tree here)") + ", spanningtrees_clusterlist schedule",
                   "clusters": len(cn), "sepsets": len(ed), "schedule_trees": len(sched), "loopy": bool(loopy),
                   "messages_per_step": int(msgs_per_cal), "max_cluster_dimension": int(st.dims.max()),
                   "parallelism": "single GPU" if world == 1 else f"{world} independent replicas"},
        "calibrate_auto": {"iterations_to_convergence": [int(r.iter_reached), int(r.tree_reached)], "messages": int(nmsg_auto),
                           "ms": 1e3 * t_auto, "messages_per_s": nmsg_auto / t_auto,
                           "note": "calibrate!(beliefs, sched, 100; auto=true) end to end: the device halts itself at the first calibrated tree, one host round trip per 4 schedule trees"},
        "roofline": {"bound": "hbm", "achieved": bytes_per_cal / (ms_step * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": bytes_per_cal / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     # HBM bytes of ONE step (calibrate iteration: every launch of it) from the committed PMC passes of the
                     # default-size run (tools/pmc_network.sh); None for other sizes / graphs
                     "traffic": (net_traffic or {}).get("hbm_bytes_per_calibrate") if default_net else None,
                     "traffic_stale": (net_traffic or {}).get("stale") if default_net else None,
                     "traffic_unit": "bytes per step (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes, every message-kernel "
                                     f"launch of one calibrate iteration; profiles/pmc_traffic_network_{args.graph}_latest.json)",
                     "kernel": "bp_level_small4 (wide levels: four rows of 16 lanes per wavefront) + bp_level_generic + bp_chunk_pair (fused narrow levels: a provider and a consumer wavefront per task)", "algorithmic_bytes_per_step": bytes_per_cal,
                     "note": "algorithmic bytes of one calibrate iteration / its wall time; launch-latency-bound (levels per tree >> width)"},
        "host_setup_s": t_host,
    }
    if not args.no_cpu_baseline and world == 1:
        try:
            from oracle import cengine
            cengine.use_native_build()
            ce = cengine.Engine(st.dims, st.sepset_clusters.reshape(-1), st.scope_off, st.scope_idx, start)
            t = time.perf_counter()
            reached = None
            for it in range(1, 101):
                for j, spt in enumerate(sched, start=1):
                    succ, iscal = ce.calibrate(spt[2], spt[3], 1, return_iscal=True)
                    if iscal:
                        reached = (it, j)
                        break
                if reached:
                    break
            t_cpu = time.perf_counter() - t
            ref = ce.packed()
            # per belief: max |difference| / max(1, max |reference|) (the tolerance of tests/test_gpu_parity.py)
            o = cgb._poff[:-1][np.diff(cgb._poff) > 0]
            err = float(np.max(np.maximum.reduceat(np.abs(converged - ref), o) /
                               np.maximum(1.0, np.maximum.reduceat(np.abs(ref), o))))
            out["cpu_baseline"] = {"value": nmsg_auto / t_cpu, "unit": "messages/s", "cores": 1, "kind": "port",
                                   "sample": f"the same calibrate!(auto) to convergence ({nmsg_auto} messages) on the oracle/c "
                                             f"sequential engine (gcc -O3 -march=native, 1 thread of {os.cpu_count()} host cpus), {t_cpu:.1f} s",
                                   "iterations_to_convergence": list(reached) if reached else None,
                                   "max_rel_belief_diff_gpu_vs_cpu": err}
            if reached != (r.iter_reached, r.tree_reached) or not err <= 1e-8:
                raise SystemExit(f"parity gate failed (network): cpu reached {reached}, gpu {(r.iter_reached, r.tree_reached)}, belief diff {err:.3e}")
        except ImportError as ex:
            out["cpu_baseline"] = {"value": None, "unit": "messages/s", "cores": 1, "kind": "port", "sample": f"unavailable: {ex}"}
    if emit:
        print(json.dumps(out))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # (cfg3: 87 ms of timed region; 20 steps = 17 ms were as noisy as the round's micro-gains)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ntips", type=int, default=None, help="default: 50000 (tree workload), 20000 (sites workload)")
    ap.add_argument("--traits", type=int, default=None, help="default: 16 (tree workload), 4 (network workload)")
    ap.add_argument("--graph", default=None, choices=["cliquetree", "bethe", "joingraph", "ltrip"],
                    help="default: cliquetree (tree workload), joingraph (network workload)")
    ap.add_argument("--blobs", type=int, default=None, help="network workload: level-3 blobs (3 reticulations each); default ntips / 12")
    ap.add_argument("--maxclustersize", type=int, default=3, help="network workload, --graph joingraph (a level-3 network's "
                    "moral graph has cliques of at most 4 nodes: 3 is the largest bound under which the join graph is loopy)")
    ap.add_argument("--blob-style", default="varied", choices=["varied", "template"], help="network workload: random blobs "
                    "of level <= 3 (default) or copies of the reference's level-3 test blob")
    ap.add_argument("--seed", type=int, default=None, help="default: SURVEY.md section 8(d)'s recorded seeds -- 3 (tree workload: cfg3), "
                    "4 (sites workload: cfg4), 5 (network workload: cfg5)")
    ap.add_argument("--no-sites-block", action="store_true",
                    help="tree workload at the cfg3 size: skip the site_sharded_cfg4 block (the site-sharded configuration, strong scaling, "
                         "with its all-gather inside the timed region) that the default line carries for every --gpus N")
    ap.add_argument("--no-network-block", action="store_true",
                    help="tree workload at the cfg3 size: skip the network_cfg5 block (loopy BP on the level-3 network's join graph)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-reading", action="store_true",
                    help="skip the 25001-tip (50k-clique) side measurement: profiler runs want one workload per kernel name")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--ll-batch", type=int, default=8, help="tree workload: parameter sets per pass of the batched log-likelihood figure (1: skip)")
    ap.add_argument("--site-model", default="ou", choices=["ou", "bm"], help="sites workload: per-problem model")
    ap.add_argument("--site-traits", type=int, default=8, help="sites workload: traits per site (each an independent univariate problem)")
    ap.add_argument("--workload", default="tree", choices=["tree", "sites", "network"],
                    help="tree: the headline one-big-tree workload (default, cfg3 / cfg2); sites: cfg4 site-sharded batch; "
                         "network: cfg5 loopy BP on a level-3 network")
    ap.add_argument("--sites", type=int, default=1000)
    args = ap.parse_args()
    if args.ntips is None:
        args.ntips = 50000 if args.workload == "tree" else 20000
    if args.traits is None:
        args.traits = 4 if args.workload == "network" else 16
    if args.graph is None:
        args.graph = "joingraph" if args.workload == "network" else "cliquetree"
    if args.blobs is None:
        args.blobs = (args.ntips + 11) // 12
    if args.seed is None:
        args.seed = {"tree": 3, "sites": 4, "network": 5}[args.workload]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the launcher as a CHILD process (this process has not
        # touched the GPU yet; never exec over a process that has) and hand its exit code back
        import subprocess
        port = 29500 + os.getpid() % 2000
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 code on a one-GPU box: every rank on device 0, gloo instead of RCCL (two RCCL ranks cannot
    # share a device); never set by the driver
    rehearsal = os.environ.get("PGBP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: a figure for another number of GPUs than asked for "
                         "is never printed")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    if args.workload == "network":
        run_network(args, torch, dist, rank, world, local_rank)
        if dist is not None:
            dist.destroy_process_group()
        return
    if args.workload == "sites":
        run_sites(args, torch, dist, rank, world, local_rank)
        if dist is not None:
            dist.destroy_process_group()
        return

    import pgbp_amd
    from pgbp_amd import _lib as L
    lib = pgbp_amd.load()

    tr, prob, packed, ll_check, (R, mu, X) = build_workload(args.ntips, args.traits, args.seed + rank, args.graph)
    cgb = pgbp_amd.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off,
                                                  prob.scope_idx, packed, device=local_rank)
    cgb.set_schedule(prob.schedule)
    eng = cgb._eng
    opts = cgb._opts()
    bytes_per_cal, msgs_per_cal = cgb.traffic_model()

    def check(code):
        if code != 0:
            raise RuntimeError(lib.pgbp_last_error(eng).decode())

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- parity gate for the timed configuration: log-likelihood vs the independent pruning check
    norm = np.zeros(1)
    info = np.zeros(1, dtype=np.int32)
    check(lib.pgbp_enqueue_loglik(eng, 1, C.byref(opts)))
    check(lib.pgbp_fetch_loglik(eng, L.f64p(norm), L.i32p(info)))
    rel = abs(norm[0] - ll_check) / max(1.0, abs(ll_check))
    if not (info[0] == 0 and rel <= 1e-8):
        raise SystemExit(f"parity gate failed: loglik {norm[0]!r} vs {ll_check!r} (rel {rel:.3e}, info {info[0]})")

    # ---- warmup + timed calibrate steps
    check(lib.pgbp_reset_from_factors(eng))
    check(lib.pgbp_enqueue_calibrate(eng, args.warmup, 0, C.byref(opts)))

    def k_steps():
        # one HIP event pair per calibrate around its message launches, INSIDE the timed steps: the kernel time below is
        # a part of `dt` by construction
        check(lib.pgbp_enqueue_calibrate_timed(eng, args.steps, 0, C.byref(opts)))
        check(lib.pgbp_sync(eng))
    dt = timed_region(k_steps, dist, torch.cuda.synchronize)
    value = whole_job_rate(msgs_per_cal, args.steps, world, dt)
    ms_k, nl_k = C.c_float(), C.c_int32()
    check(lib.pgbp_fetch_kernel_time(eng, C.byref(ms_k), C.byref(nl_k)))

    # after the timed region: no message failed, and the calibrated beliefs still integrate to the right log-likelihood
    mu_, n2, i2 = cgb.integratebelief_(prob.root_cluster, all_sites=True)
    rel2 = abs(n2[0] - ll_check) / max(1.0, abs(ll_check))
    if not (i2[0] == 0 and rel2 <= 1e-8):
        raise SystemExit(f"post-run parity failed: {n2[0]!r} vs {ll_check!r}")
    ranks_loglik, ranks_gather = None, None
    if world > 1 and dist is not None and os.environ.get("PGBP_BENCH_REHEARSAL") != "1":
        # N > 1 (replicas): every rank's log-likelihood and success flag reach every rank in ONE ncclAllGather behind the
        # C ABI (pgbp_comm).  A communicator that does not come up raises on every rank: the run exits non-zero.
        comm = make_comm(dist, torch, rank, world, local_rank)
        g_norm, g_info, all_succ, _ = comm.gather_loglik(eng, 1)
        comm.close()
        ranks_gather = "pgbp_comm (one ncclAllGather behind the C ABI)"
        if g_info.any() or not all_succ:
            raise SystemExit("a rank reported a failed message after the timed region")
        ranks_loglik = [float(v) for v in g_norm[:, 0]]

    out = None
    if rank == 0:
        # ---- secondary metric: log-likelihood evaluations / s = the body of score(theta)
        # (src/calibration.jl:195-221): assignfactors! ON THE DEVICE from (R^-1, log det R, mu), postorder
        # traversal, root integratebelief!; only the parameters and the result cross the bus.
        from pgbp_amd import synth as S
        ms = C.c_float()
        nll = max(3, args.steps // 2)
        cgb.bm_tree_setup(*S.bm_tree_table(tr, prob), X)
        cgb.assignfactors_bm_(R, mu)
        check(lib.pgbp_time_enqueued(eng, 2, nll, 1, C.byref(opts), C.byref(ms)))
        ll_evals = nll / (ms.value * 1e-3)
        check(lib.pgbp_fetch_loglik(eng, L.f64p(norm), L.i32p(info)))
        rel3 = abs(norm[0] - ll_check) / max(1.0, abs(ll_check))
        if not (info[0] == 0 and rel3 <= 1e-8):
            raise SystemExit(f"device-fill loglik parity failed: {norm[0]!r} vs {ll_check!r}")
        # same without the fill (factors copied from the resident factor pool)
        check(lib.pgbp_time_enqueued(eng, 1, nll, 1, C.byref(opts), C.byref(ms)))
        ll_evals_nofill = nll / (ms.value * 1e-3)
        # roofline of one evaluation (the other half of BASELINE.json's metric): what the score() body has to move --
        # every cluster record written by the fill, the algorithmic bytes of the postorder's messages (SURVEY.md section
        # 8(d): read the sender, read + write the sepset, read + write the receiver's block, write the residual) and the
        # root belief read by integratebelief! -- over the time of one evaluation
        dims64 = np.asarray(prob.dims, np.int64)
        nc_ = int(prob.nclusters)
        pa_, ch_ = (np.asarray(x, np.int64) for x in prob.schedule[0][-2:])
        sep_of = {}
        sc_ = np.asarray(prob.sepset_clusters, np.int64).reshape(-1, 2)
        for k, (a_, b_) in enumerate(sc_):
            sep_of[(int(a_), int(b_))] = k
            sep_of[(int(b_), int(a_))] = k
        s_dim = np.array([dims64[nc_ + sep_of[(int(a_), int(b_))]] for a_, b_ in zip(pa_, ch_)], np.int64)
        mf_post = dims64[ch_]
        post_bytes = float(8 * np.sum((mf_post * mf_post + mf_post + 1) + 4 * (s_dim * s_dim + s_dim + 1) + (s_dim * s_dim + s_dim)))
        fill_bytes = float(8 * np.sum(dims64[:nc_] * dims64[:nc_] + dims64[:nc_] + 1))
        mroot = int(dims64[prob.root_cluster])
        ll_bytes = fill_bytes + post_bytes + 8.0 * (mroot * mroot + mroot + 1)
        ll_ms = 1e3 / ll_evals
        ll_roofline = {"bound": "hbm", "achieved": ll_bytes / (ll_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": ll_bytes / (ll_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms_per_eval": ll_ms,
                       "algorithmic_bytes_per_eval": ll_bytes,
                       "bytes": {"factor_fill_written": fill_bytes, "postorder_messages": post_bytes,
                                 "root_integrate_read": 8.0 * (mroot * mroot + mroot + 1)},
                       "kernels": "bm_tree_fill_fast (assignfactors! on the device, straight into the packed layout the "
                                  "postorder reads) + bp_fast16 / bp_loop16 (postorder) + integrate_kernel (root); kernel rows: "
                                  "profiles/r04_*_cfg3_kernel_stats.csv",
                       "traffic": None}
        ll_pmc = load_pmc_traffic("pmc_traffic_ll_eval_latest.json")
        if ll_pmc and (args.traits, args.ntips, args.graph, args.seed) == (16, 50000, "cliquetree", 3):
            ll_roofline.update({"traffic": ll_pmc.get("hbm_bytes_per_calibrate"), "traffic_stale": ll_pmc.get("stale"),
                                "traffic_unit": "bytes per evaluation (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes, every "
                                                "launch of the fill, the flag reset, the postorder and the root integrate of one "
                                                "evaluation: tools/level_times.py run-ll; profiles/pmc_traffic_ll_eval_latest.json)"})
        # ---- the DROP-IN call: what a user of the kept-intact API gets (calibrate_ with its write-back, then
        # integratebelief! at the root), end to end on the host clock, beside the upload of the 0.76 GB belief state and
        # the eager pull of everything (the round-3 behaviour of every call)
        import pgbp_amd as _P

        def _wall(fn, reps=3):
            best = None
            for _ in range(reps):
                check(lib.pgbp_sync(eng))
                t0 = time.perf_counter()
                fn()
                check(lib.pgbp_sync(eng))
                dtw = time.perf_counter() - t0
                best = dtw if best is None else min(best, dtw)
            return best * 1e3

        def _lazy_call():
            assert _P.calibrate_(cgb, prob.schedule, 1)[0]
            cgb.integratebelief_(prob.root_cluster)

        def _eager_call():
            assert _P.calibrate_(cgb, prob.schedule, 1, sync=False)[0]
            cgb.pull()
        cgb.pull()
        dropin = {"calibrate_plus_root_integrate_ms": _wall(_lazy_call),
                  "calibrate_plus_full_pull_ms": _wall(_eager_call, 2),
                  "set_beliefs_upload_ms": _wall(cgb.push, 2),
                  "state_bytes": float(8 * cgb._poff[-1]),
                  "note": "host wall time of the reference-named calls (pgbp_amd.calibrate_ = calibrate!, integratebelief_): the "
                          "write-back is lazy -- the call moves the result struct, a belief's record crosses the bus when it is "
                          "first read (pgbp_get_belief), a residual's likewise (pgbp_get_residual); the full pull and the upload "
                          "are the PCIe-inclusive figures, never `value`"}
        # ---- roofline of the dominant kernel: the HIP events recorded inside the timed steps above
        nl = nl_k
        reps = args.steps
        kern_ms = ms_k.value
        pmc = load_pmc_traffic()
        achieved = bytes_per_cal * reps / (kern_ms * 1e-3) / 1e9
        traffic = (pmc or {}).get("hbm_bytes_per_launch")
        launches_pmc = (pmc or {}).get("launches_per_calibrate") or 76   # (76: the round-1 file, one launch per level)
        copy_bw = measured_copy_bandwidth(torch, local_rank) if rank == 0 else None
        out = {
            "metric": "clique-tree messages/sec (calibrate!: postorder+preorder), 16-trait BM",
            "value": value, "unit": "messages/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{'cfg3' if (args.traits, args.ntips, args.graph) == (16, 50000, 'cliquetree') else 'custom'}: homogeneous BM, {args.traits} traits, {args.ntips}-tip random "
                                   f"bifurcating tree (seed {args.seed}), {args.graph}, fixed root",
                       "clusters": int(prob.nclusters), "sepsets": int(len(prob.dims) - prob.nclusters),
                       "messages_per_step": int(msgs_per_cal), "tree_depth": int(tr.depth().max()),
                       "parallelism": "replicas only" if world > 1 else "single GPU"},
            "loglik": float(norm[0]), "loglik_rel_err_vs_pruning": float(rel),
            "ranks_loglik": ranks_loglik, "ranks_gather": ranks_gather,
            "ll_evals_per_s": ll_evals, "ll_evals_per_s_without_factor_fill": ll_evals_nofill,
            "ll_eval": {"value": ll_evals, "unit": "log-likelihood evaluations/s", "roofline": ll_roofline},
            "dropin": dropin,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # the same with the HBM bytes the counters saw (the packed layout moves about half of the
                         # algorithmic bytes): what fraction of the peak the chip actually streamed
                         "frac_physical": (traffic * launches_pmc * reps / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "traffic": traffic,
                         # true: profiles/pmc_traffic_latest.json was reduced from another state of csrc/ than the one running
                         "traffic_stale": (pmc or {}).get("stale"),
                         "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes; "
                                         "profiles/pmc_traffic_latest.json)",
                         "algorithmic_bytes_per_launch": bytes_per_cal / max(1, nl.value // reps),
                         "kernel": "bp_fast16 (level launches) + bp_loop16 (tail, chunks of fused levels)",
                         "launches_per_step": nl.value // reps,
                         "algorithmic_bytes_per_step": bytes_per_cal,
                         "kernel_ms_per_step": kern_ms / reps,
                         # context for `frac`: what a plain device-to-device copy of 1 GiB reaches on this box
                         # (bytes read + bytes written per second; SURVEY.md section 8(d) asks for it beside the peak)
                         "measured_copy_GBps": copy_bw},
        }
        if world == 1 and not args.no_alt_reading and (args.traits, args.ntips, args.graph) == (16, 50000, "cliquetree"):
            # BASELINE.json says "50k-clique" in `metric` and "50k-tip tree" in `configs[2]`: the
            # headline above is the 50k-tip tree (99 998 cliques); this is the other reading, a 25 001-tip tree
            # = exactly 50 000 cliques, same recipe, same parity gate, so that either is covered.
            tr2, prob2, packed2, ll2, _ = build_workload(25001, 16, args.seed, "cliquetree")
            cgb2 = pgbp_amd.ClusterGraphBelief.from_arrays(prob2.dims, prob2.sepset_clusters, prob2.scope_off,
                                                           prob2.scope_idx, packed2, device=local_rank)
            cgb2.set_schedule(prob2.schedule)
            _, m2 = cgb2.traffic_model()
            check2 = lambda code: code == 0 or (_ for _ in ()).throw(RuntimeError(lib.pgbp_last_error(cgb2._eng).decode()))
            check2(lib.pgbp_enqueue_loglik(cgb2._eng, 1, C.byref(opts)))
            check2(lib.pgbp_fetch_loglik(cgb2._eng, L.f64p(norm), L.i32p(info)))
            rel4 = abs(norm[0] - ll2) / max(1.0, abs(ll2))
            if not (info[0] == 0 and rel4 <= 1e-8):
                raise SystemExit(f"parity gate failed (50k-clique reading): {norm[0]!r} vs {ll2!r}")
            check2(lib.pgbp_reset_from_factors(cgb2._eng))
            check2(lib.pgbp_time_enqueued(cgb2._eng, 0, args.warmup, 0, C.byref(opts), C.byref(ms)))
            check2(lib.pgbp_time_enqueued(cgb2._eng, 0, args.steps, 0, C.byref(opts), C.byref(ms)))
            out["alt_reading_50k_cliques"] = {
                "workload": "25001-tip tree = 50000 cliques, 16 traits, clique tree", "messages_per_step": int(m2),
                "ms_per_step": ms.value / args.steps, "messages_per_s": m2 * args.steps / (ms.value * 1e-3),
                "loglik_rel_err_vs_pruning": float(rel4)}
            del cgb2
        if world == 1 and not args.no_alt_reading and (args.traits, args.ntips, args.graph) == (16, 50000, "cliquetree"):
            # BASELINE.json configs[1] (cfg2) as a secondary block: homogeneous BM, 8 traits, 10 000-tip tree, Bethe graph
            tr2, prob2, packed2, ll2, _ = build_workload(10000, 8, 2, "bethe")
            cgb2 = pgbp_amd.ClusterGraphBelief.from_arrays(prob2.dims, prob2.sepset_clusters, prob2.scope_off,
                                                           prob2.scope_idx, packed2, device=local_rank)
            cgb2.set_schedule(prob2.schedule)
            b2, m2 = cgb2.traffic_model()
            check2 = lambda code: code == 0 or (_ for _ in ()).throw(RuntimeError(lib.pgbp_last_error(cgb2._eng).decode()))
            check2(lib.pgbp_enqueue_loglik(cgb2._eng, 1, C.byref(opts)))
            check2(lib.pgbp_fetch_loglik(cgb2._eng, L.f64p(norm), L.i32p(info)))
            rel5 = abs(norm[0] - ll2) / max(1.0, abs(ll2))
            if not (info[0] == 0 and rel5 <= 1e-8):
                raise SystemExit(f"parity gate failed (cfg2): {norm[0]!r} vs {ll2!r}")
            check2(lib.pgbp_reset_from_factors(cgb2._eng))
            check2(lib.pgbp_time_enqueued(cgb2._eng, 0, args.warmup, 0, C.byref(opts), C.byref(ms)))
            check2(lib.pgbp_time_enqueued(cgb2._eng, 0, args.steps, 0, C.byref(opts), C.byref(ms)))
            out["cfg2_bethe_10k_tips_8_traits"] = {
                "workload": "cfg2: homogeneous BM, 8 traits, 10000-tip tree (seed 2), Bethe cluster graph",
                "clusters": int(prob2.nclusters), "messages_per_step": int(m2), "ms_per_step": ms.value / args.steps,
                "messages_per_s": m2 * args.steps / (ms.value * 1e-3),
                "roofline_frac": b2 * args.steps / (ms.value * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "loglik_rel_err_vs_pruning": float(rel5)}
            del cgb2
        if world == 1 and args.ll_batch > 1:
            # several parameter sets per pass: the site dimension of one engine carries B candidate models (R, mu) over
            # the same data -- what an optimiser's finite-difference gradient or a multi-start issues; the narrow levels
            # of the schedule are shared by the B evaluations
            B = args.ll_batch
            cgbB = pgbp_amd.ClusterGraphBelief.from_arrays(prob.dims, prob.sepset_clusters, prob.scope_off, prob.scope_idx,
                                                           None, n_sites=B, device=local_rank)
            cgbB.set_schedule(prob.schedule)
            cgbB.bm_tree_setup(*S.bm_tree_table(tr, prob), np.broadcast_to(X, (B,) + X.shape).copy())
            Rs = np.stack([R * (1.0 + 0.05 * b) for b in range(B)])
            cgbB.assignfactors_bm_(Rs, np.broadcast_to(mu, (B, len(mu))).copy())
            checkB = lambda code: code == 0 or (_ for _ in ()).throw(RuntimeError(lib.pgbp_last_error(cgbB._eng).decode()))
            checkB(lib.pgbp_time_enqueued(cgbB._eng, 2, 2, 1, C.byref(opts), C.byref(ms)))
            checkB(lib.pgbp_time_enqueued(cgbB._eng, 2, nll, 1, C.byref(opts), C.byref(ms)))
            normB, infoB = np.zeros(B), np.zeros(B, np.int32)
            checkB(lib.pgbp_fetch_loglik(cgbB._eng, L.f64p(normB), L.i32p(infoB)))
            refB = S.bm_loglik_pruning(tr, Rs[B - 1], mu, X)
            relB = abs(normB[B - 1] - refB) / max(1.0, abs(refB))
            if not (not infoB.any() and relB <= 1e-8 and abs(normB[0] - ll_check) <= 1e-8 * abs(ll_check)):
                raise SystemExit(f"batched loglik parity failed: {normB[B - 1]!r} vs {refB!r}")
            out["ll_evals_per_s_batched"] = {"parameter_sets_per_pass": B, "value": B * nll / (ms.value * 1e-3),
                                             "ms_per_pass": ms.value / nll, "loglik_rel_err_vs_pruning": float(relB)}
            del cgbB
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(prob, packed[0] if packed.ndim > 1 else packed, args.cpu_budget)
            if out["cpu_baseline"].get("value"):
                out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
    barrier()
    if not args.no_sites_block and (args.traits, args.ntips, args.graph) == (16, 50000, "cliquetree"):
        # BASELINE.json configs[3] (cfg4), the configuration north_star's ">= 6x at 8 GPUs" is quoted on, beside the
        # headline for EVERY N: 8000 univariate OU problems sharded over the N ranks (strong scaling), step = calibrate! +
        # score() body + the all-gather, the collective inside the timed region (run_sites)
        cgb = None
        sargs = argparse.Namespace(**vars(args))
        sargs.sites, sargs.site_traits, sargs.ntips, sargs.seed, sargs.site_model = 1000, 8, 20000, 4, "ou"
        blk = run_sites(sargs, torch, dist, rank, world, local_rank, emit=False)
        if rank == 0:
            out["site_sharded_cfg4"] = {k: blk[k] for k in ("value", "unit", "n_gpus", "ms_per_step", "scaling", "config",
                                                            "sharded_step", "calibrate_only", "roofline",
                                                            "loglik_max_rel_err_vs_pruning")}
        barrier()
    if not args.no_network_block and (args.traits, args.ntips, args.graph) == (16, 50000, "cliquetree"):
        # BASELINE.json configs[4] (cfg5) beside the headline: loopy BP on the level-3 network's join graph -- one replica
        # per rank (a cluster graph does not shard without an exchange step: DESIGN.md section 6), its cpu_baseline to
        # convergence on one rank
        nargs = argparse.Namespace(**vars(args))
        nargs.ntips, nargs.traits, nargs.seed, nargs.graph, nargs.maxclustersize = 20000, 4, 5, "joingraph", 3
        nargs.blobs, nargs.blob_style = (20000 + 11) // 12, "varied"
        nblk = run_network(nargs, torch, dist, rank, world, local_rank, emit=False)
        if rank == 0:
            out["network_cfg5"] = {k: nblk[k] for k in ("value", "unit", "n_gpus", "ms_per_step", "scaling", "config", "calibrate_auto",
                                                         "roofline", "host_setup_s", "cpu_baseline") if k in nblk}
        barrier()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
