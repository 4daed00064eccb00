"""ctypes binding of include/pgbp.h (libpgbp.so, built in-tree by csrc/Makefile).

There is no CPU fallback: if the library is missing, loading raises; if no GPU is
present, `pgbp_create` returns PGBP_ERR_NO_DEVICE and the host layer raises."""
import ctypes as C
import os

import numpy as np

# Kernel arguments in device memory: the HIP runtime reads this once when it initialises.  It is the default of this
# ROCm image; where it is off, every level launch of a narrow level costs about 1 us more (cfg3 1.04 -> 1.12 ms per
# calibrate, cfg5 3.6 -> 4.3 ms per iteration).  A value set by the user wins.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PGBP_LIB", os.path.join(_HERE, "csrc", "libpgbp.so"))

PGBP_OK, ERR_INVALID, ERR_HIP, ERR_NOT_TREE, ERR_TOO_LARGE, ERR_NO_DEVICE, ERR_STATE = range(7)
PGBP_MAX_DIM = 384


class PgbpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pgbp error {code}: {msg}")
        self.code = code
        self.msg = msg


class Desc(C.Structure):
    _fields_ = [("n_clusters", C.c_int32), ("n_sepsets", C.c_int32),
                ("dims", C.POINTER(C.c_int32)), ("sepset_clusters", C.POINTER(C.c_int32)),
                ("scope_off", C.POINTER(C.c_int64)), ("scope_idx", C.POINTER(C.c_int32)),
                ("n_sites", C.c_int32), ("device", C.c_int32)]


class BmTree(C.Structure):
    _fields_ = [("p", C.c_int32), ("n_rows", C.c_int32), ("kind", C.POINTER(C.c_int32)),
                ("length", C.POINTER(C.c_double)), ("data_row", C.POINTER(C.c_int32)),
                ("data", C.POINTER(C.c_double))]


class LgFamilies(C.Structure):
    _fields_ = [("p", C.c_int32), ("n_families", C.c_int32), ("max_parents", C.c_int32), ("n_rates", C.c_int32),
                ("n_rows", C.c_int32), ("cluster", C.POINTER(C.c_int32)), ("n_parents", C.POINTER(C.c_int32)),
                ("child_pos", C.POINTER(C.c_int32)), ("data_row", C.POINTER(C.c_int32)),
                ("parent_pos", C.POINTER(C.c_int32)), ("length", C.POINTER(C.c_double)),
                ("gamma", C.POINTER(C.c_double)), ("color", C.POINTER(C.c_int32)), ("data", C.POINTER(C.c_double)),
                ("child_mask", C.POINTER(C.c_uint64)), ("parent_mask", C.POINTER(C.c_uint64))]


class LgParams(C.Structure):
    _fields_ = [("model", C.c_int32), ("per_site", C.c_int32), ("R", C.POINTER(C.c_double)),
                ("alpha", C.POINTER(C.c_double)), ("theta", C.POINTER(C.c_double)), ("mu", C.POINTER(C.c_double))]


LG_BM, LG_OU = 0, 1


class Opts(C.Structure):
    _fields_ = [("auto_stop", C.c_int32), ("update_residualnorm", C.c_int32),
                ("update_residualkldiv", C.c_int32), ("reserved", C.c_int32), ("atol", C.c_double)]


class Result(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("succ", "iscal", "iter_reached", "tree_reached", "fail_iter",
                                         "fail_tree", "fail_dir", "fail_edge", "fail_info", "reserved")]


# every symbol include/pgbp.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_I32P = C.POINTER(C.c_int32)
_I64P = C.POINTER(C.c_int64)
_F64P = C.POINTER(C.c_double)
SYMBOLS = {
    "pgbp_plan_create": (C.c_int, [C.POINTER(Desc), C.POINTER(_P)]),
    "pgbp_plan_destroy": (None, [_P]),
    "pgbp_plan_set_schedule": (C.c_int, [_P, C.c_int32, _I32P, _I32P, _I32P]),
    "pgbp_plan_packed_size": (C.c_int64, [_P]),
    "pgbp_plan_residual_size": (C.c_int64, [_P]),
    "pgbp_plan_n_messages": (C.c_int32, [_P]),
    "pgbp_plan_traversal_sizes": (C.c_int, [_P, C.c_int32, C.c_int32, _I32P, _I32P, _I32P]),
    "pgbp_plan_traversal": (C.c_int, [_P, C.c_int32, C.c_int32, _I32P, _I32P, _I32P, _I32P, _I32P]),
    "pgbp_plan_level_nfast": (C.c_int, [_P, C.c_int32, C.c_int32, _I32P]),
    "pgbp_plan_groups": (C.c_int, [_P, C.c_int32, C.c_int32, _I32P, _I32P, _I32P, _I32P]),
    "pgbp_plan_chunks": (C.c_int, [_P, C.c_int32, C.c_int32, _I32P, _I32P, _I32P, _I32P]),
    "pgbp_plan_prologues": (C.c_int, [_P, C.c_int32, C.c_int32, _I32P, _I32P, _I32P]),
    "pgbp_plan_chains": (C.c_int, [_P, C.c_int32, C.c_int32, _I32P, _I32P]),
    "pgbp_plan_records": (C.c_int, [_P, C.c_int32, C.c_int32, _I32P, _I32P, _I32P, C.c_void_p]),
    "pgbp_plan_rows": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _I32P, _I32P]),
    "pgbp_residual_threshold": (C.c_double, [C.c_double, C.c_double]),
    "pgbp_plan_last_error": (C.c_char_p, [_P]),
    "pgbp_create": (C.c_int, [C.POINTER(Desc), C.POINTER(_P)]),
    "pgbp_destroy": (None, [_P]),
    "pgbp_last_error": (C.c_char_p, [_P]),
    "pgbp_packed_size": (C.c_int64, [_P]),
    "pgbp_residual_size": (C.c_int64, [_P]),
    "pgbp_n_messages": (C.c_int32, [_P]),
    "pgbp_belief_dim": (C.c_int32, [_P, C.c_int32]),
    "pgbp_set_beliefs": (C.c_int, [_P, _F64P, C.c_int32]),
    "pgbp_get_beliefs": (C.c_int, [_P, _F64P]),
    "pgbp_get_site_beliefs": (C.c_int, [_P, C.c_int32, _F64P]),
    "pgbp_set_belief": (C.c_int, [_P, C.c_int32, C.c_int32, _F64P]),
    "pgbp_get_belief": (C.c_int, [_P, C.c_int32, C.c_int32, _F64P]),
    "pgbp_packed_beliefs_size": (C.c_int64, [_P, C.c_int32, _I32P]),
    "pgbp_pack_beliefs": (C.c_int, [_P, C.c_int32, C.c_int32, _I32P, _F64P]),
    "pgbp_unpack_beliefs": (C.c_int, [_P, C.c_int32, C.c_int32, _I32P, _F64P]),
    "pgbp_init_factors_frombeliefs": (C.c_int, [_P]),
    "pgbp_reset_from_factors": (C.c_int, [_P]),
    "pgbp_reset_flags": (C.c_int, [_P, C.c_int32]),
    "pgbp_get_residuals": (C.c_int, [_P, _F64P, _I32P, _F64P, _I32P]),
    "pgbp_get_residual": (C.c_int, [_P, C.c_int32, C.c_int32, _F64P, _I32P, _F64P, _I32P]),
    "pgbp_residual_kldiv": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Opts), _I32P]),
    "pgbp_regularize_bycluster": (C.c_int, [_P]),
    "pgbp_set_schedule": (C.c_int, [_P, C.c_int32, _I32P, _I32P, _I32P]),
    "pgbp_propagate": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Opts), _I32P]),
    "pgbp_traverse": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(Opts), C.POINTER(Result)]),
    "pgbp_calibrate": (C.c_int, [_P, C.c_int32, C.POINTER(Opts), C.POINTER(Result)]),
    "pgbp_integrate": (C.c_int, [_P, C.c_int32, _F64P, _F64P, _I32P]),
    "pgbp_free_energy": (C.c_int, [_P, _F64P, _I32P]),
    "pgbp_bm_tree_setup": (C.c_int, [_P, C.POINTER(BmTree)]),
    "pgbp_bm_tree_assignfactors": (C.c_int, [_P, _F64P, _F64P, _F64P, C.c_int32]),
    "pgbp_enqueue_loglik_bm": (C.c_int, [_P, C.c_int32, C.POINTER(Opts)]),
    "pgbp_lg_setup": (C.c_int, [_P, C.POINTER(LgFamilies)]),
    "pgbp_lg_assignfactors": (C.c_int, [_P, C.POINTER(LgParams)]),
    "pgbp_enqueue_loglik_lg": (C.c_int, [_P, C.c_int32, C.POINTER(Opts)]),
    "pgbp_enqueue_calibrate": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(Opts)]),
    "pgbp_enqueue_calibrate_timed": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(Opts)]),
    "pgbp_fetch_kernel_time": (C.c_int, [_P, C.POINTER(C.c_float), _I32P]),
    "pgbp_enqueue_loglik": (C.c_int, [_P, C.c_int32, C.POINTER(Opts)]),
    "pgbp_fetch_loglik": (C.c_int, [_P, _F64P, _I32P]),
    "pgbp_sync": (C.c_int, [_P]),
    "pgbp_time_enqueued": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Opts), C.POINTER(C.c_float)]),
    "pgbp_time_message_kernels": (C.c_int, [_P, C.c_int32, C.POINTER(Opts), C.POINTER(C.c_float), _I32P]),
    "pgbp_traffic_model": (C.c_int, [_P, _F64P, _I64P]),
    # several GPUs: one process / several devices (pgbp_group), one process per GPU (pgbp_comm: RCCL)
    "pgbp_group_create": (C.c_int, [C.POINTER(Desc), C.c_int32, _I32P, C.POINTER(_P)]),
    "pgbp_group_destroy": (None, [_P]),
    "pgbp_group_last_error": (C.c_char_p, [_P]),
    "pgbp_group_size": (C.c_int32, [_P]),
    "pgbp_group_engine": (_P, [_P, C.c_int32]),
    "pgbp_group_range": (C.c_int, [_P, C.c_int32, _I32P, _I32P]),
    "pgbp_group_set_schedule": (C.c_int, [_P, C.c_int32, _I32P, _I32P, _I32P]),
    "pgbp_group_set_beliefs": (C.c_int, [_P, _F64P, C.c_int32]),
    "pgbp_group_get_beliefs": (C.c_int, [_P, _F64P]),
    "pgbp_group_reset_from_factors": (C.c_int, [_P]),
    "pgbp_group_calibrate": (C.c_int, [_P, C.c_int32, C.POINTER(Opts), C.POINTER(Result)]),
    "pgbp_group_integrate": (C.c_int, [_P, C.c_int32, _F64P, _F64P, _I32P]),
    "pgbp_group_lg_setup": (C.c_int, [_P, C.POINTER(LgFamilies)]),
    "pgbp_group_lg_assignfactors": (C.c_int, [_P, C.POINTER(LgParams), C.c_int32, C.c_int32]),
    "pgbp_group_enqueue_calibrate": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(Opts)]),
    "pgbp_group_enqueue_loglik": (C.c_int, [_P, C.c_int32, C.POINTER(Opts)]),
    "pgbp_group_enqueue_loglik_lg": (C.c_int, [_P, C.c_int32, C.POINTER(Opts)]),
    "pgbp_group_fetch_loglik": (C.c_int, [_P, _F64P, _I32P]),
    "pgbp_group_sync": (C.c_int, [_P]),
    "pgbp_patterns_create": (C.c_int, [C.c_int32, C.POINTER(C.POINTER(Desc)), _I32P, C.POINTER(_P)]),
    "pgbp_patterns_destroy": (None, [_P]),
    "pgbp_patterns_last_error": (C.c_char_p, [_P]),
    "pgbp_patterns_size": (C.c_int32, [_P]),
    "pgbp_patterns_engine": (_P, [_P, C.c_int32]),
    "pgbp_patterns_set_schedule": (C.c_int, [_P, C.c_int32, _I32P, _I32P, _I32P]),
    "pgbp_patterns_calibrate": (C.c_int, [_P, C.c_int32, C.POINTER(Opts), C.POINTER(Result)]),
    "pgbp_patterns_enqueue_calibrate": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(Opts)]),
    "pgbp_patterns_enqueue_loglik": (C.c_int, [_P, C.c_int32, C.POINTER(Opts)]),
    "pgbp_patterns_enqueue_loglik_lg": (C.c_int, [_P, C.c_int32, C.POINTER(Opts)]),
    "pgbp_patterns_fetch_loglik": (C.c_int, [_P, _F64P, _I32P]),
    "pgbp_patterns_integrate": (C.c_int, [_P, C.c_int32, _F64P, _I32P]),
    "pgbp_patterns_sync": (C.c_int, [_P]),
    "pgbp_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "pgbp_comm_create": (C.c_int, [C.POINTER(C.c_uint8), C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P)]),
    "pgbp_comm_destroy": (None, [_P]),
    "pgbp_comm_last_error": (C.c_char_p, [_P]),
    "pgbp_comm_gather_loglik": (C.c_int, [_P, _P, C.c_int32, _F64P, _I32P, _I32P, _I32P]),
    "pgbp_comm_precheck": (C.c_int, [C.c_int32]),
    "pgbp_comm_exchange_beliefs": (C.c_int, [_P, _P, C.c_int32, _I32P, _I32P, C.c_int32]),
    "pgbp_comm_unpack_slots": (C.c_int, [_F64P, C.c_int32, C.c_int32, _F64P, _I32P, _I32P, _I32P]),
}

_lib = None


def load():
    """Load libpgbp.so (once). Raises if it has not been built: no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP engine first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C phylogaussianbeliefprop.jl_amd/csrc). "
            "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def i32p(a):
    return a.ctypes.data_as(_I32P)


def i64p(a):
    return a.ctypes.data_as(_I64P)


def f64p(a):
    return a.ctypes.data_as(_F64P)


def make_desc(dims, sepset_clusters, scope_off, scope_idx, n_sites=1, device=0):
    """Returns (Desc, keepalive) -- keepalive holds the numpy arrays the Desc points into."""
    dims = np.ascontiguousarray(dims, dtype=np.int32)
    sc = np.ascontiguousarray(sepset_clusters, dtype=np.int32).reshape(-1)
    so = np.ascontiguousarray(scope_off, dtype=np.int64)
    si = np.ascontiguousarray(scope_idx, dtype=np.int32)
    if si.size == 0:
        si = np.zeros(1, dtype=np.int32)
    if sc.size == 0:
        sc = np.zeros(2, dtype=np.int32)
    ns = sc.size // 2 if len(np.atleast_1d(sepset_clusters)) else 0
    d = Desc()
    d.n_sepsets = int(len(so) - 1) // 2
    d.n_clusters = int(len(dims) - d.n_sepsets)
    d.dims = i32p(dims)
    d.sepset_clusters = i32p(sc)
    d.scope_off = i64p(so)
    d.scope_idx = i32p(si)
    d.n_sites = int(n_sites)
    d.device = int(device)
    return d, (dims, sc, so, si)
