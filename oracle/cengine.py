"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): ctypes wrapper of oracle/c/pgbp_oracle.c,
the plain-C sequential engine in the reference's message order.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpgbp_oracle.so")
_lib = None


def use_native_build():
    """Build (on THIS machine, with -O3 -march=native) and select a host-tuned copy of the C oracle:
    bench.py's cpu_baseline calls this so that the CPU number is not handicapped by a portable build.
    Must be called before the first lib()."""
    global _lib
    so = os.path.join(_HERE, "_build", "libpgbp_oracle_native.so")
    subprocess.check_call(["make", "-s", "-B", "-C", os.path.join(_HERE, "c"), f"OUT={so}",
                           "CFLAGS=-O3 -march=native -std=c99 -fPIC -shared -w -fopenmp"],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    _lib = None
    return _load(so)


def _load(so):
    global _lib
    _lib = C.CDLL(so)
    _lib.orc_create.restype = C.c_void_p
    _lib.orc_packed_size.restype = C.c_int64
    return _lib


def lib():
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "c")], stdout=subprocess.DEVNULL)
        _load(_SO)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Engine:
    def __init__(self, dims, sepset_clusters, scope_off, scope_idx, packed):
        L = lib()
        self.dims = np.ascontiguousarray(dims, np.int32)
        self.sepcl = np.ascontiguousarray(np.asarray(sepset_clusters, np.int32).reshape(-1))
        self.scope_off = np.ascontiguousarray(scope_off, np.int64)
        self.scope_idx = np.ascontiguousarray(scope_idx if len(scope_idx) else np.zeros(1), np.int32)
        self.ns = (len(self.scope_off) - 1) // 2
        self.nc = len(self.dims) - self.ns
        self.factors = np.ascontiguousarray(np.asarray(packed, np.float64).reshape(-1)).copy()
        sc = self.sepcl if self.sepcl.size else np.zeros(2, np.int32)
        self.h = C.c_void_p(L.orc_create(self.nc, self.ns, _p(self.dims, C.c_int32), _p(sc, C.c_int32),
                                         _p(self.scope_off, C.c_int64), _p(self.scope_idx, C.c_int32),
                                         _p(self.factors, C.c_double)))
        assert L.orc_packed_size(self.h) == self.factors.size
        self._sepmap = {}
        for k in range(self.ns):
            a, b = int(self.sepcl[2 * k]), int(self.sepcl[2 * k + 1])
            self._sepmap[(min(a, b), max(a, b))] = k

    def __del__(self):
        try:
            lib().orc_destroy(self.h)
        except Exception:
            pass

    def reset(self):
        L = lib()
        L.orc_set(self.h, _p(self.factors, C.c_double))
        L.orc_reset_flags(self.h)

    def _edges(self, pa, ch):
        pa = np.ascontiguousarray(pa, np.int32)
        ch = np.ascontiguousarray(ch, np.int32)
        sk = np.array([self._sepmap[(min(int(a), int(b)), max(int(a), int(b)))] for a, b in zip(pa, ch)], np.int32)
        return pa, ch, sk

    def prepare(self, pa, ch):
        """Resolve the sepset of every schedule edge once (outside any timed region)."""
        self._prepared = self._edges(pa, ch)

    def calibrate(self, pa=None, ch=None, niter=1, post_only=False, return_iscal=False):
        pa, ch, sk = self._prepared if pa is None else self._edges(pa, ch)
        iscal = C.c_int(0)
        succ = lib().orc_calibrate(self.h, len(pa), _p(pa, C.c_int32), _p(ch, C.c_int32), _p(sk, C.c_int32),
                                   int(niter), int(post_only), C.byref(iscal))
        return (bool(succ), bool(iscal.value)) if return_iscal else bool(succ)

    def levels_of_tree(self, pa, ch):
        """Level-synchronous form of one (postorder, preorder) pass over the schedule tree (pa, ch in preorder), for the
        all-core baseline: (level_off, task_off, ent_to, ent_k, ent_from).  Postorder: level = height of the sender,
        one task per (level, receiver) with its messages in the reference's order (decreasing edge index); preorder:
        level = depth of the sender, one task per message."""
        pa, ch, sk = self._edges(pa, ch)
        n = len(pa)
        height, depth = {}, {int(pa[0]): 0} if n else {}
        lvl_post = np.zeros(n, np.int64)
        for i in range(n - 1, -1, -1):
            h = height.get(int(ch[i]), 0)
            lvl_post[i] = h
            height[int(pa[i])] = max(height.get(int(pa[i]), 0), h + 1)
        lvl_pre = np.zeros(n, np.int64)
        for i in range(n):
            d = depth[int(pa[i])]
            lvl_pre[i] = d
            depth[int(ch[i])] = d + 1
        level_off, task_off, e_to, e_k, e_from = [0], [0], [], [], []
        for L in range(int(lvl_post.max()) + 1 if n else 0):
            by_recv = {}
            for i in np.nonzero(lvl_post == L)[0][::-1]:
                by_recv.setdefault(int(pa[i]), []).append(int(i))
            for recv, edges in by_recv.items():
                for i in edges:
                    e_to.append(recv); e_k.append(int(sk[i])); e_from.append(int(ch[i]))
                task_off.append(len(e_to))
            level_off.append(len(task_off) - 1)
        for L in range(int(lvl_pre.max()) + 1 if n else 0):
            for i in np.nonzero(lvl_pre == L)[0]:
                e_to.append(int(ch[i])); e_k.append(int(sk[i])); e_from.append(int(pa[i]))
                task_off.append(len(e_to))
            level_off.append(len(task_off) - 1)
        as32 = lambda v: np.ascontiguousarray(v, np.int32)
        return as32(level_off), as32(task_off), as32(e_to), as32(e_k), as32(e_from)

    def calibrate_levels(self, levels, niter=1, nthreads=0, return_iscal=False):
        """All-core (OpenMP) level-synchronous calibrate!(): `levels` from levels_of_tree; nthreads 0 = every core."""
        level_off, task_off, e_to, e_k, e_from = levels
        iscal = C.c_int(0)
        succ = lib().orc_calibrate_levels(self.h, len(level_off) - 1, _p(level_off, C.c_int32), _p(task_off, C.c_int32),
                                          _p(e_to, C.c_int32), _p(e_k, C.c_int32), _p(e_from, C.c_int32), int(niter),
                                          int(nthreads), C.byref(iscal))
        return (bool(succ), bool(iscal.value)) if return_iscal else bool(succ)

    def propagate(self, to, sepset_k, frm):
        return int(lib().orc_propagate(self.h, int(to), int(sepset_k), int(frm)))

    def last_failure(self):
        e, d, i = C.c_int(), C.c_int(), C.c_int()
        lib().orc_last_failure(self.h, C.byref(e), C.byref(d), C.byref(i))
        return e.value, d.value, i.value

    def integrate(self, b):
        m = int(self.dims[b])
        mu = np.zeros(max(1, m))
        norm = C.c_double()
        info = lib().orc_integrate(self.h, int(b), _p(mu, C.c_double), C.byref(norm))
        if info:
            raise np.linalg.LinAlgError(f"not positive definite (info={info})")
        return mu[:m], norm.value

    def packed(self):
        out = np.zeros(self.factors.size)
        lib().orc_get(self.h, _p(out, C.c_double))
        return out

    def residuals(self):
        sdims = np.repeat(self.dims[self.nc:].astype(np.int64), 2)
        n = int(np.sum(sdims * sdims + sdims))
        res = np.zeros(max(1, n))
        flags = np.zeros(max(1, 2 * self.ns), np.int32)
        lib().orc_get_residuals(self.h, _p(res, C.c_double), _p(flags, C.c_int32))
        return res[:n], flags[:2 * self.ns]
