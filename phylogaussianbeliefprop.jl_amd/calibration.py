"""calibrate! and the two traversals (src/calibration.jl:35-161) on the device engine."""
import ctypes as C
import logging

import numpy as np

from . import _lib as L
from .clustergraphbeliefs import _check

log = logging.getLogger("PhyloGaussianBeliefProp")


def _fail_exception(beliefs, r, tree):
    pa, ch = beliefs._schedule[tree]
    i = r.fail_edge
    sender = int(ch[i]) if r.fail_dir == 0 else int(pa[i])
    receiver = int(pa[i]) if r.fail_dir == 0 else int(ch[i])
    k = beliefs._msg_id(receiver, sender) // 2
    return beliefs._exception_for(sender, k, r.fail_info)


def _traverse(beliefs, spt, direction, verbose, update_residualnorm, update_residualkldiv, sync):
    beliefs._ensure_schedule([spt])
    res = (L.Result * beliefs.n_sites)()
    o = beliefs._opts(False, update_residualnorm, update_residualkldiv)
    _check(beliefs._lib.pgbp_traverse(beliefs._eng, 0, direction, C.byref(o), res), beliefs._eng)
    beliefs.last_results = res
    if sync:
        beliefs._invalidate()   # lazy write-back: beliefs and residuals are fetched when first read
    r = res[beliefs.site]
    if not r.succ:
        ex = _fail_exception(beliefs, r, 0)
        beliefs.last_failure = ex
        if verbose:
            log.error(ex.msg)  # @error flag.msg (src/calibration.jl:130,156)
        return False
    return True


def propagate_1traversal_postorder_(beliefs, pa_lab, ch_lab, pa_j, ch_j, verbose=True,
                                    update_residualnorm=True, update_residualkldiv=False, sync=True):
    """propagate_1traversal_postorder! (src/calibration.jl:111-135)."""
    return _traverse(beliefs, (pa_lab, ch_lab, pa_j, ch_j), 0, verbose, update_residualnorm,
                     update_residualkldiv, sync)


def propagate_1traversal_preorder_(beliefs, pa_lab, ch_lab, pa_j, ch_j, verbose=True,
                                   update_residualnorm=True, update_residualkldiv=False, sync=True):
    """propagate_1traversal_preorder! (src/calibration.jl:137-161)."""
    return _traverse(beliefs, (pa_lab, ch_lab, pa_j, ch_j), 1, verbose, update_residualnorm,
                     update_residualkldiv, sync)


def calibrate_(beliefs, schedule, niter=1, auto=False, info=False, verbose=True,
               update_residualnorm=True, update_residualkldiv=False, sync=True):
    """calibrate!(beliefs, schedule, niter; auto, info, verbose, update_residualnorm,
    update_residualkldiv) -> (succ, iscal) (src/calibration.jl:35-60).  `schedule` is a list of
    spanning trees or a single tree tuple (the :72-84 method)."""
    if isinstance(schedule, tuple):
        schedule = [schedule]
    beliefs._ensure_schedule(schedule)
    res = (L.Result * beliefs.n_sites)()
    o = beliefs._opts(auto, update_residualnorm, update_residualkldiv)
    _check(beliefs._lib.pgbp_calibrate(beliefs._eng, int(niter), C.byref(o), res), beliefs._eng)
    beliefs.last_results = res
    if sync:
        beliefs._invalidate()   # lazy write-back: beliefs and residuals are fetched when first read
    r = res[beliefs.site]
    if not r.succ:
        ex = _fail_exception(beliefs, r, r.fail_tree - 1)
        beliefs.last_failure = ex
        if verbose:
            log.error(ex.msg)
        if info:
            log.info("propagation failed: iteration %d, schedule tree %d" % (r.fail_iter, r.fail_tree))
        return (False, False)
    if r.iter_reached and info:
        log.info("calibration reached: iteration %d, schedule tree %d" % (r.iter_reached, r.tree_reached))
    return (bool(r.succ), bool(r.iscal))
