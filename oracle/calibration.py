"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restatement of src/calibration.jl:35-161 (calibrate!, the two traversals).
A schedule tree is the 4-tuple (pa_lab, ch_lab, pa_j, ch_j) of
src/clustergraph.jl:885-894 with 0-based belief indices.
"""
from __future__ import annotations

from typing import List

from . import beliefs as B


def propagate_1traversal_postorder(cgb: B.ClusterGraphBelief, pa_lab, ch_lab, pa_j, ch_j,
                                   verbose=True, update_residualnorm=True, log: List[str] = None,
                                   update_residualkldiv=False):
    """src/calibration.jl:111-135."""
    b, mr = cgb.belief, cgb.messageresidual
    for i in reversed(range(len(pa_lab))):
        sepset = b[cgb.sepsetindex(pa_lab[i], ch_lab[i])]
        mrss = mr[(pa_lab[i], ch_lab[i])]
        flag = B.propagate_belief(b[pa_j[i]], sepset, b[ch_j[i]], mrss)
        if flag is None:
            if update_residualnorm:
                B.iscalibrated_residnorm_update(mrss)
            if update_residualkldiv:
                B.residual_kldiv(mrss, sepset)
        else:
            if verbose and log is not None:
                log.append(("error", flag.msg))
            cgb.last_failure = flag
            return False
    return True


def propagate_1traversal_preorder(cgb: B.ClusterGraphBelief, pa_lab, ch_lab, pa_j, ch_j,
                                  verbose=True, update_residualnorm=True, log: List[str] = None,
                                   update_residualkldiv=False):
    """src/calibration.jl:137-161."""
    b, mr = cgb.belief, cgb.messageresidual
    for i in range(len(pa_lab)):
        sepset = b[cgb.sepsetindex(pa_lab[i], ch_lab[i])]
        mrss = mr[(ch_lab[i], pa_lab[i])]
        flag = B.propagate_belief(b[ch_j[i]], sepset, b[pa_j[i]], mrss)
        if flag is None:
            if update_residualnorm:
                B.iscalibrated_residnorm_update(mrss)
            if update_residualkldiv:
                B.residual_kldiv(mrss, sepset)
        else:
            if verbose and log is not None:
                log.append(("error", flag.msg))
            cgb.last_failure = flag
            return False
    return True


def calibrate_tree(cgb, spt, verbose=True, up_resnorm=True, log=None, up_reskldiv=False):
    """src/calibration.jl:72-84."""
    possucc = propagate_1traversal_postorder(cgb, *spt, verbose, up_resnorm, log, up_reskldiv)
    presucc = propagate_1traversal_preorder(cgb, *spt, verbose, up_resnorm, log, up_reskldiv)
    if not (possucc and presucc):
        return (False, False)
    return (True, cgb.iscalibrated_residnorm())


def calibrate(cgb, schedule, niter=1, auto=False, info=False, verbose=True,
              update_residualnorm=True, log=None, update_residualkldiv=False):
    """src/calibration.jl:35-60.  `log` collects (level, text) tuples for the
    @info/@error lines."""
    succ, iscal = False, False
    for i in range(1, niter + 1):
        for j, spt in enumerate(schedule, start=1):
            succ, iscal = calibrate_tree(cgb, spt, verbose, update_residualnorm, log, update_residualkldiv)
            if not succ:
                if info and log is not None:
                    log.append(("info", f"propagation failed: iteration {i}, schedule tree {j}"))
                return succ, iscal
            if iscal:
                if info and log is not None:
                    log.append(("info", f"calibration reached: iteration {i}, schedule tree {j}"))
                if auto:
                    return succ, iscal
    return succ, iscal
