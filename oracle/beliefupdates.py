"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restatement of the canonical-form belief updates of src/beliefupdates.jl
(numpy, fp64; 0-based indices here, 1-based in the reference).
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import lapack, solve_triangular

LOG2PI = float(np.log(2.0 * np.pi))
EPS = float(np.finfo(np.float64).eps)


class BPPosDefException(Exception):
    """src/beliefupdates.jl:11-22.  `info` as LinearAlgebra.PosDefException."""

    def __init__(self, msg: str, info: int):
        super().__init__(msg)
        self.msg = msg
        self.info = int(info)

    def showerror(self) -> str:
        tail = "Hermitian." if self.info == -1 else "positive definite."
        return f"BPPosDefException: {self.msg}\nmatrix is not {tail}"


def _chol_upper_of_symmetric(A):
    """PDMat(Symmetric(A)): Cholesky reading only the UPPER triangle of A
    (src/beliefupdates.jl:68).  Returns (U, info) with A = U'U; info as LAPACK potrf."""
    U, info = lapack.dpotrf(np.asfortranarray(A), lower=0, clean=1, overwrite_a=0)
    return U, int(info)


def marginalize(h, J, g, keep_index, integrate_index=None, metadata="?"):
    """src/beliefupdates.jl:51-83.  Returns (h_S, J_S, g) of the message."""
    h = np.asarray(h, dtype=float)
    J = np.asarray(J, dtype=float)
    keep_index = np.asarray(keep_index, dtype=int).reshape(-1)
    if integrate_index is None:
        # :52 setdiff(1:length(h), keep_index)  (sorted ascending)
        mask = np.ones(h.shape[0], dtype=bool)
        mask[keep_index] = False
        integrate_index = np.nonzero(mask)[0]
    integrate_index = np.asarray(integrate_index, dtype=int).reshape(-1)
    if integrate_index.size == 0:  # :56
        return h, J, float(g)
    Ji = J[np.ix_(integrate_index, integrate_index)]
    Jk = J[np.ix_(keep_index, keep_index)]
    Jki = J[np.ix_(keep_index, integrate_index)]
    hi = h[integrate_index]
    hk = h[keep_index]
    # :62-66 "fake" all-zero block (missing data)
    if np.all(np.abs(Ji) <= EPS) and np.all(np.abs(hi) <= EPS) and np.all(np.abs(Jki) <= EPS):
        return hk.copy(), Jk.copy(), float(g)
    U, info = _chol_upper_of_symmetric(Ji)  # :68
    if info != 0:
        # :69-76 ; Julia prints the 1-based integrate_index vector
        idx1 = "[" + ", ".join(str(int(i) + 1) for i in integrate_index) + "]"
        raise BPPosDefException(f"belief {metadata}, integrating {idx1}", info)
    # :77 X_invA_Xt(Ji, Jki) = Jki Ji^{-1} Jki' via z = Jki U^{-1}
    z = solve_triangular(U, Jki.T, trans="T", lower=False).T if Jki.size else np.zeros((keep_index.size, integrate_index.size))
    messageJ = Jk - z @ z.T
    # :78 mu_i = Ji \ hi
    y = solve_triangular(U, hi, trans="T", lower=False)
    mui = solve_triangular(U, y, lower=False)
    messageh = hk - Jki @ mui  # :79
    ni = integrate_index.size
    logdet = 2.0 * float(np.sum(np.log(np.diag(U))))
    messageg = float(g) + (ni * LOG2PI - logdet + float(hi @ mui)) / 2.0  # :81
    return messageh, messageJ, messageg


def integratebelief(h, J, g):
    """src/beliefupdates.jl:187-200.  Returns (mu, norm)."""
    h = np.asarray(h, dtype=float)
    J = np.asarray(J, dtype=float)
    if not h.any() and not J.any():  # :189-191
        return np.full(h.shape, np.inf), float(g)
    U, info = _chol_upper_of_symmetric(J)
    if info != 0:
        raise np.linalg.LinAlgError(f"PosDefException: matrix is not positive definite; info={info}")
    n = h.shape[0]
    y = solve_triangular(U, h, trans="T", lower=False)
    mu = solve_triangular(U, y, lower=False)
    logdet = 2.0 * float(np.sum(np.log(np.diag(U))))
    norm = float(g) + (n * LOG2PI - logdet + float(np.sum(h * mu))) / 2.0
    return mu, norm


def absorbevidence(h, J, g, dataindex, datavalues):
    """src/beliefupdates.jl:210-231.  `datavalues` may hold None/nan for missing.
    Returns ((h_k, J_kk, g), missingdata_indices) -- indices into the kept vars."""
    h = np.asarray(h, dtype=float)
    J = np.asarray(J, dtype=float)
    dataindex = list(dataindex)
    vals = [None if (v is None or (isinstance(v, float) and np.isnan(v))) else float(v) for v in datavalues]
    assert len(dataindex) == len(vals)
    hasdata = [v is not None for v in vals]
    absorb = [dataindex[i] for i in range(len(vals)) if hasdata[i]]
    nvar = h.shape[0]
    keep = [i for i in range(nvar) if i not in set(absorb)]
    miss_idx = [keep.index(dataindex[i]) for i in range(len(vals)) if not hasdata[i]]
    data_nm = np.array([v for v in vals if v is not None], dtype=float)
    if not absorb:
        return (h, J, float(g)), miss_idx
    Jkk = J[np.ix_(keep, keep)]
    Jk_data = J[np.ix_(keep, absorb)] @ data_nm
    Ja_data = J[np.ix_(absorb, absorb)] @ data_nm
    g = float(g) + float(np.sum(h[absorb] * data_nm)) - float(np.sum(Ja_data * data_nm)) / 2.0
    hk = h[keep] - Jk_data
    return (hk, Jkk, g), miss_idx


def absorbleaf(h, J, g, datavalues, rowlabel="?"):
    """src/beliefupdates.jl:266-274: leaf traits are the first variables."""
    (h, J, g), miss = absorbevidence(h, J, g, range(len(datavalues)), datavalues)
    if miss:
        keep = [i for i in range(len(h)) if i not in set(miss)]
        h, J, g = marginalize(h, J, g, keep, miss, f"leaf row {rowlabel}")
    return h, J, g


def divide(sep_h, sep_J, sep_g, h, J, g):
    """src/beliefupdates.jl:579-587 (functional form).
    Returns (dh, dJ, dg) and the new sepset parameters (h, J, g)."""
    dh = h - sep_h
    dJ = J - sep_J
    dg = float(g) - float(sep_g)
    return dh, dJ, dg


def mult_inplace(to_h, to_J, to_g_arr, upind, dh, dJ, dg):
    """src/beliefupdates.jl:483-488 (in place; to_g_arr is a length-1 array)."""
    upind = np.asarray(upind, dtype=int)
    to_h[upind] += dh
    to_J[np.ix_(upind, upind)] += dJ
    to_g_arr[0] += dg
