"""Callers of the hot path: maximum-likelihood / maximum-factored-energy estimation of the model parameters
(calibrate_optimize_cliquetree!, calibrate_optimize_clustergraph!: src/calibration.jl:163-359) with the whole objective
on the device -- factors assigned from the candidate parameters, messages passed, root integrated (or the free energy
evaluated) without the belief state ever leaving HBM; only the parameters and one number per evaluation cross the bus.

The reference minimises with Optim.jl's LBFGS over unconstrained parameters; here the same transforms and the same
objective go to scipy's L-BFGS-B with central-difference gradients (the optimum is a property of the objective, not of
the optimiser).  The ClusterGraphBelief must have its node families set up (`lg_setup`)."""
import ctypes as C

import numpy as np

from . import _lib as L


class _BMTransform:
    """params_optimize / params_original of UnivariateBrownianMotion (log sigma2, mu:
    src/evomodels/homogeneousbrownianmotion.jl:48-49) and MvFullBrownianMotion (log-Cholesky of R, then mu: :130-157)."""

    def __init__(self, p, diagonal=False):
        self.p = p
        self.diagonal = diagonal   # MvDiagBrownianMotion: log of the rates, then mu (:89-90)

    def forward(self, R, mu):
        R = np.atleast_2d(np.asarray(R, float))
        if self.diagonal:
            return np.array(list(np.log(np.diag(R))) + list(np.asarray(mu, float).reshape(self.p)))
        U = np.linalg.cholesky(R).T
        p = self.p
        above = [U[i, j] for j in range(1, p) for i in range(j)]
        return np.array([np.log(U[i, i]) for i in range(p)] + above + list(np.asarray(mu, float).reshape(p)))

    def back(self, theta):
        p = self.p
        if self.diagonal:
            return np.diag(np.exp(np.asarray(theta[:p], float))), np.asarray(theta[p:2 * p], float)
        U = np.zeros((p, p))
        k = 0
        for i in range(p):
            U[i, i] = np.exp(theta[k]); k += 1
        for j in range(1, p):
            for i in range(j):
                U[i, j] = theta[k]; k += 1
        return U.T @ U, np.asarray(theta[k:k + p], float)


def _minimise(score, x0, maxiter):
    from scipy.optimize import minimize

    def grad(x):
        g = np.zeros_like(x)
        for i in range(len(x)):
            h = 1e-6 * max(1.0, abs(x[i]))
            e = np.zeros_like(x); e[i] = h
            g[i] = (score(x + e) - score(x - e)) / (2 * h)
        return g
    # L-BFGS-B, restarted from its own end point while it still improves (its line search gives up early on badly scaled
    # starts, e.g. rates two orders of magnitude off), then a Nelder-Mead polish when a restart stalls above the tolerance
    best = minimize(score, x0, jac=grad, method="L-BFGS-B", options={"maxiter": maxiter, "ftol": 1e-15, "gtol": 1e-9})
    for _ in range(20):
        nxt = minimize(score, best.x, jac=grad, method="L-BFGS-B", options={"maxiter": maxiter, "ftol": 1e-15, "gtol": 1e-9})
        improved = nxt.fun < best.fun - 1e-13 * max(1.0, abs(best.fun))
        nxt.nfev += best.nfev
        if nxt.fun <= best.fun:
            best = nxt
        if not improved:
            break
    if np.linalg.norm(grad(best.x)) > 1e-5 * max(1.0, abs(best.fun)):
        pol = minimize(score, best.x, method="Nelder-Mead", options={"xatol": 1e-10, "fatol": 1e-13, "maxiter": 400 * len(x0)})
        pol.nfev += best.nfev
        if pol.fun <= best.fun:
            best = pol
            nxt = minimize(score, best.x, jac=grad, method="L-BFGS-B", options={"maxiter": maxiter, "ftol": 1e-15, "gtol": 1e-9})
            if nxt.fun <= best.fun:
                nxt.nfev += best.nfev
                best = nxt
    return best


def calibrate_optimize_cliquetree_(beliefs, schedule_tree, R0, mu0, extra_rates=(), maxiter=200, diagonal=False):
    """calibrate_optimize_cliquetree! (src/calibration.jl:183-221) for a homogeneous Brownian motion (univariate or full
    rate matrix): maximise the log-likelihood over (R, mu); the root prior variance, if the root is random, stays fixed
    (`extra_rates`: the matrices that follow R in the rate table of lg_setup, e.g. the root prior variance).
    Each evaluation = assignfactors! + postorder of `schedule_tree` + integratebelief! at its root, on the device.
    diagonal: MvDiagBrownianMotion (independent traits: only the diagonal of R is estimated).
    Returns (R, mu, loglik, scipy result)."""
    p = beliefs._lg_p
    tf = _BMTransform(p, diagonal)
    beliefs._ensure_schedule([schedule_tree])

    def score(theta):
        R, mu = tf.back(theta)
        try:
            np.linalg.cholesky(R)
        except np.linalg.LinAlgError:
            return np.inf
        rates = np.stack([R] + [np.atleast_2d(np.asarray(x, float)) for x in extra_rates])
        beliefs.assignfactors_lg_(rates, mu)
        ll, info = beliefs.loglik_lg()
        return np.inf if (info[0] or not np.isfinite(ll[0])) else -float(ll[0])
    opt = _minimise(score, tf.forward(R0, mu0), maxiter)
    R, mu = tf.back(opt.x)
    return R, mu, -float(opt.fun), opt


def calibrate_optimize_clustergraph_(beliefs, schedule, R0, mu0, extra_rates=(), maxiter_calibration=100, maxiter=200,
                                     diagonal=False):
    """calibrate_optimize_clustergraph! (src/calibration.jl:309-359): maximise the factored energy (= minus the Bethe free
    energy; the log-likelihood on a clique tree) over (R, mu).  Each evaluation = assignfactors! + factors from beliefs +
    regularizebeliefs_bycluster! + calibrate!(schedule, maxiter_calibration; auto=true) + free_energy, on the device.
    Returns (R, mu, factored energy, scipy result)."""
    from .calibration import calibrate_
    p = beliefs._lg_p
    tf = _BMTransform(p, diagonal)
    lib = beliefs._lib

    def score(theta):
        R, mu = tf.back(theta)
        try:
            np.linalg.cholesky(R)
        except np.linalg.LinAlgError:
            return np.inf
        rates = np.stack([R] + [np.atleast_2d(np.asarray(x, float)) for x in extra_rates])
        beliefs.assignfactors_lg_(rates, mu)               # also snapshots the factors and resets the flags
        if lib.pgbp_regularize_bycluster(beliefs._eng) != L.PGBP_OK:
            return np.inf
        succ, _ = calibrate_(beliefs, schedule, maxiter_calibration, auto=True, verbose=False, sync=False)
        if not succ:
            return np.inf
        out, info = beliefs.free_energy(all_sites=True)
        return np.inf if info[0] or not np.isfinite(out[0, 2]) else float(out[0, 2])
    opt = _minimise(score, tf.forward(R0, mu0), maxiter)
    R, mu = tf.back(opt.x)
    return R, mu, -float(opt.fun), opt
