"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restatement of the evolutionary-model factor formulas that produce the (h,J,g)
inputs of the hot path.  Reference: src/evomodels/*.jl.

Every factor is returned as (h, J, g) with the CHILD's traits first and then
the parents' traits (src/evomodels/evomodels.jl:185-192).
"""
from __future__ import annotations

import numpy as np

LOG2PI = float(np.log(2.0 * np.pi))


def _logdet_spd(a):
    sign, ld = np.linalg.slogdet(np.atleast_2d(a))
    assert sign > 0
    return float(ld)


class EvolutionaryModel:
    """src/evomodels/evomodels.jl:19-116 (interface)."""

    # subclasses set: p (ntraits), mu (p,), v (p,p) root prior variance
    def dimension(self):
        return self.p

    def rootpriormeanvector(self):
        return np.asarray(self.mu, dtype=float).reshape(self.p)

    def rootpriorvariance(self):
        return self.v

    def isrootfixed(self):
        # evomodels.jl:41  all(obj.v .== 0)
        return bool(np.all(np.asarray(self.v) == 0))

    # generic linear-Gaussian branch description: X_child | X_parent ~ N(q X_pa + w, V)
    def branch_qwv(self, edge):
        raise NotImplementedError

    # -- generic fallbacks: evomodels.jl:208-245 ---------------------------
    def factor_treeedge(self, edge):
        q, w, V = self.branch_qwv(edge)
        j = np.linalg.inv(V)
        g0 = (-self.p * LOG2PI + _logdet_spd(j)) / 2.0  # branch_logdet_precision :170-172
        return factor_from_qwj(q, w, j, 1, self.p, g0)

    # evomodels.jl:314-330
    def factor_hybridnode(self, pae):
        p = self.p
        npar = len(pae)
        v = np.zeros((p, p))
        w = np.zeros(p)
        q = np.zeros((p, npar * p))
        for k, e in enumerate(pae):
            qe, we, ve = self.branch_qwv(e)
            q[:, k * p:(k + 1) * p] = e.gamma * qe
            v += e.gamma ** 2 * ve
            w += e.gamma * we
        j = np.linalg.inv(v)
        g0 = (-p * LOG2PI + _logdet_spd(j)) / 2.0
        return factor_from_qwj(q, w, j, npar, p, g0)

    # evomodels.jl:377-396
    def factor_root(self):
        p = self.p
        v = np.atleast_2d(np.asarray(self.v, dtype=float))
        mu = self.rootpriormeanvector()
        improper = bool(np.any(np.isinf(np.diag(v))))
        if improper:
            return np.zeros(p), np.zeros((p, p)), 0.0
        j = np.linalg.inv(v)
        h = j @ mu
        g = (-p * LOG2PI + _logdet_spd(j) - float(mu @ h)) / 2.0
        return h, j, g


def factor_from_qwj(q, w, j, nparents, p, g0):
    """evomodels.jl:214-245: J = [j -jq; -q'j q'jq], h = [jw; -q'jw], g = g0 - w'jw/2."""
    jq = -j @ q
    qjq = -q.T @ jq
    ntot = p * (1 + nparents)
    J = np.zeros((ntot, ntot))
    J[:p, :p] = j
    J[:p, p:] = jq
    J[p:, :p] = jq.T
    J[p:, p:] = qjq
    jw = j @ w
    h = np.concatenate([jw, jq.T @ w])
    g = g0 - float(w @ jw) / 2.0
    return h, J, float(g)


class HomogeneousBM(EvolutionaryModel):
    """src/evomodels/homogeneousbrownianmotion.jl: R variance rate (p,p)."""

    def __init__(self, R, mu, v=None):
        R = np.atleast_2d(np.asarray(R, dtype=float))
        self.p = R.shape[0]
        self.R = R
        self.J = np.linalg.inv(R)
        self.mu = np.asarray(mu, dtype=float).reshape(self.p)
        if v is None:
            v = np.zeros((self.p, self.p))
        v = np.asarray(v, dtype=float)
        if v.ndim < 2:
            v = np.diag(v.reshape(self.p))
        self.v = v
        # g0 = -log(det(2 pi R))/2 : homogeneousbrownianmotion.jl:26,69,106
        self.g0 = -(self.p * LOG2PI + _logdet_spd(R)) / 2.0

    def branch_qwv(self, edge):
        return np.eye(self.p), np.zeros(self.p), self.R * edge.length

    # homogeneousbrownianmotion.jl:222-282 (t > 0 only; t == 0 is the
    # GeneralizedBelief path, out of scope)
    def factor_treeedge(self, edge):
        t = edge.length if hasattr(edge, "length") else float(edge)
        if t == 0:
            raise ValueError("degenerate (zero-length) edge: GeneralizedBelief path is out of scope")
        p = self.p
        j = self.J / t
        J = np.block([[j, -j], [-j, j]])
        h = np.zeros(2 * p)
        g = self.g0 - p * np.log(t) / 2.0
        return h, J, float(g)

    # homogeneousbrownianmotion.jl:288-351
    def factor_hybridnode(self, pae):
        t = np.array([e.length for e in pae], dtype=float)
        gam = np.array([e.gamma for e in pae], dtype=float)
        t0 = float(np.sum(gam ** 2 * t))
        if t0 == 0:
            raise ValueError("degenerate hybrid: GeneralizedBelief path is out of scope")
        p = self.p
        j = self.J / t0
        gv = np.concatenate([[1.0], -gam])  # [1 -gamma...]
        J = np.kron(np.outer(gv, gv), j)
        h = np.zeros(p * (1 + len(pae)))
        g = self.g0 - p * np.log(t0) / 2.0
        return h, J, float(g)


def UnivariateBrownianMotion(sigma2, mu, v=None):
    """homogeneousbrownianmotion.jl:16-40; v may be inf (improper root prior)."""
    return HomogeneousBM([[float(sigma2)]], [float(mu)], None if v is None else [[float(v)]])


def MvDiagBrownianMotion(R, mu, v=None):
    """homogeneousbrownianmotion.jl:51-85."""
    return HomogeneousBM(np.diag(np.asarray(R, dtype=float)), mu,
                         None if v is None else np.diag(np.asarray(v, dtype=float)))


def MvFullBrownianMotion(R, mu, v=None):
    """homogeneousbrownianmotion.jl:95-128."""
    return HomogeneousBM(R, mu, v)


class UnivariateOrnsteinUhlenbeck(EvolutionaryModel):
    """src/evomodels/homogeneousornsteinuhlenbeck.jl:18-66."""

    def __init__(self, sigma2, alpha, theta, mu, v=None):
        self.p = 1
        self.gamma2 = sigma2 / (2.0 * alpha)
        self.alpha = float(alpha)
        self.theta = float(theta)
        self.mu = np.array([float(mu)])
        self.v = np.array([[0.0 if v is None else float(v)]])
        self.g0 = -(LOG2PI + np.log(self.gamma2)) / 2.0

    def branch_qwv(self, edge):
        # :59-66
        actu = np.exp(-self.alpha * edge.length)
        facvar = 1.0 - actu ** 2
        return (np.array([[actu]]), np.array([(1.0 - actu) * self.theta]),
                np.array([[self.gamma2 * facvar]]))

    def factor_treeedge(self, edge):
        # :51-58 branch_transition_q.w.j.g then evomodels.jl:208-225
        q = np.exp(-self.alpha * edge.length)
        facvar = 1.0 - q ** 2
        j = 1.0 / self.gamma2 / facvar
        w = (1.0 - q) * self.theta
        g0 = self.g0 - np.log(facvar) / 2.0
        return factor_from_qwj(np.array([[q]]), np.array([w]), np.array([[j]]), 1, 1, g0)


class HeterogeneousBrownianMotion(EvolutionaryModel):
    """src/evomodels/heterogeneousmodels.jl:70-150; `colors` maps edge number -> 1-based rate index."""

    def __init__(self, rates, colors, mu, v=None):
        rates = [np.atleast_2d(np.asarray(R, dtype=float)) for R in rates]
        self.p = rates[0].shape[0]
        self.rates = rates
        self.inv = [np.linalg.inv(R) for R in rates]
        self.g0s = [-(self.p * LOG2PI + _logdet_spd(R)) / 2.0 for R in rates]
        self.colors = dict(colors)
        self.mu = np.asarray(mu, dtype=float).reshape(self.p)
        self.v = np.zeros((self.p, self.p)) if v is None else np.asarray(v, dtype=float)

    def _c(self, edge):
        return self.colors.get(edge.number, 1) - 1

    def branch_qwv(self, edge):
        return np.eye(self.p), np.zeros(self.p), self.rates[self._c(edge)] * edge.length

    def factor_treeedge(self, edge):
        # :128-134
        c = self._c(edge)
        j = self.inv[c] / edge.length
        g = self.g0s[c] - self.p * np.log(edge.length) / 2.0
        return factor_from_qwj(np.eye(self.p), np.zeros(self.p), j, 1, self.p, g)
    # factor_hybridnode (:135-150) == the generic fallback with w = 0.
