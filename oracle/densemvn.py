"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Independent oracle: dense multivariate-normal log-likelihood of tip data on a
tree or network.  This is the recipe the reference's authors used to generate
their golden values: `PhyloNetworks.vcv(net)` + `loglikelihood(MvNormal(..))`
(test/test_evomodels.jl:265-316, test/test_calibration.jl:119-124,
test/test_canonicalform.jl:111-115) and, for OU, the hand recursion in
test/test_evomodels.jl:121-167.  It shares no code with the message-passing
restatement (only the models' branch_qwv description of each edge).
"""
from __future__ import annotations

import numpy as np

LOG2PI = float(np.log(2.0 * np.pi))


def node_moments(net, model, root_mean=None, root_var=None):
    """Mean (N*p,), covariance (N*p,N*p) of all node states in preorder, and the
    root-propagation matrix A (N*p, p) such that E[X | x_root] = A x_root + const."""
    pre = net.vec_node
    p = model.dimension()
    N = len(pre)
    pos = {id(n): i for i, n in enumerate(pre)}
    mean = np.zeros(N * p)
    cov = np.zeros((N * p, N * p))
    A = np.zeros((N * p, p))
    sl = lambda i: slice(i * p, (i + 1) * p)
    mean[sl(0)] = model.rootpriormeanvector() if root_mean is None else root_mean
    cov[sl(0), sl(0)] = model.rootpriorvariance() if root_var is None else root_var
    A[sl(0)] = np.eye(p)
    for i in range(1, N):
        n = pre[i]
        pes = net.parent_edges(n)
        qs, ws, vs, pis = [], [], [], []
        for e in pes:
            q, w, v = model.branch_qwv(e)
            qs.append(e.gamma * q)
            ws.append(e.gamma * w)
            vs.append(e.gamma ** 2 * v)
            pis.append(pos[id(e.parent)])
        m = np.zeros(p)
        for q, w, pi in zip(qs, ws, pis):
            m += q @ mean[sl(pi)] + w
            A[sl(i)] += q @ A[sl(pi)]
        mean[sl(i)] = m
        # covariance with every earlier node
        row = np.zeros((p, i * p))
        for q, pi in zip(qs, pis):
            row += q @ cov[sl(pi), : i * p]
        cov[sl(i), : i * p] = row
        cov[: i * p, sl(i)] = row.T
        vii = sum(vs)
        for q1, p1 in zip(qs, pis):
            for q2, p2 in zip(qs, pis):
                vii = vii + q1 @ cov[sl(p1), sl(p2)] @ q2.T
        cov[sl(i), sl(i)] = vii
    return mean, cov, A


def loglik(net, model, tbl, taxa):
    """log-likelihood of the tip data under `model` (fixed, random or improper root).
    tbl: list of trait columns over `taxa`, None = missing."""
    pre = net.vec_node
    p = model.dimension()
    v = np.atleast_2d(np.asarray(model.rootpriorvariance(), dtype=float))
    improper = bool(np.any(np.isinf(np.diag(v))))
    rootvar = np.zeros((p, p)) if improper else v
    rootmean = np.zeros(p) if improper else model.rootpriormeanvector()
    mean, cov, A = node_moments(net, model, rootmean, rootvar)
    obs, y = [], []
    for i, n in enumerate(pre):
        if not n.leaf:
            continue
        r = list(taxa).index(n.name)
        for t in range(p):
            val = tbl[t][r]
            if val is not None:
                obs.append(i * p + t)
                y.append(float(val))
    obs = np.array(obs, dtype=int)
    y = np.array(y)
    S = cov[np.ix_(obs, obs)]
    r = y - mean[obs]
    L = np.linalg.cholesky(S)
    z = np.linalg.solve(L, r)
    n = len(y)
    ll = -0.5 * float(z @ z) - float(np.sum(np.log(np.diag(L)))) - 0.5 * n * LOG2PI
    if improper:
        Ao = A[obs]
        W = np.linalg.solve(L, Ao)            # L^{-1} A
        M = W.T @ W                           # A' S^{-1} A
        b = W.T @ z                           # A' S^{-1} r
        Lm = np.linalg.cholesky(M)
        u = np.linalg.solve(Lm, b)
        # integral over a flat prior on the root state
        ll += 0.5 * float(u @ u) - float(np.sum(np.log(np.diag(Lm)))) + 0.5 * p * LOG2PI
    return ll


def posterior_node_moments(net, model, tbl, taxa):
    """Conditional mean/covariance of all node states given the tip data (proper root)."""
    pre = net.vec_node
    p = model.dimension()
    mean, cov, _ = node_moments(net, model)
    obs, y = [], []
    for i, n in enumerate(pre):
        if n.leaf:
            r = list(taxa).index(n.name)
            for t in range(p):
                if tbl[t][r] is not None:
                    obs.append(i * p + t)
                    y.append(float(tbl[t][r]))
    obs = np.array(obs, dtype=int)
    y = np.array(y)
    S = cov[np.ix_(obs, obs)]
    K = np.linalg.solve(S, cov[obs, :]).T
    pm = mean + K @ (y - mean[obs])
    pc = cov - K @ cov[obs, :]
    return pm, pc
