"""
Synthetic inputs for the hot path (host side, vectorised numpy): random bifurcating
trees, the clique tree / Bethe cluster graph of a tree, homogeneous-BM factor fill,
and BM tip-data simulation -- the configurations of BASELINE.json / SURVEY.md section 8(d).

The reference builds these with O(n^2) host code (src/beliefs.jl:521-536,
src/clustergraph.jl:92-104) that cannot reach 50k tips; this module produces the same
objects directly for trees:
  * clique tree of a tree = one clique {child, parent} per edge, sepset = shared node,
    topology mirroring the phylogeny (src/clustergraph.jl:452-466 on a tree);
  * nodes inside a belief sorted by decreasing preorder index (src/clustergraph.jl:764-769),
    traits contiguous within a node (src/beliefs.jl:347-353);
  * fixed root and tips are out of scope (src/beliefs.jl:551-559);
  * factors: assignfactors! for MvFullBrownianMotion with complete data
    (src/beliefs.jl:786-861, src/evomodels/homogeneousbrownianmotion.jl:262-282,
     src/beliefupdates.jl:210-231).
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

LOG2PI = float(np.log(2.0 * np.pi))


@dataclass
class Tree:
    """Rooted tree, nodes numbered in preorder (root = 0)."""
    parent: np.ndarray   # (N,) int, parent[0] = -1
    length: np.ndarray   # (N,) float, length[0] unused
    is_leaf: np.ndarray  # (N,) bool

    @property
    def nnodes(self):
        return len(self.parent)

    @property
    def ntips(self):
        return int(self.is_leaf.sum())

    def depth(self):
        d = np.zeros(self.nnodes, dtype=np.int64)
        for i in range(1, self.nnodes):
            d[i] = d[self.parent[i]] + 1
        return d

    def newick(self, names=None):
        ch = [[] for _ in range(self.nnodes)]
        for i in range(1, self.nnodes):
            ch[self.parent[i]].append(i)
        def rec(i):
            nm = names[i] if names is not None else f"n{i}"
            if not ch[i]:
                return f"{nm}:{float(self.length[i])!r}"
            s = "(" + ",".join(rec(c) for c in ch[i]) + ")" + nm
            return s + (f":{float(self.length[i])!r}" if i else "")
        import sys
        sys.setrecursionlimit(max(10000, 4 * self.nnodes))
        return rec(0) + ";"


def random_tree(ntips: int, rng: np.random.Generator, lo=0.1, hi=1.0) -> Tree:
    """Random bifurcating tree by uniform random joins, edge lengths ~ U(lo, hi)."""
    assert ntips >= 2
    N = 2 * ntips - 1
    par = np.full(N, -1, dtype=np.int64)
    active = list(range(ntips))
    nxt = ntips
    while len(active) > 1:
        k = len(active)
        i = int(rng.integers(k))
        j = int(rng.integers(k - 1))
        if j >= i:
            j += 1
        a, b = active[i], active[j]
        par[a] = par[b] = nxt
        lo_i, hi_i = (i, j) if i < j else (j, i)
        active[lo_i] = nxt
        active[hi_i] = active[-1]
        active.pop()
        nxt += 1
    root = nxt - 1
    children = [[] for _ in range(N)]
    for v in range(N):
        if par[v] >= 0:
            children[par[v]].append(v)
    order, stack = [], [root]
    while stack:
        v = stack.pop()
        order.append(v)
        stack.extend(reversed(children[v]))
    new = np.empty(N, dtype=np.int64)
    new[np.array(order)] = np.arange(N)
    parent = np.full(N, -1, dtype=np.int64)
    for v in range(N):
        if par[v] >= 0:
            parent[new[v]] = new[par[v]]
    is_leaf = np.ones(N, dtype=bool)
    is_leaf[parent[1:]] = False
    length = rng.uniform(lo, hi, size=N)
    length[0] = 0.0
    return Tree(parent, length, is_leaf)


def random_multifurcating_tree(ntips: int, maxdeg: int, rng: np.random.Generator, lo=0.1, hi=1.0) -> Tree:
    """Random tree whose internal nodes have 2..maxdeg children (polytomies): exercises tasks with
    more than two messages per receiver / sender."""
    par = {}
    active = list(range(ntips))
    nxt = ntips
    while len(active) > 1:
        k = min(len(active), int(rng.integers(2, maxdeg + 1)))
        pick = sorted(rng.choice(len(active), size=k, replace=False).tolist(), reverse=True)
        for i in pick:
            par[active[i]] = nxt
            active.pop(i)
        active.append(nxt)
        nxt += 1
    N = nxt
    root = N - 1
    children = [[] for _ in range(N)]
    for v, q in par.items():
        children[q].append(v)
    order, stack = [], [root]
    while stack:
        v = stack.pop()
        order.append(v)
        stack.extend(reversed(sorted(children[v])))
    new = np.empty(N, dtype=np.int64)
    new[np.array(order)] = np.arange(N)
    parent = np.full(N, -1, dtype=np.int64)
    for v, q in par.items():
        parent[new[v]] = new[q]
    is_leaf = np.ones(N, dtype=bool)
    is_leaf[parent[1:]] = False
    length = rng.uniform(lo, hi, size=N)
    length[0] = 0.0
    return Tree(parent, length, is_leaf)


def caterpillar_tree(ntips: int, rng: np.random.Generator, lo=0.1, hi=1.0) -> Tree:
    """Maximally unbalanced tree (depth = ntips - 1): the worst case for level parallelism."""
    N = 2 * ntips - 1
    parent = np.full(N, -1, dtype=np.int64)
    is_leaf = np.zeros(N, dtype=bool)
    v = 0
    for i in range(ntips - 1):
        # internal node v has a leaf child v+1 and (except last) an internal child v+2
        parent[v + 1] = v
        is_leaf[v + 1] = True
        if i < ntips - 2:
            parent[v + 2] = v
            v += 2
        else:
            parent[v + 2] = v
            is_leaf[v + 2] = True
    length = rng.uniform(lo, hi, size=N)
    length[0] = 0
    return Tree(parent, length, is_leaf)


def simulate_bm(tree: Tree, R: np.ndarray, mu: np.ndarray, rng: np.random.Generator) -> np.ndarray:
    """Trait values at every node under BM(R) from a fixed root mu: (N, p)."""
    p = len(mu)
    Lc = np.linalg.cholesky(R)
    z = rng.standard_normal((tree.nnodes, p))
    x = np.zeros((tree.nnodes, p))
    x[0] = mu
    inc = (z @ Lc.T) * np.sqrt(tree.length)[:, None]
    for i in range(1, tree.nnodes):
        x[i] = x[tree.parent[i]] + inc[i]
    return x


def random_rate_matrix(p: int, rng: np.random.Generator) -> np.ndarray:
    """R = A A'/p + I."""
    A = rng.standard_normal((p, p))
    return A @ A.T / p + np.eye(p)


@dataclass
class Problem:
    """Everything pgbp_create / pgbp_set_beliefs / pgbp_set_schedule need."""
    dims: np.ndarray
    sepset_clusters: np.ndarray  # (n_sepsets, 2)
    scope_off: np.ndarray
    scope_idx: np.ndarray
    schedule: list               # [(pa_j, ch_j)]
    packed: Optional[np.ndarray] = None  # (n_sites, packed_size) factors (sepsets zero)
    packed_off: Optional[np.ndarray] = None
    nclusters: int = 0
    root_cluster: int = 0
    cluster_nodes: Optional[np.ndarray] = None  # (n_clusters, 2) 1-based node labels [child, parent] (clique tree)
    sepset_nodes: Optional[np.ndarray] = None   # (n_sepsets,) 1-based label of the shared node
    meta: dict = field(default_factory=dict)


def _packed_offsets(dims):
    m = dims.astype(np.int64)
    return np.concatenate([[0], np.cumsum(m * m + m + 1)])


def cliquetree_of_tree(tree: Tree, p: int) -> Problem:
    """Clique tree of a tree under a fixed-root model: cluster c-1 = {c, parent(c)} for every
    non-root node c (preorder), sepset e between cluster(k) and its parent cluster."""
    N = tree.nnodes
    par, leaf = tree.parent, tree.is_leaf
    c = np.arange(1, N)
    dc = np.where(leaf[c], 0, p)               # child's variables (tips out of scope)
    dpa = np.where(par[c] == 0, 0, p)          # parent's variables (fixed root out of scope)
    cdims = dc + dpa
    # schedule edge e (e = 0..N-3) joins cluster(k), k = e+2, to its parent cluster
    k = np.arange(2, N)
    pa_cluster = np.where(par[k] == 0, 0, par[k] - 1)   # cluster(parent(k)), or cluster(node 1) at the root
    ch_cluster = k - 1
    sdims = np.where(par[k] == 0, 0, p)
    dims = np.concatenate([cdims, sdims]).astype(np.int32)
    ns = len(k)
    # scope maps: side a = parent cluster (shared node is its first node), side b = child cluster
    lens = np.repeat(sdims, 2)
    scope_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    start_a = np.zeros(ns, dtype=np.int64)
    start_b = dc[k - 1].astype(np.int64)          # after k's own variables
    starts = np.stack([start_a, start_b], axis=1).reshape(-1)
    scope_idx = (np.repeat(starts, lens) + (np.arange(lens.sum()) - np.repeat(scope_off[:-1], lens))).astype(np.int32)
    prob = Problem(dims=dims, sepset_clusters=np.stack([pa_cluster, ch_cluster], axis=1).astype(np.int32),
                   scope_off=scope_off, scope_idx=scope_idx,
                   schedule=[(pa_cluster.astype(np.int32), ch_cluster.astype(np.int32))],
                   nclusters=N - 1, root_cluster=0)
    prob.packed_off = _packed_offsets(dims)
    prob.cluster_nodes = np.stack([c + 1, par[c] + 1], axis=1)
    prob.sepset_nodes = par[k] + 1
    prob.meta = {"graph": "cliquetree", "ntips": tree.ntips, "p": p, "child_dim": dc}
    return prob


def bm_factors_cliquetree(tree: Tree, prob: Problem, R: np.ndarray, mu: np.ndarray, Y: np.ndarray) -> np.ndarray:
    """Packed (J,h,g) of every cluster after assignfactors! under MvFullBrownianMotion(R, mu) with
    fixed root and complete tip data Y ((N, p): rows of tips are used). Sepsets are zero.
    Returns a (packed_size,) array."""
    N, p = tree.nnodes, len(mu)
    Rinv = np.linalg.inv(R)
    sign, logdetR = np.linalg.slogdet(R)
    g0 = -(p * LOG2PI + logdetR) / 2.0
    par, leaf, t = tree.parent, tree.is_leaf, tree.length
    packed = np.zeros(int(prob.packed_off[-1]))
    off = prob.packed_off
    c = np.arange(1, N)
    gbase = g0 - p * np.log(t[c]) / 2.0
    pa_root = par[c] == 0
    is_leaf = leaf[c]

    def put(sel, J, h, g):
        """sel: cluster ids; J (n,m,m) symmetric, h (n,m), g (n,)"""
        n = len(sel)
        if n == 0:
            return
        m = J.shape[1]
        rec = np.concatenate([J.reshape(n, m * m), h, g[:, None]], axis=1)
        L = m * m + m + 1
        for i, o in enumerate(off[sel].tolist()):  # slice copies: far faster than one fancy-index scatter
            packed[o:o + L] = rec[i]

    # internal child, internal parent: J = [j -j; -j j], h = 0
    s = np.nonzero(~is_leaf & ~pa_root)[0]
    if len(s):
        j = Rinv[None] / t[c[s]][:, None, None]
        J = np.concatenate([np.concatenate([j, -j], axis=2), np.concatenate([-j, j], axis=2)], axis=1)
        put(s, J, np.zeros((len(s), 2 * p)), gbase[s])
    # internal child of the fixed root: absorb mu on the parent's variables
    s = np.nonzero(~is_leaf & pa_root)[0]
    if len(s):
        j = Rinv[None] / t[c[s]][:, None, None]
        jm = j @ mu
        put(s, j, jm, gbase[s] - 0.5 * (jm @ mu))
    # leaf child, internal parent: absorb the data on the child's variables
    s = np.nonzero(is_leaf & ~pa_root)[0]
    if len(s):
        j = Rinv[None] / t[c[s]][:, None, None]
        y = Y[c[s]]
        jy = np.einsum("nab,nb->na", j, y)
        put(s, j, jy, gbase[s] - 0.5 * np.einsum("na,na->n", jy, y))
    # leaf child of the fixed root: a constant
    s = np.nonzero(is_leaf & pa_root)[0]
    if len(s):
        j = Rinv[None] / t[c[s]][:, None, None]
        r = Y[c[s]] - mu[None]
        jr = np.einsum("nab,nb->na", j, r)
        packed[off[s]] = gbase[s] - 0.5 * np.einsum("na,na->n", jr, r)
    return packed


def bm_tree_table(tree: Tree, prob: Problem):
    """Static description for the DEVICE factor fill (include/pgbp.h: pgbp_bm_tree): per cluster the kind of
    its node-family factor, the branch length and the data row (= node index) of a tip.  Works for
    cliquetree_of_tree and bethe_of_tree problems (variable clusters get kind -1)."""
    N = tree.nnodes
    c = np.arange(1, N)
    leaf, pa_root = tree.is_leaf[c], tree.parent[c] == 0
    kind = np.where(~leaf & ~pa_root, 0, np.where(~leaf & pa_root, 1, np.where(leaf & ~pa_root, 2, 3))).astype(np.int32)
    length = tree.length[c].astype(np.float64)
    row = np.where(leaf, c, -1).astype(np.int32)
    extra = prob.nclusters - (N - 1)
    if extra > 0:  # Bethe variable clusters
        kind = np.concatenate([kind, np.full(extra, -1, np.int32)])
        length = np.concatenate([length, np.ones(extra)])
        row = np.concatenate([row, np.full(extra, -1, np.int32)])
    return kind, length, row



def lg_tree_table(tree: Tree, prob: Problem, p: int, colors=None):
    """Node-family table for the general DEVICE factor fill (include/pgbp.h: pgbp_lg_families; the dictionary
    `factors.lg_families` returns) of a cliquetree_of_tree / bethe_of_tree problem under a fixed-root model: family of
    node c (preorder, c >= 1) = {c, parent(c)} in factor cluster c - 1; tips carry data row c.
    colors: per-node 0-based rate index of the edge above it (heterogeneous models), default 0."""
    N = tree.nnodes
    c = np.arange(1, N)
    leaf, pa_root = tree.is_leaf[c], tree.parent[c] == 0
    col = np.zeros(N - 1, np.int32) if colors is None else np.asarray(colors, np.int32)[c]
    return dict(p=int(p), max_parents=1, n_rates=int(col.max()) + 1 if col.size else 1,
                cluster=(c - 1).astype(np.int32), n_parents=np.ones(N - 1, np.int32),
                child_pos=np.where(leaf, -1, 0).astype(np.int32), data_row=np.where(leaf, c, -1).astype(np.int32),
                parent_pos=np.where(pa_root, -1, np.where(leaf, 0, p)).astype(np.int32),
                length=tree.length[c].astype(np.float64), gamma=np.ones(N - 1), color=col)

def bm_loglik_pruning(tree: Tree, R: np.ndarray, mu: np.ndarray, Y: np.ndarray) -> float:
    """Independent O(n p^3) check: Felsenstein-style pruning for BM with a fixed root
    (no shared code with the engine): each subtree is summarised as N(x_hat, v R) x const."""
    N, p = tree.nnodes, len(mu)
    Rinv = np.linalg.inv(R)
    sign, logdetR = np.linalg.slogdet(R)
    xh = np.zeros((N, p))
    v = np.zeros(N)           # variance scale of the estimate at each node (0 at tips)
    lg = np.zeros(N)          # accumulated log constant
    have = np.zeros(N, dtype=bool)
    xh[tree.is_leaf] = Y[tree.is_leaf]
    have[tree.is_leaf] = True
    # accumulate children into parents in reverse preorder
    prec = np.zeros(N)
    wsum = np.zeros((N, p))
    for i in range(N - 1, 0, -1):
        if not tree.is_leaf[i]:
            # finalise node i from its children
            v[i] = 1.0 / prec[i]
            xh[i] = wsum[i] * v[i]
        pa = tree.parent[i]
        vi = v[i] + tree.length[i]
        # product of the running estimate at pa (prec[pa], mean m) with N(x; xh_i, vi R)
        if prec[pa] == 0:
            prec[pa] = 1.0 / vi
            wsum[pa] = xh[i] / vi
            lg[pa] += lg[i]
        else:
            m_old = wsum[pa] / prec[pa]
            v_old = 1.0 / prec[pa]
            d = xh[i] - m_old
            s2 = v_old + vi
            lg[pa] += lg[i] - 0.5 * (p * LOG2PI + p * np.log(s2) + logdetR + (d @ Rinv @ d) / s2)
            prec[pa] += 1.0 / vi
            wsum[pa] += xh[i] / vi
    v0 = 1.0 / prec[0]
    x0 = wsum[0] * v0
    d = x0 - mu
    return float(lg[0] - 0.5 * (p * LOG2PI + p * np.log(v0) + logdetR + (d @ Rinv @ d) / v0))


def simulate_bm_uni_sites(tree: Tree, sigma2: np.ndarray, mu: np.ndarray, rng: np.random.Generator) -> np.ndarray:
    """Univariate BM at every node for many independent sites at once: (n_sites, N)."""
    ns = len(sigma2)
    z = rng.standard_normal((ns, tree.nnodes)) * np.sqrt(sigma2)[:, None] * np.sqrt(tree.length)[None, :]
    x = np.empty((ns, tree.nnodes))
    x[:, 0] = mu
    for i in range(1, tree.nnodes):
        x[:, i] = x[:, tree.parent[i]] + z[:, i]
    return x


def bm_loglik_pruning_uni_sites(tree: Tree, sigma2: np.ndarray, mu: np.ndarray, X: np.ndarray) -> np.ndarray:
    """bm_loglik_pruning for univariate sites, vectorised over sites: (n_sites,) log-likelihoods."""
    N = tree.nnodes
    ns = X.shape[0]
    xh = np.where(tree.is_leaf[None, :], X, 0.0)
    v = np.zeros(N)
    lg = np.zeros((ns, N))
    prec = np.zeros(N)
    wsum = np.zeros((ns, N))
    first = np.ones(N, dtype=bool)
    logs2 = np.log(sigma2)
    for i in range(N - 1, 0, -1):
        if not tree.is_leaf[i]:
            v[i] = 1.0 / prec[i]
            xh[:, i] = wsum[:, i] * v[i]
        pa = tree.parent[i]
        vi = v[i] + tree.length[i]
        if first[pa]:
            first[pa] = False
            prec[pa] = 1.0 / vi
            wsum[:, pa] = xh[:, i] / vi
            lg[:, pa] += lg[:, i]
        else:
            m_old = wsum[:, pa] / prec[pa]
            s2 = 1.0 / prec[pa] + vi
            d = xh[:, i] - m_old
            lg[:, pa] += lg[:, i] - 0.5 * (LOG2PI + np.log(s2) + logs2 + d * d / (sigma2 * s2))
            prec[pa] += 1.0 / vi
            wsum[:, pa] += xh[:, i] / vi
    v0 = 1.0 / prec[0]
    d = wsum[:, 0] * v0 - mu
    return lg[:, 0] - 0.5 * (LOG2PI + np.log(v0) + logs2 + d * d / (sigma2 * v0))



def simulate_ou_uni_sites(tree: Tree, sigma2, alpha, theta, mu, rng: np.random.Generator) -> np.ndarray:
    """Univariate Ornstein-Uhlenbeck process (src/evomodels/homogeneousornsteinuhlenbeck.jl:59-66) at every node for
    many independent sites at once, one (sigma2, alpha, theta, mu) per site, root fixed at mu: (n_sites, N)."""
    ns = len(sigma2)
    gam2 = sigma2 / (2.0 * alpha)
    z = rng.standard_normal((ns, tree.nnodes))
    x = np.empty((ns, tree.nnodes))
    x[:, 0] = mu
    for i in range(1, tree.nnodes):
        a = np.exp(-alpha * tree.length[i])
        x[:, i] = a * x[:, tree.parent[i]] + (1.0 - a) * theta + np.sqrt(gam2 * (1.0 - a * a)) * z[:, i]
    return x


def ou_loglik_pruning_uni_sites(tree: Tree, sigma2, alpha, theta, mu, X: np.ndarray) -> np.ndarray:
    """Independent check for the OU site batches: pruning in scalar canonical form, vectorised over sites.
    Node i holds L_i(x_i) = exp(-A x^2 / 2 + B x + C), the likelihood of the tips below it; an edge
    x_i | x_pa ~ N(q x_pa + w, V) turns it into a function of x_pa.  (n_sites,) log-likelihoods."""
    N = tree.nnodes
    ns = X.shape[0]
    gam2 = sigma2 / (2.0 * alpha)
    A = np.zeros((N, ns))
    B = np.zeros((N, ns))
    Cc = np.zeros((N, ns))
    for i in range(N - 1, 0, -1):
        pa = tree.parent[i]
        q = np.exp(-alpha * tree.length[i])
        w = (1.0 - q) * theta
        V = gam2 * (1.0 - q * q)
        if tree.is_leaf[i]:
            r = X[:, i] - w
            A[pa] += q * q / V
            B[pa] += q * r / V
            Cc[pa] += -0.5 * (r * r / V + LOG2PI + np.log(V))
        else:
            P = A[i] + 1.0 / V
            sc = 1.0 / (1.0 + A[i] * V)
            A[pa] += A[i] * sc * q * q
            B[pa] += q * sc * (B[i] - A[i] * w)
            Cc[pa] += sc * (B[i] * w - 0.5 * A[i] * w * w) + 0.5 * B[i] * B[i] / P + Cc[i] - 0.5 * np.log1p(A[i] * V)
    return -0.5 * A[0] * mu * mu + B[0] * mu + Cc[0]

def bethe_of_tree(tree: Tree, p: int) -> Problem:
    """Bethe cluster graph of a tree (src/clustergraph.jl:473-527): factor cluster {c, parent(c)}
    for every non-root node (ids 0..N-2, as in the clique tree) then one variable cluster {v} for
    every internal node v, in decreasing preorder (postorder), each joined to the factor clusters
    that contain v.  It is a tree; the schedule is its DFS from the variable cluster of the root."""
    N = tree.nnodes
    par, leaf = tree.parent, tree.is_leaf
    c = np.arange(1, N)
    dc = np.where(leaf[c], 0, p)
    dpa = np.where(par[c] == 0, 0, p)
    fdims = dc + dpa
    internal = np.nonzero(~leaf)[0][::-1]          # decreasing preorder
    vid = {int(v): (N - 1) + i for i, v in enumerate(internal)}
    vdims = np.where(internal == 0, 0, p)
    children = [[] for _ in range(N)]
    for i in range(1, N):
        children[par[i]].append(i)
    sep_a, sep_b, sdim, starts = [], [], [], []
    pa_j, ch_j = [], []
    # DFS from variable cluster {root}; edges: {v} - family(child k) for each child k, then
    # family(k) - {k} if k internal
    stack = [0]
    while stack:
        v = stack.pop()
        for k in children[v]:
            d = 0 if v == 0 else p
            # sepset {v} between variable cluster {v} (side a) and factor cluster (k, v) (side b)
            sep_a.append(vid[v]); sep_b.append(k - 1); sdim.append(d)
            starts += [0, int(dc[k - 1])]
            pa_j.append(vid[v]); ch_j.append(k - 1)
            if not leaf[k]:
                # sepset {k} between factor cluster (k, v) (side a) and variable cluster {k} (side b)
                sep_a.append(k - 1); sep_b.append(vid[k]); sdim.append(p)
                starts += [0, 0]
                pa_j.append(k - 1); ch_j.append(vid[k])
        # preorder DFS: visit children in order -> emit edges depth-first
        # (edges above are emitted breadth-wise per node; reorder below)
        for k in reversed(children[v]):
            if not leaf[k]:
                stack.append(k)
    # the edge list above is not in DFS preorder of the cluster tree; rebuild it properly
    sep_index = {(a, b): i for i, (a, b) in enumerate(zip(sep_a, sep_b))}
    nb = {}
    for a, b in zip(sep_a, sep_b):
        nb.setdefault(a, []).append(b)
        nb.setdefault(b, []).append(a)
    rootc = vid[0]
    pa_j, ch_j = [], []
    seen = {rootc}
    st = [(rootc, iter(nb.get(rootc, [])))]
    while st:
        v, it = st[-1]
        adv = False
        for u in it:
            if u not in seen:
                seen.add(u)
                pa_j.append(v); ch_j.append(u)
                st.append((u, iter(nb.get(u, []))))
                adv = True
                break
        if not adv:
            st.pop()
    sdim = np.array(sdim, dtype=np.int64)
    dims = np.concatenate([fdims, vdims, sdim]).astype(np.int32)
    lens = np.repeat(sdim, 2)
    scope_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    starts = np.array(starts, dtype=np.int64)
    scope_idx = (np.repeat(starts, lens) + (np.arange(lens.sum()) - np.repeat(scope_off[:-1], lens))).astype(np.int32)
    prob = Problem(dims=dims, sepset_clusters=np.stack([sep_a, sep_b], axis=1).astype(np.int32),
                   scope_off=scope_off, scope_idx=scope_idx,
                   schedule=[(np.array(pa_j, np.int32), np.array(ch_j, np.int32))],
                   nclusters=(N - 1) + len(internal), root_cluster=rootc)
    prob.packed_off = _packed_offsets(dims)
    prob.meta = {"graph": "bethe", "ntips": tree.ntips, "p": p}
    return prob


def bm_factors_bethe(tree: Tree, prob: Problem, R, mu, Y):
    """Bethe factors: the factor clusters hold the same (J,h,g) as the cliques; variable clusters = 1."""
    ct = cliquetree_of_tree(tree, len(mu))
    pk = bm_factors_cliquetree(tree, ct, R, mu, Y)
    n_f = tree.nnodes - 1
    packed = np.zeros(int(prob.packed_off[-1]))
    packed[: ct.packed_off[n_f]] = pk[: ct.packed_off[n_f]]
    return packed
