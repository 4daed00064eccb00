"""Synthetic admixture networks and scope allocation on plain arrays, for the loopy-network configuration
(BASELINE.json configs[4]) at sizes where per-node Python objects and quadratic searches are out of the question.

A network is held as a `NetArrays`: nodes numbered in preorder (1-based labels as in the reference: every node
after all of its parents, root = 1), each with its node family [child, parents by decreasing label]
(nodefamilies, src/clustergraph.jl:136-146) and, aligned with the parents, edge length, inheritance and colour."""
from dataclasses import dataclass
from typing import List, Sequence

import numpy as np

from .synth import Tree, random_tree


@dataclass
class NetArrays:
    node2family: List[List[int]]     # [child, parents...] 1-based preorder labels, parents decreasing
    length: List[List[float]]        # per parent edge, aligned with node2family[i][1:]
    gamma: List[List[float]]
    color: List[List[int]]           # 0-based rate index of each parent edge
    is_leaf: np.ndarray              # (N,) bool, index = label - 1

    @property
    def nnodes(self):
        return len(self.node2family)

    @property
    def nhybrids(self):
        return sum(1 for nf in self.node2family if len(nf) > 2)


def random_level3_network(ntips: int, nblobs: int, rng: np.random.Generator, n_colors: int = 1, lo=0.1, hi=1.0) -> NetArrays:
    """Random rooted level-3 network: a random bifurcating tree (uniform random joins, lengths U(lo, hi)) in which
    `nblobs` internal nodes w with children c1, c2 are replaced by the level-3 blob of the reference's own test
    network (test/test_calibration.jl:132, "((#H1,#H2)I1,(((A)#H1,#H3)#H2,(B)#H3)I2)I3"):
      w -> I1, I2;  I1 -> H1, H2 (minor);  I2 -> H2, H3 (major);  H2 -> H1 (major), H3 (minor);  H1 -> c1;  H3 -> c2.
    3 reticulations per blob, blobs separated by cut edges.  Minor inheritance ~ U(0.1, 0.5), hybrid edge lengths
    U(0.05, 0.3), every length > 0; each edge gets a colour in [0, n_colors)."""
    tr: Tree = random_tree(ntips, rng, lo, hi)
    N0 = tr.nnodes
    internal = np.nonzero(~tr.is_leaf)[0]
    picks = set(int(x) for x in rng.permutation(internal)[:nblobs])
    # directed acyclic graph on temporary ids: parents[v] = [(parent id, length, gamma)]
    parents: List[list] = [[] for _ in range(N0)]
    children: List[list] = [[] for _ in range(N0)]
    is_leaf = list(tr.is_leaf)
    kids = [[] for _ in range(N0)]
    for v in range(1, N0):
        kids[tr.parent[v]].append(v)

    def new_node():
        parents.append([])
        children.append([])
        is_leaf.append(False)
        return len(parents) - 1

    def link(pa, ch, t, g=1.0):
        parents[ch].append((pa, float(t), float(g)))
        children[pa].append(ch)

    for w in range(N0):
        if w in picks and len(kids[w]) == 2:
            c1, c2 = kids[w]
            if rng.random() < 0.5:
                c1, c2 = c2, c1
            i1, i2, h1, h2, h3 = (new_node() for _ in range(5))
            g1, g2, g3 = rng.uniform(0.1, 0.5, size=3)
            t = rng.uniform(0.05, 0.3, size=6)
            link(w, i1, rng.uniform(lo, hi)); link(w, i2, rng.uniform(lo, hi))
            link(i1, h1, t[0], g1); link(h2, h1, t[1], 1.0 - g1)
            link(i1, h2, t[2], g2); link(i2, h2, t[3], 1.0 - g2)
            link(h2, h3, t[4], g3); link(i2, h3, t[5], 1.0 - g3)
            link(h1, c1, tr.length[c1]); link(h3, c2, tr.length[c2])
        else:
            for c in kids[w]:
                link(w, c, tr.length[c])
    # preorder: a node is listed once all of its parents are (depth-first, children in creation order)
    n = len(parents)
    indeg = [len(p) for p in parents]
    order, stack = [], [0]
    while stack:
        v = stack.pop()
        order.append(v)
        for c in reversed(children[v]):
            indeg[c] -= 1
            if indeg[c] == 0:
                stack.append(c)
    assert len(order) == n
    label = np.empty(n, dtype=np.int64)
    label[np.array(order)] = np.arange(1, n + 1)
    fam, ln, gm, col = [], [], [], []
    for v in order:
        ps = sorted(parents[v], key=lambda q: -label[q[0]])
        fam.append([int(label[v])] + [int(label[q[0]]) for q in ps])
        ln.append([q[1] for q in ps])
        gm.append([q[2] for q in ps])
        col.append([int(x) for x in rng.integers(0, n_colors, size=len(ps))])
    return NetArrays(fam, ln, gm, col, np.array([is_leaf[v] for v in order], dtype=bool))


def random_level3_network_varied(ntips: int, nret: int, rng: np.random.Generator, n_colors: int = 1, lo=0.1, hi=1.0,
                                  max_level: int = 3) -> NetArrays:
    """Random rooted network of level <= max_level with about `nret` reticulations in VARIED blobs (BASELINE.json
    configs[4]).  A random bifurcating tree (uniform random joins); disjoint blob sites = an internal node with the
    edges of its subtree down to depth 3; inside a site, 1 .. max_level reticulations are drawn between random pairs of
    its (progressively subdivided) edges: the source edge gets a new tree node, the target edge a new hybrid node (minor
    inheritance ~ U(0.1, 0.5), all lengths > 0), rejected if it would close a directed cycle.  The reticulations of a
    site may interlock into one blob (level up to max_level; moral graphs with cliques of 4 nodes, treewidth 3) or stay
    apart; sites are separated by cut edges, so no blob has more than max_level hybrids."""
    tr: Tree = random_tree(ntips, rng, lo, hi)
    N0 = tr.nnodes
    parents: List[list] = [[] for _ in range(N0)]     # [(parent id, length, gamma)]
    children: List[list] = [[] for _ in range(N0)]
    is_leaf = list(tr.is_leaf)
    for v in range(1, N0):
        parents[v].append((int(tr.parent[v]), float(tr.length[v]), 1.0))
        children[int(tr.parent[v])].append(v)

    def new_node():
        parents.append([])
        children.append([])
        is_leaf.append(False)
        return len(parents) - 1

    def subdivide(pa, ch):
        """new node on the edge pa -> ch; returns it"""
        k = next(i for i, q in enumerate(parents[ch]) if q[0] == pa)
        _, t, g = parents[ch][k]
        cut = float(rng.uniform(0.25, 0.75))
        x = new_node()
        parents[x].append((pa, t * cut, g))
        parents[ch][k] = (x, t * (1.0 - cut), 1.0)
        children[pa][children[pa].index(ch)] = x
        children[x].append(ch)
        return x

    def reaches(a, b, limit=4096):
        """is b a descendant of a (inside the small site)?"""
        st, seen = [a], 0
        while st and seen < limit:
            x = st.pop()
            if x == b:
                return True
            seen += 1
            st.extend(children[x])
        return False

    used = np.zeros(N0, dtype=bool)
    done = 0
    for w in rng.permutation(np.nonzero(~tr.is_leaf)[0]):
        if done >= nret:
            break
        w = int(w)
        # the site: edges of the subtree of w down to depth 3, all of whose nodes are still free
        nodes, frontier, edges = [w], [w], []
        for _ in range(3):
            nxt = []
            for x in frontier:
                for c in children[x]:
                    if c < N0:
                        edges.append((x, c))
                        nodes.append(c)
                        nxt.append(c)
            frontier = nxt
        if len(edges) < 4 or used[nodes].any():
            continue
        used[nodes] = True
        want = int(rng.integers(1, max_level + 1))
        site_edges = list(edges)
        for _ in range(want):
            for _attempt in range(20):
                i, j = rng.choice(len(site_edges), size=2, replace=False)
                (p1, c1), (p2, c2) = site_edges[i], site_edges[j]
                if len(parents[c2]) > 1 or len(parents[c1]) > 1:      # keep every hybrid node at two parents
                    continue
                if c2 == c1 or reaches(c2, p1) or reaches(c2, c1):     # the hybrid would sit above its new parent
                    continue
                s_node = subdivide(p1, c1)
                h_node = subdivide(p2, c2)
                g = float(rng.uniform(0.1, 0.5))
                k = next(q for q, e in enumerate(parents[h_node]) if e[0] == p2)
                parents[h_node][k] = (p2, parents[h_node][k][1], 1.0 - g)
                parents[h_node].append((s_node, float(rng.uniform(0.05, 0.3)), g))
                children[s_node].append(h_node)
                site_edges[i] = (p1, s_node); site_edges.append((s_node, c1))
                site_edges[j] = (p2, h_node); site_edges.append((h_node, c2)); site_edges.append((s_node, h_node))
                done += 1
                break
    n = len(parents)
    indeg = [len(q) for q in parents]
    order, stack = [], [0]
    while stack:
        v = stack.pop()
        order.append(v)
        for c in reversed(children[v]):
            indeg[c] -= 1
            if indeg[c] == 0:
                stack.append(c)
    assert len(order) == n, "the generated network has a directed cycle"
    label = np.empty(n, dtype=np.int64)
    label[np.array(order)] = np.arange(1, n + 1)
    fam, ln, gm, col = [], [], [], []
    for v in order:
        ps = sorted(parents[v], key=lambda q: -label[q[0]])
        fam.append([int(label[v])] + [int(label[q[0]]) for q in ps])
        ln.append([q[1] for q in ps])
        gm.append([q[2] for q in ps])
        col.append([int(x) for x in rng.integers(0, n_colors, size=len(ps))])
    return NetArrays(fam, ln, gm, col, np.array([is_leaf[v] for v in order], dtype=bool))


def simulate_bm_network(net: NetArrays, rates: Sequence[np.ndarray], mu: np.ndarray, rng: np.random.Generator) -> np.ndarray:
    """Trait values at every node (N, p) under a heterogeneous BM on the network (weighted-average merging at hybrid
    nodes: src/evomodels/evomodels.jl:314-330), root fixed at mu."""
    p = len(mu)
    chol = [np.linalg.cholesky(np.asarray(R, float)) for R in rates]
    x = np.zeros((net.nnodes, p))
    x[0] = mu
    for i in range(1, net.nnodes):
        nf = net.node2family[i]
        acc = np.zeros(p)
        for k, pl in enumerate(nf[1:]):
            g, t, c = net.gamma[i][k], net.length[i][k], net.color[i][k]
            acc += g * (x[pl - 1] + np.sqrt(t) * (chol[c] @ rng.standard_normal(p)))
        x[i] = acc
    return x


@dataclass
class ScopeTables:
    """What allocatebeliefs (src/beliefs.jl:478-594) establishes for complete data: belief dimensions, scope index
    maps of both ends of every sepset (the arrays pgbp_desc takes), the cluster of every node family, and light
    cluster records (nodelabel, inscope) for factors.lg_families."""
    dims: np.ndarray
    sepset_clusters: np.ndarray
    scope_off: np.ndarray
    scope_idx: np.ndarray
    node2cluster: List[int]
    node2fixed: np.ndarray
    clusters: list


class _Scope:
    __slots__ = ("nodelabel", "inscope")

    def __init__(self, nodelabel, inscope):
        self.nodelabel, self.inscope = nodelabel, inscope


def allocate_scopes(cluster_nodes, edges, sepset_nodes, net: NetArrays, p: int, fixedroot: bool = True, data=None,
                    data_row=None) -> ScopeTables:
    """allocatebeliefs (src/beliefs.jl:478-594) on plain arrays: tips (and a fixed root) are out of scope; an internal node
    has trait t in scope iff some tip below it has a value for t (:509-520, 551-559) -- every trait without `data`;
    node2cluster[ni] = the first cluster that contains the family of node ni + 1 (:521-527); scopeindex(sepset, cluster)
    (:389-405) for both ends of every sepset.  data: [n_rows, p] tip values, NaN where missing; data_row[ni]: the row of
    tip ni (as for factors.lg_families)."""
    N = net.nnodes
    fixed = net.is_leaf.copy()
    if fixedroot:
        fixed[0] = True
    has = np.ones((N, p), dtype=bool)
    if data is not None:
        data = np.asarray(data, float)
        has[:] = False
        for i in range(N - 1, -1, -1):          # children before parents
            if net.is_leaf[i]:
                has[i] = np.isfinite(data[int(data_row[i])])
            for pa in net.node2family[i][1:]:
                has[pa - 1] |= has[i]
    ins = has & ~fixed[:, None]                  # (N, p)
    holders: List[List[int]] = [[] for _ in range(N + 1)]
    for ci, nodes in enumerate(cluster_nodes):
        for v in nodes:
            holders[v].append(ci)
    node2cluster = []
    csets = [None] * len(cluster_nodes)
    for ni, nf in enumerate(net.node2family):
        found = None
        for ci in holders[nf[0]]:                      # increasing cluster index = the reference's findfirst
            if csets[ci] is None:
                csets[ci] = set(cluster_nodes[ci])
            if all(v in csets[ci] for v in nf[1:]):
                found = ci
                break
        if found is None:
            raise ValueError(f"no cluster containing the node family of node {ni + 1}")
        node2cluster.append(found)
    ndim = ins.sum(axis=1)
    cdims = np.array([int(sum(ndim[v - 1] for v in nodes)) for nodes in cluster_nodes], dtype=np.int32)
    sdims = np.array([int(sum(ndim[v - 1] for v in nodes)) for nodes in sepset_nodes], dtype=np.int32)
    start = []
    for nodes in cluster_nodes:
        s, acc = {}, 0
        for v in nodes:
            s[v] = acc
            acc += int(ndim[v - 1])
        start.append(s)
    off, idx = [0], []
    for (a, b), nodes in zip(edges, sepset_nodes):
        for c in (a, b):
            cn = cluster_nodes[c]
            where = [cn.index(v) for v in nodes]         # raises ValueError if not a subset
            if any(y <= x for x, y in zip(where, where[1:])):
                raise ValueError("subset labels come in a different order in the belief")
            for v in nodes:
                idx.extend(range(start[c][v], start[c][v] + int(ndim[v - 1])))
            off.append(len(idx))
    clusters = [_Scope(list(nodes), ins[np.array(nodes) - 1].T.copy()) for nodes in cluster_nodes]
    return ScopeTables(np.concatenate([cdims, sdims]).astype(np.int32), np.array(edges, np.int32).reshape(-1, 2),
                       np.array(off, np.int64), np.array(idx, np.int32), node2cluster, fixed, clusters)


# ---------------------------------------------------------------------------------------------------------------
# data formats on the caller's side of the path: extended Newick (the reference reads networks with
# PhyloNetworks.readnewick: test/example_networks/*.phy, docs/src/man/getting_started.md:30-60)
# ---------------------------------------------------------------------------------------------------------------

def read_newick(text: str, prefix: str = "I", order: str = "phylonetworks"):
    """Rooted network from an extended Newick string: `#Hk` marks the two (or more) appearances of hybrid node k,
    `:length:support:gamma` follows a node.  A missing inheritance gamma is the complement of the given one (0.5 each if
    none is given).  A root of degree one is suppressed (as PhyloNetworks does); unnamed internal nodes are named
    `prefix`1, `prefix`2, ... in the order in which their subtrees close in the string (preprocessnet!,
    src/clustergraph.jl:18-37).  Returns (NetArrays, names): nodes in a preorder (every node after all of its parents,
    root first), `names[i]` the name of the node labelled i + 1; every edge gets colour 0.
    order = "phylonetworks": the node ordering of PhyloNetworks.preorder!, which the reference's cluster-graph builders
    break their ties with (src/clustergraph.jl:87-121): a stack; a node's children are pushed in file order, so the LAST
    child is visited first; a hybrid node is pushed by the parent that is visited last.  (It reproduces the node
    numberings that the reference's tests and doctests imply: tests/golden/reference_goldens.json.)
    order = "file": depth first with the first child first."""
    s = "".join(text.split())
    pos = 0
    parents: List[list] = []     # per temporary id: [(parent id, length, gamma or None)]
    kids: List[list] = []
    names: List[str] = []
    hybrid_id = {}

    def new(name):
        parents.append([]); kids.append([]); names.append(name)
        return len(names) - 1

    def number():
        nonlocal pos
        a = pos
        while pos < len(s) and s[pos] not in ":,();":
            pos += 1
        return float(s[a:pos]) if pos > a else None

    def subtree():
        """-> (node id, length, gamma) of the branch above the subtree that starts at `pos`"""
        nonlocal pos
        below = []
        if s[pos] == "(":
            pos += 1
            while True:
                below.append(subtree())
                if s[pos] == ",":
                    pos += 1
                    continue
                if s[pos] == ")":
                    pos += 1
                    break
                raise ValueError(f"unexpected {s[pos]!r} at position {pos}")
        a = pos
        while pos < len(s) and s[pos] not in ":,();":
            pos += 1
        name = s[a:pos]
        vals = []
        while pos < len(s) and s[pos] == ":":
            pos += 1
            vals.append(number())
        if name.startswith("#"):
            key = name[1:]
            if key not in hybrid_id:
                hybrid_id[key] = new(key)
            v = hybrid_id[key]
        else:
            v = new(name)
        for (c, t, g) in below:
            parents[c].append([v, float("nan") if t is None else t, g])   # no length given: NaN (the factor fill refuses it)
            kids[v].append(c)
        return v, (vals[0] if vals else None), (vals[2] if len(vals) > 2 else None)

    root, _, _ = subtree()
    # inheritance of hybrid edges
    for v, ps in enumerate(parents):
        if len(ps) >= 2:
            given = [q[2] for q in ps if q[2] is not None]
            rest = [q for q in ps if q[2] is None]
            for q in rest:
                q[2] = (1.0 - sum(given)) / len(rest) if given else 1.0 / len(ps)
        elif ps:
            ps[0][2] = 1.0
    while len(kids[root]) == 1 and len(parents[kids[root][0]]) == 1 and kids[kids[root][0]]:
        nxt = kids[root][0]          # degree-one root above a tree node: suppressed
        parents[nxt] = []
        kids[root] = []
        root = nxt
    indeg = [len(p) for p in parents]
    walk = order
    order, stack = [], [root]
    while stack:
        v = stack.pop()
        order.append(v)
        for c in (kids[v] if walk == "phylonetworks" else reversed(kids[v])):
            indeg[c] -= 1
            if indeg[c] == 0:
                stack.append(c)
    label = {v: i + 1 for i, v in enumerate(order)}
    # unnamed internal nodes: numbered in the order of their temporary ids = the order in which their subtrees close
    used = {nm for nm in names if nm}
    k = 1
    for v in sorted(label):
        if not names[v]:
            while f"{prefix}{k}" in used:
                k += 1
            names[v] = f"{prefix}{k}"
            used.add(names[v])
    out_names = [names[v] for v in order]
    fam, ln, gm, col = [], [], [], []
    for v in order:
        ps = sorted(parents[v], key=lambda q: -label[q[0]])
        fam.append([label[v]] + [label[q[0]] for q in ps])
        ln.append([float(q[1]) for q in ps])
        gm.append([float(q[2]) for q in ps])
        col.append([0] * len(ps))
    return NetArrays(fam, ln, gm, col, np.array([not kids[v] for v in order], dtype=bool)), out_names
