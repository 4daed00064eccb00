"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Cluster-graph construction and message schedules for small networks, after
src/clustergraph.jl.  The reference builds these with Graphs.jl/MetaGraphsNext
(maximal_cliques, kruskal_mst, dfs_parents): the *algorithms* are restated; tie
breaking and vertex numbering of those libraries are not (they change which of
several equally valid clique trees / spanning trees is produced, not any
log-likelihood or calibrated marginal).
"""
from __future__ import annotations

import itertools
from typing import List

from .beliefs import ClusterGraph


def moralize(net):
    """src/clustergraph.jl:43-77: undirected adjacency over 1-based preorder indices."""
    pre = net.vec_node
    pos = {id(n): i + 1 for i, n in enumerate(pre)}
    adj = {i: set() for i in range(1, len(pre) + 1)}
    for e in net.edges:
        a, b = pos[id(e.parent)], pos[id(e.child)]
        adj[a].add(b)
        adj[b].add(a)
    for n in pre:
        if n.hybrid:
            ps = [pos[id(p)] for p in net.parents(n)]
            for a, b in itertools.combinations(ps, 2):
                adj[a].add(b)
                adj[b].add(a)
    return adj


def triangulate_minfill(adj):
    """src/clustergraph.jl:87-121: greedy min-fill elimination; ties broken by
    post-ordering (largest preorder index first).  Adds fill edges to `adj`,
    returns the elimination ordering."""
    g2 = {k: set(v) for k, v in adj.items()}
    ordering = []

    def fill_edges(v):
        nb = sorted(g2[v])
        return [(a, b) for a, b in itertools.combinations(nb, 2) if b not in g2[a]]

    while len(g2) > 1:
        v = min(g2.keys(), key=lambda x: (len(fill_edges(x)), -x))
        for a, b in fill_edges(v):
            g2[a].add(b); g2[b].add(a)
            adj[a].add(b); adj[b].add(a)
        ordering.append(v)
        for u in g2[v]:
            g2[u].discard(v)
        del g2[v]
    ordering.append(next(iter(g2.keys())))
    return ordering


def maximal_cliques_chordal(adj, ordering):
    """Maximal cliques of a chordal graph from a perfect elimination ordering."""
    posn = {v: i for i, v in enumerate(ordering)}
    cands = []
    for v in ordering:
        later = {u for u in adj[v] if posn[u] > posn[v]}
        cands.append(frozenset(later | {v}))
    cliques = []
    for c in cands:
        if not any(c < d for d in cands):
            if c not in cliques:
                cliques.append(c)
    return cliques


def _kruskal(nv, weighted_edges, maximize=True):
    parent = list(range(nv))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    out = []
    for (w, a, b, dat) in sorted(weighted_edges, key=lambda t: (-t[0] if maximize else t[0], t[1], t[2])):
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[ra] = rb
            out.append((a, b, dat))
    return out


def cliquetree(net) -> ClusterGraph:
    """src/clustergraph.jl:452-466 + :757-820: moralize, min-fill triangulate,
    maximal cliques, maximum-weight spanning tree on sepset size."""
    adj = moralize(net)
    ordering = triangulate_minfill(adj)
    cliques = maximal_cliques_chordal(adj, ordering)
    names = [n.name for n in net.vec_node]
    clusters = []
    for c in cliques:
        nodes = sorted(c, reverse=True)
        clusters.append(("".join(names[i - 1] for i in nodes), nodes))
    wedges = []
    for i, j in itertools.combinations(range(len(clusters)), 2):
        sep = sorted(set(clusters[i][1]) & set(clusters[j][1]), reverse=True)
        if sep:
            wedges.append((len(sep), i, j, sep))
    mst = _kruskal(len(clusters), wedges, maximize=True)
    return ClusterGraph(clusters, [(a, b, sep) for a, b, sep in mst], "cliquetree")


def bethe(net) -> ClusterGraph:
    """src/clustergraph.jl:473-527: one factor cluster per node family (merged into a
    child's family when it is a subset of it), one variable cluster per node that
    sits in more than one factor cluster."""
    pre = net.vec_node
    names = [n.name for n in pre]
    pos = {id(n): i + 1 for i, n in enumerate(pre)}
    clusters = []
    node2code = {}
    node2cluster = {}
    for noi in reversed(range(len(pre))):
        n = pre[noi]
        fam = [noi + 1] + sorted((pos[id(p)] for p in net.parents(n)), reverse=True)
        if len(fam) == 1:
            continue
        sub = False
        for ch in net.children(n):
            cc = node2code[pos[id(ch)]]
            if set(fam) <= set(clusters[cc][1]):
                node2code[noi + 1] = cc
                sub = True
                break
        if sub:
            continue
        code = len(clusters)
        node2code[noi + 1] = code
        clusters.append(("".join(names[i - 1] for i in fam), fam))
        for ni in fam:
            node2cluster.setdefault(ni, []).append(code)
    edges = []
    for ni in sorted(node2cluster.keys(), reverse=True):
        cl = node2cluster[ni]
        if len(cl) <= 1:
            continue
        vcode = len(clusters)
        clusters.append((names[ni - 1], [ni]))
        for fc in cl:
            edges.append((vcode, fc, [ni]))
    return ClusterGraph(clusters, edges, "Bethe")


def default_rootcluster(cg: ClusterGraph, net) -> int:
    """src/clustergraph.jl:1022-1029 (0-based)."""
    pre = net.vec_node
    best, bestscore = None, None
    for k, (_, nodes) in enumerate(cg.clusters):
        if 1 in nodes:
            score = sum(1 for i in nodes if pre[i - 1].leaf)
            if bestscore is None or score < bestscore:
                best, bestscore = k, score
    if best is None:
        raise ValueError("no cluster contains the root")
    return best



def _graphs_jl_tree_order(n, nbrs, root):
    """Literal restatement of the two Graphs.jl routines behind `spanningtree_clusterlist` (src/clustergraph.jl:885-894;
    Graphs.jl is a dependency of the reference, not vendored: algorithms as published in Graphs.jl 1.x,
    src/traversals/dfs.jl).  (The product's version, pgbp_amd/clustergraph.py, keeps a scan pointer per vertex instead
    of rescanning: the two are written independently and compared by tests/test_plan_cpu.py.)

    dfs_parents(g, root) -> tree_dfs: `S = [root]; parents[root] = root; while S: v = S[end]; u = 0;
    for n in outneighbors(g, v): if !seen[n]: u = n; break;  if u == 0: pop!(S) else: seen[u] = true; push!(S, u);
    parents[u] = v`.
    topological_sort_by_dfs(tree(parents)): for v in vertices: if vcolor[v] == 0: S = [v]; vcolor[v] = 1; while S:
    u = S[end]; w = first out-neighbour with vcolor == 0 (a neighbour with vcolor == 1 is a cycle: error);
    if w: vcolor[w] = 1; push!(S, w) else: vcolor[u] = 2; push!(verts, u); pop!(S);  return reverse(verts).
    nbrs[v]: neighbours in increasing code (Graphs.jl adjacency lists are sorted).  Returns (parents, order after root)."""
    seen = set([root])
    parents = {root: root}
    S = [root]
    while S:
        v = S[-1]
        u = None
        for cand in nbrs[v]:                 # rescanned from the start every time, as the library does
            if cand not in seen:
                u = cand
                break
        if u is None:
            S.pop()
        else:
            seen.add(u)
            S.append(u)
            parents[u] = v
    # tree(parents): directed graph with an edge parents[v] -> v for v != root
    out = {v: [] for v in range(n)}
    for v in sorted(parents):
        if parents[v] != v:
            out[parents[v]].append(v)
    vcolor = {v: 0 for v in range(n)}
    verts = []
    for v in range(n):
        if vcolor[v] != 0 or v not in parents:   # (vertices outside the component: isolated in tree(parents))
            continue
        S = [v]
        vcolor[v] = 1
        while S:
            u = S[-1]
            w = None
            for cand in out[u]:
                if vcolor[cand] == 1:
                    raise ValueError("The input graph contains at least one loop.")
                if vcolor[cand] == 0:
                    w = cand
                    break
            if w is not None:
                vcolor[w] = 1
                S.append(w)
            else:
                vcolor[u] = 2
                verts.append(u)
                S.pop()
    order = list(reversed(verts))
    assert order and order[0] == root
    return [parents.get(v, -1) for v in range(n)], order[1:]


def spanningtree_clusterlist(cg: ClusterGraph, rootj: int, edge_subset=None, vertices=None):
    """src/clustergraph.jl:885-894: depth-first spanning tree from cluster `rootj`, clusters listed as Graphs.jl lists
    them (see _graphs_jl_tree_order).  `vertices`: the vertex numbering of the (sub)graph the reference works on, as a
    list of cluster indices (induced_subgraph renumbers); default: all clusters in index order.
    Returns (pa_lab, ch_lab, pa_j, ch_j) with 0-based cluster indices of `cg`."""
    edges = cg.edges if edge_subset is None else edge_subset
    if vertices is None:
        vertices = list(range(len(cg.clusters)))
    code = {c: i for i, c in enumerate(vertices)}
    nbrs = [[] for _ in vertices]
    for (a, b, _) in edges:
        if a in code and b in code:
            nbrs[code[a]].append(code[b])
            nbrs[code[b]].append(code[a])
    for lst in nbrs:
        lst.sort()
    par, order = _graphs_jl_tree_order(len(vertices), nbrs, code[rootj])
    ch_j = [vertices[v] for v in order]
    pa_j = [vertices[par[v]] for v in order]
    labs = cg.labels
    return ([labs[i] for i in pa_j], [labs[i] for i in ch_j], pa_j, ch_j)


def spanningtrees_clusterlist(cg: ClusterGraph, net):
    """src/clustergraph.jl:908-937: spanning trees that together cover all edges.  Each is Graphs.jl's kruskal_mst on
    "number of earlier trees using the edge" (edges in lexicographic (smaller, larger) vertex order, stable sort by
    weight, stop at n - 1 edges); `induced_subgraph(cg, mst_edges)` renumbers the clusters in the order in which they first
    appear in that edge list, and the tree is rooted and listed in THAT numbering (default_rootcluster takes the first
    minimum; the depth-first search follows increasing new numbers)."""
    pre = net.vec_node
    nclu = len(cg.clusters)
    und = sorted(range(len(cg.edges)), key=lambda k: (min(cg.edges[k][:2]), max(cg.edges[k][:2])))
    used = [0] * len(cg.edges)
    sched = []
    while any(u == 0 for u in used):
        parent = list(range(nclu))

        def find(x):
            while parent[x] != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x
        mst = []
        for k in sorted(und, key=lambda k: used[k]):      # stable: ties keep the lexicographic edge order
            a, b = sorted(cg.edges[k][:2])
            ra, rb = find(a), find(b)
            if ra != rb:
                parent[ra] = rb
                mst.append(k)
                if len(mst) >= nclu - 1:
                    break
        vmap = []
        for k in mst:
            for v in sorted(cg.edges[k][:2]):
                if v not in vmap:
                    vmap.append(v)
        best, bestscore = None, None
        for v in vmap:                                     # default_rootcluster(sg, prenodes): first minimum in sg order
            nodes = cg.clusters[v][1]
            if 1 in nodes:
                score = sum(1 for i in nodes if pre[i - 1].leaf)
                if bestscore is None or score < bestscore:
                    best, bestscore = v, score
        if best is None:
            raise ValueError("no cluster contains the root")
        sched.append(spanningtree_clusterlist(cg, best, [cg.edges[k] for k in mst], vertices=vmap))
        for k in mst:
            used[k] += 1
    return sched


# ---------------------------------------------------------------------------
# join-graph structuring: src/clustergraph.jl:605-757
# ---------------------------------------------------------------------------

def nodefamilies(net):
    """src/clustergraph.jl:136-146: [child, parents sorted decreasing], 1-based preorder indices."""
    pre = net.vec_node
    pos = {id(n): i + 1 for i, n in enumerate(pre)}
    return [[i + 1] + sorted((pos[id(p)] for p in net.parents(n)), reverse=True) for i, n in enumerate(pre)]


def _assign(bucket, new, maxsize):
    """src/clustergraph.jl:705-736 assign!, statement by statement: `for sz in sort(collect(keys(bucket)), rev=true)`,
    `for (i, minibucket) in enumerate(bucket[sz])`: `merged = sort(union(new, minibucket))`; if it fits: deleteat! the
    old one (and the key when its vector empties), push the merged one under its own size, return (merged, old);
    else the new minibucket is filed under its size and (new, []) returned.  bucket: dict size -> list of minibuckets."""
    sizes = list(bucket.keys())
    sizes.sort()
    sizes.reverse()
    for sz in sizes:
        i = 0
        while i < len(bucket[sz]):
            old = bucket[sz][i]
            union = list(new)
            for x in old:
                if x not in union:
                    union.append(x)
            union.sort()
            if len(union) <= maxsize:
                del bucket[sz][i]
                if len(bucket[sz]) == 0:
                    del bucket[sz]
                if len(union) not in bucket:
                    bucket[len(union)] = []
                bucket[len(union)].append(union)
                return union, old
            i += 1
    if len(new) not in bucket:
        bucket[len(new)] = []
    bucket[len(new)].append(new)
    return new, []


def _julia_slot(k: int) -> int:
    """(Base.hash(::Int64) & 15) of Julia 1.x: the slot of key k in a fresh 16-slot Dict."""
    M = (1 << 64) - 1
    a = k & M
    a = (~a + (a << 21)) & M
    a ^= a >> 24
    a = (a + (a << 3) + (a << 8)) & M
    a ^= a >> 14
    a = (a + (a << 2) + (a << 4)) & M
    a ^= a >> 28
    a = (a + (a << 31)) & M
    return a & 15


def joingraph(net, maxclustersize, size_order="julia") -> ClusterGraph:
    """src/clustergraph.jl:605-697 (JoinGraphStructuring).  The reference walks the minibuckets of a bucket in the
    iteration order of a Julia Dict keyed by minibucket size (`values(bd)`, :645): size_order="julia" walks the sizes by
    their hash slot (the keys 1..10 fall into distinct slots of the initial 16-slot table), which reproduces the join
    graphs of the reference's doctests cluster for cluster; "decreasing" / "increasing" are other valid orders.  Vertex
    numbers follow MetaGraphsNext: deleting a vertex gives its number to the last one."""
    maxfam = max(len(nf) for nf in nodefamilies(net))
    if maxclustersize < maxfam:
        raise ValueError(f"maxclustersize {maxclustersize} is smaller than the size of largest node family {maxfam}.")
    adj = moralize(net)
    ordering = triangulate_minfill(adj)            # preorder indices in elimination order
    e2p = list(ordering)                           # elimination position (0-based) -> preorder index
    p2e = {v: i for i, v in enumerate(e2p)}
    names = [n.name for n in net.vec_node]
    buckets = [dict() for _ in ordering]
    for nf in nodefamilies(net):
        mb = sorted(p2e[v] for v in nf)
        _assign(buckets[mb[0]], mb, maxclustersize)
    labels, nodes_of, edges = [], {}, {}           # cluster labels in creation order; label -> nodes; {l1,l2} -> sepset

    def cluster_of(mb):
        nodes = sorted((e2p[i] for i in mb), reverse=True)
        lab = "".join(names[v - 1] for v in nodes)
        if lab not in nodes_of:
            nodes_of[lab] = nodes
            labels.append(lab)
        return lab, nodes

    def add_edge(l1, l2, sep):
        if l1 != l2:
            edges[frozenset((l1, l2))] = list(sep)

    for i in range(len(ordering)):
        bd = buckets[i]
        bi = e2p[i]
        prev = None
        sizes = (sorted(bd.keys(), key=lambda k: (_julia_slot(k), k)) if size_order == "julia" else
                 sorted(bd.keys(), reverse=(size_order == "decreasing")))
        for mb in [m for sz in sizes for m in list(bd[sz])]:
            lab, nodes = cluster_of(mb)
            if prev is not None:
                add_edge(prev, lab, [bi])
            prev = lab
            mb_new = mb[1:]
            if not mb_new:
                continue
            mb1, mb2 = _assign(buckets[mb_new[0]], mb_new, maxclustersize)
            lab1, _ = cluster_of(mb1)
            add_edge(lab, lab1, [v for v in nodes if v != bi])
            if len(mb1) != len(mb2):
                nodes2 = sorted((e2p[k] for k in mb2), reverse=True)
                lab2 = "".join(names[v - 1] for v in nodes2)
                if mb2 and lab2 in nodes_of:
                    for key in [k for k in edges if lab2 in k]:
                        (labn,) = key - {lab2}
                        add_edge(lab1, labn, edges[key])
                        del edges[key]
                    del nodes_of[lab2]
                    i2 = labels.index(lab2)       # Graphs.rem_vertex!: the last vertex takes the deleted one's number
                    labels[i2] = labels[-1]
                    labels.pop()
    index = {lab: k for k, lab in enumerate(labels)}
    out_edges = []
    for key, sep in edges.items():
        a, b = sorted(index[l] for l in key)
        out_edges.append((a, b, sep))
    out_edges.sort()
    return ClusterGraph([(lab, nodes_of[lab]) for lab in labels], out_edges, "joingraph")


def isfamilypreserving(cg: ClusterGraph, net) -> bool:
    """src/clustergraph.jl:169-182."""
    sets = [set(n) for _, n in cg.clusters]
    return all(any(set(nf) <= s for s in sets) for nf in nodefamilies(net))


def check_runningintersection(cg: ClusterGraph, net) -> bool:
    """src/clustergraph.jl:200-217: for every node, the clusters holding it and the sepsets holding it form a tree."""
    for v in range(1, len(net.vec_node) + 1):
        cl = [i for i, (_, n) in enumerate(cg.clusters) if v in n]
        ed = [(a, b) for (a, b, s) in cg.edges if v in s]
        if not cl:
            return False
        if len(ed) != len(cl) - 1:
            return False
        parent = {i: i for i in cl}

        def find(x):
            while parent[x] != x:
                x = parent[x]
            return x
        for a, b in ed:
            ra, rb = find(a), find(b)
            if ra == rb:
                return False
            parent[ra] = rb
    return True


def default_rootcluster_nodes(clusters) -> int:
    """src/clustergraph.jl:1043-1053 (one-argument method): among `clusters` (lists of preorder indices, decreasing),
    the first one that minimises: 0 if it is only the smallest index, else its second-smallest index; clusters
    without the smallest index are never chosen.  Returns a position in `clusters`."""
    i0 = min(n[-1] for n in clusters)
    best, bestscore = None, None
    for k, n in enumerate(clusters):
        if i0 not in n:
            continue
        score = 0 if len(n) == 1 else n[-2]
        if bestscore is None or score < bestscore:
            best, bestscore = k, score
    return best


def nodesubtree_clusterlist(cg: ClusterGraph, v: int):
    """src/clustergraph.jl:953-962 with nodesubtree (:219-240): DFS spanning tree (preorder edge list) of the
    subgraph induced by the clusters holding node v (1-based preorder index) and the sepsets holding v."""
    cl = [i for i, (_, n) in enumerate(cg.clusters) if v in n]
    if not cl:
        raise ValueError(f"no cluster with node {v}")
    sub = [(a, b, s) for (a, b, s) in cg.edges if a in cl and b in cl and v in s]
    rootj = cl[default_rootcluster_nodes([cg.clusters[i][1] for i in cl])]
    return spanningtree_clusterlist(cg, rootj, sub, vertices=cl)      # induced_subgraph(cgraph, clusters_i): that order
