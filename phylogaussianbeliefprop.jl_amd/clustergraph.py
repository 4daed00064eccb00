"""Host-side schedule builders of src/clustergraph.jl that the hot path's callers need: the spanning-tree edge lists
`calibrate!` walks.  Cluster graphs are given as plain arrays (the same ones `pgbp_desc` takes): `edges[k] = (i, j)`
are the two clusters of sepset k, 0-based; `cluster_nodes[i]` lists the (1-based, preorder) node indices of cluster i.

A schedule tree is the 4-tuple `(parent_labels, child_labels, parent_indices, child_indices)` of
`spanningtree_clusterlist` (src/clustergraph.jl:885-894), clusters in depth-first preorder, 0-based indices;
`ClusterGraphBelief.set_schedule` / `calibrate_` take a list of them."""
from typing import List, Optional, Sequence, Tuple


def default_rootcluster(cluster_nodes: Sequence[Sequence[int]], is_leaf: Sequence[bool]) -> int:
    """default_rootcluster(clustergraph, nodevector_preordered) (src/clustergraph.jl:1022-1029): a cluster that
    contains the network's root (preorder index 1); among several, the first with the fewest leaves."""
    best, best_score = None, None
    for k, nodes in enumerate(cluster_nodes):
        if 1 in nodes:
            score = sum(1 for i in nodes if is_leaf[i - 1])
            if best_score is None or score < best_score:
                best, best_score = k, score
    if best is None:
        raise ValueError("no cluster contains the root")
    return best



def _graphs_jl_tree_order(n: int, nbrs: List[List[int]], root: int):
    """The vertex order `spanningtree_clusterlist` gets from Graphs.jl (src/clustergraph.jl:885-894):
    `par = dfs_parents(g, root)` -- an iterative depth-first search that always follows the first unseen neighbour in
    increasing vertex code -- then `topological_sort(tree(par))` = `topological_sort_by_dfs`: depth-first searches started
    from every vertex in increasing code, vertices listed by REVERSE finishing time (so of two children the one with the
    larger code comes first, and whatever hangs below vertex 0 ... comes last when vertex 0 is not the root).
    nbrs[v]: neighbours of v in increasing code.  Returns (parents, vertices after the root in that order)."""
    par = [-1] * n
    seen = [False] * n
    stack = [root]
    seen[root] = True
    par[root] = root
    nxt = [0] * n
    while stack:
        v = stack[-1]
        while nxt[v] < len(nbrs[v]) and seen[nbrs[v][nxt[v]]]:
            nxt[v] += 1
        if nxt[v] == len(nbrs[v]):
            stack.pop()
            continue
        u = nbrs[v][nxt[v]]
        seen[u] = True
        par[u] = v
        stack.append(u)
    kids = [[] for _ in range(n)]
    for v in range(n):
        if par[v] >= 0 and par[v] != v:
            kids[par[v]].append(v)          # increasing code
    color = [0] * n
    finished = []
    pos = [0] * n
    for s in range(n):
        if color[s] or par[s] < 0:
            continue
        color[s] = 1
        stack = [s]
        while stack:
            u = stack[-1]
            while pos[u] < len(kids[u]) and color[kids[u][pos[u]]]:
                pos[u] += 1
            if pos[u] == len(kids[u]):
                color[u] = 2
                finished.append(u)
                stack.pop()
            else:
                w = kids[u][pos[u]]
                color[w] = 1
                stack.append(w)
    order = finished[::-1]
    assert order and order[0] == root
    return par, order[1:]


def spanningtree_clusterlist(n_clusters: int, edges: Sequence[Tuple[int, int]], rootj: int,
                             labels: Optional[Sequence] = None, vertices: Optional[Sequence[int]] = None):
    """spanningtree_clusterlist(clustergraph, root_index) (src/clustergraph.jl:885-894): depth-first spanning tree from
    `rootj`; the clusters other than the root, each with its parent, in the order Graphs.jl's dfs_parents +
    topological_sort produce (`_graphs_jl_tree_order`).  `vertices`: the numbering of the subgraph the reference works on
    (Graphs.induced_subgraph renumbers its vertices), as a list of cluster indices; default 0 .. n_clusters - 1."""
    if vertices is None:
        vertices = range(n_clusters)
    vertices = list(vertices)
    code = {c: i for i, c in enumerate(vertices)}
    nb: List[List[int]] = [[] for _ in vertices]
    for (a, b) in edges:
        if a in code and b in code:
            nb[code[a]].append(code[b])
            nb[code[b]].append(code[a])
    for lst in nb:
        lst.sort()
    par, order = _graphs_jl_tree_order(len(vertices), nb, code[rootj])
    ch_j = [vertices[v] for v in order]
    pa_j = [vertices[par[v]] for v in order]
    lab = (lambda i: i) if labels is None else (lambda i: labels[i])
    return [lab(i) for i in pa_j], [lab(i) for i in ch_j], pa_j, ch_j


def spanningtrees_clusterlist(n_clusters: int, edges: Sequence[Tuple[int, int]],
                              cluster_nodes: Sequence[Sequence[int]], is_leaf: Sequence[bool],
                              labels: Optional[Sequence] = None):
    """spanningtrees_clusterlist(clustergraph, nodevector_preordered) (src/clustergraph.jl:908-937): spanning trees
    that together cover every edge.  Each is Graphs.jl's kruskal_mst with weight = number of earlier trees that used the
    edge (edges in lexicographic (smaller, larger) cluster order, stable sort by weight, stop at n - 1 edges); the reference
    then works on `induced_subgraph(cg, mst_edges)`, whose vertices are numbered in the order of their first appearance in
    that edge list: the root is the first cluster of that numbering among those that hold the network's root with the
    fewest tips (default_rootcluster), and the depth-first search follows that numbering."""
    und = sorted(range(len(edges)), key=lambda k: (min(edges[k]), max(edges[k])))
    used = [0] * len(edges)
    schedule = []
    while any(u == 0 for u in used):
        parent = list(range(n_clusters))

        def find(x):
            while parent[x] != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x
        chosen = []
        for k in sorted(und, key=lambda k: used[k]):       # stable: ties keep the lexicographic edge order
            ra, rb = find(min(edges[k])), find(max(edges[k]))
            if ra != rb:
                parent[ra] = rb
                chosen.append(k)
                if len(chosen) >= n_clusters - 1:
                    break
        vmap, seen = [], set()
        for k in chosen:
            for v in sorted(edges[k]):
                if v not in seen:
                    seen.add(v)
                    vmap.append(v)
        best, best_score = None, None
        for v in vmap:
            nodes = cluster_nodes[v]
            if 1 in nodes:
                score = sum(1 for i in nodes if is_leaf[i - 1])
                if best_score is None or score < best_score:
                    best, best_score = v, score
        if best is None:
            raise ValueError("no cluster contains the root")
        schedule.append(spanningtree_clusterlist(n_clusters, [edges[k] for k in chosen], best, labels, vertices=vmap))
        for k in chosen:
            used[k] += 1
    return schedule


# ---------------------------------------------------------------------------------------------------------------
# cluster-graph construction: join-graph structuring (src/clustergraph.jl:605-757), sized for networks with tens of
# thousands of nodes (the reference's own version is quadratic through Graphs.jl / MetaGraphsNext)
# ---------------------------------------------------------------------------------------------------------------

def moralize(node2family: Sequence[Sequence[int]]):
    """moralize(net) (src/clustergraph.jl:43-77) from the node families [child, parents...] (1-based preorder
    indices): adjacency sets of the moral graph (child - parent edges, parents of one child married)."""
    adj = {nf[0]: set() for nf in node2family}
    for nf in node2family:
        for a in range(len(nf)):
            for b in range(a + 1, len(nf)):
                adj[nf[a]].add(nf[b])
                adj[nf[b]].add(nf[a])
    return adj


def triangulate_minfill(adj):
    """triangulate_minfill!(graph) (src/clustergraph.jl:87-121): greedy min-fill elimination order, ties broken by
    the larger preorder index; `adj` gains the fill edges.  Same order as the reference's rescan of all vertices, with
    a heap and local updates: eliminating v changes the fill count only of v's neighbours and of the common
    neighbours of a new fill edge's ends."""
    import heapq
    g2 = {k: set(v) for k, v in adj.items()}

    def fill(v):
        nb = list(g2[v])
        c = 0
        for i, a in enumerate(nb):
            ga = g2[a]
            for b in nb[i + 1:]:
                if b not in ga:
                    c += 1
        return c

    score = {v: fill(v) for v in g2}
    heap = [(score[v], -v) for v in g2]
    heapq.heapify(heap)
    ordering = []
    while len(g2) > 1:
        while True:
            sc, negv = heapq.heappop(heap)
            v = -negv
            if v in g2 and score[v] == sc:
                break
        nb = sorted(g2[v])
        touched = set(nb)
        for i, a in enumerate(nb):
            for b in nb[i + 1:]:
                if b not in g2[a]:
                    touched |= g2[a] & g2[b]          # common neighbours see one missing edge less
                    g2[a].add(b); g2[b].add(a)
                    adj[a].add(b); adj[b].add(a)
        ordering.append(v)
        for u in nb:
            g2[u].discard(v)
        del g2[v]
        del score[v]
        touched.discard(v)
        for u in touched:
            s = fill(u)
            if s != score[u]:
                score[u] = s
                heapq.heappush(heap, (s, -u))
    ordering.append(next(iter(g2)))
    return ordering


def _julia_hash_int(n: int) -> int:
    """Base.hash(::Int64) of Julia 1.x (hash_64_64 bit mixing), for `_julia_dict_order`."""
    M = (1 << 64) - 1
    a = n & M
    a = (~a + (a << 21)) & M
    a ^= a >> 24
    a = (a + (a << 3) + (a << 8)) & M
    a ^= a >> 14
    a = (a + (a << 2) + (a << 4)) & M
    a ^= a >> 28
    a = (a + (a << 31)) & M
    return a


def _julia_dict_order(keys):
    """Iteration order of a Julia `Dict{<:Integer}` holding these small keys: slot = hash & 15 in the initial 16-slot
    table (the keys 1..10 have distinct slots, so deletions and re-insertions do not move them).  The reference walks a
    bucket's minibucket sizes in this order (`values(bd)`, src/clustergraph.jl:645); with it the join graphs of the
    reference's doctests come out cluster for cluster (docs/src/man/clustergraphs.md)."""
    return sorted(keys, key=lambda k: (_julia_hash_int(int(k)) & 15, int(k)))


def _assign(bucket, new, maxsize):
    """assign!(bucket, new_minibucket, max_minibucket_size) (src/clustergraph.jl:705-736)."""
    for sz in sorted(bucket, reverse=True):
        mbs = bucket[sz]
        for i, mb in enumerate(mbs):
            merged = sorted(set(new) | set(mb))
            if len(merged) <= maxsize:
                mbs.pop(i)
                if not mbs:
                    del bucket[sz]
                bucket.setdefault(len(merged), []).append(merged)
                return merged, mb
    bucket.setdefault(len(new), []).append(new)
    return new, []


def joingraph(node2family: Sequence[Sequence[int]], maxclustersize: int):
    """clustergraph!(net, JoinGraphStructuring(maxclustersize)) (src/clustergraph.jl:405-410, 605-697) from the node
    families of the network (nodefamilies(net), :136-146: [child, parents by decreasing index], 1-based preorder
    indices).  Returns (cluster_nodes, edges, sepset_nodes): cluster i holds the nodes cluster_nodes[i] (decreasing
    preorder index), sepset k = edges[k] = (i, j), i < j, holds sepset_nodes[k].
    The reference walks a bucket's minibuckets in the iteration order of a Julia Dict keyed by their size (:645):
    `_julia_dict_order`; clusters are numbered as the vertices of its MetaGraph end up (a deleted vertex's number goes
    to the last vertex: Graphs.rem_vertex!), which is the order LTRIP(clusters, net) and the beliefs see."""
    maxfam = max(len(nf) for nf in node2family)
    if maxclustersize < maxfam:
        raise ValueError(f"maxclustersize {maxclustersize} is smaller than the size of largest node family {maxfam}.")
    ordering = triangulate_minfill(moralize(node2family))
    e2p = ordering
    p2e = {v: i for i, v in enumerate(e2p)}
    buckets = [dict() for _ in ordering]
    for nf in node2family:
        mb = sorted(p2e[v] for v in nf)
        _assign(buckets[mb[0]], mb, maxclustersize)
    order: List[Tuple[int, ...]] = []      # cluster keys (node tuples, decreasing) in creation order
    alive = {}                             # key -> True
    nbrs = {}                              # key -> {neighbour key: sepset}

    def cluster_of(mb):
        key = tuple(sorted((e2p[i] for i in mb), reverse=True))
        if key not in alive:
            alive[key] = True
            nbrs[key] = {}
            order.append(key)
        return key

    def add_edge(k1, k2, sep):
        if k1 != k2:
            nbrs[k1][k2] = list(sep)
            nbrs[k2][k1] = nbrs[k1][k2]

    for i in range(len(ordering)):
        bd = buckets[i]
        bi = e2p[i]
        prev = None
        for mb in [m for sz in _julia_dict_order(bd) for m in list(bd[sz])]:
            key = cluster_of(mb)
            if prev is not None:
                add_edge(prev, key, [bi])          # chain of the bucket's minibuckets: sepset = the bucket's node
            prev = key
            mb_new = mb[1:]
            if not mb_new:
                continue
            mb1, mb2 = _assign(buckets[mb_new[0]], mb_new, maxclustersize)
            key1 = cluster_of(mb1)
            add_edge(key, key1, [v for v in key if v != bi])
            if len(mb1) != len(mb2) and mb2:
                key2 = tuple(sorted((e2p[k] for k in mb2), reverse=True))
                if key2 in alive:                  # mb2 was absorbed into mb1: contract the two clusters
                    for kn, sep in list(nbrs[key2].items()):
                        del nbrs[kn][key2]
                        add_edge(key1, kn, sep)
                    del nbrs[key2]
                    del alive[key2]
                    i2 = order.index(key2)         # rem_vertex!: the last vertex takes the deleted one's number
                    order[i2] = order[-1]
                    order.pop()
    keys = order
    index = {k: i for i, k in enumerate(keys)}
    edges, seps = [], []
    for k in keys:
        for kn, sep in nbrs[k].items():
            if index[k] < index[kn]:
                edges.append((index[k], index[kn]))
                seps.append(list(sep))
    o = sorted(range(len(edges)), key=lambda t: edges[t])
    return [list(k) for k in keys], [edges[t] for t in o], [seps[t] for t in o]


def default_rootcluster_nodes(cluster_nodes: Sequence[Sequence[int]]) -> int:
    """default_rootcluster(clustergraph) (src/clustergraph.jl:1043-1053): among clusters given by their nodes
    (decreasing preorder index), the first that holds the overall smallest index and minimises: 0 if it holds nothing
    else, otherwise its second-smallest index."""
    i0 = min(n[-1] for n in cluster_nodes)
    best, best_score = None, None
    for k, n in enumerate(cluster_nodes):
        if i0 in n:
            score = 0 if len(n) == 1 else n[-2]
            if best_score is None or score < best_score:
                best, best_score = k, score
    return best


def nodesubtree_clusterlist(cluster_nodes: Sequence[Sequence[int]], edges: Sequence[Tuple[int, int]],
                            sepset_nodes: Sequence[Sequence[int]], node: int, labels: Optional[Sequence] = None):
    """nodesubtree_clusterlist(clustergraph, nodesymbol) (src/clustergraph.jl:953-962, nodesubtree :219-240): the
    schedule tree of the subgraph of clusters and sepsets that hold `node` (1-based preorder index), rooted by
    default_rootcluster(subgraph); cluster indices are those of the whole graph."""
    cl = [i for i, n in enumerate(cluster_nodes) if node in n]
    if not cl:
        raise ValueError(f"no cluster with node {node}")
    inside = set(cl)
    sub = [e for e, s in zip(edges, sepset_nodes) if node in s and e[0] in inside and e[1] in inside]
    rootj = cl[default_rootcluster_nodes([cluster_nodes[i] for i in cl])]
    return spanningtree_clusterlist(len(cluster_nodes), sub, rootj, labels, vertices=cl)   # induced_subgraph: that numbering


def bethe(node2family: Sequence[Sequence[int]]):
    """clustergraph!(net, Bethe()) (src/clustergraph.jl:473-527) from the node families: one factor cluster per node
    family, visited by decreasing node index and merged into a child's factor cluster when it is a subset of it (the
    children are tried by increasing index; the reference tries them in the network's edge order), then one variable
    cluster {v} for every node in more than one factor cluster, by decreasing index, joined to each of them.
    Returns (cluster_nodes, edges, sepset_nodes) as `joingraph` does; edges are (variable cluster, factor cluster)."""
    n = len(node2family)
    children: List[List[int]] = [[] for _ in range(n + 1)]
    for nf in node2family:
        for pa in nf[1:]:
            children[pa].append(nf[0])
    cluster_nodes: List[List[int]] = []
    node2code = {}
    member: List[List[int]] = [[] for _ in range(n + 1)]
    for v in range(n, 0, -1):
        fam = list(node2family[v - 1])
        if len(fam) == 1:
            continue
        merged = False
        for ch in sorted(children[v]):
            cc = node2code[ch]
            cs = cluster_nodes[cc]
            if all(x in cs for x in fam):
                node2code[v] = cc
                merged = True
                break
        if merged:
            continue
        node2code[v] = len(cluster_nodes)
        cluster_nodes.append(fam)
        for x in fam:
            member[x].append(node2code[v])
    edges, seps = [], []
    for v in range(n, 0, -1):
        if len(member[v]) <= 1:
            continue
        vc = len(cluster_nodes)
        cluster_nodes.append([v])
        for fc in member[v]:
            edges.append((vc, fc))
            seps.append([v])
    return cluster_nodes, edges, seps


def cliquetree(node2family: Sequence[Sequence[int]]):
    """clustergraph!(net, Cliquetree()) (src/clustergraph.jl:452-466, 757-820) from the node families: moralize, min-fill
    triangulation, the maximal cliques of the chordal graph, a maximum-weight spanning tree on sepset size.  The
    reference runs Kruskal over all clique pairs; here each clique C_v = {v} + (its neighbours eliminated later) is joined
    to the clique of the first-eliminated vertex of C_v - {v}, which is a maximum-weight spanning tree as well (a clique
    tree: running intersection holds) and costs O(sum of clique sizes).  Returns (cluster_nodes, edges, sepset_nodes)."""
    adj = moralize(node2family)
    ordering = triangulate_minfill(adj)
    posn = {v: i for i, v in enumerate(ordering)}
    later = {v: sorted((u for u in adj[v] if posn[u] > posn[v]), key=lambda u: posn[u]) for v in ordering}
    # Elimination clique of v, and the elimination tree: parent(v) = its first later neighbour.  C_w is not maximal iff it
    # sits inside the clique that stands for one of w's CHILDREN in that tree (if C_w is inside C_x, x earlier, it is inside
    # the clique of every vertex on the tree path from x up to w); the cliques a maximal clique absorbs form a path upwards.
    clique = {v: frozenset([v] + later[v]) for v in ordering}
    children = {v: [] for v in ordering}
    for v in ordering:
        if later[v]:
            children[later[v][0]].append(v)
    rep = {}                      # vertex -> the vertex whose maximal clique stands for its elimination clique
    maximal = []
    for w in ordering:
        host = None
        for c in reversed(children[w]):            # latest-eliminated child first
            if clique[w] <= clique[rep[c]]:
                host = rep[c]
                break
        if host is None:
            rep[w] = w
            maximal.append(w)
        else:
            rep[w] = host
    index = {v: i for i, v in enumerate(maximal)}
    cluster_nodes = [sorted(clique[v], reverse=True) for v in maximal]
    edges, seps = [], []
    for v in maximal:
        # walk up the elimination tree (parent = first later neighbour) past the vertices whose cliques C_v absorbed,
        # to the clique that takes over what is left of C_v
        w = later[v][0] if later[v] else None
        while w is not None and rep[w] == v:
            w = later[w][0] if later[w] else None
        if w is not None:
            u = rep[w]
            a, b = sorted((index[v], index[u]))
            edges.append((a, b))
            seps.append(sorted(clique[v] & clique[u], reverse=True))
    o = sorted(range(len(edges)), key=lambda t: edges[t])
    return cluster_nodes, [edges[t] for t in o], [seps[t] for t in o]


def ltrip(node2family: Sequence[Sequence[int]], clusters: Optional[Sequence[Sequence[int]]] = None):
    """clustergraph!(net, LTRIP(net)) / LTRIP(clusters, net) (src/clustergraph.jl:330-345, 530-598; Streicher & du Preez
    2017): the given clusters (default: the node families, one of them the root alone) become the graph's clusters; for
    every node n, by decreasing index, the clusters holding n are spanned by a maximum-weight tree whose edges get n in
    their sepset -- weights = size of the clusters' intersection, plus, for both endpoints, the number of maximum-weight
    edges (within n's subgraph) they touch.  The reference takes Graphs.jl's `kruskal_mst(minimize=false)`; ties among
    equal weights are broken here by (lower, higher) cluster index.  Returns (cluster_nodes, edges, sepset_nodes)."""
    if clusters is None:
        cl = [sorted(nf, reverse=True) for nf in node2family]
    else:
        cl = [sorted(c, reverse=True) for c in clusters]
        sets = [set(c) for c in cl]
        if not all(any(set(nf) <= s for s in sets) for nf in node2family):
            raise ValueError("`clusters` is not family preserving with respect to `net`")
    holders = {}
    for ci, c in enumerate(cl):
        for v in c:
            holders.setdefault(v, []).append(ci)
    csets = [set(c) for c in cl]
    sep = {}
    for v in sorted(holders, reverse=True):
        hs = holders[v]
        if len(hs) < 2:
            continue
        # every pair of clusters holding v shares at least v: the subgraph is complete
        w = {(a, b): len(csets[a] & csets[b]) for i, a in enumerate(hs) for b in hs[i + 1:]}
        maxw = max(w.values())
        score = {c: 0 for c in hs}
        for (a, b), x in w.items():
            if x == maxw:
                score[a] += 1
                score[b] += 1
        adj_w = {e: x + score[e[0]] + score[e[1]] for e, x in w.items()}
        parent = {c: c for c in hs}

        def find(x):
            while parent[x] != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x
        for (a, b) in sorted(adj_w, key=lambda e: (-adj_w[e], e)):
            ra, rb = find(a), find(b)
            if ra != rb:
                parent[ra] = rb
                sep.setdefault((a, b), []).append(v)
    edges = sorted(sep)
    return [list(c) for c in cl], edges, [sep[e] for e in edges]
