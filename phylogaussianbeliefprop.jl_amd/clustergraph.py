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


def spanningtree_clusterlist(n_clusters: int, edges: Sequence[Tuple[int, int]], rootj: int,
                             labels: Optional[Sequence] = None):
    """spanningtree_clusterlist(clustergraph, root_index) (src/clustergraph.jl:885-894): depth-first spanning tree
    from `rootj`, neighbours in increasing cluster index; clusters other than the root in preorder, each with its
    parent."""
    nb: List[List[int]] = [[] for _ in range(n_clusters)]
    for (a, b) in edges:
        nb[a].append(b)
        nb[b].append(a)
    for lst in nb:
        lst.sort()
    seen = [False] * n_clusters
    seen[rootj] = True
    pa_j: List[int] = []
    ch_j: List[int] = []
    stack = [(rootj, 0)]
    while stack:
        v, pos = stack[-1]
        while pos < len(nb[v]) and seen[nb[v][pos]]:
            pos += 1
        if pos == len(nb[v]):
            stack.pop()
            continue
        u = nb[v][pos]
        stack[-1] = (v, pos + 1)
        seen[u] = True
        pa_j.append(v)
        ch_j.append(u)
        stack.append((u, 0))
    lab = (lambda i: i) if labels is None else (lambda i: labels[i])
    return [lab(i) for i in pa_j], [lab(i) for i in ch_j], pa_j, ch_j


def spanningtrees_clusterlist(n_clusters: int, edges: Sequence[Tuple[int, int]],
                              cluster_nodes: Sequence[Sequence[int]], is_leaf: Sequence[bool],
                              labels: Optional[Sequence] = None):
    """spanningtrees_clusterlist(clustergraph, nodevector_preordered) (src/clustergraph.jl:908-937): spanning trees
    that together cover every edge: Kruskal minimum spanning trees with weight = number of earlier trees that used
    the edge, each rooted at `default_rootcluster` and listed as by `spanningtree_clusterlist`."""
    used = [0] * len(edges)
    rootj = default_rootcluster(cluster_nodes, is_leaf)
    schedule = []
    while any(u == 0 for u in used):
        parent = list(range(n_clusters))

        def find(x):
            while parent[x] != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x
        chosen = []
        for k in sorted(range(len(edges)), key=lambda k: (used[k], edges[k][0], edges[k][1])):
            ra, rb = find(edges[k][0]), find(edges[k][1])
            if ra != rb:
                parent[ra] = rb
                chosen.append(k)
        schedule.append(spanningtree_clusterlist(n_clusters, [edges[k] for k in chosen], rootj, labels))
        for k in chosen:
            used[k] += 1
    return schedule
