"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Minimal phylogenetic-network container + extended-Newick reader.  The reference
delegates this to PhyloNetworks.jl (`readnewick`, `preorder!`, `vcv`), a
third-party dependency that is NOT under /root/reference (Project.toml:31,
compat "1"); only the behaviour the reference's tests rely on is restated:

  * extended Newick with `#Hn` hybrid labels and `:length:support:gamma`
  * a degree-1 root is removed (the tests' strings wrap the network in an
    extra pair of parentheses, e.g. test/test_canonicalform.jl:3)
  * edges are numbered in the order they are closed while reading, which is what
    `net.edge[k]` means in test/test_canonicalform.jl:15-23,75-98
  * unnamed internal nodes get names "I<k>" (src/clustergraph.jl preprocessnet!)
  * `preorder`: any topological order with the root first.  The reference's
    exact order is a PhyloNetworks implementation detail; log-likelihoods and
    calibrated marginals do not depend on it.  An explicit order can be imposed
    (`Network.set_preorder(names)`) to reproduce tests that hard-code indices.
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np


@dataclass(eq=False)
class Node:
    name: str
    leaf: bool = False
    hybrid: bool = False
    number: int = 0
    edges: List["Edge"] = field(default_factory=list, repr=False)

    def __repr__(self):
        return f"Node({self.name})"


@dataclass(eq=False)
class Edge:
    number: int
    parent: Node
    child: Node
    length: float
    gamma: float = 1.0
    hybrid: bool = False

    def __repr__(self):
        return f"Edge({self.number}:{self.parent.name}->{self.child.name},t={self.length},g={self.gamma})"


class Network:
    def __init__(self, root: Node, nodes: List[Node], edges: List[Edge]):
        self.root = root
        self.nodes = nodes
        self.edges = edges  # edges[k-1] has number k
        self.vec_node: List[Node] = []  # preorder
        self.name_internal_nodes()
        self.preorder()

    # -- structure helpers -------------------------------------------------
    def parent_edges(self, n: Node) -> List[Edge]:
        """Parent edges of n; major (largest gamma) hybrid edge first."""
        pe = [e for e in n.edges if e.child is n]
        pe.sort(key=lambda e: -e.gamma)
        return pe

    def child_edges(self, n: Node) -> List[Edge]:
        return [e for e in n.edges if e.parent is n]

    def parents(self, n: Node) -> List[Node]:
        return [e.parent for e in self.parent_edges(n)]

    def children(self, n: Node) -> List[Node]:
        return [e.child for e in self.child_edges(n)]

    def name_internal_nodes(self, prefix: str = "I"):
        used = {n.name for n in self.nodes if n.name}
        k = 1
        for n in self.nodes:
            if not n.name:
                while f"{prefix}{k}" in used:
                    k += 1
                n.name = f"{prefix}{k}"
                used.add(n.name)

    def preorder(self):
        """Topological order, root first; a node is listed once all its parents are."""
        indeg = {id(n): len(self.parent_edges(n)) for n in self.nodes}
        order, stack = [], [self.root]
        while stack:
            n = stack.pop()
            order.append(n)
            for e in reversed(self.child_edges(n)):
                indeg[id(e.child)] -= 1
                if indeg[id(e.child)] == 0:
                    stack.append(e.child)
        assert len(order) == len(self.nodes), "network is not a rooted DAG"
        self.vec_node = order
        return order

    def set_preorder(self, names: List[str]):
        byname = {n.name: n for n in self.nodes}
        order = [byname[s] for s in names]
        assert len(order) == len(self.nodes)
        pos = {id(n): i for i, n in enumerate(order)}
        for e in self.edges:
            assert pos[id(e.parent)] < pos[id(e.child)], "not a preorder"
        self.vec_node = order

    def index(self, n: Node) -> int:
        """0-based preorder index."""
        for i, m in enumerate(self.vec_node):
            if m is n:
                return i
        raise KeyError(n)

    def node(self, name: str) -> Node:
        for n in self.nodes:
            if n.name == name:
                return n
        raise KeyError(name)

    @property
    def tip_names(self):
        return [n.name for n in self.vec_node if n.leaf]


_TOK = re.compile(r"\s*([(),;:]|[^(),;:\s]+)")


def read_newick(s: str) -> Network:
    toks = _TOK.findall(s)
    pos = 0
    nodes: List[Node] = []
    edges: List[Edge] = []
    hybrids = {}

    def peek():
        return toks[pos] if pos < len(toks) else None

    def take():
        nonlocal pos
        t = toks[pos]
        pos += 1
        return t

    def read_subtree():
        """returns (node, length, gamma) for the branch above the subtree"""
        node = Node("")
        child_edges = []
        if peek() == "(":
            take()
            while True:
                ch, clen, cgam = read_subtree()
                # the edge is numbered when the child's branch closes
                e = Edge(len(edges) + 1, node, ch, clen,
                         1.0 if cgam is None else cgam, hybrid=ch.hybrid)
                e._gamma_given = cgam is not None
                edges.append(e)
                child_edges.append(e)
                t = take()
                if t == ",":
                    continue
                if t == ")":
                    break
                raise ValueError(f"unexpected token {t!r}")
        name = ""
        if peek() not in (":", ",", ")", ";", None):
            name = take()
        vals = []
        while peek() == ":":
            take()
            if peek() not in (":", ",", ")", ";", None):
                vals.append(float(take()))
            else:
                vals.append(None)
        length = vals[0] if len(vals) > 0 and vals[0] is not None else -1.0
        gamma = vals[2] if len(vals) > 2 and vals[2] is not None else None
        if name.startswith("#"):
            hname = name[1:]
            if hname in hybrids:
                existing = hybrids[hname]
                for e in child_edges:
                    e.parent = existing
                node = existing
            else:
                node.name, node.hybrid = hname, True
                hybrids[hname] = node
                nodes.append(node)
        else:
            node.name = name
            nodes.append(node)
            if not child_edges:
                node.leaf = True
        return (node, length, gamma)

    root, _, _ = read_subtree()
    if peek() == ";":
        take()
    # hybrid gammas: fill in a missing one as 1 - other
    for hn in hybrids.values():
        pe = [e for e in edges if e.child is hn]
        if len(pe) == 2:
            g0, g1 = pe[0]._gamma_given, pe[1]._gamma_given
            if g0 and not g1:
                pe[1].gamma = 1.0 - pe[0].gamma
            elif g1 and not g0:
                pe[0].gamma = 1.0 - pe[1].gamma
            elif not g0 and not g1:
                pe[0].gamma = pe[1].gamma = 0.5
    for e in edges:
        e.parent.edges.append(e)
        e.child.edges.append(e)
    # remove a degree-1 root (PhyloNetworks readnewick behaviour relied on by
    # test/test_canonicalform.jl:3, where 9 nodes and 9 edges remain)
    while not root.leaf and len([e for e in root.edges if e.parent is root]) == 1:
        e = [e for e in root.edges if e.parent is root][0]
        newroot = e.child
        if newroot.hybrid or newroot.leaf:
            break
        newroot.edges.remove(e)
        edges.remove(e)
        nodes.remove(root)
        root = newroot
    for k, e in enumerate(edges):
        e.number = k + 1
    for k, n in enumerate(nodes):
        n.number = k + 1
    return Network(root, nodes, edges)


# ---------------------------------------------------------------------------
# synthetic trees for oracle-side tests
# ---------------------------------------------------------------------------

def random_tree_newick(ntips: int, rng: np.random.Generator, lo=0.1, hi=1.0) -> str:
    """Random bifurcating tree by uniform random joins."""
    parts = [f"t{i+1}" for i in range(ntips)]
    k = 0
    while len(parts) > 1:
        i, j = sorted(rng.choice(len(parts), size=2, replace=False))
        a, b = parts[i], parts[j]
        la, lb = rng.uniform(lo, hi, size=2)
        k += 1
        new = f"({a}:{la:.6f},{b}:{lb:.6f})n{k}"
        parts = [p for t, p in enumerate(parts) if t not in (i, j)] + [new]
    return parts[0] + ";"


def random_network(ntips: int, nhybrids: int, rng: np.random.Generator, lo=0.1, hi=1.0) -> Network:
    """Random rooted network for the cfg5-shaped tests: a random bifurcating tree (uniform
    random joins, edge lengths U(lo, hi)) plus `nhybrids` reticulations, each inside the two child edges of its own
    internal node w, so that the blobs are edge-disjoint and the network has a small level:
      * triangle: x splits w->c1, hybrid y splits w->c2, minor edge x->y;
      * 4-cycle:  x splits w->c1, hybrid y splits a child edge of c2, minor edge x->y (c2 then hosts no blob itself).
    Minor inheritance gamma ~ U(0.1, 0.5); every edge length > 0 (no degenerate hybrids)."""
    nodes: List[Node] = []
    edges: List[Edge] = []

    def new_node(name="", leaf=False):
        n = Node(name=name, leaf=leaf)
        nodes.append(n)
        return n

    def new_edge(pa, ch, length, gamma=1.0, hybrid=False):
        e = Edge(number=len(edges) + 1, parent=pa, child=ch, length=float(length), gamma=float(gamma), hybrid=hybrid)
        edges.append(e)
        pa.edges.append(e)
        ch.edges.append(e)
        return e

    parts = [new_node(f"t{i+1}", leaf=True) for i in range(ntips)]
    while len(parts) > 1:
        i, j = sorted(int(x) for x in rng.choice(len(parts), size=2, replace=False))
        a, b = parts[i], parts[j]
        w = new_node()
        new_edge(w, a, rng.uniform(lo, hi))
        new_edge(w, b, rng.uniform(lo, hi))
        parts = [p for t, p in enumerate(parts) if t not in (i, j)] + [w]
    root = parts[0]

    def child_edges(n):
        return [e for e in n.edges if e.parent is n]

    def split(e, frac, hybrid_node):
        """insert a node on edge e at fraction `frac` from the parent; returns it"""
        m = new_node()
        m.hybrid = hybrid_node
        ch = e.child
        full = e.length
        e.child.edges.remove(e)
        e.child = m
        e.length = full * frac
        m.edges.append(e)
        new_edge(m, ch, full * (1.0 - frac))
        return m

    internal = [n for n in nodes if not n.leaf]
    order = rng.permutation(len(internal))
    blocked = set()
    made = 0
    for k in order:
        if made >= nhybrids:
            break
        w = internal[int(k)]
        if id(w) in blocked:
            continue
        ce = child_edges(w)
        if len(ce) != 2:
            continue
        if rng.random() < 0.5:
            ce = ce[::-1]
        e1, e2 = ce
        c2 = e2.child
        target = e2
        if (not c2.leaf) and id(c2) not in blocked and rng.random() < 0.5 and len(child_edges(c2)) == 2:
            target = child_edges(c2)[int(rng.integers(2))]      # 4-cycle through c2
            blocked.add(id(c2))
        x = split(e1, rng.uniform(0.3, 0.7), False)
        y = split(target, rng.uniform(0.3, 0.7), True)
        g = rng.uniform(0.1, 0.5)
        major = [e for e in y.edges if e.child is y][0]
        major.gamma = 1.0 - g
        major.hybrid = True
        new_edge(x, y, rng.uniform(0.05, 0.3), gamma=g, hybrid=True)
        blocked.add(id(w))
        made += 1
    return Network(root, nodes, edges)


def random_level3_network(ntips: int, nblobs: int, rng: np.random.Generator, lo=0.1, hi=1.0) -> Network:
    """Random rooted level-3 network (BASELINE.json configs[4]): a random bifurcating tree (uniform random joins) in
    which `nblobs` internal nodes w, with children c1 and c2, are replaced by the level-3 blob of the reference's own
    test network (test/test_calibration.jl:132: "((#H1,#H2)I1,(((A)#H1,#H3)#H2,(B)#H3)I2)I3"): w -> I1, I2;
    I1 -> H1, H2 (minor edges); I2 -> H2, H3 (major); H2 -> H1 (major), H3 (minor); H1 -> c1; H3 -> c2.
    3 reticulations and 5 new nodes per blob; blobs are separated by cut edges, so the level is exactly 3.
    Minor inheritance gamma ~ U(0.1, 0.5); every edge length > 0."""
    net = random_network(ntips, 0, rng, lo, hi)
    nodes, edges = net.nodes, net.edges

    def new_node(hybrid=False):
        n = Node(name="", hybrid=hybrid)
        nodes.append(n)
        return n

    def new_edge(pa, ch, length, gamma=1.0, hybrid=False):
        e = Edge(number=len(edges) + 1, parent=pa, child=ch, length=float(length), gamma=float(gamma), hybrid=hybrid)
        edges.append(e)
        pa.edges.append(e)
        ch.edges.append(e)
        return e

    internal = [n for n in nodes if not n.leaf]
    picks = rng.permutation(len(internal))[:nblobs]
    for k in picks:
        w = internal[int(k)]
        e1, e2 = [e for e in w.edges if e.parent is w]
        if rng.random() < 0.5:
            e1, e2 = e2, e1
        i1, i2 = new_node(), new_node()
        h1, h2, h3 = new_node(True), new_node(True), new_node(True)
        # the old child edges become H1 -> c1 and H3 -> c2
        for e, h in ((e1, h1), (e2, h3)):
            w.edges.remove(e)
            e.parent = h
            h.edges.append(e)
        new_edge(w, i1, rng.uniform(lo, hi))
        new_edge(w, i2, rng.uniform(lo, hi))
        g1, g2, g3 = rng.uniform(0.1, 0.5, size=3)
        new_edge(i1, h1, rng.uniform(0.05, 0.3), gamma=g1, hybrid=True)
        new_edge(h2, h1, rng.uniform(0.05, 0.3), gamma=1.0 - g1, hybrid=True)
        new_edge(i1, h2, rng.uniform(0.05, 0.3), gamma=g2, hybrid=True)
        new_edge(i2, h2, rng.uniform(0.05, 0.3), gamma=1.0 - g2, hybrid=True)
        new_edge(h2, h3, rng.uniform(0.05, 0.3), gamma=g3, hybrid=True)
        new_edge(i2, h3, rng.uniform(0.05, 0.3), gamma=1.0 - g3, hybrid=True)
    for n in nodes:
        if not n.leaf and n.name.startswith("I"):
            n.name = ""
    return Network(net.root, nodes, edges)


def set_preorder_by_tipsets(net: Network, order, internal_names=None):
    """Give `net` the node preordering `order`: entries are node names (tips, named hybrids) or, for unnamed internal
    nodes, the list of tip names below them.  Optionally (re)name those internal nodes."""
    def tips_below(n):
        out, stack = set(), [n]
        while stack:
            x = stack.pop()
            if x.leaf:
                out.add(x.name)
            stack.extend(net.children(x))
        return frozenset(out)
    named = {n.name: n for n in net.nodes if n.leaf or n.hybrid}
    by_tips = {tips_below(n): n for n in net.nodes if not n.leaf and not n.hybrid}
    if internal_names:
        for nm, tips in internal_names.items():
            by_tips[frozenset(tips)].name = nm
    seq = [named[x] if isinstance(x, str) else by_tips[frozenset(x)] for x in order]
    net.set_preorder([n.name for n in seq])
