"""
CPU test of the host side of the device factor fill: the node-family table `pgbp_amd.lg_families` builds from
allocatebeliefs' outputs, and the closed form the kernel evaluates (pgbp_lgfill.hip header), restated here in numpy,
against the oracle's assignfactors! (src/beliefs.jl:786-861) on random networks with hybrid nodes.
"""
import zlib

import numpy as np
import pytest

import pgbp_amd as P
from helpers import lg_inputs_from_oracle, oracle_setup
from oracle import clustergraph as OCG
from oracle import network as ON
from test_gpu_lgfill import _models

LOG2PI = np.log(2 * np.pi)


def fill_numpy(fam, data, kw, dims):
    """The closed form of pgbp_lgfill.hip, masks included: the factor keeps the child_mask components O of its residual
    (V_OO inverted); each in-scope node's block lists its in-scope traits in trait order."""
    p = fam['p']
    K = fam['max_parents']
    R = np.asarray(kw['R'], float)
    mu = np.asarray(kw['mu'], float).reshape(p)
    ou = kw.get('model') == 'ou'
    full = (1 << p) - 1
    cm = fam.get('child_mask')
    pm = fam.get('parent_mask')
    traits = lambda mask: [t for t in range(p) if (int(mask) >> t) & 1]
    out = [(np.zeros((m, m)), np.zeros(m), np.zeros(1)) for m in dims]
    for f in range(len(fam['cluster'])):
        c = fam['cluster'][f]
        np_ = fam['n_parents'][f]
        cp = fam['child_pos'][f]
        J, h, g = out[c]
        O = traits(cm[f] if cm is not None else full)
        if not O:
            continue
        cs, pos, masks = [1.0], [cp], [cm[f] if cm is not None else full]
        if np_ == 0:
            V = R[fam['color'][f * K]].copy()
            z = mu.copy()
        else:
            V = np.zeros((p, p))
            z = np.zeros(p)
            for k in range(np_):
                t, gam, col = fam['length'][f * K + k], fam['gamma'][f * K + k], fam['color'][f * K + k]
                if ou:
                    a = np.exp(-kw['alpha'] * t)
                    qc, vc, wc = gam * a, gam * gam * (1 - a * a), gam * (1 - a)
                else:
                    qc, vc, wc = gam, gam * gam * t, 0.0
                V += vc * R[col]
                if ou:
                    z += wc * np.asarray(kw['theta'], float)
                pp = fam['parent_pos'][f * K + k]
                if pp < 0:
                    z += qc * mu
                cs.append(-qc)
                pos.append(pp)
                masks.append(pm[f * K + k] if pm is not None else full)
            if cp < 0:
                z = z - np.nan_to_num(data[fam['data_row'][f]])
        j = np.linalg.inv(V[np.ix_(O, O)])
        zO = z[O]
        jz = j @ zO
        for a in range(len(cs)):
            if pos[a] < 0:
                continue
            ia = [pos[a] + traits(masks[a]).index(t) for t in O]
            h[ia] += cs[a] * jz
            for b in range(len(cs)):
                if pos[b] < 0:
                    continue
                ib = [pos[b] + traits(masks[b]).index(t) for t in O]
                J[np.ix_(ia, ib)] += cs[a] * cs[b] * j
        g[0] += -0.5 * (len(O) * LOG2PI + np.linalg.slogdet(V[np.ix_(O, O)])[1] + zO @ jz)
    return out


@pytest.mark.parametrize("graph", ["cliquetree", "bethe"])
@pytest.mark.parametrize("which,p", [("bm_fixed", 3), ("bm_random_root", 2), ("bm_improper_root", 2), ("hetero", 4),
                                     ("hetero_random_root", 2), ("ou_fixed", 1), ("ou_random_root", 1)])
def test_family_table_and_closed_form(graph, which, p):
    rng = np.random.default_rng(zlib.crc32(f"{graph}-{which}-{p}".encode()))
    net = ON.random_network(24, 6, rng)
    model = _models(p, rng, net, which)
    taxa = net.tip_names
    tbl = [list(rng.normal(size=len(taxa))) for _ in range(p)]
    cg = OCG.cliquetree(net) if graph == "cliquetree" else OCG.bethe(net)
    ocgb = oracle_setup(net, cg, model, tbl, taxa)
    fam, data, kw = lg_inputs_from_oracle(P, net, ocgb, model, tbl, taxa)
    assert fam["max_parents"] == 2 and np.any(fam["n_parents"] == 2)
    dims = [b.dimension for b in ocgb.belief[:ocgb.nclusters]]
    for i, (J, h, g) in enumerate(fill_numpy(fam, data, kw, dims)):
        ob = ocgb.belief[i]
        for x, y in ((J, ob.J), (h, ob.h), (g, ob.g)):
            if x.size:
                assert np.max(np.abs(x - y)) <= 1e-10 * max(1.0, np.max(np.abs(y))), (i, x, y)


def missing_pattern(net, p, rng, which):
    """Tip data with missing values the REFERENCE can handle.  Scattered values go missing only where every internal
    node keeps its full scope (a trait observed somewhere below each of them).  Where a whole subtree loses a trait -- its
    internal nodes then have a partial scope -- the reference's marginalisation of the parent's trait (src/beliefs.jl:840-852)
    meets a precision that is zero only up to rounding ("fixit" in the source) and fails its Cholesky for generic numbers;
    the reference's own test of that situation (test/test_calibration.jl:131-185) is the case used for it below."""
    taxa = net.tip_names
    tbl = [[float(x) for x in rng.normal(size=len(taxa))] for _ in range(p)]
    row = {name: r for r, name in enumerate(taxa)}

    def tips_below(n):
        out, stack = [], [n]
        while stack:
            x = stack.pop()
            if x.leaf:
                out.append(x.name)
            stack.extend(net.children(x))
        return out
    for r in range(len(taxa)):
        for v in range(p):
            if rng.random() < 0.25:
                tbl[v][r] = None
    for n in net.vec_node:                      # keep every internal node's scope full
        if not n.leaf:
            below = tips_below(n)
            for v in range(p):
                if all(tbl[v][row[t]] is None for t in below):
                    tbl[v][row[below[0]]] = 0.1 * (v + 1)
    return tbl, taxa


@pytest.mark.parametrize("graph", ["cliquetree", "bethe"])
@pytest.mark.parametrize("which,p", [("bm_fixed", 3), ("bm_random_root", 2), ("hetero", 4), ("bm_improper_root", 3)])
def test_family_table_with_missing_data(graph, which, p):
    """Missing tip values: the masks of the family table and the masked closed form against the oracle's assignfactors!."""
    rng = np.random.default_rng(zlib.crc32(f"miss-{graph}-{which}-{p}".encode()))
    net = ON.random_network(20, 4, rng)
    model = _models(p, rng, net, which)
    tbl, taxa = missing_pattern(net, p, rng, which)
    cg = OCG.cliquetree(net) if graph == "cliquetree" else OCG.bethe(net)
    ocgb = oracle_setup(net, cg, model, tbl, taxa)
    fam, data, kw = lg_inputs_from_oracle(P, net, ocgb, model, tbl, taxa)
    assert fam.get("child_mask") is not None and np.isnan(data).any()
    dims = [b.dimension for b in ocgb.belief[:ocgb.nclusters]]
    for i, (J, h, g) in enumerate(fill_numpy(fam, data, kw, dims)):
        ob = ocgb.belief[i]
        for x, y in ((J, ob.J), (h, ob.h), (g, ob.g)):
            if x.size:
                assert np.max(np.abs(x - y)) <= 1e-10 * max(1.0, np.max(np.abs(y))), (i, x, y)


@pytest.mark.parametrize("variant", ["improper", "fixed"])
@pytest.mark.parametrize("graph", ["cliquetree", "joingraph"])
def test_family_table_partial_internal_scopes_level3_golden(variant, graph):
    """test/test_calibration.jl:131-185: y2 missing at B leaves the hybrid nodes above B with one of two traits in scope
    (partial scopes of internal nodes, hybrid families among them): masked closed form == the oracle's assignfactors!."""
    from helpers import goldens, make_model
    g = goldens()["calibration_level3_joingraph"]
    net = ON.read_newick(g["net"])
    net.set_preorder(g["preorder"])
    cg = OCG.cliquetree(net) if graph == "cliquetree" else OCG.joingraph(net, 3)
    model = make_model(g["model_" + variant])
    tbl = [g["y1"], g["y2"]]
    ocgb = oracle_setup(net, cg, model, tbl, g["taxa"])
    fam, data, kw = lg_inputs_from_oracle(P, net, ocgb, model, tbl, g["taxa"])
    assert any(b.dimension % 2 for b in ocgb.belief[:ocgb.nclusters])         # partial scopes are there
    dims = [b.dimension for b in ocgb.belief[:ocgb.nclusters]]
    for i, (J, h, gg) in enumerate(fill_numpy(fam, data, kw, dims)):
        ob = ocgb.belief[i]
        for x, y in ((J, ob.J), (h, ob.h), (gg, ob.g)):
            if x.size:
                assert np.max(np.abs(x - y)) <= 1e-10 * max(1.0, np.max(np.abs(y))), (i, x, y)
