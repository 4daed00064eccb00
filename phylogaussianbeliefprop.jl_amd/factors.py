"""Host side of the device factor assignment: the static table of node families that
`pgbp_lg_setup` (include/pgbp.h) takes, derived from what `allocatebeliefs` (src/beliefs.jl:478-594)
returns -- node2cluster, node2family, node2fixed -- and the cluster beliefs' scopes."""
import numpy as np


def lg_families(beliefs, node2cluster, node2family, node2fixed, parent_edges, data_row, p, n_rates=1,
                root_prior_color=None, data=None):
    """One entry per node family that carries a factor, in the order of the loop of assignfactors!
    (src/beliefs.jl:797: `for (ni, ci) in enumerate(node2cluster)`).

    beliefs: cluster beliefs (CanonicalBelief: nodelabel, inscope), indexed by node2cluster's entries;
    node2family[ni] = [child, parent_1, ...] 1-based preorder labels; node2fixed[ni]: tip or fixed root;
    parent_edges[ni] = [(length, gamma, color), ...] aligned with node2family[ni][1:] (color: 0-based index
    into the model's variance rates); data_row[ni]: row of the tip's data (tips only);
    root_prior_color: index among the rates of the root prior variance when the root is random and its prior proper
    (None: fixed root, or an improper prior -- no factor: src/evomodels/evomodels.jl:383-385).

    Missing tip values: pass `data` ([n_rows, p], NaN where missing; one pattern for all sites).  The table then carries
    scope masks (bit t = trait t): `child_mask` = an internal child's traits in scope / a tip's observed traits,
    `parent_mask` = each parent's traits in scope (src/beliefs.jl:551-559); the factor keeps the child_mask components of
    its residual, which is what absorbleaf! (src/beliefupdates.jl:266-274) and the partial-scope marginalisation of
    assignfactors! (src/beliefs.jl:829-857) leave for the reference's models."""
    K = max([1] + [len(nf) - 1 for nf in node2family])
    out = {k: [] for k in ("cluster", "n_parents", "child_pos", "data_row")}
    ppos, length, gamma, color = [], [], [], []
    pos_cache = {}

    if p > 64:
        raise ValueError("at most 64 traits")
    full = (1 << p) - 1
    bits = 1 << np.arange(p, dtype=object)
    partial = [False]
    cmask, pmask = [], []

    def positions(ci):
        """label -> (first variable or -1, scope mask) of every node of cluster ci"""
        if ci not in pos_cache:
            b = beliefs[ci]
            insc = np.asarray(b.inscope, bool)
            ndim = insc.sum(axis=0)
            if np.any((ndim != 0) & (ndim != p)):
                partial[0] = True
            start = np.concatenate([[0], np.cumsum(ndim)])
            pos_cache[ci] = {lab: ((int(start[j]) if ndim[j] else -1), int(sum(bits[insc[:, j]])))
                             for j, lab in enumerate(b.nodelabel)}
        return pos_cache[ci]

    def observed(row):
        if data is None:
            return full
        ok = np.isfinite(np.asarray(data, float)[row])
        if not ok.all():
            partial[0] = True
        return int(sum(bits[ok]))

    for ni, ci in enumerate(node2cluster):
        nf = node2family[ni]
        pos = positions(ci)
        if len(nf) == 1:
            if ni != 0:
                raise ValueError("only the root node can belong to a family of size 1")
            if node2fixed[0] or root_prior_color is None:
                continue
            out["cluster"].append(ci); out["n_parents"].append(0); out["child_pos"].append(pos[nf[0]][0])
            out["data_row"].append(-1)
            ppos += [-1] * K; length += [1.0] * K; gamma += [1.0] * K
            color += [int(root_prior_color)] + [0] * (K - 1)
            cmask.append(pos[nf[0]][1]); pmask += [full] * K
            continue
        fixed_child = bool(node2fixed[ni])
        out["cluster"].append(ci)
        out["n_parents"].append(len(nf) - 1)
        out["child_pos"].append(-1 if fixed_child else pos[nf[0]][0])
        out["data_row"].append(int(data_row[ni]) if fixed_child else -1)
        cmask.append(observed(int(data_row[ni])) if fixed_child else pos[nf[0]][1])
        row_p, row_l, row_g, row_c, row_m = [-1] * K, [1.0] * K, [1.0] * K, [0] * K, [full] * K
        for k, (pl, (t, gam, col)) in enumerate(zip(nf[1:], parent_edges[ni])):
            row_p[k] = -1 if node2fixed[pl - 1] else pos[pl][0]
            row_m[k] = full if node2fixed[pl - 1] else pos[pl][1]
            row_l[k], row_g[k], row_c[k] = float(t), float(gam), int(col)
        ppos += row_p; length += row_l; gamma += row_g; color += row_c; pmask += row_m
    masks = {}
    if partial[0]:
        masks = dict(child_mask=np.array(cmask, np.uint64), parent_mask=np.array(pmask, np.uint64))
    return dict(**masks, p=int(p), max_parents=K, n_rates=int(n_rates), cluster=np.array(out["cluster"], np.int32),
                n_parents=np.array(out["n_parents"], np.int32), child_pos=np.array(out["child_pos"], np.int32),
                data_row=np.array(out["data_row"], np.int32), parent_pos=np.array(ppos, np.int32),
                length=np.array(length, np.float64), gamma=np.array(gamma, np.float64), color=np.array(color, np.int32))
