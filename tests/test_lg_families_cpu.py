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
    p=fam['p']; K=fam['max_parents']
    R=np.asarray(kw['R'],float); mu=np.asarray(kw['mu'],float).reshape(p)
    ou=kw.get('model')=='ou'
    out=[(np.zeros((m,m)),np.zeros(m),np.zeros(1)) for m in dims]
    for f in range(len(fam['cluster'])):
        c=fam['cluster'][f]; np_=fam['n_parents'][f]; cp=fam['child_pos'][f]
        J,h,g=out[c]
        cs=[1.0]; pos=[cp]
        if np_==0:
            V=R[fam['color'][f*K]]; z=mu.copy()
        else:
            V=np.zeros((p,p)); z=np.zeros(p)
            for k in range(np_):
                t=fam['length'][f*K+k]; gam=fam['gamma'][f*K+k]; col=fam['color'][f*K+k]
                if ou:
                    a=np.exp(-kw['alpha']*t); qc=gam*a; vc=gam*gam*(1-a*a); wc=gam*(1-a)
                else:
                    qc=gam; vc=gam*gam*t; wc=0.0
                V+=vc*R[col]
                if ou: z+=wc*np.asarray(kw['theta'],float)
                pp=fam['parent_pos'][f*K+k]
                if pp<0: z+=qc*mu
                cs.append(-qc); pos.append(pp)
            if cp<0: z-=data[fam['data_row'][f]]
        j=np.linalg.inv(V); jz=j@z
        for a in range(len(cs)):
            if pos[a]<0: continue
            h[pos[a]:pos[a]+p]+=cs[a]*jz
            for b in range(len(cs)):
                if pos[b]<0: continue
                J[pos[a]:pos[a]+p,pos[b]:pos[b]+p]+=cs[a]*cs[b]*j
        g[0]+=-0.5*(p*LOG2PI+np.linalg.slogdet(V)[1]+z@jz)
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


def test_family_table_refuses_missing_data():
    """A trait missing at both tips of a cherry leaves their parent with one of two traits in scope
    (src/beliefs.jl:829-857 marginalises the factor there): that path stays on the host."""
    rng = np.random.default_rng(1)
    net = ON.read_newick("((a:1.0,b:0.5):1.0,(c:0.3,d:0.4):0.7);")
    model = _models(2, rng, net, "bm_fixed")
    taxa = net.tip_names
    tbl = [[None if t in ("a", "b") else 0.3 for t in taxa], [0.1 * k for k in range(len(taxa))]]
    ocgb = oracle_setup(net, OCG.cliquetree(net), model, tbl, taxa)
    with pytest.raises(ValueError, match="missing data"):
        lg_inputs_from_oracle(P, net, ocgb, model, [[0.0 if v is None else v for v in col] for col in tbl], taxa)
