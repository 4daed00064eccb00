"""Shared test helpers (tests only): golden loading, model construction from the
fixture dictionaries, building oracle cluster-graph beliefs."""
import json
import os

import numpy as np

from oracle import beliefs as OB
from oracle import calibration as OC
from oracle import clustergraph as OCG
from oracle import models as OM
from oracle import network as ON

HERE = os.path.dirname(os.path.abspath(__file__))


def goldens():
    with open(os.path.join(HERE, "golden", "reference_goldens.json")) as f:
        return json.load(f)


def _num(x):
    if isinstance(x, str):
        return float(x)
    if isinstance(x, list):
        return [_num(v) for v in x]
    return x


def make_model(d):
    k = d["kind"]
    v = _num(d.get("v")) if "v" in d else None
    if k == "UnivariateBM":
        return OM.UnivariateBrownianMotion(d["sigma2"], d["mu"], v)
    if k == "MvDiagBM":
        return OM.MvDiagBrownianMotion(d["R"], d["mu"], v)
    if k == "MvFullBM":
        return OM.MvFullBrownianMotion(d["R"], d["mu"], v)
    if k == "UnivariateOU":
        return OM.UnivariateOrnsteinUhlenbeck(d["sigma2"], d["alpha"], d["theta"], d["mu"], v)
    if k == "HeteroBM":
        return OM.HeterogeneousBrownianMotion(d["rates"], {int(a): b for a, b in d["colors"].items()}, d["mu"], v)
    raise KeyError(k)


def oracle_setup(net, cg, model, tbl, taxa):
    """allocatebeliefs + assignfactors + ClusterGraphBelief with the oracle."""
    b, (n2c, n2f, n2fix, n2d, c2n) = OB.allocatebeliefs(tbl, taxa, net, cg, model)
    OB.assignfactors(b, model, tbl, taxa, net, n2c, n2f, n2fix)
    return OB.ClusterGraphBelief(b, n2c, n2f, n2fix, c2n)
