"""Host-side mirror of the belief types of src/beliefs.jl that the hot path touches.

Python names follow the reference; a trailing underscore stands for Julia's `!`.
Indices are 0-based here (positions in h/J, belief indices, cluster indices);
node labels stay the 1-based preorder indices of the reference."""
import numpy as np

bclustertype, bsepsettype = "cluster", "sepset"


class CanonicalBelief:
    """CanonicalBelief (src/beliefs.jl:72-132): C(x; J, h, g) = exp(-x'Jx/2 + h'x + g) over the
    in-scope traits of `nodelabel`; `inscope` is ntraits x nnodes.  h, J, g start at 0
    (the constant function 1).  Once the belief belongs to a ClusterGraphBelief, h/J/g are
    views into its packed host mirror of the device state."""

    # h, J, g of a belief that belongs to a device-backed ClusterGraphBelief are read THROUGH its owner: after a device
    # call the owner only marks its host mirror stale, and the first read of a belief fetches that one record
    # (clustergraphbeliefs.py: _refresh) -- an alias held since before the call sees the new values, as in the reference,
    # where every update is in place (ownership)
    def _fresh_field(self, name):
        o = self.__dict__.get("_owner")
        if o is not None:
            o._refresh(self.__dict__["_index"])
        return self.__dict__[name]

    J = property(lambda self: self._fresh_field("_J"), lambda self, v: self.__dict__.__setitem__("_J", v))
    h = property(lambda self: self._fresh_field("_h"), lambda self, v: self.__dict__.__setitem__("_h", v))
    g = property(lambda self: self._fresh_field("_g"), lambda self, v: self.__dict__.__setitem__("_g", v))

    def __init__(self, nodelabel, ntraits, inscope, btype, metadata):
        self.nodelabel = [int(x) for x in nodelabel]
        self.ntraits = int(ntraits)
        self.inscope = np.asarray(inscope, dtype=bool).reshape(self.ntraits, len(self.nodelabel))
        m = int(self.inscope.sum())
        self.mu = np.zeros(m)
        self.h = np.zeros(m)
        self.J = np.zeros((m, m), order="F")
        self.g = np.zeros(1)
        self.type = btype
        self.metadata = metadata

    @property
    def dimension(self):
        return int(self.h.shape[0])


def scopeindex(sepset, cluster):
    """scopeindex(sepset, cluster) (src/beliefs.jl:389-405): positions, in the cluster's
    variables, of the sepset's variables. Raises ValueError (ErrorException in the reference)
    if labels are out of order / not a subset / out of the cluster's scope."""
    sub_labels, bel_labels = sepset.nodelabel, cluster.nodelabel
    try:
        node_index = [bel_labels.index(lab) for lab in sub_labels]
    except ValueError:
        raise ValueError("subset_labels not a subset of belief_labels")
    if any(b <= a for a, b in zip(node_index, node_index[1:])):
        raise ValueError("subset labels come in a different order in the belief")
    if np.any(sepset.inscope & ~cluster.inscope[:, node_index]):
        raise ValueError("some variable(s) in subset's scope yet not in full belief's scope")
    mask = np.zeros_like(cluster.inscope)
    mask[:, node_index] = sepset.inscope
    return np.nonzero(mask.T.reshape(-1)[cluster.inscope.T.reshape(-1)])[0].astype(np.int32)


class MessageResidual:
    """MessageResidual (src/beliefs.jl:895-924): view of one directed message's residual
    (dh, dJ, kldiv, iscalibrated_resid) inside the host mirror of the device residual pool."""

    def __init__(self, owner, msg_id, s):
        self._o, self._d, self._s = owner, msg_id, s

    @property
    def dJ(self):
        r = self._o._residual_record(self._d)
        return r[: self._s * self._s].reshape(self._s, self._s, order="F")

    @property
    def dh(self):
        r = self._o._residual_record(self._d)
        return r[self._s * self._s: self._s * self._s + self._s]

    @property
    def iscalibrated_resid(self):
        return bool(self._o._residual_words(self._d)[0])

    @property
    def kldiv(self):
        return float(self._o._residual_words(self._d)[1])

    @property
    def iscalibrated_kl(self):
        return bool(self._o._residual_words(self._d)[2])
