"""Mirror of the entry points of src/beliefupdates.jl that live on the device."""


class BPPosDefException(Exception):
    """BPPosDefException (src/beliefupdates.jl:11-22)."""

    def __init__(self, msg, info):
        super().__init__(msg)
        self.msg = msg
        self.info = int(info)

    def showerror(self):
        tail = "Hermitian." if self.info == -1 else "positive definite."
        return f"BPPosDefException: {self.msg}\nmatrix is not {tail}"


def propagate_belief_(beliefs, cluster_to, sepset, cluster_from, withresidual=True):
    """propagate_belief!(cluster_to, sepset, cluster_from, residual) (src/beliefupdates.jl:634-649)
    on belief INDICES of the ClusterGraphBelief `beliefs`: returns None, or the BPPosDefException
    (returned, not raised).  With withresidual=False the 3-argument form (:650-665): raises it."""
    flag = beliefs._propagate(cluster_to, sepset, cluster_from)
    if flag is not None and not withresidual:
        raise flag
    return flag


def integratebelief_(beliefs, beliefindex):
    """integratebelief!(obj::ClusterGraphBelief, beliefindex) (src/clustergraphbeliefs.jl:194):
    (mu, norm); the belief's mu is updated."""
    return beliefs.integratebelief_(beliefindex)
