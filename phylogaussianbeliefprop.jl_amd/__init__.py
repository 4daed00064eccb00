"""
MI355X-native engine for the canonical-form Gaussian belief-propagation hot path of
PhyloGaussianBeliefProp.jl: host-side mirror (Python) of the reference's
ClusterGraphBelief / calibrate! / propagate_belief! surface over the C ABI of
include/pgbp.h (HIP kernels in csrc/).  Import as `pgbp_amd` (see /pgbp_amd.py).
"""
from ._lib import LIB_PATH, PgbpError, load
from .beliefs import CanonicalBelief, MessageResidual, bclustertype, bsepsettype, scopeindex
from .beliefupdates import BPPosDefException, integratebelief_, propagate_belief_
from .calibration import calibrate_, propagate_1traversal_postorder_, propagate_1traversal_preorder_
from .clustergraph import (bethe, cliquetree, default_rootcluster, ltrip, default_rootcluster_nodes, joingraph, moralize, nodesubtree_clusterlist,
                           spanningtree_clusterlist, spanningtrees_clusterlist, triangulate_minfill)
from .clustergraphbeliefs import ClusterGraphBelief
from .factors import lg_families
from .optimize import calibrate_optimize_cliquetree_, calibrate_optimize_clustergraph_
from .networks import (NetArrays, allocate_scopes, random_level3_network, random_level3_network_varied, read_newick,
                       simulate_bm_network)
from .regularization import (regularizebeliefs_bycluster_, regularizebeliefs_bynodesubtree_,
                             regularizebeliefs_onschedule_)

__all__ = [
    "CanonicalBelief", "MessageResidual", "ClusterGraphBelief", "BPPosDefException", "scopeindex",
    "bclustertype", "bsepsettype", "calibrate_", "propagate_1traversal_postorder_",
    "propagate_1traversal_preorder_", "propagate_belief_", "regularizebeliefs_bycluster_",
    "regularizebeliefs_bynodesubtree_", "regularizebeliefs_onschedule_", "default_rootcluster",
    "spanningtree_clusterlist", "spanningtrees_clusterlist", "joingraph", "bethe", "cliquetree", "ltrip", "moralize", "triangulate_minfill",
    "nodesubtree_clusterlist", "default_rootcluster_nodes", "integratebelief_", "lg_families", "calibrate_optimize_cliquetree_", "calibrate_optimize_clustergraph_", "NetArrays", "allocate_scopes", "random_level3_network", "random_level3_network_varied", "read_newick", "simulate_bm_network", "load", "LIB_PATH", "PgbpError",
]
