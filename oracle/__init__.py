"""
oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement (numpy + a plain-C sequential engine under oracle/c/) of the
canonical-form Gaussian belief-propagation hot path of
JuliaPhylo/PhyloGaussianBeliefProp.jl.  Every function cites the reference
file:line it follows (paths relative to the reference repository root).

Who may import / call / link this package:
    * tests/                       (as the checker)
    * __graft_entry__.smoke()      (as the checker)
    * bench.py's cpu_baseline leg  (as the thing timed *beside* the GPU path)
Nothing under phylogaussianbeliefprop.jl_amd/ (the product) may import it, and
the product must fail loudly when its HIP library is missing: there is no CPU
fallback.

Parity pinning: the reference is 100% Julia and Julia is absent from the build
image, so the reference itself cannot be executed.  The restatement is pinned by
(i) every literal golden value the reference's own tests/doctests hold for this
path (tests/golden/reference_goldens.json, checked by tests/test_oracle_goldens.py)
and (ii) an independent dense multivariate-normal log-likelihood (the recipe the
reference's authors used to generate those goldens, see test comments cited in
oracle/densemvn.py).
"""
