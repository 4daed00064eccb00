# bench/reference_calibrate.jl -- the OPPORTUNISTIC third baseline of BASELINE.md section 3.3.
#
# Times the real `calibrate!()` of PhyloGaussianBeliefProp.jl on beliefs produced by THIS repo's generator, bypassing the
# reference's O(n^2) set-up (src/beliefs.jl:521,526,536; src/clustergraph.jl:92-104).  It runs only where `julia` with the
# package is already installed (it cannot be installed here: no network); bench/run_reference_calibrate.sh checks
# `which julia` first and prints {"available": false} otherwise.  NOT exercised in the build container (no Julia): the
# constructors below are the reference's own (src/beliefs.jl:102-132, src/clustergraphbeliefs.jl:89-109).
#
#   python bench/dump_workload.py out.bin [--ntips 50000 --traits 16]     # this repo: dims, scopes, schedule, packed (J,h,g)
#   julia bench/reference_calibrate.jl out.bin [repetitions]
#
# File layout (little endian): int64 header [nclusters, nsepsets, p, nedges, packed_len], then
#   int32 dims[nb], int32 cluster_nodes (2 per cluster: child, parent preorder labels; 0 = absent),
#   int32 child_dim[nc], int32 sepset_node[ns], int32 sepset_clusters[2 ns], int32 pa[nedges], int32 ch[nedges] (0-based),
#   float64 packed[packed_len].
using PhyloGaussianBeliefProp
const PGBP = PhyloGaussianBeliefProp
using Printf

function load(path)
    io = open(path, "r")
    hdr = Vector{Int64}(undef, 5); read!(io, hdr)
    nc, ns, p, ne, plen = hdr
    rd(T, n) = (v = Vector{T}(undef, n); read!(io, v); v)
    dims = rd(Int32, nc + ns); cnodes = reshape(rd(Int32, 2nc), 2, nc); cdim = rd(Int32, nc)
    snode = rd(Int32, ns); sepcl = reshape(rd(Int32, 2ns), 2, ns)
    pa = rd(Int32, ne); ch = rd(Int32, ne); packed = rd(Float64, plen)
    close(io)
    beliefs = PGBP.CanonicalBelief[]
    off = 0
    labelof(i) = Symbol("c", i)
    for i in 1:nc
        labs = Int32[x for x in cnodes[:, i] if x > 0]
        insc = falses(p, length(labs))
        m = Int(dims[i])
        # (child, parent): which of the two is in scope follows from the dimensions (tips and the fixed root are not)
        length(labs) >= 1 && (insc[:, 1] .= cdim[i] > 0)
        length(labs) >= 2 && (insc[:, 2] .= (m - cdim[i]) > 0)
        b = PGBP.CanonicalBelief(labs, p, insc, PGBP.bclustertype, labelof(i))
        b.J .= reshape(packed[off+1:off+m*m], m, m); b.h .= packed[off+m*m+1:off+m*m+m]; b.g[1] = packed[off+m*m+m+1]
        off += m*m + m + 1
        push!(beliefs, b)
    end
    for k in 1:ns
        m = Int(dims[nc + k])
        b = PGBP.CanonicalBelief(Int32[snode[k]], p, trues(p, 1) .& (m > 0), PGBP.bsepsettype,
                                 (labelof(sepcl[1, k] + 1), labelof(sepcl[2, k] + 1)))
        b.J .= reshape(packed[off+1:off+m*m], m, m); b.h .= packed[off+m*m+1:off+m*m+m]; b.g[1] = packed[off+m*m+m+1]
        off += m*m + m + 1
        push!(beliefs, b)
    end
    cgb = PGBP.ClusterGraphBelief(beliefs, Int[], Vector{Int}[], falses(0), Vector{Int}[])
    spt = (labelof.(pa .+ 1), labelof.(ch .+ 1), Int.(pa .+ 1), Int.(ch .+ 1))   # src/clustergraph.jl:885-894
    return cgb, spt, ne
end

function main()
    cgb, spt, ne = load(ARGS[1])
    reps = length(ARGS) > 1 ? parse(Int, ARGS[2]) : 5
    PGBP.calibrate!(cgb, [spt], 1)                       # compile
    times = Float64[]
    for _ in 1:reps
        PGBP.init_beliefs_reset_fromfactors!(cgb); PGBP.init_messagecalibrationflags_reset!(cgb, true)
        push!(times, @elapsed PGBP.calibrate!(cgb, [spt], 1))
    end
    t = sort(times)[(reps + 1) ÷ 2]
    ll = PGBP.integratebelief!(cgb, Int(spt[3][1]))[2]
    @printf("{\"available\": true, \"kind\": \"reference\", \"messages_per_s\": %.6e, \"median_s\": %.6e, \"messages\": %d, \"threads\": %d, \"loglik\": %.15e}\n",
            2ne / t, t, 2ne, Threads.nthreads(), ll)
end
main()
