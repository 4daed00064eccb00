# PGBPDevice.jl -- the reference-side binding of include/pgbp.h.
#
# What a PhyloGaussianBeliefProp.jl maintainer adds (e.g. as ext/PGBPDeviceExt.jl) so that
# calibrate!/propagate_belief!/integratebelief! run on an MI355X.  It only packs the package's own
# CanonicalBelief arrays into flat buffers, @ccall's libpgbp.so and writes results back into the SAME
# Julia arrays, so callers holding aliases (e.g. tests that read b[6] after calibrating) keep working.
# NOT exercised in the build container (no Julia there); the Python/ctypes host mirror in this
# directory's parent performs the identical call sequence and is what the test-suite drives.
module PGBPDevice

import PhyloGaussianBeliefProp as PGBP
const LIB = get(ENV, "PGBP_LIB", joinpath(@__DIR__, "..", "csrc", "libpgbp.so"))

struct Desc            # pgbp_desc
    n_clusters::Int32; n_sepsets::Int32
    dims::Ptr{Int32}; sepset_clusters::Ptr{Int32}
    scope_off::Ptr{Int64}; scope_idx::Ptr{Int32}
    n_sites::Int32; device::Int32
end
struct Opts            # pgbp_opts
    auto_stop::Int32; update_residualnorm::Int32; update_residualkldiv::Int32; reserved::Int32
    atol::Float64
end
struct Result          # pgbp_result
    succ::Int32; iscal::Int32; iter_reached::Int32; tree_reached::Int32
    fail_iter::Int32; fail_tree::Int32; fail_dir::Int32; fail_edge::Int32; fail_info::Int32; reserved::Int32
end

mutable struct DeviceClusterGraphBelief
    cgb::PGBP.ClusterGraphBelief       # the package's own object: stays the owner of the arrays
    handle::Ptr{Cvoid}
    packed::Vector{Float64}
    offsets::Vector{Int}
    schedule_set::Any
    keep::Any                          # arrays the device factor fill's static table points into
end

check(h, rc) = rc == 0 || error(unsafe_string(@ccall LIB.pgbp_last_error(h::Ptr{Cvoid})::Cstring))

"Wrap a ClusterGraphBelief (src/clustergraphbeliefs.jl:89-109): scopeindex is evaluated once per (sepset, side)."
function DeviceClusterGraphBelief(cgb::PGBP.ClusterGraphBelief; device::Integer=0)
    b = cgb.belief; nc = cgb.nclusters; nb = length(b)
    all(x -> x isa PGBP.CanonicalBelief, b) || error("GeneralizedBelief (degenerate edges) is not supported on the device")
    dims = Int32[length(x.h) for x in b]
    sepcl = Int32[]; off = Int64[0]; idx = Int32[]
    for j in (nc+1):nb
        (l1, l2) = b[j].metadata
        for c in (cgb.cdict[l1], cgb.cdict[l2])
            push!(sepcl, c - 1)
            ind = PGBP.scopeindex(b[j], b[c])            # src/beliefs.jl:389-405, once
            append!(idx, Int32.(ind .- 1)); push!(off, off[end] + length(ind))
        end
    end
    isempty(idx) && push!(idx, 0); isempty(sepcl) && append!(sepcl, (0, 0))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve dims sepcl off idx begin
        d = Desc(nc, nb - nc, pointer(dims), pointer(sepcl), pointer(off), pointer(idx), 1, device)
        rc = @ccall LIB.pgbp_create(Ref(d)::Ref{Desc}, h::Ref{Ptr{Cvoid}})::Cint
        rc == 0 || error(unsafe_string(@ccall LIB.pgbp_last_error(C_NULL::Ptr{Cvoid})::Cstring))
    end
    offsets = cumsum(vcat(0, [m*m + m + 1 for m in Int.(dims)]))
    obj = DeviceClusterGraphBelief(cgb, h[], zeros(offsets[end]), offsets, nothing, nothing)
    push!(obj); finalizer(o -> @ccall(LIB.pgbp_destroy(o.handle::Ptr{Cvoid})::Cvoid), obj)
    return obj
end

"host -> device: J (column-major, as stored), h, g of every belief; cluster part becomes the factors"
function Base.push!(o::DeviceClusterGraphBelief)
    for (i, x) in enumerate(o.cgb.belief)
        m = length(x.h); p = o.offsets[i]
        copyto!(o.packed, p + 1, x.J, 1, m*m); copyto!(o.packed, p + m*m + 1, x.h, 1, m); o.packed[p + m*m + m + 1] = x.g[1]
    end
    check(o.handle, @ccall LIB.pgbp_set_beliefs(o.handle::Ptr{Cvoid}, o.packed::Ptr{Float64}, 1::Int32)::Cint)
end

"device -> the SAME Julia arrays (aliases held by callers stay valid), residuals and flags included"
function pull!(o::DeviceClusterGraphBelief)
    check(o.handle, @ccall LIB.pgbp_get_beliefs(o.handle::Ptr{Cvoid}, o.packed::Ptr{Float64})::Cint)
    for (i, x) in enumerate(o.cgb.belief)
        m = length(x.h); p = o.offsets[i]
        copyto!(x.J, 1, o.packed, p + 1, m*m); copyto!(x.h, 1, o.packed, p + m*m + 1, m); x.g[1] = o.packed[p + m*m + m + 1]
    end
    nm = 2 * (length(o.cgb.belief) - o.cgb.nclusters)
    rs = Int(@ccall LIB.pgbp_residual_size(o.handle::Ptr{Cvoid})::Int64)
    res = zeros(max(rs, 1)); flags = zeros(Int32, max(nm, 1))
    check(o.handle, @ccall LIB.pgbp_get_residuals(o.handle::Ptr{Cvoid}, res::Ptr{Float64}, flags::Ptr{Int32}, C_NULL::Ptr{Float64}, C_NULL::Ptr{Int32})::Cint)
    p = 0
    for (k, j) in enumerate((o.cgb.nclusters+1):length(o.cgb.belief)), (dir, key) in enumerate((o.cgb.belief[j].metadata, reverse(o.cgb.belief[j].metadata)))
        mr = o.cgb.messageresidual[key]; s = length(mr.Δh)       # key = (receiver, sender)
        copyto!(mr.ΔJ, 1, res, p + 1, s*s); copyto!(mr.Δh, 1, res, p + s*s + 1, s); p += s*s + s
        mr.iscalibrated_resid[1] = flags[2*(k-1) + dir] != 0
    end
end

function set_schedule!(o::DeviceClusterGraphBelief, schedule)
    o.schedule_set === schedule && return
    off = Int32[0]; pa = Int32[]; ch = Int32[]
    for spt in schedule                                   # (pa_lab, ch_lab, pa_j, ch_j): src/clustergraph.jl:885-894
        append!(pa, Int32.(spt[3] .- 1)); append!(ch, Int32.(spt[4] .- 1)); push!(off, length(pa))
    end
    check(o.handle, @ccall LIB.pgbp_set_schedule(o.handle::Ptr{Cvoid}, length(schedule)::Int32, off::Ptr{Int32}, pa::Ptr{Int32}, ch::Ptr{Int32})::Cint)
    o.schedule_set = schedule
end

"calibrate!(beliefs, schedule, niter; auto, info, verbose, ...) -> (succ, iscal)   (src/calibration.jl:35-60)"
function PGBP.calibrate!(o::DeviceClusterGraphBelief, schedule::AbstractVector, niter::Integer=1;
        auto::Bool=false, info::Bool=false, verbose::Bool=true,
        update_residualnorm::Bool=true, update_residualkldiv::Bool=false)
    update_residualkldiv && error("update_residualkldiv is not available on the device")
    set_schedule!(o, schedule)
    r = Ref(Result(0,0,0,0,0,0,0,0,0,0))
    check(o.handle, @ccall LIB.pgbp_calibrate(o.handle::Ptr{Cvoid}, niter::Int32,
        Ref(Opts(auto, update_residualnorm, 0, 0, 1e-5))::Ref{Opts}, r::Ref{Result})::Cint)
    pull!(o)
    res = r[]
    if res.succ == 0
        spt = schedule[res.fail_tree]; i = res.fail_edge + 1
        sender = res.fail_dir == 0 ? spt[4][i] : spt[3][i]; receiver = res.fail_dir == 0 ? spt[3][i] : spt[4][i]
        sep = o.cgb.belief[PGBP.sepsetindex(spt[1][i], spt[2][i], o.cgb)]
        integ = setdiff(1:length(o.cgb.belief[sender].h), PGBP.scopeindex(sep, o.cgb.belief[sender]))
        verbose && @error "belief $(o.cgb.belief[sender].metadata), integrating $(integ)"   # src/beliefupdates.jl:71
        info && @info "propagation failed: iteration $(res.fail_iter), schedule tree $(res.fail_tree)"
        return (false, false)
    end
    info && res.iter_reached > 0 && @info "calibration reached: iteration $(res.iter_reached), schedule tree $(res.tree_reached)"
    return (true, res.iscal != 0)
end

"integratebelief!(obj, beliefindex) -> (mu, norm)   (src/clustergraphbeliefs.jl:194)"
function PGBP.integratebelief!(o::DeviceClusterGraphBelief, j::Integer)
    m = length(o.cgb.belief[j].h); mu = zeros(max(m, 1)); norm = Ref(0.0); info = Ref(Int32(0))
    check(o.handle, @ccall LIB.pgbp_integrate(o.handle::Ptr{Cvoid}, (j-1)::Int32, mu::Ptr{Float64}, norm::Ref{Float64}, info::Ref{Int32})::Cint)
    info[] == 0 || throw(PGBP.LA.PosDefException(info[]))
    o.cgb.belief[j].μ[:] = mu[1:m]
    return (mu[1:m], norm[])
end

# ---- factor assignment on the device (include/pgbp.h: pgbp_lg_families / pgbp_lg_params) --------------------------
struct LgFamilies      # pgbp_lg_families
    p::Int32; n_families::Int32; max_parents::Int32; n_rates::Int32; n_rows::Int32
    cluster::Ptr{Int32}; n_parents::Ptr{Int32}; child_pos::Ptr{Int32}; data_row::Ptr{Int32}
    parent_pos::Ptr{Int32}; length::Ptr{Float64}; gamma::Ptr{Float64}; color::Ptr{Int32}; data::Ptr{Float64}
    child_mask::Ptr{UInt64}; parent_mask::Ptr{UInt64}     # C_NULL: complete data
end
struct LgParams        # pgbp_lg_params
    model::Int32; per_site::Int32; R::Ptr{Float64}; alpha::Ptr{Float64}; theta::Ptr{Float64}; mu::Ptr{Float64}
end

"""
    lg_setup!(o, prenodes, tbl, taxa, colorof = e -> 0; nrates = 1)

Static part of `assignfactors!` (src/beliefs.jl:786-861) for complete tip data: one entry per node family in the order
of its loop over `node2cluster`; `colorof(edge)` is the 0-based index of the parent edge's variance rate
(`HeterogeneousBrownianMotion`: its colour; homogeneous models: 0).  Call once per cluster graph.
"""
function lg_setup!(o::DeviceClusterGraphBelief, prenodes, tbl, taxa, colorof = e -> 0; nrates::Integer = 1)
    cgb = o.cgb; b = cgb.belief; p = length(tbl)
    n2c, n2fam, n2fix = cgb.node2cluster, cgb.node2family, cgb.node2fixed
    K = max(1, maximum(length(nf) - 1 for nf in n2fam))
    cl = Int32[]; np = Int32[]; cpos = Int32[]; drow = Int32[]
    ppos = Int32[]; len = Float64[]; gam = Float64[]; col = Int32[]
    blockstart(be, lab) = (ind = PGBP.scopeindex([lab], be); isempty(ind) ? Int32(-1) : Int32(ind[1] - 1))
    for (ni, ci) in enumerate(n2c)
        nf = n2fam[ni]; length(nf) == 1 && continue      # root prior: append the family with its own colour if proper
        be = b[ci]; ch = prenodes[ni]
        push!(cl, ci - 1); push!(np, length(nf) - 1)
        push!(cpos, n2fix[ni] ? Int32(-1) : blockstart(be, nf[1]))
        push!(drow, n2fix[ni] ? Int32(findfirst(isequal(ch.name), taxa) - 1) : Int32(-1))
        for k in 1:K
            if k < length(nf)
                pa = prenodes[nf[k+1]]
                e = first(e for e in pa.edge if PGBP.getchild(e) === ch)            # src/beliefs.jl:813-820
                push!(ppos, n2fix[nf[k+1]] ? Int32(-1) : blockstart(be, nf[k+1]))
                push!(len, e.length); push!(gam, length(nf) > 2 ? e.gamma : 1.0); push!(col, colorof(e))
            else
                push!(ppos, -1); push!(len, 1.0); push!(gam, 1.0); push!(col, 0)
            end
        end
    end
    data = Float64[tbl[v][r] for v in 1:p, r in 1:length(taxa)]                   # [row][trait], row-major for C
    o.keep = (cl, np, cpos, drow, ppos, len, gam, col, data)                       # keep alive
    f = LgFamilies(p, length(cl), K, nrates, length(taxa), pointer(cl), pointer(np), pointer(cpos), pointer(drow),
                   pointer(ppos), pointer(len), pointer(gam), pointer(col), pointer(data), C_NULL, C_NULL)
    # missing tip values: fill child_mask / parent_mask from `inscope(be)` and the data's `missing`s, as factors.py does
    GC.@preserve cl np cpos drow ppos len gam col data check(o.handle,
        @ccall LIB.pgbp_lg_setup(o.handle::Ptr{Cvoid}, Ref(f)::Ref{LgFamilies})::Cint)
end

"""
    loglik!(o, rates, mu; alpha = nothing, theta = nothing)

The body of `score(θ)` (src/calibration.jl:195-221) on the device: factors from the model parameters, postorder of
schedule tree 1, `integratebelief!` at its root cluster.  `rates`: vector of p×p variance matrices
(OU: the stationary variance).  Only the parameters and the result cross the bus.
"""
function loglik!(o::DeviceClusterGraphBelief, rates::Vector{<:AbstractMatrix}, mu::AbstractVector;
                 alpha = nothing, theta = nothing)
    R = Float64.(reduce(vcat, vec.(rates))); m = Float64.(mu)
    ou = alpha !== nothing
    a = ou ? Float64[alpha] : Float64[0]; th = ou ? Float64.(theta) : zeros(length(m))
    prm = LgParams(ou ? 1 : 0, 0, pointer(R), ou ? pointer(a) : C_NULL, ou ? pointer(th) : C_NULL, pointer(m))
    norm = Ref(0.0); info = Ref(Int32(0))
    GC.@preserve R m a th begin
        check(o.handle, @ccall LIB.pgbp_lg_assignfactors(o.handle::Ptr{Cvoid}, Ref(prm)::Ref{LgParams})::Cint)
        check(o.handle, @ccall LIB.pgbp_enqueue_loglik_lg(o.handle::Ptr{Cvoid}, 1::Int32, Ref(Opts(0, 1, 0, 0, 1e-5))::Ref{Opts})::Cint)
        check(o.handle, @ccall LIB.pgbp_fetch_loglik(o.handle::Ptr{Cvoid}, norm::Ref{Float64}, info::Ref{Int32})::Cint)
    end
    info[] == 0 || throw(PGBP.LA.PosDefException(info[]))
    return norm[]
end

end # module
