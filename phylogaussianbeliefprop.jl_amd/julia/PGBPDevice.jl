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
    stale::BitVector                   # lazy write-back: beliefs whose Julia arrays are older than the device's (fetch!)
    stale_residuals::Bool              # ... and the same for the message residuals as a whole (fetch_residual!)
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
    obj = DeviceClusterGraphBelief(cgb, h[], zeros(offsets[end]), offsets, nothing, nothing, falses(nb), false)
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

"""
LAZY write-back.  `calibrate!(o, ...; pull = :lazy)` moves only the result struct across the bus and marks every belief of
`o.cgb` stale; `o[j]` (= `fetch!(o, j)`) brings belief j's record over (pgbp_get_belief, a few KB) into the SAME Julia arrays
`o.cgb.belief[j].{J,h,g}` and returns that belief, so `calibrate!` followed by `integratebelief!(o, rootj)` or by reading a
handful of beliefs costs kilobytes instead of the whole state (0.76 GB at 50 000 tips x 16 traits).  `pull = :eager` (the
default, the reference's semantics for callers that read `o.cgb.belief[j]` directly) downloads everything as before.
"""
function fetch!(o::DeviceClusterGraphBelief, j::Integer)
    x = o.cgb.belief[j]
    o.stale[j] || return x
    m = length(x.h); rec = zeros(m*m + m + 1)
    check(o.handle, @ccall LIB.pgbp_get_belief(o.handle::Ptr{Cvoid}, Int32(0)::Int32, Int32(j - 1)::Int32, rec::Ptr{Float64})::Cint)
    copyto!(x.J, 1, rec, 1, m*m); copyto!(x.h, 1, rec, m*m + 1, m); x.g[1] = rec[m*m + m + 1]
    o.stale[j] = false
    return x
end
Base.getindex(o::DeviceClusterGraphBelief, j::Integer) = fetch!(o, j)
"the residual of the message `key = (receiver label, sender label)` (src/clustergraphbeliefs.jl:17-20), fetched alone (pgbp_get_residual)"
function fetch_residual!(o::DeviceClusterGraphBelief, key)
    mr = o.cgb.messageresidual[key]
    o.stale_residuals || return mr
    j = PGBP.sepsetindex(key[1], key[2], o.cgb); k = j - o.cgb.nclusters
    dir = o.cgb.belief[j].metadata == key ? 1 : 2          # message 2(k-1) + dir - 1 is the one received by the sepset's end `dir`
    s = length(mr.Δh); rec = zeros(max(s*s + s, 1)); flag = Ref(Int32(0)); kl = Ref(0.0); klflag = Ref(Int32(0))
    check(o.handle, @ccall LIB.pgbp_get_residual(o.handle::Ptr{Cvoid}, Int32(0)::Int32, Int32(2*(k-1) + dir - 1)::Int32,
        rec::Ptr{Float64}, flag::Ref{Int32}, kl::Ref{Float64}, klflag::Ref{Int32})::Cint)
    copyto!(mr.ΔJ, 1, rec, 1, s*s); copyto!(mr.Δh, 1, rec, s*s + 1, s)
    mr.iscalibrated_resid[1] = flag[] != 0; mr.kldiv[1] = kl[]; mr.iscalibrated_kl[1] = klflag[] != 0
    return mr
end

"device -> the SAME Julia arrays (aliases held by callers stay valid), residuals and flags included"
function pull!(o::DeviceClusterGraphBelief)
    fill!(o.stale, false); o.stale_residuals = false
    check(o.handle, @ccall LIB.pgbp_get_beliefs(o.handle::Ptr{Cvoid}, o.packed::Ptr{Float64})::Cint)
    for (i, x) in enumerate(o.cgb.belief)
        m = length(x.h); p = o.offsets[i]
        copyto!(x.J, 1, o.packed, p + 1, m*m); copyto!(x.h, 1, o.packed, p + m*m + 1, m); x.g[1] = o.packed[p + m*m + m + 1]
    end
    nm = 2 * (length(o.cgb.belief) - o.cgb.nclusters)
    rs = Int(@ccall LIB.pgbp_residual_size(o.handle::Ptr{Cvoid})::Int64)
    res = zeros(max(rs, 1)); flags = zeros(Int32, max(nm, 1)); kl = zeros(max(nm, 1)); klflags = zeros(Int32, max(nm, 1))
    check(o.handle, @ccall LIB.pgbp_get_residuals(o.handle::Ptr{Cvoid}, res::Ptr{Float64}, flags::Ptr{Int32}, kl::Ptr{Float64}, klflags::Ptr{Int32})::Cint)
    p = 0
    for (k, j) in enumerate((o.cgb.nclusters+1):length(o.cgb.belief)), (dir, key) in enumerate((o.cgb.belief[j].metadata, reverse(o.cgb.belief[j].metadata)))
        mr = o.cgb.messageresidual[key]; s = length(mr.Δh)       # key = (receiver, sender)
        copyto!(mr.ΔJ, 1, res, p + 1, s*s); copyto!(mr.Δh, 1, res, p + s*s + 1, s); p += s*s + s
        mr.iscalibrated_resid[1] = flags[2*(k-1) + dir] != 0
        mr.kldiv[1] = kl[2*(k-1) + dir]; mr.iscalibrated_kl[1] = klflags[2*(k-1) + dir] != 0   # src/beliefs.jl:895-924
    end
end

"all beliefs of one site of a batched engine (several sites sharing topology and scopes), packed like `o.packed`"
function site_beliefs(o::DeviceClusterGraphBelief, site::Integer)
    out = zeros(o.offsets[end])
    check(o.handle, @ccall LIB.pgbp_get_site_beliefs(o.handle::Ptr{Cvoid}, Int32(site - 1)::Int32, out::Ptr{Float64})::Cint)
    return out
end

# The exchange buffer of a cluster graph cut across devices (DESIGN.md section 6): the records of the listed beliefs
# (1-based, clusters then sepsets as in `o.belief`) of the first site, back to back, gathered on the device.
function pack_beliefs(o::DeviceClusterGraphBelief, beliefs::Vector{<:Integer})
    idx = Int32.(beliefs .- 1)
    n = @ccall LIB.pgbp_packed_beliefs_size(o.handle::Ptr{Cvoid}, Int32(length(idx))::Int32, idx::Ptr{Int32})::Int64
    n >= 0 || error("belief index out of range")
    buf = zeros(n)
    check(o.handle, @ccall LIB.pgbp_pack_beliefs(o.handle::Ptr{Cvoid}, Int32(0)::Int32, Int32(length(idx))::Int32,
                                                 idx::Ptr{Int32}, buf::Ptr{Float64})::Cint)
    return buf
end
"the reverse: overwrite the listed beliefs of the first site from a buffer another rank packed"
function unpack_beliefs!(o::DeviceClusterGraphBelief, beliefs::Vector{<:Integer}, buf::Vector{Float64})
    idx = Int32.(beliefs .- 1)
    check(o.handle, @ccall LIB.pgbp_unpack_beliefs(o.handle::Ptr{Cvoid}, Int32(0)::Int32, Int32(length(idx))::Int32,
                                                   idx::Ptr{Int32}, buf::Ptr{Float64})::Cint)
end
"propagate_1traversal_postorder! (dir 0) / _preorder! (dir 1) of schedule tree `tree` (1-based): the pieces a cut traversal is made of"
function traverse!(o::DeviceClusterGraphBelief, tree::Integer, dir::Integer; update_residualnorm=true)
    r = Ref(Result(ntuple(_ -> Int32(0), 10)...))
    check(o.handle, @ccall LIB.pgbp_traverse(o.handle::Ptr{Cvoid}, Int32(tree - 1)::Int32, Int32(dir)::Int32,
        Ref(Opts(0, update_residualnorm, 0, 0, 1e-5))::Ref{Opts}, r::Ref{Result})::Cint)
    return r[].succ != 0
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
        update_residualnorm::Bool=true, update_residualkldiv::Bool=false, pull::Symbol=:eager)
    set_schedule!(o, schedule)
    r = Ref(Result(0,0,0,0,0,0,0,0,0,0))
    check(o.handle, @ccall LIB.pgbp_calibrate(o.handle::Ptr{Cvoid}, niter::Int32,
        Ref(Opts(auto, update_residualnorm, update_residualkldiv, 0, 1e-5))::Ref{Opts}, r::Ref{Result})::Cint)
    if pull === :lazy          # nothing but `r` crosses the bus: o[j] / fetch_residual!(o, key) fetch what is read
        fill!(o.stale, true); o.stale_residuals = true
    else
        pull!(o)
    end
    res = r[]
    if res.succ == 0
        report_failure(o, schedule[res.fail_tree], res, verbose)
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

"the reference's error line for a failed message (src/beliefupdates.jl:69-76): belief metadata + integrated indices"
function report_failure(o::DeviceClusterGraphBelief, spt, res::Result, verbose::Bool)
    i = res.fail_edge + 1
    sender = res.fail_dir == 0 ? spt[4][i] : spt[3][i]
    sep = o.cgb.belief[PGBP.sepsetindex(spt[1][i], spt[2][i], o.cgb)]
    integ = setdiff(1:length(o.cgb.belief[sender].h), PGBP.scopeindex(sep, o.cgb.belief[sender]))
    verbose && @error "belief $(o.cgb.belief[sender].metadata), integrating $(integ)"
    return PGBP.BPPosDefException("belief $(o.cgb.belief[sender].metadata), integrating $(integ)", res.fail_info)
end

"""
    propagate_belief!(o, to, sepset, from; update_residualnorm = true) -> nothing | BPPosDefException

`propagate_belief!(cluster_to, sepset, cluster_from, residual)` (src/beliefupdates.jl:634-649) by belief index (1-based,
sepset index >= nclusters + 1): the exception is RETURNED, not thrown, and nothing of the message is applied.
"""
function PGBP.propagate_belief!(o::DeviceClusterGraphBelief, to::Integer, sepset::Integer, from::Integer;
                                update_residualnorm::Bool=true)
    info = Ref(Int32(0))
    check(o.handle, @ccall LIB.pgbp_propagate(o.handle::Ptr{Cvoid}, (to-1)::Int32, (sepset-1)::Int32, (from-1)::Int32,
        Ref(Opts(0, update_residualnorm, 0, 0, 1e-5))::Ref{Opts}, info::Ref{Int32})::Cint)
    pull!(o)
    info[] == 0 && return nothing
    sep = o.cgb.belief[sepset]
    integ = setdiff(1:length(o.cgb.belief[from].h), PGBP.scopeindex(sep, o.cgb.belief[from]))
    return PGBP.BPPosDefException("belief $(o.cgb.belief[from].metadata), integrating $(integ)", info[])
end

function traverse!(o::DeviceClusterGraphBelief, spt, dir::Integer, verbose, up_rn, up_kl)
    set_schedule!(o, [spt])
    r = Ref(Result(0,0,0,0,0,0,0,0,0,0))
    check(o.handle, @ccall LIB.pgbp_traverse(o.handle::Ptr{Cvoid}, 0::Int32, dir::Int32,
        Ref(Opts(0, up_rn, up_kl, 0, 1e-5))::Ref{Opts}, r::Ref{Result})::Cint)
    pull!(o)
    r[].succ != 0 && return true
    report_failure(o, spt, r[], verbose)
    return false
end
"propagate_1traversal_postorder!(beliefs, pa_lab, ch_lab, pa_j, ch_j, verbose, up_rn, up_kl) -> Bool (src/calibration.jl:111-135)"
PGBP.propagate_1traversal_postorder!(o::DeviceClusterGraphBelief, pa_lab, ch_lab, pa_j, ch_j,
        verbose::Bool=true, up_rn::Bool=true, up_kl::Bool=false) =
    traverse!(o, (pa_lab, ch_lab, pa_j, ch_j), 0, verbose, up_rn, up_kl)
"propagate_1traversal_preorder! (src/calibration.jl:137-161)"
PGBP.propagate_1traversal_preorder!(o::DeviceClusterGraphBelief, pa_lab, ch_lab, pa_j, ch_j,
        verbose::Bool=true, up_rn::Bool=true, up_kl::Bool=false) =
    traverse!(o, (pa_lab, ch_lab, pa_j, ch_j), 1, verbose, up_rn, up_kl)

"regularizebeliefs_bycluster!(beliefs, clustergraph) (src/clustergraphbeliefs.jl:235-275) on the device (the graph is the engine's)"
function PGBP.regularizebeliefs_bycluster!(o::DeviceClusterGraphBelief, _clustergraph=nothing)
    check(o.handle, @ccall LIB.pgbp_regularize_bycluster(o.handle::Ptr{Cvoid})::Cint)
    pull!(o)
end

"free_energy(beliefs) -> (average energy, approximate entropy, free energy) (src/score.jl:162-182)"
function PGBP.free_energy(o::DeviceClusterGraphBelief)
    out = zeros(3); info = Ref(Int32(0))
    check(o.handle, @ccall LIB.pgbp_free_energy(o.handle::Ptr{Cvoid}, out::Ptr{Float64}, info::Ref{Int32})::Cint)
    info[] == 0 || throw(PGBP.LA.PosDefException(info[]))
    return (out[1], out[2], out[3])
end
"factored_energy(beliefs) (src/score.jl:151-154)"
PGBP.factored_energy(o::DeviceClusterGraphBelief) = (f = PGBP.free_energy(o); (f[1], f[2], -f[3]))

"""
    residual_kldiv!(o, to, sepset, from) -> iscalibrated_kl

`residual_kldiv!(messageresidual[(to, from)], sepset)` (src/beliefs.jl:1060-1075) for the message last sent from `from`
to `to`; the residual's `kldiv` / `iscalibrated_kl` are updated in the Julia object by the pull.
"""
function PGBP.residual_kldiv!(o::DeviceClusterGraphBelief, to::Integer, sepset::Integer, from::Integer)
    flag = Ref(Int32(0))
    check(o.handle, @ccall LIB.pgbp_residual_kldiv(o.handle::Ptr{Cvoid}, (to-1)::Int32, (sepset-1)::Int32, (from-1)::Int32,
        Ref(Opts(0, 1, 1, 0, 1e-5))::Ref{Opts}, flag::Ref{Int32})::Cint)
    pull!(o)
    return flag[] != 0
end

"init_beliefs_reset_fromfactors! (src/clustergraphbeliefs.jl:126-139) / init_messagecalibrationflags_reset! (:146-150)"
function PGBP.init_beliefs_reset_fromfactors!(o::DeviceClusterGraphBelief)
    check(o.handle, @ccall LIB.pgbp_reset_from_factors(o.handle::Ptr{Cvoid})::Cint); pull!(o)
end
function PGBP.init_messagecalibrationflags_reset!(o::DeviceClusterGraphBelief, reset_kl::Bool=true)
    check(o.handle, @ccall LIB.pgbp_reset_flags(o.handle::Ptr{Cvoid}, reset_kl::Int32)::Cint); pull!(o)
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
    lg_setup!(o, prenodes, tbl, taxa, colorof = e -> 0; nrates = 1, rootpriorcolor = nothing)

Static part of `assignfactors!` (src/beliefs.jl:786-861): one entry per node family that carries a factor, in the order of
its loop over `node2cluster`; `colorof(edge)` is the 0-based index of the parent edge's variance rate
(`HeterogeneousBrownianMotion`: its colour; homogeneous models: 0).  A RANDOM root with a PROPER prior (`factor_root`,
src/evomodels/evomodels.jl:377-396) needs `rootpriorcolor` = the 0-based index, among the `rates` later given to
`loglik!`, of the prior variance (append it to the model's rates and count it in `nrates`); a fixed root or an improper
prior has no factor and takes `nothing` -- asking for a factor the model does not have, or omitting one it has, is an
error here rather than a silently different likelihood.  Missing tip values (`missing` in `tbl`) and partial scopes are
passed as scope masks (bit t = trait t), as `pgbp_amd/factors.py` does.  Call once per cluster graph.
"""
function lg_setup!(o::DeviceClusterGraphBelief, prenodes, tbl, taxa, colorof = e -> 0; nrates::Integer = 1,
                   rootpriorcolor::Union{Nothing,Integer} = nothing)
    cgb = o.cgb; b = cgb.belief; p = length(tbl)
    p <= 64 || error("at most 64 traits")
    n2c, n2fam, n2fix = cgb.node2cluster, cgb.node2family, cgb.node2fixed
    K = max(1, maximum(length(nf) - 1 for nf in n2fam))
    cl = Int32[]; np = Int32[]; cpos = Int32[]; drow = Int32[]
    ppos = Int32[]; len = Float64[]; gam = Float64[]; col = Int32[]
    cmask = UInt64[]; pmask = UInt64[]
    full = p == 64 ? typemax(UInt64) : (UInt64(1) << p) - UInt64(1)
    partial = false
    # (first variable of the node's block or -1, mask of its traits in scope) inside cluster belief `be`
    function place(be, lab)
        j = findfirst(isequal(lab), PGBP.nodelabels(be))
        insc = PGBP.inscope(be)                                   # p x k BitArray (src/beliefs.jl:72-100)
        nd = [count(insc[:, c]) for c in 1:size(insc, 2)]
        any(x -> x != 0 && x != p, nd) && (partial = true)
        m = reduce(|, (UInt64(1) << (t - 1) for t in 1:p if insc[t, j]); init = UInt64(0))
        return (nd[j] == 0 ? Int32(-1) : Int32(sum(nd[1:j-1])), m)
    end
    observed(row) = (ok = [!ismissing(tbl[v][row]) for v in 1:p]; all(ok) || (partial = true);
                     reduce(|, (UInt64(1) << (t - 1) for t in 1:p if ok[t]); init = UInt64(0)))
    for (ni, ci) in enumerate(n2c)
        nf = n2fam[ni]; be = b[ci]; ch = prenodes[ni]
        if length(nf) == 1                                        # the root's own family: its prior, if it has one
            ni == 1 || error("only the root node can belong to a family of size 1")
            if n2fix[1]
                rootpriorcolor === nothing || error("the root is fixed: it has no prior factor")
                continue
            end
            rootpriorcolor === nothing && continue                # improper prior: no factor (evomodels.jl:383-385)
            (pos, m) = place(be, nf[1])
            push!(cl, ci - 1); push!(np, 0); push!(cpos, pos); push!(drow, -1); push!(cmask, m)
            for k in 1:K
                push!(ppos, -1); push!(len, 1.0); push!(gam, 1.0); push!(col, k == 1 ? Int32(rootpriorcolor) : Int32(0))
                push!(pmask, full)
            end
            continue
        end
        push!(cl, ci - 1); push!(np, length(nf) - 1)
        if n2fix[ni]                                              # a tip: its data row is absorbed
            row = findfirst(isequal(ch.name), taxa)
            push!(cpos, -1); push!(drow, Int32(row - 1)); push!(cmask, observed(row))
        else
            (pos, m) = place(be, nf[1])
            push!(cpos, pos); push!(drow, -1); push!(cmask, m)
        end
        for k in 1:K
            if k < length(nf)
                pa = prenodes[nf[k+1]]
                e = first(e for e in pa.edge if PGBP.getchild(e) === ch)            # src/beliefs.jl:813-820
                if n2fix[nf[k+1]]
                    push!(ppos, -1); push!(pmask, full)
                else
                    (pos, m) = place(be, nf[k+1]); push!(ppos, pos); push!(pmask, m)
                end
                push!(len, e.length); push!(gam, length(nf) > 2 ? e.gamma : 1.0); push!(col, colorof(e))
            else
                push!(ppos, -1); push!(len, 1.0); push!(gam, 1.0); push!(col, 0); push!(pmask, full)
            end
        end
    end
    data = Float64[ismissing(tbl[v][r]) ? NaN : tbl[v][r] for v in 1:p, r in 1:length(taxa)]   # [row][trait], row-major for C
    o.keep = (cl, np, cpos, drow, ppos, len, gam, col, data, cmask, pmask)            # keep alive
    f = LgFamilies(p, length(cl), K, nrates, length(taxa), pointer(cl), pointer(np), pointer(cpos), pointer(drow),
                   pointer(ppos), pointer(len), pointer(gam), pointer(col), pointer(data),
                   partial ? pointer(cmask) : C_NULL, partial ? pointer(pmask) : C_NULL)
    GC.@preserve cl np cpos drow ppos len gam col data cmask pmask check(o.handle,
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

# ---- several GPUs (include/pgbp.h "several GPUs") -----------------------------------------------------------------
# One process, several devices: a group of engines over contiguous site ranges; every call fans out inside libpgbp.so.
mutable struct DeviceGroup
    handle::Ptr{Cvoid}
    n_sites::Int
    packed_size::Int
end
checkg(g, rc) = rc == 0 || error(unsafe_string(@ccall LIB.pgbp_group_last_error(g::Ptr{Cvoid})::Cstring))

"`n_sites` independent replicas (same graph and scopes as `o`, different numbers) sharded over `devices` (0-based ordinals)"
function DeviceGroup(dims::Vector{Int32}, sepcl::Vector{Int32}, off::Vector{Int64}, idx::Vector{Int32},
                     nclusters::Integer, n_sites::Integer, devices::Vector{Int32})
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve dims sepcl off idx devices begin
        d = Desc(nclusters, length(dims) - nclusters, pointer(dims), pointer(sepcl), pointer(off), pointer(idx), n_sites, 0)
        rc = @ccall LIB.pgbp_group_create(Ref(d)::Ref{Desc}, length(devices)::Int32, devices::Ptr{Int32}, h::Ref{Ptr{Cvoid}})::Cint
        rc == 0 || error(unsafe_string(@ccall LIB.pgbp_group_last_error(C_NULL::Ptr{Cvoid})::Cstring))
    end
    g = DeviceGroup(h[], n_sites, sum(m*m + m + 1 for m in Int.(dims)))
    finalizer(x -> @ccall(LIB.pgbp_group_destroy(x.handle::Ptr{Cvoid})::Cvoid), g)
    return g
end
"packed: [packed_size, n_sites] (one column per site)"
set_beliefs!(g::DeviceGroup, packed::Matrix{Float64}) =
    checkg(g.handle, @ccall LIB.pgbp_group_set_beliefs(g.handle::Ptr{Cvoid}, packed::Ptr{Float64}, 1::Int32)::Cint)
function set_schedule!(g::DeviceGroup, schedule)
    off = Int32[0]; pa = Int32[]; ch = Int32[]
    for spt in schedule
        append!(pa, Int32.(spt[3] .- 1)); append!(ch, Int32.(spt[4] .- 1)); push!(off, length(pa))
    end
    checkg(g.handle, @ccall LIB.pgbp_group_set_schedule(g.handle::Ptr{Cvoid}, length(schedule)::Int32, off::Ptr{Int32}, pa::Ptr{Int32}, ch::Ptr{Int32})::Cint)
end
"calibrate! on every site -> vector of (succ, iscal), one per site"
function PGBP.calibrate!(g::DeviceGroup, niter::Integer=1; auto::Bool=false, update_residualnorm::Bool=true)
    res = Vector{Result}(undef, g.n_sites)
    checkg(g.handle, @ccall LIB.pgbp_group_calibrate(g.handle::Ptr{Cvoid}, niter::Int32,
        Ref(Opts(auto, update_residualnorm, 0, 0, 1e-5))::Ref{Opts}, res::Ptr{Result})::Cint)
    return [(r.succ != 0, r.iscal != 0) for r in res]
end
"per-site integratebelief! norms (the log-likelihoods on a calibrated clique tree)"
function loglik(g::DeviceGroup, j::Integer)
    norm = zeros(g.n_sites); info = zeros(Int32, g.n_sites)
    checkg(g.handle, @ccall LIB.pgbp_group_integrate(g.handle::Ptr{Cvoid}, (j-1)::Int32, C_NULL::Ptr{Float64}, norm::Ptr{Float64}, info::Ptr{Int32})::Cint)
    any(!=(0), info) && throw(PGBP.LA.PosDefException(first(filter(!=(0), info))))
    return norm
end

# Sites with different missing-data patterns behind one handle: one pgbp_desc per pattern (its own scopes), `sites` = for
# every pattern in turn the (1-based) global indices of its sites.  Pattern k's engine takes everything pattern-specific.
mutable struct DevicePatterns
    handle::Ptr{Cvoid}
    n_sites::Int
end
function DevicePatterns(descs::Vector{Desc}, sites::Vector{<:Integer})
    refs = [Ref(d) for d in descs]
    ptrs = [Base.unsafe_convert(Ptr{Desc}, r) for r in refs]
    h = Ref{Ptr{Cvoid}}(C_NULL)
    s0 = Int32.(sites .- 1)
    rc = GC.@preserve refs @ccall LIB.pgbp_patterns_create(Int32(length(descs))::Int32, ptrs::Ptr{Ptr{Desc}}, s0::Ptr{Int32},
                                                         h::Ref{Ptr{Cvoid}})::Cint
    rc == 0 || error(unsafe_string(@ccall LIB.pgbp_patterns_last_error(C_NULL::Ptr{Cvoid})::Cstring))
    g = DevicePatterns(h[], length(sites))
    finalizer(x -> @ccall(LIB.pgbp_patterns_destroy(x.handle::Ptr{Cvoid})::Cvoid), g)
    return g
end
checkp(h, rc) = rc == 0 || error(unsafe_string(@ccall LIB.pgbp_patterns_last_error(h::Ptr{Cvoid})::Cstring))
pattern_engine(g::DevicePatterns, k::Integer) = @ccall LIB.pgbp_patterns_engine(g.handle::Ptr{Cvoid}, Int32(k - 1)::Int32)::Ptr{Cvoid}
function set_schedule!(g::DevicePatterns, off::Vector{Int32}, pa::Vector{Int32}, ch::Vector{Int32})
    checkp(g.handle, @ccall LIB.pgbp_patterns_set_schedule(g.handle::Ptr{Cvoid}, Int32(length(off) - 1)::Int32, off::Ptr{Int32},
                                                           pa::Ptr{Int32}, ch::Ptr{Int32})::Cint)
end
function PGBP.calibrate!(g::DevicePatterns, niter::Integer=1; auto=false, update_residualnorm=true)
    res = Vector{Result}(undef, g.n_sites)
    checkp(g.handle, @ccall LIB.pgbp_patterns_calibrate(g.handle::Ptr{Cvoid}, Int32(niter)::Int32,
        Ref(Opts(auto, update_residualnorm, 0, 0, 1e-5))::Ref{Opts}, res::Ptr{Result})::Cint)
    return [(r.succ != 0, r.iscal != 0) for r in res]
end
"score() body of every site (device factor fill + postorder + root integrate), in the caller's site order"
function loglik_lg(g::DevicePatterns)
    checkp(g.handle, @ccall LIB.pgbp_patterns_enqueue_loglik_lg(g.handle::Ptr{Cvoid}, Int32(1)::Int32,
        Ref(Opts(0, 1, 0, 0, 1e-5))::Ref{Opts})::Cint)
    norm = zeros(g.n_sites); info = zeros(Int32, g.n_sites)
    checkp(g.handle, @ccall LIB.pgbp_patterns_fetch_loglik(g.handle::Ptr{Cvoid}, norm::Ptr{Float64}, info::Ptr{Int32})::Cint)
    return norm, info
end

# One process per GPU (Distributed.jl / MPI.jl workers): ONE ncclAllGather per gather.
const COMM_ID_BYTES = 128
"rank 0: the id to hand to the other ranks (e.g. `MPI.bcast`, `remotecall_fetch`)"
function comm_unique_id()
    id = zeros(UInt8, COMM_ID_BYTES)
    rc = @ccall LIB.pgbp_comm_unique_id(id::Ptr{UInt8})::Cint
    rc == 0 || error(unsafe_string(@ccall LIB.pgbp_comm_last_error(C_NULL::Ptr{Cvoid})::Cstring))
    return id
end
"0 if this rank can open its communicator (RCCL loadable, the device exists): agree on the minimum over the ranks
BEFORE the collective `comm_create` -- a rank that failed there alone would leave its peers inside ncclCommInitRank"
comm_precheck(device::Integer) = @ccall LIB.pgbp_comm_precheck(device::Int32)::Cint
function comm_create(id::Vector{UInt8}, n_ranks::Integer, rank::Integer, device::Integer)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = @ccall LIB.pgbp_comm_create(id::Ptr{UInt8}, n_ranks::Int32, rank::Int32, device::Int32, h::Ref{Ptr{Cvoid}})::Cint
    rc == 0 || error(unsafe_string(@ccall LIB.pgbp_comm_last_error(C_NULL::Ptr{Cvoid})::Cstring))
    return h[]
end
"every rank's per-site log-likelihoods on every rank + min over all sites of (succ, iscal): one collective"
function comm_gather_loglik(comm::Ptr{Cvoid}, o::DeviceClusterGraphBelief, n_ranks::Integer, slot_sites::Integer)
    norm = zeros(slot_sites, n_ranks); info = zeros(Int32, slot_sites, n_ranks); succ = Ref(Int32(0)); iscal = Ref(Int32(0))
    rc = @ccall LIB.pgbp_comm_gather_loglik(comm::Ptr{Cvoid}, o.handle::Ptr{Cvoid}, slot_sites::Int32, norm::Ptr{Float64},
                                            info::Ptr{Int32}, succ::Ref{Int32}, iscal::Ref{Int32})::Cint
    rc == 0 || error(unsafe_string(@ccall LIB.pgbp_comm_last_error(comm::Ptr{Cvoid})::Cstring))
    return (norm, info, succ[] != 0, iscal[] != 0)
end

end # module
