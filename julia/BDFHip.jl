# BDFHip.jl -- Julia-side binding of libbdf_hip.so (include/bdf.h) for BayesianDataFusion.jl.
#
# UNTESTED IN THIS REPOSITORY: neither the build image nor the GPU box has Julia, and the reference itself is Julia 0.4
# syntax.  This file is written for Julia >= 1.6 and shows the binding a maintainer would add; the same entry points are
# exercised from Python (ctypes) by tests/.  Device buffers are owned through bdf_dev_alloc / bdf_h2d / bdf_d2h.
module BDFHip

const lib = get(ENV, "BDF_HIP_LIB", "libbdf_hip.so")

struct BDFError <: Exception
    code::Cint
    msg::String
end

function check(rc::Cint)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:bdf_last_error, lib), Cstring, ()))
    rc == -1 && throw(ArgumentError(msg))
    rc == -2 && throw(BoundsError(msg))
    throw(BDFError(rc, msg))
end

mutable struct Context
    h::Ptr{Cvoid}
    function Context(device::Integer=0; seed::Integer=0)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:bdf_ctx_create, lib), Cint, (Cint, Ptr{Cvoid}, UInt64, Ref{Ptr{Cvoid}}), device, C_NULL, seed % UInt64, out))
        c = new(out[])
        finalizer(x -> ccall((:bdf_ctx_destroy, lib), Cint, (Ptr{Cvoid},), x.h), c)
        c
    end
    function Context(h::Ptr{Cvoid})                 # a context the library created (bdf_ctx_create_rows / _side)
        c = new(h)
        finalizer(x -> ccall((:bdf_ctx_destroy, lib), Cint, (Ptr{Cvoid},), x.h), c)
        c
    end
end

set_sweep!(c::Context, i) = check(ccall((:bdf_ctx_set_sweep, lib), Cint, (Ptr{Cvoid}, UInt32), c.h, i))
function sync(c::Context)
    check(ccall((:bdf_ctx_sync, lib), Cint, (Ptr{Cvoid},), c.h))
    bits = Ref{UInt32}(0)
    check(ccall((:bdf_ctx_warnings, lib), Cint, (Ptr{Cvoid}, Ref{UInt32}), c.h, bits))
    # BDF_WARN_CG_MAXITER: cg_AtA (parallel_cg.jl:73-93) returns such a column silently
    (bits[] & 0x40) != 0 && @warn "beta update: a conjugate-gradient column was still above its tolerance after maxiter iterations"
    nothing
end
# K1 tuning: rows with more than `item` observations are split into pieces of at most `piece` (defaults 192 / 128)
set_item_size!(c::Context, item) = check(ccall((:bdf_ctx_set_item_size, lib), Cint, (Ptr{Cvoid}, Cint), c.h, item))
"D <= 16: rows of at most `max_obs` observations of an entity with at least `min_rows` rows are sampled four to a wave (0: off)"
set_small_rows!(c::Context, max_obs, min_rows) = check(ccall((:bdf_ctx_set_small_rows, lib), Cint, (Ptr{Cvoid}, Cint, Int64), c.h, max_obs, min_rows))
# rows of few observations by the low-rank sampler (same distribution as sample_user_basic, other values); max_obs = 0: off
set_lowrank!(c::Context, max_obs=-1, min_rows=8192) = check(ccall((:bdf_ctx_set_lowrank, lib), Cint, (Ptr{Cvoid}, Cint, Int64), c.h, max_obs, min_rows))
# 16 < D <= 32, one two-mode relation: the rows four to a wave in the column layout (K1c, k_rows_col.hip), cut into pieces of at most `max_piece`
# observations; 0: off (the wave-per-row kernel), -1: the default again (128, larger for entities of many observations)
set_col_rows!(c::Context, max_piece=-1) = check(ccall((:bdf_ctx_set_col_rows, lib), Cint, (Ptr{Cvoid}, Cint), c.h, max_piece))
# how the latest row launch under `entity_tag` was dispatched: rows by K1-lr, K1s, K1c, K1; K1's items; K1c's waves
function rows_dispatch(c::Context, entity_tag::Integer)
    out = zeros(Int64, 6)
    check(ccall((:bdf_ctx_rows_dispatch, lib), Cint, (Ptr{Cvoid}, UInt32, Ptr{Int64}), c.h, UInt32(entity_tag), out))
    return out
end
set_piece_size!(c::Context, piece) = check(ccall((:bdf_ctx_set_piece_size, lib), Cint, (Ptr{Cvoid}, Cint), c.h, piece))
# a row context on a library-owned stream that leaves `reserve_cus` CUs (0, 8, 16, ...) free, and side contexts that really
# run beside it -- on the reserved CUs (`reserved = true`: the hyperprior's small kernels) or on the others
function rows_context(device::Integer=0; seed::Integer=0, reserve_cus::Integer=8)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:bdf_ctx_create_rows, lib), Cint, (Cint, UInt64, Cint, Ref{Ptr{Cvoid}}), device, seed % UInt64, reserve_cus, out))
    return Context(out[])
end
function side_context(main::Context; apart::Vector{Context}=Context[], reserved::Bool=false)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    hs = Ptr{Cvoid}[a.h for a in apart]
    check(ccall((:bdf_ctx_create_side, lib), Cint, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Cint, Cint, Ref{Ptr{Cvoid}}), main.h, hs, length(hs), reserved, out))
    return Context(out[])
end

"device copy of a Julia array (column-major as is)"
mutable struct DevArray{T}
    ctx::Context
    p::Ptr{Cvoid}
    dims::Tuple
end
function DevArray(c::Context, a::Array{T}) where T
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:bdf_dev_alloc, lib), Cint, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), c.h, sizeof(a), p))
    check(ccall((:bdf_h2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), c.h, p[], a, sizeof(a)))
    d = DevArray{T}(c, p[], size(a))
    finalizer(x -> ccall((:bdf_dev_free, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), x.ctx.h, x.p), d)
    d
end
function Base.Array(d::DevArray{T}) where T
    a = Array{T}(undef, d.dims...)
    check(ccall((:bdf_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), d.ctx.h, a, d.p, sizeof(a)))
    a
end

"Relation.data (IndexedDF / FastIDF) on the device: replaces FastIDF(rel.data) + @spawnat (src/macau.jl:50-52)"
mutable struct DevRelation
    h::Ptr{Cvoid}
    ctx::Context                      # keeps the context alive for as long as the relation (the library frees through it)
    function DevRelation(c::Context, ids::Matrix{Int64}, values::Vector{Float64}, dims::Vector{Int64})
        out = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:bdf_relation_create, lib), Cint,
                    (Ptr{Cvoid}, Cint, Ptr{Int64}, Int64, Ptr{Cvoid}, Cint, Ptr{Float64}, Ref{Ptr{Cvoid}}),
                    c.h, size(ids, 2), dims, size(ids, 1), ids, 8, values, out))
        r = new(out[], c)
        finalizer(x -> ccall((:bdf_relation_destroy, lib), Cint, (Ptr{Cvoid},), x.h), r)
        r
    end
    """
    Several GPUs: the relation of rank `rank` of `world` -- the full IndexedDF index, but on the device only the observations
    of the rows this rank owns, at the internal positions `pos[m]` (0-based, from `layout`) of every mode (bdf_relation_create_sharded).
    """
    function DevRelation(c::Context, ids::Matrix{Int64}, values::Vector{Float64}, dims::Vector{Int64},
                         pos::Vector{Vector{Int32}}, cmax::Vector{Int64}, rank::Integer, world::Integer, chunks::Integer)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        pp = Ptr{Int32}[pointer(p) for p in pos]
        GC.@preserve pos check(ccall((:bdf_relation_create_sharded, lib), Cint,
                    (Ptr{Cvoid}, Cint, Ptr{Int64}, Int64, Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Ptr{Int32}}, Ptr{Int64}, Cint, Cint, Cint, Ref{Ptr{Cvoid}}),
                    c.h, size(ids, 2), dims, size(ids, 1), ids, 8, values, pp, cmax, rank, world, chunks, out))
        r = new(out[], c)
        finalizer(x -> ccall((:bdf_relation_destroy, lib), Cint, (Ptr{Cvoid},), x.h), r)
        r
    end
end

# struct bdf_term of include/bdf.h (BDF_MAX_MODES = 4)
struct Term
    rel::Ptr{Cvoid}
    mode::Int32
    _pad::Int32
    alpha::Float64
    mean_value::Float64
    linear_values::Ptr{Cvoid}
    factors::NTuple{4,Ptr{Cvoid}}
    alpha_dev::Ptr{Cvoid}             # C_NULL, or rel.model.alpha in device memory (sampled there: sample_alpha inside sweep!)
end

"""
    sample_rows!(ctx, D, N, terms, mu, Lambda, entity_tag, out; shard=0, n_shards=1)

Replaces `sample_latent_all2!` (src/sampling.jl:149-172) and `sample_user2_all!` (:251-264): every row of the entity
(or the rows of one shard) is drawn into the device sample matrix `out` (D x N).
"""
function sample_rows!(c::Context, D, N, terms::Vector{Term}, mu::DevArray{Float64}, Lambda::DevArray{Float64}, entity_tag,
                      out::DevArray{Float64}; shard=0, n_shards=1, prior_pack=nothing)
    check(ccall((:bdf_sample_rows, lib), Cint,
                (Ptr{Cvoid}, Cint, Int64, Cint, Ptr{Term}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, UInt32, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}),
                c.h, D, N, length(terms), terms, mu.p, length(mu.dims) == 2 ? 1 : 0, Lambda.p, entity_tag, shard, n_shards, out.p,
                prior_pack === nothing ? C_NULL : prior_pack.p))
end

"ConditionalNormalWishart + rand (src/sampling.jl:116-127, src/normal_wishart.jl:38-42; call site src/macau.jl:120-134)"
function update_prior!(c::Context, D, N, sample, uhat, sumU, UUt, mu0, b0, Tinv, nu, entity_tag, mu, Lambda;
                       prior_pack=nothing, draws=nothing)   # prior_pack: bdf_prior_pack_doubles(D) doubles; draws: bdf_hyper_draws
    check(ccall((:bdf_hyper_sums, lib), Cint, (Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                c.h, D, N, sample.p, uhat === nothing ? C_NULL : uhat.p, sumU.p, UUt.p))
    check(ccall((:bdf_hyper_sample, lib), Cint,
                (Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Ptr{Cvoid}, Float64, UInt32, Ptr{Cvoid}, Ptr{Cvoid},
                 Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                c.h, D, N, sumU.p, UUt.p, mu0.p, b0, Tinv.p, nu, entity_tag, mu.p, Lambda.p, C_NULL,
                prior_pack === nothing ? C_NULL : prior_pack.p, draws === nothing ? C_NULL : draws.p))
end

"update_beta! (src/sampling.jl:361-370): feat is a bdf_feat handle from bdf_feat_create_{dense,csr,bin}"
function update_beta!(c::Context, feat::Ptr{Cvoid}, D, sample, mu, Lambda, lambda_beta_dev, use_ff::Bool, tol, sample_lambda::Bool,
                      nu, mu_h, entity_tag, beta)
    check(ccall((:bdf_sample_beta, lib), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64, Cint, Cint, Float64, Float64,
                 UInt32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                c.h, feat, D, sample.p, mu.p, Lambda.p, lambda_beta_dev.p, use_ff, tol, 0, sample_lambda, nu, mu_h, entity_tag,
                beta.p, C_NULL, C_NULL))
end

# ---- Entity.F operators (S4: F*B, At_mul_B(F,B); RelationData.jl:314-329) ----------------------------------------------
"dense feature matrix (N x numF, column-major as a Julia Matrix{Float64})"
function feat_dense(c::Context, F::Matrix{Float64})
    out = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:bdf_feat_create_dense, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Ref{Ptr{Cvoid}}),
                c.h, size(F, 1), size(F, 2), F, out))
    out[]
end
"sparse features from 1-based COO rows/cols (Int32) and values; `vals === nothing`: binary (SparseBinMatrix / SparseBinMatrixCSR)"
function feat_sparse(c::Context, m, n, rows::Vector{Int32}, cols::Vector{Int32}, vals=nothing)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    if vals === nothing
        check(ccall((:bdf_feat_create_bin, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Int32}, Ptr{Int32}, Ref{Ptr{Cvoid}}),
                    c.h, m, n, length(rows), rows, cols, out))
    else
        check(ccall((:bdf_feat_create_csr, lib), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ref{Ptr{Cvoid}}),
                    c.h, m, n, length(rows), rows, cols, convert(Vector{Float64}, vals), out))
    end
    out[]
end
feat_destroy(f::Ptr{Cvoid}) = check(ccall((:bdf_feat_destroy, lib), Cint, (Ptr{Cvoid},), f))
"out = F * B (transpose = false) or F' * B; B, out: device, column-major with `ncol` columns"
feat_mul!(c::Context, f::Ptr{Cvoid}, B::DevArray, ncol, out::DevArray; transpose=false) =
    check(ccall((:bdf_feat_mul, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint), c.h, f, B.p, ncol, out.p, transpose))
"uhat = (F beta)' and mu .+ uhat (F_mul_beta, RelationData.jl:314-320; macau.jl:103-104)"
uhat!(c::Context, f::Ptr{Cvoid}, D, beta::DevArray, mu::DevArray, uhat::DevArray, mu_matrix::DevArray) =
    check(ccall((:bdf_uhat, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                c.h, f, D, beta.p, mu.p, uhat.p, mu_matrix.p))
"T^-1 += beta' beta * lambda_beta (macau.jl:124-129)"
hyper_feature_terms!(c::Context, D, numF, beta::DevArray, WI::DevArray, lambda_beta::DevArray, Tinv::DevArray) =
    check(ccall((:bdf_hyper_feature_terms, lib), Cint, (Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                c.h, D, numF, beta.p, WI.p, lambda_beta.p, Tinv.p))

# ---- test-set prediction (pred(rel, test_vec, F) + running mean; src/sampling.jl:9-45, src/macau.jl:142-203) ----------
mutable struct DevPairs
    h::Ptr{Cvoid}
    n::Int
    function DevPairs(c::Context, ids::Matrix{Int64}, values::Vector{Float64})
        out = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:bdf_pairs_create, lib), Cint, (Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Cint, Ptr{Float64}, Ref{Ptr{Cvoid}}),
                    c.h, size(ids, 2), size(ids, 1), ids, 8, values, out))
        p = new(out[], size(ids, 1))
        finalizer(x -> ccall((:bdf_pairs_destroy, lib), Cint, (Ptr{Cvoid},), x.h), p)
        p
    end
end
"running posterior mean / sum of squares / clamped errors / class hits; stats: DevArray of 4 doubles"
function predict_update!(c::Context, p::DevPairs, D, factors::Vector{<:DevArray}, mean_value, phase, clamp_lo, clamp_hi, class_cut, stats::DevArray)
    fp = Ptr{Cvoid}[f.p for f in factors]
    check(ccall((:bdf_predict_update, lib), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Ptr{Cvoid}}, Float64, Cint, Float64, Float64, Float64, Ptr{Cvoid}),
                c.h, p.h, D, fp, mean_value, phase, clamp_lo, clamp_hi, class_cut, stats.p))
end

# ---- relation model (src/macau.jl:83-92) ------------------------------------------------------------------------------
"alpha = sample_alpha(alpha_lambda0, alpha_nu0, err) (src/sampling.jl:129-134); stats from predict_sse!, alpha_out: 1 double"
function sample_alpha!(c::Context, p::DevPairs, D, factors::Vector{<:DevArray}, mean_value, lambda0, nu0, rel_tag, stats::DevArray, alpha_out::DevArray)
    fp = Ptr{Cvoid}[f.p for f in factors]
    check(ccall((:bdf_predict_sse, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Ptr{Cvoid}}, Float64, Ptr{Cvoid}, Ptr{Cvoid}),
                c.h, p.h, D, fp, mean_value, C_NULL, stats.p))
    check(ccall((:bdf_sample_alpha, lib), Cint, (Ptr{Cvoid}, Float64, Float64, Int64, Ptr{Cvoid}, UInt32, Ptr{Cvoid}),
                c.h, lambda0, nu0, p.n, stats.p + 8, rel_tag, alpha_out.p))
end
"beta = sample_beta_rel(r); linear_values = mean_value + F beta (src/sampling.jl:322-337, src/macau.jl:89-92)"
function sample_beta_rel!(c::Context, f::Ptr{Cvoid}, train::DevPairs, D, factors::Vector{<:DevArray}, mean_value, alpha, lambda_beta,
                          rel_tag, beta::DevArray, linear_values::DevArray)
    fp = Ptr{Cvoid}[x.p for x in factors]
    check(ccall((:bdf_sample_beta_rel, lib), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Ptr{Cvoid}}, Float64, Float64, Float64, UInt32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                c.h, f, train.h, D, fp, mean_value, alpha, lambda_beta, rel_tag, beta.p, linear_values.p, C_NULL))
end

# ---- the whole iteration (src/macau.jl:80-203 without side information) and the multi-GPU exchange ------------------------
struct GibbsTerm
    rel::Ptr{Cvoid}
    mode::Int32
    entity_of_mode::NTuple{4,Int32}
    alpha::Float64
    mean_value::Float64
end

struct GibbsEntity                                # bdf_gibbs_entity, field for field
    N::Int64
    n_real::Int64
    tag::UInt32
    n_terms::Int32
    terms::NTuple{4,GibbsTerm}
    sample::NTuple{3,Ptr{Cvoid}}
    mu::Ptr{Cvoid}; Lambda::Ptr{Cvoid}; mu0::Ptr{Cvoid}; WI::Ptr{Cvoid}; sumU::Ptr{Cvoid}; UUt::Ptr{Cvoid}
    params::Ptr{Cvoid}; prior_pack::Ptr{Cvoid}; draws::Ptr{Cvoid}
    b0::Float64
    nu0::Float64
    # side information of the entity (Entity.F): C_NULL / zeros = none.  With it the iteration runs F_mul_beta and the per-row
    # prior means before the rows (macau.jl:103-104), the feature terms of the hyperprior (:124-129) and update_beta! (:138-140)
    feat::Ptr{Cvoid}
    beta::Ptr{Cvoid}; uhat::Ptr{Cvoid}; mu_matrix::Ptr{Cvoid}; Tinv::Ptr{Cvoid}; lambda_beta::Ptr{Cvoid}; cg_iters::Ptr{Cvoid}
    use_ff::Int32; sample_lambda_beta::Int32; full_lambda_u::Int32; _pad::Int32
    tol::Float64; lb_nu::Float64; lb_mu::Float64
end

mutable struct Gibbs
    h::Ptr{Cvoid}
    # what the native object dereferences for as long as it lives: its row context (bdf_gibbs_sweep / _destroy synchronise
    # its stream) and, once set, the communicator -- held here so that the GC cannot finalize them first
    rows::Context
    comm::Any
    keep::Vector{Any}                 # device arrays and relations the entity descriptions point into
    function Gibbs(rows::Context, num_latent::Integer, entities::Vector{GibbsEntity}; keep::Vector=Any[])
        out = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:bdf_gibbs_create, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{GibbsEntity}, Ref{Ptr{Cvoid}}),
                    rows.h, num_latent, length(entities), entities, out))
        g = new(out[], rows, nothing, collect(Any, keep))
        finalizer(x -> ccall((:bdf_gibbs_destroy, lib), Cint, (Ptr{Cvoid},), x.h), g)
        g
    end
end
"set-up: full iterations for about `ms` milliseconds whose results are discarded -- the chain's state is put back bit for bit, only the
buffers have rotated (ask `current_buffer`) -- to bring the device to its working state (bdf_gibbs_warm_device)"
warm_device!(g::Gibbs, ms::Real) = check(ccall((:bdf_gibbs_warm_device, lib), Cint, (Ptr{Cvoid}, Float64), g.h, ms))

"one Gibbs iteration; phase: 0 burn-in, 1 first posterior sample, 2 later ones, -1 no prediction update"
sweep!(g::Gibbs, i::Integer, phase::Integer=-1) = check(ccall((:bdf_gibbs_sweep, lib), Cint, (Ptr{Cvoid}, UInt32, Cint), g.h, i, phase))
sync(g::Gibbs) = check(ccall((:bdf_gibbs_sync, lib), Cint, (Ptr{Cvoid},), g.h))
function current_buffer(g::Gibbs, entity::Integer)
    b = Ref{Cint}(0)
    check(ccall((:bdf_gibbs_current, lib), Cint, (Ptr{Cvoid}, Cint, Ref{Cint}), g.h, entity - 1, b))
    return b[] + 1
end
set_test!(g::Gibbs, pairs::Ptr{Cvoid}, entity_of_mode::Vector{Int32}, mean_value, clamp_lo, clamp_hi, class_cut, stats::DevArray) =
    check(ccall((:bdf_gibbs_set_test, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int32}, Float64, Float64, Float64, Float64, Ptr{Cvoid}),
                g.h, pairs, entity_of_mode, mean_value, clamp_lo, clamp_hi, class_cut, stats.p))

"internal row positions of an entity shared by `world` GPUs (bdf_layout_build): (pos::Vector{Int32} 0-based, cmax)"
function layout(degree::Vector{Int64}, world::Integer, chunks::Integer)
    pos = zeros(Int32, length(degree)); cmax = Ref{Int64}(0)
    check(ccall((:bdf_layout_build, lib), Cint, (Int64, Ptr{Int64}, Cint, Cint, Ptr{Int32}, Ref{Int64}),
                length(degree), degree, world, chunks, pos, cmax))
    return pos, cmax[]
end

"rank 0: the 128-byte RCCL id to hand to the other workers (e.g. with remotecall_fetch)"
function comm_unique_id()
    id = zeros(UInt8, 128)
    check(ccall((:bdf_comm_unique_id, lib), Cint, (Ptr{UInt8},), id))
    return id
end

mutable struct Comm
    h::Ptr{Cvoid}
    ctx::Context                      # bdf_comm_destroy uses the context's device
    function Comm(c::Context, rank::Integer, world::Integer, id::Vector{UInt8})
        out = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:bdf_comm_create, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt8}, Ref{Ptr{Cvoid}}), c.h, rank, world, id, out))
        m = new(out[], c)
        finalizer(x -> ccall((:bdf_comm_destroy, lib), Cint, (Ptr{Cvoid},), x.h), m)
        m
    end
end
function set_comm!(g::Gibbs, m::Comm)
    check(ccall((:bdf_gibbs_set_comm, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), g.h, m.h))
    g.comm = m                        # the native object keeps the pointer: keep the communicator alive with it
    nothing
end
"""large exchanges by direct all-pairs copies over IPC mappings (bdf_comm_enable_peer).  `fn`: a `@cfunction` pointer to the host's
all-gather, `(user::Ptr{Cvoid}, send::Ptr{Cvoid}, recv::Ptr{Cvoid}, bytes::Csize_t) -> Cint` (e.g. a remotecall round over the
workers): it carries the control messages and orders the copies"""
enable_peer!(m::Comm, fn::Ptr{Cvoid}, user::Ptr{Cvoid}=C_NULL, min_bytes::Integer=4 << 20) =
    check(ccall((:bdf_comm_enable_peer, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), m.h, fn, user, min_bytes))
disable_peer!(m::Comm) = check(ccall((:bdf_comm_disable_peer, lib), Cint, (Ptr{Cvoid},), m.h))
"(collective) one exchange by peer copies whatever its size, complete on return: `buf` (device) holds world blocks of `bytes`, this rank's filled in"
peer_selftest!(c::Context, m::Comm, buf::Ptr{Cvoid}, bytes::Integer) =
    check(ccall((:bdf_comm_peer_selftest, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), c.h, m.h, buf, bytes))
function peer_stats(m::Comm)
    n = Ref{Int64}(0); b = Ref{Int64}(0)
    check(ccall((:bdf_comm_peer_stats, lib), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), m.h, n, b))
    return n[], b[]
end
"in-place exchange of chunk `chunk` (0-based) of the D x N sample matrix between the ranks, then `allgather_join!` (bdf_allgather_rows / _join)"
allgather_rows!(c::Context, m::Comm, D, N, sample::DevArray, chunk::Integer, chunks::Integer) =
    check(ccall((:bdf_allgather_rows, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Cint, Cint), c.h, m.h, D, N, sample.p, chunk, chunks))
allgather_join!(c::Context, m::Comm) = check(ccall((:bdf_allgather_join, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), c.h, m.h))
"update_beta! on several ranks: the conjugate-gradient columns shared out and all-gathered (parallel_matrix.jl:488-507; bdf_sample_beta_ranks)"
function update_beta!(c::Context, m::Comm, feat::Ptr{Cvoid}, D, sample, mu, Lambda, lambda_beta_dev, use_ff::Bool, tol, sample_lambda::Bool,
                      nu, mu_h, entity_tag, beta)
    check(ccall((:bdf_sample_beta_ranks, lib), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64, Cint, Cint, Float64, Float64,
                 UInt32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                c.h, m.h, feat, D, sample.p, mu.p, Lambda.p, lambda_beta_dev.p, use_ff, tol, 0, sample_lambda, nu, mu_h, entity_tag,
                beta.p, C_NULL, C_NULL))
end

"the relation model on several ranks (rank r holds the block of observations first_obs+1 : first_obs+train.n -- f: those rows of the
relation's feature matrix, train: the same observations as pairs): the squared-error sum is added up over the ranks in rank order
(bdf_sum_ranks) before sample_alpha; n_total: the relation's observations"
function sample_alpha!(c::Context, m::Comm, p::DevPairs, n_total::Integer, D, factors::Vector{<:DevArray}, mean_value, lambda0, nu0, rel_tag,
                       stats::DevArray, alpha_out::DevArray)
    fp = Ptr{Cvoid}[f.p for f in factors]
    check(ccall((:bdf_predict_sse, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Ptr{Cvoid}}, Float64, Ptr{Cvoid}, Ptr{Cvoid}),
                c.h, p.h, D, fp, mean_value, C_NULL, stats.p))
    check(ccall((:bdf_sum_ranks, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64), c.h, m.h, stats.p + 8, 1))
    check(ccall((:bdf_sample_alpha, lib), Cint, (Ptr{Cvoid}, Float64, Float64, Int64, Ptr{Cvoid}, UInt32, Ptr{Cvoid}),
                c.h, lambda0, nu0, n_total, stats.p + 8, rel_tag, alpha_out.p))
end
"sample_beta_rel on several ranks (bdf_sample_beta_rel_ranks): F'v and, once, F'F summed over the ranks, the same beta on every rank;
linear_values: world blocks of `block` values, this rank's block written, then gathered in place (bdf_allgather_block)"
function sample_beta_rel!(c::Context, m::Comm, f::Ptr{Cvoid}, train::DevPairs, first_obs::Integer, block::Integer, D, factors::Vector{<:DevArray},
                          mean_value, alpha, lambda_beta, rel_tag, beta::DevArray, linear_values::DevArray)
    fp = Ptr{Cvoid}[x.p for x in factors]
    check(ccall((:bdf_sample_beta_rel_ranks, lib), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cint, Ptr{Ptr{Cvoid}}, Float64, Float64, Float64, UInt32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                c.h, m.h, f, train.h, first_obs, D, fp, mean_value, alpha, lambda_beta, rel_tag, beta.p, linear_values.p + 8 * first_obs, C_NULL))
    check(ccall((:bdf_allgather_block, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), c.h, m.h, linear_values.p, 8 * block))
    allgather_join!(c, m)
end

# ---- IndexedDF index, launch order, value mean (a1; src/IndexedDF.jl:10-21, 26, 41-43) ----------------------------------------
"IndexedDF.index on the host, without a device: per mode (rowptr[dims[m]+1] 0-based offsets, rowids[nnz] 1-based COO row numbers)"
function index_build(ids::Matrix{Int64}, dims::Vector{Int64})
    nnz, nm = size(ids)
    rp = [zeros(Int64, d + 1) for d in dims]; ri = [zeros(Int64, max(nnz, 1)) for _ in dims]
    rpp = Ptr{Int64}[pointer(x) for x in rp]; rip = Ptr{Int64}[pointer(x) for x in ri]
    GC.@preserve rp ri check(ccall((:bdf_index_build, lib), Cint, (Cint, Ptr{Int64}, Int64, Ptr{Cvoid}, Cint, Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}),
                                   nm, dims, nnz, ids, 8, rpp, rip))
    return [(rp[m], ri[m][1:nnz]) for m in 1:nm]
end
"the device relation's own index of `mode` (1-based mode): getData / getCount / getI (IndexedDF.jl:41-43, 67-70)"
function relation_index(r::DevRelation, mode::Integer, dim::Integer, nnz::Integer)
    rp = Ref{Ptr{Int64}}(C_NULL); ri = Ref{Ptr{Int64}}(C_NULL)
    check(ccall((:bdf_relation_index, lib), Cint, (Ptr{Cvoid}, Cint, Ref{Ptr{Int64}}, Ref{Ptr{Int64}}), r.h, mode - 1, rp, ri))
    return copy(unsafe_wrap(Array, rp[], dim + 1)), copy(unsafe_wrap(Array, ri[], nnz))
end
function relation_value_mean(r::DevRelation)            # valueMean (IndexedDF.jl:26)
    m = Ref{Float64}(0.0)
    check(ccall((:bdf_relation_value_mean, lib), Cint, (Ptr{Cvoid}, Ref{Float64}), r.h, m))
    return m[]
end
"rows of `mode` (1-based) by falling number of observations: the order the row kernel deals its shards from (sampling.jl:154)"
function relation_order(r::DevRelation, mode::Integer, dim::Integer)
    o = zeros(Int32, dim)
    check(ccall((:bdf_relation_order, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Int32}), r.h, mode - 1, o))
    return o .+ Int32(1)
end

# ---- Block / sample_users_blocked (src/sampling.jl:236-249) ---------------------------------------------------------------
"the users of a Block share one covariance: vx (device Int32, 0-based item ids), Yma (device nv x nu), factor (device D x M) -> out (device D x nu)"
sample_block!(c::Context, D, nu, nv, vx::DevArray{Int32}, Yma::DevArray{Float64}, factor::DevArray{Float64}, alpha, mu::DevArray{Float64},
              Lambda::DevArray{Float64}, entity_tag, out::DevArray{Float64}) =
    check(ccall((:bdf_sample_block, lib), Cint,
                (Ptr{Cvoid}, Cint, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Ptr{Cvoid}),
                c.h, D, nu, nv, vx.p, Yma.p, factor.p, alpha, mu.p, Lambda.p, entity_tag, out.p))

# ---- prediction (src/sampling.jl:9-45) --------------------------------------------------------------------------------------
"pred(r, probe_vec): udot over the pairs + mean_value (or the pairs' baseline, set_baseline!) -> out (device, one double per pair)"
function predict!(c::Context, p::DevPairs, D, factors::Vector{<:DevArray}, mean_value, out::DevArray{Float64})
    fp = Ptr{Cvoid}[f.p for f in factors]
    check(ccall((:bdf_predict, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Ptr{Cvoid}}, Float64, Ptr{Cvoid}), c.h, p.h, D, fp, mean_value, out.p))
end
"pred_all(r) (sampling.jl:91-97): every cell of the relation, the last mode fastest -> out (device, prod(dims) doubles)"
function predict_all!(c::Context, dims::Vector{Int64}, D, factors::Vector{<:DevArray}, mean_value, out::DevArray{Float64})
    fp = Ptr{Cvoid}[f.p for f in factors]
    check(ccall((:bdf_predict_all, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Int64}, Cint, Ptr{Ptr{Cvoid}}, Float64, Ptr{Cvoid}),
                c.h, length(dims), dims, D, fp, mean_value, out.p))
end
"per-pair baseline replacing mean_value: mean_value + F_test beta of pred(r, probe_vec, F) (sampling.jl:9-14); `nothing` clears it"
set_baseline!(p::DevPairs, baseline) =
    check(ccall((:bdf_pairs_set_baseline, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), p.h, baseline === nothing ? C_NULL : baseline.p))
"out = mean_value + F beta: linear_values (macau.jl:91) and the test rows' baseline"
feat_linear!(c::Context, f::Ptr{Cvoid}, beta::DevArray{Float64}, mean_value, out::DevArray{Float64}) =
    check(ccall((:bdf_feat_linear, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Ptr{Cvoid}), c.h, f, beta.p, mean_value, out.p))
"store the pairs sorted by their id in `mode` (1-based): halves the gather traffic of the updates; results stay in the caller's order"
pairs_sort!(p::DevPairs, mode::Integer) = check(ccall((:bdf_pairs_sort, lib), Cint, (Ptr{Cvoid}, Cint), p.h, mode - 1))
"storage position -> the caller's index (1-based) after pairs_sort!"
function pairs_order(p::DevPairs)
    o = zeros(Int64, p.n)
    check(ccall((:bdf_pairs_order, lib), Cint, (Ptr{Cvoid}, Ptr{Int64}), p.h, o))
    return o .+ 1
end
"running posterior mean and sum of squares of the pairs (macau.jl:171-183, 235-241), in STORAGE order (pairs_order)"
function pairs_state(c::Context, p::DevPairs)
    a = Ref{Ptr{Cvoid}}(C_NULL); q = Ref{Ptr{Cvoid}}(C_NULL); n = Ref{Int64}(0)
    check(ccall((:bdf_pairs_state, lib), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ref{Ptr{Cvoid}}, Ref{Int64}), p.h, a, q, n))
    avg = zeros(n[]); sq = zeros(n[])
    if n[] > 0
        check(ccall((:bdf_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), c.h, avg, a[], 8 * n[]))
        check(ccall((:bdf_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), c.h, sq, q[], 8 * n[]))
    end
    return avg, sq
end

# ---- feature operators on several ranks; the hyperprior's sums over the ranks ----------------------------------------------
"original id (0-based) of every row of F: rows moved to an entity's internal positions keep their noise streams (several GPUs)"
feat_set_row_ids!(f::Ptr{Cvoid}, row_ids::Vector{Int32}) = check(ccall((:bdf_feat_set_row_ids, lib), Cint, (Ptr{Cvoid}, Ptr{Int32}), f, row_ids))
function feat_size(f::Ptr{Cvoid})
    m = Ref{Int64}(0); n = Ref{Int64}(0); z = Ref{Int64}(0)
    check(ccall((:bdf_feat_size, lib), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Int64}), f, m, n, z))
    return m[], n[], z[]
end
"sum_i U_i and U U' over the rows this rank owns, the ranks' partial sums added in rank order (src/sampling.jl:117-119 on the master)"
hyper_sums!(c::Context, m::Comm, D, N, chunks, sample::DevArray, uhat, sumU::DevArray, UUt::DevArray) =
    check(ccall((:bdf_hyper_sums_ranks, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                c.h, m.h, D, N, chunks, sample.p, uhat === nothing ? C_NULL : uhat.p, sumU.p, UUt.p))
"the data-independent part of the hyperprior draw (Bartlett matrix, mean normals), ahead of the rows: D*D + D doubles"
hyper_draws!(c::Context, D, N, nu, entity_tag, draws::DevArray{Float64}) =
    check(ccall((:bdf_hyper_draws, lib), Cint, (Ptr{Cvoid}, Cint, Int64, Float64, UInt32, Ptr{Cvoid}), c.h, D, N, nu, entity_tag, draws.p))
prior_pack_doubles(D::Integer) = Int(ccall((:bdf_prior_pack_doubles, lib), Cint, (Cint,), D))

# ---- the relation model inside the native iteration (src/macau.jl:83-92; bdf_gibbs_set_relations) ----------------------------
struct GibbsRelation                              # bdf_gibbs_relation, field for field
    rel::Ptr{Cvoid}
    entity_of_mode::NTuple{4,Int32}
    mean_value::Float64
    alpha_dev::Ptr{Cvoid}
    alpha_sample::Int32
    rel_tag::UInt32
    alpha_lambda0::Float64
    alpha_nu0::Float64
    nnz::Int64
    train::Ptr{Cvoid}
    first_obs::Int64
    obs_block::Int64
    feat::Ptr{Cvoid}
    beta::Ptr{Cvoid}
    linear::Ptr{Cvoid}
    lambda_beta::Float64
    feat_test::Ptr{Cvoid}
    test_baseline::Ptr{Cvoid}
end
"register the relations whose alpha is sampled and / or that carry features: sweep! then runs sample_alpha, sample_beta_rel and
linear_values before the rows of every iteration; `keep`: what the records point into"
function set_relations!(g::Gibbs, rels::Vector{GibbsRelation}; keep::Vector=Any[])
    check(ccall((:bdf_gibbs_set_relations, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{GibbsRelation}), g.h, length(rels), rels))
    append!(g.keep, keep)
    nothing
end

end # module
