# NNopHIPExt.jl -- package extension that routes NNop's Flash Attention to libnnop_hip.so on AMD GPUs.
#
# SOURCE ONLY: there is no Julia in the build image, so this file is not executed by the test-suite;
# the identical C calls are exercised from Python (nnop.jl_amd/_lib.py + attention.py).  It has the
# shape of the reference's own extension (ext/NNopAMDGPUExt.jl:1-11): it only ADDS METHODS to
# functions owned by NNop, so nothing else in the package changes:
#
#   NNop._shared_memory(::ROCBackend, device_id)             ext/NNopAMDGPUExt.jl:6-9
#   NNop._flash_attention(q,k,v,pair; causal,kpad_mask)       src/attention.jl:133-177   -> (o, ms, ls)
#   NNop.∇flash_attention(Δ,o,ms,ls,q,k,v,pair; causal,...)   src/attention_bwd.jl:199-275 -> (dq,dk,dv,dpair)
#   NNop._llama_rope(q, k, cos, sin; bwd)                     src/rope/llama_rope.jl:69-89 -> (q', k')
#   NNop.online_softmax(x) / NNop.∇online_softmax(Δ, y)       src/softmax.jl:60-80
#   NNop._rms_norm / NNop.∇rms_norm                           src/rms_norm.jl:117-169
#   NNop._layer_norm / NNop.∇layer_norm                       src/layer_norm.jl:150-204
#
# `flash_attention` and the ChainRules `rrule` (src/attention_crc.jl:4-31) call these generically, so
# `NNop.flash_attention(q, k, v; causal)` and `Zygote.gradient` keep working unchanged.
#
# Install (see INTEGRATION.md):  in NNop's Project.toml
#     [extensions]
#     NNopHIPExt = "AMDGPU"          # REPLACES the line `NNopAMDGPUExt = "AMDGPU"`
# and put this file at ext/NNopHIPExt.jl; point ENV["NNOP_HIP_LIB"] at libnnop_hip.so.
# It replaces NNopAMDGPUExt, it cannot sit beside it: both define NNop._shared_memory(::ROCBackend, ::Integer), and
# two extensions defining one method is a method overwrite, which Julia rejects during precompilation.
#
# GC / stream safety of the ccalls: every array whose raw device pointer is passed is held by GC.@preserve for the
# duration of the call; the library only ENQUEUES kernels on the task-local HIP stream and keeps no pointer.  Arrays
# that die right after the call (the backward workspace `ws`) are safe because AMDGPU.jl frees device memory in stream
# order (hipFreeAsync on the same task-local stream), i.e. after the kernels enqueued here.
module NNopHIPExt

using AMDGPU
using NNop

const LIB = Ref{String}("")
libnnop() = isempty(LIB[]) ? (LIB[] = get(ENV, "NNOP_HIP_LIB", "libnnop_hip.so")) : LIB[]

# struct nnop_fa_desc (include/nnop_hip.h)
struct FaDesc
    dtype::Int32; emb::Int32; ql::Int32; kl::Int32; qh::Int32; kh::Int32; batch::Int32; causal::Int32
    emb_k::Int32; emb_v::Int32; kl_v::Int32; kh_v::Int32
end

# nnop_dtype
nnop_dtype(::Type{Float32}) = Int32(0)
nnop_dtype(::Type{Float16}) = Int32(1)
# BFloat16 is a Core type from Julia 1.11 on; on older Julia the extension still loads, without the bf16 methods
@static if isdefined(Core, :BFloat16)
    nnop_dtype(::Type{Core.BFloat16}) = Int32(2)      # BFloat16s.BFloat16 (>= 0.5 on 1.11) is this same type
    const HipFloat = Union{Float32, Float16, Core.BFloat16}
else
    const HipFloat = Union{Float32, Float16}
end

# nnop_status -> the reference's ErrorException messages (src/attention.jl:141-144, :204)
function check(st::Cint, q, k, v)
    st == 0 && return
    QE, KE = size(q, 1), size(k, 1)
    st == -1 && error("Embedding dim of Q `$QE` must be the same as of K `$KE`.")
    st == -2 && error("Shapes of K `$(size(k))` and V `$(size(v))` must be the same.")
    st == -3 && error("Only power-of-2 embedding dims are supported.")
    st == -4 && error("Number of query heads `$(size(q, 3))` must be divisible by number of KV heads `$(size(k, 3))`.")
    st == -7 && error("Failed to find groupsize for Flash Attention that satisfies Shared Memory constraint.")
    st == -11 && error("libnnop_hip: a tensor base address is not 16-byte aligned (views with odd element offsets are not supported: copy them)")
    msg = unsafe_string(ccall((:nnop_strerror, libnnop()), Cstring, (Cint,), st))
    error("libnnop_hip: $msg")
end

desc(q, k, v, causal) = FaDesc(nnop_dtype(eltype(q)), size(q, 1), size(q, 2), size(k, 2), size(q, 3), size(k, 3),
                               size(q, 4), causal ? 1 : 0, size(k, 1), size(v, 1), size(v, 2), size(v, 3))

devptr(x) = Ptr{Cvoid}(UInt(pointer(x)))             # device VA of a ROCArray
devptr(::Nothing) = Ptr{Cvoid}(0)
hipstream() = Ptr{Cvoid}(UInt(AMDGPU.stream().stream))   # the task-local HIP stream the arrays live on

# ext/NNopAMDGPUExt.jl:6-9 (device_id is 1-based in AMDGPU.devices())
function NNop._shared_memory(::ROCBackend, device_id::Integer)
    out = Ref{UInt64}(0)
    st = ccall((:nnop_shared_memory, libnnop()), Cint, (Cint, Ptr{UInt64}), device_id - 1, out)
    st == 0 || error("nnop_shared_memory failed ($st)")
    return out[]
end

# src/attention.jl:133-177
function NNop._flash_attention(
    q::ROCArray{T,4}, k::ROCArray{T,4}, v::ROCArray{T,4},
    pair::Union{Nothing,ROCArray{T,4}} = nothing;
    causal::Bool, kpad_mask::Union{Nothing,ROCMatrix{Bool}} = nothing,
) where T <: HipFloat
    d = Ref(desc(q, k, v, causal))
    o  = similar(q)                                            # :166
    ms = ROCArray{T}(undef, size(q, 2), size(q, 3), size(q, 4))  # :167  (QL, QH, B)
    ls = similar(ms)                                           # :168
    st = GC.@preserve o ms ls q k v pair kpad_mask ccall((:nnop_fa_fwd, libnnop()), Cint,
        (Ptr{FaDesc}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        d, devptr(o), devptr(ms), devptr(ls), devptr(q), devptr(k), devptr(v), devptr(pair), devptr(kpad_mask), hipstream())
    check(st, q, k, v)
    return o, ms, ls                                           # asynchronous, like the KA launch (:170-176)
end

# src/attention_bwd.jl:199-275
function NNop.∇flash_attention(
    Δ::ROCArray{T,4},
    o::ROCArray{T,4}, ms::ROCArray{T,3}, ls::ROCArray{T,3},
    q::ROCArray{T,4}, k::ROCArray{T,4}, v::ROCArray{T,4},
    pair::Union{Nothing,ROCArray{T,4}} = nothing;
    causal::Bool, kpad_mask::Union{Nothing,ROCMatrix{Bool}} = nothing,
) where T <: HipFloat
    d = Ref(desc(q, k, v, causal))
    nbytes = ccall((:nnop_fa_bwd_workspace_bytes, libnnop()), Csize_t, (Ptr{FaDesc},), d)
    if nbytes != 0 && pair !== nothing
        # scratch for the head-major copies of the bias: the backward then runs on 16-byte accesses (include/nnop_hip.h)
        nbytes = max(nbytes, ccall((:nnop_fa_bwd_workspace_bytes_pair, libnnop()), Csize_t, (Ptr{FaDesc},), d))
    end
    if nbytes == 0
        # invalid descriptor (the same four checks as the forward, src/attention_bwd.jl:210-213): a call with NULL
        # tensors returns the status of the failed check before it looks at any pointer
        null = Ptr{Cvoid}(0)
        st = ccall((:nnop_fa_bwd, libnnop()), Cint,
            (Ptr{FaDesc}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
             Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
             Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
             Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
            d, null, null, null, null, null, null, null, null, null, null, null, null, null, null, Csize_t(0), null)
        check(st, q, k, v)
        error("libnnop_hip: invalid attention descriptor")    # unreachable when check() raised
    end
    dq, dk, dv = similar(q), similar(k), similar(v)           # fully overwritten by the library
    dp = isnothing(pair) ? nothing : similar(pair)
    ws = ROCArray{UInt8}(undef, nbytes)                        # replaces Δ_scaled / δ (:224-225)
    st = GC.@preserve dq dk dv dp Δ o ms ls q k v pair kpad_mask ws ccall((:nnop_fa_bwd, libnnop()), Cint,
        (Ptr{FaDesc}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},          # dq dk dv dpair
         Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},                      # Δ o ms ls
         Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},          # q k v pair kpad_mask
         Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),                                    # workspace, bytes, stream
        d, devptr(dq), devptr(dk), devptr(dv), devptr(dp), devptr(Δ), devptr(o), devptr(ms), devptr(ls),
        devptr(q), devptr(k), devptr(v), devptr(pair), devptr(kpad_mask), devptr(ws), nbytes, hipstream())
    check(st, q, k, v)
    return dq, dk, dv, dp
end

# A cotangent that is not a ROCArray -- the lazy `FillArrays.Fill` Zygote produces for `sum(flash_attention(...))`
# (test/attention_tests.jl:36-41), a Broadcasted, a host Array -- would miss the method above and fall through to the
# reference's generic KernelAbstractions backward.  Materialise it on the device and stay on the HIP path.
_to_roc(Δ::ROCArray, like) = Δ
_to_roc(Δ::Array, like) = copyto!(similar(like), Δ)
_to_roc(Δ, like) = (d = similar(like); d .= Δ; d)           # Fill / lazy wrappers broadcast without scalar indexing
function NNop.∇flash_attention(
    Δ::AbstractArray{<:Real,4},
    o::ROCArray{T,4}, ms::ROCArray{T,3}, ls::ROCArray{T,3},
    q::ROCArray{T,4}, k::ROCArray{T,4}, v::ROCArray{T,4},
    pair::Union{Nothing,ROCArray{T,4}} = nothing;
    causal::Bool, kpad_mask::Union{Nothing,ROCMatrix{Bool}} = nothing,
) where T <: HipFloat
    return NNop.∇flash_attention(_to_roc(Δ, o)::ROCArray{T,4}, o, ms, ls, q, k, v, pair; causal, kpad_mask)
end

# ---- several devices (include/nnop_hip.h: nnop_fa_shards, ABI version 6) -------------------------------------------------------------
# The reference has no multi-GPU code; (batch, kv-head) slices are independent in forward and backward (src/attention.jl:27-28,33;
# src/attention_bwd.jl:28-29,34), so the H x B axis shards with pointer offsets and NO data-path collective.  nnop_fa_shards gives
# rank g of `world` its contiguous unit range as <= 3 dense rectangles; each is a problem of its own for _flash_attention at the
# element offsets it reports.  Two ways to use it from Julia:
#   * one process per GPU (MPI.jl / Distributed.jl): call `flash_attention_shard(q, k, v, world, rank; causal)` on the rank's device
#     with the FULL arrays resident there, or with the rank's slices and world = 1;
#   * one process driving all GPUs (below): `flash_attention_multi(qs, ks, vs; causal)` with one (q, k, v) replica or slice per device
#     -- a task per device, each on its own task-local HIP stream (AMDGPU.device! is task-local), joined at the end.
# struct nnop_fa_shard
struct FaShard
    desc::FaDesc
    b0::Int32; b1::Int32; kh0::Int32; kh1::Int32
    q_off::UInt64; kv_off::UInt64; row_off::UInt64; mask_off::UInt64; pair_off::Int64
end

function fa_shards(q, k, v, causal::Bool, world::Integer, rank::Integer)
    out = Vector{FaShard}(undef, 3)
    n = ccall((:nnop_fa_shards, libnnop()), Cint, (Ptr{FaDesc}, Cint, Cint, Ptr{FaShard}), Ref(desc(q, k, v, causal)), world, rank, out)
    n < 0 && check(n, q, k, v)
    return out[1:n]
end

# Rank `rank` (0-based) of `world`: its rectangles of o, ms, ls are written in place into full-size outputs (the rest is left
# untouched: another rank's).  No copies: `view`s of the (E, L, H, B) arrays over head / batch ranges are the rectangles.
function flash_attention_shard!(o, ms, ls, q::ROCArray{T,4}, k::ROCArray{T,4}, v::ROCArray{T,4}, world::Integer, rank::Integer;
                                causal::Bool, kpad_mask::Union{Nothing,ROCMatrix{Bool}} = nothing) where T <: HipFloat
    rep = size(q, 3) ÷ size(k, 3)
    for s in fa_shards(q, k, v, causal, world, rank)
        b, h, qh = (s.b0 + 1):s.b1, (s.kh0 + 1):s.kh1, (s.kh0 * rep + 1):(s.kh1 * rep)
        # the four views are dense (whole batches, or a head range of one batch): the library sees plain [B'][H'][L][E] problems
        oo, mm, ll = NNop._flash_attention(q[:, :, qh, b], k[:, :, h, b], v[:, :, h, b]; causal,
                                           kpad_mask = isnothing(kpad_mask) ? nothing : kpad_mask[:, b])
        o[:, :, qh, b] .= oo; ms[:, qh, b] .= mm; ls[:, qh, b] .= ll
    end
    return o, ms, ls
end
# (The indexing above copies the rectangle -- the simple, allocation-tolerant form.  The zero-copy form passes
#  devptr(q) + s.q_off * sizeof(T), devptr(k) + s.kv_off * sizeof(T), ... with Ref(s.desc) straight to :nnop_fa_fwd / :nnop_fa_bwd,
#  exactly as _flash_attention does with offset 0; the offsets are multiples of L * E elements, so 16-byte alignment is kept.)

# One process, all devices: qs[g], ks[g], vs[g] live on device g (replicas or batch slices of the global problem).
function flash_attention_multi(qs::Vector, ks::Vector, vs::Vector; causal::Bool)
    world = length(qs)
    tasks = map(1:world) do g
        Threads.@spawn begin
            AMDGPU.device!(AMDGPU.devices()[g])                 # task-local: this task's arrays and stream live on device g
            o, ms, ls = similar(qs[g]), ROCArray{eltype(qs[g])}(undef, size(qs[g])[2:4]...), ROCArray{eltype(qs[g])}(undef, size(qs[g])[2:4]...)
            flash_attention_shard!(o, ms, ls, qs[g], ks[g], vs[g], world, g - 1; causal)
            AMDGPU.synchronize()
            (o, ms, ls)
        end
    end
    return fetch.(tasks)
end

# struct nnop_rope_desc (include/nnop_hip.h)
struct RopeDesc
    dtype::Int32; cs_dtype::Int32; dim::Int32; seq::Int32; qh::Int32; kh::Int32; batch::Int32
end

# src/rope/llama_rope.jl:69-89.  `llama_rope`, `∇llama_rope` and the rrule (:67, :91-98) call this generically.
# One pass, out of place: replaces copy(q), copy(k) + the in-place kernel (:75-86).
function NNop._llama_rope(
    q::ROCArray{T,4}, k::ROCArray{T,4}, cos::ROCArray{C,3}, sin::ROCArray{C,3}; bwd::Bool,
) where {T <: HipFloat, C <: HipFloat}
    head_dim, q_seq, q_heads, batch = size(q)
    @assert size(k, 1) == head_dim "Head dimension mismatch"    # :72
    @assert size(k, 2) == q_seq && size(k, 4) == batch          # :73
    @assert size(cos) == size(sin) == (head_dim, q_seq, batch)
    d = Ref(RopeDesc(nnop_dtype(T), nnop_dtype(C), head_dim, q_seq, q_heads, size(k, 3), batch))
    qo, ko = similar(q), similar(k)
    st = GC.@preserve qo ko q k cos sin ccall((:nnop_llama_rope, libnnop()), Cint,
        (Ptr{RopeDesc}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cfloat, Ptr{Cvoid}),
        d, devptr(qo), devptr(ko), devptr(q), devptr(k), devptr(cos), devptr(sin), bwd ? -1f0 : 1f0, hipstream())
    st == 0 || error("libnnop_hip: " * unsafe_string(ccall((:nnop_strerror, libnnop()), Cstring, (Cint,), st)))
    return qo, ko
end

# struct nnop_softmax_desc (include/nnop_hip.h)
struct SoftmaxDesc
    dtype::Int32; n::Int32; batch::Int64
end

# src/softmax.jl:60-68.  The rrule (:82-86) calls online_softmax / ∇online_softmax generically.
function NNop.online_softmax(x::ROCMatrix{T}) where T <: HipFloat
    y = similar(x)
    d = Ref(SoftmaxDesc(nnop_dtype(T), size(x, 1), size(x, 2)))
    st = GC.@preserve y x ccall((:nnop_online_softmax, libnnop()), Cint, (Ptr{SoftmaxDesc}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        d, devptr(y), devptr(x), hipstream())
    st == 0 || error("libnnop_hip: " * unsafe_string(ccall((:nnop_strerror, libnnop()), Cstring, (Cint,), st)))
    return y
end

# src/softmax.jl:70-80: the two broadcasts + reduction fused into one pass (first derivatives only, like the
# reference's fast path :75-78; under nested differentiation the generic method still applies).
function NNop.∇online_softmax(Δ::ROCMatrix{T}, y::ROCMatrix{T}) where T <: HipFloat
    NNop.within_gradient(y) && return invoke(NNop.∇online_softmax, Tuple{AbstractArray, AbstractArray}, Δ, y)
    dx = similar(y)
    d = Ref(SoftmaxDesc(nnop_dtype(T), size(y, 1), size(y, 2)))
    st = GC.@preserve dx Δ y ccall((:nnop_online_softmax_bwd, libnnop()), Cint,
        (Ptr{SoftmaxDesc}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), d, devptr(dx), devptr(Δ), devptr(y), hipstream())
    st == 0 || error("libnnop_hip: " * unsafe_string(ccall((:nnop_strerror, libnnop()), Cstring, (Cint,), st)))
    return dx
end

# struct nnop_norm_desc (include/nnop_hip.h)
struct NormDesc
    dtype::Int32; w_dtype::Int32; emb::Int32; reserved::Int32; n::Int64
end
normdesc(x, w) = Ref(NormDesc(nnop_dtype(eltype(x)), nnop_dtype(eltype(w)), size(x, 1), 0, size(x, 2)))
ok(st) = st == 0 || error("libnnop_hip: " * unsafe_string(ccall((:nnop_strerror, libnnop()), Cstring, (Cint,), st)))

# src/rms_norm.jl:117-137 (the public rms_norm and the rrule, :171-185, call these generically)
function NNop._rms_norm(x::ROCMatrix{T}, w::ROCVector{W}; ϵ::Float32, offset::Float32 = 0f0) where {T <: HipFloat, W <: HipFloat}
    @assert size(x, 1) == length(w)
    y = similar(x); rms = ROCArray{Float32}(undef, size(x, 2))
    ok(GC.@preserve y rms x w ccall((:nnop_rms_norm, libnnop()), Cint,
        (Ptr{NormDesc}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cfloat, Cfloat, Ptr{Cvoid}),
        normdesc(x, w), devptr(y), devptr(rms), devptr(x), devptr(w), offset, ϵ, hipstream()))
    return y, rms
end

# src/rms_norm.jl:139-169: dw comes back already summed over rows (Float32, :146)
function NNop.∇rms_norm(Δ::ROCMatrix{T}, rms::ROCVector{Float32}, x::ROCMatrix{T}, w::ROCVector{W};
                        offset::Float32) where {T <: HipFloat, W <: HipFloat}
    d = normdesc(x, w)
    dx = similar(x); dw = ROCArray{Float32}(undef, size(x, 1))
    nbytes = ccall((:nnop_norm_bwd_workspace_bytes, libnnop()), Csize_t, (Ptr{NormDesc}, Cint), d, 0)
    ws = ROCArray{UInt8}(undef, nbytes)
    ok(GC.@preserve dx dw Δ rms x w ws ccall((:nnop_rms_norm_bwd, libnnop()), Cint,
        (Ptr{NormDesc}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cfloat, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
        d, devptr(dx), devptr(dw), devptr(Δ), devptr(rms), devptr(x), devptr(w), offset, devptr(ws), nbytes, hipstream()))
    return dx, dw
end

# src/layer_norm.jl:150-170
function NNop._layer_norm(x::ROCMatrix{T}, w::ROCVector{W}, b::ROCVector{W}; ϵ::Float32 = 1f-6) where {T <: HipFloat, W <: HipFloat}
    y = similar(x); μ = ROCArray{Float32}(undef, size(x, 2)); Σ = similar(μ)
    ok(GC.@preserve y μ Σ x w b ccall((:nnop_layer_norm, libnnop()), Cint,
        (Ptr{NormDesc}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cfloat, Ptr{Cvoid}),
        normdesc(x, w), devptr(y), devptr(μ), devptr(Σ), devptr(x), devptr(w), devptr(b), ϵ, hipstream()))
    return y, μ, Σ
end

# src/layer_norm.jl:172-204: dw, db in eltype(w) (:179-180), already summed over rows
function NNop.∇layer_norm(Δ::ROCMatrix{T}, μ::ROCVector{Float32}, Σ::ROCVector{Float32}, x::ROCMatrix{T},
                          w::ROCVector{W}, b::ROCVector{W}) where {T <: HipFloat, W <: HipFloat}
    d = normdesc(x, w)
    dx = similar(x); dw = similar(w); db = similar(b)
    nbytes = ccall((:nnop_norm_bwd_workspace_bytes, libnnop()), Csize_t, (Ptr{NormDesc}, Cint), d, 1)
    ws = ROCArray{UInt8}(undef, nbytes)
    ok(GC.@preserve dx dw db Δ μ Σ x w ws ccall((:nnop_layer_norm_bwd, libnnop()), Cint,
        (Ptr{NormDesc}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
        d, devptr(dx), devptr(dw), devptr(db), devptr(Δ), devptr(μ), devptr(Σ), devptr(x), devptr(w), devptr(ws), nbytes, hipstream()))
    return dx, dw, db
end

end # module
