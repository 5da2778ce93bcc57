# runtests_shim.jl -- for a maintainer WITH Julia + AMDGPU.jl on an MI355X: checks that NNopHIPExt routes
# NNop.flash_attention and its rrule to libnnop_hip.so and passes the reference's own attention grids.
#
# SOURCE ONLY in this repository: the build image has no Julia, so this file is never executed by the test-suite.  The
# same grids run against the same library through the Python mirror in tests/test_reference_grids_gpu.py.
#
#   NNOP_HIP_LIB=/path/to/libnnop_hip.so julia --project=<NNop checkout with ext/NNopHIPExt.jl> runtests_shim.jl
#
# Grids and tolerances: test/attention_tests.jl:6-48, test/causal_attention_tests.jl:6-46,
# test/gqa_attention_tests.jl:6-33 of the reference, verbatim; the naive formula is test/attention_testsetup.jl:21-45
# without its Einops / NNlib dependencies.
using Test
using AMDGPU
using NNop
import Zygote

const Ext = Base.get_extension(NNop, :NNopHIPExt)
@assert !isnothing(Ext) "NNopHIPExt is not loaded: see INTEGRATION.md section 2"

# --- naive attention (test/attention_testsetup.jl:21-45), batched over (head, batch) with plain loops ---------------
function naive_softmax(x; dims = 1)
    mx = maximum(x; dims)
    tmp = exp.(x .- mx)
    return tmp ./ sum(tmp; dims)
end
function naive_attention(q, k, v, pair = nothing; causal::Bool, kpad_mask = nothing)
    E, QL, QH, B = size(q)
    KL, KVH = size(k, 2), size(k, 3)
    n = QH ÷ KVH
    scale = inv(sqrt(eltype(q)(E)))
    os = map(Iterators.product(1:QH, 1:B)) do (h, b)
        kh = cld(h, n)                                            # src/attention.jl:28
        a = (permutedims(k[:, :, kh, b]) * q[:, :, h, b]) .* scale   # (KL, QL)
        if causal
            a = a .+ ifelse.((1:KL) .<= permutedims(1:QL), zero(scale), typemin(scale))
        end
        if !isnothing(kpad_mask)
            a = a .+ log.(eltype(q).(kpad_mask[:, b]))
        end
        if !isnothing(pair)
            a = a .+ permutedims(pair[h, :, :, b])
        end
        v[:, :, kh, b] * naive_softmax(a; dims = 1)               # (E, QL)
    end
    return cat((cat(os[:, b]...; dims = 3) for b in 1:B)...; dims = 4)
end

roc(x) = ROCArray(x)
roc(::Nothing) = nothing

function check_case(q, k, v, pair, kpad_mask; causal)
    o1, g1 = Zygote.withgradient(q, k, v, pair) do q, k, v, pair
        sum(naive_attention(q, k, v, pair; causal, kpad_mask))
    end
    dq, dk, dv, dp, dm = roc(q), roc(k), roc(v), roc(pair), roc(kpad_mask)
    # the cotangent of sum(...) is a lazy FillArrays.Fill: NNopHIPExt materialises it on the device (no fall-through to the
    # reference's KernelAbstractions backward)
    o2, g2 = Zygote.withgradient(dq, dk, dv, dp) do q, k, v, pair
        sum(NNop.flash_attention(q, k, v, pair; causal, kpad_mask = dm))
    end
    @test isapprox(o1, o2; atol = 1e-3, rtol = 1e-3)
    for i in 1:3
        @test isapprox(g1[i], Array(g2[i]); atol = 1e-3, rtol = 1e-3)
    end
    isnothing(pair) || @test isapprox(g1[4], Array(g2[4]); atol = 1e-3, rtol = 1e-3)
end

@testset "NNopHIPExt dispatch" begin
    q = AMDGPU.randn(Float32, 64, 256, 2, 1)
    @test which(NNop._flash_attention, typeof.((q, q, q))).module === Ext
    o, ms, ls = NNop._flash_attention(q, q, q; causal = false)
    @test which(NNop.∇flash_attention, typeof.((o, o, ms, ls, q, q, q))).module === Ext
    @test_throws ErrorException NNop.flash_attention(AMDGPU.randn(Float32, 48, 64, 2, 1), AMDGPU.randn(Float32, 48, 64, 2, 1),
                                                     AMDGPU.randn(Float32, 48, 64, 2, 1); causal = false)   # :143
end

@testset "Flash Attention padmask=$pm pair=$up E=$E QL=$QL KL=$KL" for pm in (false, true), up in (false, true),
        E in (16, 32, 64), QL in (255, 256, 511, 512, 1024), KL in (255, 256, 511, 512, 1024)
    T, H, B = Float32, 2, 3
    q, k, v = randn(T, E, QL, H, B), randn(T, E, KL, H, B), randn(T, E, KL, H, B)
    mask = pm ? (m = ones(Bool, KL, B); m[end-10:end, end] .= false; m) : nothing
    pair = up ? randn(T, H, QL, KL, B) : nothing
    check_case(q, k, v, pair, mask; causal = false)
end

@testset "Causal Flash Attention padmask=$pm pair=$up E=$E L=$L" for pm in (false, true), up in (false, true),
        E in (16, 32, 64), L in (255, 256, 511, 512, 1024)
    T, H, B = Float32, 2, 3
    q, k, v = randn(T, E, L, H, B), randn(T, E, L, H, B), randn(T, E, L, H, B)
    mask = pm ? (m = ones(Bool, L, B); m[end-10:end, end] .= false; m) : nothing
    pair = up ? randn(T, H, L, L, B) : nothing
    check_case(q, k, v, pair, mask; causal = true)
end

@testset "GQA QH=$QH KVH=$KVH causal=$causal E=$E L=$L" for QH in (4, 6, 8), KVH in (1, 2), causal in (false, true),
        E in (32, 64), L in (255, 256, 257, 512)
    T, B = Float32, 2
    q, k, v = randn(T, E, L, QH, B), randn(T, E, L, KVH, B), randn(T, E, L, KVH, B)
    check_case(q, k, v, nothing, nothing; causal)
end
