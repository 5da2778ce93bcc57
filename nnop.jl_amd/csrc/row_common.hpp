// row_common.hpp -- building blocks of the row-wise HBM-streaming operators (softmax, RMSNorm, LayerNorm).
//
//  * group reduce: the gfx950 counterpart of NNop's `@groupreduce op val` (src/groupreduce.jl:13-43), which folds a
//    workgroup's values through an LDS array in log2(groupsize) barrier-separated steps.  Here a wave folds its 64
//    values in registers with a butterfly (DPP quad_perm / row_half_mirror / row_mirror, then v_permlane16_swap and
//    v_permlane32_swap -- no LDS, no barrier) and waves meet through ONE LDS hop; every lane gets the result (the
//    reference returns it on lane 1 and re-broadcasts through another LDS slot, src/softmax.jl:43-49).
//  * RowRegs: one row of N elements held in the registers of a group of G lanes (G = 64: wave per row; G = 256:
//    workgroup per row) as C 16-byte chunks per lane, chunk j of lane l covering elements [(j*G + l)*VEC, +VEC): every
//    load/store instruction of a wave is one contiguous 1 KiB segment, and x is read from HBM exactly once (the
//    reference kernels re-read the row for every pass: src/softmax.jl:35-57, src/layer_norm.jl:24-62).
#pragma once
#include <type_traits>
#include "fa_common.hpp"

namespace nnop {

// ---- cross-lane exchange: value of lane (l ^ STEP) -------------------------------------------------------------
template <int CTRL> NNOP_DEV float dpp_mov(float x) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), CTRL, 0xf, 0xf, true));
}
// After the previous butterfly steps all lanes of a 2^k-lane block hold the same value, so any lane of the partner
// block will do: mirrors stand in for xor 4 / xor 8.
template <int STEP> NNOP_DEV float xlane(float x) {
    if constexpr (STEP == 1) return dpp_mov<0xB1>(x);            // quad_perm [1,0,3,2]
    else if constexpr (STEP == 2) return dpp_mov<0x4E>(x);       // quad_perm [2,3,0,1]
    else if constexpr (STEP == 4) return dpp_mov<0x141>(x);      // row_half_mirror: l -> 7 - l within 8
    else if constexpr (STEP == 8) return dpp_mov<0x140>(x);      // row_mirror: l -> 15 - l within 16
    else {
        const uint32_t u = __float_as_uint(x);
        auto r = (STEP == 16) ? __builtin_amdgcn_permlane16_swap(u, u, false, false)
                              : __builtin_amdgcn_permlane32_swap(u, u, false, false);
        // {own, partner} in an order that depends on the lane's row/half: whichever is not bit-identical to the own
        // value is the partner's (if both are identical either is right)
        return __uint_as_float(r[0] == u ? r[1] : r[0]);
    }
}

struct SumOp { NNOP_DEV float operator()(float a, float b) const { return a + b; } };

// Online-softmax pair (running max, denominator): MD / md_reduce of src/softmax.jl:1-16, including the NaN guard
// for (-Inf) - (-Inf).
struct MD { float m, d; };
struct Sum2 { float a, b; };

NNOP_DEV MD md_reduce(MD a, MD b) {
    const bool a_bigger = a.m > b.m;
    const MD big = a_bigger ? a : b, small = a_bigger ? b : a;
    float diff = small.m - big.m;
    diff = (diff != diff) ? -INFINITY : diff;
    return MD{big.m, big.d + small.d * __expf(diff)};
}

template <int STEP> NNOP_DEV float fold(float v, SumOp op) { return op(v, xlane<STEP>(v)); }
template <int STEP> NNOP_DEV MD fold(MD v, int) { return md_reduce(v, MD{xlane<STEP>(v.m), xlane<STEP>(v.d)}); }
template <int STEP> NNOP_DEV Sum2 fold(Sum2 v, int) { return Sum2{v.a + xlane<STEP>(v.a), v.b + xlane<STEP>(v.b)}; }

// all-reduce over the 64 lanes of a wave; OP: SumOp for float, ignored (int) for MD / Sum2
template <typename V, typename OP> NNOP_DEV V wave_allreduce(V v, OP op) {
    v = fold<1>(v, op); v = fold<2>(v, op); v = fold<4>(v, op);
    v = fold<8>(v, op); v = fold<16>(v, op); v = fold<32>(v, op);
    return v;
}

NNOP_DEV float combine(float a, float b, SumOp) { return a + b; }
NNOP_DEV MD combine(MD a, MD b, int) { return md_reduce(a, b); }
NNOP_DEV Sum2 combine(Sum2 a, Sum2 b, int) { return Sum2{a.a + b.a, a.b + b.b}; }

// all-reduce over a group of G lanes (G = 64: the wave; G = NW*64: the workgroup, one LDS hop, `slots` holds NW
// values and may be reused after the call returns only behind another barrier -- callers alternate two slot sets).
template <int G, typename V, typename OP> NNOP_DEV V group_allreduce(V v, OP op, V* slots) {
    v = wave_allreduce(v, op);
    if constexpr (G > 64) {
        constexpr int NW = G / 64;
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) slots[wave] = v;
        __syncthreads();
        V r = slots[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) r = combine(r, slots[w], op);
        return r;
    }
    return v;
}

// ---- a row in registers ----------------------------------------------------------------------------------------
template <typename T, int G, int C> struct RowRegs {
    static constexpr int VEC = 16 / (int)sizeof(T);
    static constexpr int CAP = G * C * VEC;                    // longest row this shape holds
    typedef T tv __attribute__((ext_vector_type(VEC)));
    tv raw[C];                                                 // kept in T: a 16-bit row costs half the registers

    NNOP_DEV static bool in_row(int j, int lane, int N) { return (j * G + lane) * VEC < N; }
    // lane: 0..G-1 within the row's group; chunks past N read as `fill`
    NNOP_DEV void load(const T* __restrict__ row, int N, int lane, float fill) {
        const T f = from_f32<T>(fill);
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int e = (j * G + lane) * VEC;
            tv t;
#pragma unroll
            for (int i = 0; i < VEC; ++i) t[i] = f;
            if (e < N) t = *reinterpret_cast<const tv*>(row + e);
            raw[j] = t;
        }
    }
    NNOP_DEV float get(int j, int i) const { return to_f32(raw[j][i]); }
    NNOP_DEV void set(int j, int i, float x) { raw[j][i] = from_f32<T>(x); }
    NNOP_DEV void store(T* __restrict__ row, int N, int lane) const {
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int e = (j * G + lane) * VEC;
            if (e < N) *reinterpret_cast<tv*>(row + e) = raw[j];
        }
    }
};

// Pick the register shape for a row of N elements: wave per row up to 8 chunks per lane, then a 256-lane workgroup
// per row, then a 1024-lane workgroup per row (<= 128 VGPRs per lane: 16 chunks of a 16-bit type, 16 of fp32 for
// kernels holding one row); longer rows, or rows whose byte length is not a multiple of 16, take an operator's
// generic (strided, two-pass) kernel.  ROWS_HELD: rows of registers the kernel keeps live (1 or 2).
template <int G_, int C_> struct RowShape { static constexpr int G = G_, C = C_; };

template <typename T, int ROWS_HELD, typename F> static inline bool dispatch_row_shape(long long N, F&& f) {
    constexpr int VEC = 16 / (int)sizeof(T);
    if (N % VEC != 0) return false;
    const long long chunks = N / VEC;
    if (chunks <= 64 * 1) { f(RowShape<64, 1>{}); return true; }
    if (chunks <= 64 * 2) { f(RowShape<64, 2>{}); return true; }
    if (chunks <= 64 * 4) { f(RowShape<64, 4>{}); return true; }
    if (chunks <= 64 * 8) { f(RowShape<64, 8>{}); return true; }
    if (chunks <= 256 * 4) { f(RowShape<256, 4>{}); return true; }
    if (chunks <= 256 * 8) { f(RowShape<256, 8>{}); return true; }
    if (chunks <= 1024 * 4) { f(RowShape<1024, 4>{}); return true; }
    if (chunks <= 1024 * 8) { f(RowShape<1024, 8>{}); return true; }
    if constexpr (ROWS_HELD == 1)
        if (chunks <= 1024 * 16) { f(RowShape<1024, 16>{}); return true; }
    return false;
}

// Shapes for kernels that keep several rows plus per-column fp32 accumulators live (the norm pullbacks): at most 16
// columns per lane (4 chunks of fp32, 2 of a 16-bit type), wider groups instead.
template <typename T, typename F> static inline bool dispatch_row_shape_narrow(long long N, F&& f) {
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr bool F32 = sizeof(T) == 4;
    if (N % VEC != 0) return false;
    const long long chunks = N / VEC;
    if (chunks <= 64 * 1) { f(RowShape<64, 1>{}); return true; }
    if (chunks <= 64 * 2) { f(RowShape<64, 2>{}); return true; }
    if constexpr (F32) { if (chunks <= 64 * 4) { f(RowShape<64, 4>{}); return true; } }
    if (chunks <= 256 * 1) { f(RowShape<256, 1>{}); return true; }
    if (chunks <= 256 * 2) { f(RowShape<256, 2>{}); return true; }
    if constexpr (F32) { if (chunks <= 256 * 4) { f(RowShape<256, 4>{}); return true; } }
    if (chunks <= 1024 * 1) { f(RowShape<1024, 1>{}); return true; }
    if (chunks <= 1024 * 2) { f(RowShape<1024, 2>{}); return true; }
    if constexpr (F32) { if (chunks <= 1024 * 4) { f(RowShape<1024, 4>{}); return true; } }
    return false;
}

// Opaque fence on a register row: values derived from it before the fence (fp32 expansions of a 16-bit row) cannot be
// kept alive across it, so a second pass re-converts from the packed registers instead of doubling the footprint.
template <typename R> NNOP_DEV void repack(R& r) {
#pragma unroll
    for (int j = 0; j < (int)(sizeof(r.raw) / sizeof(r.raw[0])); ++j) asm volatile("" : "+v"(r.raw[j]));
}

}  // namespace nnop
