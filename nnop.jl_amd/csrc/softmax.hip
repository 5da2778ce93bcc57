// softmax.hip -- NNop.online_softmax and its pullback for gfx950 (HBM-streaming).
//
//   y[:, b] = softmax(x[:, b])                      online_softmax!  src/softmax.jl:19-58 (host :60-68)
//   dx = dy .* y .- y .* sum(dy .* y; dims=1)        ∇online_softmax  src/softmax.jl:70-80 (broadcasts in the reference)
//
// Memory: Julia (N, batch) column-major == row-major [batch][N]; softmax runs along the contiguous axis.
// A row lives in registers (row_common.hpp) so x is read once and y written once; the (max, denominator) pair is
// reduced with the MD monoid of the reference (:1-16).  Arithmetic fp32, exp via exp2 of a pre-scaled argument.
// Bound: HBM -- N*sizeof(T) read + N*sizeof(T) written per row (backward: 2 reads + 1 write).
#include <type_traits>
#include "row_common.hpp"
#include "fa_launch.hpp"

namespace nnop {

struct SoftmaxParams {
    void* out;              // y (fwd) / dx (bwd)
    const void* a;          // x (fwd) / dy (bwd)
    const void* b;          // unused (fwd) / y (bwd)
    int N;
    long long rows;
};

template <typename T, int G, int C, bool BWD>
__global__ __launch_bounds__(G > 256 ? G : 256, G > 256 ? 1 : 2) void softmax_reg_kernel(const SoftmaxParams p) {
    constexpr int RPB = G >= 256 ? 1 : 256 / G;                   // rows per workgroup
    __shared__ MD md_slots[G > 64 ? G / 64 : 1];
    __shared__ float f_slots[G > 64 ? G / 64 : 1];
    const int lane = threadIdx.x % G;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / G;
    if (G == 64 && row >= p.rows) return;                         // whole waves only: no barrier is skipped
    const size_t off = (size_t)row * p.N;
    RowRegs<T, G, C> r;
    constexpr int VEC = RowRegs<T, G, C>::VEC;
    if constexpr (!BWD) {
        r.load((const T*)p.a + off, p.N, lane, -INFINITY);
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < C; ++j)
#pragma unroll
            for (int i = 0; i < VEC; ++i) m = fmaxf(m, r.get(j, i));
        repack(r);
        // exp(x - m) of the lane's own elements: one MD per lane, as the reference accumulates (:33-40)
        float d = 0.f;
        const float ml = m * kLog2e;
#pragma unroll
        for (int j = 0; j < C; ++j)
#pragma unroll
            for (int i = 0; i < VEC; ++i) d += fast_exp2(__builtin_fmaf(r.get(j, i), kLog2e, -ml));
        if (m == -INFINITY) d = 0.f;                              // (-inf) - (-inf): the reference's NaN guard (:11)
        const MD tot = group_allreduce<G>(MD{m, d}, 0, md_slots);
        repack(r);
        const float inv = 1.0f / tot.d;                           // :49
        const float mt = tot.m * kLog2e;
#pragma unroll
        for (int j = 0; j < C; ++j)
#pragma unroll
            for (int i = 0; i < VEC; ++i) r.set(j, i, fast_exp2(__builtin_fmaf(r.get(j, i), kLog2e, -mt)) * inv);   // :54
        r.store((T*)p.out + off, p.N, lane);
    } else {
        RowRegs<T, G, C> y;
        r.load((const T*)p.a + off, p.N, lane, 0.f);
        y.load((const T*)p.b + off, p.N, lane, 0.f);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < C; ++j)
#pragma unroll
            for (int i = 0; i < VEC; ++i) s = __builtin_fmaf(r.get(j, i), y.get(j, i), s);
        s = group_allreduce<G>(s, SumOp{}, f_slots);
        repack(r); repack(y);
#pragma unroll
        for (int j = 0; j < C; ++j)
#pragma unroll
            for (int i = 0; i < VEC; ++i) r.set(j, i, y.get(j, i) * (r.get(j, i) - s));
        r.store((T*)p.out + off, p.N, lane);
    }
}

// Any N (odd lengths, rows longer than the register shapes): workgroup of 1024 lanes per row, the reference's own
// two-pass structure (the second pass re-reads the row, from L2 / MALL when it is still there).  VEC = 16-byte
// accesses when the row length allows, else element accesses.
template <typename T, bool BWD, int VEC>
__global__ __launch_bounds__(1024) void softmax_generic_kernel(const SoftmaxParams p) {
    typedef T tv __attribute__((ext_vector_type(VEC)));
    __shared__ MD md_slots[16];
    __shared__ float f_slots[16];
    const size_t off = (size_t)blockIdx.x * p.N;
    const tv* __restrict__ a = reinterpret_cast<const tv*>((const T*)p.a + off);
    tv* __restrict__ out = reinterpret_cast<tv*>((T*)p.out + off);
    const int n = p.N / VEC;
    if constexpr (!BWD) {
        MD md{-INFINITY, 0.f};
        for (int e = threadIdx.x; e < n; e += 1024) {
            const tv x = a[e];
            float m = to_f32(x[0]);
#pragma unroll
            for (int i = 1; i < VEC; ++i) m = fmaxf(m, to_f32(x[i]));
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) d += fast_exp2((to_f32(x[i]) - m) * kLog2e);
            md = md_reduce(md, MD{m, m == -INFINITY ? 0.f : d});
        }
        const MD tot = group_allreduce<1024>(md, 0, md_slots);
        const float inv = 1.0f / tot.d, mt = tot.m * kLog2e;
        for (int e = threadIdx.x; e < n; e += 1024) {
            const tv x = a[e];
            tv o;
#pragma unroll
            for (int i = 0; i < VEC; ++i) o[i] = from_f32<T>(fast_exp2(__builtin_fmaf(to_f32(x[i]), kLog2e, -mt)) * inv);
            out[e] = o;
        }
    } else {
        const tv* __restrict__ y = reinterpret_cast<const tv*>((const T*)p.b + off);
        float s = 0.f;
        for (int e = threadIdx.x; e < n; e += 1024) {
            const tv dy = a[e], yy = y[e];
#pragma unroll
            for (int i = 0; i < VEC; ++i) s = __builtin_fmaf(to_f32(dy[i]), to_f32(yy[i]), s);
        }
        s = group_allreduce<1024>(s, SumOp{}, f_slots);
        for (int e = threadIdx.x; e < n; e += 1024) {
            const tv dy = a[e], yy = y[e];
            tv o;
#pragma unroll
            for (int i = 0; i < VEC; ++i) o[i] = from_f32<T>(to_f32(yy[i]) * (to_f32(dy[i]) - s));
            out[e] = o;
        }
    }
}

template <typename T, bool BWD>
static int launch_softmax_t(const SoftmaxParams& p, hipStream_t s) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const bool aligned = (((uintptr_t)p.out | (uintptr_t)p.a | (uintptr_t)p.b) & 15) == 0;
    bool done = false;
    if (aligned)
        done = dispatch_row_shape<T, BWD ? 2 : 1>(p.N, [&](auto shape) {
            constexpr int G = decltype(shape)::G, C = decltype(shape)::C;
            constexpr int RPB = G >= 256 ? 1 : 256 / G, NT = G > 256 ? G : 256;
            const long long grid = (p.rows + RPB - 1) / RPB;
            hipLaunchKernelGGL((softmax_reg_kernel<T, G, C, BWD>), dim3((unsigned)grid), dim3(NT), 0, s, p);
        });
    if (!done) {
        if (aligned && p.N % VEC == 0)
            hipLaunchKernelGGL((softmax_generic_kernel<T, BWD, VEC>), dim3((unsigned)p.rows), dim3(1024), 0, s, p);
        else
            hipLaunchKernelGGL((softmax_generic_kernel<T, BWD, 1>), dim3((unsigned)p.rows), dim3(1024), 0, s, p);
    }
    return hipGetLastError() == hipSuccess ? NNOP_OK : NNOP_ERR_HIP;
}

int launch_softmax(const nnop_softmax_desc& d, void* out, const void* a, const void* b, bool bwd, hipStream_t s) {
    SoftmaxParams p{out, a, b, d.n, d.batch};
    if (d.batch > 0x7fffffffLL) return NNOP_ERR_SHAPE;
    switch (d.dtype) {
        case NNOP_F32:  return bwd ? launch_softmax_t<float, true>(p, s) : launch_softmax_t<float, false>(p, s);
        case NNOP_F16:  return bwd ? launch_softmax_t<_Float16, true>(p, s) : launch_softmax_t<_Float16, false>(p, s);
        case NNOP_BF16: return bwd ? launch_softmax_t<__bf16, true>(p, s) : launch_softmax_t<__bf16, false>(p, s);
    }
    return NNOP_ERR_DTYPE;
}

}  // namespace nnop
