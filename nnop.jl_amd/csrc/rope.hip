// rope.hip -- Llama rotary embedding for gfx950 (HBM-streaming).
//
// Replaces llama_rope! (src/rope/llama_rope.jl:24-65), which runs one workgroup per (head, batch) with every thread
// walking the head dim of "its" sequence position serially (lane stride = D elements: uncoalesced) after the host has
// copied q and k (:75-76).  Here: out of place in ONE pass (no copy), every lane moves 16 bytes of the first half
// of a row and the matching 16 bytes of the second half, rows of q and k form one flat grid, cos/sin (shared by all
// heads of a (position, batch)) are read through L2.  Arithmetic in fp32, one rounding on store.
// Bound: HBM -- q, k read once + written once (+ the D/2-wide cos/sin rows).
#include "fa_common.hpp"
#include "fa_launch.hpp"

namespace nnop {

struct RopeParams {
    void *qo, *ko;
    const void *q, *k, *cos, *sin;
    int D, L, QH, KH, B;
    float sin_sign;
    long long n_rows_q, n_rows;     // rows of q; rows of q + rows of k
};

// Measured on MI355X (tools/vars_rope.sh, profiles/r01/NOTES.md): U in {1,2,4} x nontemporal stores on/off are all
// within noise of each other (5.4-5.7 TB/s at Llama-8B shapes); nontemporal LOADS cost 10 % in fp32.  Default: the
// simplest form.
#ifndef NNOP_ROPE_U
#define NNOP_ROPE_U 1
#endif
#ifndef NNOP_ROPE_NT
#define NNOP_ROPE_NT 0
#endif

template <typename V> __device__ __forceinline__ void rope_store(V* dst, const V& v) {
#if NNOP_ROPE_NT
    __builtin_nontemporal_store(v, dst);          // outputs are not re-read by this kernel: keep L2 for cos/sin
#else
    *dst = v;
#endif
}

// VEC: elements per lane per half (8 on the vector path, 1 on the generic path when D/2 % 8 != 0).
// U: chunks per lane, strided by the grid so that a wave's accesses stay contiguous; all 2U row loads are issued
// before the first use.
template <typename T, typename CS, int VEC, int U>
__global__ __launch_bounds__(256) void rope_kernel(const RopeParams p) {
    typedef T tv __attribute__((ext_vector_type(VEC)));
    typedef CS cv __attribute__((ext_vector_type(VEC)));
    const int half = p.D >> 1;
    const int cpr = half / VEC;                                   // lanes per row
    const long long n_chunks = p.n_rows * cpr;
    const long long stride = (long long)gridDim.x * 256;
    const long long gid0 = (long long)blockIdx.x * 256 + threadIdx.x;
    tv x1[U], x2[U];
    cv cc[U], ss[U];
    T* y[U];
    int c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        long long gid = gid0 + u * stride;
        y[u] = nullptr;
        if (gid >= n_chunks) continue;
        const long long row = gid / cpr;
        c[u] = (int)(gid - row * cpr) * VEC;                      // first element of this lane's chunk
        // row -> (tensor, batch, head, position)
        const bool is_k = row >= p.n_rows_q;
        const long long r = is_k ? row - p.n_rows_q : row;
        const int H = is_k ? p.KH : p.QH;
        const int l = (int)(r % p.L);
        const int b = (int)(r / ((long long)p.L * H));
        const T* __restrict__ x = (const T*)(is_k ? p.k : p.q) + r * p.D;
        y[u] = (T*)(is_k ? p.ko : p.qo) + r * p.D;
        const CS* __restrict__ cs = (const CS*)p.cos + ((size_t)b * p.L + l) * p.D;
        const CS* __restrict__ sn = (const CS*)p.sin + ((size_t)b * p.L + l) * p.D;
        if constexpr (VEC > 1) {
            x1[u] = *reinterpret_cast<const tv*>(x + c[u]);
            x2[u] = *reinterpret_cast<const tv*>(x + half + c[u]);
            cc[u] = *reinterpret_cast<const cv*>(cs + c[u]);
            ss[u] = *reinterpret_cast<const cv*>(sn + c[u]);
        } else {
            x1[u][0] = x[c[u]]; x2[u][0] = x[half + c[u]];
            cc[u][0] = cs[c[u]]; ss[u][0] = sn[c[u]];
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (!y[u]) continue;
        tv o1, o2;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float a = to_f32(x1[u][j]), bq = to_f32(x2[u][j]);
            const float co = to_f32(cc[u][j]), si = to_f32(ss[u][j]) * p.sin_sign;
            o1[j] = from_f32<T>(a * co - bq * si);
            o2[j] = from_f32<T>(bq * co + a * si);
        }
        if constexpr (VEC > 1) {
            rope_store(reinterpret_cast<tv*>(y[u] + c[u]), o1);
            rope_store(reinterpret_cast<tv*>(y[u] + half + c[u]), o2);
        } else {
            y[u][c[u]] = o1[0];
            y[u][half + c[u]] = o2[0];
        }
    }
}

template <typename T, typename CS>
static int launch_rope_t(const RopeParams& p, hipStream_t s) {
    constexpr int U = NNOP_ROPE_U;
    const int half = p.D >> 1;
    const bool vec = (half % 8) == 0;
    const long long n_chunks = p.n_rows * (vec ? half / 8 : half);
    const long long grid = (n_chunks + 256LL * U - 1) / (256LL * U);
    if (grid <= 0 || grid > 0x7fffffffLL) return NNOP_ERR_SHAPE;
    if (vec) hipLaunchKernelGGL((rope_kernel<T, CS, 8, U>), dim3((unsigned)grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((rope_kernel<T, CS, 1, U>), dim3((unsigned)grid), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? NNOP_OK : NNOP_ERR_HIP;
}

int launch_rope(const nnop_rope_desc& d, void* qo, void* ko, const void* q, const void* k, const void* cos,
                const void* sin, float sin_sign, hipStream_t s) {
    RopeParams p;
    p.qo = qo; p.ko = ko; p.q = q; p.k = k; p.cos = cos; p.sin = sin;
    p.D = d.dim; p.L = d.seq; p.QH = d.qh; p.KH = d.kh; p.B = d.batch;
    p.sin_sign = sin_sign;
    p.n_rows_q = (long long)d.batch * d.qh * d.seq;
    p.n_rows = p.n_rows_q + (long long)d.batch * d.kh * d.seq;
    const bool cs32 = d.cs_dtype == NNOP_F32;
    switch (d.dtype) {
        case NNOP_F32:  return launch_rope_t<float, float>(p, s);
        case NNOP_F16:  return cs32 ? launch_rope_t<_Float16, float>(p, s) : launch_rope_t<_Float16, _Float16>(p, s);
        case NNOP_BF16: return cs32 ? launch_rope_t<__bf16, float>(p, s) : launch_rope_t<__bf16, __bf16>(p, s);
    }
    return NNOP_ERR_DTYPE;
}

}  // namespace nnop
