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
    long long n_items_q, n_items;   // work items over q; over q and k
};

// Heads per lane.  The cos / sin half-rows of a (position, batch) are shared by every head; at D = 128 they are 512 B of
// L2 -> L1 traffic against a 256-B bf16 row.  A lane can keep its cos / sin chunk in registers and apply it to HG heads.
// Measured on MI355X (profiles/r01/NOTES.md): HG = 1, 2, 4, 8 are within +-5 % of each other (HG = 2: +4..10 % at B = 4,
// -4 % at L = 32768 B = 1 where two heads are 8 MB apart) -- that traffic is not the limiter.  Default: one head.
#ifndef NNOP_ROPE_HG
#define NNOP_ROPE_HG 1
#endif

// VEC: elements per lane per half (8 on the vector path, 1 on the generic path when D/2 % 8 != 0).
// Work item = (batch, group of HG heads, position, chunk), chunk fastest then position: a wave's lanes cover consecutive
// chunks of consecutive rows of one head, so every load / store instruction is contiguous in memory.
template <typename T, typename CS, int VEC, int HG>
__global__ __launch_bounds__(256) void rope_kernel(const RopeParams p) {
    typedef T tv __attribute__((ext_vector_type(VEC)));
    typedef CS cv __attribute__((ext_vector_type(VEC)));
    const int half = p.D >> 1;
    const int cpr = half / VEC;                                   // lanes per row
    long long it = (long long)blockIdx.x * 256 + threadIdx.x;
    if (it >= p.n_items) return;
    const bool is_k = it >= p.n_items_q;
    if (is_k) it -= p.n_items_q;
    const int H = is_k ? p.KH : p.QH;
    const int n_hg = (H + HG - 1) / HG;
    const int c = (int)(it % cpr) * VEC;                          // first element of this lane's chunk
    long long t = it / cpr;
    const int l = (int)(t % p.L); t /= p.L;
    const int hg = (int)(t % n_hg);
    const int b = (int)(t / n_hg);
    const int h0 = hg * HG;
    const T* __restrict__ x = (const T*)(is_k ? p.k : p.q) + (((size_t)b * H + h0) * p.L + l) * p.D;
    T* __restrict__ y = (T*)(is_k ? p.ko : p.qo) + (((size_t)b * H + h0) * p.L + l) * p.D;
    const size_t hstride = (size_t)p.L * p.D;
    const CS* __restrict__ cs = (const CS*)p.cos + ((size_t)b * p.L + l) * p.D;
    const CS* __restrict__ sn = (const CS*)p.sin + ((size_t)b * p.L + l) * p.D;
    cv cc, ss;
    tv x1[HG], x2[HG];
    if constexpr (VEC > 1) {
        // 16-byte accesses only: an 8 x fp32 chunk is loaded as two 16-byte halves, so 16-byte alignment of the
        // tensors (checked by the launcher) is all the vector path needs
        typedef CS ch __attribute__((ext_vector_type(sizeof(CS) == 4 ? VEC / 2 : VEC)));
        if constexpr (sizeof(CS) == 4) {
            const ch c0 = *reinterpret_cast<const ch*>(cs + c), c1 = *reinterpret_cast<const ch*>(cs + c + VEC / 2);
            const ch s0 = *reinterpret_cast<const ch*>(sn + c), s1 = *reinterpret_cast<const ch*>(sn + c + VEC / 2);
#pragma unroll
            for (int j = 0; j < VEC / 2; ++j) { cc[j] = c0[j]; cc[VEC / 2 + j] = c1[j]; ss[j] = s0[j]; ss[VEC / 2 + j] = s1[j]; }
        } else {
            cc = *reinterpret_cast<const ch*>(cs + c);
            ss = *reinterpret_cast<const ch*>(sn + c);
        }
    } else {
        cc[0] = cs[c]; ss[0] = sn[c];
    }
#pragma unroll
    for (int u = 0; u < HG; ++u) {
        if (h0 + u < H) {
            if constexpr (VEC > 1) {
                x1[u] = *reinterpret_cast<const tv*>(x + u * hstride + c);
                x2[u] = *reinterpret_cast<const tv*>(x + u * hstride + half + c);
            } else {
                x1[u][0] = x[u * hstride + c]; x2[u][0] = x[u * hstride + half + c];
            }
        }
    }
    float co[VEC], si[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { co[j] = to_f32(cc[j]); si[j] = to_f32(ss[j]) * p.sin_sign; }
#pragma unroll
    for (int u = 0; u < HG; ++u) {
        if (h0 + u < H) {
            tv o1, o2;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float a = to_f32(x1[u][j]), bq = to_f32(x2[u][j]);
                o1[j] = from_f32<T>(a * co[j] - bq * si[j]);
                o2[j] = from_f32<T>(bq * co[j] + a * si[j]);
            }
            if constexpr (VEC > 1) {
                *reinterpret_cast<tv*>(y + u * hstride + c) = o1;
                *reinterpret_cast<tv*>(y + u * hstride + half + c) = o2;
            } else {
                y[u * hstride + c] = o1[0];
                y[u * hstride + half + c] = o2[0];
            }
        }
    }
}

template <typename T, typename CS>
static int launch_rope_t(RopeParams p, hipStream_t s) {
    constexpr int HG = NNOP_ROPE_HG;
    const int half = p.D >> 1;
    // the vector path issues 16-byte loads / stores: every tensor must be 16-byte aligned (rows then are, since
    // D % 16 == 0 there); an offset view from a C or Julia caller takes the element-wise kernel instead
    const bool aligned = ((reinterpret_cast<uintptr_t>(p.q) | reinterpret_cast<uintptr_t>(p.k) |
                           reinterpret_cast<uintptr_t>(p.qo) | reinterpret_cast<uintptr_t>(p.ko) |
                           reinterpret_cast<uintptr_t>(p.cos) | reinterpret_cast<uintptr_t>(p.sin)) & 15) == 0;
    const bool vec = (half % 8) == 0 && aligned;
    const long long cpr = vec ? half / 8 : half;
    p.n_items_q = (long long)p.B * ((p.QH + HG - 1) / HG) * p.L * cpr;
    p.n_items = p.n_items_q + (long long)p.B * ((p.KH + HG - 1) / HG) * p.L * cpr;
    const long long grid = (p.n_items + 255) / 256;
    if (grid <= 0 || grid > 0x7fffffffLL) return NNOP_ERR_SHAPE;
    if (vec) hipLaunchKernelGGL((rope_kernel<T, CS, 8, HG>), dim3((unsigned)grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((rope_kernel<T, CS, 1, HG>), dim3((unsigned)grid), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? NNOP_OK : NNOP_ERR_HIP;
}

int launch_rope(const nnop_rope_desc& d, void* qo, void* ko, const void* q, const void* k, const void* cos,
                const void* sin, float sin_sign, hipStream_t s) {
    RopeParams p;
    p.qo = qo; p.ko = ko; p.q = q; p.k = k; p.cos = cos; p.sin = sin;
    p.D = d.dim; p.L = d.seq; p.QH = d.qh; p.KH = d.kh; p.B = d.batch;
    p.sin_sign = sin_sign;
    p.n_items_q = p.n_items = 0;                                 // set per instantiation in launch_rope_t
    const bool cs32 = d.cs_dtype == NNOP_F32;
    switch (d.dtype) {
        case NNOP_F32:  return launch_rope_t<float, float>(p, s);
        case NNOP_F16:  return cs32 ? launch_rope_t<_Float16, float>(p, s) : launch_rope_t<_Float16, _Float16>(p, s);
        case NNOP_BF16: return cs32 ? launch_rope_t<__bf16, float>(p, s) : launch_rope_t<__bf16, __bf16>(p, s);
    }
    return NNOP_ERR_DTYPE;
}

}  // namespace nnop
