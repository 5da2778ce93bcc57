// Compile-only probe: does hipcc keep O / Q pinned in AGPRs when every MFMA is inline asm with explicit
// register-class constraints?  hipcc --offload-arch=gfx950 -O3 -S probe_w64.hip, then count v_accvgpr / scratch.
#include "../../nnop.jl_amd/csrc/fa_common.hpp"
using namespace nnop;

NNOP_DEV f32x16 mfma_qk0(bf16x8 a, bf16x8 bq) {
    f32x16 d;
    asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(bq));
    return d;
}
NNOP_DEV void mfma_qk(f32x16& d, bf16x8 a, bf16x8 bq) {
    asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(bq));
}
NNOP_DEV void mfma_pv(f32x16& o, bf16x8 a, bf16x8 b) {
    asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "v"(a), "v"(b));
}

template <int E>
__global__ __launch_bounds__(256, 1) void probe(const __bf16* q, const __bf16* k, float* out, int n_tiles, float c2) {
    using KImg = RowImg<__bf16, E>;
    using VImg = ColImg<__bf16, E>;
    constexpr int KS = E / 16, EB = E / 32, KB = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    bf16x8 qf[2][KS];
    for (int z = 0; z < 2; ++z)
        for (int ks = 0; ks < KS; ++ks) qf[z][ks] = *reinterpret_cast<const bf16x8*>(q + (z * 32 + r) * E + 16 * ks + 8 * h);
    f32x16 oacc[2][EB];
    for (int z = 0; z < 2; ++z)
        for (int eb = 0; eb < EB; ++eb)
            for (int i = 0; i < 16; ++i) oacc[z][eb][i] = 0.f;
    float m2[2] = {0.f, 0.f}, lsum[2] = {0.f, 0.f};
    const int vbase = VImg::lane_base(lane);
    for (int t = 0; t < n_tiles; ++t) {
        const char* kimg = smem + (t & 1) * 65536;
        const char* vimg = kimg + 32768;
        f32x16 s[2][KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 a = KImg::read_row_frag(kimg, 32 * kb + r, h, ks);
#pragma unroll
                for (int z = 0; z < 2; ++z) {
                    if (ks == 0) s[z][kb] = mfma_qk0(a, qf[z][ks]);
                    else mfma_qk(s[z][kb], a, qf[z][ks]);
                }
            }
        asm("s_nop 11" : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[1][0]), "+v"(s[1][1]));
#pragma unroll
        for (int z = 0; z < 2; ++z) {
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[z][kb][i]);
            mx = half_swap_max(mx * c2);
            if (__any(mx > m2[z] + 8.f)) {
                const float mn = fmaxf(m2[z], mx);
                const float alpha = fast_exp2(m2[z] - mn);
#pragma unroll
                for (int eb = 0; eb < EB; ++eb) {
                    asm("s_nop 15\n\ts_nop 3" : "+a"(oacc[z][eb]));
#pragma unroll
                    for (int i = 0; i < 16; ++i) oacc[z][eb][i] *= alpha;
                }
                lsum[z] *= alpha;
                m2[z] = mn;
            }
        }
#pragma unroll
        for (int z = 0; z < 2; ++z) {
            bf16x8 pf[2 * KB];
#pragma unroll
            for (int kk = 0; kk < 2 * KB; ++kk) {
                const int kb = kk >> 1, i0 = 8 * (kk & 1);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    s[z][kb][i0 + j] = fast_exp2(__builtin_fmaf(s[z][kb][i0 + j], c2, -m2[z]));
                    lsum[z] += s[z][kb][i0 + j];
                }
                pf[kk] = (kk & 1) ? acc_frag<__bf16, 1>(s[z][kb]) : acc_frag<__bf16, 0>(s[z][kb]);
            }
            asm("s_nop 1" : "+v"(pf[0]), "+v"(pf[1]), "+v"(pf[2]), "+v"(pf[3]));
#pragma unroll
            for (int kk = 0; kk < 2 * KB; ++kk)
#pragma unroll
                for (int eb = 0; eb < EB; ++eb) mfma_pv(oacc[z][eb], VImg::read_col_frag(vimg + vbase, kk, eb), pf[kk]);
        }
        __syncthreads();
    }
    for (int z = 0; z < 2; ++z)
        for (int eb = 0; eb < EB; ++eb) {
            asm("s_nop 15\n\ts_nop 3" : "+a"(oacc[z][eb]));
            for (int i = 0; i < 16; ++i) out[((z * EB + eb) * 16 + i) * 256 + threadIdx.x] = oacc[z][eb][i] / lsum[z];
        }
}
template __global__ void probe<128>(const __bf16*, const __bf16*, float*, int, float);
template __global__ void probe<64>(const __bf16*, const __bf16*, float*, int, float);
