// pingpong.hip -- dev microbenchmark (not part of the library): what two waves per SIMD buy an E = 64 attention forward when one
// wave's MATRIX phase (QK^T of one 64 x 64 tile + PV of the previous one: 32 v_mfma_f32_32x32x16_bf16, 8 ds_read_b128 + 16
// ds_read_b64_tr_b16) runs beside its SIMD partner's VECTOR phase (softmax of a 64 x 64 tile: 32 v_max3, 64 v_fma + 64 v_exp,
// 64 v_add or 8 ones-MFMAs, 32 v_cvt_pk), the two separated by s_barrier.  Prints cycles per (M + V) iteration per wave.
//   mode 0: 4 waves (one per SIMD), M then V serially           mode 1: 8 waves in lockstep (all M, then all V)
//   mode 2: 8 waves, waves 4-7 half an iteration behind         mode 3: as 2, the M-phase wave at s_setprio 1
//   mode 4: as 2, waves 4-7 at s_setprio 1 throughout
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

#define SB() __builtin_amdgcn_sched_barrier(0)

template <int NSUM_MFMA, bool LDS_READS, int VSCALE>
__device__ __forceinline__ void m_phase(f32x16 (&o)[4], f32x16 (&s)[4], f32x16 (&la)[2], bf16x8 (&fr)[4], const bf16x8 (&q)[8], const u32x4 (&pw)[8],
                                        uint32_t lbase) {
    // sums first (operands in registers), then QK^T (S = K Q^T, fresh accumulators), then PV
#pragma unroll
    for (int i = 0; i < NSUM_MFMA; ++i) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(la[i & 1]) : "v"(fr[0]), "v"(__builtin_bit_cast(bf16x8, pw[i & 7])));
        SB();
    }
#pragma unroll
    for (int f = 0; f < 8; ++f) {
        if (LDS_READS) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(f + 2) & 3]) : "v"(lbase), "n"(0) : "memory");
        if (LDS_READS) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
#pragma unroll
        for (int z = 0; z < 2; ++z) {
            if ((f & 3) == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s[2 * (f >> 2) + z]) : "v"(fr[f & 3]), "a"(q[4 * z + (f & 3)]));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s[2 * (f >> 2) + z]) : "v"(fr[f & 3]), "a"(q[4 * z + (f & 3)]));
            SB();
        }
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        if (LDS_READS) {
            u32x2 a, c;
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a) : "v"(lbase), "n"(2048) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(c) : "v"(lbase), "n"(4096) : "memory");
            asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            u32x4 w = {a[0], a[1], c[0], c[1]};
            fr[(g + 2) & 3] = __builtin_bit_cast(bf16x8, w);
        }
#pragma unroll
        for (int z = 0; z < 2; ++z) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o[2 * z + (g & 1)]) : "v"(fr[g & 3]), "v"(__builtin_bit_cast(bf16x8, pw[2 * (g >> 1) + z])));
            SB();
        }
    }
    if (LDS_READS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <bool VSUM, int VSCALE>
__device__ __forceinline__ void v_phase(f32x16 (&s)[4], u32x4 (&pw)[8], float (&l)[4], float& mref, float c2) {
    float mx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("v_max_f32 %0, %1, %2" : "=v"(mx[j]) : "v"(s[j][0]), "v"(s[j][1]));
#pragma unroll
    for (int i = 2; i < 16; i += 2)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mx[j]) : "v"(s[j][i]), "v"(s[j][i + 1]));
    float m0, m1;
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(m0) : "v"(mx[0]), "v"(mx[1]));
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(m1) : "v"(mx[2]), "v"(mx[3]));
    asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mref) : "v"(m0), "v"(m1));
    float nm;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(nm) : "v"(mref), "v"(-c2));
    // groups of 8 elements: 8 fma, 8 exp, then the adds and converts of the previous group
#pragma unroll
    for (int g = 0; g <= 8; ++g) {
        if (g < 8) {
            if (VSCALE) {
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[g >> 1][8 * (g & 1) + e]) : "v"(c2), "v"(nm));
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("v_exp_f32 %0, %0" : "+v"(s[g >> 1][8 * (g & 1) + e]));
        }
        if (g > 0) {
            const int h = g - 1;
            if (VSUM) {
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("v_add_f32 %0, %0, %1" : "+v"(l[e & 3]) : "v"(s[h >> 1][8 * (h & 1) + e]));
            }
#pragma unroll
            for (int e = 0; e < 8; e += 2)
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pw[h][e >> 1]) : "v"(s[h >> 1][8 * (h & 1) + e]), "v"(s[h >> 1][8 * (h & 1) + e + 1]));
        }
    }
}

// one wave-tile here = 64 query rows x 64 keys: the 4 S tiles are [kb][z], elements: 64 per lane
template <int MODE, int NSUM_MFMA, bool LDS_READS, int VSCALE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((float*)lds)[i] = 0.001f * (i & 1023);
    __syncthreads();
    f32x16 o[4], s[4], la[2];
    bf16x8 fr[4], q[8];
    u32x4 pw[8];
    float l[4] = {0, 0, 0, 0}, mref = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) { o[j][i] = 0.f; s[j][i] = 0.001f * (lane + i + j); }
    for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 16; ++i) la[j][i] = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 8; ++i) fr[j][i] = (__bf16)(0.01f * ((lane + i + j) & 15));
    for (int j = 0; j < 8; ++j)
        for (int i = 0; i < 8; ++i) q[j][i] = (__bf16)(0.02f * ((lane - i + j) & 15));
    for (int j = 0; j < 8; ++j) pw[j] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(o[j]));
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(q[j]));
    asm volatile("" : "+a"(la[0]), "+a"(la[1]));
    const uint32_t lbase = (uint32_t)(uintptr_t)lds + lane * 16;
    const float c2 = 0.18033688f;
    constexpr bool VSUM = NSUM_MFMA < 8;
    const bool second = wave >= 4;
    if (MODE == 4 && second) __builtin_amdgcn_s_setprio(1);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE >= 2 && second) asm volatile("s_barrier" ::: "memory");       // half an iteration behind
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) __builtin_amdgcn_s_setprio(1);
        m_phase<NSUM_MFMA, LDS_READS, VSCALE>(o, s, la, fr, q, pw, lbase);
        if (MODE == 3) __builtin_amdgcn_s_setprio(0);
        SB();
        if (MODE != 0) asm volatile("s_barrier" ::: "memory");
        SB();
        v_phase<VSUM, VSCALE>(s, pw, l, mref, c2);
        SB();
        if (MODE != 0) asm volatile("s_barrier" ::: "memory");
        SB();
    }
    if (MODE >= 2 && !second) asm volatile("s_barrier" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = l[0] + l[1] + l[2] + l[3] + mref;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) acc += o[j][i] + s[j][i];
    for (int i = 0; i < 16; ++i) acc += la[0][i] + la[1][i];
    acc += (float)pw[0][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

template <int MODE, int NSUM_MFMA, bool LDS_READS, int VSCALE> void run(const char* name) {
    const int iters = 1000, nblk = 256;
    const int threads = MODE == 0 ? 256 : 512;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * nblk * threads);
    hipMalloc(&cyc, sizeof(unsigned long long) * nblk * threads / 64);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<MODE, NSUM_MFMA, LDS_READS, VSCALE>), dim3(nblk), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nblk * threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double per_iter = (double)h[h.size() / 2] / iters;
    const double per_tile = MODE == 0 ? per_iter : per_iter / 2;           // cycles per SIMD per 64 x 64 wave-tile
    printf("%-34s mode=%d sumMFMA=%d lds=%d scale=%d : %8.1f cycles/iter  -> %7.1f cycles per wave-tile per SIMD\n", name, MODE, NSUM_MFMA,
           (int)LDS_READS, VSCALE, per_iter, per_tile);
    hipFree(out); hipFree(cyc);
}


// ---- E = 128, 32 query rows per wave (NZ = 1): every K / V fragment feeds ONE MFMA.  M: 4 row-sum MFMAs (16x16x32) + 16 PV + 16 QK^T
// MFMAs (32x32x16), 16 ds_read_b128 + 32 ds_read_b64_tr_b16;  V: 32 logits per lane (16 v_max3, 32 v_fma + v_exp, 16 v_cvt_pk)
template <int MODE>
__global__ __launch_bounds__(512) void k128(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((float*)lds)[i] = 0.001f * (i & 1023);
    __syncthreads();
    f32x16 o[4], s[2];
    f32x4v la;
    bf16x8 fr[4], q[8];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) o[j][i] = 0.f;
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 16; ++i) s[j][i] = 0.001f * (lane + i + j);
    for (int i = 0; i < 4; ++i) la[i] = 0.f;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 8; ++i) fr[j][i] = (__bf16)(0.01f * ((lane + i + j) & 15));
    for (int j = 0; j < 8; ++j) for (int i = 0; i < 8; ++i) q[j][i] = (__bf16)(0.02f * ((lane - i + j) & 15));
    const uint32_t lbase = (uint32_t)(uintptr_t)lds + lane * 16;
    const float c2 = 0.12751743f;
    const bool second = wave >= 4;
    float mref = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE >= 2 && second) asm volatile("s_barrier" ::: "memory");
    for (int it = 0; it < iters; ++it) {
        // ---- M
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4v w = {s[i >> 1][8 * (i & 1)], s[i >> 1][8 * (i & 1) + 1], s[i >> 1][8 * (i & 1) + 2], s[i >> 1][8 * (i & 1) + 3]};
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(la) : "v"(fr[0]), "v"(__builtin_bit_cast(bf16x8, w)));
            SB();
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {                       // PV: kk = g / 4, eb = g % 4
            u32x2 a, c;
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a) : "v"(lbase), "n"(2048) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(c) : "v"(lbase), "n"(4096) : "memory");
            asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            u32x4 w4 = {a[0], a[1], c[0], c[1]};
            fr[(g + 2) & 3] = __builtin_bit_cast(bf16x8, w4);
            const int kk = g >> 2;
            f32x4v w = {s[kk >> 1][8 * (kk & 1)], s[kk >> 1][8 * (kk & 1) + 1], s[kk >> 1][8 * (kk & 1) + 2], s[kk >> 1][8 * (kk & 1) + 3]};
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(o[g & 3]) : "v"(fr[g & 3]), "v"(__builtin_bit_cast(bf16x8, w)));
            SB();
        }
#pragma unroll
        for (int f = 0; f < 16; ++f) {                       // QK^T: kb = f / 8, ks = f % 8
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(f + 2) & 3]) : "v"(lbase), "n"(0) : "memory");
            asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
            if ((f & 7) == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s[f >> 3]) : "v"(fr[f & 3]), "v"(q[f & 7]));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s[f >> 3]) : "v"(fr[f & 3]), "v"(q[f & 7]));
            SB();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SB();
        if (MODE != 0) asm volatile("s_barrier" ::: "memory");
        SB();
        // ---- V: 32 logits
        float mx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("v_max_f32 %0, %1, %2" : "=v"(mx[j]) : "v"(s[j >> 1][8 * (j & 1)]), "v"(s[j >> 1][8 * (j & 1) + 1]));
#pragma unroll
        for (int i = 2; i < 8; i += 2)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mx[j]) : "v"(s[j >> 1][8 * (j & 1) + i]), "v"(s[j >> 1][8 * (j & 1) + i + 1]));
        float m0, m1;
        asm volatile("v_max_f32 %0, %1, %2" : "=v"(m0) : "v"(mx[0]), "v"(mx[1]));
        asm volatile("v_max_f32 %0, %1, %2" : "=v"(m1) : "v"(mx[2]), "v"(mx[3]));
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mref) : "v"(m0), "v"(m1));
        float nm;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(nm) : "v"(mref), "v"(-c2));
#pragma unroll
        for (int g = 0; g <= 4; ++g) {
            if (g < 4) {
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[g >> 1][8 * (g & 1) + e]) : "v"(c2), "v"(nm));
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("v_exp_f32 %0, %0" : "+v"(s[g >> 1][8 * (g & 1) + e]));
            }
            if (g > 0) {
                const int h = g - 1;
#pragma unroll
                for (int e = 0; e < 8; e += 2)
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(s[h >> 1][8 * (h & 1) + (e >> 1)]) : "v"(s[h >> 1][8 * (h & 1) + e]), "v"(s[h >> 1][8 * (h & 1) + e + 1]));
            }
        }
        SB();
        if (MODE != 0) asm volatile("s_barrier" ::: "memory");
        SB();
    }
    if (MODE >= 2 && !second) asm volatile("s_barrier" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = mref;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc += o[j][i];
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 16; ++i) acc += s[j][i];
    for (int i = 0; i < 4; ++i) acc += la[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

template <int MODE> void run128(const char* name) {
    const int iters = 1000, nblk = 256;
    const int threads = MODE == 0 ? 256 : 512;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * nblk * threads);
    hipMalloc(&cyc, sizeof(unsigned long long) * nblk * threads / 64);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k128<MODE>), dim3(nblk), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nblk * threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double per_iter = (double)h[h.size() / 2] / iters;
    // one iteration = one (32 rows x 64 keys) wave-tile per wave; a SIMD holds 1 (mode 0) or 2 waves: cycles per 64 x 64 tile-equivalent
    printf("E=128 NZ=1 %-28s mode=%d : %8.1f cycles/iter -> %7.1f cycles per 64x64 tile-equivalent per SIMD (one-wave form: 2770 folded / 3072 exact)\n",
           name, MODE, per_iter, MODE == 0 ? 2 * per_iter : per_iter);
    hipFree(out); hipFree(cyc);
}

int main() {
    run128<0>("1 wave/SIMD serial"); run128<1>("2 waves/SIMD lockstep"); run128<2>("2 waves/SIMD alternating");
    run<0, 0, true, 1>("1 wave/SIMD serial");
    run<0, 8, true, 1>("1 wave/SIMD serial");
    run<1, 0, true, 1>("2 waves/SIMD lockstep");
    run<2, 0, true, 1>("2 waves/SIMD alternating");
    run<2, 4, true, 1>("2 waves/SIMD alternating");
    run<2, 8, true, 1>("2 waves/SIMD alternating");
    run<2, 0, false, 1>("2 waves/SIMD alternating");
    run<2, 0, true, 0>("2 waves/SIMD alternating (folded)");
    run<2, 8, true, 0>("2 waves/SIMD alternating (folded)");
    run<3, 0, true, 1>("alternating, M phase prio 1");
    run<3, 8, true, 1>("alternating, M phase prio 1");
    run<4, 0, true, 1>("alternating, waves 4-7 prio 1");
    run<4, 8, true, 1>("alternating, waves 4-7 prio 1");
    return 0;
}
