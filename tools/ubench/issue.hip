// issue.hip -- what a SIMD can issue beside MFMAs on gfx950 (dev microbenchmark, not part of the library).
// Each wave runs ITERS x { 1 v_mfma_f32_32x32x16_bf16 ; N x <op> } with independent operands, W waves per SIMD.
// Prints cycles per iteration per SIMD (s_memtime), for op in {v_fma_f32, v_exp_f32, v_pk_fma_f32, v_max3_f32,
// v_cvt_pk_bf16_f32, ds_read_b128} and N = 0..; the slope over N is the op's issue cost, the plateau its "free" slots.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int OP, int N, bool MFMA>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x;
    f32x16 acc0 = {0}, acc1 = {0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (lane + i)); b[i] = (__bf16)(0.02f * (lane - i)); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 0.001f * (lane + i);
    u32x4 ld = {0, 0, 0, 0};
    for (int i = lane; i < 16384; i += blockDim.x) ((float*)lds)[i] = i;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MFMA) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < N; ++n) {
            float& x = v[n % 16];
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(v[(n + 1) % 16]));
            if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
            if (OP == 2) { f32x2& y = *(f32x2*)&v[2 * (n % 8)]; asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(y)); }
            if (OP == 3) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(v[(n + 1) % 16]));
            if (OP == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(v[(n + 1) % 16]));
            if (OP == 5) { asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"((lane & 63) * 16 + (n & 3) * 1024) : "memory"); }
        }
        if (OP == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i] + acc0[i] + acc1[i];
    s += (float)ld[0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP, int N, bool MFMA> void run(int waves_per_simd, const char* name) {
    const int iters = 2000, nblk = 256;                         // one workgroup per CU
    const int threads = 256 * waves_per_simd;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * nblk * threads);
    hipMalloc(&cyc, sizeof(unsigned long long) * nblk * threads / 64);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<OP, N, MFMA>), dim3(nblk), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nblk * threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    // s_memtime counts at 100 MHz-independent "shader clock"? report raw counts per iteration (median wave)
    printf("%-10s mfma=%d N=%2d waves/simd=%d : %8.2f memtime-ticks/iter (2 MFMA + N ops per wave)\n", name, (int)MFMA, N,
           waves_per_simd, (double)h[h.size() / 2] / iters);
    hipFree(out); hipFree(cyc);
}

template <int OP> void sweep(const char* name) {
    for (int w : {1, 2, 4}) {
        run<OP, 0, true>(w, name); run<OP, 4, true>(w, name); run<OP, 8, true>(w, name); run<OP, 16, true>(w, name);
        run<OP, 32, true>(w, name); run<OP, 16, false>(w, name); run<OP, 32, false>(w, name);
    }
}

int main() {
    sweep<0>("fma"); sweep<1>("exp"); sweep<2>("pk_fma"); sweep<3>("max3"); sweep<4>("cvt_pk"); sweep<5>("ds_b128");
    return 0;
}
