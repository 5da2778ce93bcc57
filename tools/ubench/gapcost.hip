// gapcost.hip -- what one MFMA gap costs on gfx950 with ONE wave per SIMD (dev microbenchmark, not part of the library).
// Every wave runs ITERS x { MFMA ; F x <op> ; MFMA ; F x <op> } with independent operands; printed: cycles per gap (s_memtime)
// for each op and F.  The slope over F past the plateau is the op's issue cost; the plateau is what hides behind the MFMA.
// The model of tools/w64_gaps.py (MFMA 8, v_exp 8, VALU 4, DS 4, scalar 4, waits 0) is calibrated against this.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

enum { FMA, EXP, CVT, MAX3, SNOP, SMOV, WAITL, WAITV, DSB128, DSTR, DMA, BARRIER, PERM, MIX, NOPS };
static const char* kName[NOPS] = {"v_fma_f32", "v_exp_f32", "v_cvt_pk_bf16", "v_max3_f32", "s_nop 0", "s_mov_b32", "s_waitcnt lgkmcnt(0)", "s_waitcnt vmcnt(0)",
                                  "ds_read_b128", "ds_read_b64_tr_b16", "m0 + global_load_lds_dwordx4", "s_barrier", "v_permlane32_swap",
                                  "mix: 2 exp + cvt + ds_read_b128 + wait"};

template <int OP, int F>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, const char* src, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc0 = {0}, acc1 = {0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (lane + i)); b[i] = (__bf16)(0.02f * (lane - i)); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 0.001f * (lane + i);
    u32x4 ld[4] = {};
    u32x2 lt[4] = {};
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((float*)lds)[i] = i;
    __syncthreads();
    const uint32_t laddr = (uint32_t)(uintptr_t)lds + lane * 16 + wave * 4096;
    const uint32_t ldst = (uint32_t)(uintptr_t)lds + 32768 + wave * 4096;
    const uint32_t voff = lane * 16 + wave * 1024;
    auto fill = [&](auto) {
#pragma unroll
        for (int n = 0; n < F; ++n) {
            float& x = v[n % 16];
            float& y = v[(n + 5) % 16];
            if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
            if (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
            if (OP == CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(y));
            if (OP == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
            if (OP == SNOP) asm volatile("s_nop 0");
            if (OP == SMOV) { uint32_t s; asm volatile("s_mov_b32 %0, 17" : "=s"(s)); }
            if (OP == WAITL) asm volatile("s_waitcnt lgkmcnt(0)");
            if (OP == WAITV) asm volatile("s_waitcnt vmcnt(0)");
            if (OP == DSB128) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[n % 4]) : "v"(laddr + (n & 3) * 1024) : "memory");
            if (OP == DSTR) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lt[n % 4]) : "v"(laddr + (n & 3) * 1024) : "memory");
            if (OP == DMA) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(src), "s"(ldst + (n & 3) * 1024) : "memory");
            if (OP == BARRIER) asm volatile("s_barrier");
            if (OP == PERM) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
            if (OP == MIX) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(x));
                asm volatile("v_exp_f32 %0, %0" : "+v"(y));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v[(n + 9) % 16]) : "v"(v[(n + 10) % 16]), "v"(v[(n + 11) % 16]));
                asm volatile("ds_read_b128 %0, %1" : "=v"(ld[n % 4]) : "v"(laddr + (n & 3) * 1024) : "memory");
                asm volatile("s_waitcnt lgkmcnt(2)");
            }
        }
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
        fill(0);
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(b));
        fill(0);
        if (OP == DSB128 || OP == DSTR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (OP == DMA) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i] + acc0[i] + acc1[i];
    for (int i = 0; i < 4; ++i) s += (float)ld[i][0] + (float)lt[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

static float* g_out; static unsigned long long* g_cyc; static char* g_src;

template <int OP, int F> double run() {
    const int iters = 4000, nblk = 256;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<OP, F>), dim3(nblk), dim3(256), 0, 0, g_out, g_cyc, g_src, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nblk * 4);
    hipMemcpy(h.data(), g_cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    return (double)h[h.size() / 2] / iters / 2.0;
}
template <int OP> void sweep() {
    printf("%-40s F=0 %6.1f  1 %6.1f  2 %6.1f  3 %6.1f  4 %6.1f  6 %6.1f  8 %6.1f   cycles per MFMA gap\n", kName[OP], run<OP, 0>(), run<OP, 1>(), run<OP, 2>(),
           run<OP, 3>(), run<OP, 4>(), run<OP, 6>(), run<OP, 8>());
    fflush(stdout);
}
int main() {
    hipMalloc(&g_out, sizeof(float) * 256 * 256);
    hipMalloc(&g_cyc, 8 * 256 * 4);
    hipMalloc(&g_src, 1 << 20);
    hipMemset(g_src, 0, 1 << 20);
    sweep<FMA>(); sweep<EXP>(); sweep<CVT>(); sweep<MAX3>(); sweep<SNOP>(); sweep<SMOV>(); sweep<WAITL>(); sweep<WAITV>(); sweep<DSB128>(); sweep<DSTR>();
    sweep<DMA>(); sweep<BARRIER>(); sweep<PERM>(); sweep<MIX>();
    return 0;
}
