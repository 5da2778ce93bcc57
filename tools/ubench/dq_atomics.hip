// dq_atomics.hip -- the floor a 5-product ("single sweep") attention backward would run into on MI355X: its dQ is
// summed across key blocks with fp32 atomics.  This program issues EXACTLY that atomic traffic and nothing else -- one
// workgroup per 256-key block of a (batch, kv-head), which adds one [32 queries x E] fp32 tile per query slice and query
// head to dQ (global_atomic_add_f32, no return; a wave adds 32x32 blocks register by register: two 128-byte row segments
// per wave-instruction, the shape the guide measures at the full chip-wide atomic rate) -- and reports the time.  If that
// time alone is not clearly below what the two-kernel (7-product, atomic-free) backward takes for all of its work, the
// 5-product form cannot win, whatever its schedule.  ("measure the alternative instead of arguing it".)
//   build: hipcc -O3 --offload-arch=gfx950 dq_atomics.hip -o dq_atomics      run: ./dq_atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int E>
__global__ __launch_bounds__(256) void dq_atomics(float* dq, int QL, int KL, int QH, int KH, int causal) {
    const int nkb = (KL + 255) / 256;
    const int kb = blockIdx.x % nkb, bh = blockIdx.x / nkb;         // key block, (batch, kv-head)
    const int b = bh / KH, kvh = bh % KH, rep = QH / KH;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int q_first = causal ? (kb * 256) / 32 : 0;                 // causal: query slices at or below the block's first key
    for (int g = 0; g < rep; ++g) {
        float* base = dq + ((size_t)(b * QH + kvh * rep + g) * QL) * E;
        for (int qs = q_first; qs < QL / 32; ++qs) {
            // the [32 x E] tile = E/32 blocks of 32 x 32; block eb belongs to wave eb % 4
            for (int eb = wave; eb < E / 32; eb += 4) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;    // accumulator register i of lane half h
                    // transposed accumulator (embedding on the registers, query on the lane) would scatter; use the
                    // row-segment form: register i of all lanes = 2 rows x 32 consecutive floats
                    atomicAdd(base + (size_t)(qs * 32 + row) * E + eb * 32 + r, 1.0f);
                }
            }
        }
    }
}

static double run(int E, int L, int QH, int KH, int B, int causal, int iters) {
    float* dq;
    const size_t n = (size_t)B * QH * L * E;
    hipMalloc(&dq, n * sizeof(float));
    hipMemset(dq, 0, n * sizeof(float));
    const int grid = ((L + 255) / 256) * B * KH;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() {
        if (E == 64) hipLaunchKernelGGL(dq_atomics<64>, dim3(grid), dim3(256), 0, 0, dq, L, L, QH, KH, causal);
        else hipLaunchKernelGGL(dq_atomics<128>, dim3(grid), dim3(256), 0, 0, dq, L, L, QH, KH, causal);
    };
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(dq);
    return ms / iters;
}

int main() {
    struct { const char* name; int E, L, QH, KH, B, causal, iters; double bwd_ms; } cfg[] = {
        {"C2  bf16 E64  L4096  H4    B4  non-causal", 64, 4096, 4, 4, 4, 0, 20, 0.0},
        {"C3  bf16 E128 L8192  H32   B8  causal    ", 128, 8192, 32, 32, 8, 1, 3, 0.0},
        {"C4  fp16 E128 L4096  H32/8 B16 (dense)   ", 128, 4096, 32, 8, 16, 0, 3, 0.0},
    };
    for (auto& c : cfg) {
        const double ms = run(c.E, c.L, c.QH, c.KH, c.B, c.causal, c.iters);
        const double nkb = (c.L + 255) / 256;
        const double tiles = c.causal ? (nkb + 1) / 2.0 : nkb;          // key blocks a query slice is added from, on average
        const double bytes = tiles * c.L * (double)c.E * 4.0 * c.QH * c.B;
        printf("%s : dQ atomics alone %8.3f ms  (%.2f GB of fp32 adds -> %.2f TB/s)\n", c.name, ms, bytes / 1e9, bytes / ms / 1e9);
    }
    return 0;
}
