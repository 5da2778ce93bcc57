// dev microbenchmark: what the dQ hand-off of a 5-product backward would cost on its own.  One workgroup per 256-key block of a
// (batch, q-head) column (as fa_bwd_w64's dK/dV pass), walking the column's 64-row q tiles (causal: from the block's diagonal down) and
// adding one 64 x E fp32 tile into dQ per (block, q tile) -- after the in-LDS reduction over the workgroup's 4 waves that the scheme
// assumes -- with nothing else in the loop.  Compare with 5/7 of the measured 7-product backward (profiles/r04/bench_c3.json).
//   build: hipcc -O3 --offload-arch=gfx950 -o atomic_dq atomic_dq.hip        usage: atomic_dq [E L H B causal]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int SCOPE, bool XCD_COLUMNS>
__global__ __launch_bounds__(256) void add_tiles(float* dq, int E, int L, int n_kvb, int n_cols, int causal) {
    // XCD_COLUMNS: workgroup ids go round-robin over the 8 XCDs; give every XCD whole columns, so that all adds to one dQ row meet in one L2
    int col, kvb;
    if (XCD_COLUMNS) {
        const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
        col = x * (n_cols / 8) + i / n_kvb;
        kvb = i % n_kvb;
    } else {
        col = blockIdx.x / n_kvb;
        kvb = blockIdx.x % n_kvb;
    }
    float* base = dq + (size_t)col * L * E;
    const int t0 = causal ? kvb * 4 : 0, nt = L / 64;
    const int per_thread = 64 * E / 256;
    for (int t = t0; t < nt; ++t) {
        float* tile = base + (size_t)t * 64 * E;
#pragma unroll 8
        for (int j = 0; j < per_thread; ++j) {
            float* p = tile + j * 256 + threadIdx.x;           // lane-contiguous: one 256-byte line pair per wave and add
            if (SCOPE == 0) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

int main(int argc, char** argv) {
    const int E = argc > 1 ? atoi(argv[1]) : 128, L = argc > 2 ? atoi(argv[2]) : 8192, H = argc > 3 ? atoi(argv[3]) : 32, B = argc > 4 ? atoi(argv[4]) : 8;
    const int causal = argc > 5 ? atoi(argv[5]) : 1;
    const int n_cols = H * B, n_kvb = L / 256;
    float* dq;
    const size_t bytes = (size_t)n_cols * L * E * 4;
    hipMalloc(&dq, bytes);
    hipMemset(dq, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double tiles = 0;
    for (int k = 0; k < n_kvb; ++k) tiles += (double)(L / 64 - (causal ? 4 * k : 0));
    tiles *= n_cols;
    const double gb = tiles * 64 * E * 4 / 1e9;
    auto run = [&](const char* name, auto kern) {
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kern, dim3(n_cols * n_kvb), dim3(256), 0, 0, dq, E, L, n_kvb, n_cols, causal);
        hipEventRecord(e0);
        const int it = 3;
        for (int r = 0; r < it; ++r) hipLaunchKernelGGL(kern, dim3(n_cols * n_kvb), dim3(256), 0, 0, dq, E, L, n_kvb, n_cols, causal);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
        printf("E=%d L=%d H=%d B=%d causal=%d  %-44s %9.3f ms  %7.1f GB of fp32 adds  %7.2f TB/s\n", E, L, H, B, causal, name, ms, gb, gb / ms);
        fflush(stdout);
    };
    run("agent scope (sc1: beyond the L2)", add_tiles<0, false>);
    run("agent scope, columns per XCD", add_tiles<0, true>);
    run("workgroup scope (in the L2), columns per XCD", add_tiles<1, true>);
    // check of the last variant's premise: every element of dQ received exactly (number of blocks that reach its tile) adds
    std::vector<float> h((size_t)L * E);
    hipMemcpy(h.data(), dq + (size_t)(n_cols - 1) * L * E, h.size() * 4, hipMemcpyDeviceToHost);
    const int runs = 3 * 5;
    int bad = 0;
    for (int t = 0; t < L / 64; ++t) {
        const float want = (float)runs * (causal ? (t / 4 + 1) : n_kvb);
        for (int i = 0; i < 64 * E; ++i) if (h[(size_t)t * 64 * E + i] != want) { ++bad; break; }
    }
    printf("sum check over the three variants (last column): %s\n", bad ? "MISMATCH" : "exact");
    return 0;
}
