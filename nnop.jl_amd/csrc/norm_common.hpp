// norm_common.hpp -- RMSNorm and LayerNorm, forward and pullback, for gfx950 (HBM-streaming).  One set of templated
// kernels; rms_norm.hip and layer_norm.hip instantiate them (LN = false / true).
//
//   RMSNorm   y = (offset + w) * x * rstd,  rstd = 1/sqrt(mean(x^2) + eps)           src/rms_norm.jl:3-38 (host :117-137)
//             dx = rstd*m - rstd^3 * (dd/N) * x,  m = dy*(w+offset), dd = sum(m*x);  dw = sum_rows dy*x*rstd   :43-115
//   LayerNorm y = (x - mu) * rstd * w + b,  two-pass variance                        src/layer_norm.jl:8-63 (host :150-170)
//             xn = (x-mu)*rstd, wdy = w*dy, c1 = mean(wdy*xn), c2 = mean(wdy);
//             dx = (wdy - (xn*c1 + c2))*rstd;  dw = sum_rows dy*xn;  db = sum_rows dy                         :65-148
//
// Memory: x, y, dy, dx [n][emb] == Julia (emb, n); w, b [emb]; rms / mu / sigma fp32 [n] (the reference's caches for the
// pullback).  Forward: the row lives in registers (row_common.hpp), x is read once (the reference reads it 2-3 times with
// element-strided accesses).  Pullback: the reference gives a workgroup 4 rows and keeps dw/db in LDS with a barrier per
// element step (:101-103), then sums n/4 partial rows with a separate `sum`; here a group of lanes walks rows g, g+P, ..
// and every lane owns fixed columns, so dw/db accumulate in REGISTERS; the <= 1024 partial rows are folded by one small
// second kernel in a fixed order (deterministic, no atomics).
// Bound: HBM -- forward 2*emb*sizeof(T) per row; pullback 3*emb*sizeof(T) per row (+ the partials, <= 8 %).
#pragma once
#include "row_common.hpp"
#include "fa_launch.hpp"

namespace nnop {

struct NormParams {
    // forward: out = y, a = x;  pullback: out = dx, a = dy, x = x
    void* out;
    const void* a;
    const void* x;
    const void* w;
    const void* b;            // LayerNorm forward only
    float* stat0;             // rms (RMSNorm) / mu (LayerNorm): written by forward, read by pullback
    float* stat1;             // sigma (LayerNorm)
    float* part_w;            // pullback: [n_parts][emb] fp32 partial dw
    float* part_b;            // pullback, LayerNorm: partial db
    int emb;
    long long n;
    float offset, eps, inv_emb;
    int n_groups;             // pullback: row stride of a group
};

// VEC consecutive elements of a [emb] vector of type W starting at e, as fp32
template <typename W, int VEC> NNOP_DEV void load_vec(const W* __restrict__ w, int e, float (&out)[VEC]) {
    constexpr int PER = 16 / (int)sizeof(W);                      // elements per 16-byte load
    typedef W wv __attribute__((ext_vector_type(PER)));
    static_assert(VEC % PER == 0, "chunk must be whole 16-byte loads of W");
#pragma unroll
    for (int k = 0; k < VEC / PER; ++k) {
        const wv t = *reinterpret_cast<const wv*>(w + e + k * PER);
#pragma unroll
        for (int i = 0; i < PER; ++i) out[k * PER + i] = to_f32(t[i]);
    }
}

// ---- forward ---------------------------------------------------------------------------------------------------
template <typename T, typename W, int G, int C, bool LN>
__global__ __launch_bounds__(G > 256 ? G : 256, G > 256 ? 1 : 2) void norm_fwd_kernel(const NormParams p) {
    constexpr int RPB = G >= 256 ? 1 : 256 / G;
    constexpr int VEC = RowRegs<T, G, C>::VEC;
    __shared__ float slots[2][G > 64 ? G / 64 : 1];
    const int lane = threadIdx.x % G;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / G;
    if (G == 64 && row >= p.n) return;
    const size_t off = (size_t)row * p.emb;
    RowRegs<T, G, C> r;
    r.load((const T*)p.a + off, p.emb, lane, 0.f);
    float mu = 0.f;
    if constexpr (LN) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < C; ++j)
#pragma unroll
            for (int i = 0; i < VEC; ++i) s += r.get(j, i);
        mu = group_allreduce<G>(s, SumOp{}, slots[0]) * p.inv_emb;       // src/layer_norm.jl:22-35
        repack(r);
    }
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < C; ++j)
        if (!LN || r.in_row(j, lane, p.emb)) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) { const float xc = r.get(j, i) - mu; q = __builtin_fmaf(xc, xc, q); }
        }
    q = group_allreduce<G>(q, SumOp{}, slots[1]) * p.inv_emb;            // rms_norm.jl:16-25, layer_norm.jl:38-48
    const float rstd = 1.0f / sqrtf(q + p.eps);                          // rms_norm.jl:27, layer_norm.jl:50
    if (lane == 0) {
        if constexpr (LN) { p.stat0[row] = mu; p.stat1[row] = rstd; }
        else p.stat0[row] = rstd;
    }
    repack(r);
#pragma unroll
    for (int j = 0; j < C; ++j) {
        const int e = (j * G + lane) * VEC;
        if (e < p.emb) {
            float wv[VEC], bv[VEC];
            load_vec<W, VEC>((const W*)p.w, e, wv);
            if constexpr (LN) load_vec<W, VEC>((const W*)p.b, e, bv);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                if constexpr (LN) r.set(j, i, (r.get(j, i) - mu) * rstd * wv[i] + bv[i]);       // layer_norm.jl:58-59
                else r.set(j, i, (p.offset + wv[i]) * r.get(j, i) * rstd);                      // rms_norm.jl:34
            }
        }
    }
    r.store((T*)p.out + off, p.emb, lane);
}

// any emb: workgroup per row, element accesses, the reference's pass structure
template <typename T, typename W, bool LN>
__global__ __launch_bounds__(256) void norm_fwd_generic_kernel(const NormParams p) {
    __shared__ float slots[2][4];
    const long long row = blockIdx.x;
    const T* __restrict__ x = (const T*)p.a + (size_t)row * p.emb;
    T* __restrict__ y = (T*)p.out + (size_t)row * p.emb;
    const W* __restrict__ w = (const W*)p.w;
    const W* __restrict__ b = (const W*)p.b;
    float mu = 0.f;
    if constexpr (LN) {
        float s = 0.f;
        for (int e = threadIdx.x; e < p.emb; e += 256) s += to_f32(x[e]);
        mu = group_allreduce<256>(s, SumOp{}, slots[0]) * p.inv_emb;
    }
    float q = 0.f;
    for (int e = threadIdx.x; e < p.emb; e += 256) { const float xc = to_f32(x[e]) - mu; q = __builtin_fmaf(xc, xc, q); }
    q = group_allreduce<256>(q, SumOp{}, slots[1]) * p.inv_emb;
    const float rstd = 1.0f / sqrtf(q + p.eps);
    if (threadIdx.x == 0) {
        if constexpr (LN) { p.stat0[row] = mu; p.stat1[row] = rstd; }
        else p.stat0[row] = rstd;
    }
    for (int e = threadIdx.x; e < p.emb; e += 256) {
        if constexpr (LN) y[e] = from_f32<T>((to_f32(x[e]) - mu) * rstd * to_f32(w[e]) + to_f32(b[e]));
        else y[e] = from_f32<T>((p.offset + to_f32(w[e])) * to_f32(x[e]) * rstd);
    }
}

// ---- pullback --------------------------------------------------------------------------------------------------
template <typename T, typename W, int G, int C, bool LN>
__global__ __launch_bounds__(G > 256 ? G : 256, G > 256 ? 1 : 2) void norm_bwd_kernel(const NormParams p) {
    constexpr int RPB = G >= 256 ? 1 : 256 / G;
    constexpr int VEC = RowRegs<T, G, C>::VEC;
    __shared__ float slots_f[2][G > 64 ? G / 64 : 1];
    __shared__ Sum2 slots_2[2][G > 64 ? G / 64 : 1];
    __shared__ float comb[RPB > 1 ? (LN ? 2 : 1) * 64 * C * VEC : 1];   // wave-per-row shapes: fold the 4 waves
    const int lane = threadIdx.x % G;
    const int sub = threadIdx.x / G;                                     // group within the workgroup
    const long long g0 = (long long)blockIdx.x * RPB + sub;

    float wv[C][VEC], dw[C][VEC], db[LN ? C : 1][VEC];
#pragma unroll
    for (int j = 0; j < C; ++j) {
        const int e = (j * G + lane) * VEC;
#pragma unroll
        for (int i = 0; i < VEC; ++i) { wv[j][i] = 0.f; dw[j][i] = 0.f; if constexpr (LN) db[j][i] = 0.f; }
        if (e < p.emb) {
            load_vec<W, VEC>((const W*)p.w, e, wv[j]);
            if constexpr (!LN)
#pragma unroll
                for (int i = 0; i < VEC; ++i) wv[j][i] += p.offset;      // w + offset, rms_norm.jl:78,95
        }
    }
    int it = 0;
    // the next row's dy / x are requested before the current row's reduce-and-store chain starts: a group otherwise
    // has nothing in flight while it waits on its own reduction
    RowRegs<T, G, C> d, x, d_nx, x_nx;
    if (g0 < p.n) {
        d.load((const T*)p.a + (size_t)g0 * p.emb, p.emb, lane, 0.f);
        x.load((const T*)p.x + (size_t)g0 * p.emb, p.emb, lane, 0.f);
    }
    for (long long row = g0; row < p.n; row += p.n_groups, ++it) {
        const size_t off = (size_t)row * p.emb;
        const long long nx = row + p.n_groups;
        if (nx < p.n) {
            d_nx.load((const T*)p.a + (size_t)nx * p.emb, p.emb, lane, 0.f);
            x_nx.load((const T*)p.x + (size_t)nx * p.emb, p.emb, lane, 0.f);
        }
        if constexpr (!LN) {
            const float rstd = p.stat0[row];
            float dd = 0.f;
#pragma unroll
            for (int j = 0; j < C; ++j)
#pragma unroll
                for (int i = 0; i < VEC; ++i) dd = __builtin_fmaf(d.get(j, i) * wv[j][i], x.get(j, i), dd);     // :72-81
            dd = group_allreduce<G>(dd, SumOp{}, slots_f[it & 1]);
            const float k = -p.inv_emb * rstd * rstd * dd * rstd;      // rstd * (-1/N * rstd^2 * dd), :96
#pragma unroll
            for (int j = 0; j < C; ++j)
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float de = d.get(j, i), xe = x.get(j, i);
                    dw[j][i] = __builtin_fmaf(de, xe * rstd, dw[j][i]);                                          // :98,101
                    d.set(j, i, __builtin_fmaf(de * wv[j][i], rstd, k * xe));                                    // :95-96
                }
        } else {
            const float mu = p.stat0[row], rstd = p.stat1[row];
            const float nmr = -mu * rstd;                          // xn = (x - mu)*rstd = fma(x, rstd, -mu*rstd)
            Sum2 c{0.f, 0.f};
#pragma unroll
            for (int j = 0; j < C; ++j)
                if (x.in_row(j, lane, p.emb)) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        const float xn = __builtin_fmaf(x.get(j, i), rstd, nmr), wdy = wv[j][i] * d.get(j, i);  // :102-106
                        c.a = __builtin_fmaf(wdy, xn, c.a);
                        c.b += wdy;
                    }
                }
            c = group_allreduce<G>(c, 0, slots_2[it & 1]);
            // dx = (wdy - (xn*c1 + c2))*rstd = fma(wdy, rstd, fma(xn, -c1*rstd, -c2*rstd))                     // :110-111,128
            const float k1 = -c.a * p.inv_emb * rstd, k2 = -c.b * p.inv_emb * rstd;
#pragma unroll
            for (int j = 0; j < C; ++j)
                if (x.in_row(j, lane, p.emb)) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        const float de = d.get(j, i);
                        const float xn = __builtin_fmaf(x.get(j, i), rstd, nmr), wdy = wv[j][i] * de;
                        dw[j][i] = __builtin_fmaf(de, xn, dw[j][i]);                                             // :129,132
                        db[j][i] += de;                                                                          // :133
                        d.set(j, i, __builtin_fmaf(wdy, rstd, __builtin_fmaf(xn, k1, k2)));
                    }
                }
        }
        d.store((T*)p.out + off, p.emb, lane);
        d = d_nx; x = x_nx;
    }
    // fold the workgroup's groups (wave-per-row shapes) in a fixed order, then one partial row per workgroup
    if constexpr (RPB > 1) {
        constexpr int STR = 64 * C * VEC;
        for (int s = 0; s < RPB; ++s) {
            if (sub == s) {
#pragma unroll
                for (int j = 0; j < C; ++j)
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        const int e = (j * G + lane) * VEC + i;
                        if (s == 0) { comb[e] = dw[j][i]; if constexpr (LN) comb[STR + e] = db[j][i]; }
                        else { comb[e] += dw[j][i]; if constexpr (LN) comb[STR + e] += db[j][i]; }
                    }
            }
            __syncthreads();
        }
        if (sub == 0) {
#pragma unroll
            for (int j = 0; j < C; ++j)
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const int e = (j * G + lane) * VEC + i;
                    dw[j][i] = comb[e];
                    if constexpr (LN) db[j][i] = comb[STR + e];
                }
        }
    }
    if (sub == 0) {
        float* pw = p.part_w + (size_t)blockIdx.x * p.emb;
        float* pb = LN ? p.part_b + (size_t)blockIdx.x * p.emb : nullptr;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int e = (j * G + lane) * VEC;
            if (e < p.emb) {
#pragma unroll
                for (int i = 0; i < VEC; i += 4) {
                    *reinterpret_cast<f32x4*>(pw + e + i) = f32x4{dw[j][i], dw[j][i + 1], dw[j][i + 2], dw[j][i + 3]};
                    if constexpr (LN)
                        *reinterpret_cast<f32x4*>(pb + e + i) = f32x4{db[j][i], db[j][i + 1], db[j][i + 2], db[j][i + 3]};
                }
            }
        }
    }
}

// any emb: a workgroup walks rows blockIdx.x, +gridDim.x, ..; a lane owns columns tid, tid+256, .. and accumulates
// its partial sums directly in its workgroup's partial row (same lane, same address: no race)
template <typename T, typename W, bool LN>
__global__ __launch_bounds__(256) void norm_bwd_generic_kernel(const NormParams p) {
    __shared__ float slots_f[2][4];
    __shared__ Sum2 slots_2[2][4];
    const T* __restrict__ dy = (const T*)p.a;
    const T* __restrict__ xx = (const T*)p.x;
    const W* __restrict__ w = (const W*)p.w;
    T* __restrict__ dx = (T*)p.out;
    float* pw = p.part_w + (size_t)blockIdx.x * p.emb;
    float* pb = LN ? p.part_b + (size_t)blockIdx.x * p.emb : nullptr;
    for (int e = threadIdx.x; e < p.emb; e += 256) { pw[e] = 0.f; if (LN) pb[e] = 0.f; }
    int it = 0;
    for (long long row = blockIdx.x; row < p.n; row += gridDim.x, ++it) {
        const size_t off = (size_t)row * p.emb;
        if constexpr (!LN) {
            const float rstd = p.stat0[row];
            float dd = 0.f;
            for (int e = threadIdx.x; e < p.emb; e += 256)
                dd = __builtin_fmaf(to_f32(dy[off + e]) * (to_f32(w[e]) + p.offset), to_f32(xx[off + e]), dd);
            dd = group_allreduce<256>(dd, SumOp{}, slots_f[it & 1]);
            const float k = -p.inv_emb * rstd * rstd * dd;
            for (int e = threadIdx.x; e < p.emb; e += 256) {
                const float de = to_f32(dy[off + e]), xe = to_f32(xx[off + e]);
                pw[e] = __builtin_fmaf(de, xe * rstd, pw[e]);
                dx[off + e] = from_f32<T>(rstd * (de * (to_f32(w[e]) + p.offset)) + rstd * (k * xe));
            }
        } else {
            const float mu = p.stat0[row], rstd = p.stat1[row];
            Sum2 c{0.f, 0.f};
            for (int e = threadIdx.x; e < p.emb; e += 256) {
                const float xn = (to_f32(xx[off + e]) - mu) * rstd, wdy = to_f32(w[e]) * to_f32(dy[off + e]);
                c.a = __builtin_fmaf(wdy, xn, c.a);
                c.b += wdy;
            }
            c = group_allreduce<256>(c, 0, slots_2[it & 1]);
            const float c1 = c.a * p.inv_emb, c2 = c.b * p.inv_emb;
            for (int e = threadIdx.x; e < p.emb; e += 256) {
                const float de = to_f32(dy[off + e]);
                const float xn = (to_f32(xx[off + e]) - mu) * rstd, wdy = to_f32(w[e]) * de;
                pw[e] = __builtin_fmaf(de, xn, pw[e]);
                pb[e] += de;
                dx[off + e] = from_f32<T>((wdy - (xn * c1 + c2)) * rstd);
            }
        }
    }
}

// dw[e] (blockIdx.y = 0) / db[e] (blockIdx.y = 1) = sum over the partial rows in a fixed order: a workgroup takes 32
// columns x 32 slices of rows (four independent accumulators per lane keep loads in flight), slices meet through LDS
template <typename O>
__global__ __launch_bounds__(1024) void norm_fold_kernel(O* __restrict__ out_w, O* __restrict__ out_b,
                                                          const float* __restrict__ part_w,
                                                          const float* __restrict__ part_b, int n_parts, int emb) {
    __shared__ float sm[32][33];
    const float* __restrict__ part = blockIdx.y ? part_b : part_w;
    O* __restrict__ out = blockIdx.y ? out_b : out_w;
    const int c = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (col < emb) {
        int g = sl;
        for (; g + 96 < n_parts; g += 128) {
            s0 += part[(size_t)g * emb + col];
            s1 += part[(size_t)(g + 32) * emb + col];
            s2 += part[(size_t)(g + 64) * emb + col];
            s3 += part[(size_t)(g + 96) * emb + col];
        }
        for (; g < n_parts; g += 32) s0 += part[(size_t)g * emb + col];
    }
    sm[sl][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && col < emb) {
        float t = sm[0][c];
#pragma unroll
        for (int k = 1; k < 32; ++k) t += sm[k][c];
        out[col] = from_f32<O>(t);
    }
}

// Number of partial rows (= persistent pullback workgroups) for n rows of emb columns.  Measured on MI355X
// (tools/norm_cap.sh): more workgroups hide the per-row reduce latency, but every partial row is emb*4 bytes written
// and read again by the fold -- best is ~2 Mi partial elements: 256 rows at emb >= 8192, 512 at 4096, 1024 below 2048.
// Never more than n/4 (every group gets >= 4 rows when n allows).
constexpr int kNormBwdCap = 1024;                                  // max partial rows (workspace sizing)
static inline long long norm_bwd_max_parts(long long n) {
    const long long q = (n + 3) / 4;
    return q < 1 ? 1 : (q > kNormBwdCap ? kNormBwdCap : q);
}
static inline int norm_bwd_parts(long long n, int emb, int rpb) {
    long long cap = (2LL << 20) / emb;
    cap = cap < 256 ? 256 : (cap > kNormBwdCap ? kNormBwdCap : cap);
    if (const int t = tune_get(kTuneNormBwdCap); t > 0) cap = t;
    long long wgs = (n + (long long)rpb * 4 - 1) / ((long long)rpb * 4);
    if (wgs > cap) wgs = cap;
    if (wgs > norm_bwd_max_parts(n)) wgs = norm_bwd_max_parts(n);
    return (int)(wgs < 1 ? 1 : wgs);
}
// workspace: dw partials, then db partials for LayerNorm
static inline size_t norm_bwd_ws_bytes(const nnop_norm_desc& d, bool ln) {
    return (size_t)norm_bwd_max_parts(d.n) * (size_t)d.emb * sizeof(float) * (ln ? 2 : 1);
}

template <typename T, typename W, bool LN> static int launch_norm_fwd_t(NormParams p, hipStream_t s) {
    const bool aligned = (((uintptr_t)p.out | (uintptr_t)p.a | (uintptr_t)p.w | (uintptr_t)p.b) & 15) == 0;
    bool done = false;
    if (aligned)
        done = dispatch_row_shape<T, 1>(p.emb, [&](auto shape) {
            constexpr int G = decltype(shape)::G, C = decltype(shape)::C;
            constexpr int RPB = G >= 256 ? 1 : 256 / G, NT = G > 256 ? G : 256;
            const long long grid = (p.n + RPB - 1) / RPB;
            hipLaunchKernelGGL((norm_fwd_kernel<T, W, G, C, LN>), dim3((unsigned)grid), dim3(NT), 0, s, p);
        });
    if (!done) hipLaunchKernelGGL((norm_fwd_generic_kernel<T, W, LN>), dim3((unsigned)p.n), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? NNOP_OK : NNOP_ERR_HIP;
}

// O: element type of dw / db (fp32 for RMSNorm, W for LayerNorm)
template <typename T, typename W, typename O, bool LN>
static int launch_norm_bwd_t(NormParams p, void* dw, void* db, hipStream_t s) {
    const bool aligned = (((uintptr_t)p.out | (uintptr_t)p.a | (uintptr_t)p.x | (uintptr_t)p.w | (uintptr_t)p.part_w |
                           (uintptr_t)p.part_b) & 15) == 0;
    int parts = 0;
    bool done = false;
    if (aligned)
        done = dispatch_row_shape_narrow<T>(p.emb, [&](auto shape) {
            constexpr int G = decltype(shape)::G, C = decltype(shape)::C;
            constexpr int RPB = G >= 256 ? 1 : 256 / G, NT = G > 256 ? G : 256;
            parts = norm_bwd_parts(p.n, p.emb, RPB);
            p.n_groups = parts * RPB;
            hipLaunchKernelGGL((norm_bwd_kernel<T, W, G, C, LN>), dim3(parts), dim3(NT), 0, s, p);
        });
    if (!done) {
        parts = norm_bwd_parts(p.n, p.emb, 1);
        hipLaunchKernelGGL((norm_bwd_generic_kernel<T, W, LN>), dim3(parts), dim3(256), 0, s, p);
    }
    const unsigned fg = (unsigned)((p.emb + 31) / 32);
    hipLaunchKernelGGL((norm_fold_kernel<O>), dim3(fg, LN ? 2 : 1), dim3(1024), 0, s, (O*)dw, (O*)db,
                       (const float*)p.part_w, (const float*)p.part_b, parts, p.emb);
    return hipGetLastError() == hipSuccess ? NNOP_OK : NNOP_ERR_HIP;
}

}  // namespace nnop
