// fa_generic.hpp -- attention for the embedding dims the MFMA kernels do not cover.
//
// The reference accepts any power-of-two embedding dim whose tiles fit shared memory (src/attention.jl:143,193-205); the tiled
// kernels of this library exist for E in {16, 32, 64, 128}.  So that the boundary never rejects what the reference accepts,
// every other power of two from 1 to 512 runs here: plain HIP, one wave per row, fp32 arithmetic, no matrix cores, the same
// contract (causal top-left aligned, key padding, grouped-query heads, pair bias and dpair, ragged lengths, residuals ms / ls
// per src/attention.jl:128-129, a row without a visible key gives NaN in o and zero gradients).  Correctness path, not a
// fast one (1-7 TFLOP/s).  E = 256 does NOT come here in the 16-bit types (the one-wave-per-SIMD kernels run it, DESIGN.md
// section 6) nor for the fp32 FORWARD (the 32-row tiled kernel); the fp32 E = 256 backward does.
//
//   forward  : wave = one query row; a lane scores one key of the 64-key tile (dot product over E from its own K row, the Q row
//              broadcast from LDS), online softmax with wave reductions, then O += p_k * V[k] with lane = embedding column.
//   backward : preprocess (lse, delta per row) -> dQ kernel (wave = query row, lanes = keys, writes dpair) -> dK/dV kernel
//              (wave = key row of a kv head, loops over the group's query heads, lanes = queries): no atomics, deterministic.
#pragma once
#include "fa_bwd.hpp"
#include "fa_fwd.hpp"

namespace nnop {

constexpr int kGenericMaxE = 512;
inline bool emb_generic(int e) { return e >= 1 && e <= kGenericMaxE && (e & (e - 1)) == 0 && !(e == 16 || e == 32 || e == 64 || e == 128); }

NNOP_DEV float wave_max64(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
NNOP_DEV float wave_sum64(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
NNOP_DEV float lane_bcast(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
template <typename T> NNOP_DEV float dot_lds(const float* a, const T* __restrict__ b, int E) {
    float s = 0.f;
    for (int e = 0; e < E; ++e) s += a[e] * to_f32(b[e]);
    return s;
}

template <typename T>
__global__ __launch_bounds__(256) void fa_fwd_generic_kernel(const FwdParams p, int E, long long n_rows) {
    __shared__ float qs_all[4][kGenericMaxE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + wave;
    if (row >= n_rows) return;                                    // wave-uniform; no workgroup barrier below
    const int qi = (int)(row % p.QL);
    const int bh = (int)(row / p.QL);
    const int b = bh / p.QH, qh = bh - b * p.QH;
    const int kvh = qh / (p.QH / p.KH);
    float* qs = qs_all[wave];
    const T* q = (const T*)p.q + row * E;
    for (int e = lane; e < E; e += 64) qs[e] = to_f32(q[e]);
    __builtin_amdgcn_wave_barrier();
    const T* kb = (const T*)p.k + ((size_t)(b * p.KH + kvh) * p.KL) * E;
    const T* vb = (const T*)p.v + ((size_t)(b * p.KH + kvh) * p.KL) * E;
    const uint8_t* mp = p.kpad ? p.kpad + (size_t)b * p.KL : nullptr;
    const int kend = p.causal ? (qi + 1 < p.KL ? qi + 1 : p.KL) : p.KL;
    float m = -INFINITY, l = 0.f, oacc[kGenericMaxE / 64];
#pragma unroll
    for (int j = 0; j < kGenericMaxE / 64; ++j) oacc[j] = 0.f;
    for (int k0 = 0; k0 < kend; k0 += 64) {
        const int k = k0 + lane;
        const bool valid = k < kend && (!mp || mp[k] != 0);
        const int kc = k < p.KL ? k : p.KL - 1;
        float s = dot_lds(qs, kb + (size_t)kc * E, E) * p.scale;
        if (p.pair) s += to_f32(((const T*)p.pair)[(((size_t)b * p.KL + kc) * p.QL + qi) * p.QH + qh]);
        if (!valid) s = -INFINITY;
        const float m_new = fmaxf(m, wave_max64(s));
        if (m_new == -INFINITY) continue;                         // no visible key so far (wave-uniform)
        const float pr = valid ? __expf(s - m_new) : 0.f;
        const float alpha = __expf(m - m_new);                   // m = -inf -> 0
        l = l * alpha + wave_sum64(pr);
        m = m_new;
#pragma unroll
        for (int j = 0; j < kGenericMaxE / 64; ++j) oacc[j] *= alpha;
        const int nk = kend - k0 < 64 ? kend - k0 : 64;
        for (int kk = 0; kk < nk; ++kk) {
            const float pk = lane_bcast(pr, kk);
            const T* vr = vb + (size_t)(k0 + kk) * E;
#pragma unroll
            for (int j = 0; j < kGenericMaxE / 64; ++j) {
                const int e = lane + 64 * j;
                if (e < E) oacc[j] += pk * to_f32(vr[e]);
            }
        }
    }
    const float inv = 1.0f / l;                                   // l == 0 (no visible key) -> NaN row, as the naive formula gives
    T* o = (T*)p.o + row * E;
#pragma unroll
    for (int j = 0; j < kGenericMaxE / 64; ++j) {
        const int e = lane + 64 * j;
        if (e < E) o[e] = from_f32<T>(oacc[j] * inv);
    }
    if (lane == 0) {
        // residual contract (src/attention.jl:128-129): ms = row max rounded to T, ls relative to the ROUNDED ms
        const T m_t = from_f32<T>(m);
        float l_out = l;
        if (m != -INFINITY) l_out = l * __expf(m - to_f32(m_t));
        ((T*)p.ms)[row] = m_t;
        ((T*)p.ls)[row] = from_f32<T>(l_out);
    }
}

// lse = ms + log(ls) (natural units; -inf for a row without a visible key) and delta = sum_e dO * o, per query row
template <typename T>
__global__ __launch_bounds__(256) void fa_bwd_generic_pre_kernel(const BwdParams p, int E, long long n_rows) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + wave;
    if (row >= n_rows) return;
    float acc = 0.f;
    for (int e = lane; e < E; e += 64) acc += to_f32(((const T*)p.d_o)[row * E + e]) * to_f32(((const T*)p.o)[row * E + e]);
    acc = wave_sum64(acc);
    if (lane == 0) {
        const float m = to_f32(((const T*)p.ms)[row]), l = to_f32(((const T*)p.ls)[row]);
        // a row that sees no key: o = 0 / 0 = NaN there, so delta would be NaN and dS = P (dP - delta) = 0 * NaN would poison dq and
        // dpair of the row and, through the shared key block, dK of live keys -- the tiled preprocess writes 0 as well (fa_bwd.hpp)
        const bool live = l > 0.f && m != -INFINITY;
        p.nl[row] = live ? m + __logf(l) : -INFINITY;
        p.delta[row] = live ? acc : 0.f;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void fa_bwd_generic_dq_kernel(const BwdParams p, int E, long long n_rows) {
    __shared__ float qs_all[4][kGenericMaxE], dos_all[4][kGenericMaxE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + wave;
    if (row >= n_rows) return;
    const int qi = (int)(row % p.QL);
    const int bh = (int)(row / p.QL);
    const int b = bh / p.QH, qh = bh - b * p.QH;
    const int kvh = qh / (p.QH / p.KH);
    float *qs = qs_all[wave], *dos = dos_all[wave];
    for (int e = lane; e < E; e += 64) {
        qs[e] = to_f32(((const T*)p.q)[row * E + e]);
        dos[e] = to_f32(((const T*)p.d_o)[row * E + e]);
    }
    __builtin_amdgcn_wave_barrier();
    const T* kb = (const T*)p.k + ((size_t)(b * p.KH + kvh) * p.KL) * E;
    const T* vb = (const T*)p.v + ((size_t)(b * p.KH + kvh) * p.KL) * E;
    const uint8_t* mp = p.kpad ? p.kpad + (size_t)b * p.KL : nullptr;
    const int kend = p.causal ? (qi + 1 < p.KL ? qi + 1 : p.KL) : p.KL;
    const float lse = p.nl[row], delta = p.delta[row];
    const bool dead = !(lse > -INFINITY);                        // no visible key: zero gradients (deviation 3)
    float acc[kGenericMaxE / 64];
#pragma unroll
    for (int j = 0; j < kGenericMaxE / 64; ++j) acc[j] = 0.f;
    for (int k0 = 0; k0 < kend; k0 += 64) {
        const int k = k0 + lane;
        const bool valid = !dead && k < kend && (!mp || mp[k] != 0);
        const int kc = k < p.KL ? k : p.KL - 1;
        float s = dot_lds(qs, kb + (size_t)kc * E, E) * p.scale;
        const size_t po = (((size_t)b * p.KL + kc) * p.QL + qi) * p.QH + qh;
        if (p.pair) s += to_f32(((const T*)p.pair)[po]);
        const float pr = valid ? __expf(s - lse) : 0.f;
        const float dp = dot_lds(dos, vb + (size_t)kc * E, E);
        const float ds = pr * (dp - delta);
        if (p.dpair && k < kend) ((T*)p.dpair)[po] = from_f32<T>(ds);      // dpair = dS (src/attention_bwd.jl:123-132); the rest is zero-filled
        const int nk = kend - k0 < 64 ? kend - k0 : 64;
        for (int kk = 0; kk < nk; ++kk) {
            const float dk_ = lane_bcast(ds, kk);
            const T* kr = kb + (size_t)(k0 + kk) * E;
#pragma unroll
            for (int j = 0; j < kGenericMaxE / 64; ++j) {
                const int e = lane + 64 * j;
                if (e < E) acc[j] += dk_ * to_f32(kr[e]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < kGenericMaxE / 64; ++j) {
        const int e = lane + 64 * j;
        if (e < E) ((T*)p.dq)[row * E + e] = from_f32<T>(acc[j] * p.scale);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void fa_bwd_generic_dkdv_kernel(const BwdParams p, int E, long long n_krows) {
    __shared__ float ks_all[4][kGenericMaxE], vs_all[4][kGenericMaxE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long krow = (long long)blockIdx.x * 4 + wave;      // over [B][KH][KL]
    if (krow >= n_krows) return;
    const int k = (int)(krow % p.KL);
    const int bk = (int)(krow / p.KL);
    const int b = bk / p.KH, kvh = bk - b * p.KH;
    const int rep = p.QH / p.KH;
    float *ks = ks_all[wave], *vs = vs_all[wave];
    for (int e = lane; e < E; e += 64) {
        ks[e] = to_f32(((const T*)p.k)[krow * E + e]);
        vs[e] = to_f32(((const T*)p.v)[krow * E + e]);
    }
    __builtin_amdgcn_wave_barrier();
    float dk[kGenericMaxE / 64], dv[kGenericMaxE / 64];
#pragma unroll
    for (int j = 0; j < kGenericMaxE / 64; ++j) { dk[j] = 0.f; dv[j] = 0.f; }
    const bool kvalid = !p.kpad || p.kpad[(size_t)b * p.KL + k] != 0;
    if (kvalid) {
        for (int g = 0; g < rep; ++g) {
            const int qh = kvh * rep + g;
            const size_t rbase = ((size_t)b * p.QH + qh) * p.QL;
            const T* qb = (const T*)p.q + rbase * E;
            const T* dob = (const T*)p.d_o + rbase * E;
            const int qbeg = p.causal ? (k / 64) * 64 : 0;        // queries < k see nothing of this key
            for (int q0 = qbeg; q0 < p.QL; q0 += 64) {
                const int q = q0 + lane;
                const int qc = q < p.QL ? q : p.QL - 1;
                const float lse = p.nl[rbase + qc];
                const bool valid = q < p.QL && (!p.causal || k <= q) && lse > -INFINITY;
                float s = dot_lds(ks, qb + (size_t)qc * E, E) * p.scale;
                if (p.pair) s += to_f32(((const T*)p.pair)[(((size_t)b * p.KL + k) * p.QL + qc) * p.QH + qh]);
                const float pr = valid ? __expf(s - lse) : 0.f;
                const float dp = dot_lds(vs, dob + (size_t)qc * E, E);
                const float ds = pr * (dp - p.delta[rbase + qc]);
                const int nq = p.QL - q0 < 64 ? p.QL - q0 : 64;
                for (int qq = 0; qq < nq; ++qq) {
                    const float pq = lane_bcast(pr, qq), dsq = lane_bcast(ds, qq);
                    const T* qr = qb + (size_t)(q0 + qq) * E;
                    const T* dor = dob + (size_t)(q0 + qq) * E;
#pragma unroll
                    for (int j = 0; j < kGenericMaxE / 64; ++j) {
                        const int e = lane + 64 * j;
                        if (e < E) { dv[j] += pq * to_f32(dor[e]); dk[j] += dsq * to_f32(qr[e]); }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < kGenericMaxE / 64; ++j) {
        const int e = lane + 64 * j;
        if (e < E) {
            ((T*)p.dk)[krow * E + e] = from_f32<T>(dk[j] * p.scale);
            ((T*)p.dv)[krow * E + e] = from_f32<T>(dv[j]);
        }
    }
}

}  // namespace nnop
