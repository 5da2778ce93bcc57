// fa_fwd_split.hpp -- forward kernel, "split-KV inside the workgroup" form (plain mode, 16-bit types).
//
// Same math and the same MFMA/LDS building blocks as fa_fwd.hpp, different occupancy strategy.
// At the headline shape (E=64, L=4096, H=4, B=4) the chip has 65,536 query rows for 256 CUs: 256 rows
// = eight 32-row waves per CU = 2 waves per SIMD, and every in-order stall of a wave (LDS round trip,
// MFMA -> VALU dependency, barrier) leaves the matrix pipe idle -- measured ~50 % busy at the clock the
// chip holds, and insensitive to software pipelining / prefetch inside a wave.
// More rows per CU do not exist, so the extra waves come from the KEY axis:
//
//   workgroup = 16 waves = 2 groups x 8 waves; wave (g, w) owns query rows [32w, 32w+32) of the
//   workgroup's 256 rows and kv tiles t = 2i + g.  Both groups run the plain online-softmax loop on
//   their own tiles (4 waves per SIMD, <= 128 VGPRs each); at the end group 1 hands its (O, m, l) to
//   group 0 through LDS and group 0 merges, normalises and stores.
//
// LDS: per group a K ring of 2 and a V ring of 2 (one tile pair in use, the next pair being written);
// the rings are reused for the final hand-off.
#pragma once
#include "fa_fwd.hpp"

// row sums on the matrix pipe: measured 5 % slower; experiment switch of make DEV=1 builds only
#if !defined(NNOP_DEV_BUILD)
#undef NNOP_SPLIT_MFMASUM
#endif
#ifndef NNOP_SPLIT_MFMASUM
#define NNOP_SPLIT_MFMASUM 0
#endif

namespace nnop {

template <typename T, int E> constexpr int fa_fwd_split_lds_bytes() {
    constexpr int rings = 2 * 2 * (RowImg<T, E>::bytes(64) + ColImg<T, E>::bytes(64));   // [group][slot](K + V)
    constexpr int handoff = 8 * 64 * 4 * (((E + 31) / 32) * 16 + 3);                     // [wave][reg][lane] fp32
    return rings > handoff ? rings : handoff;
}

template <typename T, int E>
__global__ __launch_bounds__(1024) void fa_fwd_split_kernel(const FwdParams p) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    using frag_t = typename Elem<T>::frag;
    using KImg   = RowImg<T, E>;
    using VImg   = ColImg<T, E>;
    constexpr int BK = 64, KB = 2, KS = E / 16, EB = (E + 31) / 32;
    constexpr int NT = 1024;
    constexpr int KBYTES = KImg::bytes(BK), VBYTES = VImg::bytes(BK);
    constexpr int GRP = 2 * (KBYTES + VBYTES);            // one group's rings: [slot](K, V)
    constexpr int N16 = E * (int)sizeof(T) / 16;           // 16-byte chunks per row
    constexpr int NCH = BK * N16;                          // chunks per tile per tensor
    constexpr int NLD = (4 * NCH + NT - 1) / NT;           // chunks per thread per step (2 K + 2 V tiles)
    constexpr float kThr = 8.0f;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 3, w8 = wave & 7;
    const int r = lane & 31, h = lane >> 5;

    const int lin = xcd_remap_chunked((int)blockIdx.x, p.n_wg, p.n_qblk * (p.QH / p.KH));
    const int qblk = lin % p.n_qblk;
    const int bh = lin / p.n_qblk;
    const int b = bh / p.QH, qh = bh - b * p.QH;
    const int kvh = qh / (p.QH / p.KH);
    const int qi = qblk * 256 + w8 * 32 + r;
    const int qc = qi < p.QL ? qi : p.QL - 1;

    const T* __restrict__ qp = (const T*)p.q + ((size_t)bh * p.QL) * E;
    const char* __restrict__ kp = (const char*)((const T*)p.k + ((size_t)(b * p.KH + kvh) * p.KL) * E);
    const char* __restrict__ vp = (const char*)((const T*)p.v + ((size_t)(b * p.KH + kvh) * p.KL) * E);

    const int n_tiles = p.KL / BK;                         // plain mode: KL % 64 == 0
    const int n_steps = (n_tiles + 1) / 2;
    const float c2 = p.scale * kLog2e;

    frag_t qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const frag_t*>(qp + (size_t)qc * E + 16 * ks + 8 * h);

    // ---- staging: 1024 threads move the step's 4 tiles (K even, K odd, V even, V odd) -------------
    u32x4 sreg[NLD];
    auto stage_load = [&](int step) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int c = tid + i * NT;
            if ((4 * NCH) % NT != 0) c = c < 4 * NCH ? c : 4 * NCH - 1;
            const int which = c / NCH, cc = c % NCH;       // 0: K(2s) 1: K(2s+1) 2: V(2s) 3: V(2s+1)
            int t = 2 * step + (which & 1);
            t = t < n_tiles ? t : n_tiles - 1;             // odd tile count: group 1 ignores its last tile
            const char* base = (which < 2 ? kp : vp) + (size_t)t * ((size_t)BK * E * sizeof(T));
            sreg[i] = *reinterpret_cast<const u32x4*>(base + (size_t)cc * 16);
        }
    };
    auto stage_write = [&](int slot) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + i * NT;
            if ((4 * NCH) % NT == 0 || c < 4 * NCH) {
                const int which = c / NCH, cc = c % NCH;
                char* gbase = smem + (which & 1) * GRP + slot * (KBYTES + VBYTES);
                if (which < 2) KImg::write16(gbase, cc / N16, cc % N16, sreg[i]);
                else VImg::write16(gbase + KBYTES, cc / N16, cc % N16, sreg[i]);
            }
        }
    };

    f32x16 oacc[EB];
#pragma unroll
    for (int eb = 0; eb < EB; ++eb)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[eb][i] = 0.f;
    float m2 = -INFINITY, mt = -INFINITY, lsum = 0.f;      // reference max (log2 units), true max, row sum
#if NNOP_SPLIT_MFMASUM
    frag_t ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = from_f32<T>(1.0f);
#endif
    const int vbase = VImg::lane_base(lane);

    stage_load(0);
    stage_write(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) landed(qf[ks]);
    __syncthreads();

    for (int step = 0; step < n_steps; ++step) {
        const bool more = step + 1 < n_steps;
        if (more) stage_load(step + 1);
        const char* kimg = smem + grp * GRP + (step & 1) * (KBYTES + VBYTES);
        const char* vimg = kimg + KBYTES;
        if (2 * step + grp < n_tiles) {
            // S^T = K Q^T (raw units)
            f32x16 s[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
                for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    s[kb] = mma16<T>(KImg::read_row_frag(kimg, 32 * kb + r, h, ks), qf[ks], s[kb]);
            }
            float mxp[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int i = 0; i < 16; i += 2)
                    mxp[(i >> 1) & 3] = fmaxf(fmaxf(mxp[(i >> 1) & 3], s[kb][i]), s[kb][i + 1]);
            const float mx = half_swap_max(fmaxf(fmaxf(mxp[0], mxp[1]), fmaxf(mxp[2], mxp[3])) * c2);
            mt = fmaxf(mt, mx);
            if (__any(mx > m2 + kThr)) {                   // deferred-max rescale (rare after the first tiles)
                const float mn = fmaxf(m2, mx);
                const float alpha = fast_exp2(m2 - mn);    // m2 = -inf at the first tile -> 0
#pragma unroll
                for (int eb = 0; eb < EB; ++eb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) oacc[eb][i] *= alpha;
                lsum *= alpha;
                m2 = mn;
            }
            const char* vb = vimg + vbase;
#if NNOP_SPLIT_MFMASUM
            // Row sums on the matrix pipe (which has slack at E <= 64; the vector-issue port does not, DESIGN.md
            // section 5): ones(32 x 16) x P(16 keys x 32 queries) leaves sum_k P[k][query] in every register of the
            // result; the result tile lives only within this kv tile (in the registers of the consumed score tile),
            // one v_add per tile folds it into the running sum.  Sums the bf16/fp16-ROUNDED P, i.e. exactly the
            // weights the PV product uses.
            f32x16 lt;
#pragma unroll
            for (int kk = 0; kk < 2 * KB; ++kk) {
                const int kb = kk >> 1, i0 = 8 * (kk & 1);
#pragma unroll
                for (int j = 0; j < 8; ++j) s[kb][i0 + j] = fast_exp2(__builtin_fmaf(s[kb][i0 + j], c2, -m2));
                const frag_t pf = (kk & 1) ? acc_frag<T, 1>(s[kb]) : acc_frag<T, 0>(s[kb]);
                if (kk == 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) lt[i] = 0.f;
                }
                lt = mma16<T>(ones, pf, lt);
#pragma unroll
                for (int eb = 0; eb < EB; ++eb)
                    oacc[eb] = mma16<T>(VImg::read_col_frag(vb, kk, eb), pf, oacc[eb]);
            }
            lsum += lt[0];
#else
            float lp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2 * KB; ++kk) {
                const int kb = kk >> 1, i0 = 8 * (kk & 1);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    s[kb][i0 + j] = fast_exp2(__builtin_fmaf(s[kb][i0 + j], c2, -m2));
                    lp[j & 3] += s[kb][i0 + j];
                }
                const frag_t pf = (kk & 1) ? acc_frag<T, 1>(s[kb]) : acc_frag<T, 0>(s[kb]);
#pragma unroll
                for (int eb = 0; eb < EB; ++eb)
                    oacc[eb] = mma16<T>(VImg::read_col_frag(vb, kk, eb), pf, oacc[eb]);
            }
            lsum += (lp[0] + lp[1]) + (lp[2] + lp[3]);
#endif
        }
        if (more) stage_write((step + 1) & 1);
        __syncthreads();
    }

    // ---- merge the two key halves: group 1 -> LDS -> group 0 -------------------------------------
    constexpr int NREG = EB * 16 + 3;
    float* xch = reinterpret_cast<float*>(smem) + (size_t)w8 * NREG * 64 + lane;     // [wave][reg][lane]
    if (grp == 1) {
#pragma unroll
        for (int eb = 0; eb < EB; ++eb)
#pragma unroll
            for (int i = 0; i < 16; ++i) xch[(eb * 16 + i) * 64] = oacc[eb][i];
        xch[(EB * 16 + 0) * 64] = m2;
        xch[(EB * 16 + 1) * 64] = mt;
        xch[(EB * 16 + 2) * 64] = lsum;
    }
    __syncthreads();
    if (grp == 0) {
        const float m2b = xch[(EB * 16 + 0) * 64], mtb = xch[(EB * 16 + 1) * 64], lb = xch[(EB * 16 + 2) * 64];
        const float mn = fmaxf(m2, m2b);
        const float fa = (m2 == -INFINITY) ? 0.f : fast_exp2(m2 - mn);
        const float fb = (m2b == -INFINITY) ? 0.f : fast_exp2(m2b - mn);
        const float mtt = fmaxf(mt, mtb);
#if NNOP_SPLIT_MFMASUM
        const float ltot = lsum * fa + lb * fb;            // the MFMA sums already span both lane halves' keys
#else
        const float ltot = half_swap_sum(lsum * fa + lb * fb);
#endif
        const float inv = 1.0f / ltot;
        if (qi < p.QL) {
            T* orow = (T*)p.o + ((size_t)bh * p.QL + qi) * E;
#pragma unroll
            for (int eb = 0; eb < EB; ++eb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int e = 32 * eb + 8 * g + 4 * h;
                    if (e < E) {
                        f32x4 w;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            w[j] = (oacc[eb][4 * g + j] * fa + xch[(eb * 16 + 4 * g + j) * 64] * fb) * inv;
                        typedef T t4 __attribute__((ext_vector_type(4)));
                        *reinterpret_cast<t4*>(orow + e) = __builtin_convertvector(w, t4);
                    }
                }
            if (h == 0) {
                // residual contract as in fa_fwd.hpp (src/attention.jl:128-129)
                const size_t so = (size_t)bh * p.QL + qi;
                const float m_nat = mtt * kLn2;
                const T m_t = from_f32<T>(m_nat);
                const float m_back = to_f32(m_t);
                float l_out = ltot;
                if (mtt != -INFINITY) l_out = ltot * fast_exp2(mn - m_back * kLog2e);
                ((T*)p.ms)[so] = m_t;
                ((T*)p.ls)[so] = from_f32<T>(l_out);
            }
        }
    }
}

}  // namespace nnop
