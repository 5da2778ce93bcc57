// fa_fwd_split16.hpp -- forward kernel, split-KV workgroup form (fa_fwd_split.hpp) on v_mfma_f32_16x16x32_{bf16,f16}.
//
// Same schedule as fa_fwd_split_kernel: 16 waves = 2 groups x 8 waves, wave (g, w) owns 32 query rows and the kv
// tiles t = 2i + g, (O, m, l) of group 1 merged into group 0 through LDS.  Different matrix instruction: the
// 16x16x32 shape holds a higher clock than 32x32x16 on this power-limited part at equal cycles per flop
// (MI355X_MICROARCH.md, "Shape": ~1.15x in bare loops).  Fragment geometry (lane l = (i = l & 15, g = l >> 4)):
//   A operand 16 x 32 : row i, k = 8g + j          B operand 32 x 16 : column i, k = 8g + j
//   C / D     16 x 16 : column i, rows 4g + r (4 registers)
// S^T = K Q^T per (16 keys kb) x (16 queries qb): A = K rows from the swizzled row image (one ds_read_b128 per
// (kb, 32-wide contraction step), shared by the two query blocks), B = Q from registers; a lane ends up with keys
// 16 kb + 4g + r of query i.  O^T += V^T P per (16 columns eb) x (16 queries): the contraction runs over the 32 keys
// of a key-block pair (2p, 2p+1) in the order k = 8g + j <-> key 16(2p + (j >> 2)) + 4g + (j & 3), which is exactly
// what a lane already holds in its S registers (no lane movement), and what two ds_read_b64_tr_b16 of the blocked V
// image return for a lane (column i of the 4 x 16 block of rows 4g..4g+3).
#pragma once
#include "fa_fwd_split.hpp"

namespace nnop {

template <typename T> NNOP_DEV f32x4 mma32(typename Elem<T>::frag a, typename Elem<T>::frag b, f32x4 c);
template <> NNOP_DEV f32x4 mma32<__bf16>(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <> NNOP_DEV f32x4 mma32<_Float16>(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// max / sum over the four 16-lane quarters of a wave (lanes l, l^16, l^32, l^48), result in every lane
NNOP_DEV float quarters_max(float x) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return half_swap_max(fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])));
}
NNOP_DEV float quarters_sum(float x) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return half_swap_sum(__uint_as_float(r[0]) + __uint_as_float(r[1]));
}

// V image for the 16x16x32 column read: ColImg's 4-row x 32-column blocks (256 B = one LDS bank row), with the two
// 16-column halves of odd blocks swapped.  A ds_read_b64_tr_b16 is serviced in two 32-lane halves; here a half is two
// quarters g = 2h, 2h+1 reading the SAME 16 columns of two adjacent blocks (rows 4g..4g+3): without the swap they
// would sit on the same banks.
template <typename T, int E> struct ColImg16 {
    static_assert(sizeof(T) == 2 && E % 32 == 0, "16-bit types, whole 32-column blocks");
    static constexpr int kEB = E / 32;
    static constexpr int bytes(int rows) { return rows * E * (int)sizeof(T); }
    NNOP_DEV static void write16(char* img, int row, int c16, u32x4 v) {      // chunk c16 = 8 columns
        const int blk = row >> 2;
        const int off = ((blk * kEB + (c16 >> 2)) << 8) + ((row & 3) << 6) + ((((c16 & 3) ^ ((blk & 1) << 1))) << 4);
        *reinterpret_cast<u32x4*>(img + off) = v;
    }
    NNOP_DEV static int lane_base(int lane) {
        const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
        return g * kEB * 256 + q * 64 + p * 8;
    }
    // A operand (V^T) for embedding columns [16 eb, 16 eb + 16) and key blocks 2*pair, 2*pair + 1 of a 64-key tile.
    // base = img + lane_base(lane); g = lane >> 4.
    NNOP_DEV static typename Elem<T>::frag read(const char* base, int pair, int eb, int g) {
        typedef __attribute__((address_space(3))) s16x4* lds_p;
        const int seg = (((eb & 1) ^ (g & 1)) << 5) + ((eb >> 1) << 8);
        const char* p0 = base + ((8 * pair * kEB) << 8) + seg;
        const char* p1 = base + (((8 * pair + 4) * kEB) << 8) + seg;
        s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
        s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p1));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        return __builtin_bit_cast(typename Elem<T>::frag, r);
    }
};

template <typename T, int E> constexpr int fa_fwd_split16_lds_bytes() {
    constexpr int rings = 2 * 2 * (RowImg<T, E>::bytes(64) + ColImg16<T, E>::bytes(64));   // [group][slot](K + V)
    constexpr int handoff = 8 * 64 * 4 * (2 * (E / 16) * 4 + 6);                           // [wave][reg][lane] fp32
    return rings > handoff ? rings : handoff;
}

template <typename T, int E>
__global__ __launch_bounds__(1024) void fa_fwd_split16_kernel(const FwdParams p) {
    static_assert(sizeof(T) == 2 && E % 32 == 0, "16-bit element types, E a multiple of 32");
    using frag_t = typename Elem<T>::frag;
    using KImg   = RowImg<T, E>;
    using VImg   = ColImg16<T, E>;
    constexpr int BK = 64, NKB = 4, KS = E / 32, EB = E / 16, QB = 2;
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
    const int li = lane & 15, g = lane >> 4;

    const int lin = xcd_remap_chunked((int)blockIdx.x, p.n_wg, p.n_qblk * (p.QH / p.KH));
    const int qblk = lin % p.n_qblk;
    const int bh = lin / p.n_qblk;
    const int b = bh / p.QH, qh = bh - b * p.QH;
    const int kvh = qh / (p.QH / p.KH);
    const int q0 = qblk * 256 + w8 * 32;

    const T* __restrict__ qp = (const T*)p.q + ((size_t)bh * p.QL) * E;
    const char* __restrict__ kp = (const char*)((const T*)p.k + ((size_t)(b * p.KH + kvh) * p.KL) * E);
    const char* __restrict__ vp = (const char*)((const T*)p.v + ((size_t)(b * p.KH + kvh) * p.KL) * E);

    const int n_tiles = p.KL / BK;                         // plain mode: KL % 64 == 0
    const int n_steps = (n_tiles + 1) / 2;
    const float c2 = p.scale * kLog2e;

    frag_t qf[QB][KS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qi = q0 + 16 * qb + li;
        const int qc = qi < p.QL ? qi : p.QL - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[qb][ks] = *reinterpret_cast<const frag_t*>(qp + (size_t)qc * E + 32 * ks + 8 * g);
    }

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

    f32x4 oacc[QB][EB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int eb = 0; eb < EB; ++eb) oacc[qb][eb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m2[QB], mt[QB], lsum[QB];                         // reference max (log2 units), true max, row-sum partial
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) { m2[qb] = -INFINITY; mt[qb] = -INFINITY; lsum[qb] = 0.f; }
    const int vbase = VImg::lane_base(lane);

    stage_load(0);
    stage_write(0);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) landed(qf[qb][ks]);
    __syncthreads();

    for (int step = 0; step < n_steps; ++step) {
        const bool more = step + 1 < n_steps;
        if (more) stage_load(step + 1);
        const char* kimg = smem + grp * GRP + (step & 1) * (KBYTES + VBYTES);
        const char* vimg = kimg + KBYTES;
        if (2 * step + grp < n_tiles) {
            // S^T = K Q^T (raw units): s[qb][kb][r] = key 16 kb + 4g + r  x  query 16 qb + li
            f32x4 s[QB][NKB];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) s[qb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const frag_t kf = *reinterpret_cast<const frag_t*>(kimg + KImg::off(16 * kb + li, 4 * ks + g));
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) s[qb][kb] = mma32<T>(kf, qf[qb][ks], s[qb][kb]);
                }
            }
            float mx[QB];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                float a0 = fmaxf(fmaxf(s[qb][0][0], s[qb][0][1]), fmaxf(s[qb][0][2], s[qb][0][3]));
#pragma unroll
                for (int kb = 1; kb < NKB; ++kb)
                    a0 = fmaxf(fmaxf(fmaxf(a0, s[qb][kb][0]), s[qb][kb][1]), fmaxf(s[qb][kb][2], s[qb][kb][3]));
                mx[qb] = quarters_max(a0 * c2);
                mt[qb] = fmaxf(mt[qb], mx[qb]);
            }
            if (__any(mx[0] > m2[0] + kThr || mx[1] > m2[1] + kThr)) {    // deferred-max rescale (rare after the first tiles)
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    const float mn = fmaxf(m2[qb], mx[qb]);
                    const float alpha = fast_exp2(m2[qb] - mn);            // m2 = -inf at the first tile -> 0
#pragma unroll
                    for (int eb = 0; eb < EB; ++eb) oacc[qb][eb] *= alpha;
                    lsum[qb] *= alpha;
                    m2[qb] = mn;
                }
            }
            const char* vb = vimg + vbase;
#pragma unroll
            for (int pair = 0; pair < 2; ++pair) {
                frag_t pf[QB];
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    f32x8 t;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float e = fast_exp2(__builtin_fmaf(s[qb][2 * pair + (j >> 2)][j & 3], c2, -m2[qb]));
                        t[j] = e;
                        lsum[qb] += e;
                    }
                    pf[qb] = __builtin_convertvector(t, frag_t);
                }
#pragma unroll
                for (int eb = 0; eb < EB; ++eb) {
                    const frag_t vf = VImg::read(vb, pair, eb, g);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) oacc[qb][eb] = mma32<T>(vf, pf[qb], oacc[qb][eb]);
                }
            }
        }
        if (more) stage_write((step + 1) & 1);
        __syncthreads();
    }

    // ---- merge the two key halves: group 1 -> LDS -> group 0 -------------------------------------
    constexpr int NO = QB * EB * 4;                        // O registers per lane
    constexpr int NREG = NO + 3 * QB;
    float* xch = reinterpret_cast<float*>(smem) + (size_t)w8 * NREG * 64 + lane;     // [wave][reg][lane]
    if (grp == 1) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
            for (int eb = 0; eb < EB; ++eb)
#pragma unroll
                for (int r = 0; r < 4; ++r) xch[((qb * EB + eb) * 4 + r) * 64] = oacc[qb][eb][r];
            xch[(NO + 3 * qb + 0) * 64] = m2[qb];
            xch[(NO + 3 * qb + 1) * 64] = mt[qb];
            xch[(NO + 3 * qb + 2) * 64] = lsum[qb];
        }
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const float m2b = xch[(NO + 3 * qb + 0) * 64], mtb = xch[(NO + 3 * qb + 1) * 64], lb = xch[(NO + 3 * qb + 2) * 64];
            const float mn = fmaxf(m2[qb], m2b);
            const float fa = (m2[qb] == -INFINITY) ? 0.f : fast_exp2(m2[qb] - mn);
            const float fb = (m2b == -INFINITY) ? 0.f : fast_exp2(m2b - mn);
            const float mtt = fmaxf(mt[qb], mtb);
            const float ltot = quarters_sum(lsum[qb] * fa + lb * fb);
            const float inv = 1.0f / ltot;
            const int qi = q0 + 16 * qb + li;
            if (qi < p.QL) {
                T* orow = (T*)p.o + ((size_t)bh * p.QL + qi) * E;
#pragma unroll
                for (int eb = 0; eb < EB; ++eb) {
                    f32x4 w;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        w[r] = (oacc[qb][eb][r] * fa + xch[((qb * EB + eb) * 4 + r) * 64] * fb) * inv;
                    typedef T t4 __attribute__((ext_vector_type(4)));
                    *reinterpret_cast<t4*>(orow + 16 * eb + 4 * g) = __builtin_convertvector(w, t4);
                }
                if (g == 0) {
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
}

}  // namespace nnop
