// fa_fwd.hpp -- gfx950 flash-attention forward kernel (template; instantiated per dtype in
// fa_fwd_*.hip).
//
// Computes what `_flash_attention_fwd!` computes (src/attention.jl:1-131): per (q-tile, q-head,
// batch)  S = scale*Q K^T (+pair), causal / key-padding mask -> -inf, online softmax, O = P V,
// and the residuals ms (row max) and ls (row sum of exp(s - ms)).  It is a different program:
//
//   reference (attention.jl)                      this kernel
//   -----------------------------------------    ---------------------------------------------
//   1 thread = 1 query row, scalar FMAs out of    1 wave = 32 query rows, both contractions on
//   LDS (mma!, mma.jl:6-48), T accumulation       MFMA 32x32x16 (bf16/f16) / 32x32x2 (f32), fp32 acc
//   S, P round-trip through s_shm                 S^T = K Q^T ("swapped"): the query sits on the
//                                                 lane, its 32 keys in registers -> softmax is
//                                                 in-register, and exp(S^T) IS the B operand of
//                                                 O^T = V^T P^T (no LDS for S or P)
//   O normalised after every tile (FA-1)          O un-normalised, one divide in the epilogue
//   5 barriers / kv tile                          1 barrier / kv tile (double-buffered K,V images)
//   uncoalesced per-row loads                     16-byte coalesced tile loads, issued one tile
//                                                 ahead, written to LDS after the compute phase
//
// Work decomposition: workgroup = NW waves = 32*NW consecutive query rows of one (batch, q-head);
// kv tiles of BK keys; linear workgroup ids are remapped so that the workgroups sharing one
// (batch, kv-head) -- i.e. the same K/V bytes -- run on one XCD's L2.
#pragma once
#include "fa_common.hpp"

namespace nnop {

struct FwdParams {
    void*       o;
    void*       ms;
    void*       ls;
    const void* q;
    const void* k;
    const void* v;
    const void* pair;        // nullable, [B][KL][QL][QH]
    const uint8_t* kpad;     // nullable, [B][KL]
    int   QL, KL, QH, KH, B;
    int   causal;
    int   n_qblk;            // ceil(QL / (32*NW))
    int   n_wg;              // n_qblk * QH * B
    float scale;             // 1/sqrt(E)
};

// kGeneral = false: KL % BK == 0, no causal, no kpad, no pair (every logit is live).
template <typename T, int E, int NW, int BK, bool kGeneral>
__global__ __launch_bounds__(NW * 64) void fa_fwd_kernel(const FwdParams p) {
    using frag_t = typename Elem<T>::frag;
    using KImg   = RowImg<T, E>;
    using VImg   = ColImg<T, E>;
    constexpr int NT  = NW * 64;
    constexpr int KS  = E / 16;                 // contraction steps of Q K^T
    constexpr int KB  = BK / 32;                // 32-key blocks per kv tile
    constexpr int EB  = (E + 31) / 32;          // 32-column blocks of O^T
    constexpr int KBYTES = KImg::bytes(BK);
    constexpr int VBYTES = VImg::bytes(BK);
    constexpr int TILE_BYTES = KBYTES + VBYTES;
    constexpr int N16 = E * (int)sizeof(T) / 16;          // 16-byte chunks per row in HBM
    constexpr int NCH = BK * N16;                          // chunks per tile per tensor
    constexpr int NLD = (NCH + NT - 1) / NT;               // chunks per thread per tensor
    constexpr float kThr = 0.0f;                           // defer-max threshold (log2 units)

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    // ---- which (batch, q-head, q-block) -------------------------------------------------
    int lin = xcd_remap((int)blockIdx.x, p.n_wg);
    int qblk = lin % p.n_qblk;
    const int bh = lin / p.n_qblk;
    if (p.causal) qblk = p.n_qblk - 1 - qblk;              // heaviest q-blocks first
    const int b   = bh / p.QH;
    const int qh  = bh - b * p.QH;
    const int kvh = qh / (p.QH / p.KH);                    // cld(q_head, n_q_per_kv), 0-based
    const int q0w = qblk * (32 * NW) + wave * 32;          // first query row of this wave
    const int qi  = q0w + r;                               // this lane's query row
    const int qi_c = qi < p.QL ? qi : p.QL - 1;            // clamped for loads

    const T* __restrict__ qp = (const T*)p.q + ((size_t)bh * p.QL) * E;
    const T* __restrict__ kp = (const T*)p.k + ((size_t)(b * p.KH + kvh) * p.KL) * E;
    const T* __restrict__ vp = (const T*)p.v + ((size_t)(b * p.KH + kvh) * p.KL) * E;
    const uint8_t* __restrict__ mp = kGeneral && p.kpad ? p.kpad + (size_t)b * p.KL : nullptr;

    // ---- number of kv tiles this workgroup walks ---------------------------------------
    int n_tiles = (p.KL + BK - 1) / BK;
    if (kGeneral && p.causal) {
        int q_last = qblk * (32 * NW) + 32 * NW - 1;
        if (q_last > p.QL - 1) q_last = p.QL - 1;
        const int t_c = q_last / BK + 1;                   // keys <= q_last
        if (t_c < n_tiles) n_tiles = t_c;
    }

    // ---- Q fragments: B operand of S^T = K Q^T, straight from HBM into registers --------
    frag_t qf[KS];
    {
        const T* qrow = qp + (size_t)qi_c * E;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const frag_t*>(qrow + 16 * ks + 8 * h);
    }

    // ---- staging: thread -> 16-byte chunks of the [BK][E] K and V tiles -----------------
    u32x4 kreg[NLD], vreg[NLD];
    auto stage_load = [&](int t) {
        const int k0 = t * BK;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + i * NT;
            const int row = c / N16;
            const bool ok = (NCH % NT == 0 || c < NCH) && (!kGeneral || k0 + row < p.KL);
            u32x4 z = {0u, 0u, 0u, 0u};
            kreg[i] = z;
            vreg[i] = z;
            if (ok) {
                const size_t off = ((size_t)k0 * N16 + c) * 16;
                kreg[i] = *reinterpret_cast<const u32x4*>((const char*)kp + off);
                vreg[i] = *reinterpret_cast<const u32x4*>((const char*)vp + off);
            }
        }
    };
    auto stage_write = [&](char* buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + i * NT;
            if (NCH % NT == 0 || c < NCH) {
                const int row = c / N16, c16 = c % N16;
                KImg::write16(buf, row, c16, kreg[i]);
                VImg::write16(buf + KBYTES, row, c16, vreg[i]);
            }
        }
    };

    // ---- per-lane LDS read addresses -----------------------------------------------------
    const int vbase = VImg::lane_base(lane);

    // ---- accumulators ---------------------------------------------------------------------
    f32x16 oacc[EB];
#pragma unroll
    for (int eb = 0; eb < EB; ++eb)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[eb][i] = 0.f;
    float m2 = -INFINITY;      // running max, log2 units, shared by lanes r and r+32
    float lsum = 0.f;          // running sum over THIS lane's keys only (halves added at the end)
    const float c2 = p.scale * kLog2e;

    stage_load(0);
    stage_write(smem);
    __syncthreads();

    for (int t = 0; t < n_tiles; ++t) {
        char* cur = smem + (t & 1) * TILE_BYTES;
        char* nxt = smem + ((t + 1) & 1) * TILE_BYTES;
        const bool more = t + 1 < n_tiles;
        if (more) stage_load(t + 1);

        const int k0 = t * BK;
        // wave-uniform tile classification (general mode only)
        bool skip = false, need_mask = false;
        uint64_t valid = ~0ull;
        if constexpr (kGeneral) {
            if (p.causal && k0 > q0w + 31) skip = true;                 // tile entirely above diagonal
            if (BK < 64) valid = (1ull << BK) - 1ull;
            if (k0 + BK > p.KL) valid &= (p.KL - k0 >= 64) ? ~0ull : ((1ull << (p.KL - k0)) - 1ull);
            if (mp) {
                const int kk = k0 + lane;
                const bool lv = (lane < BK && kk < p.KL) ? (mp[kk] != 0) : false;
                valid &= __ballot(lv);
            }
            if (valid == 0ull) skip = true;
            need_mask = (valid != ((BK < 64) ? ((1ull << BK) - 1ull) : ~0ull)) ||
                        (p.causal && k0 + BK - 1 > q0w) || (p.pair != nullptr);
        }

        if (!skip) {
            // ---- S^T = K Q^T : keys in registers, query on the lane -----------------------
            f32x16 s[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
                for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    frag_t kf = KImg::read_row_frag(cur, 32 * kb + r, h, ks);
                    s[kb] = mma16<T>(kf, qf[ks], s[kb]);
                }
            }

            // ---- logits in log2 units, masks ----------------------------------------------
            float mx = -INFINITY;
            if constexpr (kGeneral) {
                if (need_mask) {
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb) {
                        const uint32_t w = (uint32_t)(valid >> (32 * kb + 4 * h));
                        const int lim = qi - k0 - 32 * kb - 4 * h;   // causal: local row <= lim
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            constexpr int dummy = 0; (void)dummy;
                            const int lr = (i & 3) + 8 * (i >> 2);
                            bool ok = (w >> lr) & 1u;
                            if (p.causal) ok = ok && (lr <= lim);
                            float x = s[kb][i] * c2;
                            if (p.pair) {
                                const int key = k0 + 32 * kb + lr + 4 * h;
                                if (ok && qi < p.QL) {
                                    const size_t po = (((size_t)b * p.KL + key) * p.QL + qi) * p.QH + qh;
                                    x += to_f32(((const T*)p.pair)[po]) * kLog2e;
                                }
                            }
                            s[kb][i] = ok ? x : -INFINITY;
                            mx = fmaxf(mx, s[kb][i]);
                        }
                    }
                }
            }
            const bool premul = kGeneral && need_mask;     // s already in log2 units
            if (!premul) {
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[kb][i]);
                mx *= c2;
            }
            mx = half_swap_max(mx);

            // ---- online softmax: rescale only when some row's max grew ---------------------
            if (__any(mx > m2 + kThr)) {
                const float mn = fmaxf(m2, mx);
                const float alpha = (mn == -INFINITY) ? 1.f : fast_exp2(m2 - mn);
#pragma unroll
                for (int eb = 0; eb < EB; ++eb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) oacc[eb][i] *= alpha;
                lsum *= alpha;
                m2 = mn;
            }
            const float msub = (kGeneral && m2 == -INFINITY) ? 0.f : m2;
            if (premul) {
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        s[kb][i] = fast_exp2(s[kb][i] - msub);
                        lsum += s[kb][i];
                    }
            } else {
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        s[kb][i] = fast_exp2(__builtin_fmaf(s[kb][i], c2, -msub));
                        lsum += s[kb][i];
                    }
            }

            // ---- O^T += V^T P^T : P^T comes straight from the S^T accumulators -------------
            frag_t pf[2 * KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                pf[2 * kb]     = acc_frag<T, 0>(s[kb]);
                pf[2 * kb + 1] = acc_frag<T, 1>(s[kb]);
            }
            const char* vb = cur + KBYTES + vbase;
#pragma unroll
            for (int eb = 0; eb < EB; ++eb)
#pragma unroll
                for (int kk = 0; kk < 2 * KB; ++kk) {
                    frag_t vf = VImg::read_col_frag(vb, kk, eb);
                    oacc[eb] = mma16<T>(vf, pf[kk], oacc[eb]);
                }
        }

        if (more) stage_write(nxt);
        __syncthreads();
    }

    // ---- epilogue: normalise, store o, ms, ls -----------------------------------------------
    const float ltot = half_swap_sum(lsum);
    const float inv = 1.0f / ltot;                         // ltot == 0 (no visible key) -> NaN rows,
                                                           // as the naive formula gives
    if (qi < p.QL) {
        T* orow = (T*)p.o + ((size_t)bh * p.QL + qi) * E;
#pragma unroll
        for (int eb = 0; eb < EB; ++eb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int e = 32 * eb + 8 * g + 4 * h;
                if (e < E) {
                    if constexpr (sizeof(T) == 4) {
                        f32x4 w = {oacc[eb][4 * g] * inv, oacc[eb][4 * g + 1] * inv,
                                   oacc[eb][4 * g + 2] * inv, oacc[eb][4 * g + 3] * inv};
                        *reinterpret_cast<f32x4*>(orow + e) = w;
                    } else {
                        typedef T t4 __attribute__((ext_vector_type(4)));
                        f32x4 w = {oacc[eb][4 * g] * inv, oacc[eb][4 * g + 1] * inv,
                                   oacc[eb][4 * g + 2] * inv, oacc[eb][4 * g + 3] * inv};
                        *reinterpret_cast<t4*>(orow + e) = __builtin_convertvector(w, t4);
                    }
                }
            }
        if (h == 0) {
            // Residual contract (src/attention.jl:128-129): ms = row max (natural-log units),
            // ls = sum exp(s - ms), both in T.  ms is rounded to T first and ls is expressed
            // relative to the ROUNDED ms, so the pair stays self-consistent in 16-bit types.
            const size_t so = (size_t)bh * p.QL + qi;
            const float m_nat = m2 * kLn2;
            const T m_t = from_f32<T>(m_nat);
            const float m_back = to_f32(m_t);
            float l_out = ltot;
            if (m2 != -INFINITY) l_out = ltot * fast_exp2((m_nat - m_back) * kLog2e);
            ((T*)p.ms)[so] = m_t;
            ((T*)p.ls)[so] = from_f32<T>(l_out);
        }
    }
}

// LDS bytes the kernel needs (two buffers of a K image + a V image).
template <typename T, int E, int BK> constexpr int fa_fwd_lds_bytes() {
    return 2 * (RowImg<T, E>::bytes(BK) + ColImg<T, E>::bytes(BK));
}

}  // namespace nnop
