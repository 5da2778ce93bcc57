// fa_fwd.hpp -- gfx950 flash-attention forward kernel (template; instantiated per dtype in
// fa_fwd_*.hip).
//
// Computes what `_flash_attention_fwd!` computes (src/attention.jl:1-131): per (q-tile, q-head,
// batch)  S = scale*Q K^T (+pair), causal / key-padding mask -> -inf, online softmax, O = P V,
// and the residuals ms (row max) and ls (row sum of exp(s - ms)).  It is a different program:
//
//   reference (attention.jl)                      this kernel
//   -----------------------------------------    ---------------------------------------------
//   1 thread = 1 query row, scalar FMAs out of    1 wave = QB blocks of 32 query rows, both
//   LDS (mma!, mma.jl:6-48), T accumulation       contractions on MFMA 32x32x16 (bf16/f16) /
//                                                 32x32x2 (f32), fp32 accumulation
//   S, P round-trip through s_shm                 S^T = K Q^T ("swapped"): the query sits on the
//                                                 lane, its 32 keys in registers -> softmax is
//                                                 in-register, and exp(S^T) IS the B operand of
//                                                 O^T = V^T P^T (no LDS for S or P)
//   O normalised after every tile (FA-1)          O un-normalised, one divide in the epilogue
//   5 barriers / kv tile                          1 barrier / kv tile (K ring + V ring in LDS)
//   uncoalesced per-row loads                     16-byte coalesced tile loads, issued ahead,
//                                                 written to LDS after the compute phase
//   QK^T, softmax, PV strictly one after the      software-pipelined across tiles INSIDE a wave:
//   other                                         one straight-line block holds the QK^T MFMAs of
//                                                 tile t+1, the exp/convert of tile t, the PV
//                                                 MFMAs of tile t and the row max of tile t+1
//   scale, max-subtraction, exp in T              one fp32 fma + v_exp per logit: P = exp2(s*c - m);
//                                                 rescale deferred (threshold 2^8); row sums by MFMA
//                                                 (all-ones A operand) where registers allow
//
// Work decomposition: workgroup = NW waves x QB x 32 consecutive query rows of one (batch, q-head);
// kv tiles of BK keys.  QB = 2 (64 rows per wave, one wave per SIMD, 512-register budget) halves the
// LDS fragment bytes per MFMA -- every K / V fragment read from LDS feeds two MFMAs -- which is what
// bounds the QB = 1 form (measured: per kv tile a wave waited ~700 cycles for its fragments and
// ~470 at the barrier against ~1100 in the MFMA block; profiles/r01/).  Linear workgroup ids are
// remapped so that the workgroups sharing one (batch, kv-head) -- the same K/V bytes -- run on one
// XCD's L2.
#pragma once
#include <type_traits>
#include "fa_common.hpp"

// Timing-only ablation builds (make DEV=1 ABL=n OUTDIR=../lib_abl<n>): results are WRONG by construction.
//   1 no per-interval barrier   2 no exp   3 no PV MFMAs   4 no QK^T MFMAs   5 no LDS fragment reads
//   6 no HBM->LDS staging       7 no row-sum MFMAs          8 no row max
// The release build (no NNOP_DEV_BUILD) pins NNOP_ABL to 0: none of that code exists in the shipped library.
#if !defined(NNOP_DEV_BUILD)
#undef NNOP_ABL
#endif
#ifndef NNOP_ABL
#define NNOP_ABL 0
#endif
// Variant switches for A/B timing (make DEV=1 VAR="-DNNOP_V_...=0"); defaults are the shipped configuration and the
// only one a release build can have.
#if !defined(NNOP_DEV_BUILD)
#undef NNOP_V_PREFETCH
#undef NNOP_V_DEEP
#undef NNOP_V_MFMASUM
#undef NNOP_V_SETPRIO
#endif
#ifndef NNOP_V_PREFETCH
#define NNOP_V_PREFETCH 1
#endif
#ifndef NNOP_V_DEEP
#define NNOP_V_DEEP 1
#endif
#ifndef NNOP_V_MFMASUM
#define NNOP_V_MFMASUM 1
#endif
#ifndef NNOP_V_SETPRIO
#define NNOP_V_SETPRIO 0
#endif

namespace nnop {

// key padding: one 64-bit validity word per 64 keys is kept in LDS for up to this many words (64K keys)
constexpr int kMaxMaskTiles = 1024;

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
    int   n_qblk;            // ceil(QL / (32*QB*NW))
    int   n_wg;              // n_qblk * QH * B
    float scale;             // 1/sqrt(E)
    int   persist_hx = 0;    // persistent form: heads per XCD when QH % 8 == 0 (the XCD's columns = those heads of every batch), else 0
    int   persist = 0;       // fa_fwd_w64_kernel: blocks per workgroup of the persistent form (grid = 256 workgroups), 0 = one block per workgroup
    int   persist_asc = 0;   // persistent form: q-blocks of a column in ASCENDING order (light blocks first), see fwd_persist_plan
    int   causal_alt = 0;    // fa_fwd_kernel, causal: g > 0 -- every second run of g consecutive blocks of an XCD's dispatch order (whole columns) runs its q-blocks ascending (launch_fwd_cfg)
#ifdef NNOP_DEV_BUILD
    int   stagger = 0;       // experiment: s_sleep units for the odd co-resident workgroup (0 = off)
#endif
};

// MODE 0: plain   -- KL % BK == 0, no causal, no kpad, no pair: every logit is live
// MODE 1: masked  -- causal and/or key padding and/or ragged KL
// MODE 2: pair    -- masked + additive pair bias
template <typename T, int E, int NW, int BK, int MODE, int QB>
__global__ __launch_bounds__(NW * 64, (QB == 2 || (sizeof(T) == 4 && E >= 256)) ? 1 : 2) void fa_fwd_kernel(const FwdParams p) {
    using frag_t = typename Elem<T>::frag;
    using KImg   = RowImg<T, E>;
    using VImg   = ColImg<T, E>;
    constexpr bool kGeneral = MODE != 0;
    constexpr bool kPair = MODE == 2;
    constexpr int NT  = NW * 64;
    constexpr int KS  = E / 16;                 // contraction steps of Q K^T
    constexpr int KB  = BK / 32;                // 32-key blocks per kv tile
    constexpr int EB  = (E + 31) / 32;          // 32-column blocks of O^T
    constexpr int WROWS = 32 * QB;              // query rows per wave
    constexpr int KBYTES = KImg::bytes(BK);
    constexpr int VBYTES = VImg::bytes(BK);
    constexpr uint64_t kFull = (BK < 64) ? ((1ull << BK) - 1ull) : ~0ull;
    constexpr int NKF = KB * KS;                // K fragments per tile
    constexpr int NVF = EB * 2 * KB;            // V fragments per tile

    // ---- feature switches, set by the register budget (256 VGPRs at QB = 1, 512 at QB = 2) --------
    // kPipe    : software-pipelined body (two score tiles live per query block)
    // kPrefetch: all LDS fragment reads of an interval issued up front (PFK / PFV fragments)
    // kDeep    : HBM loads run two intervals ahead of the LDS writes (two register sets)
    // kMfmaSum : row sums on the matrix pipe
    constexpr bool k16 = sizeof(T) == 2;
    constexpr bool kPipe = k16 && (E <= 64 || MODE == 0);      // E = 128 masked, pipelined: 716 B/lane of spills, 2x slower
    constexpr bool kPrefetch = NNOP_V_PREFETCH && k16 && (QB == 2 ? E <= 64 : (MODE == 0 && E <= 64));
    constexpr int  PFK = kPrefetch ? (NKF <= 8 ? NKF : 8) : 0;
    constexpr int  PFV = kPrefetch ? (NVF <= 8 ? NVF : 8) : 0;
    constexpr bool kDeep = NNOP_V_DEEP && kPipe && E <= 64 && (QB == 2 || (MODE == 0 && NW == 8));
    constexpr bool kMfmaSum = NNOP_V_MFMASUM && k16 && E <= 64 && (QB == 2 || MODE == 0);

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

#ifdef NNOP_DEV_BUILD
    // experiment (NNOP_FWD_STAGGER): de-phase co-resident workgroups.  HW_ID.TG_ID (bits 19:16) numbers the
    // workgroups resident on this CU; the odd one starts late so that its LDS / barrier phases fall into
    // the other's MFMA phase.  Timing only, never correctness.
    if (p.stagger > 0) {
        const unsigned tg = __builtin_amdgcn_s_getreg(4 | (16 << 6) | (3 << 11));
        if (tg & 1) for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(8);
    }
#endif

    // ---- which (batch, q-head, q-block) -------------------------------------------------
    int lin = kPair ? xcd_remap_heads((int)blockIdx.x, p.n_qblk, p.QH, p.n_wg / p.QH)
                    : xcd_remap_chunked((int)blockIdx.x, p.n_wg, p.n_qblk * (p.QH / p.KH));
    int qblk = lin % p.n_qblk;
    const int bh = lin / p.n_qblk;
    // causal: heaviest q-blocks first -- except (causal_alt) in every second column of this XCD's dispatch order: the workgroups that
    // share a CU at the same time then come from a descending and an ascending column, heavy beside light
    if (p.causal && !(p.causal_alt > 0 && (((((int)blockIdx.x >> 3) / p.causal_alt) & 1) != 0))) qblk = p.n_qblk - 1 - qblk;
    const int b   = bh / p.QH;
    const int qh  = bh - b * p.QH;
    const int kvh = qh / (p.QH / p.KH);                    // cld(q_head, n_q_per_kv), 0-based
    const int q0w = qblk * (WROWS * NW) + wave * WROWS;    // first query row of this wave
    int qi[QB];                                            // this lane's query row in block z
#pragma unroll
    for (int z = 0; z < QB; ++z) qi[z] = q0w + 32 * z + r;

    const T* __restrict__ qp = (const T*)p.q + ((size_t)bh * p.QL) * E;
    const char* __restrict__ kp = (const char*)((const T*)p.k + ((size_t)(b * p.KH + kvh) * p.KL) * E);
    const char* __restrict__ vp = (const char*)((const T*)p.v + ((size_t)(b * p.KH + kvh) * p.KL) * E);
    const uint8_t* __restrict__ mp = kGeneral && p.kpad ? p.kpad + (size_t)b * p.KL : nullptr;

    char* const kring = smem;
    char* const vring = smem + 2 * KBYTES;

    // ---- number of kv tiles this workgroup walks ---------------------------------------
    int n_tiles = (p.KL + BK - 1) / BK;
    if constexpr (kGeneral) {
        if (p.causal) {
            int q_last = qblk * (WROWS * NW) + WROWS * NW - 1;
            if (q_last > p.QL - 1) q_last = p.QL - 1;
            const int t_c = q_last / BK + 1;                   // keys <= q_last
            if (t_c < n_tiles) n_tiles = t_c;
        }
        if (mp) {
            // variable sequence length: one pass over the mask row builds the per-64-key validity words in LDS and
            // finds the last valid key; the walk stops after the tile holding it (none -> 0 tiles -> NaN rows)
            uint64_t* vbits = reinterpret_cast<uint64_t*>(smem + 2 * KBYTES + 2 * VBYTES + 16);
            int* slot = reinterpret_cast<int*>(smem + 2 * KBYTES + 2 * VBYTES);
            const int nk = n_tiles * BK < p.KL ? n_tiles * BK : p.KL;
            const int last = kpad_scan(mp, p.KL, nk, vbits, kMaxMaskTiles, slot, tid, NT);
            const int t_m = last / BK + 1;
            if (t_m < n_tiles) n_tiles = t_m;
        }
    }
    // tiles that are live for THIS wave (causal: up to the diagonal of its last row)
    int n_live = n_tiles;
    if (kGeneral && p.causal) {
        const int t_w = (q0w + WROWS - 1) / BK + 1;
        if (t_w < n_live) n_live = t_w;
    }

    // Leading run of tiles that are PLAIN for this wave: fully inside KL, every key valid, not clipped by the causal
    // diagonal of the wave's first row, live.  The pipelined loop below runs the plain-mode interval (one basic block,
    // no validity fetch, no branch) while tile t + 1 is still in that run and switches to the general interval from
    // there on -- once per wave instead of a decision per tile (measured on an all-valid mask: the general interval
    // alone is 1.45x slower per tile than plain mode at E = 64).  Every interval holds exactly one barrier, so waves of
    // a workgroup may switch at different tiles.
    int first_special = n_live;
    if constexpr (kGeneral) {
        if (p.causal) {
            const int t_c = (q0w + 1) / BK;                   // first tile the wave's first row does not fully see
            if (t_c < first_special) first_special = t_c;
        }
        if (p.KL / BK < first_special) first_special = p.KL / BK;          // ragged last tile
        if (mp) {
            // first 64-key validity word that is not all ones (words built by kpad_scan above)
            const uint64_t* vbits = reinterpret_cast<const uint64_t*>(smem + 2 * KBYTES + 2 * VBYTES + 16);
            int n_words = (first_special * BK + 63) >> 6;
            if (n_words > kMaxMaskTiles) n_words = kMaxMaskTiles;
            int first_bad = n_words;
            for (int base = 0; base < n_words; base += 64) {
                const int w = base + lane;
                const bool bad = w < n_words && vbits[w] != ~0ull;
                const uint64_t bm = __ballot(bad);
                if (bm) { first_bad = base + __builtin_ctzll(bm); break; }
            }
            const int t_bad = (first_bad << 6) / BK;
            if (t_bad < first_special) first_special = t_bad;
        }
    }
    const int plain_end = first_special > 1 ? ((first_special - 1) & ~1) : 0;   // intervals [0, plain_end), even

    // ---- Q fragments: B operand of S^T = K Q^T, straight from HBM into registers (raw: the scale is applied
    // in fp32 inside the exp argument -- pre-scaling Q in T was measured: no faster, and 10-50x less accurate
    // on large logits).
    const float c2 = p.scale * kLog2e;
    frag_t qf[QB][KS];
#pragma unroll
    for (int z = 0; z < QB; ++z) {
        const int qc = qi[z] < p.QL ? qi[z] : p.QL - 1;    // clamped for loads
        const T* qrow = qp + (size_t)qc * E;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[z][ks] = *reinterpret_cast<const frag_t*>(qrow + 16 * ks + 8 * h);
    }

    // ---- staging (pipelined: K runs ONE TILE AHEAD of V) -----------------------------------
    //   interval t (between two barriers) reads K(t+1) and V(t); at its end K(t+2) replaces K(t)
    //   and V(t+1) replaces V(t-1): a K ring of 2 and a V ring of 2.  With kDeep the HBM/L2 loads run
    //   one interval further ahead than the LDS writes (two register sets).
    Stager<T, E, BK, NT> sk0, sv0, sk1, sv1;
    auto stage = [&](Stager<T, E, BK, NT>& st, const char* base, int t) {
        if constexpr (kGeneral) st.load(base + (size_t)t * ((size_t)BK * E * sizeof(T)), p.KL - t * BK, tid);
        else st.load_full(base + (size_t)t * ((size_t)BK * E * sizeof(T)), tid);
    };

    const int vbase = VImg::lane_base(lane);

    f32x16 oacc[QB][EB];
#pragma unroll
    for (int z = 0; z < QB; ++z)
#pragma unroll
        for (int eb = 0; eb < EB; ++eb)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[z][eb][i] = 0.f;

    // Deferred-max online softmax.  m2 is the exponent REFERENCE (log2 units, per query row, shared by lanes r and
    // r+32): P = exp2(s*c2 - m2).  It is raised only when a row's max outgrows it by more than kThr (then
    // P <= 2^kThr: exact for the fp32 accumulation, inside fp16/bf16 range).  mt is the TRUE running row max, kept
    // because the residual contract wants it (ms = row max, src/attention.jl:128).
    constexpr float kThr = 8.0f;
    float m2[QB], mt[QB], lsum[QB];
    f32x16 lacc[QB];
#pragma unroll
    for (int z = 0; z < QB; ++z) {
        m2[z] = -INFINITY; mt[z] = -INFINITY; lsum[z] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) lacc[z][i] = 0.f;
    }
    frag_t ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = from_f32<T>(1.0f);

    // ---- LDS fragment reads.  With kPrefetch they are issued EARLY (top of an interval, pinned by a
    // sched_barrier) and consumed late; every fragment is shared by the wave's QB query blocks.
    auto kf_load = [&](const char* kimg, frag_t (&kf)[PFK > 0 ? PFK : 1]) {
#pragma unroll
        for (int f = 0; f < PFK; ++f) {
#if NNOP_ABL != 5
            kf[f] = KImg::read_row_frag(kimg, 32 * (f / KS) + r, h, f % KS);
#else
            kf[f] = qf[0][f % KS];
#endif
        }
    };
    auto vf_load = [&](const char* vimg, frag_t (&vf)[PFV > 0 ? PFV : 1]) {
#pragma unroll
        for (int f = 0; f < PFV; ++f) {
#if NNOP_ABL != 5
            vf[f] = VImg::read_col_frag(vimg + vbase, f % (2 * KB), f / (2 * KB));
#else
            vf[f] = qf[0][f % KS];
#endif
        }
    };
    // ---- X(t): S^T = K Q^T for kv tile t (raw units), all query blocks (MFMA) --------------------
    auto qk_tile = [&](const char* kimg, const frag_t (&kf)[PFK > 0 ? PFK : 1], f32x16 (&s)[QB][KB]) {
#if NNOP_V_SETPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int f = kb * KS + ks;
                frag_t a;
                if (f < PFK) a = kf[f < PFK ? f : 0];
                else a = KImg::read_row_frag(kimg, 32 * kb + r, h, ks);
#pragma unroll
                for (int z = 0; z < QB; ++z) {
                    if (ks == 0) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) s[z][kb][i] = 0.f;
                    }
#if NNOP_ABL != 4
                    s[z][kb] = mma16<T>(a, qf[z][ks], s[z][kb]);
#else
                    s[z][kb][ks] += (float)a[0];
#endif
                }
            }
        }
#if NNOP_V_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };
    // wave-uniform: which keys of tile t are valid (bounds + key padding); does it need masking
    auto tile_valid = [&](int t) -> uint64_t {
        uint64_t valid = kFull;
        if constexpr (kGeneral) {
            const int k0 = t * BK;
            if (k0 + BK > p.KL) valid &= (p.KL - k0 >= 64) ? ~0ull : ((1ull << (p.KL - k0)) - 1ull);
            if (mp) {
                if ((t * BK) >> 6 < kMaxMaskTiles) {
                    valid &= kpad_tile_bits<BK>(reinterpret_cast<const uint64_t*>(smem + 2 * KBYTES + 2 * VBYTES + 16), t);
                } else {                                   // sequences beyond 64K keys: read the mask per tile
                    const int kk = k0 + lane;
                    const bool lv = (lane < BK && kk < p.KL) ? (mp[kk] != 0) : false;
                    valid &= __ballot(lv);
                }
            }
        }
        return valid;
    };
    auto tile_needs_mask = [&](int t, uint64_t valid) {
        return kGeneral && (valid != kFull || (p.causal && t * BK + BK - 1 > q0w));
    };
    // mask (-> -inf) / bias tile t of query block z in place and return its row max in log2 units (both halves).
    // Plain / masked: logits stay in raw units; kPair: they become log2 units (s*c2 + pair*log2e).
    auto finish_x = [&](auto masked, int z, f32x16 (&s)[KB], int t, uint64_t valid) -> float {
        constexpr bool MASKED = decltype(masked)::value;
        const int k0 = t * BK;
        float mxp[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};      // 4 independent chains
        if constexpr (MASKED || kPair) {
            // pair bias [B][KL][QL][QH]: one 64-bit base per (tile, lane), 32-bit element offsets, addresses clamped
            // into the tensor (no divergent branch around the loads), masked-out logits dropped by the select below
            const T* pbase = nullptr;
            int kstride = 0, kmax = 0;
            if constexpr (kPair) {
                const int qc = qi[z] < p.QL ? qi[z] : p.QL - 1;
                kstride = p.QL * p.QH;                                      // elements between consecutive keys
                kmax = p.KL - 1 - k0;                                       // last in-range local key of this tile
                pbase = (const T*)p.pair + (((size_t)b * p.KL + k0) * p.QL + qc) * p.QH + qh;
            }
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const uint32_t w = (uint32_t)(valid >> (32 * kb + 4 * h));
                const int lim = qi[z] - k0 - 32 * kb - 4 * h;               // causal: local row <= lim
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int lr = (i & 3) + 8 * (i >> 2);
                    bool ok = true;
                    if constexpr (MASKED) {
                        ok = (w >> lr) & 1u;
                        if (p.causal) ok = ok && (lr <= lim);
                    }
                    float x = s[kb][i];
                    if constexpr (kPair) {
                        int kl = 32 * kb + lr + 4 * h;
                        kl = kl < kmax ? kl : kmax;
                        x = __builtin_fmaf(x, c2, to_f32(pbase[kl * kstride]) * kLog2e);
                    }
                    s[kb][i] = ok ? x : -INFINITY;
                    mxp[i & 3] = fmaxf(mxp[i & 3], s[kb][i]);
                }
            }
        } else {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int i = 0; i < 16; i += 2)
                    mxp[(i >> 1) & 3] = fmaxf(fmaxf(mxp[(i >> 1) & 3], s[kb][i]), s[kb][i + 1]);
        }
#if NNOP_ABL == 8
        return s[0][0];
#endif
        float mx = fmaxf(fmaxf(mxp[0], mxp[1]), fmaxf(mxp[2], mxp[3]));
        if constexpr (!kPair) mx *= c2;
        return half_swap_max(mx);
    };
    auto finish_all = [&](auto masked, f32x16 (&s)[QB][KB], float (&mx)[QB], int t, uint64_t valid) {
#pragma unroll
        for (int z = 0; z < QB; ++z) mx[z] = finish_x(masked, z, s[z], t, valid);
    };
    // Y(t): exp / convert / O^T += V^T P^T (+ row sums) for tile t, one 16-key step at a time so
    // that the exps of step kk+1 sit beside the MFMAs of step kk; V fragments shared by the blocks.
    auto softmax_pv = [&](f32x16 (&s)[QB][KB], const char* vimg, const frag_t (&vfp)[PFV > 0 ? PFV : 1]) {
        float lp[QB][4], msub[QB];
#pragma unroll
        for (int z = 0; z < QB; ++z) {
            msub[z] = (kGeneral && m2[z] == -INFINITY) ? 0.f : m2[z];     // a row that has seen no key yet: P = 0
#pragma unroll
            for (int c = 0; c < 4; ++c) lp[z][c] = 0.f;
        }
        const char* vb = vimg + vbase;
#pragma unroll
        for (int kk = 0; kk < 2 * KB; ++kk) {
            const int kb = kk >> 1, i0 = 8 * (kk & 1);
            frag_t pf[QB];
#pragma unroll
            for (int z = 0; z < QB; ++z) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
#if NNOP_ABL != 2
                    if constexpr (kPair) s[z][kb][i0 + j] = fast_exp2(s[z][kb][i0 + j] - msub[z]);
                    else s[z][kb][i0 + j] = fast_exp2(__builtin_fmaf(s[z][kb][i0 + j], c2, -msub[z]));
#endif
                    if constexpr (!kMfmaSum) lp[z][j & 3] += s[z][kb][i0 + j];
                }
                // P^T comes straight from the S^T accumulators (acc_frag): no LDS, no lane movement
                pf[z] = (kk & 1) ? acc_frag<T, 1>(s[z][kb]) : acc_frag<T, 0>(s[z][kb]);
                // row sums: ones[32 x 16] * P^T -> every accumulator row holds sum_k P^T[k][query]
#if NNOP_ABL != 7
                if constexpr (kMfmaSum) lacc[z] = mma16<T>(ones, pf[z], lacc[z]);
#endif
            }
#pragma unroll
            for (int eb = 0; eb < EB; ++eb) {
                const int f = eb * 2 * KB + kk;
                frag_t vf;
                if (f < PFV) vf = vfp[f < PFV ? f : 0];
                else vf = VImg::read_col_frag(vb, kk, eb);
#pragma unroll
                for (int z = 0; z < QB; ++z) {
#if NNOP_ABL != 3
                    oacc[z][eb] = mma16<T>(vf, pf[z], oacc[z][eb]);
#else
                    oacc[z][eb][kk] += (float)vf[0] * (float)pf[z][0];
#endif
                }
            }
        }
        if constexpr (!kMfmaSum) {
#pragma unroll
            for (int z = 0; z < QB; ++z) lsum[z] += (lp[z][0] + lp[z][1]) + (lp[z][2] + lp[z][3]);
        }
    };
    // Before tile t is exponentiated: `mxr` is its row max (log2 units).  Rare path: some row's max outgrew the
    // reference by > kThr (always at a row's first visible key, m2 = -inf) -> raise the reference.  Everything
    // accumulated at the old reference (O, l) is scaled exactly once; tile t has not been exponentiated yet.
    auto rescale = [&](f32x16 (&)[QB][KB], const float (&mxr)[QB]) {
        bool any = false;
#pragma unroll
        for (int z = 0; z < QB; ++z) {
            mt[z] = fmaxf(mt[z], mxr[z]);
            any = any || (mxr[z] > m2[z] + kThr);
        }
        if (__any(any)) {
#pragma unroll
            for (int z = 0; z < QB; ++z) {
                const bool up = mxr[z] > m2[z] + kThr;
                const float mn = up ? mxr[z] : m2[z];
                const float alpha = up ? fast_exp2(m2[z] - mn) : 1.f;      // m2 = -inf -> 0 (nothing accumulated yet)
#pragma unroll
                for (int eb = 0; eb < EB; ++eb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) oacc[z][eb][i] *= alpha;
                if constexpr (kMfmaSum) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) lacc[z][i] *= alpha;
                } else {
                    lsum[z] *= alpha;
                }
                m2[z] = mn;
            }
        }
    };

    if constexpr (!kPipe) {
        // ---- one tile per interval: K(t), V(t) in ring slot t&1; K(t+1), V(t+1) requested at the top
        // of interval t and written to the other slot at its end (one barrier per tile).
        if (n_tiles > 0) {
            stage(sk0, kp, 0);
            stage(sv0, vp, 0);
            sk0.template write<KImg, kGeneral>(kring, tid);
            sv0.template write<VImg, kGeneral>(vring, tid);
        }
#pragma unroll
        for (int z = 0; z < QB; ++z)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) landed(qf[z][ks]);
        __syncthreads();
        for (int t = 0; t < n_tiles; ++t) {
            const bool more1 = t + 1 < n_tiles;
            if (more1) { stage(sk0, kp, t + 1); stage(sv0, vp, t + 1); }
            if (t < n_live) {
                f32x16 sc[QB][KB];
                float mxr[QB];
                frag_t kfr[PFK > 0 ? PFK : 1], vfr[PFV > 0 ? PFV : 1];
                kf_load(kring + (t & 1) * KBYTES, kfr);
                vf_load(vring + (t & 1) * VBYTES, vfr);
                qk_tile(kring + (t & 1) * KBYTES, kfr, sc);
                const uint64_t vt = tile_valid(t);
                if (tile_needs_mask(t, vt)) finish_all(std::true_type{}, sc, mxr, t, vt);
                else finish_all(std::false_type{}, sc, mxr, t, vt);
                rescale(sc, mxr);
                softmax_pv(sc, vring + (t & 1) * VBYTES, vfr);
            }
            if (more1) {
                sk0.template write<KImg, kGeneral>(kring + ((t + 1) & 1) * KBYTES, tid);
                sv0.template write<VImg, kGeneral>(vring + ((t + 1) & 1) * VBYTES, tid);
            }
            __syncthreads();
        }
    } else {
        // ---- prologue: K(0), V(0), K(1) -> LDS; (kDeep: K(2), V(1) -> register set 1); X(0) ---------
        f32x16 sa[QB][KB], sb[QB][KB];
        float mxa[QB], mxb[QB];
#pragma unroll
        for (int z = 0; z < QB; ++z) { mxa[z] = -INFINITY; mxb[z] = -INFINITY; }
        if (n_tiles > 0) {
            stage(sk0, kp, 0);
            stage(sv0, vp, 0);
            if (n_tiles > 1) stage(sk1, kp, 1);
            sk0.template write<KImg, kGeneral>(kring, tid);
            sv0.template write<VImg, kGeneral>(vring, tid);
            if (n_tiles > 1) {
                sk1.template write<KImg, kGeneral>(kring + KBYTES, tid);
                if constexpr (kDeep) stage(sv1, vp, 1);
            }
            if constexpr (kDeep) {
                if (n_tiles > 2) stage(sk1, kp, 2);
            }
        }
#pragma unroll
        for (int z = 0; z < QB; ++z)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) landed(qf[z][ks]);
        __syncthreads();
        if (n_live > 0) {
            frag_t kf0[PFK > 0 ? PFK : 1];
            kf_load(kring, kf0);
            qk_tile(kring, kf0, sa);
            const uint64_t v0 = tile_valid(0);
            if (tile_needs_mask(0, v0)) finish_all(std::true_type{}, sa, mxa, 0, v0);
            else finish_all(std::false_type{}, sa, mxa, 0, v0);
        }
        __syncthreads();      // every wave has read K(0) before interval 0 ends by overwriting it

        // one interval: Y(t) on `sc` (row max `mxc` known) together with X(t+1) into `sn` / `mxn`.
        // (skl, svl): register set loaded this interval; (skw, svw): set written at its end.
        auto interval = [&](auto plain, int t, f32x16 (&sc)[QB][KB], const float (&mxc)[QB], f32x16 (&sn)[QB][KB],
                            float (&mxn)[QB], Stager<T, E, BK, NT>& skl, Stager<T, E, BK, NT>& svl,
                            Stager<T, E, BK, NT>& skw, Stager<T, E, BK, NT>& svw) {
            constexpr bool PLAIN = !kGeneral || decltype(plain)::value;
            const bool more1 = t + 1 < n_tiles, more2 = t + 2 < n_tiles, more3 = t + 3 < n_tiles;
#if NNOP_ABL != 6
            if constexpr (kDeep) {
                if (more3) stage(skl, kp, t + 3);
                if (more2) stage(svl, vp, t + 2);
            } else {
                if (more2) stage(skw, kp, t + 2);
                if (more1) stage(svw, vp, t + 1);
            }
#endif
            const char* knext = kring + ((t + 1) & 1) * KBYTES;
            const char* vcur = vring + (t & 1) * VBYTES;
            frag_t kfr[PFK > 0 ? PFK : 1], vfr[PFV > 0 ? PFV : 1];
            if constexpr (PLAIN) {
                rescale(sc, mxc);
                // ONE basic block: LDS fragment reads first, then QK^T(t+1) MFMAs | exp, convert (t)
                // | PV(t) MFMAs | row max (t+1).  Past the last tile the K ring holds a stale tile:
                // the result is never used.
                kf_load(knext, kfr);
                vf_load(vcur, vfr);
                __builtin_amdgcn_sched_barrier(0);
                qk_tile(knext, kfr, sn);
                softmax_pv(sc, vcur, vfr);
                finish_all(std::false_type{}, sn, mxn, t + 1, kFull);
            } else {
                if (t < n_live) {
                    rescale(sc, mxc);
                    if (t + 1 < n_live) {
                        const uint64_t vn = tile_valid(t + 1);
                        const bool nm = tile_needs_mask(t + 1, vn);
                        kf_load(knext, kfr);
                        vf_load(vcur, vfr);
                        __builtin_amdgcn_sched_barrier(0);
                        if (nm) {
                            qk_tile(knext, kfr, sn);
                            softmax_pv(sc, vcur, vfr);
                            finish_all(std::true_type{}, sn, mxn, t + 1, vn);
                        } else {
                            qk_tile(knext, kfr, sn);
                            softmax_pv(sc, vcur, vfr);
                            finish_all(std::false_type{}, sn, mxn, t + 1, vn);
                        }
                    } else {
                        vf_load(vcur, vfr);
                        softmax_pv(sc, vcur, vfr);
                    }
                }
            }
#if NNOP_ABL != 6
            if (more2) skw.template write<KImg, kGeneral>(kring + (t & 1) * KBYTES, tid);
            if (more1) svw.template write<VImg, kGeneral>(vring + ((t + 1) & 1) * VBYTES, tid);
#endif
#if NNOP_ABL != 1
            __syncthreads();
#endif
        };

        int t = 0;
        if constexpr (kGeneral) {
            // plain run of this wave (kDeep is a plain-mode-only feature: one register set here)
            for (; t < plain_end; t += 2) {
                interval(std::true_type{}, t, sa, mxa, sb, mxb, sk0, sv0, sk0, sv0);
                interval(std::true_type{}, t + 1, sb, mxb, sa, mxa, sk0, sv0, sk0, sv0);
            }
        }
        for (; t < n_tiles; t += 2) {
            if constexpr (kDeep) {
                interval(std::false_type{}, t, sa, mxa, sb, mxb, sk0, sv0, sk1, sv1);
                if (t + 1 < n_tiles) interval(std::false_type{}, t + 1, sb, mxb, sa, mxa, sk1, sv1, sk0, sv0);
            } else {        // one register set: loaded at the top of an interval, written at its end
                interval(std::false_type{}, t, sa, mxa, sb, mxb, sk0, sv0, sk0, sv0);
                if (t + 1 < n_tiles) interval(std::false_type{}, t + 1, sb, mxb, sa, mxa, sk0, sv0, sk0, sv0);
            }
        }
    }   // kPipe

    // ---- epilogue: normalise, store o, ms, ls -----------------------------------------------
#pragma unroll
    for (int z = 0; z < QB; ++z) {
        const float ltot = kMfmaSum ? lacc[z][0] : half_swap_sum(lsum[z]);
        const float inv = 1.0f / ltot;                     // ltot == 0 (no visible key) -> NaN rows,
                                                           // as the naive formula gives
        if (qi[z] < p.QL) {
            T* orow = (T*)p.o + ((size_t)bh * p.QL + qi[z]) * E;
#pragma unroll
            for (int eb = 0; eb < EB; ++eb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int e = 32 * eb + 8 * g + 4 * h;
                    if (e < E) {
                        f32x4 w = {oacc[z][eb][4 * g] * inv, oacc[z][eb][4 * g + 1] * inv,
                                   oacc[z][eb][4 * g + 2] * inv, oacc[z][eb][4 * g + 3] * inv};
                        if constexpr (sizeof(T) == 4) {
                            *reinterpret_cast<f32x4*>(orow + e) = w;
                        } else {
                            typedef T t4 __attribute__((ext_vector_type(4)));
                            *reinterpret_cast<t4*>(orow + e) = __builtin_convertvector(w, t4);
                        }
                    }
                }
            if (h == 0) {
                // Residual contract (src/attention.jl:128-129): ms = row max (natural-log units),
                // ls = sum exp(s - ms), both in T.  ms is rounded to T first and ls is expressed
                // relative to the ROUNDED ms, so the pair stays self-consistent in 16-bit types.
                const size_t so = (size_t)bh * p.QL + qi[z];
                const float m_nat = mt[z] * kLn2;
                const T m_t = from_f32<T>(m_nat);
                const float m_back = to_f32(m_t);
                float l_out = ltot;                        // sum exp2(x - m2) -> sum exp(s - ms)
                if (mt[z] != -INFINITY) l_out = ltot * fast_exp2(m2[z] - m_back * kLog2e);   // both finite here
                ((T*)p.ms)[so] = m_t;
                ((T*)p.ls)[so] = from_f32<T>(l_out);
            }
        }
    }
}

// LDS bytes the kernel needs: K ring of 2 + V ring of 2 + one scratch slot + (key padding) one 64-bit validity
// word per kv tile for up to kMaxMaskTiles tiles (longer sequences fall back to reading the mask per tile).
template <typename T, int E, int BK> constexpr int fa_fwd_lds_bytes() {
    return 2 * (RowImg<T, E>::bytes(BK) + ColImg<T, E>::bytes(BK)) + 16 + 8 * kMaxMaskTiles;
}

}  // namespace nnop
