// fa_bwd.hpp -- gfx950 flash-attention backward kernels (templates; instantiated per dtype in
// fa_bwd_*.hip).
//
// Computes what `∇flash_attention` computes (src/attention_bwd.jl:199-275) from the residuals
// (o, ms, ls) of the forward:
//     P  = exp(S - ms) / ls            S = scale*Q K^T (+pair), masked
//     dV = P^T dO                      dP = dO V^T          delta = rowsum(dO o)
//     dS = P o (dP - delta)            dpair = dS
//     dQ = scale dS K                  dK = scale dS^T Q
// (the reference folds 1/ls into dO -- "Δ_scaled", :183-188 -- and scale into dS, :118; same math).
//
// It is a different program from `_flash_attention_bwd!` (src/attention_bwd.jl:1-161):
//
//   reference                                        here
//   ---------------------------------------------    -------------------------------------------
//   ONE workgroup per (q-head, batch) walks all       two kernels, each with >= 256 workgroups:
//   kv-tile x q-tile pairs (<= QH*B CUs busy)           dkdv: workgroup = key block of a (batch,
//   dq/dk/dv read-modify-written in HBM every                 kv-head); dK^T, dV^T stay in MFMA
//   inner iteration; KA.@atomic for GQA (:99-103)             accumulators across the sweep over the
//   5 scalar-FMA LDS GEMMs + 9 barriers per pair              group's q-heads x q-tiles: no atomics,
//   Q, K rounded to Float16 even for fp32 (:19-20)            GQA included, deterministic
//   Δ_scaled materialised in HBM (:224)                 dq  : workgroup = query block; dQ^T in
//                                                             accumulators across the key sweep
//                                                     all five products on MFMA, fp32 accumulate,
//                                                     operands in the input dtype (no Float16 detour)
//                                                     1/ls and ms folded into ONE fp32 row constant
//                                                     that is the INITIAL ACCUMULATOR of S (so
//                                                     P = exp2(c*S') needs no subtraction), -delta
//                                                     the initial accumulator of dP
//
// S and dP are recomputed in both kernels (7 products instead of 5): the price of having no
// cross-workgroup dQ reduction (f32 atomics would bound the pass at ~1.3 TB/s of added bytes).
#pragma once
#include <type_traits>
#include "fa_common.hpp"
#include "pair_tile.hpp"

// 16-bit dQ kernel: up to which E the wave keeps its Q and dO fragments in registers (above: re-read from LDS every tile)
// dK/dV kernel at E = 128 (16-bit, plain / masked modes): K fragments in registers, V fragments from LDS (both in registers on
// top of the 128 registers of dK^T / dV^T accumulators: 36-124 spilled registers, measured).  0 = both from LDS (round 1).
#ifndef NNOP_DKDV_KREGS128
#define NNOP_DKDV_KREGS128 1
#endif

namespace nnop {

struct BwdParams {
    void *dq, *dk, *dv, *dpair;
    const void *d_o, *o, *ms, *ls, *q, *k, *v, *pair;
    const uint8_t* kpad;
    float* nl;       // [B][QH][QLs]  -(ms*log2e + log2(ls)) / (scale*log2e);  -inf for dead rows
    float* delta;    // [B][QH][QLs]  MINUS sum_e dO*o (the initial accumulator of dP); the plain-HIP kernels of fa_generic.hpp keep +delta here
    int   QL, KL, QH, KH, B, causal;
    int   QLs;       // row stride of nl / delta per (batch, q-head): QL rounded up to 64; the padding holds nl = -inf, delta = 0
    int   fused;     // 1: no preprocess launch -- the dQ kernel of fa_bwd_w64.hpp computes the row constants of its own rows (and
                     // writes `rcf` for the dK/dV kernel, which runs behind it)
    void* rcf;       // [B][QH][QLs][2][8] T, or null: nl and -delta once more, each as three 16-bit terms (hi, mid, lo, 0 ...) that sum to
                     // the fp32 value -- the form in which the dK/dV kernel of fa_bwd_w64.hpp feeds them to the matrix pipe
    int   n_blk;     // blocks along the workgroup's sequence axis
    int   n_wg;
    float scale;
    // MODE 3 (pair_tile.hpp): head-major scratch copies of the pair bias and the dS scratch, zero-padded to QLp x KLp
    const void* pair_a;
    void* dpair_s;
    int   QLp, KLp;
    int   persist_hx = 0;   // persistent form: heads of the column axis per XCD when they divide by 8 (fa_fwd.hpp), else 0
    int   causal_alt = 0;   // fa_bwd_dkdv_kernel / fa_bwd_dq_kernel, causal: g > 0 -- every second run of g consecutive blocks of an XCD's dispatch order runs its blocks the other way round (fa_launch.hpp causal_alt_run)
    int   persist = 0;   // fa_bwd_w64_kernel: blocks per workgroup of the persistent form (grid = 256 workgroups), 0 = one block per workgroup
};

// An fp32 row constant as three 16-bit terms (hi, mid, lo, 0 ...) whose sum is the value to ~24 bits: the form in which the dK/dV
// kernel of fa_bwd_w64.hpp feeds it to the matrix pipe (one extra contraction step against (1, 1, 1, 0 ...)).  -inf -> (-inf, 0, 0).
template <typename T> NNOP_DEV auto rc_split3(float x) {
    typedef T t8 __attribute__((ext_vector_type(8)));
    t8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = from_f32<T>(0.f);
    const T a = from_f32<T>(x);
    o[0] = a;
    const float r1 = x - to_f32(a);
    if (r1 == r1 && r1 - r1 == 0.f) {                    // finite remainder
        const T bb = from_f32<T>(r1);
        o[1] = bb;
        o[2] = from_f32<T>(r1 - to_f32(bb));
    }
    return o;
}

// -------------------------------------------------------------------------------------------------
// Preprocess (replaces _flash_attention_bwd_preprocess!, src/attention_bwd.jl:163-197).
// HBM-streaming: reads dO and o once (16 bytes per lane), writes two fp32 per query row.
// -------------------------------------------------------------------------------------------------
template <typename T, int E>
__global__ __launch_bounds__(256) void fa_bwd_pre_kernel(const BwdParams p, long long n_rows) {
    // n_rows counts PADDED rows (B * QH * QLs): the rows QL .. QLs-1 of every (batch, head) get nl = -inf, delta = 0, so that
    // the 64-row forms (fa_bwd_w64.hpp) can copy whole steps of row constants without a bounds test.
    constexpr int LPR = E / 8;                       // lanes per row, 8 elements per lane
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long prow = gid / LPR;
    const int sub = (int)(gid % LPR);
    const long long bh = (long long)((unsigned)prow / (unsigned)p.QLs);          // padded rows < 2^31 (launcher)
    const int qi = (int)(prow - bh * p.QLs);
    const bool live = prow < n_rows && qi < p.QL;
    const long long row = bh * p.QL + qi;            // row of the dense tensors
    float acc = 0.f;
    if (live) {
        typedef T t8 __attribute__((ext_vector_type(8)));
        const t8 a = *reinterpret_cast<const t8*>((const T*)p.d_o + row * E + sub * 8);
        const t8 b = *reinterpret_cast<const t8*>((const T*)p.o + row * E + sub * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += to_f32(a[j]) * to_f32(b[j]);
    }
#pragma unroll
    for (int off = LPR / 2; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if (prow < n_rows && sub == 0) {
        float nl = -INFINITY, dl = 0.f;
        if (live) {
            const float m = to_f32(((const T*)p.ms)[row]);
            const float l = to_f32(((const T*)p.ls)[row]);
            const float c2 = p.scale * kLog2e;
            nl = -(m * kLog2e + __builtin_amdgcn_logf(l)) / c2;   // v_log_f32 = log2
            dl = acc;
            if (!(l > 0.f) || !(nl == nl) || m == -INFINITY) {           // row with no visible key
                nl = -INFINITY;
                dl = 0.f;
            }
        }
        p.nl[prow] = nl;
        p.delta[prow] = -dl;                         // as the kernels consume it: dP' = dO V^T - delta
        if constexpr (sizeof(T) == 2) {
            if (p.rcf) {
                typedef T t8 __attribute__((ext_vector_type(8)));
                auto split3 = [](float x) -> t8 { return rc_split3<T>(x); };
                t8* dst = reinterpret_cast<t8*>(p.rcf) + 2 * prow;
                dst[0] = split3(nl);
                dst[1] = split3(-dl);
            }
        }
    }
}

// fp32 E = 256: the gradient accumulators of a 32-row wave (dK^T + dV^T: 256 registers, dQ^T: 128) do not fit beside the fragments and
// tiles -- every block runs as fa_bwd_split() workgroups that each recompute the score tiles (full E) and accumulate ONE slice of the
// columns (blockIdx.x / n_wg picks it), as the 16-bit E = 256 one-wave kernels do (fa_bwd_w64.hpp NSPLIT).  Two slices: 6 product-units
// for 4 (dK/dV), 5 for 3 (dQ), and twice the workgroups for the one-round launches this embedding dim mostly sees.
#ifndef NNOP_F32_E256_SPLIT
#define NNOP_F32_E256_SPLIT 2
#endif
template <typename T, int E> constexpr int fa_bwd_split() { return (sizeof(T) == 4 && E == 256) ? NNOP_F32_E256_SPLIT : 1; }

// store a transposed accumulator (rows = embedding in registers, column = this lane's sequence row); EBA column blocks from block eb0 on
template <typename T, int E, int EBA = (E + 31) / 32>
NNOP_DEV void store_acc_row(T* rowp, const f32x16* acc, float mul, int h, int eb0 = 0) {
    constexpr int EB = EBA;
    rowp += 32 * eb0;
#pragma unroll
    for (int eb = 0; eb < EB; ++eb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int e = 32 * eb + 8 * g + 4 * h;
            if (e < E) {                                   // (E < 32: the padded columns; a slice never reaches past E)
                f32x4 w = {acc[eb][4 * g] * mul, acc[eb][4 * g + 1] * mul, acc[eb][4 * g + 2] * mul,
                           acc[eb][4 * g + 3] * mul};
                if constexpr (sizeof(T) == 4) {
                    *reinterpret_cast<f32x4*>(rowp + e) = w;
                } else {
                    typedef T t4 __attribute__((ext_vector_type(4)));
                    *reinterpret_cast<t4*>(rowp + e) = __builtin_convertvector(w, t4);
                }
            }
        }
}

template <typename T, int E> struct BwdImgs {
    using Row = RowImg<T, E>;
    using Col = ColImg<T, E>;
    // a tile that is read both by rows and by columns: fp32 uses ONE swizzled image, the 16-bit
    // types a row image + a transposed-read image.
    static constexpr bool kDual = sizeof(T) == 4;
    static constexpr int both(int rows) { return kDual ? Row::bytes(rows) : Row::bytes(rows) + Col::bytes(rows); }
};

// -------------------------------------------------------------------------------------------------
// dK, dV.  Workgroup = 32*NW keys of one (batch, kv-head); each wave owns 32 keys (key on the lane)
// and keeps dK^T[e][key], dV^T[e][key] in accumulators while the workgroup sweeps the group's
// q-heads x q-tiles of BQ queries (double-buffered Q / dO images, one barrier per q-tile).
//   S  [q][key] = Q K^T   : A = Q rows (LDS),  B = K (registers / LDS)
//   dP [q][key] = dO V^T  : A = dO rows (LDS), B = V
//   dV^T += dO^T P        : A = dO columns (LDS transposed read), B = P  straight from accumulators
//   dK^T += Q^T dS        : A = Q columns,                        B = dS straight from accumulators
// -------------------------------------------------------------------------------------------------
// kSingle (E = 128, 16-bit): the streamed q-tile is single-buffered in LDS -- the next tile is prefetched into
// registers during the compute phase and written after a barrier -- which frees the LDS for 7-wave workgroups
// (224 keys: 112 KiB of K, V images + 32 KiB of tile), i.e. ~2 waves per SIMD instead of 1.
// (chosen by the launcher: NW == 7 <=> single-buffered)
// (fp32 E = 256: 1 KiB rows -- one wave's K, V images (64 KiB) + one Q, dO tile (64 KiB) is what 160 KiB hold)
template <typename T, int E, int NW> constexpr bool fa_bwd_single() { return (sizeof(T) == 2 && ((E > 64 && NW == 7) || E > 128)) || (sizeof(T) == 4 && E > 128); }

// dK/dV kernel: which of the workgroup's K / V fragments live in registers for the whole kernel (else: LDS row images)
// fp32 E = 128 (one wave per SIMD, 512 registers): K and V fragments (64 registers each) beside the 128 accumulator registers --
// with the images out of LDS a 4-wave workgroup (every SIMD of the CU) fits with double-buffered tiles; the 2-wave form with both
// images in LDS left half the SIMDs idle (24 % of the fp32 peak)
#ifndef NNOP_F32_E128_REGS
#define NNOP_F32_E128_REGS 1
#endif
template <typename T, int E> constexpr bool fa_bwd_f32_wide() { return NNOP_F32_E128_REGS && sizeof(T) == 4 && E == 128; }
// fp32 E = 256 (round 4; the plain-HIP kernels of fa_generic.hpp before: 1.3 TFLOP/s): the same form one size up -- K AND V fragments
// (dK/dV) / Q and dO fragments (dQ) in registers (2 x 128), so that LDS holds only the streamed tiles and a 4-wave workgroup fits, with
// the gradient accumulators column-split over fa_bwd_split() workgroups per block (below).  hipcc still spills ~90-130 registers there;
// measured 20-24 TFLOP/s at L2048-4096 H8 B2 (profiles/r04/f32_e256_bwd.log: 1 wave with everything in LDS 7.5, K in registers and 2 waves
// 8.3, this form 24.0 / 14.5 causal unsplit, 22.4 / 20.5 split in two).  NNOP_F32_E256_FORM: 0 / 1 / 2 = those three (A/B builds).
#ifndef NNOP_F32_E256_FORM
#define NNOP_F32_E256_FORM 2
#endif
template <typename T, int E> constexpr bool fa_bwd_f32_e256() { return sizeof(T) == 4 && E == 256; }
template <typename T, int E, int MODE> constexpr bool fa_bwd_dkdv_vregs() {
    return E <= 64 || (NNOP_F32_E128_REGS == 2 && fa_bwd_f32_wide<T, E>()) || (NNOP_F32_E256_FORM == 2 && fa_bwd_f32_e256<T, E>());
}
template <typename T, int E, int MODE, int NW = 8> constexpr bool fa_bwd_dkdv_kregs() {
    // E = 128: only the 8-wave form (the 4-wave masked kernel spills 25 registers with K on top and measured 2 % slower)
    return E <= 64 || (NNOP_DKDV_KREGS128 && sizeof(T) == 2 && E == 128 && MODE <= 1 && NW == 8) || fa_bwd_f32_wide<T, E>() ||
           (NNOP_F32_E256_FORM >= 1 && fa_bwd_f32_e256<T, E>());
}
// with K out of LDS the streamed tiles fit twice even at 7 / 8 waves
template <typename T, int E, int NW, int MODE> constexpr bool fa_bwd_dkdv_single() {
    return fa_bwd_single<T, E, NW>() && !(E == 128 && fa_bwd_dkdv_kregs<T, E, MODE, NW>());
}
template <typename T, int E, int NW, int BQ, int MODE>
constexpr int fa_bwd_dkdv_lds_bytes() {
    constexpr int nbuf = fa_bwd_dkdv_single<T, E, NW, MODE>() ? 1 : 2;
    return (fa_bwd_dkdv_kregs<T, E, MODE, NW>() ? 0 : RowImg<T, E>::bytes(32 * NW)) + (fa_bwd_dkdv_vregs<T, E, MODE>() ? 0 : RowImg<T, E>::bytes(32 * NW)) +
           nbuf * (2 * BwdImgs<T, E>::both(BQ) + 2 * BQ * 4);
}

// MODE 0 plain / 1 masked (causal, key padding) / 2 masked + pair bias and dpair  (as in fa_fwd.hpp)
template <typename T, int E, int NW, int BQ, int MODE>
__global__ __launch_bounds__(NW * 64, (sizeof(T) == 4 && E > 64) ? 1 : 2) void fa_bwd_dkdv_kernel(const BwdParams p) {
    constexpr bool kGeneral = MODE != 0;
    constexpr bool kPair = MODE >= 2;
    constexpr bool kStaged = MODE == 3;          // pair bias through head-major scratch + LDS tiles (pair_tile.hpp)
    using frag_t = typename Elem<T>::frag;
    using Imgs = BwdImgs<T, E>;
    using Row = typename Imgs::Row;
    using Col = typename Imgs::Col;
    constexpr int NT = NW * 64;
    constexpr int KS = E / 16;
    constexpr int EB = (E + 31) / 32;
    constexpr int QB = BQ / 32;
    constexpr bool kKRegs = fa_bwd_dkdv_kregs<T, E, MODE, NW>(), kVRegs = fa_bwd_dkdv_vregs<T, E, MODE>();
    constexpr int KIMG_B = kKRegs ? 0 : Row::bytes(32 * NW), VIMG_B = kVRegs ? 0 : Row::bytes(32 * NW);
    constexpr int QIMG = Imgs::both(BQ);
    constexpr bool kSingle = fa_bwd_dkdv_single<T, E, NW, MODE>();
    constexpr int BUF = kSingle ? 0 : 2 * QIMG + 2 * BQ * 4;      // distance between the two buffers (0: one buffer)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    constexpr int NSPLIT = fa_bwd_split<T, E>(), EBA = EB / NSPLIT;
    const int bid = NSPLIT > 1 ? (int)blockIdx.x % p.n_wg : (int)blockIdx.x;
    const int eb0 = NSPLIT > 1 ? ((int)blockIdx.x / p.n_wg) * EBA : 0;      // first column block this workgroup accumulates
    const int lin  = (kPair && !kStaged) ? xcd_remap_heads(bid, p.n_blk, p.KH, p.n_wg / p.KH)
                                         : xcd_remap_chunked(bid, p.n_wg, p.n_blk);      // chunk = one (batch, kv-head)
    int kblk = lin % p.n_blk;                  // (causal: block 0 sees every query -- ascending is heaviest first)
    if (p.causal && p.causal_alt > 0 && (((bid >> 3) / p.causal_alt) & 1) != 0) kblk = p.n_blk - 1 - kblk;
    const int bk   = lin / p.n_blk;
    const int b    = bk / p.KH;
    const int kvh  = bk - b * p.KH;
    const int rep  = p.QH / p.KH;
    const int k0wg = kblk * (32 * NW);
    const int kw0  = k0wg + wave * 32;
    const int key  = kw0 + r;
    const int key_c = key < p.KL ? key : p.KL - 1;
    const float c2 = p.scale * kLog2e;

    const T* __restrict__ kp = (const T*)p.k + ((size_t)(b * p.KH + kvh) * p.KL) * E;
    const T* __restrict__ vp = (const T*)p.v + ((size_t)(b * p.KH + kvh) * p.KL) * E;

    bool kvalid = key < p.KL;
    if (kGeneral && p.kpad && kvalid) kvalid = p.kpad[(size_t)b * p.KL + key] != 0;
    if constexpr (kGeneral) {
        // variable sequence length: a key block with no valid key gets dK = dV = 0 and does no work
        if (p.kpad && !__syncthreads_or(kvalid ? 1 : 0)) {
            if (key < p.KL) {
                f32x16 zero[EBA];
#pragma unroll
                for (int eb = 0; eb < EBA; ++eb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) zero[eb][i] = 0.f;
                const size_t ro = ((size_t)(b * p.KH + kvh) * p.KL + key) * E;
                store_acc_row<T, E, EBA>((T*)p.dk + ro, zero, 0.f, h, eb0);
                store_acc_row<T, E, EBA>((T*)p.dv + ro, zero, 0.f, h, eb0);
            }
            return;
        }
    }

    // ---- K, V fragments (B operands: k = embedding, column = key on the lane) ------------------
    frag_t kf[kKRegs ? KS : 1], vf[kVRegs ? KS : 1];
    char* kimg = smem;
    char* vimg = smem + KIMG_B;
    char* bufs = smem + KIMG_B + VIMG_B;
    if constexpr (kKRegs) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[ks] = *reinterpret_cast<const frag_t*>(kp + (size_t)key_c * E + 16 * ks + 8 * h);
    } else {
        Stager<T, E, 32 * NW, NT> sk;
        sk.load(kp + (size_t)k0wg * E, p.KL - k0wg, tid);
        sk.template write<Row>(kimg, tid);
    }
    if constexpr (kVRegs) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) vf[ks] = *reinterpret_cast<const frag_t*>(vp + (size_t)key_c * E + 16 * ks + 8 * h);
    } else {
        Stager<T, E, 32 * NW, NT> sv;
        sv.load(vp + (size_t)k0wg * E, p.KL - k0wg, tid);
        sv.template write<Row>(vimg, tid);
    }

    // ---- iteration space: (q-head of the group) x (q-tile) --------------------------------------
    const int n_qt = (p.QL + BQ - 1) / BQ;
    int qt0 = 0;
    if (kGeneral && p.causal) qt0 = k0wg / BQ;           // queries < first key of the block see none of it
    const int nqt = n_qt > qt0 ? n_qt - qt0 : 0;
    const int n_it = nqt * rep;

    Stager<T, E, BQ, NT> sq, sdo;
    // Row constants of the tile: nl (tid < BQ) or delta (BQ <= tid < 2BQ).  Loaded UNCONDITIONALLY from a clamped
    // address and turned into the stored value (sign, -inf / 0 for rows past QL) only in stage_write: a load inside a
    // divergent branch whose result merges with a constant makes the compiler wait for it at the end of the branch,
    // i.e. right behind the tile loads issued above -- the whole memory latency exposed once per q tile.
    float rowc_raw = 0.f;
    bool rowc_in = false;
    const float* __restrict__ rowc_src = tid < BQ ? p.nl : p.delta;
    const int rowc_rr = tid < BQ ? tid : (tid < 2 * BQ ? tid - BQ : 0);
    auto stage_load = [&](int it) {
        const int g = it / nqt, qt = qt0 + it - g * nqt;
        const int qh = kvh * rep + g;
        const size_t row0 = (size_t)(b * p.QH + qh) * p.QL + (size_t)qt * BQ;
        const size_t rc0 = (size_t)(b * p.QH + qh) * p.QLs + (size_t)qt * BQ;
        sq.load((const T*)p.q + row0 * E, p.QL - qt * BQ, tid);
        sdo.load((const T*)p.d_o + row0 * E, p.QL - qt * BQ, tid);
        const int last = p.QL - 1 - qt * BQ;               // last existing row of this tile (>= 0)
        rowc_in = rowc_rr <= last;
        rowc_raw = rowc_src[rc0 + (rowc_in ? rowc_rr : last)];
    };
    auto stage_write = [&](char* buf) {
        sq.template write<Row>(buf, tid);
        sdo.template write<Row>(buf + QIMG, tid);
        if constexpr (!Imgs::kDual) {
            sq.template write<Col>(buf + Row::bytes(BQ), tid);
            sdo.template write<Col>(buf + QIMG + Row::bytes(BQ), tid);
        }
        if (tid < 2 * BQ) {
            const float rowc = tid < BQ ? (rowc_in ? rowc_raw : -INFINITY) : (rowc_in ? rowc_raw : 0.f);
            reinterpret_cast<float*>(buf + 2 * QIMG)[tid] = rowc;
        }
    };

    f32x16 dka[EBA], dva[EBA];
#pragma unroll
    for (int eb = 0; eb < EBA; ++eb)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dka[eb][i] = 0.f; dva[eb][i] = 0.f; }

    const int cbase = Col::lane_base(lane);

    if (n_it > 0) {
        stage_load(0);
        stage_write(bufs);
    }
    if constexpr (kKRegs) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) landed(kf[ks]);
    }
    if constexpr (kVRegs) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) landed(vf[ks]);
    }
    __syncthreads();

    for (int it = 0; it < n_it; ++it) {
        char* cur = bufs + (it & 1) * BUF;
        char* nxt = bufs + ((it + 1) & 1) * BUF;
        const bool more = it + 1 < n_it;
        if (more) stage_load(it + 1);

        const int g = it / nqt, qt = qt0 + it - g * nqt;
        const int qh = kvh * rep + g;
        const char* qrow = cur;
        const char* dorow = cur + QIMG;
        const char* qcol = cur + Row::bytes(BQ) + cbase;
        const char* docol = cur + QIMG + Row::bytes(BQ) + cbase;
        const float* rc = reinterpret_cast<const float*>(cur + 2 * QIMG);

#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const int q0 = qt * BQ + 32 * qb;
            if (q0 >= p.QL) continue;
            if (kGeneral && p.causal && q0 + 31 < kw0) continue;       // block entirely above the diagonal

            typename PairTile<T>::Regs pregs;                              // kStaged: the bias tile, fetched ahead of the MFMAs
            if constexpr (kStaged) {
                // rows = this wave's 32 keys (its LANE axis), columns = queries q0 .. q0+31 (its register axis)
                const int krow = kw0 < p.KLp - 32 ? kw0 : p.KLp - 32;     // a wave past KL stays inside the scratch
                pregs = PairTile<T>::fetch((const T*)p.pair_a + (((size_t)b * p.QH + qh) * p.KLp + krow) * p.QLp + q0, (size_t)p.QLp, lane);
            }
            f32x16 s, dp;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(rc + 32 * qb + 8 * g4 + 4 * h);
                const f32x4 d = *reinterpret_cast<const f32x4*>(rc + BQ + 32 * qb + 8 * g4 + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) { s[4 * g4 + j] = a[j]; dp[4 * g4 + j] = d[j]; }
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                frag_t kfr, vfr;
                if constexpr (kKRegs) kfr = kf[ks];
                else kfr = Row::read_row_frag(kimg, 32 * wave + r, h, ks);
                if constexpr (kVRegs) vfr = vf[ks];
                else vfr = Row::read_row_frag(vimg, 32 * wave + r, h, ks);
                s  = mma16<T>(Row::read_row_frag(qrow, 32 * qb + r, h, ks), kfr, s);
                dp = mma16<T>(Row::read_row_frag(dorow, 32 * qb + r, h, ks), vfr, dp);
            }

            // P = exp2(c*S'), dS = P*dP'   (rows = queries in registers, key on the lane)
            const bool diag = kGeneral && p.causal && (q0 < kw0 + 31);
            // pair / dpair [B][KL][QL][QH]: one 64-bit base per (q-block, lane); rows are QH elements apart
            // (dpair is written by the dQ kernel, where the lane axis is the tensor's contiguous direction)
            const T* pbase = nullptr;
            int qmax = 0;
            float pv[16];                                                  // kStaged: this lane's 16 bias values
            if constexpr (kStaged) {
                PairTile<T>::unpack_rows(pregs, smem + fa_bwd_dkdv_lds_bytes<T, E, NW, BQ, MODE>() + wave * PairTile<T>::kBytes, lane, pv);
            } else if constexpr (kPair) {
                pbase = (const T*)p.pair + (((size_t)b * p.KL + key_c) * p.QL + q0) * p.QH + qh;
                qmax = p.QL - 1 - q0;                                      // last in-range local query row
            }
            // The key sits on the lane and every product contracts over QUERIES, so a masked-out key only ever
            // pollutes its own lane's dK / dV rows: key validity costs nothing here (those rows are zeroed before the
            // store).  Only the causal diagonal needs a per-element select, and only in diagonal blocks (wave-uniform).
            f32x16 ds;
            auto p_ds = [&](auto masked) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float x = s[i] * c2;
                    const int lrow = acc_row(i, h);
                    if constexpr (kStaged) {
                        x += pv[i] * kLog2e;
                    } else if constexpr (kPair) {
                        const int lr = lrow < qmax ? lrow : qmax;            // clamped: always inside the tensor
                        x += to_f32(pbase[lr * p.QH]) * kLog2e;
                    }
                    float pr = fast_exp2(x);
                    if constexpr (decltype(masked)::value) pr = (!diag || q0 + lrow >= key) ? pr : 0.f;
                    s[i] = pr;
                    ds[i] = pr * dp[i];
                }
            };
            if constexpr (kPair && !kStaged) {
                p_ds(std::true_type{});                       // one body: the pair gather is not duplicated
            } else if constexpr (kGeneral) {
                if (diag) p_ds(std::true_type{});
                else p_ds(std::false_type{});
            } else {
                p_ds(std::false_type{});
            }
            const frag_t p0 = acc_frag<T, 0>(s), p1 = acc_frag<T, 1>(s);
            const frag_t d0 = acc_frag<T, 0>(ds), d1 = acc_frag<T, 1>(ds);
#pragma unroll
            for (int eb = 0; eb < EBA; ++eb) {
                frag_t a0, a1, b0, b1;
                if constexpr (Imgs::kDual) {
                    a0 = Row::read_col_frag_f32(dorow, r, h, 2 * qb, eb0 + eb);
                    a1 = Row::read_col_frag_f32(dorow, r, h, 2 * qb + 1, eb0 + eb);
                    b0 = Row::read_col_frag_f32(qrow, r, h, 2 * qb, eb0 + eb);
                    b1 = Row::read_col_frag_f32(qrow, r, h, 2 * qb + 1, eb0 + eb);
                } else {
                    a0 = Col::read_col_frag(docol, 2 * qb, eb0 + eb);
                    a1 = Col::read_col_frag(docol, 2 * qb + 1, eb0 + eb);
                    b0 = Col::read_col_frag(qcol, 2 * qb, eb0 + eb);
                    b1 = Col::read_col_frag(qcol, 2 * qb + 1, eb0 + eb);
                }
                dva[eb] = mma16<T>(a0, p0, dva[eb]);
                dva[eb] = mma16<T>(a1, p1, dva[eb]);
                dka[eb] = mma16<T>(b0, d0, dka[eb]);
                dka[eb] = mma16<T>(b1, d1, dka[eb]);
            }
        }

        if constexpr (kSingle) __syncthreads();          // every wave is done reading the (only) buffer
        if (more) stage_write(nxt);
        __syncthreads();
    }

    if (key < p.KL) {
        if constexpr (kGeneral) {
            if (!kvalid) {                                  // padded-out key: its lane accumulated garbage (see above)
#pragma unroll
                for (int eb = 0; eb < EBA; ++eb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) { dka[eb][i] = 0.f; dva[eb][i] = 0.f; }
            }
        }
        const size_t ro = ((size_t)(b * p.KH + kvh) * p.KL + key) * E;
        store_acc_row<T, E, EBA>((T*)p.dk + ro, dka, p.scale, h, eb0);
        store_acc_row<T, E, EBA>((T*)p.dv + ro, dva, 1.0f, h, eb0);
    }
}

// -------------------------------------------------------------------------------------------------
// dQ.  Workgroup = 32*NW queries of one (batch, q-head); each wave owns 32 queries (query on the
// lane) and keeps dQ^T[e][query] in accumulators while the workgroup sweeps kv tiles of BK keys
// (double-buffered, one barrier per tile) -- the forward's structure with
//   S^T [key][q] = K Q^T   (init: nl[q])      dP^T [key][q] = V dO^T   (init: -delta[q])
//   dQ^T += K^T dS^T : A = K columns (LDS transposed read), B = dS^T straight from accumulators
// -------------------------------------------------------------------------------------------------
constexpr int kMaxMaskTilesBwd = 1024;       // key padding: one 64-bit validity word per kv tile (see fa_fwd.hpp)
// dQ kernel: does a wave keep its Q and dO fragments in registers for the whole kernel (else they are re-read from LDS row
// images every tile)?  16-bit: E <= 64 always; E = 128 in the plain / masked modes (252-256 registers, no spills; the
// pair-bias modes would spill ~50) -- which also frees 112 KiB of LDS: those kernels are double-buffered and run 8 waves.
template <typename T, int E, int MODE> constexpr bool fa_bwd_dq_qregs() {
    return sizeof(T) == 2 ? (E <= 64 || (E == 128 && MODE <= 1)) : (E <= 32 || fa_bwd_f32_wide<T, E>() || (NNOP_F32_E256_FORM == 2 && fa_bwd_f32_e256<T, E>()));
}
template <typename T, int E, int NW, int BK, int MODE>
constexpr int fa_bwd_dq_lds_bytes() {
    constexpr bool qdo_regs = fa_bwd_dq_qregs<T, E, MODE>();
    constexpr int nbuf = (fa_bwd_single<T, E, NW>() && !qdo_regs) ? 1 : 2;
    return (qdo_regs ? 0 : 2 * RowImg<T, E>::bytes(32 * NW)) +
           nbuf * (BwdImgs<T, E>::both(BK) + RowImg<T, E>::bytes(BK)) + 8 * kMaxMaskTilesBwd;   // + validity words
}

template <typename T, int E, int NW, int BK, int MODE>
__global__ __launch_bounds__(NW * 64, (sizeof(T) == 4 && E > 64) ? 1 : 2) void fa_bwd_dq_kernel(const BwdParams p) {
    constexpr bool kGeneral = MODE != 0;
    constexpr bool kPair = MODE >= 2;
    constexpr bool kStaged = MODE == 3;          // pair bias / dS through head-major scratch + LDS tiles (pair_tile.hpp)
    using frag_t = typename Elem<T>::frag;
    using Imgs = BwdImgs<T, E>;
    using Row = typename Imgs::Row;
    using Col = typename Imgs::Col;
    constexpr int NT = NW * 64;
    constexpr int KS = E / 16;
    constexpr int EB = (E + 31) / 32;
    constexpr int KB = BK / 32;
    constexpr bool kQRegs = fa_bwd_dq_qregs<T, E, MODE>();     // else Q, dO fragments come from LDS row images
    constexpr int QIMG = kQRegs ? 0 : Row::bytes(32 * NW);
    constexpr int KIMG = Imgs::both(BK);
    constexpr bool kSingle = fa_bwd_single<T, E, NW>() && !kQRegs;
    constexpr int BUF = kSingle ? 0 : KIMG + Row::bytes(BK);      // distance between the two buffers (0: one buffer)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    constexpr int NSPLIT = fa_bwd_split<T, E>(), EBA = EB / NSPLIT;
    const int bid = NSPLIT > 1 ? (int)blockIdx.x % p.n_wg : (int)blockIdx.x;
    const int eb0 = NSPLIT > 1 ? ((int)blockIdx.x / p.n_wg) * EBA : 0;      // first column block this workgroup accumulates
    int lin = (kPair && !kStaged) ? xcd_remap_heads(bid, p.n_blk, p.QH, p.n_wg / p.QH)
                                  : xcd_remap_chunked(bid, p.n_wg, p.n_blk * (p.QH / p.KH));
    int qblk = lin % p.n_blk;
    const int bh = lin / p.n_blk;
    if (p.causal && !(p.causal_alt > 0 && (((bid >> 3) / p.causal_alt) & 1) != 0)) qblk = p.n_blk - 1 - qblk;
    const int b = bh / p.QH;
    const int qh = bh - b * p.QH;
    const int kvh = qh / (p.QH / p.KH);
    const int q0wg = qblk * (32 * NW);
    const int q0w = q0wg + wave * 32;
    const int qi = q0w + r;
    const int qi_c = qi < p.QL ? qi : p.QL - 1;
    const float c2 = p.scale * kLog2e;

    const T* __restrict__ qp  = (const T*)p.q + ((size_t)bh * p.QL) * E;
    const T* __restrict__ dop = (const T*)p.d_o + ((size_t)bh * p.QL) * E;
    const T* __restrict__ kp = (const T*)p.k + ((size_t)(b * p.KH + kvh) * p.KL) * E;
    const T* __restrict__ vp = (const T*)p.v + ((size_t)(b * p.KH + kvh) * p.KL) * E;
    const uint8_t* __restrict__ mp = kGeneral && p.kpad ? p.kpad + (size_t)b * p.KL : nullptr;

    int n_tiles = (p.KL + BK - 1) / BK;
    if (kGeneral && p.causal) {
        int q_last = q0wg + 32 * NW - 1;
        if (q_last > p.QL - 1) q_last = p.QL - 1;
        const int t_c = q_last / BK + 1;
        if (t_c < n_tiles) n_tiles = t_c;
    }

    if constexpr (kGeneral) {
        if (mp) {
            // variable sequence length: validity words in LDS + stop after the tile holding the last valid key (as the
            // forward does); no valid key at all -> 0 tiles -> dQ = 0
            uint64_t* vbits = reinterpret_cast<uint64_t*>(smem + (fa_bwd_dq_lds_bytes<T, E, NW, BK, MODE>() - 8 * kMaxMaskTilesBwd));
            int* slot = reinterpret_cast<int*>(smem);
            const int nk = n_tiles * BK < p.KL ? n_tiles * BK : p.KL;
            const int last = kpad_scan(mp, p.KL, nk, vbits, kMaxMaskTilesBwd, slot, tid, NT);
            const int t_m = last / BK + 1;
            if (t_m < n_tiles) n_tiles = t_m;
            __syncthreads();                                   // the slot is reused by the LDS images below
        }
    }

    const float nlq = qi < p.QL ? p.nl[(size_t)bh * p.QLs + qi] : -INFINITY;
    const float ndl = qi < p.QL ? p.delta[(size_t)bh * p.QLs + qi] : 0.f;

    frag_t qf[kQRegs ? KS : 1], dof[kQRegs ? KS : 1];
    char* qimg = smem;
    char* doimg = smem + QIMG;
    char* bufs = smem + 2 * QIMG;
    if constexpr (kQRegs) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks]  = *reinterpret_cast<const frag_t*>(qp + (size_t)qi_c * E + 16 * ks + 8 * h);
            dof[ks] = *reinterpret_cast<const frag_t*>(dop + (size_t)qi_c * E + 16 * ks + 8 * h);
        }
    } else {
        Stager<T, E, 32 * NW, NT> s1, s2;
        s1.load(qp + (size_t)q0wg * E, p.QL - q0wg, tid);
        s2.load(dop + (size_t)q0wg * E, p.QL - q0wg, tid);
        s1.template write<Row>(qimg, tid);
        s2.template write<Row>(doimg, tid);
    }

    Stager<T, E, BK, NT> sk, sv;
    auto stage_load = [&](int t) {
        sk.load(kp + (size_t)t * BK * E, p.KL - t * BK, tid);
        sv.load(vp + (size_t)t * BK * E, p.KL - t * BK, tid);
    };
    auto stage_write = [&](char* buf) {
        sk.template write<Row>(buf, tid);
        if constexpr (!Imgs::kDual) sk.template write<Col>(buf + Row::bytes(BK), tid);
        sv.template write<Row>(buf + KIMG, tid);
    };

    f32x16 dqa[EBA];
#pragma unroll
    for (int eb = 0; eb < EBA; ++eb)
#pragma unroll
        for (int i = 0; i < 16; ++i) dqa[eb][i] = 0.f;
    const int cbase = Col::lane_base(lane);

    if (n_tiles > 0) {
        stage_load(0);
        stage_write(bufs);
    }
    if constexpr (kQRegs) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { landed(qf[ks]); landed(dof[ks]); }
    }
    {
        float a = nlq, b2 = ndl;
        landed(a); landed(b2);
    }
    __syncthreads();

    for (int t = 0; t < n_tiles; ++t) {
        char* cur = bufs + (t & 1) * BUF;
        char* nxt = bufs + ((t + 1) & 1) * BUF;
        const bool more = t + 1 < n_tiles;
        if (more) stage_load(t + 1);

        const int k0 = t * BK;
        bool skip = false, need_mask = false;
        uint64_t valid = ~0ull;
        if constexpr (kGeneral) {
            if (p.causal && k0 > q0w + 31) skip = true;
            if (BK < 64) valid = (1ull << BK) - 1ull;
            if (k0 + BK > p.KL) valid &= (p.KL - k0 >= 64) ? ~0ull : ((1ull << (p.KL - k0)) - 1ull);
            if (mp) {
                if ((t * BK) >> 6 < kMaxMaskTilesBwd) {
                    valid &= kpad_tile_bits<BK>(reinterpret_cast<const uint64_t*>(
                        smem + (fa_bwd_dq_lds_bytes<T, E, NW, BK, MODE>() - 8 * kMaxMaskTilesBwd)), t);
                } else {
                    const int kk = k0 + lane;
                    const bool lv = (lane < BK && kk < p.KL) ? (mp[kk] != 0) : false;
                    valid &= __ballot(lv);
                }
            }
            if (valid == 0ull) skip = true;
            need_mask = (valid != ((BK < 64) ? ((1ull << BK) - 1ull) : ~0ull)) ||
                        (p.causal && k0 + BK - 1 > q0w) || (kPair && !kStaged);
        }

        if (!skip) {
            const char* krow = cur;
            const char* kcol = cur + Row::bytes(BK) + cbase;
            const char* vrow = cur + KIMG;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                typename PairTile<T>::Regs pregs;                          // kStaged: the bias tile, fetched ahead of the MFMAs
                const int qcol = q0w < p.QLp - 32 ? q0w : p.QLp - 32;     // a wave past QL stays inside the scratch
                if constexpr (kStaged) {
                    // rows = keys k0 + 32 kb .. (register axis), columns = this wave's 32 queries (lane axis): copy A
                    pregs = PairTile<T>::fetch((const T*)p.pair_a + (((size_t)b * p.QH + qh) * p.KLp + k0 + 32 * kb) * p.QLp + qcol, (size_t)p.QLp, lane);
                }
                f32x16 s, dp;
#pragma unroll
                for (int i = 0; i < 16; ++i) { s[i] = nlq; dp[i] = ndl; }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    frag_t qfr, dofr;
                    if constexpr (kQRegs) { qfr = qf[ks]; dofr = dof[ks]; }
                    else {
                        qfr = Row::read_row_frag(qimg, 32 * wave + r, h, ks);
                        dofr = Row::read_row_frag(doimg, 32 * wave + r, h, ks);
                    }
                    s  = mma16<T>(Row::read_row_frag(krow, 32 * kb + r, h, ks), qfr, s);
                    dp = mma16<T>(Row::read_row_frag(vrow, 32 * kb + r, h, ks), dofr, dp);
                }
                const uint32_t w = (uint32_t)(valid >> (32 * kb + 4 * h));
                const int lim = qi - k0 - 32 * kb - 4 * h;
                const T* pbase = nullptr;
                T* dpbase = nullptr;
                int kstride = 0, kmax = 0;
                float pv[16];                                              // kStaged: this lane's 16 bias values
                char* ptile = nullptr;
                if constexpr (kStaged) {
                    ptile = smem + fa_bwd_dq_lds_bytes<T, E, NW, BK, MODE>() + wave * PairTile<T>::kBytes;
                    PairTile<T>::unpack(pregs, ptile, lane, pv);
                } else if constexpr (kPair) {
                    kstride = p.QL * p.QH;
                    kmax = p.KL - 1 - k0;
                    const size_t po = (((size_t)b * p.KL + k0) * p.QL + qi_c) * p.QH + qh;
                    pbase = (const T*)p.pair + po;
                    dpbase = (T*)p.dpair + po;
                }
                f32x16 ds;
                // masked / plain body chosen per tile (wave-uniform `need_mask`): a fully valid, unclipped tile pays no select
                auto p_ds = [&](auto masked) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float x = s[i] * c2;
                        if constexpr (kStaged) x += pv[i] * kLog2e;
                        bool ok = true;
                        if constexpr (decltype(masked)::value) {
                            const int lr = (i & 3) + 8 * (i >> 2);
                            ok = (w >> lr) & 1u;
                            if (p.causal) ok = ok && (lr <= lim);
                            if constexpr (kPair && !kStaged) {
                                int kl = 32 * kb + lr + 4 * h;
                                kl = kl < kmax ? kl : kmax;
                                x += to_f32(pbase[kl * kstride]) * kLog2e;
                            }
                        }
                        float pr = fast_exp2(x);
                        if constexpr (decltype(masked)::value) pr = ok ? pr : 0.f;
                        ds[i] = pr * dp[i];
                        if constexpr (kPair && !kStaged) {
                            // dpair = dS (the reference's dS / scale, src/attention_bwd.jl:123-132); lanes = consecutive queries
                            const int klr = 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;
                            if (klr <= kmax && qi < p.QL) dpbase[klr * kstride] = from_f32<T>(ds[i]);
                        }
                    }
                };
                if constexpr (kPair && !kStaged) {
                    p_ds(std::true_type{});                   // need_mask is always set with a pair bias
                } else if constexpr (kGeneral) {
                    if (need_mask) p_ds(std::true_type{});
                    else p_ds(std::false_type{});
                } else {
                    p_ds(std::false_type{});
                }
                if constexpr (kStaged) {
                    // dpair = dS (the reference's dS / scale, src/attention_bwd.jl:123-132) into the scratch matrix of this
                    // (batch, head); masked elements are exact zeros, tiles never visited are zero-filled by the unpack kernel
                    if (q0w < p.QL) {
                        T* g = PairTile<T>::kStoreLaneMajor
                                   ? (T*)p.dpair_s + (((size_t)b * p.QH + qh) * p.QLp + q0w) * p.KLp + k0 + 32 * kb
                                   : (T*)p.dpair_s + (((size_t)b * p.QH + qh) * p.KLp + k0 + 32 * kb) * p.QLp + q0w;
                        PairTile<T>::store(g, PairTile<T>::kStoreLaneMajor ? (size_t)p.KLp : (size_t)p.QLp, ptile, lane, ds);
                    }
                }
                const frag_t d0 = acc_frag<T, 0>(ds), d1 = acc_frag<T, 1>(ds);
#pragma unroll
                for (int eb = 0; eb < EBA; ++eb) {
                    frag_t a0, a1;
                    if constexpr (Imgs::kDual) {
                        a0 = Row::read_col_frag_f32(krow, r, h, 2 * kb, eb0 + eb);
                        a1 = Row::read_col_frag_f32(krow, r, h, 2 * kb + 1, eb0 + eb);
                    } else {
                        a0 = Col::read_col_frag(kcol, 2 * kb, eb0 + eb);
                        a1 = Col::read_col_frag(kcol, 2 * kb + 1, eb0 + eb);
                    }
                    dqa[eb] = mma16<T>(a0, d0, dqa[eb]);
                    dqa[eb] = mma16<T>(a1, d1, dqa[eb]);
                }
            }
        }

        if constexpr (kSingle) __syncthreads();          // every wave is done reading the (only) buffer
        if (more) stage_write(nxt);
        __syncthreads();
    }

    if (qi < p.QL) store_acc_row<T, E, EBA>((T*)p.dq + ((size_t)bh * p.QL + qi) * E, dqa, p.scale, h, eb0);
}

}  // namespace nnop
