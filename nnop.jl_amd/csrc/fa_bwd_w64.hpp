// fa_bwd_w64.hpp -- backward kernels in the forward's "one wave per SIMD" form (16-bit types, E = 64 / 128, no pair bias).
//
// What `_flash_attention_bwd!` computes (src/attention_bwd.jl:1-161), as the same two passes as fa_bwd.hpp (dK/dV with the key
// block stationary, dQ with the query block stationary: no atomics, GQA included, bitwise reproducible), but both passes are ONE
// kernel template here, because they are the same program with the roles of the two sequence axes exchanged:
//
//   a wave keeps "stationary" rows (KIND dK/dV: keys, KIND dQ: queries) for the whole kernel -- their fragments of the two
//   operand tensors (K, V / Q, dO) and their gradient accumulators (dK^T, dV^T / dQ^T), all in the ACCUMULATOR file -- while the
//   workgroup streams tiles of the other axis (Q, dO / K, V) through an LDS ring:
//
//     X products   S [t][s] = T1 rows . B1 (+ row constants)      dP [t][s] = T2 rows . B2 (- delta)
//                  A = a row fragment of the streamed tile (LDS, ds_read_b128), B = a stationary fragment (accumulator file):
//                  the streamed index lands in the accumulator REGISTERS, the stationary one on the LANE
//     softmax'     P = exp2(c S'),  dS = P dP'      (the forward's residuals are folded into S' and dP': no row max, no sum)
//     Y products   acc^T[e][s] += T^T[e][t] . F[t][s]   for (dV: T = dO, F = P), (dK: T = Q, F = dS)  /  (dQ: T = K, F = dS)
//                  A = a COLUMN fragment of the streamed tile (ds_read_b64_tr_b16), B = P / dS straight from the registers of the
//                  X products (accumulator-as-operand: no lane movement, no LDS)
//
//   * one wave per SIMD owning the whole 512-register file; every MFMA is inline asm with explicit register classes and the
//     loop body is HAND-PLACED slot by slot (sched_barrier pins), as in fa_fwd_w64.hpp -- hipcc neither splits the two files
//     that way nor interleaves asm MFMAs with VALU work on its own.
//   * the streamed tiles arrive by LDS-DMA into ONE dual-use image per tensor (DualImg: row reads and transposed column
//     reads both conflict-free, tools/dual_image.py), ring of 4 slots, a batch stays in flight for two iterations (counted
//     vmcnt), one barrier per iteration.
//   * three-stage software pipeline over the streamed steps: iteration u runs Y(u-1), then X(u+1), with the element-wise work of
//     step u dealt out over all of their MFMA slots by issue cost.  The three stages touch disjoint registers (two sets of
//     score tiles, two sets of P / dS fragments), so nothing inside an iteration waits for anything else inside it.
//   * wave shapes (ZS stationary 32-row blocks per wave x ZT streamed 32-row blocks per step): 2 x 1 up to E = 128 -- 64 stationary
//     rows per wave, so that each streamed fragment feeds two MFMAs.  dK/dV at E = 128 fills the accumulator file with accumulators
//     alone (256): its K fragments live in the arch VGPRs and its V fragments are a second fragment stream from an LDS image of the
//     workgroup's V rows (BwdW64Shape::kVLds).  E = 256: 1 x 1, and dK/dV as two workgroups per key block that each accumulate one
//     128-column half of dK^T / dV^T (NSPLIT).
//   * row constants: dQ has them per lane (fma(s, c2, nl2); -delta as dP's initial accumulator).  dK/dV has them per accumulator
//     REGISTER: they enter the score tiles through the matrix pipe, as one extra contraction step of (hi, mid, lo) 16-bit terms
//     against (1, 1, 1, 0 ...) -- "rcf".  When both passes run in this form the dQ kernel computes them in its prologue (no
//     preprocess launch) and the dK/dV kernel runs behind it.
//
// Modes: 0 plain / 1 masked (causal, key padding, ragged streamed length).  The pair-bias modes stay on fa_bwd.hpp.
#pragma once
#include "fa_bwd.hpp"
#include "fa_fwd_w64.hpp"

// diagnostic (make DEV=1 VAR=-DNNOP_BW64_STAMP=1; results WRONG by construction): s_memtime stamps, see tools/bw64_stamp.py
#if !defined(NNOP_DEV_BUILD)
#undef NNOP_BW64_STAMP
#endif
#ifndef NNOP_BW64_STAMP
#define NNOP_BW64_STAMP 0
#endif
#ifndef NNOP_BW64_LAG
#define NNOP_BW64_LAG 3
#endif
#ifndef NNOP_BW64_PF
#define NNOP_BW64_PF 3
#endif

namespace nnop {

enum : int { kBwdDKDV = 0, kBwdDQ = 1 };

// ---- DualImg: one LDS copy of a [rows][E] 16-bit tile for row reads AND transposed column reads -----------------------------
// Row-major with the 16-byte chunks of a row XOR-swizzled so that BOTH MFMA operand reads are bank-conflict free:
//   row read    (ds_read_b128, four 16-lane groups of distinct rows mod 16): x(row) must be a bijection of row & 15;
//   column read (ds_read_b64_tr_b16, per 32-lane half: 4 consecutive rows x 64 bytes): the four rows must land in four different
//               64-byte bank groups -- which the row swizzles of RowImg (row & 15, (row >> 1) & 7) do not give (4-way conflict).
// E = 128 (256-byte rows): x = ((row & 3) << 2) | ((row >> 2) & 3)      (the guide's layout (b)); E = 256: the same on the low 4 chunk bits
// E =  64 (128-byte rows, two rows per bank row): x = bit1 << 2 | bit3 << 1 | bit2 of row
// tools/dual_image.py restates every formula below on the CPU and checks data mapping and bank conflicts (tests/test_dual_image.py).
template <typename T, int E> struct DualImg {
    static_assert(sizeof(T) == 2 && (E == 64 || E == 128 || E == 256), "16-bit element types, E = 64, 128 or 256");
    static constexpr int kRowBytes = 2 * E;
    static constexpr int bytes(int rows) { return rows * kRowBytes; }
    NNOP_DEV static constexpr int xor_of(int row) {
        if constexpr (E >= 128) return ((row & 3) << 2) | ((row >> 2) & 3);      // rows of 256 bytes and more: the low 4 chunk bits
        else return (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 2) & 1);
    }
    // LDS-DMA: source byte (inside the dense tile) of the 16 bytes that land at image byte `o` (the destination is lane-linear)
    NNOP_DEV static int src_of(int o) {
        const int row = o / kRowBytes, phys = (o % kRowBytes) >> 4;
        return row * kRowBytes + ((phys ^ xor_of(row)) << 4);
    }
    // row read of lane (r, h), 32-row block zb, contraction step ks:  (row_lane_base ^ (ks << 5)) + zb * 32 * kRowBytes
    NNOP_DEV static int row_lane_base(int lane) {
        const int r = lane & 31, h = lane >> 5;
        return r * kRowBytes + ((xor_of(r) ^ h) << 4);
    }
    // column read, 16-row step kk, 32-column block eb, half s:  (col_lane_base ^ (eb << 6) ^ (s << 5)) + (16 kk + 8 s) * kRowBytes
    NNOP_DEV static int col_lane_base(int lane) {
        const int h = lane >> 5, g1 = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
        const int qx = E >= 128 ? q : (q >> 1);
        return (4 * h + q) * kRowBytes + 16 * (4 * qx + ((2 * g1 + (p >> 1)) ^ h)) + 8 * (p & 1);
    }
};

// ---- shapes -------------------------------------------------------------------------------------------------------------------
// NARROW: 32 stationary rows per wave x 64 streamed rows per step (1 x 2) at E = 64 / 128 -- 128-row workgroups, for launches whose
// 256-row blocks leave CUs idle (fa_bwd_inst.hpp bwd_w64_narrow): every streamed fragment feeds one MFMA instead of two.
template <int E, int KIND, int NARROW = 0> struct BwdW64Shape {
    static constexpr bool kDQ = KIND == kBwdDQ;
    static_assert(!NARROW || E == 64 || E == 128, "narrow form: E = 64 / 128");
    // E = 256: one block each way (the accumulator file holds 128 accumulator + 128 fragment registers of ONE 32-row block), and the
    // dK/dV pass runs as NSPLIT = 2 launches-in-one: each workgroup keeps full-E K / V fragments (S and dP contract over all of E)
    // but only half of the dK^T / dV^T accumulators -- 6 product-units instead of 4, spill-free.
    // dK/dV at E = 128 (kVLds): 64 keys per wave as at E = 64 -- every streamed fragment feeds two MFMAs, half the LDS read and LDS-DMA
    // traffic per MFMA of the 1 x 2 shape it replaces -- although their dK^T / dV^T accumulators are the whole accumulator file (256):
    // the K fragments live in the arch VGPRs (64) and the V fragments are read from an LDS image of the workgroup's V rows, as a
    // second fragment stream beside the dO rows.
    static constexpr bool kVLds = !kDQ && E == 128 && !NARROW;
    static constexpr int ZS = (E == 256 || NARROW) ? 1 : ((kDQ || E == 64 || kVLds) ? 2 : 1);       // stationary 32-row blocks per wave
    static constexpr int ZT = E == 256 ? 1 : 2 / ZS;                           // streamed 32-row blocks per step
    static constexpr int NSPLIT = (!kDQ && E == 256) ? 2 : 1;
    static constexpr int NYP = kDQ ? 1 : 2;                   // Y products
    static constexpr int KS = E / 16, EB = E / 32, EBA = EB / NSPLIT;          // EBA: 32-column blocks a workgroup accumulates
    static constexpr int NEL = 16 * ZS * ZT;                  // score elements per lane and step
    static constexpr int RT = 32 * ZT, SW = 32 * ZS, RB = 2 * E;
    static constexpr int IMG = RT * RB;                       // one streamed tile image
    // dK/dV: the row constants of a step travel as MFMA operand fragments (32 bytes per streamed row, see "row constants")
    static constexpr int RCM = kDQ ? 0 : 1;                   // extra MFMA per score tile
    static constexpr int RCB = kDQ ? 0 : RT * 32;
    static constexpr int SLOT = 2 * IMG + RCB, NS = 4;
    static constexpr int NJ = IMG / 4096;                     // LDS-DMA pieces per wave, tile and tensor
    static constexpr int NPB = 2 * NJ + RCM;                  // DMA instructions per wave and step
    static constexpr int NFY = NYP * 2 * ZT * EBA, NFX = 2 * ZT * KS, NF = NFY + NFX;    // A-fragment stream of an iteration
    static constexpr int TB = ZS * (KS + RCM);                // X slots per (product, zt): [row-constant MFMA,] KS steps, each x ZS
    static constexpr int NY = NFY * ZS, NX = 2 * ZT * TB, NSLOT = NY + NX;             // MFMA slots
    static constexpr int WG_ROWS = 4 * SW;
    // E = 64 (the wave's issue is the bound there): the XOR-ed lane bases of the fragment reads are computed once per iteration (KS row
    // bases in slot 1, 2 EBA column bases when phase Y has issued its last read) instead of one v_xor per read.  (At E = 128, where the
    // kernels are power-bound, the same change measured +-0.2 % and is not used.)
    static constexpr bool kBases = E == 64 && !kDQ;            // (dQ at E = 64 has too few reads per iteration for it to pay)
    static constexpr int VIMG = kVLds ? WG_ROWS * RB : 0;     // LDS image of the workgroup's V rows (behind the ring)
    static constexpr int NVB = kVLds ? ZS * KS : 0, PFB = 3, RFB = 4;      // V fragment stream of an iteration, its read-ahead / ring
    // dP slot of V fragment b = 2 ks + zs (relative to the iteration): behind phase Y, the S tiles and dP's row-constant MFMAs
    static constexpr int vb_slot(int b) { return NY + TB + ZS * RCM + b; }
    static_assert(IMG % 4096 == 0 && NJ >= 1 && NJ <= 4, "four waves x NJ pieces = one image; 12-bit immediate");
    static_assert(SLOT % (RB > 256 ? RB : 256) == 0, "XOR-addressed fragment reads: slot bases aligned to a row / 256 bytes");
};
template <typename T, int E, int KIND, int NARROW = 0> constexpr int fa_bwd_w64_lds_bytes(bool masked) {
    using SH = BwdW64Shape<E, KIND, NARROW>;
    return SH::NS * SH::SLOT + SH::VIMG + (masked ? (SH::kDQ ? 16 + 8 * kMaxMaskTiles : 16) : 0);
}

// ---- the slot plan ------------------------------------------------------------------------------------------------------------
// An iteration is NSLOT slots of one MFMA each: Y(u-1) first, then X(u+1).  The wave issues in order, so the time between two
// MFMAs ("gap") is max(32, 8 + issue cost of what lies between them): the fixed work (what follows MFMA g: row-constant fragment
// reads, LDS-DMA pieces and their address arithmetic; what precedes MFMA g+1: the barrier, the fragment read-ahead; behind the
// last one the iteration's bookkeeping) plus a share of the element-wise stream of step u.  The stream is a list of small items
// in a fixed order -- A(n): scale + v_exp_f32 of element n;  B(m), LAG elements behind: dS = P dP';  C(m) for odd m: the 16-bit
// converts of the pair it closes -- and nothing in it depends on anything else in the iteration, so it is dealt out by cost alone:
// the smallest per-gap budget for which a greedy in-order fill places everything (prices: tools/w64_gaps.py, calibrated on
// tools/ubench/gapcost.hip).  One constraint: no B item in gap 0 -- the dP tiles are written by the last MFMAs of the previous
// iteration, and a VALU read needs two MFMA slots of distance (tools/audit_w64.py checks the generated code).
template <int E, int KIND, bool MASKED, int LAG, int PF, int NARROW = 0> struct BwdW64Plan {
    using SH = BwdW64Shape<E, KIND, NARROW>;
    static constexpr int NSLOT = SH::NSLOT, NEL = SH::NEL, NITEM = NEL + NEL + NEL / 2;
    static constexpr int BAR_SLOT = (SH::NFY - PF) * SH::ZS;  // the barrier opens this slot (all column reads of Y are issued)
    int kind[NITEM] = {};                                     // 0 A, 1 B, 2 C
    int el[NITEM] = {};
    int it_end[NSLOT] = {};                                   // items [it_end[g-1], it_end[g]) go into gap g
    int cost[NSLOT] = {};
    int cap = 0;

    // X slot j (relative to NY): tile tq = (product, zt), step st (0 = the row-constant MFMA when RCM), block zs
    static constexpr bool x_is_rc(int j) { return SH::RCM && (j % SH::TB) / SH::ZS == 0; }
    static constexpr int x_frag(int j) { return (j / SH::TB) * SH::KS + (j % SH::TB) / SH::ZS - SH::RCM; }        // row fragment index
    static constexpr int rc_slot(int tq) { return SH::NY + tq * SH::TB; }                                       // its row-constant MFMA (zs = 0)
    static constexpr int rc_read_slot(int tq) { return rc_slot(tq) - 3 > BAR_SLOT ? rc_slot(tq) - 3 : BAR_SLOT; }
    static constexpr int dma_slot(int d) { return BAR_SLOT + 1 + 2 * d; }          // piece d of the batch: one per second slot
    // does slot s open with a fragment read-ahead, and of which stream position
    static constexpr int slot_frag(int s) {                   // stream position whose first MFMA slot s is, or -1
        if (s < SH::NY) return s % SH::ZS == 0 ? s / SH::ZS : -1;
        const int j = s - SH::NY;
        if (x_is_rc(j) || j % SH::ZS != 0) return -1;
        return SH::NFY + x_frag(j);
    }
    static constexpr int pre_cost(int s) {                    // in front of MFMA s
        int c = 0;
        if (s == BAR_SLOT) c += 8;
        const int f = slot_frag(s % NSLOT);
        if (f >= 0) c += ((f + PF) % SH::NF < SH::NFY ? 24 : 16) + 2 - (SH::kBases ? 4 : 0);   // two transposed reads / one ds_read_b128, xor, wait
        if (s == NSLOT) c += 24 + 18 + 16 + (MASKED ? 30 : 0);                     // ring rotation, loop control, image bases, mask test
        return c;
    }
    static constexpr int post_cost(int s) {                   // behind MFMA s (before the movable share)
        int c = 0;
        for (int tq = 0; tq < 2 * SH::ZT * SH::RCM; ++tq)
            if (s == rc_read_slot(tq)) c += 12;
        if (s == BAR_SLOT + 1) c += SH::kDQ ? 24 : 60;
        if (SH::kBases && s == 1) c += 4 * SH::KS;
        if (SH::kBases && s == SH::NY + 1) c += 8 * SH::EBA;
        for (int b = 0; b < SH::NVB; ++b)
            if (s == SH::vb_slot(b) - SH::PFB) c += 16;       // V fragment read-ahead (ds_read_b128 + xor)
        for (int d = 0; d < SH::NPB; ++d)
            if (s == dma_slot(d)) c += 40 + ((d % SH::NJ == 0 || d >= 2 * SH::NJ) ? 14 : 0);
        return c;
    }
    static constexpr int item_cost(int k) { return k == 0 ? 12 : (k == 1 ? 4 : 4 * SH::NYP); }
    constexpr void list() {
        int n = 0;
        for (int st = 0; st < NEL + LAG; ++st) {
            if (st < NEL) { kind[n] = 0; el[n++] = st; }
            const int m = st - LAG;
            if (m >= 0) {
                kind[n] = 1; el[n++] = m;
                if (m & 1) { kind[n] = 2; el[n++] = m; }
            }
        }
    }
    constexpr bool fill(int budget) {
        int n = 0;
        for (int g = 0; g < NSLOT; ++g) {
            int used = post_cost(g) + pre_cost(g + 1);
            while (n < NITEM && used + item_cost(kind[n]) <= budget && !(g == 0 && kind[n] != 0)) used += item_cost(kind[n++]);
            it_end[g] = n;
            cost[g] = used;
        }
        cap = budget;
        return n == NITEM;
    }
    static constexpr BwdW64Plan make() {
        BwdW64Plan pl{};
        pl.list();
        for (int b = 8; b <= 200; b += 2)
            if (pl.fill(b)) break;
        return pl;
    }
};

// 32 x E transposed accumulator tiles (rows = embedding in the registers, column = this lane's sequence row) -> one row of a
// [rows][E] tensor, 16-byte stores (lane halves paired with v_permlane32_swap, as the forward's epilogue)
// (the caller has fenced the accumulators: one fence_acc_result in front of the epilogue)
template <typename T, int EB> NNOP_DEV void store_acc_row16(T* rowp, f32x16 (&acc)[EB], float mul, int h, bool ok) {
#pragma unroll
    for (int eb = 0; eb < EB; ++eb) {
        acc_after_fence(acc[eb]);
        uint32_t pk[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            typedef T t4 __attribute__((ext_vector_type(4)));
            const f32x4 w = {acc[eb][4 * g] * mul, acc[eb][4 * g + 1] * mul, acc[eb][4 * g + 2] * mul, acc[eb][4 * g + 3] * mul};
            const u32x2 u = __builtin_bit_cast(u32x2, __builtin_convertvector(w, t4));
            pk[g][0] = u[0];
            pk[g][1] = u[1];
        }
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
            const u32x4 lo = {s0[0], s1[0], s0[1], s1[1]};
            if (ok) *reinterpret_cast<u32x4*>(rowp + 32 * eb + 8 * g + 8 * h) = lo;
        }
    }
}

template <typename T, int E, int KIND, int MODE, int NARROW = 0>
__global__ __launch_bounds__(256, 1) void fa_bwd_w64_kernel(const BwdParams p_arg) {
    using SH = BwdW64Shape<E, KIND, NARROW>;
    using frag_t = typename Elem<T>::frag;
    using Img = DualImg<T, E>;
    using MM = MfmaAsm<T>;
    constexpr bool kDQ = SH::kDQ, kGeneral = MODE != 0;
    constexpr int ZS = SH::ZS, ZT = SH::ZT, NYP = SH::NYP, KS = SH::KS, EB = SH::EBA;       // EB: the blocks THIS workgroup accumulates
    constexpr int RT = SH::RT, SW = SH::SW, RB = SH::RB, IMG = SH::IMG, SLOT = SH::SLOT, NS = SH::NS, NJ = SH::NJ, NPB = SH::NPB;
    constexpr int NFY = SH::NFY, NF = SH::NF, NY = SH::NY, NX = SH::NX, NSLOT = SH::NSLOT;
    constexpr int PF = NNOP_BW64_PF, RF = 4, LAG = NNOP_BW64_LAG;
    static_assert(NF % RF == 0 && PF < RF && PF < NFY, "static fragment ring; the barrier sits inside phase Y");

    extern __shared__ __attribute__((aligned(16))) char smem[];
#if NNOP_BW64_STAMP
    uint64_t stamp[6];
    stamp[0] = __builtin_amdgcn_s_memtime();
    stamp[1] = __builtin_amdgcn_s_memrealtime();
#endif
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);

    // ---- persistent form (p.persist = blocks per workgroup, masked-mode kernels; fa_fwd_w64.hpp has the reasoning and the list):
    // 256 workgroups, XCD x owns the (batch, head) columns [x C/8, (x+1) C/8), its blocks -- heaviest first inside a column: the last
    // query block for dQ, the first key block for dK/dV under the causal mask -- dealt out 32 at a time, alternately forwards and
    // backwards over the XCD's workgroups.  The parameters are re-read from the kernel-argument segment per block.
    constexpr bool kPersist = kGeneral && SH::NSPLIT == 1;
    const int n_steps_pers = (kPersist && p_arg.persist > 0) ? p_arg.persist : 1;
    for (int pstep = 0; pstep < n_steps_pers; ++pstep) {
    // lane indices are re-derived per block from a value the compiler cannot hoist: kept live across the register-full hand-placed
    // loop of the previous block they were spilled to scratch around it (E = 128)
    int lane0 = 0;
    if constexpr (kPersist) asm volatile("s_mov_b32 %0, 0" : "=s"(lane0));
    const int lane = kPersist ? (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (uint32_t)lane0)) : ((int)threadIdx.x & 63);
    const int tid = wave * 64 + lane;
    const int r = lane & 31, h = lane >> 5;
    BwdParams p_blk;
    if constexpr (!kPersist) p_blk = p_arg;
    if constexpr (kPersist) {
        p_blk.dpair = nullptr; p_blk.pair = nullptr; p_blk.pair_a = nullptr; p_blk.dpair_s = nullptr; p_blk.QLp = 0; p_blk.KLp = 0;
        typedef const BwdParams __attribute__((address_space(4))) * params_cp;
        params_cp pp = (params_cp)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(pp));
        p_blk.dq = pp->dq; p_blk.dk = pp->dk; p_blk.dv = pp->dv; p_blk.d_o = pp->d_o; p_blk.o = pp->o; p_blk.ms = pp->ms; p_blk.ls = pp->ls;
        p_blk.q = pp->q; p_blk.k = pp->k; p_blk.v = pp->v; p_blk.kpad = pp->kpad; p_blk.nl = pp->nl; p_blk.delta = pp->delta;
        p_blk.QL = pp->QL; p_blk.KL = pp->KL; p_blk.QH = pp->QH; p_blk.KH = pp->KH; p_blk.B = pp->B; p_blk.causal = pp->causal;
        p_blk.QLs = pp->QLs; p_blk.fused = pp->fused; p_blk.rcf = pp->rcf; p_blk.n_blk = pp->n_blk; p_blk.n_wg = pp->n_wg;
        p_blk.scale = pp->scale; p_blk.persist = pp->persist; p_blk.persist_hx = pp->persist_hx;
    }
    const BwdParams& p = p_blk;
    const float c2 = p.scale * kLog2e;
    const int rep = p.QH / p.KH;
    int pers_col = 0, pers_k = 0;      // persistent form: column inside the XCD's share, rank of the block inside the column (0 = heaviest)
    if (kPersist && p.persist > 0) {
        const int c = (int)blockIdx.x >> 3;
        const int pos = 32 * pstep + ((pstep & 1) ? 31 - c : c);
        pers_col = pos / p.n_blk;
        pers_k = pos - pers_col * p.n_blk;
    }

    // ---- which block; the stationary rows of this wave; the streamed sequence ------------------------------------------------
    int b, kvh, bh_s;                  // bh_s: (batch, head) index of the stationary tensors
    int s0wg;                          // first stationary row of the workgroup
    int SL, TL;                        // stationary / streamed sequence lengths
    int heads, u0 = 0, nps;            // streamed heads (dK/dV under GQA), first step, steps per head
    if constexpr (kDQ) {
        int blk;
        if (kPersist && p.persist > 0) {
            blk = p.n_blk - 1 - pers_k;
            if (p.persist_hx > 0) bh_s = (pers_col / p.persist_hx) * p.QH + ((int)blockIdx.x & 7) * p.persist_hx + pers_col % p.persist_hx;
            else bh_s = ((int)blockIdx.x & 7) * ((p.B * p.QH) >> 3) + pers_col;
        } else {
            const int lin = xcd_remap_chunked((int)blockIdx.x, p.n_wg, p.n_blk * rep);
            blk = lin % p.n_blk;
            bh_s = lin / p.n_blk;
            if (kGeneral && p.causal) blk = p.n_blk - 1 - blk;             // heaviest query blocks first
        }
        b = bh_s / p.QH;
        kvh = (bh_s - b * p.QH) / rep;
        s0wg = blk * SH::WG_ROWS;
        SL = p.QL; TL = p.KL; heads = 1;
        nps = (TL + RT - 1) / RT;
        if (kGeneral && p.causal) {
            int last = s0wg + SH::WG_ROWS - 1;
            if (last > SL - 1) last = SL - 1;
            const int t_c = last / RT + 1;
            if (t_c < nps) nps = t_c;
        }
    } else {
        // (E = 256: the grid holds every workgroup NSPLIT times; copy `esplit` accumulates the column blocks esplit * EB ..)
        int blk;
        if (kPersist && p.persist > 0) {
            blk = pers_k;                                                  // causal: the first key block sees every query
            if (p.persist_hx > 0) bh_s = (pers_col / p.persist_hx) * p.KH + ((int)blockIdx.x & 7) * p.persist_hx + pers_col % p.persist_hx;
            else bh_s = ((int)blockIdx.x & 7) * ((p.B * p.KH) >> 3) + pers_col;
        } else {
            const int lin = xcd_remap_chunked((int)blockIdx.x % p.n_wg, p.n_wg, p.n_blk);
            blk = lin % p.n_blk;
            bh_s = lin / p.n_blk;
        }
        b = bh_s / p.KH;
        kvh = bh_s - b * p.KH;
        s0wg = blk * SH::WG_ROWS;
        SL = p.KL; TL = p.QL; heads = rep;
        const int nst = (TL + RT - 1) / RT;
        if (kGeneral && p.causal) u0 = s0wg / RT < nst ? s0wg / RT : nst;   // queries in front of the block's first key see none of it
        nps = nst - u0;
    }
    const int esplit = SH::NSPLIT > 1 ? (int)blockIdx.x / p.n_wg : 0;
    const int s0w = s0wg + wave * SW;
    int sidx[ZS], sidx_c[ZS];
#pragma unroll
    for (int zs = 0; zs < ZS; ++zs) {
        sidx[zs] = s0w + 32 * zs + r;
        sidx_c[zs] = sidx[zs] < SL ? sidx[zs] : SL - 1;
    }

    const T* __restrict__ b1p;         // stationary operand of S  (dK/dV: K, dQ: Q)
    const T* __restrict__ b2p;         // stationary operand of dP (dK/dV: V, dQ: dO)
    const char* __restrict__ t1p;      // streamed tensor read by rows for S  and by columns for dK / dQ  (dK/dV: Q, dQ: K)
    const char* __restrict__ t2p;      // streamed tensor read by rows for dP and by columns for dV       (dK/dV: dO, dQ: V)
    uint32_t t_bytes;
    if constexpr (kDQ) {
        b1p = (const T*)p.q + (size_t)bh_s * p.QL * E;
        b2p = (const T*)p.d_o + (size_t)bh_s * p.QL * E;
        t1p = (const char*)((const T*)p.k + (size_t)(b * p.KH + kvh) * p.KL * E);
        t2p = (const char*)((const T*)p.v + (size_t)(b * p.KH + kvh) * p.KL * E);
        t_bytes = (uint32_t)p.KL * (uint32_t)RB;
    } else {
        b1p = (const T*)p.k + (size_t)bh_s * p.KL * E;
        b2p = (const T*)p.v + (size_t)bh_s * p.KL * E;
        const size_t g0 = (size_t)(b * p.QH + kvh * rep) * p.QL * E;       // the group's q-heads are contiguous
        t1p = (const char*)((const T*)p.q + g0);
        t2p = (const char*)((const T*)p.d_o + g0);
        t_bytes = (uint32_t)rep * (uint32_t)p.QL * (uint32_t)RB;           // < 4 GiB (launcher)
    }
    const int U = heads * nps;                                             // streamed steps of this workgroup

    // key padding, dK/dV: a key block with no valid key gets dK = dV = 0 and does no work; dQ: validity words in LDS
    const uint8_t* __restrict__ mp = kGeneral && p.kpad ? p.kpad + (size_t)b * p.KL : nullptr;
    bool svalid[ZS];
#pragma unroll
    for (int zs = 0; zs < ZS; ++zs) svalid[zs] = sidx[zs] < SL;
    uint64_t* const vbits = reinterpret_cast<uint64_t*>(smem + NS * SLOT + SH::VIMG + 16);
    int n_steps = U;
    if constexpr (kGeneral && !kDQ) {
        if (mp) {
            bool any = false;
#pragma unroll
            for (int zs = 0; zs < ZS; ++zs) {
                if (svalid[zs]) svalid[zs] = mp[sidx[zs]] != 0;
                any = any || svalid[zs];
            }
            // (no __syncthreads_or: its static LDS word would sit in front of the dynamic segment, whose base the XOR-addressed
            // fragment reads need 256-byte aligned)
            int* flag = reinterpret_cast<int*>(smem + NS * SLOT + SH::VIMG);
            if (tid == 0) *flag = 0;
            __syncthreads();
            if (any) *flag = 1;
            __syncthreads();
            if (*flag == 0) n_steps = 0;
        }
    }
    if constexpr (kGeneral && kDQ) {
        int* slot = reinterpret_cast<int*>(smem + NS * SLOT + SH::VIMG);
        const int nk = n_steps * RT < p.KL ? n_steps * RT : p.KL;
        if (mp) {
            const int last = kpad_scan(mp, p.KL, nk, vbits, kMaxMaskTiles, slot, tid, 256);
            const int t_m = last / RT + 1;
            if (t_m < n_steps) n_steps = t_m;
        } else {
            for (int w = tid; w * 64 < nk; w += 256) {
                const int left = p.KL - w * 64;
                vbits[w] = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
            }
            __syncthreads();
        }
    }

    n_steps = __builtin_amdgcn_readfirstlane(n_steps);      // workgroup-uniform by construction; the LDS reads above made it a VGPR

    // ---- dQ: the row constants of this wave's queries: nl = -(ms log2e + log2 ls) / c2 (-inf: no visible key) and -delta ----------
    // Either from the preprocess launch, or (p.fused: both passes run in this form) computed HERE, which saves that launch: the wave
    // holds the rows anyway -- delta = sum_e dO o over the lane pair (l, l + 32) that shares a row -- and writes them in fragment
    // form for the dK/dV kernel, which then runs BEHIND this one (src/attention_bwd.jl:163-197 is the reference's preprocess).
    // (fused: the loads are issued HERE, in front of the stationary fragments and the first LDS-DMA batches, and consumed behind
    // them -- computing right away would put a full memory latency in front of everything else the prologue has to fetch)
    float rc_nl[ZS], rc_nd[ZS];
    frag_t rc_do[kDQ ? ZS : 1][kDQ ? KS : 1], rc_o[kDQ ? ZS : 1][kDQ ? KS : 1];
    T rc_m[ZS], rc_l[ZS];
    if constexpr (kDQ) {
#pragma unroll
        for (int zs = 0; zs < ZS; ++zs) {
            const bool in = sidx[zs] < SL;
            const size_t row = (size_t)bh_s * p.QL + sidx_c[zs];
            const T* drow = (const T*)p.d_o + row * E;
            // the dO fragments: compiler-visible loads (they are both the stationary operand of dP -- moved to the accumulator file
            // below -- and, fused, one factor of delta: one fetch serves both)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) rc_do[zs][ks] = *reinterpret_cast<const frag_t*>(drow + 16 * ks + 8 * h);
            if (p.fused) {
                const T* orow = (const T*)p.o + row * E;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) rc_o[zs][ks] = *reinterpret_cast<const frag_t*>(orow + 16 * ks + 8 * h);
                rc_m[zs] = ((const T*)p.ms)[row];
                rc_l[zs] = ((const T*)p.ls)[row];
            } else {
                const size_t ro = (size_t)bh_s * p.QLs + sidx_c[zs];
                rc_nl[zs] = in ? p.nl[ro] : -INFINITY;
                rc_nd[zs] = in ? p.delta[ro] : 0.f;                                      // the workspace holds -delta
            }
        }
    }
    auto finish_rc = [&]() {
        if constexpr (kDQ) {
            if (!p.fused) return;
#pragma unroll
            for (int zs = 0; zs < ZS; ++zs) {
                float part = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int j = 0; j < 8; ++j) part += to_f32(rc_do[zs][ks][j]) * to_f32(rc_o[zs][ks][j]);
                const float dl = half_swap_sum(part);
                const float m = to_f32(rc_m[zs]), l = to_f32(rc_l[zs]);
                float nlv = -(m * kLog2e + __builtin_amdgcn_logf(l)) / c2;               // v_log_f32 = log2
                float ndv = -dl;
                if (!(l > 0.f) || !(nlv == nlv) || m == -INFINITY) { nlv = -INFINITY; ndv = 0.f; }
                if (!(sidx[zs] < SL)) { nlv = -INFINITY; ndv = 0.f; }
                if (p.rcf && sidx[zs] < p.QLs && h == 0) {              // rows QL .. QLs-1: the neutral padding
                    typedef T t8 __attribute__((ext_vector_type(8)));
                    t8* dst = reinterpret_cast<t8*>(p.rcf) + 2 * ((size_t)bh_s * p.QLs + sidx[zs]);
                    dst[0] = rc_split3<T>(nlv);
                    dst[1] = rc_split3<T>(ndv);
                }
                rc_nl[zs] = nlv;
                rc_nd[zs] = ndv;
            }
        }
    };
    if (n_steps == 0) finish_rc();                           // no loop: the fragment form is still owed to the dK/dV kernel
    // ---- accumulators, stationary fragments (accumulator file) ---------------------------------------------------------------
    f32x16 acc[NYP][ZS][EB];
#pragma unroll
    for (int y = 0; y < NYP; ++y)
#pragma unroll
        for (int zs = 0; zs < ZS; ++zs)
#pragma unroll
            for (int eb = 0; eb < EB; ++eb) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[y][zs][eb][i] = 0.f;
                // zeroed HERE: left to itself hipcc sinks the v_accvgpr_writes to the first use -- directly in front of the asm MFMA
                // that takes the tile as its C operand (VALU write -> MFMA read needs wait states; tools/audit_w64.py, E = 256)
                asm volatile("" : "+a"(acc[y][zs][eb]));
            }

    if (n_steps > 0) {
        const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
        // ragged streamed length: rows of the last step past the end are outside the descriptor's range.  Whatever the DMA does
        // with such a lane (zeros, or nothing), the ring must hold finite data there: they are multiplied by P = 0 / dS = 0.
        if ((TL & (RT - 1)) != 0) {
            for (int i = tid * 16; i < NS * SLOT; i += 256 * 16) *reinterpret_cast<u32x4*>(smem + i) = u32x4{0, 0, 0, 0};
            __syncthreads();
        }
        // ---- LDS-DMA: per-lane source offsets (the image's layout applied to the SOURCE), descriptors, step offsets -----------
        // wave w copies image bytes [NJ KiB * w, NJ KiB * (w + 1)) of both tensors; voff[j] = source byte of its chunk of piece j
        // MINUS 1024 j (the load's immediate adds it back on both sides).
        uint32_t voff[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) voff[j] = (uint32_t)(Img::src_of((wave * NJ + j) * 1024 + lane * 16) - j * 1024);
        const uint32_t wave_off = (uint32_t)(wave * NJ * 1024);
        const u32x4 rs1 = make_rsrc(t1p, t_bytes), rs2 = make_rsrc(t2p, t_bytes);
        // row constants (dK/dV): 32 bytes per streamed row of the workspace (p.rcf, [B][QH][QLs] rows; the padding rows hold
        // nl = -inf, delta = 0, so a row past QL gives P = 0 whatever its Q / dO rows hold), copied as they lie: 1 KiB = 32 rows per
        // piece, wave w copies piece w mod (RT / 32) (the same bytes twice: every wave issues the same number of loads, which is
        // what the counted vmcnt in front of the barrier needs)
        u32x4 rsc = rs1;
        uint32_t rc_voff = 0;
        if constexpr (!kDQ) {
            rsc = make_rsrc(p.rcf, (uint32_t)((size_t)p.B * p.QH * p.QLs * 32));
            rc_voff = (uint32_t)lane * 16u + (uint32_t)((wave % (RT / 32)) * 1024);
        }
        const uint32_t rc_dst_off = (uint32_t)(2 * IMG) + (uint32_t)((wave % (RT / 32 > 0 ? RT / 32 : 1)) * 1024);
        // running byte offsets of the NEXT step to copy (streamed tensors / row constants); dK/dV under GQA: the q-heads of the group
        // one after the other, each from its step u0.  Branch-free (a branch would split the hand-placed loop body) and NOT clamped
        // at the end of the stream: the copies past the last step are out of the descriptors' range (they read as zeros, or not
        // at all) or fetch the next group's rows, into slots whose score tiles nobody uses.
        int d_s = 0;
        uint32_t d_off = (uint32_t)u0 * (uint32_t)(RT * RB);
        uint32_t d_rc = kDQ ? 0u : ((uint32_t)(b * p.QH + kvh * rep) * (uint32_t)p.QLs + (uint32_t)(u0 * RT)) * 32u;
        const uint32_t jump_off = ((uint32_t)p.QL - (uint32_t)((nps - 1) * RT)) * (uint32_t)RB;      // last step of a head -> step u0 of the next
        const uint32_t jump_rc = ((uint32_t)p.QLs - (uint32_t)((nps - 1) * RT)) * 32u;
        auto advance_dma = [&]() {
            if constexpr (kDQ) {
                d_off += (uint32_t)(RT * RB);
            } else {
                const int s1 = d_s + 1;
                const bool wrap = s1 == nps;
                d_s = wrap ? 0 : s1;
                d_off += wrap ? jump_off : (uint32_t)(RT * RB);
                d_rc += wrap ? jump_rc : (uint32_t)(RT * 32);
            }
        };
        auto issue_step = [&](uint32_t slot) {          // slot: LDS byte address of the ring slot + this wave's share
            static_for<NJ>([&](auto jc) { constexpr int j = decltype(jc)::value; dma_piece<j, j == 0>(rs1, voff[j], d_off, slot); });
            static_for<NJ>([&](auto jc) { constexpr int j = decltype(jc)::value; dma_piece<j, j == 0>(rs2, voff[j], d_off, slot + IMG); });
            if constexpr (!kDQ) dma_piece<0, true>(rsc, rc_voff, d_rc, slot - wave_off + rc_dst_off);
        };
        // Ring slots as rotating scalars (each = slot address + wave_off):
        //   sY: step u-1 (column reads of phase Y), sM: step u, sX: step u+1 (row reads of phase X), sF: step u+2 (in flight),
        //   sD: where the batch issued behind this iteration's barrier goes (step u+3) = the slot of step u-1, which every wave has
        //       finished reading by then.  In iteration 0 there is no step -1: phase Y runs on zero fragments and reads the columns
        //       of step 0 (finite data; sY = sM), and the batch goes to the fourth slot.
        uint32_t sM = lds0 + wave_off, sX = sM + SLOT, sF = sM + 2 * SLOT, sD = sM + 3 * SLOT, sY = sM;
        auto rotate_slots = [&]() { sY = sM; sM = sX; sX = sF; sF = sD; sD = sY; };

        // ---- prologue: the stationary fragments first (X(0) needs them and step 0), then steps 0, 1, 2 ------------------------------
        constexpr bool kVLds = SH::kVLds;
        frag_t b1[ZS][KS], b2[kVLds ? 1 : ZS][kVLds ? 1 : KS];
#pragma unroll
        for (int zs = 0; zs < ZS; ++zs)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if constexpr (kVLds) {
                    b1[zs][ks] = *reinterpret_cast<const frag_t*>(b1p + (size_t)sidx_c[zs] * E + 16 * ks + 8 * h);      // arch VGPRs
                } else {
                    b1[zs][ks] = load_q_frag<frag_t>(b1p + (size_t)sidx_c[zs] * E + 16 * ks + 8 * h);
                    if constexpr (kDQ) b2[zs][ks] = rc_do[zs][ks];           // already requested (row constants above)
                    else b2[zs][ks] = load_q_frag<frag_t>(b2p + (size_t)sidx_c[zs] * E + 16 * ks + 8 * h);
                }
            }
        // kVLds: this wave's 64 V rows -> its quarter of the V image (DualImg row layout; 16 KiB = 16 pieces in groups of four per
        // M0).  Only this wave reads them, so its own vmcnt (the prologue's wait below) is all the synchronisation they need.  Rows
        // past KL are outside the descriptor: whatever lands there stays on the lanes of keys whose results are discarded.
        const uint32_t vimg0 = lds0 + (uint32_t)(NS * SLOT);
        if constexpr (kVLds) {
            const u32x4 rsv = make_rsrc(b2p, (uint32_t)SL * (uint32_t)RB);
            const uint32_t vsoff = (uint32_t)s0wg * (uint32_t)RB;
            static_for<16>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const uint32_t vo = (uint32_t)(Img::src_of((wave * 16 + j) * 1024 + lane * 16) - (j & 3) * 1024);
                dma_piece<(j & 3), (j & 3) == 0>(rsv, vo, vsoff, vimg0 + (uint32_t)(wave * 16384 + (j >> 2) * 4096));
            });
        }
        issue_step(sM);
        advance_dma();
        issue_step(sX);
        advance_dma();
        issue_step(sF);
        advance_dma();
        finish_rc();
        // dQ: the query's row constants are per lane: nl2 = c2 * nl (exponent offset), -delta as the initial accumulator of dP
        float nl2[ZS];
        f32x16 ndl[ZS];
        if constexpr (kDQ) {
#pragma unroll
            for (int zs = 0; zs < ZS; ++zs) {
                nl2[zs] = rc_nl[zs] * c2;                                  // nl = -inf (dead row / row past QL) stays -inf: c2 > 0
#pragma unroll
                for (int i = 0; i < 16; ++i) ndl[zs][i] = rc_nd[zs];
            }
            if constexpr (ZS == 2) fence_valu_operand(ndl[0], ndl[1]);
            else asm volatile("s_nop 1" : "+v"(ndl[0]));
        }
        // dK/dV: B operand of the row-constant MFMAs: (1, 1, 1, 0 ...) in lane half 0 (it sums the three 16-bit terms of the fp32
        // constant), zeros in lane half 1.  Opaque, and in the accumulator file like the other stationary operands.
        // (E = 256: the accumulator file is full -- 128 accumulator + 128 fragment registers -- and it lives in the arch VGPRs)
        constexpr bool kBonesV = E == 256 || kVLds;
        frag_t bones;
        if constexpr (!kDQ) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bones[j] = from_f32<T>((h == 0 && j < 3) ? 1.0f : 0.0f);
            if constexpr (kBonesV) asm volatile("s_nop 1" : "+v"(bones));
            else asm volatile("s_nop 1" : "+a"(bones));
        }
        auto rc_tile = [&](frag_t rc) -> f32x16 { if constexpr (kBonesV) return MM::qk_first_v(rc, bones); else return MM::qk_first(rc, bones); };

        // score tiles: two sets (roles swap every iteration), P / dS fragments as words: two sets
        f32x16 sA[ZS][ZT], dA[ZS][ZT], sB[ZS][ZT], dB[ZS][ZT];
        u32x4 fA[NYP][ZS][2 * ZT], fB[NYP][ZS][2 * ZT];
#pragma unroll
        for (int y = 0; y < NYP; ++y)
#pragma unroll
            for (int zs = 0; zs < ZS; ++zs)
#pragma unroll
                for (int kk = 0; kk < 2 * ZT; ++kk) { fA[y][zs][kk] = u32x4{0, 0, 0, 0}; fB[y][zs][kk] = u32x4{0, 0, 0, 0}; }

        // ---- LDS reads from integer addresses -----------------------------------------------------------------------------------
        typedef __attribute__((address_space(3))) const frag_t* lds_frag_p;
        typedef __attribute__((address_space(3))) s16x4* lds_tr_p;
        const uint32_t row_lane = (uint32_t)Img::row_lane_base(lane) - wave_off;     // ring scalars include wave_off
        const uint32_t col_lane = (uint32_t)Img::col_lane_base(lane) + (uint32_t)(esplit * EB * 64) - wave_off;   // + this split's first column block
        // row-constant fragments (dK/dV): streamed row 32 zt + r of the step, 32 bytes per row: [nl'] [-delta], each 8 elements of T
        // (hi, mid, lo of the fp32 value, 0 ...).  Lane half 0 takes the product's chunk; lane half 1 meets zeros of the B operand,
        // so it only has to read something finite: the delta chunk (nl may be -inf).
        const uint32_t rc_lane0 = (uint32_t)(2 * IMG + r * 32 + 16 * h) - wave_off;      // S:  h = 0 -> nl chunk, h = 1 -> delta chunk
        const uint32_t rc_lane1 = (uint32_t)(2 * IMG + r * 32 + 16) - wave_off;          // dP: delta chunk
        auto opaque = [](uint32_t x) { asm volatile("" : "+v"(x)); return x; };
        auto pin = [](auto& x) { asm volatile("" : "+v"(x)); };
        // row fragment g of phase X (g = (product * ZT + zt) * KS + ks) from the slot behind `ra` (= slot + row_lane)
        auto read_row = [&](uint32_t ra, int g) -> frag_t {
            const int prod = g / (ZT * KS), zt = (g / KS) % ZT, ks = g % KS;
            return *(lds_frag_p)(uintptr_t)((ra ^ (uint32_t)(ks << 5)) + (uint32_t)(prod * IMG + zt * 32 * RB));
        };
        // column fragment f of phase Y (f = (y * 2 ZT + kk) * EB + eb) from the slot behind `ca` (= slot + col_lane)
        auto read_col = [&](uint32_t ca, int f) -> frag_t {
            const int y = f / (2 * ZT * EB), kk = (f / EB) % (2 * ZT), eb = f % EB;
            const int img = kDQ ? 0 : (y == 0 ? IMG : 0);                  // dV: dO columns, dK: Q columns; dQ: K columns
            const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(uintptr_t)((ca ^ (uint32_t)(eb << 6)) + (uint32_t)(img + 16 * kk * RB)));
            const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(uintptr_t)((ca ^ (uint32_t)((eb << 6) | 32)) + (uint32_t)(img + (16 * kk + 8) * RB)));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 v8 = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
            return __builtin_bit_cast(frag_t, v8);
        };
        // the same reads from precomputed bases (kBases): rb[ks] = ra ^ (ks << 5), cb[2 eb + s] = ca ^ (eb << 6 | s << 5)
        auto read_row_b = [&](const uint32_t (&rb)[KS], int g) -> frag_t {
            const int prod = g / (ZT * KS), zt = (g / KS) % ZT, ks = g % KS;
            return *(lds_frag_p)(uintptr_t)(rb[ks] + (uint32_t)(prod * IMG + zt * 32 * RB));
        };
        auto read_col_b = [&](const uint32_t (&cb)[2 * EB], int f) -> frag_t {
            const int y = f / (2 * ZT * EB), kk = (f / EB) % (2 * ZT), eb = f % EB;
            const int img = kDQ ? 0 : (y == 0 ? IMG : 0);
            const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(uintptr_t)(cb[2 * eb] + (uint32_t)(img + 16 * kk * RB)));
            const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(uintptr_t)(cb[2 * eb + 1] + (uint32_t)(img + (16 * kk + 8) * RB)));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 v8 = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
            return __builtin_bit_cast(frag_t, v8);
        };
        // kVLds: V fragment b = 2 ks + zs (B operand of dP) from the V image: the row read of DualImg on this wave's rows
        const uint32_t v_lane = vimg0 + (uint32_t)(wave * 64 * RB + Img::row_lane_base(lane));
        auto read_vb = [&](int b) -> frag_t {
            const int ks = b / ZS, zs = b % ZS;
            return *(lds_frag_p)(uintptr_t)((v_lane ^ (uint32_t)(ks << 5)) + (uint32_t)(zs * 32 * RB));
        };
        // row constants of tile (product, zt) as an A fragment (see rc_lane0 / rc_lane1); rca = slot + rc_lane<product>
        auto read_rcf = [&](uint32_t rca, int zt) -> frag_t { return *(lds_frag_p)(uintptr_t)(rca + (uint32_t)(zt * 1024)); };

        // ---- masks (masked mode): keep-bits of one score tile per lane, bit lr <-> accumulator register i, lr = acc_row(i, 0) ----
        // dK/dV (causal): register row = query t0 + 32 zt + lr + 4 h, lane = key: keep iff query >= key
        // dQ: register row = key t0 + lr + 4 h, lane = query: keep iff the key is valid and key <= query (causal)
        uint64_t vword_next = 0;
        auto vword_fetch = [&](int u) { const int w = (u * RT) >> 6; vword_next = vbits[w < kMaxMaskTiles ? w : kMaxMaskTiles - 1]; };
        // (one bit per streamed key of a step: 32, or 64 in the narrow shape -- RT keys per step)
        using vword_t = std::conditional_t<(kDQ && RT == 64), uint64_t, uint32_t>;
        constexpr vword_t kAllValid = ~(vword_t)0;
        auto vword_take = [&](int u) -> vword_t {
            const uint64_t w = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(vword_next >> 32)) << 32) |
                               (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)vword_next);
            return (vword_t)(w >> ((u * RT) & 63));
        };
        auto apply_mask = [&](f32x16 (&s)[ZS][ZT], int t0, vword_t valid) {
#pragma unroll
            for (int zs = 0; zs < ZS; ++zs)
#pragma unroll
                for (int zt = 0; zt < ZT; ++zt) {
                    uint32_t m;
                    if constexpr (kDQ) {
                        const int lim = (p.causal ? sidx[zs] : 0x3fffffff) - t0 - 32 * zt - 4 * h;
                        const uint32_t cm = lim >= 31 ? ~0u : (lim < 0 ? 0u : ((2u << lim) - 1u));
                        m = (uint32_t)(valid >> (32 * zt + 4 * h)) & cm;
                    } else {
                        const int lim = sidx[zs] - t0 - 32 * zt - 4 * h;                 // keep iff lr >= lim
                        m = lim <= 0 ? ~0u : (lim >= 32 ? 0u : ~((1u << lim) - 1u));
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int lr = (i & 3) + 8 * (i >> 2);
                        s[zs][zt][i] = ((m >> lr) & 1u) ? s[zs][zt][i] : -INFINITY;
                    }
                }
        };
        // first streamed row of compute step u (dK/dV: a query index inside its head; dQ: a key index)
        int c_s = 0;                                                        // step inside the head, of the step being exponentiated
        auto step_t0 = [&]() -> int { return (u0 + c_s) * RT; };
        auto step_needs_mask = [&](int t0, vword_t valid) -> bool {
            if constexpr (kDQ) return valid != kAllValid || (p.causal && t0 + RT - 1 > s0w);
            else return p.causal && t0 < s0w + SW - 1;
        };

        // ---- the stationary fragments and step 0 have landed (every wave's pieces: barrier); steps 1 and 2 stay in flight -- the
        // barrier of iteration 0 waits for step 1 with the same counted wait as every other iteration -----------------------------------
        if constexpr (kVLds) {
#pragma unroll
            for (int zs = 0; zs < ZS; ++zs)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) landed(b1[zs][ks]);
            asm volatile("s_waitcnt vmcnt(%c[nfl])\n\ts_barrier" :: [nfl] "n"(2 * NPB) : "memory");
        } else if constexpr (KS == 16) {
            static_assert(ZS == 1, "");
            asm volatile("s_waitcnt vmcnt(%c[nfl])\n\ts_barrier"
                         : "+a"(b1[0][0]), "+a"(b1[0][1]), "+a"(b1[0][2]), "+a"(b1[0][3]), "+a"(b1[0][4]), "+a"(b1[0][5]), "+a"(b1[0][6]), "+a"(b1[0][7])
                         : [nfl] "n"(2 * NPB) : "memory");
            asm volatile("" : "+a"(b1[0][8]), "+a"(b1[0][9]), "+a"(b1[0][10]), "+a"(b1[0][11]), "+a"(b1[0][12]), "+a"(b1[0][13]), "+a"(b1[0][14]), "+a"(b1[0][15]) :: "memory");
            asm volatile("" : "+a"(b2[0][0]), "+a"(b2[0][1]), "+a"(b2[0][2]), "+a"(b2[0][3]), "+a"(b2[0][4]), "+a"(b2[0][5]), "+a"(b2[0][6]), "+a"(b2[0][7]) :: "memory");
            asm volatile("" : "+a"(b2[0][8]), "+a"(b2[0][9]), "+a"(b2[0][10]), "+a"(b2[0][11]), "+a"(b2[0][12]), "+a"(b2[0][13]), "+a"(b2[0][14]), "+a"(b2[0][15]) :: "memory");
        } else if constexpr (KS == 8 && ZS == 2) {
            asm volatile("s_waitcnt vmcnt(%c[nfl])\n\ts_barrier"
                         : "+a"(b1[0][0]), "+a"(b1[0][1]), "+a"(b1[0][2]), "+a"(b1[0][3]), "+a"(b1[0][4]), "+a"(b1[0][5]), "+a"(b1[0][6]), "+a"(b1[0][7]),
                           "+a"(b1[1][0]), "+a"(b1[1][1]), "+a"(b1[1][2]), "+a"(b1[1][3]), "+a"(b1[1][4]), "+a"(b1[1][5]), "+a"(b1[1][6]), "+a"(b1[1][7])
                         : [nfl] "n"(2 * NPB) : "memory");
            asm volatile(""
                         : "+a"(b2[0][0]), "+a"(b2[0][1]), "+a"(b2[0][2]), "+a"(b2[0][3]), "+a"(b2[0][4]), "+a"(b2[0][5]), "+a"(b2[0][6]), "+a"(b2[0][7]),
                           "+a"(b2[1][0]), "+a"(b2[1][1]), "+a"(b2[1][2]), "+a"(b2[1][3]), "+a"(b2[1][4]), "+a"(b2[1][5]), "+a"(b2[1][6]), "+a"(b2[1][7])
                         :: "memory");
        } else if constexpr (KS == 8) {
            asm volatile("s_waitcnt vmcnt(%c[nfl])\n\ts_barrier"
                         : "+a"(b1[0][0]), "+a"(b1[0][1]), "+a"(b1[0][2]), "+a"(b1[0][3]), "+a"(b1[0][4]), "+a"(b1[0][5]), "+a"(b1[0][6]), "+a"(b1[0][7]),
                           "+a"(b2[0][0]), "+a"(b2[0][1]), "+a"(b2[0][2]), "+a"(b2[0][3]), "+a"(b2[0][4]), "+a"(b2[0][5]), "+a"(b2[0][6]), "+a"(b2[0][7])
                         : [nfl] "n"(2 * NPB) : "memory");
        } else if constexpr (KS == 4 && ZS == 1) {
            asm volatile("s_waitcnt vmcnt(%c[nfl])\n\ts_barrier"
                         : "+a"(b1[0][0]), "+a"(b1[0][1]), "+a"(b1[0][2]), "+a"(b1[0][3]), "+a"(b2[0][0]), "+a"(b2[0][1]), "+a"(b2[0][2]), "+a"(b2[0][3])
                         : [nfl] "n"(2 * NPB) : "memory");
        } else {
            static_assert(KS == 4 && ZS == 2, "");
            asm volatile("s_waitcnt vmcnt(%c[nfl])\n\ts_barrier"
                         : "+a"(b1[0][0]), "+a"(b1[0][1]), "+a"(b1[0][2]), "+a"(b1[0][3]), "+a"(b1[1][0]), "+a"(b1[1][1]), "+a"(b1[1][2]), "+a"(b1[1][3]),
                           "+a"(b2[0][0]), "+a"(b2[0][1]), "+a"(b2[0][2]), "+a"(b2[0][3]), "+a"(b2[1][0]), "+a"(b2[1][1]), "+a"(b2[1][2]), "+a"(b2[1][3])
                         : [nfl] "n"(2 * NPB) : "memory");
        }

        // ---- X(0): the score tiles of step 0 (plain order; once per workgroup) --------------------------------------------------
        {
            const uint32_t ra = opaque(sM + row_lane);
            const uint32_t rca[2] = {opaque(sM + rc_lane0), opaque(sM + rc_lane1)};
#pragma unroll
            for (int prod = 0; prod < 2; ++prod)
#pragma unroll
                for (int zt = 0; zt < ZT; ++zt) {
                    if constexpr (!kDQ) {
                        const frag_t rc = read_rcf(rca[prod], zt);
#pragma unroll
                        for (int zs = 0; zs < ZS; ++zs) (prod ? dA : sA)[zs][zt] = rc_tile(rc);
                    }
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const frag_t a = read_row(ra, (prod * ZT + zt) * KS + ks);
#pragma unroll
                        for (int zs = 0; zs < ZS; ++zs) {
                            f32x16& d = (prod ? dA : sA)[zs][zt];
                            if constexpr (kVLds) {
                                const frag_t bq = prod ? read_vb(ks * ZS + zs) : b1[zs][ks];
                                MM::qk_acc_v(d, a, bq);
                            } else {
                                const frag_t& bq = (prod ? b2 : b1)[zs][ks];
                                if (kDQ && ks == 0) d = prod ? MM::qk_init(a, bq, ndl[zs]) : MM::qk_first(a, bq);
                                else MM::qk_acc(d, a, bq);
                            }
                        }
                    }
                }
            if constexpr (ZS == 2) {
                fence_mfma_result(sA[0][0], sA[1][0], dA[0][0], dA[1][0]);
            } else if constexpr (ZT == 2) {
                fence_mfma_result(sA[0][0], sA[0][1], dA[0][0], dA[0][1]);
            } else {
                asm volatile(NNOP_FENCE_128 : "+v"(sA[0][0]), "+v"(dA[0][0]));
            }
        }
        if constexpr (kGeneral && kDQ) vword_fetch(0);

        // fragments 0 .. PF-1 of the first iteration's stream: columns of "step -1" (= step 0 again, under zero fragments)
        frag_t fr[RF];
        {
            const uint32_t ca0 = opaque(sY + col_lane);
#pragma unroll
            for (int f = 0; f < PF; ++f) fr[f] = read_col(ca0, f);
        }

        constexpr bool kBases = SH::kBases;
        uint32_t cbase[2 * EB], rbase[KS];                  // (kBases) this iteration's column bases / row bases
        auto set_cbase = [&](uint32_t slot) {
            const uint32_t ca = opaque(slot + col_lane);
#pragma unroll
            for (int j = 0; j < 2 * EB; ++j) { cbase[j] = ca ^ (uint32_t)(((j >> 1) << 6) | ((j & 1) << 5)); pin(cbase[j]); }
        };
        if constexpr (kBases) set_cbase(sY);

        // ---- one iteration: Y(u-1) on the fragments `fp`, X(u+1) into (sn, dn), element-wise work of step u: (sc, dc) -> `fw` -----
        using Plan = BwdW64Plan<E, KIND, kGeneral, LAG, PF, NARROW>;
        auto iteration = [&](int u, f32x16 (&sc)[ZS][ZT], f32x16 (&dc)[ZS][ZT], f32x16 (&sn)[ZS][ZT], f32x16 (&dn)[ZS][ZT],
                             u32x4 (&fw)[NYP][ZS][2 * ZT], u32x4 (&fp)[NYP][ZS][2 * ZT]) {
            constexpr Plan plan = Plan::make();
            static_assert(plan.it_end[NSLOT - 1] == Plan::NITEM, "every item placed");
            static_assert(Plan::dma_slot(NPB - 1) < NSLOT, "the DMA batch fits behind the barrier");
            // masked mode: the tile of step u is masked before it is exponentiated (rare: diagonal blocks, ragged / padded tiles)
            if constexpr (kGeneral) {
                const int t0 = step_t0();
                vword_t valid = kAllValid;
                if constexpr (kDQ) {
                    valid = vword_take(u);
                    vword_fetch(u + 1);
                }
                if (__builtin_expect(step_needs_mask(t0, valid), 0)) apply_mask(sc, t0, valid);
            }
            const uint32_t cimg = opaque(sY + col_lane);               // step u-1: columns
            const uint32_t cimg2 = opaque(sM + col_lane);              // step u: the next iteration's first column fragments
            const uint32_t rimg = opaque(sX + row_lane);               // step u+1: rows
            uint32_t rcimg[2] = {0, 0};                                // step u+1: row-constant fragments
            if constexpr (!kDQ) { rcimg[0] = opaque(sX + rc_lane0); rcimg[1] = opaque(sX + rc_lane1); }
            frag_t rcf[2 * ZT];
            frag_t vb[kVLds ? SH::RFB : 1];                             // kVLds: V fragment ring (B operand of dP)
            uint32_t dst = 0, soff = 0, srow = 0;                      // this iteration's DMA batch

            // element n of step u: tile n / 16 = (zs, zt), accumulator register i = n % 16.
            // A: P = exp2(c S')    B: dS = P dP'    C (odd register): the 16-bit words of the pair it closes
            auto item = [&](auto kc) {
                constexpr int k = decltype(kc)::value, kind = plan.kind[k], n = plan.el[k];
                constexpr int t = n >> 4, zs = ZS == 2 ? t : 0, zt = ZS == 2 ? 0 : t, i = n & 15;
                if constexpr (kind == 0) {
                    float x;
                    if constexpr (kDQ) x = __builtin_fmaf(sc[zs][zt][i], c2, nl2[zs]);
                    else x = sc[zs][zt][i] * c2;
                    float e = fast_exp2(x);
                    pin(e);
                    sc[zs][zt][i] = e;
                } else if constexpr (kind == 1) {
                    float ds = sc[zs][zt][i] * dc[zs][zt][i];
                    pin(ds);
                    dc[zs][zt][i] = ds;
                } else {
                    typedef T t2 __attribute__((ext_vector_type(2)));
                    constexpr int kk = 2 * zt + (i >> 3), w = (i & 7) >> 1;
                    if constexpr (!kDQ) {                              // P first: its operands are the older ones
                        const f32x2 pw = {sc[zs][zt][i - 1], sc[zs][zt][i]};
                        uint32_t word2 = __builtin_bit_cast(uint32_t, __builtin_convertvector(pw, t2));
                        pin(word2);
                        fw[0][zs][kk][w] = word2;                      // P: dV
                    }
                    const f32x2 dsw = {dc[zs][zt][i - 1], dc[zs][zt][i]};
                    uint32_t word = __builtin_bit_cast(uint32_t, __builtin_convertvector(dsw, t2));
                    pin(word);
                    fw[NYP - 1][zs][kk][w] = word;                     // dS: the last Y product (dK / dQ)
                }
            };
            auto movable = [&](auto sc_) {
                constexpr int sl = decltype(sc_)::value;
                constexpr int k0 = sl ? plan.it_end[sl - 1] : 0, k1 = plan.it_end[sl];
                static_for<k1 - k0>([&](auto dk_) { item(std::integral_constant<int, k0 + decltype(dk_)::value>{}); });
            };
            // fragment read PF ahead of stream position f (wraps into the next iteration's column fragments)
            auto read_ahead = [&](auto fc) {
                constexpr int g = decltype(fc)::value + PF;
                if constexpr (kBases) {
                    // cbase: the columns of step u-1 until phase Y has issued its last read (slot NY + 1 re-bases it to step u)
                    if constexpr (g < NFY) fr[g % RF] = read_col_b(cbase, g);
                    else if constexpr (g < NF) fr[g % RF] = read_row_b(rbase, g - NFY);
                    else fr[g % RF] = read_col_b(cbase, g - NF);
                } else {
                    if constexpr (g < NFY) fr[g % RF] = read_col(cimg, g);
                    else if constexpr (g < NF) fr[g % RF] = read_row(rimg, g - NFY);
                    else fr[g % RF] = read_col(cimg2, g - NF);
                }
            };

            static_for<NSLOT>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int f = Plan::slot_frag(i);                  // >= 0: this slot opens stream position f
                if constexpr (i == Plan::BAR_SLOT) {
                    // every wave's pieces of step u+1 (issued two barriers ago) have landed; one batch (step u+2) stays in flight.
                    // Behind the barrier every wave is done with the columns of step u-1: its slot takes step u+3.
                    static_assert(NPB <= 15, "vmcnt literal");
                    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(NPB) : "memory");
                }
                if constexpr (f >= 0) read_ahead(std::integral_constant<int, f>{});
                if constexpr (i < NY) {
                    constexpr int fy = i / ZS, zs = i % ZS;
                    constexpr int y = fy / (2 * ZT * EB), kk = (fy / EB) % (2 * ZT), eb = fy % EB;
                    MM::pv_acc(acc[y][zs][eb], fr[fy % RF], __builtin_bit_cast(frag_t, fp[y][zs][kk]));
                } else {
                    constexpr int j = i - NY, tq = j / SH::TB, prod = tq / ZT, zt = tq % ZT, zs = (j % SH::TB) % ZS;
                    f32x16& d = (prod ? dn : sn)[zs][zt];
                    if constexpr (Plan::x_is_rc(j)) {
                        d = rc_tile(rcf[tq]);                           // the tile starts as its row constants
                    } else {
                        constexpr int g = Plan::x_frag(j), ks = g % KS;
                        if constexpr (kVLds) {
                            if constexpr (prod == 0) MM::qk_acc_v(d, fr[(NFY + g) % RF], b1[zs][ks]);
                            else MM::qk_acc_v(d, fr[(NFY + g) % RF], vb[(ks * ZS + zs) % SH::RFB]);
                        } else {
                            const frag_t& bq = (prod ? b2 : b1)[zs][ks];
                            if constexpr (kDQ && ks == 0) d = prod ? MM::qk_init(fr[(NFY + g) % RF], bq, ndl[zs]) : MM::qk_first(fr[(NFY + g) % RF], bq);
                            else MM::qk_acc(d, fr[(NFY + g) % RF], bq);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!kDQ) {
                    static_for<2 * ZT>([&](auto tc) {
                        constexpr int tq = decltype(tc)::value;
                        if constexpr (i == Plan::rc_read_slot(tq)) rcf[tq] = read_rcf(rcimg[tq / ZT], tq % ZT);
                    });
                }
                if constexpr (kVLds) {
                    static_for<SH::NVB>([&](auto bc) {
                        constexpr int b = decltype(bc)::value;
                        if constexpr (i == SH::vb_slot(b) - SH::PFB) vb[b % SH::RFB] = read_vb(b);
                    });
                }
                if constexpr (kBases && i == 1) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) { rbase[ks] = rimg ^ (uint32_t)(ks << 5); pin(rbase[ks]); }
                }
                if constexpr (kBases && i == NY + 1) set_cbase(sM);     // every column read of step u-1 has been issued (slot (NFY - PF) ZS)
                if constexpr (i == Plan::BAR_SLOT + 1) {
                    dst = sD;
                    soff = d_off;
                    srow = d_rc;
                    advance_dma();
                    asm volatile("" : "+s"(d_off));
                }
                static_for<NPB>([&](auto dc_) {
                    constexpr int d = decltype(dc_)::value;
                    if constexpr (i == Plan::dma_slot(d)) {
                        if constexpr (d < NJ) dma_piece<d, d == 0>(rs1, voff[d < NJ ? d : 0], soff, dst);
                        else if constexpr (d < 2 * NJ) dma_piece<d - NJ, d == NJ>(rs2, voff[d < 2 * NJ ? d - NJ : 0], soff, dst + IMG);
                        else dma_piece<0, true>(rsc, rc_voff, srow, dst - wave_off + rc_dst_off);
                    }
                });
                movable(std::integral_constant<int, i>{});
                __builtin_amdgcn_sched_barrier(0);
            });
            rotate_slots();
            if constexpr (kGeneral) {
                if constexpr (kDQ) ++c_s;
                else { if (++c_s == nps) c_s = 0; }
            }
        };

        // a wave idles out its last MFMA before it leaves a copy of the loop body (see fa_fwd_w64.hpp: register-allocator copies of
        // accumulator tiles on the exit edges)
        auto leave_fence = []() { asm volatile(NNOP_FENCE_128 ::: "memory"); __builtin_amdgcn_sched_barrier(0); };       // (nothing is scheduled across: the exit edge's register copies stay behind the idle time)
#if NNOP_BW64_STAMP
        stamp[2] = __builtin_amdgcn_s_memtime();
        stamp[3] = __builtin_amdgcn_s_memrealtime();
#endif
        // iterations u = 0 .. n_steps (the last one runs Y(n_steps - 1) beside work on a copy of the last step that nobody uses)
        // dQ under the causal mask: a wave whose last query precedes the keys of the workgroup's later steps stops computing there
        // (the waves of a 256-query block see 64 .. 256 more keys than its first query) and only keeps the workgroup's DMA / barrier
        // schedule going: one barrier and one batch per remaining iteration, as every live iteration has.
        int n_live = n_steps;
        if constexpr (kDQ && kGeneral) {
            if (p.causal) {
                const int t_w = (s0w + SW - 1) / RT + 1;
                if (t_w < n_live) n_live = t_w;
            }
        }
        int u = 0;
        const int n_it = n_live + 1;
        if (n_it >= 2) {
            for (;;) {
                iteration(u, sA, dA, sB, dB, fA, fB);
                iteration(u + 1, sB, dB, sA, dA, fB, fA);
                u += 2;
                if (u + 1 >= n_it) {
                    leave_fence();
                    break;
                }
            }
        }
        if (u < n_it) {
            iteration(u, sA, dA, sB, dB, fA, fB);
            leave_fence();
            ++u;
        }
        if constexpr (kDQ && kGeneral) {
            for (; u <= n_steps; ++u) {
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(NPB) : "memory");
                issue_step(sD);
                advance_dma();
                rotate_slots();
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if NNOP_BW64_STAMP
        stamp[4] = __builtin_amdgcn_s_memtime();
        stamp[5] = __builtin_amdgcn_s_memrealtime();
#endif
    }

    // ---- epilogue: store the gradient rows of this wave's stationary rows ---------------------------------------------------------
    fence_acc_result(acc[0][0][0]);                         // the one fence of the epilogue (fa_fwd_w64.hpp, "RULE")
    // (persistent form: the lane's row indices are derived AGAIN here -- kept live across the register-full loop they are what hipcc
    // spills around it)
    int lane1 = 0;
    if constexpr (kPersist) asm volatile("s_mov_b32 %0, 0" : "=s"(lane1));
    const int lane_e = kPersist ? (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (uint32_t)lane1)) : lane;
    const int h_e = lane_e >> 5;
#pragma unroll
    for (int zs = 0; zs < ZS; ++zs) {
        const int sidx_e = kPersist ? s0w + 32 * zs + (lane_e & 31) : sidx[zs];
        const int sidx_ce = sidx_e < SL ? sidx_e : SL - 1;
        const bool in = sidx_e < SL;
        if constexpr (kDQ) {
            T* row = (T*)p.dq + ((size_t)bh_s * p.QL + sidx_ce) * E + esplit * EB * 32;
            store_acc_row16<T, EB>(row, acc[0][zs], p.scale, h_e, in);
        } else {
            if constexpr (kGeneral) {
                if (!svalid[zs]) {                           // padded-out key: its lane accumulated garbage (key on the lane)
#pragma unroll
                    for (int y = 0; y < NYP; ++y)
#pragma unroll
                        for (int eb = 0; eb < EB; ++eb) {
                            acc_after_fence(acc[y][zs][eb]);
#pragma unroll
                            for (int i = 0; i < 16; ++i) acc[y][zs][eb][i] = 0.f;
                        }
                }
            }
            const size_t ro = ((size_t)bh_s * p.KL + sidx_ce) * E + esplit * EB * 32;
            store_acc_row16<T, EB>((T*)p.dv + ro, acc[0][zs], 1.0f, h_e, in);
            store_acc_row16<T, EB>((T*)p.dk + ro, acc[1][zs], p.scale, h_e, in);
        }
    }
#if NNOP_BW64_STAMP
    if (tid == 0) {
        uint64_t* dbg = kDQ ? reinterpret_cast<uint64_t*>((T*)p.dq + ((size_t)bh_s * p.QL + s0w) * E)
                            : reinterpret_cast<uint64_t*>((T*)p.dk + ((size_t)bh_s * p.KL + s0w) * E);
        for (int i = 0; i < 6; ++i) dbg[i] = stamp[i];
        dbg[6] = (uint64_t)n_steps;
    }
#endif
    // the next block's prologue overwrites the ring, the V image and the flag / validity words: every wave is done reading them
    if (pstep + 1 < n_steps_pers) __syncthreads();
    }   // blocks of this workgroup
}

}  // namespace nnop
