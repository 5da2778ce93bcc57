// fa_fwd_w64.hpp -- forward kernel, "64 query rows per wave, one wave per SIMD" form (16-bit types, E = 64 / 128).
//
// What `_flash_attention_fwd!` computes (src/attention.jl:1-131), third program form next to fa_fwd.hpp (32 rows per
// wave, 2 waves per SIMD) and fa_fwd_split.hpp (32 rows per wave, 4 waves per SIMD).  The structure is the one the CDNA4
// guide measures fastest for attention on this part: a workgroup = 4 waves = 256 query rows, each wave owns 64 rows (two
// 32-row blocks z = 0, 1) and the WHOLE register file of its SIMD (512 registers per lane):
//
//   * O^T accumulators (2 x E/32 tiles = 128 registers at E = 128) and the Q fragments (64) live in the ACCUMULATOR
//     file; the arch VGPRs hold two score tiles (this kv tile's and the next one's), K / V fragments and the softmax
//     temporaries.  hipcc cannot be made to split the files that way with MFMA builtins (it moves score tiles through
//     AGPRs: ~440 v_accvgpr moves per kv tile, measured 1.3x slower -- DESIGN.md section 5), so every MFMA here is
//     inline asm with explicit register classes ("a" = accumulator file): the compiler only allocates.
//   * every K / V fragment read from LDS feeds TWO MFMAs (z = 0, 1): half the LDS bytes per FLOP of the 32-row forms.
//   * K / V tiles arrive by LDS-DMA (buffer_load_dwordx4 ... lds: 1 KiB per wave-instruction, no staging registers, no
//     ds_write; see "LDS-DMA" below): the LDS images are the same swizzled RowImg / blocked ColImg as everywhere else, the
//     swizzle is applied to each lane's SOURCE offset (the DMA destination is lane-linear); rings of 3 slots, loads issued
//     two tiles ahead right after the tile's barrier, retired by a wait before the next barrier (raw s_barrier: the DMA
//     stays in flight across everything else).
//   * one barrier per kv tile.  The loop body is software-pipelined across tiles and HAND-PLACED: it is a sequence of
//     "slots" -- the LDS fragment read three fragments ahead, one MFMA, and a share of the other tile's softmax VALU work
//     dealt out by issue COST (W64Plan below: QK^T of tile t+1 beside exp / sum / convert of tile t; PV of tile t beside the
//     rest of the softmax and the row max of tile t+1) -- each pinned by sched_barrier(0), because hipcc does not interleave
//     inline-asm MFMAs with VALU work on its own (it models an asm statement as a 1-cycle instruction).
//   * wait states the compiler would pad around a builtin MFMA are explicit (hipcc pads nothing around asm):
//     MFMA result -> VALU read (fence_mfma_result), VALU result -> MFMA operand (fence_valu_operand).
//
// Modes: 0 plain (KL % 64 == 0, every logit live) / 1 masked (causal, key padding, ragged KL); the pair-bias mode stays
// on fa_fwd.hpp.  Same numerics contract as the other forms (fp32 softmax, deferred row max with threshold 2^8, O
// normalised once in the epilogue, residuals ms / ls per src/attention.jl:128-129).
#pragma once
#include <utility>
#include "fa_fwd.hpp"

// timing-only ablations of the hand-placed loop (make DEV=1 VAR=-DNNOP_W64_ABL=mask; results WRONG by construction), a bit mask:
//   1 no LDS-DMA in the loop   2 no tile barrier   4 no softmax arithmetic   8 no row max   16 no LDS fragment reads
#if !defined(NNOP_DEV_BUILD)
#undef NNOP_W64_ABL
#endif
#ifndef NNOP_W64_ABL
#define NNOP_W64_ABL 0
#endif
// diagnostic (make DEV=1 VAR=-DNNOP_W64_STAMP=1; results WRONG by construction): wave 0 of every workgroup overwrites the
// start of its first output row with s_memtime / s_memrealtime stamps (kernel entry, loop entry, loop exit, end) --
// tools/w64_stamp.py turns them into cycles per kv tile and the in-kernel clock (MI355X_MICROARCH.md, "in-kernel clock")
#if !defined(NNOP_DEV_BUILD)
#undef NNOP_W64_STAMP
#endif
#ifndef NNOP_W64_STAMP
#define NNOP_W64_STAMP 0
#endif
// E = 64 and E = 128: scale * log2(e) folded into Q (rounded to T once) and the exponent reference -m2 loaded as the INITIAL
// accumulator of QK^T, so that a logit leaves the matrix pipe ready for v_exp_f32 (no v_fma per logit).  See the header.
#ifndef NNOP_W64_PRESCALE
#define NNOP_W64_PRESCALE 1
#endif
// row sums of P on the matrix pipe (ones x P^T, one MFMA per 16-key step and query block, accumulated in the accumulator
// file) instead of one v_add per logit: in this form the WAVE'S ISSUE is the bound, the matrix pipe has slack
#ifndef NNOP_W64_MFMASUM
#define NNOP_W64_MFMASUM 1
#endif
#ifndef NNOP_W64_SUM_MAXE
#define NNOP_W64_SUM_MAXE 64
#endif
#ifndef NNOP_W64_RF8
#define NNOP_W64_RF8 0
#endif
// o stored with the nontemporal hint (streamed past L2: less dirty data to write back when the kernel ends)
#ifndef NNOP_W64_NT
#define NNOP_W64_NT 0
#endif
// softmax: how many exp issues lie between an element's v_exp_f32 and its first consumer (row-sum add / convert)
#ifndef NNOP_W64_LAG
#define NNOP_W64_LAG 3
#endif
#ifndef NNOP_W64_PF64
#define NNOP_W64_PF64 3
#endif
#ifndef NNOP_W64_PF128
#define NNOP_W64_PF128 3
#endif

namespace nnop {

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>).  The hand-placed loop body must not
// depend on the unroller (its size estimate of a body full of `if (i == ...)` blocks exceeds the pragma threshold and the
// loop then stays rolled, with every register array indexed at run time).
template <int... I, typename F> NNOP_DEV void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> NNOP_DEV void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// ---- inline-asm MFMAs with explicit register files ------------------------------------------------------------------
// volatile: an MFMA keeps its place among the other volatile statements of its slot (the pins of the softmax work) -- hipcc
// otherwise sinks it below the slot's VALU work, and the matrix pipe idles meanwhile.
template <typename T> struct MfmaAsm;
#define NNOP_MFMA_ASM(TYPE, FRAG, MNEMONIC)                                                                     \
    template <> struct MfmaAsm<TYPE> {                                                                          \
        /* D(vgpr) = A(vgpr) x B(acc file) */                                                                   \
        static NNOP_DEV f32x16 qk_first(FRAG a, FRAG bq) {                                                      \
            f32x16 d;                                                                                           \
            asm volatile(MNEMONIC " %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(bq));                                        \
            return d;                                                                                           \
        }                                                                                                       \
        /* D(vgpr) = A(vgpr) x B(vgpr) */                                                                       \
        static NNOP_DEV f32x16 qk_first_v(FRAG a, FRAG b) {                                                     \
            f32x16 d;                                                                                           \
            asm volatile(MNEMONIC " %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));                                         \
            return d;                                                                                           \
        }                                                                                                       \
        static NNOP_DEV void qk_acc_v(f32x16& d, FRAG a, FRAG b) {                                              \
            asm volatile(MNEMONIC " %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));                                         \
        }                                                                                                       \
        /* D(vgpr) = A(vgpr) x B(acc file) + C(vgpr), C kept */                                                 \
        static NNOP_DEV f32x16 qk_init(FRAG a, FRAG bq, const f32x16& c) {                                      \
            f32x16 d;                                                                                           \
            asm volatile(MNEMONIC " %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(bq), "v"(c));                               \
            return d;                                                                                           \
        }                                                                                                       \
        static NNOP_DEV void qk_acc(f32x16& d, FRAG a, FRAG bq) {                                               \
            asm volatile(MNEMONIC " %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(bq));                                        \
        }                                                                                                       \
        /* O(acc file) += A(vgpr) x B(vgpr) */                                                                  \
        static NNOP_DEV void pv_acc(f32x16& o, FRAG a, FRAG b) {                                                \
            asm volatile(MNEMONIC " %0, %1, %2, %0" : "+a"(o) : "v"(a), "v"(b));                                         \
        }                                                                                                       \
    };
NNOP_MFMA_ASM(__bf16, bf16x8, "v_mfma_f32_32x32x16_bf16")
NNOP_MFMA_ASM(_Float16, f16x8, "v_mfma_f32_32x32x16_f16")
#undef NNOP_MFMA_ASM

// MFMA result -> first non-MFMA reader: the wait states hipcc does not insert after an asm MFMA.  The data dependence
// through the operands keeps every reader below the statement.  Not the table's 12 states: the last MFMA may itself have been
// issued behind another one that still occupied the matrix pipe, and its final pass (accumulator registers 12..15) then
// lands up to two MFMA times = 64 cycles after its issue -- measured: with `s_nop 15; s_nop 3` (20 cycles) the epilogue read
// stale registers 13..15 of the last-written O tile in ~0.1 % of the rows, differently from run to run; 64 cycles were clean.
// RULE: a fence idles 128 cycles = 2 x the longest distance that reasoning allows (and > 6 x the distance measured unsafe).
// All sites (prologue score tile, rescale, epilogue, leaving a copy of the loop body) run once per workgroup or rarer.  A run
// of reads behind ONE fence needs the idle time once: the first statement is the fence, the others only carry the data
// dependence (acc_after_fence) -- asm volatile statements keep their order.
#define NNOP_FENCE_128 "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
NNOP_DEV void fence_mfma_result(f32x16& a, f32x16& b, f32x16& c, f32x16& d) {
    asm volatile(NNOP_FENCE_128 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
NNOP_DEV void fence_acc_result(f32x16& a) { asm volatile(NNOP_FENCE_128 : "+a"(a)); }
NNOP_DEV void acc_after_fence(f32x16& a) { asm volatile("" : "+a"(a)); }
// VALU-written registers -> MFMA A/B operand inside an asm statement: 2 wait states
template <typename F> NNOP_DEV void fence_valu_operand(F& a, F& b) { asm volatile("s_nop 1" : "+v"(a), "+v"(b)); }

// Q fragment: 16 bytes per lane straight from HBM into the accumulator file (an MFMA B operand may be an AGPR).  The
// load is invisible to hipcc's wait-count bookkeeping: the destination is valid only after the `s_waitcnt vmcnt(0)` of
// q_landed(), which takes the fragments as operands so that no consumer can be scheduled above it.
template <typename F> NNOP_DEV F load_q_frag(const void* gptr) {
    F d;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(d) : "v"(gptr) : "memory");
    return d;
}

// ---- LDS-DMA ---------------------------------------------------------------------------------------------------------
// K / V tiles go HBM -> LDS with buffer_load_dwordx4 ... lds: lane l of a wave copies 16 bytes from
//   base(V#) + soffset + voffset(lane) + imm      to LDS byte      M0 + imm + 16 l .
// A wave owns NJ CONSECUTIVE 1-KiB pieces of a tile image, so one M0 (the image's ring slot + the wave's share) and one scalar
// offset (the tile) serve all of them, the piece being selected by the immediate (which both addresses add): per tile and
// tensor 2 scalar instructions + NJ loads instead of 3 scalar instructions per piece -- on a wave whose ISSUE is the bound, a
// scalar instruction costs as much as a vector one (tools/ubench/gapcost.hip).  The image's swizzle is applied to the lane's
// SOURCE offset (the destination is lane-linear).  The descriptor's NUM_RECORDS is the byte size of the (batch, kv-head)
// tensor; on gfx950 the range check covers soffset + voffset + imm (measured: tools/ubench/probe_buf.hip,
// profiles/r02/probe_buf.log), so the rows of the last tile past KL are out of range and read as zeros -- finite data
// behind the mask -- with no per-lane clamping.
// M0 is written in the statement of a tensor's first piece and stays for its other pieces; nothing else in these kernels uses
// M0 (gfx950 DS instructions need no M0 setup; tools/audit_w64.py fails on any other M0 access).
NNOP_DEV u32x4 make_rsrc(const void* base, uint32_t bytes) {
    const uint64_t a = (uint64_t)(uintptr_t)base;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);       // stride 0: raw buffer
    r[2] = bytes;
    r[3] = 0x00020000u;                                                         // DATA_FORMAT = 32 (untyped dword access)
    return r;
}
template <int J, bool FIRST> NNOP_DEV void dma_piece(u32x4 rsrc, uint32_t voff, uint32_t soff, uint32_t lds_dst) {
    if constexpr (FIRST)
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen offset:%4 lds"
                     :: "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst), "n"(J * 1024) : "memory");
    else
        asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds" :: "v"(voff), "s"(rsrc), "s"(soff), "n"(J * 1024) : "memory");
}

// ---- the slot plan ------------------------------------------------------------------------------------------------
// One iteration of the loop is NX + NY2 "slots" (one MFMA each).  Besides its MFMA a slot carries fixed work (the LDS
// fragment read-ahead, an LDS-DMA piece, ...) and a share of two streams of movable work: the softmax steps of tile t
// (0 .. 63 + LAG: step n issues the exp of element n and finishes element n - LAG -- row-sum add, and the 16-bit convert of
// the pair it closes) and the row-max items of tile t + 1.  A wave issues in order: a slot lasts max(32, 8 + issue cost of its
// fillers) cycles (v_exp_f32 8, other VALU / DS / scalar 4, an LDS-DMA piece ~24 with its M0 setup: MI355X_MICROARCH.md,
// tools/w64_gaps.py), so the streams are dealt out by COST: the smallest per-slot budget `cap` for which a greedy in-order
// fill places everything inside its window -- a softmax step no later than the slot before the first MFMA that reads the P^T
// word it completes; a row-max item no earlier than two slots behind the last MFMA that writes the logits it reads.
// NJ2: LDS-DMA pieces per wave and iteration (both tensors); EV: columns of V / O the workgroup handles (E, or E / 2 at E = 256)
template <int E, bool SUM, bool MASKED, int NJ2, int LAG, int EV = E> struct W64Plan {
    static constexpr int KS = E / 16, EB = EV / 32, NKF = 2 * KS;
    static constexpr int NX = 2 * NKF, G = 2 * EB + (SUM ? 2 : 0), NY2 = 4 * G, NSLOT = NX + NY2, NYB = NY2 / 2;
    static constexpr int NSTEP = 64 + LAG, NMX = 34;
    static constexpr int DSTRIDE = NYB + 2 * NJ2 <= NY2 ? 2 : 1;       // a DMA piece every second slot behind the barrier where that fits
    static_assert(NYB + DSTRIDE * (NJ2 - 1) + 2 <= NY2, "the DMA batch fits behind the barrier");
    static constexpr bool dma_at(int i) { return i > NYB && (i - NYB - 1) % DSTRIDE == 0 && (i - NYB - 1) / DSTRIDE < NJ2; }   // i: phase-Y slot
    static constexpr int dma_index(int i) { return (i - NYB - 1) / DSTRIDE; }
    static constexpr int MASK_SLOT = NX + 2;            // masked mode: tile t+1 is masked here, its row max starts behind it
    static constexpr int ADDR_SLOT = NX + 1;            // scalar address arithmetic of the iteration's DMA batch
    int sm_end[NSLOT] = {};                             // softmax steps [sm_end[s-1], sm_end[s]) run in slot s
    int mx_end[NSLOT] = {};
    int cost[NSLOT] = {};                               // modelled filler cost per slot (reporting)
    int cap = 0;

    // fixed work in front of the MFMA of slot s (s = NSLOT: slot 0 of the next iteration, behind the loop's back edge)
    static constexpr int pre_cost(int s) {
        if (s >= NSLOT) return 8 + 12 + 28;             // read-ahead + image bases + ring rotation, loop control, rescale branch
        if (s < NX) return (s & 1) == 0 ? 8 : 0;        // K fragment read-ahead: v_xor + ds_read_b128
        const int i = s - NX, w = i % G, wp = w - (SUM ? 2 : 0);
        const bool is_sum = SUM && w < 2;
        int c = 0;
        if (!is_sum && (wp & 1) == 0) c += 8;           // V fragment (two transposed reads) or a K(t+2) fragment
        if (dma_at(i)) c += 24;                         // LDS-DMA piece
        return c;
    }
    // fixed work between the MFMA of slot s and the next one (the movable streams share this gap)
    static constexpr int fixed_cost(int s) {
        int c = pre_cost(s + 1);
        if (s == ADDR_SLOT) c += 40;
        if (MASKED && s == MASK_SLOT) c += 16;
        if (s == NSLOT - 1) c += 24;                    // rescale test of tile t+1
        return c;
    }
    static constexpr int step_cost(int n) {
        int c = n < 64 ? 8 : 0;
        const int m = n - LAG;
        if (m >= 0) c += (SUM ? 0 : 4) + ((m & 1) ? 4 : 0);
        return c;
    }
    // last slot that may hold step n: the one before the first MFMA reading the P^T word that element n - LAG belongs to
    static constexpr int step_deadline(int n) {
        const int m = n < LAG ? 0 : n - LAG, c = m >> 3, kk = c >> 1, z = c & 1;
        return NX + kk * G + z - 1;
    }
    static constexpr int mx_cost(int u) { return u < 32 ? 4 : 28; }
    static constexpr int mx_earliest(int u) {
        // (masked mode: the same -- a tile that turns out to need its mask redoes the items that ran before the mask slot)
        const int q = u >> 1;
        return q < 8 ? NX / 2 + 2 : NX + 2;             // logits of key block 0 are complete half way through phase X
    }
    static constexpr int kMxDeadline = NSLOT - 2;       // the rescale test of the last slot reads the finished row max

    constexpr bool fill(int budget) {
        // row-max items as LATE as their deadline allows (the tail of phase Y has nothing else to do: every P^T word is due
        // before the last 16-key step), softmax steps as EARLY as the budget allows
        int u = NMX;
        for (int sl = NSLOT - 1; sl >= 0; --sl) {
            int used = fixed_cost(sl);
            mx_end[sl] = u;
            if (sl <= kMxDeadline)
                while (u > 0 && mx_earliest(u - 1) <= sl && used + mx_cost(u - 1) <= budget) used += mx_cost(--u);
            cost[sl] = used;
        }
        if (u > 0) return false;
        int n = 0, over = 0;                            // over: worst overfill caused by a deadline-forced placement
        for (int sl = 0; sl < NSLOT; ++sl) {
            int used = cost[sl];
            while (n < NSTEP && (used + step_cost(n) <= budget || step_deadline(n) <= sl)) {
                used += step_cost(n++);
                if (used > budget && used - budget > over) over = used - budget;
            }
            sm_end[sl] = n;
            cost[sl] = used;
        }
        cap = budget;
        return n == NSTEP && over <= 4;
    }
    static constexpr W64Plan make() {
        W64Plan pl{};
        for (int b = 12; b <= 96; b += 2)
            if (pl.fill(b)) break;
        return pl;
    }
};

template <typename T, int E, int EV = E> constexpr int fa_fwd_w64_lds_bytes(bool masked) {
    return 3 * (RowImg<T, E>::bytes(64) + ColImg<T, EV>::bytes(64)) + (masked ? 16 + 8 * kMaxMaskTiles : 0);
}

// EV: the columns of V / O this workgroup handles.  EV = E except at E = 256, where the O^T accumulators of 64 rows x 256 columns would
// be the whole accumulator file: the grid then holds every block twice, each copy contracting Q K^T over all of E (Q fragments: 128
// registers) but accumulating one 128-column half of O (+33 % MFMA work for a spill-free kernel; the 32-row form spills 62-152).
template <typename T, int E, int MODE, bool PRE, int EV = E>
__global__ __launch_bounds__(256, 1) void fa_fwd_w64_kernel(const FwdParams p_arg) {
    static_assert(sizeof(T) == 2 && ((EV == E && (E == 64 || E == 128)) || (E == 256 && EV == 128)), "16-bit element types, E = 64, 128 or 256 (two 128-column halves)");
    using frag_t = typename Elem<T>::frag;
    using KImg   = RowImg<T, E>;
    using VImg   = ColImg<T, EV>;
    using MM     = MfmaAsm<T>;
    constexpr bool kGeneral = MODE != 0;
    constexpr int BK = 64, KB = 2, KS = E / 16, EB = EV / 32, NS = 3;
    constexpr int KBYTES = KImg::bytes(BK), VBYTES = VImg::bytes(BK);
    constexpr int TILE_BYTES = BK * E * (int)sizeof(T);       // one kv tile in HBM
    constexpr int NJK = KBYTES / 4096, NJV = VBYTES / 4096;   // DMA pieces per wave and tile: K image, V image
    constexpr int NJ2 = NJK + NJV;
    static_assert(KBYTES % 4096 == 0 && VBYTES % 4096 == 0 && (EV != E || NJK == NJV), "");
    constexpr int NKF = KB * KS;                              // K fragments per tile (each feeds z = 0, 1)
    constexpr int NVF = 2 * KB * EB;                          // V fragments per tile
    constexpr int NF = NKF + NVF;                             // fragment stream of one iteration
    constexpr int NX = 2 * NKF, NY = 2 * NVF;                 // MFMA slots of phase X / phase Y
    constexpr int PF = E >= 128 ? NNOP_W64_PF128 : NNOP_W64_PF64, RF = (PF < 4 && !NNOP_W64_RF8) ? 4 : 8;      // fragments read ahead / fragment ring
    static_assert(NF % RF == 0, "the fragment ring index must be static across iterations");
    constexpr float kThr = 8.0f;
    constexpr uint64_t kFull = ~0ull;
    constexpr bool kPre = PRE && NNOP_W64_PRESCALE != 0;     // logits leave the MFMA as (s * scale * log2e - reference)
    constexpr bool kSum = NNOP_W64_MFMASUM != 0 && E <= NNOP_W64_SUM_MAXE;  // row sums on the matrix pipe (E = 128: 1.7 % slower before the planner, 3.5 % slower with it: 2863 vs 2770 cycles per tile)

    extern __shared__ __attribute__((aligned(16))) char smem[];
#if NNOP_W64_STAMP
    uint64_t stamp[8], stamp_p[2] = {0, 0};
    stamp[0] = __builtin_amdgcn_s_memtime();
    stamp[1] = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    // ---- persistent form (p.persist = blocks per workgroup; 0: one block per workgroup, the grid holds them all) ----------------------
    // A 512-register, 96-KiB-of-LDS workgroup owns its CU, and the hand-over to the next one costs ~8-10 us of an idle CU per block
    // (in-kernel stamps, tools/w64_stamp.py: the CUs hold a workgroup 90.5 % of the launch at C3, 93.7 % at the C5 shard).  With
    // 256 workgroups that each walk a static list of blocks the hand-over is one barrier.  The list: XCD x (= blockIdx % 8, the
    // observed round-robin dispatch -- a speed assumption only) owns the (batch, q-head) columns [x BH/8, (x+1) BH/8), so its 32
    // workgroups stream the same K / V through one L2 as before; the XCD's blocks, columns in order and q-blocks DESCENDING inside a
    // column, are dealt out 32 at a time, alternately forwards and backwards over the XCD's workgroups -- under a causal mask a
    // workgroup's q-blocks (n-1-c, c, ...) then sum to the same work every two steps, and all workgroups end together.
    constexpr bool kPersist = kGeneral && EV == E;          // masked-mode kernels of E = 64 / 128 only: elsewhere the loop folds away
    const int n_steps_pers = (kPersist && p_arg.persist > 0) ? p_arg.persist : 1;
    for (int pstep = 0; pstep < n_steps_pers; ++pstep) {
    // The parameters are re-read from the kernel-argument segment for every block (through a pointer the compiler cannot see
    // through): kept live across the block loop they cost ~25 scalar registers that the hand-placed loop needs (hipcc spilled 3-15).
    typedef const FwdParams __attribute__((address_space(4))) * params_cp;
    params_cp pp = (params_cp)__builtin_amdgcn_kernarg_segment_ptr();
    if constexpr (kPersist) asm volatile("" : "+s"(pp));
    FwdParams p_blk;
    if constexpr (!kPersist) p_blk = p_arg;
    if constexpr (kPersist) {
        p_blk.pair = nullptr;
        p_blk.o = pp->o; p_blk.ms = pp->ms; p_blk.ls = pp->ls; p_blk.q = pp->q; p_blk.k = pp->k; p_blk.v = pp->v; p_blk.kpad = pp->kpad;
        p_blk.QL = pp->QL; p_blk.KL = pp->KL; p_blk.QH = pp->QH; p_blk.KH = pp->KH; p_blk.B = pp->B; p_blk.causal = pp->causal;
        p_blk.n_qblk = pp->n_qblk; p_blk.n_wg = pp->n_wg; p_blk.scale = pp->scale; p_blk.persist = pp->persist; p_blk.persist_hx = pp->persist_hx; p_blk.persist_asc = pp->persist_asc;
    }
    const FwdParams& p = p_blk;
    const int vsplit = EV == E ? 0 : (int)blockIdx.x / p.n_wg;     // which column half (E = 256)
    int qblk, bh;
    if (kPersist && p.persist > 0) {
        const int x = (int)blockIdx.x & 7, c = (int)blockIdx.x >> 3;
        const int pos = 32 * pstep + ((pstep & 1) ? 31 - c : c);
        const int col = pos / p.n_qblk;
        qblk = p.persist_asc ? pos - col * p.n_qblk : p.n_qblk - 1 - (pos - col * p.n_qblk);
        // the XCD's columns: an eighth of the HEADS of every batch (batch-major) when the heads divide -- with per-batch key lengths
        // every XCD, and every step of 32 blocks, then sees every batch alike -- else a contiguous eighth of the (batch, head) pairs
        if (p.persist_hx > 0) bh = (col / p.persist_hx) * p.QH + x * p.persist_hx + col % p.persist_hx;
        else bh = x * ((p.B * p.QH) >> 3) + col;
    } else {
        const int lin = xcd_remap_chunked(EV == E ? (int)blockIdx.x : (int)blockIdx.x % p.n_wg, p.n_wg, p.n_qblk * (p.QH / p.KH));
        qblk = lin % p.n_qblk;
        bh = lin / p.n_qblk;
        if (kGeneral && p.causal) qblk = p.n_qblk - 1 - qblk;    // heaviest q-blocks first
    }
    const int b = bh / p.QH, qh = bh - b * p.QH;
    const int kvh = qh / (p.QH / p.KH);                      // cld(q_head, n_q_per_kv), 0-based (src/attention.jl:28)
    const int q0w = qblk * 256 + wave * 64;                  // first query row of this wave
    int qi[2];
    qi[0] = q0w + r;
    qi[1] = q0w + 32 + r;

    const T* __restrict__ qp = (const T*)p.q + ((size_t)bh * p.QL) * E;
    const char* __restrict__ kp = (const char*)((const T*)p.k + ((size_t)(b * p.KH + kvh) * p.KL) * E);
    const char* __restrict__ vp = (const char*)((const T*)p.v + ((size_t)(b * p.KH + kvh) * p.KL) * E);
    const uint8_t* __restrict__ mp = kGeneral && p.kpad ? p.kpad + (size_t)b * p.KL : nullptr;

    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;         // LDS byte address of the dynamic segment
    const uint32_t kring = lds0, vring = lds0 + NS * KBYTES;
    uint64_t* const vbits = reinterpret_cast<uint64_t*>(smem + NS * (KBYTES + VBYTES) + 16);

    // ---- number of kv tiles (workgroup) / live tiles (this wave) ------------------------------------------------
    // Masked mode keeps ONE description of "which keys exist and are valid": a 64-bit word per kv tile in LDS (key
    // padding: built from the mask row by kpad_scan; otherwise ones up to KL).  The loop body then has no branch on a
    // launch-constant (causal? mask given?) -- hipcc would unswitch the whole hand-placed loop on each of them.  The
    // launcher sends sequences beyond kMaxMaskTiles tiles to the 32-row kernel.
    int n_tiles = (p.KL + BK - 1) / BK;
    int causal_q0 = 0x3fffffff;                              // first query row of the wave if causal, else "never clipped"
    int qlim[2] = {0x3fffffff, 0x3fffffff};                  // per lane: last visible key (causal: the query index)
    if constexpr (kGeneral) {
        if (p.causal) {
            int q_last = qblk * 256 + 255;
            if (q_last > p.QL - 1) q_last = p.QL - 1;
            const int t_c = q_last / BK + 1;
            if (t_c < n_tiles) n_tiles = t_c;
            causal_q0 = q0w;
            qlim[0] = qi[0];
            qlim[1] = qi[1];
        }
        if (mp) {
            int* slot = reinterpret_cast<int*>(smem + NS * (KBYTES + VBYTES));
            const int nk = n_tiles * BK < p.KL ? n_tiles * BK : p.KL;
            const int last = kpad_scan(mp, p.KL, nk, vbits, kMaxMaskTiles, slot, tid, 256);
            const int t_m = last / BK + 1;
            if (t_m < n_tiles) n_tiles = t_m;
        } else {
            for (int w = tid; w < n_tiles; w += 256) {
                const int left = p.KL - w * BK;
                vbits[w] = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
            }
            __syncthreads();
        }
    }
    int n_live = n_tiles;
    if (kGeneral && p.causal) {
        const int t_w = (q0w + 63) / BK + 1;
        if (t_w < n_live) n_live = t_w;
    }

    // ---- per-lane DMA source offsets inside a tile (the image's layout, applied to the SOURCE) -------------------
    // wave w copies the image bytes [NJ KiB * w, NJ KiB * (w + 1)), piece j = its j-th KiB; lane l the 16 bytes at 16 l.
    // k_voff[j] / v_voff: source byte of that chunk inside the tile MINUS 1024 j (the load's immediate adds it back).
    // More than 4 pieces per wave (E = 256: the K image is 32 KiB) go in groups of 4: the immediate has 12 bits, the next group moves M0.
    static_assert(NJK * 1024 * 4 == KBYTES && NJV * 1024 * 4 == VBYTES && NJV <= 4 && NJK <= 8, "four waves x NJ pieces = one image");
    constexpr int NVO = EV == E ? 1 : NJV;                    // EV == E: +1 KiB in the V image = +1 KiB in the source: one offset serves all pieces
    uint32_t k_voff[NJK], v_voff[NVO];
#pragma unroll
    for (int j = 0; j < NJK; ++j) {
        const int off = (wave * NJK + j) * 1024 + lane * 16;  // LDS byte inside the image
        const int row = off / KImg::kRowBytes, phys = (off % KImg::kRowBytes) >> 4;
        k_voff[j] = (uint32_t)(row * KImg::kRowBytes + ((phys ^ KImg::xor_of(row)) << 4) - (j & 3) * 1024);
    }
#pragma unroll
    for (int j = 0; j < NVO; ++j) {
        // blocked V image [row >> 2][column block][row & 3][32 columns]; source rows are E elements long, this workgroup's columns
        // start at vsplit * EV
        const int off = (wave * NJV + j) * 1024 + lane * 16;
        const int blk = off >> 8, rg = blk / VImg::kEB, eb = blk % VImg::kEB, rr = (off >> 6) & 3, c4 = (off >> 4) & 3;
        v_voff[j] = (uint32_t)((4 * rg + rr) * KImg::kRowBytes + ((4 * eb + c4) << 4) + vsplit * EV * (int)sizeof(T) - j * 1024);
    }
    static_assert((1024 / 256) % VImg::kEB == 0 && (4 * (1024 / 256 / VImg::kEB)) * VImg::kRowBytes == 1024, "V image: 1 KiB = whole row groups");
    const uint32_t wave_off_k = (uint32_t)(wave * NJK * 1024), wave_off_v = (uint32_t)(wave * NJV * 1024);
    const uint32_t kv_bytes = (uint32_t)p.KL * (uint32_t)KImg::kRowBytes;     // one (batch, kv-head) tensor; < 4 GiB (launcher)
    const u32x4 krs = make_rsrc(kp, kv_bytes), vrs = make_rsrc(vp, kv_bytes);
    // Past the last tile the LAST tile is copied again (into a ring slot nobody reads any more) instead of branching
    // around the issue: a branch inside the loop body splits its basic block, and hipcc then sinks the softmax
    // arithmetic of the earlier slots below the branch, next to its first use (see pin() below).
    auto tile_off = [&](int t) -> uint32_t { return (uint32_t)(t < n_tiles ? t : n_tiles - 1) * (uint32_t)TILE_BYTES; };
    // piece j of the tile at byte `soff` of the tensor behind `rs`, into the ring slot whose wave share starts at LDS byte `dst`
    auto issue_piece = [&](u32x4 rs, uint32_t soff, uint32_t dst, uint32_t voff, auto jc) {
        constexpr int j = decltype(jc)::value;                // piece index inside the tensor's share; groups of 4 per M0
        dma_piece<(j & 3), (j & 3) == 0>(rs, voff, soff, dst + (uint32_t)((j >> 2) * 4096));        // (the lane's source offset holds the rest)
    };
    auto issue_k = [&](int t, uint32_t slot) {
        const uint32_t so = tile_off(t);
        static_for<NJK>([&](auto jc) { issue_piece(krs, so, slot, k_voff[decltype(jc)::value], jc); });
    };
    auto issue_v = [&](int t, uint32_t slot) {
        const uint32_t so = tile_off(t);
        static_for<NJV>([&](auto jc) { issue_piece(vrs, so, slot, v_voff[NVO == 1 ? 0 : decltype(jc)::value], jc); });
    };
    // Ring slots (LDS byte addresses) as rotating scalars -- no t % 3 arithmetic in the loop:
    //   kA, kB, kC = slots of K(t+1), K(t+2), K(t+3) (= K(t)'s, free);   vA, vB, vC = slots of V(t), V(t+1), V(t+2) (free)
    // Each holds slot + this wave's DMA share (wave_off): the DMA uses it as is, the fragment reads add a lane base that has
    // wave_off subtracted.
    uint32_t kA = kring + wave_off_k + 1 * KBYTES, kB = kring + wave_off_k + 2 * KBYTES, kC = kring + wave_off_k;
    uint32_t vA = vring + wave_off_v, vB = vring + wave_off_v + 1 * VBYTES, vC = vring + wave_off_v + 2 * VBYTES;
    // byte offsets (inside the tensor) of the tiles the next DMA batch copies: K(t+3), V(t+2), clamped to the last tile
    const uint32_t last_off = (uint32_t)(n_tiles - 1) * (uint32_t)TILE_BYTES;
    uint32_t off_k3 = tile_off(3), off_v2 = tile_off(2);
    auto advance_offsets = [&]() {
        off_v2 = off_k3;
        const uint32_t nx = off_k3 + (uint32_t)TILE_BYTES;
        off_k3 = nx < last_off ? nx : last_off;
    };
    auto rotate_slots = [&]() {
        const uint32_t k0 = kA, v0 = vA;
        kA = kB; kB = kC; kC = k0;
        vA = vB; vB = vC; vC = v0;
    };

    // ragged KL: the rows of the last tile past KL are outside the descriptor's range.  Whatever the DMA does with such a lane
    // (zeros, or nothing), the ring must not hold non-finite garbage there -- V rows of masked keys are multiplied by P = 0.
    if (kGeneral && (p.KL & (BK - 1)) != 0) {
        for (int i = tid * 16; i < NS * (KBYTES + VBYTES); i += 256 * 16) *reinterpret_cast<u32x4*>(smem + i) = u32x4{0, 0, 0, 0};
        __syncthreads();
    }
    // ---- prologue: K(0..2), V(0..1) in flight; Q fragments straight to registers ----------------------------------
    issue_k(0, kC);                                          // K(0) first: S(0) needs only K(0) and Q
    const float c2 = p.scale * kLog2e;
    frag_t qf[2][KS];                                        // accumulator file, for the whole kernel
#pragma unroll
    for (int z = 0; z < 2; ++z) {
        const int qc = qi[z] < p.QL ? qi[z] : p.QL - 1;
        const T* qrow = qp + (size_t)qc * E;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if constexpr (kPre) {
                const frag_t raw = *reinterpret_cast<const frag_t*>(qrow + 16 * ks + 8 * h);
                f32x8 w = __builtin_convertvector(raw, f32x8);
#pragma unroll
                for (int j = 0; j < 8; ++j) w[j] *= c2;
                qf[z][ks] = __builtin_convertvector(w, frag_t);
            } else {
                qf[z][ks] = load_q_frag<frag_t>(qrow + 16 * ks + 8 * h);
            }
        }
    }
    issue_v(0, vA); issue_k(1, kA); issue_v(1, vB); issue_k(2, kB);      // 2 NJ2 pieces that may still be in flight below
    // kPre: -(exponent reference) per query row, broadcast over a 16-register tuple = the initial accumulator of QK^T
    f32x16 negm[2];
    if constexpr (kPre) {
#pragma unroll
        for (int z = 0; z < 2; ++z)
#pragma unroll
            for (int i = 0; i < 16; ++i) negm[z][i] = 0.f;
        fence_valu_operand(negm[0], negm[1]);               // VALU-written -> MFMA operand: 2 wait states, and opaque
    }

    f32x16 oacc[2][EB];
#pragma unroll
    for (int z = 0; z < 2; ++z)
#pragma unroll
        for (int eb = 0; eb < EB; ++eb)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[z][eb][i] = 0.f;
    float m2[2] = {-INFINITY, -INFINITY}, mt[2] = {-INFINITY, -INFINITY};
    float mbase[2] = {0.f, 0.f};                             // the finite part of m2 (0 while no key has been seen): kept, not re-derived per tile
    float lp[2][2] = {{0.f, 0.f}, {0.f, 0.f}};               // row sums (VALU form): two chains per query block
    f32x16 lacc[2];                                          // row sums (matrix-pipe form): every register = sum_k P[k][query]
    frag_t ones;
    if constexpr (kSum) {
#pragma unroll
        for (int z = 0; z < 2; ++z)
#pragma unroll
            for (int i = 0; i < 16; ++i) lacc[z][i] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) ones[j] = from_f32<T>(1.0f);
        // opaque: as a known constant hipcc re-materialises it (v_mov) right in front of the asm MFMA that reads it -- a
        // VALU write -> MFMA operand read without the 2 wait states (measured: garbage row sums of query block 0 only,
        // the first of the two MFMAs behind the v_movs).  tools/audit_w64.py checks for this pattern.
        asm volatile("" : "+v"(ones));
    }

    // ---- LDS fragment reads from integer addresses -------------------------------------------------------------------
    // K row read (RowImg): row 32 kb + r, 16-byte chunk (2 ks + h) ^ xor_of(row).  xor_of(32 kb + r) = xor_of(r), and with
    // x = xor_of(r): (2 ks + h) ^ x = (x ^ h) ^ (2 ks), so   addr(kb, ks) = (A ^ (ks << 5)) + kb * 32 * row bytes   with
    // A = image + r * row bytes + ((x ^ h) << 4): ONE lane-dependent base per iteration, one v_xor per fragment.
    // V transposed read (ColImg): image + lane_base + compile-time offsets.
    typedef __attribute__((address_space(3))) const frag_t* lds_frag_p;
    typedef __attribute__((address_space(3))) s16x4* lds_tr_p;
    const uint32_t k_lane = (uint32_t)(r * KImg::kRowBytes + ((KImg::xor_of(r) ^ h) << 4)) - wave_off_k;   // ring scalars include wave_off
    const uint32_t v_lane = (uint32_t)VImg::lane_base(lane) - wave_off_v;
    auto read_kfrag = [&](uint32_t ka, int f) -> frag_t {          // ka = image address + k_lane
        const int kb = f / KS, ks = f % KS;
        return *(lds_frag_p)(uintptr_t)((ka ^ (uint32_t)(ks << 5)) + (uint32_t)(kb * 32 * KImg::kRowBytes));
    };
    auto read_vfrag = [&](uint32_t va, int g) -> frag_t {          // va = image address + v_lane;  g = kk * EB + eb
        const int kk = g / EB, eb = g % EB;
        const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(uintptr_t)(va + (uint32_t)(((4 * kk) * VImg::kEB + eb) << 8)));
        const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(uintptr_t)(va + (uint32_t)(((4 * kk + 2) * VImg::kEB + eb) << 8)));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 v8 = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
        return __builtin_bit_cast(frag_t, v8);
    };
    // per-iteration image bases as opaque registers: everything derived from them is base + immediate (hipcc otherwise
    // hoists one lane-constant address per fragment out of the loop and runs out of registers)
    auto opaque = [](uint32_t x) { asm volatile("" : "+v"(x)); return x; };
    // "computed HERE": hipcc sinks pure arithmetic to the basic block of its first use, i.e. out of the slot it was
    // placed in and below any branch in between; a value that passes through a volatile statement stays put.
    auto pin = [](auto& x) { asm volatile("" : "+v"(x)); };
    // which keys of tile t exist and are valid, wave-uniform (masked mode only)
    auto tile_valid = [&](int t) -> uint64_t { return kpad_tile_bits<BK>(vbits, t); };
    // The loop reads the word of tile t+2 one iteration before it needs it (as a vector register pair, made uniform only
    // when used): a read that is consumed right away would wait for every fragment read issued before it (LDS returns in
    // order) -- measured on all-valid masks: masked mode 5.5 % (E = 128) / 9 % (E = 64) slower than plain mode.
    uint64_t vword_next = 0;
    auto vword_fetch = [&](int t) { vword_next = vbits[t < kMaxMaskTiles ? t : kMaxMaskTiles - 1]; };
    auto vword_take = [&]() -> uint64_t {
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(vword_next >> 32)) << 32) |
               (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)vword_next);
    };
    auto tile_needs_mask = [&](int t, uint64_t valid) { return valid != kFull || t * BK + BK - 1 > causal_q0; };
    // causal / padding mask of tile t applied to its raw logits (-> -inf), both query blocks.  Per (z, kb) ONE 32-bit
    // lane mask: validity bits of the lane's key rows AND the causal prefix (local key row <= lim).
    auto apply_mask = [&](f32x16 (&s)[2][KB], int t, uint64_t valid) {
        const int k0 = t * BK;
#pragma unroll
        for (int z = 0; z < 2; ++z)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int lim = qlim[z] - k0 - 32 * kb - 4 * h;
                const uint32_t cm = lim >= 31 ? ~0u : (lim < 0 ? 0u : ((2u << lim) - 1u));
                const uint32_t m = (uint32_t)(valid >> (32 * kb + 4 * h)) & cm;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int lr = (i & 3) + 8 * (i >> 2);
                    s[z][kb][i] = ((m >> lr) & 1u) ? s[z][kb][i] : -INFINITY;
                }
            }
    };
    // row max (log2 units, both lane halves) of one query block's raw score tile
    auto row_max = [&](const f32x16 (&s)[KB]) -> float {
        float mxp[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int i = 0; i < 16; i += 2)
                mxp[(i >> 1) & 3] = fmaxf(fmaxf(mxp[(i >> 1) & 3], s[kb][i]), s[kb][i + 1]);
        return half_swap_max(fmaxf(fmaxf(mxp[0], mxp[1]), fmaxf(mxp[2], mxp[3])) * (kPre ? 1.0f : c2));
    };
    // Before a tile is exponentiated: has some row's max outgrown the reference by > kThr (or does the row see its first
    // key)?  `mx`: row max of the tile in log2 units -- absolute, or (kPre) relative to the reference that was baked into the
    // tile's logits when its QK^T ran.  The test runs in the LAST slot of the iteration before (the prologue for tile 0), so
    // that an iteration opens with nothing but the branch; m2 does not change in between.
    // `live`: the tile exists for this wave (the last iteration computes the logits of one tile too many: in plain mode a copy
    // of the last tile, harmless; in masked mode unmasked garbage that must not reach the row max).
    auto rescale_test = [&](const float (&mx)[2], bool live) -> int {
        bool any = false;
        const float lim = (kGeneral && !live) ? -INFINITY : INFINITY;          // scalar select
#pragma unroll
        for (int z = 0; z < 2; ++z) {
            float mabs = kPre ? mx[z] + mbase[z] : mx[z];                    // mbase: what the tile's logits have subtracted
            if constexpr (kGeneral) mabs = fminf(mabs, lim);
            mt[z] = fmaxf(mt[z], mabs);
            any = any || (mabs > m2[z] + kThr);
        }
        return __builtin_amdgcn_ballot_w64(any) != 0 ? 1 : 0;    // wave-uniform, a scalar register across the loop's back edge
    };
    // Rare path: raise the reference; everything accumulated at the old one (O, l) is scaled exactly once, and (kPre) the
    // tile `sc` is re-based onto the new reference.
    auto rescale = [&](const float (&mx)[2], f32x16 (&sc)[2][KB], bool first) {
        // The fence of this (rare) block carries NO operand: with the accumulator file full (E = 256) hipcc satisfies a tied "+a"
        // operand by copying the tile through arch VGPRs -- and put the first v_accvgpr_read in FRONT of the statement that was
        // meant to fence it (tools/audit_w64.py).  An operand-free volatile statement with a memory clobber stays first.
        if (!first) {
            asm volatile(NNOP_FENCE_128 ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int z = 0; z < 2; ++z) {
            const float base0 = kPre ? mbase[z] : 0.f;
            const float mabs = kPre ? mx[z] + base0 : mx[z];
            const bool up = mabs > m2[z] + kThr;
            const float mn = up ? mabs : m2[z];
            const float alpha = up ? fast_exp2(m2[z] - mn) : 1.f;         // m2 = -inf -> 0 (nothing accumulated yet)
            if (!first) {                                                 // first tile: O and l are still zero
#pragma unroll
                for (int eb = 0; eb < EB; ++eb) {
                    acc_after_fence(oacc[z][eb]);
#pragma unroll
                    for (int i = 0; i < 16; ++i) oacc[z][eb][i] *= alpha;
                    // back in the accumulator file BEFORE the paths merge: otherwise the merged value is allocated in
                    // arch VGPRs and the common path pays 128 v_accvgpr_read + 128 v_accvgpr_write per tile for it
                    asm volatile("" : "+a"(oacc[z][eb]));
                }
                if constexpr (kSum) {
                    acc_after_fence(lacc[z]);
#pragma unroll
                    for (int i = 0; i < 16; ++i) lacc[z][i] *= alpha;
                    asm volatile("" : "+a"(lacc[z]));
                } else {
                    lp[z][0] *= alpha;
                    lp[z][1] *= alpha;
                }
            }
            if constexpr (kPre) {
                const float nbase = mn != -INFINITY ? mn : 0.f;
                const float shift = base0 - nbase;                         // logits already hold -base0
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) sc[z][kb][i] += shift;
#pragma unroll
                for (int i = 0; i < 16; ++i) negm[z][i] = -nbase;
            }
            m2[z] = mn;
            mbase[z] = mn != -INFINITY ? mn : 0.f;
        }
        if constexpr (kPre) fence_valu_operand(negm[0], negm[1]);
    };

    // ---- prologue, continued: wait for the first tiles, S(0) = K(0) Q^T, its mask and row max -----------------------
    // K(0) and Q landed (every wave's pieces: barrier); the other four tiles of the prologue stay in flight behind the
    // counted wait while S(0) is computed.  The Q fragments pass through the statement.
    static_assert(2 * NJ2 <= 63, "vmcnt literal below");
    if constexpr (KS == 16) {
        asm volatile("s_waitcnt vmcnt(%c[nfl])\n\ts_barrier"
                     : "+a"(qf[0][0]), "+a"(qf[0][1]), "+a"(qf[0][2]), "+a"(qf[0][3]), "+a"(qf[0][4]), "+a"(qf[0][5]), "+a"(qf[0][6]), "+a"(qf[0][7])
                     : [nfl] "n"(2 * NJ2) : "memory");
        asm volatile("" : "+a"(qf[0][8]), "+a"(qf[0][9]), "+a"(qf[0][10]), "+a"(qf[0][11]), "+a"(qf[0][12]), "+a"(qf[0][13]), "+a"(qf[0][14]), "+a"(qf[0][15]) :: "memory");
        asm volatile("" : "+a"(qf[1][0]), "+a"(qf[1][1]), "+a"(qf[1][2]), "+a"(qf[1][3]), "+a"(qf[1][4]), "+a"(qf[1][5]), "+a"(qf[1][6]), "+a"(qf[1][7]) :: "memory");
        asm volatile("" : "+a"(qf[1][8]), "+a"(qf[1][9]), "+a"(qf[1][10]), "+a"(qf[1][11]), "+a"(qf[1][12]), "+a"(qf[1][13]), "+a"(qf[1][14]), "+a"(qf[1][15]) :: "memory");
    } else if constexpr (KS == 8) {
        asm volatile("s_waitcnt vmcnt(%c[nfl])\n\ts_barrier"
                     : "+a"(qf[0][0]), "+a"(qf[0][1]), "+a"(qf[0][2]), "+a"(qf[0][3]), "+a"(qf[0][4]), "+a"(qf[0][5]),
                       "+a"(qf[0][6]), "+a"(qf[0][7]), "+a"(qf[1][0]), "+a"(qf[1][1]), "+a"(qf[1][2]), "+a"(qf[1][3]),
                       "+a"(qf[1][4]), "+a"(qf[1][5]), "+a"(qf[1][6]), "+a"(qf[1][7])
                     : [nfl] "n"(2 * NJ2) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(%c[nfl])\n\ts_barrier"
                     : "+a"(qf[0][0]), "+a"(qf[0][1]), "+a"(qf[0][2]), "+a"(qf[0][3]), "+a"(qf[1][0]), "+a"(qf[1][1]),
                       "+a"(qf[1][2]), "+a"(qf[1][3])
                     : [nfl] "n"(2 * NJ2) : "memory");
    }

#if NNOP_W64_STAMP
    stamp_p[0] = __builtin_amdgcn_s_memtime();               // K(0) and Q have landed
#endif
    f32x16 sa[2][KB], sb[2][KB];                             // score tiles: current / next (roles swap every iteration)
    float mxa[2] = {-INFINITY, -INFINITY}, mxb[2] = {-INFINITY, -INFINITY};
    int need = 0;                                            // wave-uniform: the next tile raises a reference before its softmax
    frag_t fr[RF];                                           // fragment ring
    if (n_live > 0) {
        const uint32_t ka0 = opaque(kC + k_lane);
#pragma unroll
        for (int f = 0; f < NKF; ++f) {
            const frag_t a = read_kfrag(ka0, f);
#pragma unroll
            for (int z = 0; z < 2; ++z) {
                if (f % KS == 0) sa[z][f / KS] = kPre ? MM::qk_init(a, qf[z][f % KS], negm[z]) : MM::qk_first(a, qf[z][f % KS]);
                else MM::qk_acc(sa[z][f / KS], a, qf[z][f % KS]);
            }
        }
        fence_mfma_result(sa[0][0], sa[0][1], sa[1][0], sa[1][1]);
        if constexpr (kGeneral) {
            const uint64_t v0 = tile_valid(0);
            if (tile_needs_mask(0, v0)) apply_mask(sa, 0, v0);
        }
        mxa[0] = row_max(sa[0]);
        mxa[1] = row_max(sa[1]);
        need = rescale_test(mxa, true);
    }
#if NNOP_W64_STAMP
    stamp_p[1] = __builtin_amdgcn_s_memtime();               // S(0), its mask and row max are done
#endif
    // the rest of the prologue's tiles landed, every wave's pieces
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (n_live > 0) {
        // fragments 0 .. PF-1 of the first iteration's stream: K(1)
        const uint32_t ka1 = opaque(kA + k_lane);
#pragma unroll
        for (int f = 0; f < PF; ++f) fr[f] = read_kfrag(ka1, f);
    }

    // ---- one iteration: softmax + PV of tile t on `sc` (row max `mxc` known) beside QK^T of tile t+1 into `sn` ------
    // `need`: in -- tile t must raise a reference first (rescale_test of its row max); out -- the same for tile t+1.
    constexpr int LAG = NNOP_W64_LAG;
    using Plan = W64Plan<E, kSum, kGeneral, NJ2, LAG, EV>;
    auto iteration = [&](int t, int& need_io, f32x16 (&sc)[2][KB], const float (&mxc)[2], f32x16 (&sn)[2][KB], float (&mxn)[2]) {
        if (__builtin_expect(need_io != 0, 0)) rescale(mxc, sc, t == 0);
        float msub[2];
#pragma unroll
        for (int z = 0; z < 2; ++z) msub[z] = (kGeneral && m2[z] == -INFINITY) ? 0.f : m2[z];   // no key seen yet: P = 0
        const uint32_t kimg = opaque(kA + k_lane);                // K(t+1)
        const uint32_t vimg = opaque(vA + v_lane);                // V(t)
        const uint32_t kimg2 = opaque(kB + k_lane);               // K(t+2): the next iteration's first fragments
        u32x4 pw[2 * KB][2];                                      // P^T fragments of tile t as words: [16-key step kk][z]
        uint32_t ksoff = 0, vsoff = 0, kdst = 0, vdst = 0;       // this iteration's DMA batch: K(t+3), V(t+2)

        // softmax element n of tile t: chunk c = n / 8 = 2 kk + z, element j = n % 8 of that chunk.  Step n issues the
        // exp of element n and THEN finishes element n - LAG: the row-sum add and, for an odd element, the convert of the
        // pair it closes (one word of the P^T fragment) -- a consumer directly behind its v_exp_f32 stalls on the
        // transcendental unit's latency.
        auto sm_elem = [&](auto nc) {
            constexpr int n = decltype(nc)::value;
#if !(NNOP_W64_ABL & 4)
            if constexpr (n < 64) {
                constexpr int c = n >> 3, j = n & 7, kk = c >> 1, z = c & 1, kb = kk >> 1, i = 8 * (kk & 1) + j;
                float e = kPre ? fast_exp2(sc[z][kb][i]) : fast_exp2(__builtin_fmaf(sc[z][kb][i], c2, -msub[z]));
                pin(e);
                sc[z][kb][i] = e;
            }
#endif
            if constexpr (n >= LAG) {
                constexpr int m = n - LAG, c = m >> 3, j = m & 7, kk = c >> 1, z = c & 1, kb = kk >> 1, i = 8 * (kk & 1) + j;
#if !(NNOP_W64_ABL & 4)
                if constexpr (!kSum) {
                    lp[z][j & 1] += sc[z][kb][i];
                    pin(lp[z][j & 1]);
                }
#endif
                if constexpr (j & 1) {
                    typedef T t2 __attribute__((ext_vector_type(2)));
                    const f32x2 w = {sc[z][kb][i - 1], sc[z][kb][i]};
                    uint32_t word = __builtin_bit_cast(uint32_t, __builtin_convertvector(w, t2));
                    pin(word);
                    pw[kk][z][j >> 1] = word;
                }
            }
        };
        // fragment read PF ahead of stream position f (wraps into the next iteration's K fragments)
        auto read_ahead = [&](auto fc) {
            constexpr int g = decltype(fc)::value + PF;
#if !(NNOP_W64_ABL & 16)
            if constexpr (g < NKF) fr[g % RF] = read_kfrag(kimg, g);
            else if constexpr (g < NF) fr[g % RF] = read_vfrag(vimg, g - NKF);
            else fr[g % RF] = read_kfrag(kimg2, g - NF);
#endif
        };

        // row max of tile t+1, one query block, as 17 small items (4 independent v_max3 chains, then the combine) so that
        // they can be dealt out over the slots: item q < 16 folds two logits, item 16 finishes (scale, lane-half swap)
        float mxp[2][4];
        auto mx_item = [&](auto uc) {
            constexpr int u = decltype(uc)::value, z = u & 1, q = u >> 1;
#if !(NNOP_W64_ABL & 8)
            if constexpr (q < 16) {
                constexpr int kb = q >> 3, i0 = 2 * (q & 7);
                // single instructions: fmaxf() on values hipcc cannot prove canonical (asm MFMA results) costs an extra
                // canonicalising v_max_f32 x, x, x per operand
                if constexpr (q < 4) asm volatile("v_max_f32 %0, %1, %2" : "=v"(mxp[z][q]) : "v"(sn[z][kb][i0]), "v"(sn[z][kb][i0 + 1]));
                else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mxp[z][q & 3]) : "v"(sn[z][kb][i0]), "v"(sn[z][kb][i0 + 1]));
            } else {
                mxn[z] = half_swap_max(fmaxf(fmaxf(mxp[z][0], mxp[z][1]), fmaxf(mxp[z][2], mxp[z][3])) * (kPre ? 1.0f : c2));
                pin(mxn[z]);
            }
#endif
        };

        // -------- the schedule (W64Plan): softmax steps and row-max items per slot by issue cost -------------------------
        constexpr Plan plan = Plan::make();
        constexpr int G = Plan::G, NY2 = Plan::NY2, NYB = Plan::NYB;
        static_assert(Plan::NX == NX && plan.sm_end[Plan::NSLOT - 1] == Plan::NSTEP && plan.mx_end[Plan::NSLOT - 1] == Plan::NMX, "every item placed");
        // the first fragment of K(t+2) is read ahead from the PV slot of V fragment NVF - PF: behind the tile barrier
        static_assert(((NVF - PF) / EB) * G + (kSum ? 2 : 0) + 2 * ((NVF - PF) % EB) >= NYB, "K(t+2) reads behind the barrier");
        auto movable = [&](auto sc_) {                            // the slot's share of the two streams
            constexpr int sl = decltype(sc_)::value;
            constexpr int n0 = sl ? plan.sm_end[sl - 1] : 0, n1 = plan.sm_end[sl];
            static_for<n1 - n0>([&](auto dn) { sm_elem(std::integral_constant<int, n0 + decltype(dn)::value>{}); });
            constexpr int u0 = sl ? plan.mx_end[sl - 1] : 0, u1 = plan.mx_end[sl];
            static_for<u1 - u0>([&](auto du) { mx_item(std::integral_constant<int, u0 + decltype(du)::value>{}); });
        };

        // -------- phase X ----------------------------------------------------------------------------------------------
        static_for<NX>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int f = i >> 1, z = i & 1, kb = f / KS, ks = f % KS;
            if constexpr (z == 0) read_ahead(std::integral_constant<int, f>{});
            if constexpr (ks == 0) sn[z][kb] = kPre ? MM::qk_init(fr[f % RF], qf[z][ks], negm[z]) : MM::qk_first(fr[f % RF], qf[z][ks]);
            else MM::qk_acc(sn[z][kb], fr[f % RF], qf[z][ks]);
            __builtin_amdgcn_sched_barrier(0);                    // the MFMA opens its slot: nothing of the slot's VALU work above it
            movable(std::integral_constant<int, i>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        // -------- phase Y ----------------------------------------------------------------------------------------------
        // No explicit wait states are needed inside the loop: every P^T word is written (v_cvt_pk) at least one slot
        // (>= one MFMA issue) before the slot whose MFMA reads it, and the score tile `sn` is first read by VALU code two
        // MFMA slots after the last MFMA that wrote it -- the slot order is pinned by the sched_barrier(0) closing each slot;
        // tools/audit_w64.py checks both on the generated code.
        static_for<NY2>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int kk = i / G, w = i % G;
            constexpr bool is_sum = kSum && w < 2;
            constexpr int wp = w - (kSum ? 2 : 0);                // position among the PV slots of this kk
            constexpr int z = is_sum ? w : (wp & 1), eb = is_sum ? 0 : (wp >> 1), g = kk * EB + eb, f = NKF + g;
            if constexpr (i == NYB) {
                // tile barrier: this wave's DMA batch (issued behind the previous barrier) has landed; after the barrier
                // every wave's has, and every wave is done with the ring slots the next batch overwrites
#if NNOP_W64_ABL & 2
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#endif
            }
            // the next batch -- K(t+3), V(t+2): 2 NJ pieces -- one piece per odd slot behind the barrier (an LDS-DMA
            // instruction occupies the wave's issue for ~16 cycles; a burst of 2 NJ of them idles the matrix pipe)
            if constexpr (Plan::dma_at(i)) {
#if !(NNOP_W64_ABL & 1)
                constexpr int d = Plan::dma_index(i);
                if constexpr (d < NJK) issue_piece(krs, ksoff, kdst, k_voff[d < NJK ? d : 0], std::integral_constant<int, d>{});
                else issue_piece(vrs, vsoff, vdst, v_voff[NVO == 1 ? 0 : (d >= NJK ? d - NJK : 0)], std::integral_constant<int, d - NJK>{});
#endif
            }
            if constexpr (is_sum) {
                MM::pv_acc(lacc[z], ones, __builtin_bit_cast(frag_t, pw[kk][z]));   // ones[32 x 16] x P^T[16 x 32]: every row = sum over the 16 keys
            } else {
                if constexpr (z == 0) read_ahead(std::integral_constant<int, f>{});
                MM::pv_acc(oacc[z][eb], fr[f % RF], __builtin_bit_cast(frag_t, pw[kk][z]));
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NX + i == Plan::ADDR_SLOT) {
                ksoff = off_k3;
                vsoff = off_v2;
                kdst = kC;
                vdst = vC;
                advance_offsets();
                asm volatile("" : "+s"(off_k3));                  // computed HERE (scalar arithmetic sinks to its first use otherwise)
            }
            if constexpr (NX + i == Plan::MASK_SLOT && kGeneral) {
                const uint64_t vn = vword_take();                // fetched one iteration ago
                vword_fetch(t + 2);
                if (t + 1 < n_live && tile_needs_mask(t + 1, vn)) {
                    // rare (diagonal / ragged / padded tiles): mask, then redo the row-max items that already ran on the
                    // unmasked logits (the chains restart from their first item)
                    apply_mask(sn, t + 1, vn);
                    constexpr int done = plan.mx_end[Plan::MASK_SLOT - 1];
                    static_for<done>([&](auto du) { mx_item(du); });
                }
            }
            movable(std::integral_constant<int, NX + i>{});
            if constexpr (i == NY2 - 1) need_io = rescale_test(mxn, t + 1 < n_live);
            __builtin_amdgcn_sched_barrier(0);
        });
        rotate_slots();
    };

    // The loop body exists twice (the two score tiles swap roles) plus once more for an odd tile count.  Where these
    // copies meet -- loop exit, entry of the remainder -- hipcc's register allocator may give an O tile a different
    // accumulator tuple on either side and copy it on the edge (v_accvgpr_mov), i.e. directly behind the last asm MFMA of
    // the copy it leaves: a reader hipcc inserts, so it has no wait states in front of it (measured: stale registers 13..15
    // of one O tile at E = 64).  A wave therefore idles out its last MFMA at the END of the copy it is about to leave, before
    // the edge (the loop's exit branch leaves from the block that holds the fence); tools/audit_w64.py checks the generated code.
#ifdef NNOP_W64_NO_LEAVE_FENCE            // self-test of tools/audit_w64.py: it must flag the build without the fences
    auto leave_fence = []() {};
#else
    auto leave_fence = []() { asm volatile(NNOP_FENCE_128 ::: "memory"); __builtin_amdgcn_sched_barrier(0); };       // (nothing is scheduled across: the exit edge's register copies stay behind the idle time)
#endif
#if NNOP_W64_STAMP
    stamp[2] = __builtin_amdgcn_s_memtime();
    stamp[3] = __builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (kGeneral) vword_fetch(1);
    int t = 0;
    if (n_live >= 2) {
        for (;;) {
            iteration(t, need, sa, mxa, sb, mxb);
            iteration(t + 1, need, sb, mxb, sa, mxa);
            t += 2;
            if (t + 1 >= n_live) {       // the exit edge starts BEHIND the fence
                leave_fence();
                break;
            }
        }
    }
    if (t < n_live) {
        iteration(t, need, sa, mxa, sb, mxb);
        leave_fence();
        ++t;
    }
    // waves whose causal range ended early keep the workgroup's DMA / barrier schedule
    for (; t < n_tiles; ++t) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        issue_k(t + 3, kC);
        issue_v(t + 2, vC);
        rotate_slots();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if NNOP_W64_STAMP
    stamp[4] = __builtin_amdgcn_s_memtime();
    stamp[5] = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- epilogue: normalise, store o (16-byte stores: lane halves paired with v_permlane32_swap), ms, ls ------------
    fence_acc_result(oacc[0][0]);                          // the one fence of the epilogue; every other read is acc_after_fence
#pragma unroll
    for (int z = 0; z < 2; ++z) {
        float ltot;
        if constexpr (kSum) {
            acc_after_fence(lacc[z]);
            ltot = lacc[z][0];                             // the MFMA already summed the keys of both lane halves
        } else {
            ltot = half_swap_sum(lp[z][0] + lp[z][1]);
        }
        const float inv = 1.0f / ltot;                     // ltot == 0 (no visible key) -> NaN rows, as the naive formula gives
        T* orow = (T*)p.o + ((size_t)bh * p.QL + (qi[z] < p.QL ? qi[z] : p.QL - 1)) * E + vsplit * EV;
#pragma unroll
        for (int eb = 0; eb < EB; ++eb) {
            acc_after_fence(oacc[z][eb]);
            uint32_t pk[4][2];                             // [g][word]: this lane's 4 elements e = 32 eb + 8 g + 4 h + (0..3)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                typedef T t4 __attribute__((ext_vector_type(4)));
                const f32x4 w = {oacc[z][eb][4 * g] * inv, oacc[z][eb][4 * g + 1] * inv, oacc[z][eb][4 * g + 2] * inv,
                                 oacc[z][eb][4 * g + 3] * inv};
                const u32x2 u = __builtin_bit_cast(u32x2, __builtin_convertvector(w, t4));
                pk[g][0] = u[0];
                pk[g][1] = u[1];
            }
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                // lanes 0-31 end up with e = 32 eb + 8 g + (0..7), lanes 32-63 with e = 32 eb + 8 (g+1) + (0..7)
                const auto s0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
                const u32x4 lo = {s0[0], s1[0], s0[1], s1[1]};
                // s?[0]: vdst after the swap (lower lanes: own g; upper lanes: lower's g+1), s?[1]: src after the swap
                // (lower lanes: upper's g; upper lanes: own g+1)
#if NNOP_W64_NT
                if (qi[z] < p.QL) __builtin_nontemporal_store(lo, reinterpret_cast<u32x4*>(orow + 32 * eb + 8 * g + 8 * h));
#else
                if (qi[z] < p.QL) *reinterpret_cast<u32x4*>(orow + 32 * eb + 8 * g + 8 * h) = lo;
#endif
            }
        }
        if (qi[z] < p.QL && h == 0 && vsplit == 0) {
            // residual contract (src/attention.jl:128-129): ms = row max (natural-log units) rounded to T, ls relative to
            // the ROUNDED ms so that the pair stays self-consistent in 16-bit types
            const size_t so = (size_t)bh * p.QL + qi[z];
            const T m_t = from_f32<T>(mt[z] * kLn2);
            const float m_back = to_f32(m_t);
            float l_out = ltot;
            if (mt[z] != -INFINITY) l_out = ltot * fast_exp2(m2[z] - m_back * kLog2e);
            ((T*)p.ms)[so] = m_t;
            ((T*)p.ls)[so] = from_f32<T>(l_out);
        }
    }
#if NNOP_W64_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp[6] = __builtin_amdgcn_s_memtime();
    stamp[7] = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
        uint64_t* dbg = reinterpret_cast<uint64_t*>((T*)p.o + ((size_t)bh * p.QL + q0w) * E);
        for (int i = 0; i < 8; ++i) dbg[i] = stamp[i];
        dbg[8] = (uint64_t)n_tiles;
        dbg[9] = stamp_p[0];
        dbg[10] = stamp_p[1];
    }
#endif
    // the next block's prologue overwrites the rings and the validity words: every wave is done reading them
    if (pstep + 1 < n_steps_pers) __syncthreads();
    }   // blocks of this workgroup
}

}  // namespace nnop
