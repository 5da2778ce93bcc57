// fa_fwd_duo.hpp -- forward kernel, "two waves per SIMD in alternating phases" form (16-bit types; E = 64 with 64- or 32-row waves,
// E = 128 with 32-row waves, E = 32 with 64-row waves).  The text below describes the E = 64, 64-row form; the variants are at its end.
//
// What `_flash_attention_fwd!` computes (src/attention.jl:1-131; its hot loop :49-121), fourth program form.  Why it exists: at
// E = 64 one 32x32x16 MFMA covers only two score elements per lane, and each element costs one v_exp_f32 (8 cycles of the SIMD's
// vector port) plus ~3 plain VALU instructions (4 each for a single wave) -- a wave that owns its SIMD alone (fa_fwd_w64.hpp)
// issues 2067 cycles of instructions per 64 x 64 tile against 1280 cycles of matrix-pipe work, i.e. the WAVE, not the pipe, is the
// bound.  Two waves on a SIMD issue into the vector port independently, and one wave's VALU work runs beside the other's MFMAs
// (tools/ubench/pingpong.hip, profiles/r04/pingpong.log: the same instruction mix runs 2461 cycles per tile serially in one wave,
// 1482 as two waves in opposite phases).
//
//   * workgroup = 8 waves = 256 query rows of one (batch, q-head).  Waves w and w + 4 (SIMD partners: a workgroup's waves go to the
//     SIMDs cyclically) own the SAME 64 query rows and split the KEYS: group 0 (waves 0-3) takes the even kv tiles, group 1 the odd
//     ones, each with its own online-softmax state (reference, sum, O); the two partial results are merged once, through LDS, in
//     the epilogue (each partner finishes and stores 32 of the 64 rows).  64 rows per wave keep what the one-wave form has: every K /
//     V fragment read from LDS feeds two MFMAs (z = 0, 1).
//   * a wave alternates a MATRIX phase M(t) -- row sums of P(t-2) (8 MFMAs 16x16x32 with a selector operand, see below), O +=
//     V(t-2)^T P(t-2)^T (16 MFMAs), S(t) = K(t) Q^T (16), fragment reads three ahead, and at its END the LDS-DMA batch K(t+4), V(t+2) --
//     and a VECTOR phase V(t): [rare: mask] row max, test, [rare: raise the reference] P = exp2(s c - m), packed in place.  The matrix
//     phase holds NOTHING but MFMAs and fragment reads: a VALU or LDS-DMA instruction between them stalls the in-order wave while its
//     partner's vector phase holds the port, and the pipe idles (measured both ways: DESIGN.md section 4.1d, profiles/r04/
//     duo_ablations.log).  Group 1 runs a phase behind group 0, so each SIMD always holds one wave in M and one in V; ONE s_barrier per
//     iteration (group 0 behind V, group 1 behind M).  Nothing is software-pipelined INSIDE a wave; the hardware overlaps the partners.
//   * registers (256 per wave, all arch VGPRs): O^T (64), Q fragments (32), row-sum accumulators (8), one score tile (64) whose
//     registers also take the packed P^T words, a fragment ring (16), state and temporaries (32); 32 are hipcc's across the loop.
//     hipcc could not be made to allocate this (given virtual 16-register tuples it moved whole tiles between phases and spilled the
//     Q fragments: 228-660 bytes of scratch per lane in every C++ form tried; naming the accumulator file halves the arch budget), so
//     every tile has a HOME register and the whole loop is ONE asm statement per mode, generated with those registers
//     (tools/gen_duo_asm.py -> fa_fwd_duo_asm.inc, register map and loop structure in the generator's header); hipcc copies each tile
//     in once.  It pads nothing inside asm: the stream carries its own s_waitcnt and wait states, audited by the generator
//     (check_stream) and tests/test_duo_codegen.py.
//   * K / V rings of 6 slots per tensor, 3 per key group (read now / landed or landing / free), filled by LDS-DMA (the group that
//     reads a tile also copies it): the batch K(t+4), V(t+2) issued at the tail of M(t) goes to the group's free slots (what they
//     held was read two barriers back), is waited for one iteration later (vmcnt(4)) and read in M(t+4) / M(t+4).
//
//   * variants (template parameter NZ = 32-row query blocks per wave; the generator's set_nz):
//     NZ = 1, E = 64   the same loop without its z = 1 half: 32 rows per wave, 128 per workgroup -- for launches whose 256-row blocks
//                      would leave CUs idle (fa_launch.hpp small_grid_prefers_32_row_waves).  The partners finish the same 32 rows, group 0
//                      the first half of the columns and the residuals, group 1 the second.
//     NZ = 2, E = 32   the E = 64 loop with two contraction steps and one column block of O^T per query block (8 + 8 + 8 MFMAs per tile against
//                      the same 128 logits per lane: the vector phase is the bound outright); grids of >= 256 blocks.
//     NZ = 1, E = 128  O^T is again 64 registers (32 rows x 128 columns), Q eight fragments; tiles of 16 KiB -> 2 ring slots per key group,
//                      the LDS-DMA batch (8 pieces per wave) in the VECTOR phase behind the barrier that closes the matrix phase whose
//                      slots it overwrites, a barrier behind every phase.  LDS-bound per tile (slower than fa_fwd_w64.hpp on grids that
//                      fill the chip), launched on small grids only (fa_fwd_inst.hpp fwd_form_of).
//
// Modes: 0 plain / 1 masked (causal, key padding, ragged KL).  Exact fp32 scale only.  Same numerics contract as the other forms
// (fp32 softmax, deferred row max with threshold 2^8, O normalised once, residuals ms / ls per src/attention.jl:128-129); the
// summation order over keys differs from the one-wave form (two partial sums per row), so results agree to rounding, not bitwise.
#pragma once
#include "fa_fwd_w64.hpp"
#include "fa_fwd_duo_asm.inc"

#if !defined(NNOP_DEV_BUILD)
#undef NNOP_DUO_STAMP
#undef NNOP_DUO_PRIO
#endif
#ifndef NNOP_DUO_STAMP
#define NNOP_DUO_STAMP 0
#endif
#ifndef NNOP_DUO_PRIO
#define NNOP_DUO_PRIO 0
#endif

namespace nnop {

// Operands of the generated loop statement: every tile in its home register (register map: tools/gen_duo_asm.py)
#if NNOP_DUO_VALU_SUMS
#define NNOP_DUO_SUMS_OPERAND , "+{v[248:251]}"(lsum)
#else
#define NNOP_DUO_SUMS_OPERAND
#endif
#if NNOP_DUO_STAMP
#define NNOP_DUO_PROF_OPERAND , "+{v[224:231]}"(profv)
#else
#define NNOP_DUO_PROF_OPERAND
#endif
#define NNOP_DUO_OPERANDS                                                                                                        \
    "+{v[0:15]}"(oacc[0][0]), "+{v[16:31]}"(oacc[0][1]), "+{v[32:47]}"(oacc[1][0]), "+{v[48:63]}"(oacc[1][1]), "+{v[64:79]}"(qf[0]),   \
        "+{v[80:95]}"(qf[1]), "+{v[96:99]}"(lacc[0]), "+{v[100:103]}"(lacc[1]), "+{v[104:107]}"(sel), "+{v[112:127]}"(sc[0][0]),      \
        "+{v[128:143]}"(sc[0][1]), "+{v[144:159]}"(sc[1][0]), "+{v[160:175]}"(sc[1][1]), "+{v[192:195]}"(mstate), [st] "+s"(s_t),      \
        [ska] "+s"(s_ka), [skb] "+s"(s_kb), [skc] "+s"(s_kc), [sva] "+s"(s_va), [svb] "+s"(s_vb), [svc] "+s"(s_vc) NNOP_DUO_PROF_OPERAND NNOP_DUO_SUMS_OPERAND \
        : "{v[196:203]}"(vconst), [sh] "s"(s_h), [snlive] "s"(s_nlive), [slast] "s"(s_last), [sc2] "s"(c2), [scq0] "s"(s_cq0),            \
          [svbits] "s"(s_vbits), [krs] "s"(krs), [vrs] "s"(vrs)                                                                       \
        : "memory", "vcc", "scc", "v108", "v109", "v110", "v111", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", \
          "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212",   \
          "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "s56", "s57", "s58", "s59", "s60", "s61",  \
          "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77"

// NZ = 1: the z = 0 tiles only; v[144:175] (the z = 1 score tile's registers) are the loop's 8-slot fragment ring
#define NNOP_DUO1_OPERANDS                                                                                                       \
    "+{v[0:15]}"(oacc[0][0]), "+{v[16:31]}"(oacc[0][1]), "+{v[64:79]}"(qf[0]), "+{v[96:99]}"(lacc[0]), "+{v[104:107]}"(sel),              \
        "+{v[112:127]}"(sc[0][0]), "+{v[128:143]}"(sc[0][1]), "+{v[192:195]}"(mstate), [st] "+s"(s_t), [ska] "+s"(s_ka), [skb] "+s"(s_kb), \
        [skc] "+s"(s_kc), [sva] "+s"(s_va), [svb] "+s"(s_vb), [svc] "+s"(s_vc)                                                         \
        : "{v[196:203]}"(vconst), [sh] "s"(s_h), [snlive] "s"(s_nlive), [slast] "s"(s_last), [sc2] "s"(c2), [scq0] "s"(s_cq0),            \
          [svbits] "s"(s_vbits), [krs] "s"(krs), [vrs] "s"(vrs)                                                                       \
        : "memory", "vcc", "scc", "v108", "v109", "v110", "v111", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", \
          "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168",   \
          "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v204",   \
          "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220",   \
          "v221", "v222", "v223", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70",    \
          "s71", "s72", "s73", "s74", "s75", "s76", "s77"

// E = 32 (NZ = 2): one column block of O^T per query block (v[0:15], v[32:47]), two Q fragments each (v[64:71], v[80:87]); the rest as E = 64
#define NNOP_DUO32_OPERANDS                                                                                                      \
    "+{v[0:15]}"(oacc[0][0]), "+{v[32:47]}"(oacc[NZ - 1][0]), "+{v[64:79]}"(qf[0]), "+{v[80:95]}"(qf[NQT - 1]), "+{v[96:99]}"(lacc[0]),  \
        "+{v[100:103]}"(lacc[NZ - 1]), "+{v[104:107]}"(sel), "+{v[112:127]}"(sc[0][0]), "+{v[128:143]}"(sc[0][1]),                       \
        "+{v[144:159]}"(sc[NZ - 1][0]), "+{v[160:175]}"(sc[NZ - 1][1]), "+{v[192:195]}"(mstate), [st] "+s"(s_t), [ska] "+s"(s_ka),        \
        [skb] "+s"(s_kb), [skc] "+s"(s_kc), [sva] "+s"(s_va), [svb] "+s"(s_vb), [svc] "+s"(s_vc)                                        \
        : "{v[196:203]}"(vconst), [sh] "s"(s_h), [snlive] "s"(s_nlive), [slast] "s"(s_last), [sc2] "s"(c2), [scq0] "s"(s_cq0),            \
          [svbits] "s"(s_vbits), [krs] "s"(krs), [vrs] "s"(vrs)                                                                       \
        : "memory", "vcc", "scc", "v108", "v109", "v110", "v111", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", \
          "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212",   \
          "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "s56", "s57", "s58", "s59", "s60", "s61",  \
          "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77"

// E = 128 (NZ = 1): O^T = 4 column blocks, Q = 8 fragments; v[100:103] more K fragment addresses, v[180:182] more DMA offsets
#define NNOP_DUO128_OPERANDS                                                                                                     \
    "+{v[0:15]}"(oacc[0][0]), "+{v[16:31]}"(oacc[0][1]), "+{v[32:47]}"(oacc[0][EB - 2]), "+{v[48:63]}"(oacc[0][EB - 1]),                  \
        "+{v[64:79]}"(qf[0]), "+{v[80:95]}"(qf[NQT - 1]), "+{v[96:99]}"(lacc[0]), "+{v[104:107]}"(sel), "+{v[112:127]}"(sc[0][0]),        \
        "+{v[128:143]}"(sc[0][1]), "+{v[192:195]}"(mstate), [st] "+s"(s_t), [ska] "+s"(s_ka), [skb] "+s"(s_kb), [skc] "+s"(s_kc),         \
        [sva] "+s"(s_va), [svb] "+s"(s_vb), [svc] "+s"(s_vc)                                                                            \
        : "{v[196:203]}"(vconst), [sh] "s"(s_h), [snlive] "s"(s_nlive), [slast] "s"(s_last), [sc2] "s"(c2), [scq0] "s"(s_cq0),            \
          [svbits] "s"(s_vbits), [krs] "s"(krs), [vrs] "s"(vrs)                                                                       \
        : "memory", "vcc", "scc", "v100", "v101", "v102", "v103", "v108", "v109", "v110", "v111", "v144", "v145", "v146", "v147", "v148",  \
          "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164",   \
          "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180",   \
          "v181", "v182", "v183", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216",   \
          "v217", "v218", "v219", "v220", "v221", "v222", "v223", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65",     \
          "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77"

constexpr int kDuoXchgBytes = 8 * (8192 + 3 * 256);          // epilogue exchange: per wave 32 fp32 per lane + (l, m2, mt)
// ring slots per key group and ring / barriers per iteration: properties of the generated loop of an embedding dim (fa_fwd_duo_asm.inc)
template <int E> constexpr int duo_slots_per_group() { return E == 128 ? NNOP_DUO128_SLOTS_PER_GROUP : NNOP_DUO_SLOTS_PER_GROUP; }
template <int E> constexpr bool duo_sync_one() { return (E == 128 ? NNOP_DUO128_SYNC_ONE : NNOP_DUO_SYNC_ONE) != 0; }
template <typename T, int E> constexpr int fa_fwd_duo_lds_bytes(bool masked) {
    constexpr int ring = 2 * duo_slots_per_group<E>() * (RowImg<T, E>::bytes(64) + ColImg<T, E>::bytes(64));
    return (ring > kDuoXchgBytes ? ring : kDuoXchgBytes) + (masked ? 16 + 8 * kMaxMaskTiles : 0);
}

// NZ: 32-row query blocks per wave.  2: 64 rows per wave, 256 per workgroup (the form described above).  1: the same loop with the z = 1
// half left out -- 32 rows per wave, 128 per workgroup, for problems whose 256-row blocks cannot fill the chip (twice the workgroups;
// every fragment then feeds one MFMA and the per-iteration overheads are paid per 32 rows: ~8 % more cycles per row).
template <typename T, int E, int MODE, int NZ = 2>
__global__ __launch_bounds__(512) void fa_fwd_duo_kernel(const FwdParams p) {
    static_assert(sizeof(T) == 2 && ((E == 64 && (NZ == 1 || NZ == 2)) || (E == 128 && NZ == 1) || (E == 32 && NZ == 2)),
                  "16-bit element types; E = 64, E = 128 with 32-row waves, E = 32 with 64-row waves");
    constexpr int RW = 32 * NZ, RB = 4 * RW;                  // query rows per wave / per workgroup
    using frag_t = typename Elem<T>::frag;
    using KImg   = RowImg<T, E>;
    using VImg   = ColImg<T, E>;
    constexpr bool kGeneral = MODE != 0;
    constexpr int BK = 64, KB = 2, KS = E / 16, EB = E / 32, NS = 2 * duo_slots_per_group<E>();
    constexpr int KBYTES = KImg::bytes(BK), VBYTES = VImg::bytes(BK);
    constexpr int RING = NS * (KBYTES + VBYTES);
    constexpr int MASK_OFF = RING > kDuoXchgBytes ? RING : kDuoXchgBytes;
    constexpr int TILE_BYTES = BK * E * (int)sizeof(T);
    constexpr int NJK = KBYTES / 4096, NJV = VBYTES / 4096;   // DMA pieces per wave and tile (the 4 waves of a group copy a tile)
    static_assert(NJK * 4096 == KBYTES && NJV * 4096 == VBYTES && NJK <= 4 && NJV <= 4, "four waves x NJ pieces = one image");

    extern __shared__ __attribute__((aligned(16))) char smem[];
#if NNOP_DUO_STAMP
    uint64_t stamp[8];
    stamp[0] = __builtin_amdgcn_s_memtime();
    stamp[1] = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;                 // key group (tile parity) / 64-row slice of the block
    const int r = lane & 31, h = lane >> 5;

    // ---- persistent form (p.persist = blocks per workgroup; 0: one block per workgroup): the static, balanced block list of
    // fa_fwd_w64.hpp -- XCD x owns an eighth of the (batch, q-head) columns, its blocks (q-blocks descending inside a column) are dealt
    // out 32 at a time alternately forwards and backwards over its 32 workgroups ----------------------------------------------------
    const int n_steps_pers = (kGeneral && p.persist > 0) ? p.persist : 1;
    for (int pstep = 0; pstep < n_steps_pers; ++pstep) {
    int qblk, bh;
    if (kGeneral && p.persist > 0) {
        const int x = (int)blockIdx.x & 7, c = (int)blockIdx.x >> 3;
        const int pos = 32 * pstep + ((pstep & 1) ? 31 - c : c);
        const int col = pos / p.n_qblk;
        qblk = p.persist_asc ? pos - col * p.n_qblk : p.n_qblk - 1 - (pos - col * p.n_qblk);
        if (p.persist_hx > 0) bh = (col / p.persist_hx) * p.QH + x * p.persist_hx + col % p.persist_hx;
        else bh = x * ((p.B * p.QH) >> 3) + col;
    } else {
        const int lin = xcd_remap_chunked((int)blockIdx.x, p.n_wg, p.n_qblk * (p.QH / p.KH));
        qblk = lin % p.n_qblk;
        bh = lin / p.n_qblk;
        if (kGeneral && p.causal) qblk = p.n_qblk - 1 - qblk; // heaviest q-blocks first
    }
    const int b = bh / p.QH, qh = bh - b * p.QH;
    const int kvh = qh / (p.QH / p.KH);                       // cld(q_head, n_q_per_kv), 0-based (src/attention.jl:28)
    const int q0w = qblk * RB + wq * RW;                      // first query row of this wave (and of its partner)
    int qi[NZ];
#pragma unroll
    for (int z = 0; z < NZ; ++z) qi[z] = q0w + 32 * z + r;

    const T* __restrict__ qp = (const T*)p.q + ((size_t)bh * p.QL) * E;
    const char* __restrict__ kp = (const char*)((const T*)p.k + ((size_t)(b * p.KH + kvh) * p.KL) * E);
    const char* __restrict__ vp = (const char*)((const T*)p.v + ((size_t)(b * p.KH + kvh) * p.KL) * E);
    const uint8_t* __restrict__ mp = kGeneral && p.kpad ? p.kpad + (size_t)b * p.KL : nullptr;

    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const uint32_t kring = lds0, vring = lds0 + NS * KBYTES;
    uint64_t* const vbits = reinterpret_cast<uint64_t*>(smem + MASK_OFF + 16);

    // ---- number of kv tiles (workgroup) / live tiles (this wave): as in fa_fwd_w64.hpp ---------------------------------------------
    int n_tiles = (p.KL + BK - 1) / BK;
    int causal_q0 = 0x3fffffff;
    int qlim[2] = {0x3fffffff, 0x3fffffff};                  // (entry 1 unused at NZ = 1)
    if constexpr (kGeneral) {
        if (p.causal) {
            int q_last = qblk * RB + RB - 1;
            if (q_last > p.QL - 1) q_last = p.QL - 1;
            const int t_c = q_last / BK + 1;
            if (t_c < n_tiles) n_tiles = t_c;
            causal_q0 = q0w;
#pragma unroll
            for (int z = 0; z < NZ; ++z) qlim[z] = qi[z];
        }
        if (mp) {
            int* slot = reinterpret_cast<int*>(smem + MASK_OFF);
            const int nk = n_tiles * BK < p.KL ? n_tiles * BK : p.KL;
            const int last = kpad_scan(mp, p.KL, nk, vbits, kMaxMaskTiles, slot, tid, 512);
            const int t_m = last / BK + 1;
            if (t_m < n_tiles) n_tiles = t_m;
        } else {
            for (int w = tid; w < n_tiles; w += 512) {
                const int left = p.KL - w * BK;
                vbits[w] = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
            }
            __syncthreads();
        }
    }
    int n_live = n_tiles;
    if (kGeneral && p.causal) {
        const int t_w = (q0w + RW - 1) / BK + 1;
        if (t_w < n_live) n_live = t_w;
    }

    // ---- per-lane DMA source offsets inside a tile (the image's layout applied to the SOURCE; fa_fwd_w64.hpp) ---------------------
    uint32_t k_voff[NJK];
#pragma unroll
    for (int j = 0; j < NJK; ++j) {
        const int off = (wq * NJK + j) * 1024 + lane * 16;
        const int row = off / KImg::kRowBytes, phys = (off % KImg::kRowBytes) >> 4;
        k_voff[j] = (uint32_t)(row * KImg::kRowBytes + ((phys ^ KImg::xor_of(row)) << 4) - j * 1024);
    }
    uint32_t v_voff;
    {
        const int off = (wq * NJV) * 1024 + lane * 16;
        const int blk = off >> 8, rg = blk / VImg::kEB, eb = blk % VImg::kEB, rr = (off >> 6) & 3, c4 = (off >> 4) & 3;
        v_voff = (uint32_t)((4 * rg + rr) * KImg::kRowBytes + ((4 * eb + c4) << 4));
    }
    static_assert((1024 / 256) % VImg::kEB == 0 && (4 * (1024 / 256 / VImg::kEB)) * VImg::kRowBytes == 1024, "V image: 1 KiB = whole row groups");
    const uint32_t wave_off_k = (uint32_t)(wq * NJK * 1024), wave_off_v = (uint32_t)(wq * NJV * 1024);
    const uint32_t kv_bytes = (uint32_t)p.KL * (uint32_t)KImg::kRowBytes;
    const u32x4 krs = make_rsrc(kp, kv_bytes), vrs = make_rsrc(vp, kv_bytes);
    // past the last tile the LAST tile is copied again (into a ring slot nobody reads any more): no branch around an issue
    auto tile_off = [&](int t) -> uint32_t { return (uint32_t)(t < n_tiles ? t : n_tiles - 1) * (uint32_t)TILE_BYTES; };
    auto issue_k_piece = [&](uint32_t soff, uint32_t dst, auto jc) {
        constexpr int j = decltype(jc)::value;
        dma_piece<j, j == 0>(krs, k_voff[j], soff, dst);
    };
    // Rings of NS slots, NS / 2 per key group.  At iteration t (LDS byte address + this wave's DMA share): kA = slot of K(t), read in M(t);
    // kB = slot of K(t+2); kC = the free slot, target of K(t+4) (with 2 slots per group: kA again, and the batch is then issued behind the
    // barrier that closes M(t)).  vA = slot of V(t-2), read in M(t); vB = slot of V(t); vC = free, target of V(t+2).  The group's slots
    // rotate (A, B, C) <- (B, C, A) per iteration.
    constexpr int SPG = duo_slots_per_group<E>();
    const uint32_t kA = kring + wave_off_k + (uint32_t)(SPG * grp) * KBYTES, kB = kA + KBYTES, kC = kA + (SPG - 1) * KBYTES;
    const uint32_t vA = vring + wave_off_v + (uint32_t)(SPG * grp) * VBYTES, vB = vA + VBYTES, vC = vA + (SPG - 1) * VBYTES;
    // ragged KL: rows of the last tile past KL are outside the descriptor's range -- the ring must not hold non-finite garbage there
    if (kGeneral && (p.KL & (BK - 1)) != 0) {
        for (int i = tid * 16; i < RING; i += 512 * 16) *reinterpret_cast<u32x4*>(smem + i) = u32x4{0, 0, 0, 0};
        __syncthreads();
    }
    // ---- prologue: the group's first tiles in flight -- K(g), K(g+2), V(g) (the loop's first batch is K(g+4), V(g+2)) -----------------
    static_for<NJK>([&](auto jc) { issue_k_piece(tile_off(grp), kA, jc); });
    const float c2 = p.scale * kLog2e;
    // Q fragments: asm loads (invisible to hipcc's wait-count bookkeeping, like the LDS-DMA around them), valid behind the counted wait
    // below, which takes them as operands
    f32x4 qw[NZ][KS];
#pragma unroll
    for (int z = 0; z < NZ; ++z) {
        const int qc = qi[z] < p.QL ? qi[z] : p.QL - 1;
        const T* qrow = qp + (size_t)qc * E;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qw[z][ks]) : "v"(qrow + 16 * ks + 8 * h) : "memory");
    }
    f32x16 oacc[NZ][EB];
    f32x4 lacc[NZ];                                            // row sums: registers 0 / 1 of lanes 0..15 = queries lane, lane + 16 (SumMfma)
#pragma unroll
    for (int z = 0; z < NZ; ++z) {
#pragma unroll
        for (int eb = 0; eb < EB; ++eb)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[z][eb][i] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) lacc[z][i] = 0.f;
    }
    // Row sums of P on the matrix pipe with the 16x16x32 shape: half the pipe time and a quarter of the accumulator registers of a
    // ones x P^T product in the 32x32x16 shape.  The B operand is a P^T fragment AS IT IS (lane (r, h) = lane l: 8 keys of query r): the
    // 16x16x32 instruction reads lane l as column l % 16, contraction group l / 16, i.e. it would add queries r and r + 16 together --
    // unless the A operand separates them: row m of A is one where contraction group g has g % 2 == m (rows 2..15 zero), so
    //   D[0][n] = sum over the keys of query n,   D[1][n] = the same for query n + 16      (n = 0..15),
    // which land in accumulator registers 0 and 1 of lanes 0..15 (the other lanes and registers hold zero rows).
    f32x4 sel;
    {
        frag_t sf;
#pragma unroll
        for (int j = 0; j < 8; ++j) sf[j] = from_f32<T>((((lane >> 4) & 1) == (lane & 15)) ? 1.0f : 0.0f);
        sel = __builtin_bit_cast(f32x4, sf);
    }
    static_for<NJK>([&](auto jc) { issue_k_piece(tile_off(grp + 2), kB, jc); });
    static_for<NJV>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        dma_piece<j, j == 0>(vrs, v_voff, tile_off(grp), vB);
    });
    const uint32_t k_lane = (uint32_t)(r * KImg::kRowBytes + ((KImg::xor_of(r) ^ h) << 4)) - wave_off_k;
    const uint32_t v_lane = (uint32_t)VImg::lane_base(lane) - wave_off_v;
    // K(grp) and Q landed (every wave's pieces: barrier); K(grp+2) and V(grp) -- the NJK + NJV pieces issued last -- stay in flight: the
    // loop's first counted wait (end of V(grp)) retires them, in time for M(grp+2).  The Q fragments pass through the statement.
    if constexpr (E == 32)
        asm volatile("s_waitcnt vmcnt(%c[n])\n\ts_barrier" : "+v"(qw[0][0]), "+v"(qw[0][KS - 1]), "+v"(qw[NZ - 1][0]), "+v"(qw[NZ - 1][KS - 1]) : [n] "n"(NJK + NJV) : "memory");
    else if constexpr (E == 128)
        asm volatile("s_waitcnt vmcnt(%c[n])\n\ts_barrier"
                     : "+v"(qw[0][0]), "+v"(qw[0][1]), "+v"(qw[0][2]), "+v"(qw[0][3]), "+v"(qw[0][KS - 4]), "+v"(qw[0][KS - 3]), "+v"(qw[0][KS - 2]),
                       "+v"(qw[0][KS - 1])
                     : [n] "n"(NJK + NJV) : "memory");
    else if constexpr (NZ == 2)
        asm volatile("s_waitcnt vmcnt(%c[n])\n\ts_barrier"
                     : "+v"(qw[0][0]), "+v"(qw[0][1]), "+v"(qw[0][2]), "+v"(qw[0][3]), "+v"(qw[NZ - 1][0]), "+v"(qw[NZ - 1][1]), "+v"(qw[NZ - 1][2]),
                       "+v"(qw[NZ - 1][3])
                     : [n] "n"(NJK + NJV) : "memory");
    else
        asm volatile("s_waitcnt vmcnt(%c[n])\n\ts_barrier" : "+v"(qw[0][0]), "+v"(qw[0][1]), "+v"(qw[0][2]), "+v"(qw[0][3]) : [n] "n"(NJK + NJV) : "memory");
    // the fragments (16-deep steps of E) as 16-register tuples: v[64:79], v[80:95] = query blocks 0, 1 at E = 64 / steps 0-3, 4-7 at E = 128
    // (E = 32: two fragments per query block, in the first half of that block's tuple)
    constexpr int NQT = E == 32 ? NZ : NZ * KS / 4;
    f32x16 qf[NQT];
    if constexpr (E == 32) {
#pragma unroll
        for (int z = 0; z < NZ; ++z)
#pragma unroll
            for (int i = 0; i < 16; ++i) qf[z][i] = i < 4 * KS ? qw[z][(i >> 2) % KS][i & 3] : 0.f;
    } else {
#pragma unroll
        for (int z = 0; z < NZ; ++z)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) qf[(z * KS + ks) / 4][4 * (ks & 3) + i] = qw[z][ks][i];
    }
#if NNOP_DUO_STAMP
    stamp[2] = __builtin_amdgcn_s_memtime();
    stamp[3] = __builtin_amdgcn_s_memrealtime();
#endif

    // the score tile S(t)^T: [z][key block].  V(t) packs P(t)^T IN PLACE: the 8 logits of 16-key step kk (registers 8 (kk & 1) .. + 7 of
    // sc[z][kk >> 1]) become 4 operand words in the first 4 of those registers.
    f32x16 sc[NZ][KB];
#pragma unroll
    for (int z = 0; z < NZ; ++z)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) sc[z][kb][i] = 0.f;

    // ---- the phase loop (generated, tools/gen_duo_asm.py): half-steps h = 0 .. n_tiles + 1, one barrier each; group g runs the matrix
    // phase M(t) at h = t for t = g (mod 2) -- [row sums of P(t-2)] [O += V(t-2)^T P(t-2)^T] [S(t) = K(t) Q^T], the LDS-DMA of K(t+2) and
    // V(t) in its gaps -- and the vector phase V(t) at h = t + 1: mask, row max, (rare) raise of the reference, P = exp2(s c - m) packed
    // as MFMA operand words ------------------------------------------------------------------------------------------------------------
    f32x4 mstate = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};       // m2[0..1] (exponent reference), mt[0..1] (true row max)
    typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
    u32x8 vconst;
    vconst[0] = k_voff[0]; vconst[1] = k_voff[NJK > 1 ? 1 : 0]; vconst[2] = v_voff; vconst[3] = k_lane; vconst[4] = v_lane;
    vconst[5] = (uint32_t)qlim[0]; vconst[6] = (uint32_t)qlim[1]; vconst[7] = (uint32_t)(4 * h);
    // the loop's scalar state (wave-uniform: hipcc hands them over in scalar registers)
    int s_t = grp, s_h = n_tiles + 2, s_nlive = n_live, s_cq0 = causal_q0;
    uint32_t s_ka = kA, s_kb = kB, s_kc = kC, s_va = vA, s_vb = vB, s_vc = vC;
    const uint32_t s_last = (uint32_t)(n_tiles - 1) * (uint32_t)TILE_BYTES, s_vbits = (uint32_t)(uintptr_t)vbits;
#if NNOP_DUO_STAMP
    f32x8 profv = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#endif
#if NNOP_DUO_VALU_SUMS
    f32x4 lsum = {0.f, 0.f, 0.f, 0.f};                        // row sums, VALU form: two chains per query block, this lane's half of the keys
#endif
#if NNOP_DUO_PRIO == 2
    if (grp) __builtin_amdgcn_s_setprio(1);
#endif
    if constexpr (!duo_sync_one<E>()) {
        if (grp) asm volatile("s_barrier" ::: "memory");      // half-step 0: group 1 has nothing to do yet
    }
#if NNOP_DUO_STAMP
#define NNOP_DUO_LOOP_MASKED NNOP_DUO_LOOP_MASKED_PROF
#define NNOP_DUO_LOOP_PLAIN NNOP_DUO_LOOP_PLAIN_PROF
#endif
    // (experiments, make DEV=1 VAR=-DNNOP_DUO_PRIO=n: 1 matrix phase at s_setprio 1, 2 waves 4-7 at priority 1 throughout, 3 vector phase at 1)
#if NNOP_DUO_PRIO == 1
#define NNOP_DUO_PRIO_ARGS "s_setprio 1", "s_setprio 0", "", ""
#elif NNOP_DUO_PRIO == 3
#define NNOP_DUO_PRIO_ARGS "", "", "s_setprio 1", "s_setprio 0"
#else
#define NNOP_DUO_PRIO_ARGS "", "", "", ""
#endif
#define NNOP_DUO_X(M, ...) M(__VA_ARGS__)
    if constexpr (E == 32) {
        if constexpr (std::is_same<T, __bf16>::value) {
            if constexpr (kGeneral) asm volatile(NNOP_DUO_X(NNOP_DUO32_LOOP_MASKED, "bf16", NNOP_DUO_PRIO_ARGS) : NNOP_DUO32_OPERANDS);
            else asm volatile(NNOP_DUO_X(NNOP_DUO32_LOOP_PLAIN, "bf16", NNOP_DUO_PRIO_ARGS) : NNOP_DUO32_OPERANDS);
        } else {
            if constexpr (kGeneral) asm volatile(NNOP_DUO_X(NNOP_DUO32_LOOP_MASKED, "f16", NNOP_DUO_PRIO_ARGS) : NNOP_DUO32_OPERANDS);
            else asm volatile(NNOP_DUO_X(NNOP_DUO32_LOOP_PLAIN, "f16", NNOP_DUO_PRIO_ARGS) : NNOP_DUO32_OPERANDS);
        }
    } else if constexpr (NZ == 2) {
        if constexpr (std::is_same<T, __bf16>::value) {
            if constexpr (kGeneral) asm volatile(NNOP_DUO_X(NNOP_DUO_LOOP_MASKED, "bf16", NNOP_DUO_PRIO_ARGS) : NNOP_DUO_OPERANDS);
            else asm volatile(NNOP_DUO_X(NNOP_DUO_LOOP_PLAIN, "bf16", NNOP_DUO_PRIO_ARGS) : NNOP_DUO_OPERANDS);
        } else {
            if constexpr (kGeneral) asm volatile(NNOP_DUO_X(NNOP_DUO_LOOP_MASKED, "f16", NNOP_DUO_PRIO_ARGS) : NNOP_DUO_OPERANDS);
            else asm volatile(NNOP_DUO_X(NNOP_DUO_LOOP_PLAIN, "f16", NNOP_DUO_PRIO_ARGS) : NNOP_DUO_OPERANDS);
        }
    } else if constexpr (E == 128) {
        if constexpr (std::is_same<T, __bf16>::value) {
            if constexpr (kGeneral) asm volatile(NNOP_DUO_X(NNOP_DUO128_LOOP_MASKED, "bf16", NNOP_DUO_PRIO_ARGS) : NNOP_DUO128_OPERANDS);
            else asm volatile(NNOP_DUO_X(NNOP_DUO128_LOOP_PLAIN, "bf16", NNOP_DUO_PRIO_ARGS) : NNOP_DUO128_OPERANDS);
        } else {
            if constexpr (kGeneral) asm volatile(NNOP_DUO_X(NNOP_DUO128_LOOP_MASKED, "f16", NNOP_DUO_PRIO_ARGS) : NNOP_DUO128_OPERANDS);
            else asm volatile(NNOP_DUO_X(NNOP_DUO128_LOOP_PLAIN, "f16", NNOP_DUO_PRIO_ARGS) : NNOP_DUO128_OPERANDS);
        }
    } else {
        if constexpr (std::is_same<T, __bf16>::value) {
            if constexpr (kGeneral) asm volatile(NNOP_DUO_X(NNOP_DUO1_LOOP_MASKED, "bf16", NNOP_DUO_PRIO_ARGS) : NNOP_DUO1_OPERANDS);
            else asm volatile(NNOP_DUO_X(NNOP_DUO1_LOOP_PLAIN, "bf16", NNOP_DUO_PRIO_ARGS) : NNOP_DUO1_OPERANDS);
        } else {
            if constexpr (kGeneral) asm volatile(NNOP_DUO_X(NNOP_DUO1_LOOP_MASKED, "f16", NNOP_DUO_PRIO_ARGS) : NNOP_DUO1_OPERANDS);
            else asm volatile(NNOP_DUO_X(NNOP_DUO1_LOOP_PLAIN, "f16", NNOP_DUO_PRIO_ARGS) : NNOP_DUO1_OPERANDS);
        }
    }
    // (the loop keeps the true row max per lane half -- lane l ^ 32 holds the same query's other keys: combined here, once)
    float m2[NZ], mt[NZ];
#pragma unroll
    for (int z = 0; z < NZ; ++z) {
        m2[z] = mstate[z];
        mt[z] = half_swap_max(mstate[2 + z]);
    }
#if NNOP_DUO_STAMP
    stamp[4] = __builtin_amdgcn_s_memtime();
    stamp[5] = __builtin_amdgcn_s_memrealtime();
#endif
    // the rings are dead from here on (the exchange buffer overlays them): every DMA landed, every wave past its last fragment read
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    // ---- epilogue: the partners exchange one 32-row half each through LDS, merge the two key groups, normalise, store --------------
    asm volatile(NNOP_FENCE_128 ::: "memory");                // the last MFMAs have written O and the row sums
    __builtin_amdgcn_sched_barrier(0);
    // wave w writes the half it gives away into its own exchange block; after the barrier it reads its partner's block (the other
    // key group's partial result for the rows it keeps)
    char* const mine = smem + wave * (8192 + 3 * 256);
    const char* const theirs = smem + (wave ^ 4) * (8192 + 3 * 256);
#if NNOP_DUO_VALU_SUMS
    lacc[0][0] = half_swap_sum(lsum[0] + lsum[1]);            // (both lane halves: lane l ^ 32 holds the same query's other keys)
    if constexpr (NZ == 2) lacc[NZ - 1][0] = half_swap_sum(lsum[2] + lsum[3]);
    auto row_sum = [&](const f32x4& l) -> float { return l[0]; };
#else
    auto row_sum = [&](const f32x4& l) -> float {            // this lane's query r: lane r % 16, register r / 16
        const float l0 = __shfl(l[0], r & 15), l1 = __shfl(l[1], r & 15);
        return (r & 16) ? l1 : l0;
    };
#endif
    // what a wave gives away / keeps: at NZ = 2 a whole 32-row block z (all E columns), at NZ = 1 one 32-column half eb of its only block
    auto give = [&](auto givec, auto eb0c, auto nebc) {
        constexpr int ZG = decltype(givec)::value, EB0 = decltype(eb0c)::value, NEB = decltype(nebc)::value;
#pragma unroll
        for (int eb = EB0; eb < EB0 + NEB; ++eb) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 w = {oacc[ZG][eb][4 * g4], oacc[ZG][eb][4 * g4 + 1], oacc[ZG][eb][4 * g4 + 2], oacc[ZG][eb][4 * g4 + 3]};
                *reinterpret_cast<f32x4*>(mine + (((eb - EB0) * 4 + g4) * 64 + lane) * 16) = w;
            }
        }
        float* sm = reinterpret_cast<float*>(mine + 8192);
        sm[lane] = row_sum(lacc[ZG]);
        sm[64 + lane] = m2[ZG];
        sm[128 + lane] = mt[ZG];
    };
    auto take = [&](auto keepc, auto eb0c, auto nebc, auto statsc) {
        constexpr int ZK = decltype(keepc)::value, EB0 = decltype(eb0c)::value, NEB = decltype(nebc)::value;
        constexpr bool kStats = decltype(statsc)::value;
        const float* so_ = reinterpret_cast<const float*>(theirs + 8192);
        const float l_o = so_[lane], m_o = so_[64 + lane], mt_o = so_[128 + lane];
        const float l_m = row_sum(lacc[ZK]), m_m = m2[ZK];
        const float mm = fmaxf(m_m, m_o);
        const float a = m_m == -INFINITY ? 0.f : fast_exp2(m_m - mm);
        const float bsc = m_o == -INFINITY ? 0.f : fast_exp2(m_o - mm);
        const float ltot = a * l_m + bsc * l_o;
        const float mtt = fmaxf(mt[ZK], mt_o);
        const float inv = 1.0f / ltot;                       // ltot == 0 (no visible key) -> NaN rows, as the naive formula gives
        const float ai = a * inv, bi = bsc * inv;
        T* orow = (T*)p.o + ((size_t)bh * p.QL + (qi[ZK] < p.QL ? qi[ZK] : p.QL - 1)) * E;
#pragma unroll
        for (int eb = EB0; eb < EB0 + NEB; ++eb) {
            uint32_t pk[4][2];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                typedef T t4 __attribute__((ext_vector_type(4)));
                const f32x4 ot = *reinterpret_cast<const f32x4*>(theirs + (((eb - EB0) * 4 + g4) * 64 + lane) * 16);
                const f32x4 w = {oacc[ZK][eb][4 * g4] * ai + ot[0] * bi, oacc[ZK][eb][4 * g4 + 1] * ai + ot[1] * bi,
                                 oacc[ZK][eb][4 * g4 + 2] * ai + ot[2] * bi, oacc[ZK][eb][4 * g4 + 3] * ai + ot[3] * bi};
                const u32x2 u = __builtin_bit_cast(u32x2, __builtin_convertvector(w, t4));
                pk[g4][0] = u[0];
                pk[g4][1] = u[1];
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; g4 += 2) {
                // lanes 0-31 end up with e = 32 eb + 8 g + (0..7), lanes 32-63 with e = 32 eb + 8 (g+1) + (0..7)
                const auto x0 = __builtin_amdgcn_permlane32_swap(pk[g4][0], pk[g4 + 1][0], false, false);
                const auto x1 = __builtin_amdgcn_permlane32_swap(pk[g4][1], pk[g4 + 1][1], false, false);
                const u32x4 lo = {x0[0], x1[0], x0[1], x1[1]};
                if (qi[ZK] < p.QL) *reinterpret_cast<u32x4*>(orow + 32 * eb + 8 * g4 + 8 * h) = lo;
            }
        }
        if (kStats && qi[ZK] < p.QL && h == 0) {
            // residual contract (src/attention.jl:128-129): ms = row max (natural-log units) rounded to T, ls relative to the ROUNDED ms
            const size_t so = (size_t)bh * p.QL + qi[ZK];
            const T m_t = from_f32<T>(mtt * kLn2);
            const float m_back = to_f32(m_t);
            float l_out = ltot;
            if (mtt != -INFINITY) l_out = ltot * fast_exp2(mm - m_back * kLog2e);
            ((T*)p.ms)[so] = m_t;
            ((T*)p.ls)[so] = from_f32<T>(l_out);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using IE = std::integral_constant<int, EB>;
    if constexpr (NZ == 2) {
        // group 0 keeps the rows of z = 0 and gives z = 1 away, group 1 the other way round
        if (grp == 0) give(I1{}, I0{}, IE{});
        else give(I0{}, I0{}, IE{});
        __syncthreads();
        if (grp == 0) take(I0{}, I0{}, IE{}, std::true_type{});
        else take(I1{}, I0{}, IE{}, std::true_type{});
    } else {
        // both partners finish the same 32 rows: group 0 the first half of the columns (and the residuals), group 1 the second
        using IH = std::integral_constant<int, EB / 2>;       // (E = 64: 1 block of 32 columns each, E = 128: 2)
        if (grp == 0) give(I0{}, IH{}, IH{});
        else give(I0{}, I0{}, IH{});
        __syncthreads();
        if (grp == 0) take(I0{}, I0{}, IH{}, std::true_type{});
        else take(I0{}, IH{}, IH{}, std::false_type{});
    }
#if NNOP_DUO_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp[6] = __builtin_amdgcn_s_memtime();
    stamp[7] = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
        uint64_t* dbg = reinterpret_cast<uint64_t*>((T*)p.o + ((size_t)bh * p.QL + q0w) * E);
        for (int i = 0; i < 8; ++i) dbg[i] = stamp[i];
        dbg[8] = (uint64_t)n_tiles;
        dbg[9] = stamp[2];
        dbg[10] = stamp[2];
        for (int i = 0; i < 5; ++i) dbg[11 + i] = (uint64_t)__float_as_uint(profv[i]);     // cycles in M, barrier, V, DMA wait, barrier (wave 0)
    }
    if (tid == 256) {                                         // the same five of wave 4 (key group 1), in the block's second row
        uint64_t* dbg = reinterpret_cast<uint64_t*>((T*)p.o + ((size_t)bh * p.QL + q0w + 1) * E);
        for (int i = 0; i < 5; ++i) dbg[i] = (uint64_t)__float_as_uint(profv[i]);
    }
#endif
    // the next block's prologue overwrites the rings / the exchange buffer / the validity words: every wave is done with them
    if (pstep + 1 < n_steps_pers) __syncthreads();
    }   // blocks of this workgroup
}

}  // namespace nnop
