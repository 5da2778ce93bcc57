// fa_fwd_inst.hpp -- host-side dispatch for the forward kernel; included by the three per-dtype
// translation units so that they compile in parallel.
//
// Replaces the reference's host launcher heuristics (src/attention.jl:146-163, :193-218): there the
// workgroup size is the largest of (256..16) whose backward LDS footprint fits; here the tile is
// fixed by the MFMA shape (32 queries per wave, BK keys per step) and only the number of waves per
// workgroup is chosen, from how many workgroups the problem yields against the chip's 256 CUs.
#pragma once
#include "fa_fwd.hpp"
#include "fa_fwd_split.hpp"
#include "fa_fwd_w64.hpp"
#include "fa_fwd_duo.hpp"
#include "fa_generic.hpp"
#ifdef NNOP_DEV_BUILD
#include <stdlib.h>
#endif
#include "fa_launch.hpp"
#include <math.h>

namespace nnop {

#ifdef NNOP_DEV_BUILD
static inline int dev_env_int(const char* name, int dflt) {
    const char* s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}
#endif

template <typename T, int E, int NW, int MODE, int QB>
static int launch_fwd_cfg(const nnop_fa_desc& d, const FwdArgs& a, hipStream_t s) {
    // 64-key tiles, except the E = 128 pair-bias body (register budget) and fp32 E = 128
    // (LDS: 2 x (K + V) x 64 keys x 512 B = 128 KiB would leave one workgroup per CU)
    constexpr int BK = (E >= 256 || (E >= 128 && (MODE == 2 || sizeof(T) == 4))) ? 32 : 64;
    constexpr int lds = fa_fwd_lds_bytes<T, E, BK>();
    static_assert(lds <= 160 * 1024, "LDS budget (160 KiB per CU on gfx950)");
    auto kern = fa_fwd_kernel<T, E, NW, BK, MODE, QB>;
    static unsigned long long lds_done = 0;
    if (ensure_dynamic_lds(kern, lds, &lds_done) != NNOP_OK) return NNOP_ERR_HIP;
    FwdParams p;
    p.o = a.o; p.ms = a.ms; p.ls = a.ls;
    p.q = a.q; p.k = a.k; p.v = a.v; p.pair = a.pair; p.kpad = a.kpad;
    p.QL = d.ql; p.KL = d.kl; p.QH = d.qh; p.KH = d.kh; p.B = d.batch;
    p.causal = d.causal ? 1 : 0;
    constexpr int rows = 32 * QB * NW;
    p.n_qblk = (d.ql + rows - 1) / rows;
    const long long n_wg = (long long)p.n_qblk * d.qh * d.batch;
    if (n_wg <= 0 || n_wg > 0x7fffffffLL) return NNOP_ERR_SHAPE;
    p.n_wg = (int)n_wg;
    p.scale = (float)(1.0 / sqrt((double)E));        // T(inv(sqrt(QE))), src/attention.jl:154
    // Causal launches of a few rounds: pair heavy with light blocks on a CU (fa_launch.hpp causal_alt_run).  Measured (round 4,
    // profiles/r04/causal_alt.log): fp32 E64 L4096 H4 B4 545.6 -> 319.6 us, fp32 E32 301 -> 180, fp32 E16 233 -> 142, bf16 E16 69.8 -> 53.0,
    // fp32 E64 L2048 H8 B8 (4 rounds) 473 -> 428; GQA at 4 rounds -5 %, 8 rounds and more -12 % (the rule stops before).  Where ONE workgroup
    // fills a CU there is nothing to pair (E = 128 with 8 waves; fp32 E = 128 with 4 waves fits two).
    p.causal_alt = (MODE != 2 && (E <= 64 || (E == 128 && sizeof(T) == 4 && NW == 4)))
                       ? causal_alt_run(d.causal != 0, n_wg, (long long)d.qh * d.batch, d.qh / d.kh, p.n_qblk) : 0;
    int lds_launch = lds;
#ifdef NNOP_DEV_BUILD
    // experiments (make DEV=1 only): de-phase co-resident workgroups; pad LDS to limit workgroups per CU
    p.stagger = dev_env_int("NNOP_FWD_STAGGER", 0);
    lds_launch += dev_env_int("NNOP_FWD_LDS_PAD", 0);
    if (lds_launch > lds && lds_launch > 64 * 1024 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_launch) != hipSuccess) {
        (void)hipGetLastError();
        return NNOP_ERR_HIP;
    }
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(NW * 64), lds_launch, s, p);
    return hipGetLastError() == hipSuccess ? NNOP_OK : NNOP_ERR_HIP;
}

// split-KV form (fa_fwd_split.hpp): 16 waves per workgroup, plain mode, 16-bit types, E <= 64.
// (A v_mfma_f32_16x16x32 body of this form was built and measured 7 % slower in round 1 -- twice the MFMA issues on the port
// that is the bottleneck; DESIGN.md section 5 -- and is no longer in the tree.)
template <typename T, int E>
static int launch_fwd_split(const nnop_fa_desc& d, const FwdArgs& a, hipStream_t s) {
    constexpr int lds = fa_fwd_split_lds_bytes<T, E>();
    void (*kern)(const FwdParams) = fa_fwd_split_kernel<T, E>;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static unsigned long long lds_done = 0;
    if (ensure_dynamic_lds(kern, lds, &lds_done) != NNOP_OK) return NNOP_ERR_HIP;
    FwdParams p;
    p.o = a.o; p.ms = a.ms; p.ls = a.ls;
    p.q = a.q; p.k = a.k; p.v = a.v; p.pair = nullptr; p.kpad = nullptr;
    p.QL = d.ql; p.KL = d.kl; p.QH = d.qh; p.KH = d.kh; p.B = d.batch;
    p.causal = 0;
    p.n_qblk = (d.ql + 255) / 256;
    const long long n_wg = (long long)p.n_qblk * d.qh * d.batch;
    if (n_wg <= 0 || n_wg > 0x7fffffffLL) return NNOP_ERR_SHAPE;
    p.n_wg = (int)n_wg;
    p.scale = (float)(1.0 / sqrt((double)E));
    hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(1024), lds, s, p);
    return hipGetLastError() == hipSuccess ? NNOP_OK : NNOP_ERR_HIP;
}

// Persistent form of the masked-mode kernels (fa_fwd_w64.hpp, fa_fwd_duo.hpp): 256 workgroups that each walk `persist` blocks of a
// static, balanced list instead of one workgroup per block -- on a 256-CU device, when the list divides: (batch x q-head) columns in
// eighths (one per XCD), a power-of-two number of q-blocks per column, whole steps of 32 blocks per XCD, and more than one step.
// Fills p.persist / p.persist_hx and the grid when the form applies.
static inline void fwd_persist_plan(const nnop_fa_desc& d, const FwdArgs& a, FwdParams& p, long long& grid) {
    const long long bh = (long long)d.batch * d.qh;
    const int n = p.n_qblk;
    const long long per_xcd = (bh / 8) * n;
    // (auto: under a causal mask without key padding -- there the static list balances exactly and measured +3..4 % at E = 128
    // L 8192-16384, +23 % at E = 64 L4096 H16 B4; equal-work blocks gain nothing (+-0.3 %), and per-batch key lengths make a
    // static list 15 % SLOWER than the dispatcher's dynamic order: C4)
    const int knob = tune_get(kTuneFwdPersist);
    // per-batch key lengths WITHOUT a causal mask: balanced as well once every XCD takes an eighth of the heads of every batch
    const int hx = d.qh % 8 == 0 ? d.qh / 8 : 0;
    const bool pays = (d.causal && !a.kpad) || (!d.causal && a.kpad && hx > 0);
    if ((knob == 1 || (knob < 0 && pays)) && device_cu_count() == 256 && bh % 8 == 0 && (n & (n - 1)) == 0 &&
        per_xcd % 32 == 0 && per_xcd / 32 >= 2 && per_xcd / 32 <= (1 << 24)) {
        p.persist = (int)(per_xcd / 32);
        p.persist_hx = hx;
        // Order of a column's q-blocks: descending (heaviest first).  The ascending order (light blocks first, so that every
        // workgroup starts at kv tile 0 together and the heavy blocks follow staggered inside each other's L2 window) was built and
        // measured in round 4: same time (+-0.1 %: C3 3912 vs 3918 us, C5 shard 15084 vs 15100 us), MORE fabric traffic at the C5
        // shard (7.93 vs 6.11 GB per launch; C3 2.34 vs 2.32 GB) -- profiles/r04/persist_order.log.  Knob kTuneFwdPersistAsc (1 = ascending).
        {
            const int asc = tune_get(kTuneFwdPersistAsc);
            p.persist_asc = asc >= 0 ? (asc != 0) : 0;
        }
        grid = 256;
    }
}

// 64-rows-per-wave form (fa_fwd_w64.hpp): 4 waves x 64 rows, 16-bit types, E = 64 / 128, plain and masked modes
template <typename T, int E, int MODE, bool PRE>
static int launch_fwd_w64(const nnop_fa_desc& d, const FwdArgs& a, hipStream_t s) {
    constexpr int EV = E == 256 ? 128 : E;                   // E = 256: two 128-column halves of O per block (fa_fwd_w64.hpp)
    constexpr int lds = fa_fwd_w64_lds_bytes<T, E, EV>(MODE != 0);
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = fa_fwd_w64_kernel<T, E, MODE, PRE, EV>;
    static unsigned long long lds_done = 0;
    if (ensure_dynamic_lds(kern, lds, &lds_done) != NNOP_OK) return NNOP_ERR_HIP;
    FwdParams p;
    p.o = a.o; p.ms = a.ms; p.ls = a.ls;
    p.q = a.q; p.k = a.k; p.v = a.v; p.pair = nullptr; p.kpad = a.kpad;
    p.QL = d.ql; p.KL = d.kl; p.QH = d.qh; p.KH = d.kh; p.B = d.batch;
    p.causal = d.causal ? 1 : 0;
    p.n_qblk = (d.ql + 255) / 256;
    const long long n_wg = (long long)p.n_qblk * d.qh * d.batch;
    if (n_wg <= 0 || n_wg > 0x7fffffffLL) return NNOP_ERR_SHAPE;
    p.n_wg = (int)n_wg;
    p.scale = (float)(1.0 / sqrt((double)E));
    if (n_wg * (E / EV) > 0x7fffffffLL) return NNOP_ERR_SHAPE;
    long long grid = n_wg * (E / EV);
    if constexpr (EV == E && MODE == 1) fwd_persist_plan(d, a, p, grid);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, p);
    return hipGetLastError() == hipSuccess ? NNOP_OK : NNOP_ERR_HIP;
}

// two waves per SIMD in alternating phases (fa_fwd_duo.hpp): 8 waves, 256 query rows per workgroup, 16-bit types, E = 64
// (NZ = 1: 32 rows per wave, 128 per workgroup -- fwd_duo_nz below)
template <typename T, int E, int MODE, int NZ>
static int launch_fwd_duo(const nnop_fa_desc& d, const FwdArgs& a, hipStream_t s) {
    constexpr int lds = fa_fwd_duo_lds_bytes<T, E>(MODE != 0);
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = fa_fwd_duo_kernel<T, E, MODE, NZ>;
    static unsigned long long lds_done = 0;
    if (ensure_dynamic_lds(kern, lds, &lds_done) != NNOP_OK) return NNOP_ERR_HIP;
    FwdParams p;
    p.o = a.o; p.ms = a.ms; p.ls = a.ls;
    p.q = a.q; p.k = a.k; p.v = a.v; p.pair = nullptr; p.kpad = a.kpad;
    p.QL = d.ql; p.KL = d.kl; p.QH = d.qh; p.KH = d.kh; p.B = d.batch;
    p.causal = d.causal ? 1 : 0;
    p.n_qblk = (d.ql + 128 * NZ - 1) / (128 * NZ);
    const long long n_wg = (long long)p.n_qblk * d.qh * d.batch;
    if (n_wg <= 0 || n_wg > 0x7fffffffLL) return NNOP_ERR_SHAPE;
    p.n_wg = (int)n_wg;
    p.scale = (float)(1.0 / sqrt((double)E));
    long long grid = n_wg;
    if constexpr (MODE == 1) fwd_persist_plan(d, a, p, grid);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, s, p);
    return hipGetLastError() == hipSuccess ? NNOP_OK : NNOP_ERR_HIP;
}

template <typename T, int E, int NW, int QB>
static int launch_fwd_mode(const nnop_fa_desc& d, const FwdArgs& a, hipStream_t s, int mode) {
    if (mode == 0) return launch_fwd_cfg<T, E, NW, 0, QB>(d, a, s);
    if (mode == 1) return launch_fwd_cfg<T, E, NW, 1, QB>(d, a, s);
    return launch_fwd_cfg<T, E, NW, 2, QB>(d, a, s);
}

// The launcher's choice of kernel form, as a plain function of the descriptor (also reported through nnop_debug_fwd_form).
//   mode: 0 plain (every logit live), 1 masked (causal / key padding / ragged KL), 2 + pair bias
static inline int fwd_mode(const nnop_fa_desc& d, bool has_pair, bool has_mask) {
    return has_pair ? 2 : ((d.causal || has_mask || (d.kl % 64) != 0) ? 1 : 0);
}
// Rows per wave of the two-waves-per-SIMD form at E = 64: 64 (NZ = 2, 256-row workgroups), or 32 (NZ = 1: twice the workgroups) where the
// 256-row blocks leave CUs idle (small_grid_prefers_32_row_waves, fa_launch.hpp; L1024 H8 B4: 19.8 -> 14.0 us; L2048 H4 B4: 30.9 -> 21.4 us;
// causal L4096 H8 B2: 57.6 -> 45.1 us; L8192 H8 B1: 104.6 -> 79.6 us).  Knob kTuneFwdDuo: 2 / 3 force NZ = 2 / 1.
static inline int fwd_duo_nz(const nnop_fa_desc& d) {
    const int duo = tune_get(kTuneFwdDuo);
    if (duo == 2) return 2;
    if (duo == 3) return 1;
    return small_grid_prefers_32_row_waves(d.ql, (long long)d.qh * d.batch, d.causal != 0) ? 1 : 2;
}
static inline int fwd_form_of(const nnop_fa_desc& d, int mode) {
    const bool b16 = d.dtype != NNOP_F32;
    const int E = d.emb;
    // the early exits of launch_fwd: embedding dims outside the tiled set (16-bit E = 256 runs the 32-row tiled kernel)
    if (E != 256 && emb_generic(E)) return kFormGeneric;
    const long long wg256 = (long long)((d.ql + 255) / 256) * d.qh * d.batch;
    if (b16 && E == 256) {
        // E = 256: the 64-row form with two 128-column halves of O per block (spill-free; the 32-row form spills 62-152 registers
        // and measured 214 TFLOP/s at L2048 H8 B2).  No pair bias; masked mode up to 64 Ki keys; a few kv tiles.
        const int w64 = tune_get(kTuneFwdW64);
        const bool fits = mode != 2 && (mode == 0 || d.kl <= 64 * kMaxMaskTiles) && (long long)d.kl * E * 2 < (1LL << 32);
        if (fits && w64 != 0 && d.kl >= 128) return kFormW64;
    }
    if (b16 && (E == 64 || E == 128)) {
        // 64-row waves (fa_fwd_w64.hpp): 4 waves x 64 rows, one wave per SIMD with the whole register file.  Measured
        // against the 32-row forms after the round-2 schedule work (bf16 / fp16, MI355X, tools/w64_check.py): faster wherever
        // there are a few kv tiles and enough 256-row workgroups -- E = 64 plain +13..21 % from KL = 256 and 64 workgroups
        // up; E = 64 causal / padded +10..20 % from KL = 512 and 128 workgroups; E = 128 +9..47 % from KL = 256, except plain
        // problems with <= 128 workgroups (-3 %: the 32-row form splits them into twice as many) and KL = 128 (-33 %: the
        // prologue).  Masked mode needs the per-tile validity words in LDS (KL <= 64 Ki), the pair-bias mode stays on the
        // 32-row kernel.  Knob kTuneFwdW64: 0 never, 1 wherever instantiated, auto = the rule below.
        const int w64 = tune_get(kTuneFwdW64);
        // (its LDS-DMA addresses a (batch, kv-head) tensor through a 32-bit buffer offset)
        const bool fits = mode != 2 && (mode == 0 || d.kl <= 64 * kMaxMaskTiles) && (long long)d.kl * E * 2 < (1LL << 32);
        const bool masked = mode == 1;
        const bool pays = E == 128 ? (d.kl >= 256 && (wg256 >= 160 || (masked && wg256 >= 64)))
                                   : (masked ? (d.kl >= 512 && wg256 >= 128) : (d.kl >= 256 && wg256 >= 64));
        // E = 64, exact scale: two waves per SIMD in alternating matrix / vector phases (fa_fwd_duo.hpp).  Measured against the
        // one-wave form on one box (tools/duo_check.py, profiles/r04/duo_sweep.log): 10-22 % faster from KL = 1024 up in plain mode at
        // every grid size (32 .. 4096 workgroups), 3-20 % faster in masked mode from KL = 2048 (from KL = 1024 with >= 128
        // workgroups); slower below (KL = 512: +10..19 %: the prologue holds three tiles and the epilogue merges two key groups).
        // Knob kTuneFwdDuo: 0 never, 1 wherever instantiated, auto = this rule.
        if (E == 128 && fits && tune_get(kTuneFwdExactScale) != 0) {
            // E = 128: the two-wave form exists with 32-row waves only (fa_fwd_duo.hpp: 128-row workgroups whose partner waves split the
            // keys).  Per tile it is 10-14 % SLOWER than the one-wave form (every fragment read feeds one MFMA: LDS-bound; C3 4388 vs
            // 3972 us), but it has twice the workgroups and halves a block's critical path: measured (profiles/r04/duo128_sweep.log)
            // 1.1-1.3x faster while the 128-row blocks fit one round (L2048 H8 B2: 42.9 -> 36.5 us), 1.3-1.7x under a causal mask
            // (L2048 H8 B2: 54.0 -> 33.4 us; L4096 H8 B1: 97.2 -> 57.8; L8192 H8 B1: 189 -> 135), and still ahead up to two rounds in
            // masked mode from KL = 2048 (causal L2048 H8 B4: 56.6 -> 44.3; lens L2048 B4: 68.8 -> 62.4).  KL = 256: a tie.
            const int duo = tune_get(kTuneFwdDuo);
            const long long w1 = (long long)((d.ql + 127) / 128) * d.qh * d.batch;
            const long long cus = device_cu_count() > 0 ? device_cu_count() : 256;
            const bool duo_pays = d.kl >= 512 && (w1 <= cus || (masked && d.kl >= 2048 && w1 <= 2 * cus));
            if (duo >= 1 || (duo < 0 && w64 != 1 && duo_pays)) return kFormDuo;
        }
        if (E == 64 && fits && tune_get(kTuneFwdExactScale) != 0) {
            const int duo = tune_get(kTuneFwdDuo);
            // With 32-row waves (small grids, fwd_duo_nz) it wins from KL = 256 in every mode: 9-11 us against 13-16 us at KL = 512, 64-256
            // 128-row blocks; 7.4-9.4 us against 9.6-10.4 us at KL = 256 (profiles/r04/nz1_small.log).
            const bool duo_pays = fwd_duo_nz(d) == 1 ? d.kl >= 256 : (masked ? (d.kl >= 2048 || (d.kl >= 1024 && wg256 >= 128)) : d.kl >= 1024);
            if (duo >= 1 || (duo < 0 && w64 != 1 && duo_pays)) return kFormDuo;
        }
        if (fits && (w64 == 1 || (w64 < 0 && pays))) return kFormW64;
    }
    if (b16 && E == 32 && mode != 2 && (mode == 0 || d.kl <= 64 * kMaxMaskTiles) && (long long)d.kl * E * 2 < (1LL << 32)) {
        // E = 32: the two-waves-per-SIMD form with 64-row waves (fa_fwd_duo.hpp).  The softmax, not the matrix pipe, is the bound of every
        // form at E = 32 (they take as long as at E = 64).  Measured (profiles/r04/duo32_sweep.log), grids of >= 256 blocks: masked mode
        // 1.1-1.5x faster from KL = 1024 (causal L2048 H8 B8 61.2 -> 40.8 us, L4096 H8 B8 164 -> 125; key padding +2..12 %), plain mode
        // +3..6 % from KL = 2048; smaller grids lose (the 32-row forms have twice the workgroups).  Knob kTuneFwdDuo.
        const int duo = tune_get(kTuneFwdDuo);
        const bool duo_pays = wg256 >= 256 && (mode == 1 ? d.kl >= 1024 : d.kl >= 2048);
        if (duo >= 1 || (duo < 0 && duo_pays)) return kFormDuo;
    }
    if (b16 && E <= 64) {
        // plain mode: 16-wave split-KV workgroups (4 waves per SIMD); measured 5-13 % faster than the 8-wave form from 64 to
        // 4096 workgroups.  Knob kTuneFwdSplit: 0 off, 1 / auto on.
        if (mode == 0 && d.ql > 128 && d.kl >= 128 && tune_get(kTuneFwdSplit) != 0) return kFormSplit;
    }
    return kFormRow32;
}

template <typename T, int E>
static int launch_fwd_e(const nnop_fa_desc& d, const FwdArgs& a, hipStream_t s) {
    const int mode = fwd_mode(d, a.pair != nullptr, a.kpad != nullptr);
    const int form = fwd_form_of(d, mode);
    // Workgroup shape of the 32-row form: 8 waves x 32 rows (256-row workgroups) when that still yields >= one
    // workgroup per CU, else 4 waves x 32 rows so that small problems spread over more CUs.
    const long long wg256 = (long long)((d.ql + 255) / 256) * d.qh * d.batch;
    if constexpr (sizeof(T) == 2 && E == 256) {
        if (form == kFormW64) return mode == 0 ? launch_fwd_w64<T, E, 0, false>(d, a, s) : launch_fwd_w64<T, E, 1, false>(d, a, s);    // exact scale only
    }
    if constexpr (sizeof(T) == 2 && E == 128) {
        if (form == kFormDuo) return mode == 0 ? launch_fwd_duo<T, E, 0, 1>(d, a, s) : launch_fwd_duo<T, E, 1, 1>(d, a, s);
    }
    if constexpr (sizeof(T) == 2 && E == 32) {
        if (form == kFormDuo) return mode == 0 ? launch_fwd_duo<T, E, 0, 2>(d, a, s) : launch_fwd_duo<T, E, 1, 2>(d, a, s);
    }
    if constexpr (sizeof(T) == 2 && E == 64) {
        if (form == kFormDuo) {
            if (fwd_duo_nz(d) == 1) return mode == 0 ? launch_fwd_duo<T, E, 0, 1>(d, a, s) : launch_fwd_duo<T, E, 1, 1>(d, a, s);
            return mode == 0 ? launch_fwd_duo<T, E, 0, 2>(d, a, s) : launch_fwd_duo<T, E, 1, 2>(d, a, s);
        }
    }
    if constexpr (sizeof(T) == 2 && (E == 64 || E == 128)) {
        if (form == kFormW64) {
            // The scale.  DEFAULT: exact -- scale * log2(e) applied in fp32 inside the exponent (one v_fma per logit).  Opt-in
            // (kTuneFwdExactScale = 0 / NNOP_FWD_EXACT_SCALE=0): folded into Q, rounded to T once, 8-12 % faster -- but every
            // channel of a query then carries a relative rounding of 2^-9 (bf16) / 2^-12 (fp16), and two near-tied keys that load
            // DIFFERENT channels get different errors: measured (tools/fold_sweep.py, profiles/r03/fold_sweep.log) the folded form
            // leaves the standard parity tolerance from |s * scale| ~ 5 (one outlier channel per key) / ~ 20 (dense directions) at
            // E = 64 -- inside what trained models produce -- while the exact form stays at 0.3 of the tolerance up to 65.  The
            // reference scales S itself in T (src/attention.jl:55), i.e. is coarser than either; parity is judged against the oracle.
            const bool exact = tune_get(kTuneFwdExactScale) != 0;
            if (mode == 0) return exact ? launch_fwd_w64<T, E, 0, false>(d, a, s) : launch_fwd_w64<T, E, 0, true>(d, a, s);
            return exact ? launch_fwd_w64<T, E, 1, false>(d, a, s) : launch_fwd_w64<T, E, 1, true>(d, a, s);
        }
    }
    if constexpr (sizeof(T) == 2 && E <= 64) {
        if (form == kFormSplit) return launch_fwd_split<T, E>(d, a, s);
    }
    int nw = 8, qb = 1;
    if (wg256 < 256 || d.ql <= 128) nw = 4;
    // causal, E <= 64: the waves of a 256-row workgroup see 1..4 x the keys of its first wave and wait for the last one
    // at every barrier; 128-row workgroups waste half as much.  Measured (bf16, MI355X): E64 L4096 H16 B4 356 -> 304 us
    // (+17 %).  Only while that still leaves >= 2 workgroups per CU.  At E = 128 and long sequences the 8-wave form is
    // equal (C3: 5.60 vs 5.62-5.70 ms) or better (C5 shard: 20.4 vs 21.2-21.6 ms).
    // At E = 128 only for short sequences (QL <= 2048: L1024 314 -> 269 us, L2048 485 -> 469 us; from L4096 on the 8-wave
    // form wins: 723 vs 762-858 us).
    {
        const long long wg128 = (long long)((d.ql + 127) / 128) * d.qh * d.batch;
        if (d.causal && wg128 >= 512 && (E <= 64 || d.ql <= 2048)) nw = 4;
    }
    // fp32 E = 128 under a causal mask: two 4-wave workgroups fit a CU where one 8-wave workgroup does, and with the alternating block order
    // (launch_fwd_cfg) they pair heavy with light: L4096 H8 B2 1019.6 -> 627.0 us.  Only where that order applies.
    if constexpr (sizeof(T) == 4 && E == 128) {
        const long long wg128 = (long long)((d.ql + 127) / 128) * d.qh * d.batch;
        if (mode == 1 && causal_alt_run(d.causal != 0, wg128, (long long)d.qh * d.batch, d.qh / d.kh, (d.ql + 127) / 128) > 0) nw = 4;
    }
    if (E >= 128 && mode == 2) nw = 4;               // the E=128 pair-bias body: 4 waves per workgroup
    if (E >= 256) nw = 4;                            // E = 256: one form (it spills at the 256-register cap as it is)
    if (const int t = tune_get(kTuneFwdNW); t == 4 || t == 8) nw = t;
#ifdef NNOP_DEV_BUILD
    // QB = 2 with builtin MFMAs (experiment, make DEV=1): hipcc shuttles the score tiles between AGPRs and VGPRs
    // (~440 v_accvgpr moves per kv tile), 1.3x slower than QB = 1.  The shipped 64-row form is fa_fwd_w64.hpp.
    qb = dev_env_int("NNOP_FWD_QB", qb);
    if constexpr (sizeof(T) == 2 && E <= 64) {
        if (qb == 2 && mode == 0) return launch_fwd_cfg<T, E, 4, 0, 2>(d, a, s);
    }
#endif
    if constexpr (E < 256) {
        if (nw == 8) return launch_fwd_mode<T, E, 8, 1>(d, a, s, mode);
    }
    return launch_fwd_mode<T, E, 4, 1>(d, a, s, mode);
}

template <typename T> static int launch_fwd_generic(const nnop_fa_desc& d, const FwdArgs& a, hipStream_t s) {
    FwdParams p;
    p.o = a.o; p.ms = a.ms; p.ls = a.ls;
    p.q = a.q; p.k = a.k; p.v = a.v; p.pair = a.pair; p.kpad = a.kpad;
    p.QL = d.ql; p.KL = d.kl; p.QH = d.qh; p.KH = d.kh; p.B = d.batch;
    p.causal = d.causal ? 1 : 0;
    p.n_qblk = 0; p.n_wg = 0;
    p.scale = (float)(1.0 / sqrt((double)d.emb));
    const long long n_rows = (long long)d.batch * d.qh * d.ql;
    const long long grid = (n_rows + 3) / 4;
    if (grid > 0x7fffffffLL) return NNOP_ERR_SHAPE;
    hipLaunchKernelGGL((fa_fwd_generic_kernel<T>), dim3((unsigned)grid), dim3(256), 0, s, p, d.emb, n_rows);
    return hipGetLastError() == hipSuccess ? NNOP_OK : NNOP_ERR_HIP;
}

template <typename T> int launch_fwd(const nnop_fa_desc& d, const FwdArgs& a, hipStream_t s) {
    // E = 256: 16-bit -- the 64-row form (two column halves), or the 32-row tiled kernel with 32-key tiles for what that form does not
    // take (pair bias, short / very long key axes).  fp32 -- the 32-row tiled kernel, 4 waves with the whole register file each (O^T
    // 128 + Q fragments 128 registers; K / V rings of 32-key tiles = 128 KiB of LDS); its BACKWARD does not fit and stays on fa_generic.hpp.
    if (d.emb == 256) return launch_fwd_e<T, 256>(d, a, s);
    if (emb_generic(d.emb)) return launch_fwd_generic<T>(d, a, s);
    switch (d.emb) {
        case 16:  return launch_fwd_e<T, 16>(d, a, s);
        case 32:  return launch_fwd_e<T, 32>(d, a, s);
        case 64:  return launch_fwd_e<T, 64>(d, a, s);
        case 128: return launch_fwd_e<T, 128>(d, a, s);
        default:  return NNOP_ERR_EMB_UNSUPPORTED;
    }
}

}  // namespace nnop
