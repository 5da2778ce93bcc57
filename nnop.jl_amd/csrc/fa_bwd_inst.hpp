// fa_bwd_inst.hpp -- host-side dispatch for the backward kernels; included by the three per-dtype
// translation units.  Replaces the reference's `∇flash_attention` host body
// (src/attention_bwd.jl:215-272): three launches on the caller's stream (preprocess, dK/dV, dQ),
// scratch from the caller's workspace, dpair zero-filled only when a pair bias is given.
#pragma once
#include "fa_bwd.hpp"
#include "fa_bwd_w64.hpp"
#include "fa_launch.hpp"
#include "fa_generic.hpp"
#include <math.h>

namespace nnop {

template <typename T, int E> struct BwdCfg {
    static constexpr bool kF32 = sizeof(T) == 4;
    // fp32 at E=128 keeps K,V (dkdv) / Q,dO (dq) in LDS instead of registers: fewer waves, smaller tiles
    // (16-bit E = 256: 2 waves, single-buffered tiles -- what fits 160 KiB of LDS)
    // (fp32 E = 256: 4 waves with the stationary fragments in registers, fa_bwd.hpp NNOP_F32_E256_FORM)
    static constexpr int NW_KV = (kF32 && E > 128) ? (NNOP_F32_E256_FORM == 2 ? 4 : NNOP_F32_E256_FORM == 1 ? 2 : 1) : ((kF32 && E > 64 && !fa_bwd_f32_wide<T, E>()) || E > 128) ? 2 : 4;
    static constexpr int BQ    = (kF32 || E > 64) ? 32 : 64;
    static constexpr int NW_Q  = (kF32 && E > 128) ? (NNOP_F32_E256_FORM == 2 ? 4 : 1) : ((kF32 && E > 64 && !fa_bwd_f32_wide<T, E>()) || E > 128) ? 2 : 4;
    // 16-bit E = 128, large grids: 7 waves (224 keys / queries per workgroup) with single-buffered tiles ->
    // ~2 waves per SIMD instead of 1 (LDS-limited); small grids keep 4 waves (finer quantization over 256 CUs)
    static constexpr bool kBig7 = !kF32 && E == 128;
    // 16-bit E <= 64 (K, V / Q, dO fragments live in registers): 8 waves per workgroup share each staged tile -- half the
    // staging work and LDS traffic per wave at the same 2 waves per SIMD; used when the grid still fills the chip
    static constexpr bool kWide8 = !kF32 && E <= 64;
    static constexpr int BK    = (E > 64) ? 32 : 64;
};

// The forward's rule for causal launches of a few rounds (fa_launch.hpp causal_alt_run), only where two workgroups share a CU (E <= 64:
// the fp32 E = 128 kernels fill a CU alone, and there the alternate order is just a worse list -- measured fp32 E128 L4096 H8 B2 causal
// backward 2606 -> 3763 us).
static inline int bwd_causal_alt(const nnop_fa_desc& d, long long n_wg, long long cols, int rep, int n_blk, bool pair) {
    if (d.emb > 64 || pair) return 0;
    return causal_alt_run(d.causal != 0, n_wg, cols, rep, n_blk);
}

template <typename T, int E, int NW, int BQ, int MODE>
static int launch_dkdv(const nnop_fa_desc& d, const BwdParams& p, hipStream_t s) {
    constexpr int lds = fa_bwd_dkdv_lds_bytes<T, E, NW, BQ, MODE>() + (MODE == 3 ? NW * PairTile<T>::kBytes : 0);
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = fa_bwd_dkdv_kernel<T, E, NW, BQ, MODE>;
    static unsigned long long lds_done = 0;
    if (ensure_dynamic_lds(kern, lds, &lds_done) != NNOP_OK) return NNOP_ERR_HIP;
    BwdParams pk = p;
    pk.n_blk = (d.kl + 32 * NW - 1) / (32 * NW);
    const long long n_wg = (long long)pk.n_blk * d.kh * d.batch;
    if (n_wg > 0x7fffffffLL) return NNOP_ERR_SHAPE;
    pk.n_wg = (int)n_wg;
    if (n_wg * fa_bwd_split<T, E>() > 0x7fffffffLL) return NNOP_ERR_SHAPE;
    pk.causal_alt = bwd_causal_alt(d, n_wg, (long long)d.kh * d.batch, 1, pk.n_blk, MODE >= 2);
    hipLaunchKernelGGL(kern, dim3((unsigned)(n_wg * fa_bwd_split<T, E>())), dim3(NW * 64), lds, s, pk);   // (fp32 E = 256: every block once per column slice)
    return NNOP_OK;
}

template <typename T, int E, int NW, int BK, int MODE>
static int launch_dq(const nnop_fa_desc& d, const BwdParams& p, hipStream_t s) {
    constexpr int lds = fa_bwd_dq_lds_bytes<T, E, NW, BK, MODE>() + (MODE == 3 ? NW * PairTile<T>::kBytes : 0);
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = fa_bwd_dq_kernel<T, E, NW, BK, MODE>;
    static unsigned long long lds_done = 0;
    if (ensure_dynamic_lds(kern, lds, &lds_done) != NNOP_OK) return NNOP_ERR_HIP;
    BwdParams pq = p;
    pq.n_blk = (d.ql + 32 * NW - 1) / (32 * NW);
    const long long n_wg = (long long)pq.n_blk * d.qh * d.batch;
    if (n_wg > 0x7fffffffLL) return NNOP_ERR_SHAPE;
    pq.n_wg = (int)n_wg;
    if (n_wg * fa_bwd_split<T, E>() > 0x7fffffffLL) return NNOP_ERR_SHAPE;
    pq.causal_alt = bwd_causal_alt(d, n_wg, (long long)d.qh * d.batch, d.qh / d.kh, pq.n_blk, MODE >= 2);
    hipLaunchKernelGGL(kern, dim3((unsigned)(n_wg * fa_bwd_split<T, E>())), dim3(NW * 64), lds, s, pq);
    return NNOP_OK;
}

// The one-wave-per-SIMD form (fa_bwd_w64.hpp): KIND = kBwdDKDV / kBwdDQ
template <typename T, int E, int KIND, int MODE, int NARROW = 0>
static int launch_bwd_w64(const nnop_fa_desc& d, const BwdParams& p, hipStream_t s) {
    using SH = BwdW64Shape<E, KIND, NARROW>;
    constexpr int lds = fa_bwd_w64_lds_bytes<T, E, KIND, NARROW>(MODE != 0);
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = fa_bwd_w64_kernel<T, E, KIND, MODE, NARROW>;
    static unsigned long long lds_done = 0;
    if (ensure_dynamic_lds(kern, lds, &lds_done) != NNOP_OK) return NNOP_ERR_HIP;
    BwdParams pk = p;
    const int len = KIND == kBwdDQ ? d.ql : d.kl, hd = KIND == kBwdDQ ? d.qh : d.kh;
    pk.n_blk = (len + SH::WG_ROWS - 1) / SH::WG_ROWS;
    const long long n_wg = (long long)pk.n_blk * hd * d.batch;
    if (n_wg > 0x7fffffffLL) return NNOP_ERR_SHAPE;
    pk.n_wg = (int)n_wg;
    if (n_wg * SH::NSPLIT > 0x7fffffffLL) return NNOP_ERR_SHAPE;
    // Persistent form (as the forward's, fa_fwd_inst.hpp launch_fwd_w64): 256 workgroups over a static balanced block list, under a
    // causal mask without key padding, when the list divides.  Knob kTuneBwdPersist (0 never, 1 wherever it divides).
    long long grid = n_wg * SH::NSPLIT;
    pk.persist = 0;
    if constexpr (MODE == 1 && SH::NSPLIT == 1) {
        const long long cols = (long long)d.batch * hd;
        const int n = pk.n_blk;
        const long long per_xcd = (cols / 8) * n;
        const int knob = tune_get(kTuneBwdPersist);
        const int hx = hd % 8 == 0 ? hd / 8 : 0;
        // (key padding without a causal mask: the dQ pass only -- every query block of a batch costs the same there, while the key
        // blocks of the dK/dV pass are full or EMPTY by the batch's length and a static list cannot pair them up: measured -17 % at C4)
        const bool pays = (d.causal && !p.kpad) || (KIND == kBwdDQ && !d.causal && p.kpad && hx > 0);
        if ((knob == 1 || (knob < 0 && pays)) && device_cu_count() == 256 && cols % 8 == 0 && (n & (n - 1)) == 0 &&
            per_xcd % 32 == 0 && per_xcd / 32 >= 2 && per_xcd / 32 <= (1 << 24)) {
            pk.persist = (int)(per_xcd / 32);
            pk.persist_hx = hx;
            grid = 256;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, pk);      // E = 256 dK/dV: every block twice (column halves)
    return NNOP_OK;
}
// Is the one-wave-per-SIMD form instantiated for this problem (16-bit, E = 64 / 128, no pair bias), and do its 32-bit descriptor
// ranges hold it?  A plain function of the descriptor: also behind nnop_debug_bwd_form (bench.py names the kernels it times).
static inline bool bwd_w64_ok(const nnop_fa_desc& d, int kind) {
    if (d.dtype == NNOP_F32 || (d.emb != 64 && d.emb != 128 && d.emb != 256)) return false;
    const long long rb = 2LL * d.emb;
    if (kind == kBwdDKDV) {
        // one descriptor spans the q-heads of a kv head; the row-constant fragments of the whole launch behind another
        if ((long long)(d.qh / d.kh) * d.ql * rb >= (1LL << 32)) return false;
        if ((long long)bwd_rows_padded(d) * 32 >= (1LL << 32)) return false;
        return true;
    }
    return (long long)d.kl * rb < (1LL << 32) && d.kl <= 64 * kMaxMaskTiles;
}
// The narrow shape of that form (BwdW64Shape NARROW: 32 stationary rows per wave, 128-row workgroups) where the 256-row blocks of a pass
// leave CUs idle -- the forward's rule (small_grid_prefers_32_row_waves, fa_launch.hpp), per pass: the dK/dV pass counts kv heads.
// Knob kTuneBwdNarrow: 0 never, 1 wherever instantiated.
static inline bool bwd_w64_narrow(const nnop_fa_desc& d, int kind) {
    if (d.emb != 64 && d.emb != 128) return false;
    const int knob = tune_get(kTuneBwdNarrow);
    if (knob >= 0) return knob != 0;
    const int len = kind == kBwdDQ ? d.ql : d.kl, hd = kind == kBwdDQ ? d.qh : d.kh;
    return small_grid_prefers_32_row_waves(len, (long long)hd * d.batch, d.causal != 0);
}
// bit 0: dK/dV runs fa_bwd_w64_kernel, bit 1: dQ does (knob kTuneBwdW64: 0 never, 1 both, 2 dK/dV only, 3 dQ only, 4 both with the
// preprocess launch kept, auto = both)
static inline int bwd_w64_forms(const nnop_fa_desc& d, bool has_pair, bool ws_aligned16) {
    if (has_pair) return 0;
    const int t = tune_get(kTuneBwdW64);
    const bool want_kv = t < 0 || t == 1 || t == 2 || t == 4, want_q = t < 0 || t == 1 || t == 3 || t == 4;
    return ((want_kv && ws_aligned16 && bwd_w64_ok(d, kBwdDKDV)) ? 1 : 0) | ((want_q && bwd_w64_ok(d, kBwdDQ)) ? 2 : 0);
}

template <typename T, int E, int MODE>
static int launch_bwd_cfg(const nnop_fa_desc& d, const BwdArgs& a, hipStream_t s) {
    using C = BwdCfg<T, E>;
    BwdParams p;
    p.dq = a.dq; p.dk = a.dk; p.dv = a.dv; p.dpair = a.pair ? a.dpair : nullptr;
    p.d_o = a.d_o; p.o = a.o; p.ms = a.ms; p.ls = a.ls;
    p.q = a.q; p.k = a.k; p.v = a.v; p.pair = a.pair; p.kpad = a.kpad;
    p.QLs = (d.ql + 63) & ~63;
    const long long n_rows = (long long)d.batch * d.qh * p.QLs;            // padded rows (bwd_workspace_bytes)
    if (n_rows > 0x7fffffffLL) return NNOP_ERR_SHAPE;
    p.nl = (float*)a.workspace;
    p.delta = p.nl + n_rows;
    // the fragment form of the row constants (fa_bwd_w64.hpp, dK/dV): behind the two vectors (n_rows is a multiple of 64: the offset
    // keeps the workspace's alignment); 16-byte stores / LDS-DMA -> only with a 16-byte aligned workspace
    p.rcf = (bwd_has_rcf(d) && ((uintptr_t)a.workspace & 15) == 0) ? (void*)(p.delta + n_rows) : nullptr;
    p.fused = 0;
    p.QL = d.ql; p.KL = d.kl; p.QH = d.qh; p.KH = d.kh; p.B = d.batch;
    p.causal = d.causal ? 1 : 0;
    p.scale = (float)(1.0 / sqrt((double)E));
    p.n_blk = 0; p.n_wg = 0;
    p.pair_a = nullptr; p.dpair_s = nullptr;
    p.QLp = pair_pad(d.ql); p.KLp = pair_pad(d.kl);
    if constexpr (MODE == 3) {
        // scratch behind the two row vectors: the head-major copy of the bias, dS (bwd_workspace_bytes_pair)
        const size_t base = (bwd_workspace_bytes(d) + 255) & ~(size_t)255;
        const size_t one = (pair_scratch_elems(d) * sizeof(T) + 255) & ~(size_t)255;
        char* w = (char*)a.workspace + base;
        p.pair_a = w; p.dpair_s = w + one;
        const long long nblk = (long long)d.batch * (p.KLp / 32) * (p.QLp / 32);
        if (nblk > 0x7fffffffLL) return NNOP_ERR_SHAPE;
        PairPackParams pp{a.pair, (void*)p.pair_a, d.ql, d.kl, d.qh, d.batch, p.QLp, p.KLp, d.causal ? 1 : 0};
        hipLaunchKernelGGL((pair_pack_kernel<T>), dim3((unsigned)nblk), dim3(256), 1024 * d.qh * sizeof(T), s, pp);
    }

    // measurement only (kTuneBwdStages, bench.py's per-kernel times): run a subset of the passes -- 1 preprocess, 2 dK/dV, 4 dQ
    const int stages = tune_get(kTuneBwdStages) < 0 ? 7 : tune_get(kTuneBwdStages);
    // the one-wave-per-SIMD form (fa_bwd_w64.hpp): 16-bit, E = 64 / 128, plain / masked modes
    const int w64_forms = MODE <= 1 ? bwd_w64_forms(d, false, p.rcf != nullptr) : 0;
    const bool w64_kv = (w64_forms & 1) != 0, w64_q = (w64_forms & 2) != 0;
    // Both passes in that form: no preprocess launch -- the dQ kernel computes the row constants of its own rows and leaves them
    // (fragment form) for the dK/dV kernel, which therefore runs BEHIND it.  (kTuneBwdW64 = 4: both passes, preprocess kept: A/B)
    p.fused = (w64_kv && w64_q && tune_get(kTuneBwdW64) != 4) ? 1 : 0;
    // 1. preprocess
    if ((stages & 1) && !p.fused) {
        const long long n_thr = n_rows * (E / 8);
        const long long grid = (n_thr + 255) / 256;
        if (grid > 0x7fffffffLL) return NNOP_ERR_SHAPE;
        hipLaunchKernelGGL((fa_bwd_pre_kernel<T, E>), dim3((unsigned)grid), dim3(256), 0, s, p, n_rows);
    }
    // 2. dpair is written only where a (query, key) pair is visited: zero it first (MODE 3: the unpack kernel writes all of it)
    if (p.dpair && MODE != 3) {
        const size_t bytes = (size_t)d.batch * d.kl * d.ql * d.qh * sizeof(T);
        if (hipMemsetAsync(p.dpair, 0, bytes, s) != hipSuccess) { (void)hipGetLastError(); return NNOP_ERR_HIP; }
    }
    const int big_tune = tune_get(kTuneBwdBig7);
    // workgroups of the 7 / 8-wave E = 128 forms from which they are used: one per CU when every block has the same work,
    // two per CU under a causal mask (measured, tools/bwd_ab.py: 256 blocks non-causal +27 %, causal -9 %)
    const int big_thr = big_tune >= 0 ? big_tune : (d.causal ? 512 : 256);
    // 3. dK, dV
    auto run_dkdv = [&]() -> int {
        int st = NNOP_OK;
        bool done = false;
        if constexpr (MODE <= 1 && sizeof(T) == 2 && (E == 64 || E == 128 || E == 256)) {
            if (w64_kv) {
                if constexpr (E != 256) {
                    if (bwd_w64_narrow(d, kBwdDKDV)) { st = launch_bwd_w64<T, E, kBwdDKDV, MODE, 1>(d, p, s); done = true; }
                }
                if (!done) { st = launch_bwd_w64<T, E, kBwdDKDV, MODE>(d, p, s); done = true; }
            }
        }
        if constexpr (C::kBig7) if (!done) {
            // 8 waves double-buffered where K lives in registers (64 KiB of V images + 2 x 32 KiB of tiles), else 7 single-buffered
            constexpr int NWB = fa_bwd_dkdv_kregs<T, E, MODE>() ? 8 : 7;
            const long long nb = (long long)((d.kl + 32 * NWB - 1) / (32 * NWB)) * d.kh * d.batch;
            if (nb >= big_thr) { st = launch_dkdv<T, E, NWB, C::BQ, MODE>(d, p, s); done = true; }
        }
        if constexpr (C::kWide8) if (!done) {
            const long long n8 = (long long)((d.kl + 255) / 256) * d.kh * d.batch;
            const int nw = tune_get(kTuneBwdNW);
            if (MODE != 2 && (nw == 8 || nw == 81 || (nw < 0 && !d.causal && n8 >= 256))) { st = launch_dkdv<T, E, 8, C::BQ, MODE>(d, p, s); done = true; }
        }
        if (!done) st = launch_dkdv<T, E, C::NW_KV, C::BQ, MODE>(d, p, s);
        return st;
    };
    // 4. dQ
    auto run_dq = [&]() -> int {
        int st = NNOP_OK;
        bool done = false;
        if constexpr (MODE <= 1 && sizeof(T) == 2 && (E == 64 || E == 128 || E == 256)) {
            // plain mode needs whole steps of keys; a ragged KL takes the masked kernel
            if (w64_q) {
                if constexpr (E != 256) {
                    if (bwd_w64_narrow(d, kBwdDQ)) {                                  // (64 streamed keys per step there)
                        if (MODE == 0 && (d.kl & 63) != 0) st = launch_bwd_w64<T, E, kBwdDQ, 1, 1>(d, p, s);
                        else st = launch_bwd_w64<T, E, kBwdDQ, MODE, 1>(d, p, s);
                        done = true;
                    }
                }
                if (!done) {
                    if (MODE == 0 && (d.kl & 31) != 0) st = launch_bwd_w64<T, E, kBwdDQ, 1>(d, p, s);
                    else st = launch_bwd_w64<T, E, kBwdDQ, MODE>(d, p, s);
                    done = true;
                }
            }
        }
        if constexpr (C::kBig7) if (!done) {
            // 8 waves where Q, dO live in registers (no LDS images of them: double-buffered tiles fit), else 7 single-buffered
            constexpr int NWB = fa_bwd_dq_qregs<T, E, MODE>() ? 8 : 7;
            const long long nb = (long long)((d.ql + 32 * NWB - 1) / (32 * NWB)) * d.qh * d.batch;
            if (nb >= big_thr) { st = launch_dq<T, E, NWB, C::BK, MODE>(d, p, s); done = true; }
        }
        if constexpr (C::kWide8) if (!done) {
            const long long n8 = (long long)((d.ql + 255) / 256) * d.qh * d.batch;
            const int nw = tune_get(kTuneBwdNW);
            if (MODE != 2 && (nw == 8 || nw == 82 || (nw < 0 && !d.causal && n8 >= 256))) { st = launch_dq<T, E, 8, C::BK, MODE>(d, p, s); done = true; }
        }
        if (!done) st = launch_dq<T, E, C::NW_Q, C::BK, MODE>(d, p, s);
        return st;
    };
    if (p.fused) {
        if (stages & 4) { const int st = run_dq(); if (st != NNOP_OK) return st; }
        if (stages & 2) { const int st = run_dkdv(); if (st != NNOP_OK) return st; }
    } else {
        if (stages & 2) { const int st = run_dkdv(); if (st != NNOP_OK) return st; }
        if (stages & 4) { const int st = run_dq(); if (st != NNOP_OK) return st; }
    }
    if constexpr (MODE == 3) {
        const long long nblk = (long long)d.batch * (p.KLp / 32) * (p.QLp / 32);
        PairUnpackParams up{a.dpair, p.dpair_s, a.kpad, d.ql, d.kl, d.qh, d.batch, p.QLp, p.KLp, d.causal ? 1 : 0,
                            PairTile<T>::kStoreLaneMajor ? 1 : 0};
        hipLaunchKernelGGL((dpair_unpack_kernel<T>), dim3((unsigned)nblk), dim3(256), 1024 * d.qh * sizeof(T), s, up);
    }
    return hipGetLastError() == hipSuccess ? NNOP_OK : NNOP_ERR_HIP;
}

template <typename T, int E>
static int launch_bwd_e(const nnop_fa_desc& d, const BwdArgs& a, hipStream_t s) {
    if constexpr (sizeof(T) == 4 && E == 256) {
        if (a.pair) return NNOP_ERR_EMB_UNSUPPORTED;             // (launch_bwd sends fp32 E = 256 with a pair bias to the plain-HIP kernels)
        return (d.causal || a.kpad) ? launch_bwd_cfg<T, E, 1>(d, a, s) : launch_bwd_cfg<T, E, 0>(d, a, s);
    } else
    if (a.pair) {
        // staged pair path when the caller brought the scratch for it (nnop_fa_bwd_workspace_bytes_pair)
        // (the scratch matrices are addressed with 16-byte vectors: a workspace that is not 16-byte aligned takes the direct path)
        if (pair_staged_ok(d) && a.workspace_bytes >= bwd_workspace_bytes_pair(d) && bwd_workspace_bytes_pair(d) > bwd_workspace_bytes(d) &&
            ((uintptr_t)a.workspace & 15) == 0)
            return launch_bwd_cfg<T, E, 3>(d, a, s);
        return launch_bwd_cfg<T, E, 2>(d, a, s);
    }
    if (d.causal || a.kpad) return launch_bwd_cfg<T, E, 1>(d, a, s);
    return launch_bwd_cfg<T, E, 0>(d, a, s);
}

template <typename T> static int launch_bwd_generic(const nnop_fa_desc& d, const BwdArgs& a, hipStream_t s) {
    BwdParams p;
    p.dq = a.dq; p.dk = a.dk; p.dv = a.dv; p.dpair = a.pair ? a.dpair : nullptr;
    p.d_o = a.d_o; p.o = a.o; p.ms = a.ms; p.ls = a.ls;
    p.q = a.q; p.k = a.k; p.v = a.v; p.pair = a.pair; p.kpad = a.kpad;
    const long long n_rows = (long long)d.batch * d.qh * d.ql, n_krows = (long long)d.batch * d.kh * d.kl;
    p.QLs = d.ql;                                                          // dense rows here
    p.rcf = nullptr; p.fused = 0;
    p.nl = (float*)a.workspace;
    p.delta = p.nl + n_rows;
    p.QL = d.ql; p.KL = d.kl; p.QH = d.qh; p.KH = d.kh; p.B = d.batch;
    p.causal = d.causal ? 1 : 0;
    p.scale = (float)(1.0 / sqrt((double)d.emb));
    p.n_blk = 0; p.n_wg = 0;
    p.pair_a = nullptr; p.dpair_s = nullptr; p.QLp = p.KLp = 0;
    const long long gq = (n_rows + 3) / 4, gk = (n_krows + 3) / 4;
    if (gq > 0x7fffffffLL || gk > 0x7fffffffLL) return NNOP_ERR_SHAPE;
    if (p.dpair) {      // the dQ kernel writes dS only where a query sees the key
        const size_t bytes = (size_t)d.batch * d.kl * d.ql * d.qh * sizeof(T);
        if (hipMemsetAsync(p.dpair, 0, bytes, s) != hipSuccess) { (void)hipGetLastError(); return NNOP_ERR_HIP; }
    }
    hipLaunchKernelGGL((fa_bwd_generic_pre_kernel<T>), dim3((unsigned)gq), dim3(256), 0, s, p, d.emb, n_rows);
    hipLaunchKernelGGL((fa_bwd_generic_dq_kernel<T>), dim3((unsigned)gq), dim3(256), 0, s, p, d.emb, n_rows);
    hipLaunchKernelGGL((fa_bwd_generic_dkdv_kernel<T>), dim3((unsigned)gk), dim3(256), 0, s, p, d.emb, n_krows);
    return hipGetLastError() == hipSuccess ? NNOP_OK : NNOP_ERR_HIP;
}

template <typename T> int launch_bwd(const nnop_fa_desc& d, const BwdArgs& a, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        if (d.emb == 256) return launch_bwd_e<T, 256>(d, a, s);     // tiled kernels, 2 waves, single-buffered (see launch_fwd)
    } else {
        if (d.emb == 256 && !a.pair) return launch_bwd_e<T, 256>(d, a, s);     // fp32: tiled kernels, 1 wave (no pair-bias mode: plain HIP)
    }
    if (emb_generic(d.emb)) return launch_bwd_generic<T>(d, a, s);
    switch (d.emb) {
        case 16:  return launch_bwd_e<T, 16>(d, a, s);
        case 32:  return launch_bwd_e<T, 32>(d, a, s);
        case 64:  return launch_bwd_e<T, 64>(d, a, s);
        case 128: return launch_bwd_e<T, 128>(d, a, s);
        default:  return NNOP_ERR_EMB_UNSUPPORTED;
    }
}

}  // namespace nnop
