// fa_launch.hpp -- internal (C++) launcher interface between the C ABI (nnop_capi.cpp) and the
// per-dtype kernel translation units.  Not part of the public boundary.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nnop_hip.h"
#include "tuning.hpp"

namespace nnop {

struct FwdArgs {
    void *o, *ms, *ls;
    const void *q, *k, *v, *pair;
    const uint8_t* kpad;
};

struct BwdArgs {
    void *dq, *dk, *dv, *dpair;
    const void *d_o, *o, *ms, *ls, *q, *k, *v, *pair;
    const uint8_t* kpad;
    void* workspace;
    size_t workspace_bytes;
};

// One per dtype (fa_fwd_{f32,f16,bf16}.hip).  Return an nnop_status.
template <typename T> int launch_fwd(const nnop_fa_desc& d, const FwdArgs& a, hipStream_t s);
// Which kernel form launch_fwd picks (no launch): 0 = 32-row waves, 1 = split-KV, 2 = 64-row waves, 3 = the plain-HIP kernel of
// fa_generic.hpp (embedding dims outside the tiled set), 4 = two waves per SIMD in alternating phases (fa_fwd_duo.hpp).
enum FwdForm { kFormRow32 = 0, kFormSplit = 1, kFormW64 = 2, kFormGeneric = 3, kFormDuo = 4 };
int fwd_form(const nnop_fa_desc& d, bool has_pair, bool has_mask);
// Which backward kernels launch_bwd picks (no launch): bit 0: dK/dV on fa_bwd_w64_kernel, bit 1: dQ on it (else fa_bwd.hpp's)
int bwd_forms(const nnop_fa_desc& d, bool has_pair);
// One per dtype (fa_bwd_{f32,f16,bf16}.hip).
template <typename T> int launch_bwd(const nnop_fa_desc& d, const BwdArgs& a, hipStream_t s);

// rope.hip
int launch_rope(const nnop_rope_desc& d, void* qo, void* ko, const void* q, const void* k, const void* cos,
                const void* sin, float sin_sign, hipStream_t s);

// softmax.hip (a = x | dy, b = nullptr | y)
int launch_softmax(const nnop_softmax_desc& d, void* out, const void* a, const void* b, bool bwd, hipStream_t s);

// rms_norm.hip / layer_norm.hip (ws: caller scratch of norm_ws_bytes)
int launch_rms_norm(const nnop_norm_desc& d, void* y, float* rms, const void* x, const void* w, float offset, float eps,
                    hipStream_t s);
int launch_rms_norm_bwd(const nnop_norm_desc& d, void* dx, float* dw, const void* dy, const float* rms, const void* x,
                        const void* w, float offset, void* ws, hipStream_t s);
int launch_layer_norm(const nnop_norm_desc& d, void* y, float* mu, float* sigma, const void* x, const void* w,
                      const void* b, float eps, hipStream_t s);
int launch_layer_norm_bwd(const nnop_norm_desc& d, void* dx, void* dw, void* db, const void* dy, const float* mu,
                          const float* sigma, const void* x, const void* w, void* ws, hipStream_t s);
size_t norm_ws_bytes(const nnop_norm_desc& d, bool ln);

// Embedding dims the MFMA kernels are instantiated for.
// MFMA-tiled kernels: 16, 32, 64, 128; every other power of two up to 512: fa_generic.hpp (correctness path)
inline bool emb_tiled(int e) { return e == 16 || e == 32 || e == 64 || e == 128; }
inline bool emb_supported(int e) { return e >= 1 && e <= 512 && (e & (e - 1)) == 0; }

// bytes of backward scratch: two fp32 per query row (folded log-sum-exp, -delta), [2][B][QH][QLs], QLs = QL rounded up to 64
// (the one-wave-per-SIMD kernels copy whole steps of these rows; the padding holds neutral values), and for the problems those
// kernels take (16-bit, E = 64 / 128) the same two values once more as 2 x 8 elements of T per row (operand fragments)
inline size_t bwd_rows_padded(const nnop_fa_desc& d) { return (size_t)d.batch * d.qh * (size_t)((d.ql + 63) & ~63); }
inline bool bwd_has_rcf(const nnop_fa_desc& d) { return d.dtype != NNOP_F32 && (d.emb == 64 || d.emb == 128 || d.emb == 256); }
inline size_t bwd_workspace_bytes(const nnop_fa_desc& d) {
    return bwd_rows_padded(d) * (2 * sizeof(float) + (bwd_has_rcf(d) ? 32 : 0));
}
// With a pair bias: the same + two head-major scratch matrices (a copy of the bias, dS), each
// [B][QH][pad64(KL)][pad64(QL)] elements, 256-byte aligned (pair_tile.hpp).  0 when the staged path does not apply
// (the pack kernel's LDS block holds 32 x 32 x QH elements).
inline size_t pair_scratch_elems(const nnop_fa_desc& d) {
    return (size_t)d.batch * d.qh * (size_t)((d.kl + 63) & ~63) * (size_t)((d.ql + 63) & ~63);
}
inline bool pair_staged_ok(const nnop_fa_desc& d) {
    const size_t es = d.dtype == NNOP_F32 ? 4 : 2;
    return (size_t)d.qh * 1024 * es <= 64 * 1024;
}
inline size_t bwd_workspace_bytes_pair(const nnop_fa_desc& d) {
    const size_t base = (bwd_workspace_bytes(d) + 255) & ~(size_t)255;
    // the staged path exists for the tiled MFMA kernels only (16 <= E <= 128, and 16-bit E = 256): other embedding dims run the
    // plain-HIP kernels, which never touch the scratch
    const bool tiled = emb_tiled(d.emb) || (d.emb == 256 && d.dtype != NNOP_F32);
    if (!pair_staged_ok(d) || !tiled) return bwd_workspace_bytes(d);
    const size_t es = d.dtype == NNOP_F32 ? 4 : 2;
    const size_t one = (pair_scratch_elems(d) * es + 255) & ~(size_t)255;
    return base + 2 * one;
}


// Compute units of the current device (cached per device ordinal < 64; 0 if the query fails)
inline int device_cu_count() {
    static int cache[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (dev >= 0 && dev < 64) {
        const int c = __atomic_load_n(&cache[dev], __ATOMIC_RELAXED);
        if (c > 0) return c;
    }
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (dev >= 0 && dev < 64) __atomic_store_n(&cache[dev], v, __ATOMIC_RELAXED);
    return v;
}

// Causal launches of a few rounds in the 32-row kernels, which run two workgroups per CU side by side: the dispatch order hands a CU the
// SAME block index of two columns -- two heavy blocks on one CU, two light ones on another.  Returns g > 0 when runs of g consecutive
// blocks of an XCD's dispatch order (whole columns; one "layer" of the XCD's CUs when a column is shorter) should alternate their
// direction, so that the workgroups sharing a CU pair heavy with light; 0 = plain heaviest-first order.  `cols` columns of `n_blk`
// blocks, dealt to the XCDs in chunks of `rep` columns (xcd_remap_chunked).  Knob kTuneFwdCausalAlt: 0 never, 1 whenever the layout allows.
inline int causal_alt_run(bool causal, long long n_wg, long long cols, int rep, int n_blk) {
    if (!causal || n_blk <= 0 || rep <= 0) return 0;
    const long long cus = device_cu_count() > 0 ? device_cu_count() : 256;
    const int per_xcd = (int)(cus / 8);
    if ((cols / rep) % 8 != 0) return 0;                         // the chunked remap does not apply: no whole columns per XCD
    int g = 0;
    if (n_blk == per_xcd) g = n_blk;                             // a column is one layer of the XCD's CUs: alternate whole columns
    else if (n_blk < per_xcd && per_xcd % n_blk == 0) g = per_xcd;      // several columns per layer: alternate layers
    // (a column longer than a layer pairs its own blocks j and j + per_xcd on a CU: alternating columns makes that worse -- measured
    // fp32 E128 L8192 H8 B2, 4 waves: 2441 -> 3036 us)
    if (g == 0 || (((cols / 8) * n_blk) / g) % 2 != 0) return 0;   // an even number of runs per XCD
    const int knob = tune_get(kTuneFwdCausalAlt);
    if (knob == 0) return 0;
    if (knob == 1) return g;
    return (n_wg > cus && n_wg <= (rep == 1 ? 4 : 2) * cus) ? g : 0;
}

// Do 128-row workgroups of 32-row waves beat 256-row workgroups of 64-row waves for `len` stationary rows x `cols` (batch x head)
// columns?  The 32-row forms (fa_fwd_duo.hpp NZ = 1, BwdW64Shape NARROW) do ~0.62 of a 256-row block's work time per 128-row block
// (one MFMA per fragment read), so they pay where they turn idle CUs into busy ones.  Measured on 256 CUs (profiles/r04/nz1_sweep.log,
// bwd_narrow.log): equal-work blocks -- 1.35-1.6x faster while the 128-row blocks fit one round, +7 % where they make 3 rounds of 2,
// slower from there on; causal -- the finer blocks also even out the triangle: ahead up to a round and a half of 256-row blocks.
inline bool small_grid_prefers_32_row_waves(int len, long long cols, bool causal) {
    const long long w2 = (long long)((len + 255) / 256) * cols, w1 = (long long)((len + 127) / 128) * cols;
    const long long cus = device_cu_count() > 0 ? device_cu_count() : 256;
    if (causal) return 2 * w2 <= 3 * cus;
    const long long r2 = (w2 + cus - 1) / cus, r1 = (w1 + cus - 1) / cus;
    return 62 * r1 < 100 * r2;
}

// Kernels that need more than 64 KiB of dynamic LDS must opt in once per (kernel, device).  `done` is a per-kernel
// static bitmap (one bit per device ordinal < 64); setting the attribute twice is harmless, so a benign race between
// host threads only costs a redundant call.
template <typename K> inline int ensure_dynamic_lds(K kern, int lds_bytes, unsigned long long* done) {
    if (lds_bytes <= 64 * 1024) return NNOP_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return NNOP_ERR_HIP; }
    const unsigned long long bit = 1ull << (dev & 63);
    if (dev < 64 && (__atomic_load_n(done, __ATOMIC_RELAXED) & bit)) return NNOP_OK;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) {
        (void)hipGetLastError();
        return NNOP_ERR_HIP;
    }
    if (dev < 64) __atomic_fetch_or(done, bit, __ATOMIC_RELAXED);
    return NNOP_OK;
}

}  // namespace nnop
