/*
 * nnop_hip.h -- C ABI of libnnop_hip.so: MI355X (gfx950) Flash Attention behind NNop.jl's
 * operator boundary.
 *
 * These are exactly the entry points a Julia package extension of the shape of
 * ext/NNopAMDGPUExt.jl:1-11 `ccall`s to replace, for ROCArray arguments, the two generic
 * host functions of the reference:
 *
 *   NNop._flash_attention(q,k,v,pair; causal,kpad_mask) -> (o, ms, ls)   src/attention.jl:133-177
 *   NNop.∇flash_attention(Δ,o,ms,ls,q,k,v,pair; causal,kpad_mask)
 *                                      -> (dq, dk, dv, dpair|nothing)   src/attention_bwd.jl:199-275
 *   NNop._shared_memory(backend, device_id) -> UInt64                    ext/NNopAMDGPUExt.jl:6-9
 *
 * (binding shown in INTEGRATION.md; shipped, source-only, in nnop.jl_amd/julia/).
 *
 * Conventions
 *   - Plain pointers and sizes only; no C++ / torch types cross this boundary.
 *   - Every pointer is a DEVICE pointer on the current HIP device; the caller owns every
 *     buffer, outputs and workspace included (replaces the reference's internal
 *     `similar` / `KA.zeros`, src/attention.jl:166-168, src/attention_bwd.jl:224-238).
 *   - All tensors dense, contiguous, E fastest -- Julia column-major (E,L,H,B) is the same
 *     memory as C row-major [B][H][L][E]:
 *        q,o,dO,dq : [B][QH][QL][E]      k,v,dk,dv : [B][KH][KL][E]
 *        ms,ls     : [B][QH][QL]         (dtype T, src/attention.jl:167-168)
 *        pair,dpair: [B][KL][QL][QH]     (Julia (QH,QL,KL,B), src/attention.jl:62)
 *        kpad_mask : [B][KL], 1 byte per element, nonzero = key is valid (Julia Bool matrix
 *                    (KL,B), src/attention.jl:76)
 *   - Launches are asynchronous on the passed stream; nothing synchronises, allocates or
 *     frees; no pointer is retained after return (src/attention.jl:170-176 never syncs).
 *   - Re-entrant; no mutable global state.  Safe from concurrent host threads on different
 *     streams / devices.  The caller selects the device (hipSetDevice) before calling.
 *   - Errors are returned, never thrown: 0 = success, negative = nnop_status.  The host shim
 *     re-raises with the reference's messages (src/attention.jl:141-144).
 */
#ifndef NNOP_HIP_H
#define NNOP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element type T of q,k,v,o,ms,ls,pair and all gradients (they share one T,
 * src/attention.jl:134-137) */
typedef enum nnop_dtype {
    NNOP_F32  = 0,
    NNOP_F16  = 1,
    NNOP_BF16 = 2
} nnop_dtype;

typedef enum nnop_status {
    NNOP_OK                 =  0,
    NNOP_ERR_EMB_MISMATCH   = -1,  /* "Embedding dim of Q `..` must be the same as of K" (attention.jl:141) */
    NNOP_ERR_KV_SHAPE       = -2,  /* "Shapes of K and V must be the same"               (attention.jl:142) */
    NNOP_ERR_EMB_NOT_POW2   = -3,  /* "Only power-of-2 embedding dims are supported."     (attention.jl:143) */
    NNOP_ERR_HEADS          = -4,  /* "Number of query heads must be divisible by ..."    (attention.jl:144) */
    NNOP_ERR_DTYPE          = -5,  /* dtype not one of nnop_dtype (Julia: MethodError)                       */
    NNOP_ERR_NULL           = -6,  /* a required pointer is NULL                                             */
    NNOP_ERR_EMB_UNSUPPORTED= -7,  /* power of two above 512 ("Failed to find groupsize
                                      ... Shared Memory constraint", attention.jl:204)                       */
    NNOP_ERR_SHAPE          = -8,  /* a non-positive / overflowing dimension                                 */
    NNOP_ERR_WORKSPACE      = -9,  /* workspace smaller than nnop_fa_bwd_workspace_bytes                      */
    NNOP_ERR_HIP            = -10, /* HIP runtime error at launch (hipGetLastError)                          */
    NNOP_ERR_ALIGN          = -11  /* a tensor or workspace base address is not aligned as the kernels need:
                                      16 bytes for q, k, v, o, the gradients, pair / dpair and the workspace
                                      (the MFMA kernels move them with 16-byte vector accesses and LDS-DMA),
                                      the element size for ms, ls.  Embedding dims that run the plain-HIP
                                      kernels (not 16 / 32 / 64 / 128 / 256) need element alignment only.      */
} nnop_status;

/*
 * Problem descriptor.  The reference reads all of this from the array sizes
 * (src/attention.jl:138-139); across a C boundary the caller states it.
 * `emb_k`, `kl_v`, `kh_v`, `emb_v` carry the K / V sizes that the reference's checks
 * compare (E of K vs Q; size(k) == size(v)), so that those checks -- and their error
 * codes -- live behind the boundary like they do in the reference.
 */
typedef struct nnop_fa_desc {
    int32_t dtype;    /* nnop_dtype */
    int32_t emb;      /* E  = size(q,1) */
    int32_t ql;       /* QL = size(q,2) */
    int32_t kl;       /* KL = size(k,2) */
    int32_t qh;       /* QH = size(q,3) */
    int32_t kh;       /* KH = size(k,3) */
    int32_t batch;    /* B  = size(q,4) */
    int32_t causal;   /* keyword `causal` (required in the reference, attention_crc.jl:7) */
    int32_t emb_k;    /* size(k,1); 0 means "same as emb" */
    int32_t emb_v;    /* size(v,1); 0 means "same as emb" */
    int32_t kl_v;     /* size(v,2); 0 means "same as kl"  */
    int32_t kh_v;     /* size(v,3); 0 means "same as kh"  */
} nnop_fa_desc;

/* Opaque to the ABI: a hipStream_t.  Declared void* so that C callers need no HIP headers. */
typedef void* nnop_stream_t;

/*
 * Forward: contract of NNop._flash_attention (src/attention.jl:133-177) + kernel
 * _flash_attention_fwd! (src/attention.jl:1-131).
 *   o  = softmax(scale * q k^T (+pair) masked) v
 *   ms = row max of the scaled+biased+masked logits   (src/attention.jl:128)
 *   ls = sum_j exp(logit_j - ms)                       (src/attention.jl:129)
 * `pair` and `kpad_mask` may be NULL (Julia `nothing`).
 */
int nnop_fa_fwd(const nnop_fa_desc* d,
                void* o, void* ms, void* ls,
                const void* q, const void* k, const void* v,
                const void* pair, const uint8_t* kpad_mask,
                nnop_stream_t stream);

/*
 * Scratch the backward needs (replaces the reference's internal Δ_scaled / δ temporaries,
 * src/attention_bwd.jl:224-225).  Returns 0 for an invalid descriptor.
 */
size_t nnop_fa_bwd_workspace_bytes(const nnop_fa_desc* d);

/*
 * The same for a call WITH a pair bias (ABI version 5).  `pair` / `dpair` are [B][KL][QL][QH] with the head fastest
 * (src/attention.jl:62): per head their elements are QH apart, which a (batch, head) kernel can only touch one element per
 * lane.  Given this much scratch -- the small workspace + two head-major bias-sized matrices (a copy of the bias, dS), padded to
 * multiples of 64 -- nnop_fa_bwd re-packs the bias once, runs its kernels on 16-byte accesses and unpacks dpair at the end
 * (bf16 E=64 L=2048 H=4 B=4: see DESIGN.md).  A caller that passes only nnop_fa_bwd_workspace_bytes() gets the same results
 * from the direct (slow) path; nnop_fa_bwd picks by `workspace_bytes`.  Returns 0 for an invalid descriptor.
 */
size_t nnop_fa_bwd_workspace_bytes_pair(const nnop_fa_desc* d);

/*
 * Backward: contract of NNop.∇flash_attention (src/attention_bwd.jl:199-275) + kernels
 * _flash_attention_bwd_preprocess! (:163-197) and _flash_attention_bwd! (:1-161).
 * Writes dq, dk, dv (fully overwritten -- the caller need not zero them) and, when `pair`
 * is non-NULL, dpair (which must then be non-NULL too).  `d_o` is the cotangent Δ.
 */
int nnop_fa_bwd(const nnop_fa_desc* d,
                void* dq, void* dk, void* dv, void* dpair,
                const void* d_o, const void* o, const void* ms, const void* ls,
                const void* q, const void* k, const void* v,
                const void* pair, const uint8_t* kpad_mask,
                void* workspace, size_t workspace_bytes,
                nnop_stream_t stream);

/*
 * Llama rotary embedding: contract of NNop._llama_rope(q, k, cos, sin; bwd)
 * (src/rope/llama_rope.jl:69-89) + kernel llama_rope! (:24-65).  For every row x of q and of k:
 *     out[i]       = x[i] * cos[i] - x[i + D/2] * (sin_sign * sin[i])
 *     out[i + D/2] = x[i + D/2] * cos[i] + x[i] * (sin_sign * sin[i])          i < D/2
 * with cos, sin indexed by (position, batch) and shared by all heads.  sin_sign = +1: llama_rope;
 * sin_sign = -1: the pullback ∇llama_rope applied to (dq, dk) (:92).  Out of place (the reference copies q, k and
 * rotates the copies, :75-76); q_out == q / k_out == k (in place) is allowed.
 *   q, q_out : [B][QH][L][D]    k, k_out : [B][KH][L][D]    cos, sin : [B][L][D] (only the first D/2 of a row are read)
 *   dtype: element type of q, k;  cs_dtype: element type of cos, sin -- NNOP_F32 (what LlamaRotaryEmbedding
 *   returns, :15-22) or the same as dtype.  D must be even.  The arithmetic is fp32, one rounding to T on store.
 */
typedef struct nnop_rope_desc {
    int32_t dtype;     /* nnop_dtype of q, k */
    int32_t cs_dtype;  /* nnop_dtype of cos, sin */
    int32_t dim;       /* D  = size(q,1) */
    int32_t seq;       /* L  = size(q,2) == size(k,2) */
    int32_t qh;        /* size(q,3) */
    int32_t kh;        /* size(k,3) */
    int32_t batch;     /* size(q,4) == size(k,4) */
} nnop_rope_desc;

int nnop_llama_rope(const nnop_rope_desc* d, void* q_out, void* k_out, const void* q, const void* k,
                    const void* cos, const void* sin, float sin_sign, nnop_stream_t stream);

/*
 * Online softmax: NNop.online_softmax(x) (src/softmax.jl:60-68, kernel
 * online_softmax! :19-58, reduction monoid MD / md_reduce :1-16 through @groupreduce, src/groupreduce.jl:13-43) and
 * its pullback ∇online_softmax(Δ, y) (src/softmax.jl:70-80).
 *   x, y, dy, dx : [batch][N]  == Julia matrix (N, batch); softmax along N (dims = 1 in Julia).
 *   y[b][:]  = exp(x[b][:] - max) / sum(exp(x[b][:] - max))
 *   dx[b][:] = y[b][:] * (dy[b][:] - sum(dy[b][:] * y[b][:]))
 * One element type T for all arrays (the reference: y = similar(x)); fp32 arithmetic, one rounding on store.
 * Any N >= 1; rows whose byte length is a multiple of 16 (and up to 16384 fp32 / 32768 16-bit elements) are
 * held in registers and touch HBM once.
 */
typedef struct nnop_softmax_desc {
    int32_t dtype;   /* nnop_dtype */
    int32_t n;       /* N = size(x,1) */
    int64_t batch;   /* size(x,2) */
} nnop_softmax_desc;

int nnop_online_softmax(const nnop_softmax_desc* d, void* y, const void* x, nnop_stream_t stream);
int nnop_online_softmax_bwd(const nnop_softmax_desc* d, void* dx, const void* dy, const void* y,
                            nnop_stream_t stream);

/*
 * RMSNorm and LayerNorm.
 *   NNop._rms_norm(x, w; ϵ, offset) -> (y, rms)                  src/rms_norm.jl:117-137 (kernel :3-38)
 *   NNop.∇rms_norm(Δ, rms, x, w; offset) -> (dx, dw)              src/rms_norm.jl:139-169 (kernel :43-115)
 *   NNop._layer_norm(x, w, b; ϵ) -> (y, μ, Σ)                     src/layer_norm.jl:150-170 (kernel :8-63)
 *   NNop.∇layer_norm(Δ, μ, Σ, x, w, b) -> (dx, dw, db)            src/layer_norm.jl:172-204 (kernel :65-148)
 * x, y, dy, dx : [n][emb] == Julia (emb, n), element type `dtype`;  w, b : [emb], element type `w_dtype`
 * (NNOP_F32 or the same as dtype);  rms, mu, sigma : fp32 [n] (the caches the reference keeps for the pullback:
 * rms = sigma = 1/sqrt(var + eps));  RMSNorm dw : fp32 [emb] (rms_norm.jl:146);  LayerNorm dw, db : `w_dtype` [emb]
 * (layer_norm.jl:179-180).  Any emb >= 1.  fp32 arithmetic, one rounding on store.
 * The pullbacks need caller-owned scratch of nnop_norm_bwd_workspace_bytes() (replaces the reference's
 * (n/4, emb) partial-sum arrays, rms_norm.jl:146, layer_norm.jl:179-180); dx, dw, db are fully overwritten.
 */
typedef struct nnop_norm_desc {
    int32_t dtype;     /* nnop_dtype of x, y, dy, dx */
    int32_t w_dtype;   /* nnop_dtype of w, b (and of LayerNorm's dw, db) */
    int32_t emb;       /* size(x,1) */
    int32_t reserved;  /* must be 0 */
    int64_t n;         /* size(x,2) */
} nnop_norm_desc;

int nnop_rms_norm(const nnop_norm_desc* d, void* y, float* rms, const void* x, const void* w,
                  float offset, float eps, nnop_stream_t stream);
int nnop_rms_norm_bwd(const nnop_norm_desc* d, void* dx, float* dw, const void* dy, const float* rms,
                      const void* x, const void* w, float offset,
                      void* workspace, size_t workspace_bytes, nnop_stream_t stream);
int nnop_layer_norm(const nnop_norm_desc* d, void* y, float* mu, float* sigma, const void* x, const void* w,
                    const void* b, float eps, nnop_stream_t stream);
int nnop_layer_norm_bwd(const nnop_norm_desc* d, void* dx, void* dw, void* db, const void* dy, const float* mu,
                        const float* sigma, const void* x, const void* w,
                        void* workspace, size_t workspace_bytes, nnop_stream_t stream);
/* layer_norm: 0 = RMSNorm pullback, 1 = LayerNorm pullback.  Returns 0 for an invalid descriptor. */
size_t nnop_norm_bwd_workspace_bytes(const nnop_norm_desc* d, int layer_norm);

/*
 * Sharding over several devices (ABI version 6).  The reference has no multi-GPU code; its (batch, kv-head) slices -- a KV head with its
 * QH / KH query heads -- are independent in forward and backward (src/attention.jl:27-28,33; src/attention_bwd.jl:28-29,34), so a host
 * that owns several devices gives rank g of `world` the contiguous unit range [g U / world, (g+1) U / world), U = batch * kh, units
 * numbered u = b * kh + h (batch slowest, as in memory).  In the [B][H][L][E] layout that range is at most three DENSE rectangles
 * (tail of the first batch, whole batches, head of the last batch); each is a problem of its own for nnop_fa_fwd / nnop_fa_bwd at the
 * element offsets below -- pointer arithmetic only, no copies, no collective.  Host-only: touches no device.
 * Returns the number of rectangles written to `out` (0..3), or a negative nnop_status for an invalid descriptor / world / rank.
 */
typedef struct nnop_fa_shard {
    nnop_fa_desc desc;   /* the rectangle as a problem: batch, qh, kh replaced; everything else as in the full problem */
    int32_t  b0, b1;     /* batches  [b0, b1)  of the full problem */
    int32_t  kh0, kh1;   /* kv heads [kh0, kh1) (query heads [kh0, kh1) * qh / kh) */
    uint64_t q_off;      /* ELEMENT offset of the rectangle in q, o, dO, dq   ([B][QH][QL][E]) */
    uint64_t kv_off;     /* ... in k, v, dk, dv                               ([B][KH][KL][E]) */
    uint64_t row_off;    /* ... in ms, ls                                     ([B][QH][QL])    */
    uint64_t mask_off;   /* BYTE offset in kpad_mask                          ([B][KL])        */
    int64_t  pair_off;   /* element offset in pair / dpair ([B][KL][QL][QH]) when the rectangle holds ALL heads of its batches;
                            -1 otherwise: part of a batch's heads is not a dense sub-array of the head-fastest bias layout (the caller
                            then hands that rectangle a head-sliced copy, or shards whole batches: world <= batch) */
} nnop_fa_shard;
int nnop_fa_shards(const nnop_fa_desc* d, int world, int rank, nnop_fa_shard out[3]);

/* NNop._shared_memory(::ROCBackend, device_id) (ext/NNopAMDGPUExt.jl:6-9):
 * hipDeviceProp_t.sharedMemPerBlock of `device` (0-based HIP ordinal). */
int nnop_shared_memory(int device, uint64_t* bytes);

/* Human-readable text for an nnop_status (static storage; never NULL). */
const char* nnop_strerror(int status);

/* ABI version of this header: bumped on any incompatible change. */
#define NNOP_HIP_ABI_VERSION 6
int nnop_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NNOP_HIP_H */
