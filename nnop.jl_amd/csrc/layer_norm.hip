// layer_norm.hip -- LayerNorm forward and pullback (kernels in norm_common.hpp, LN = true).
#include "norm_common.hpp"

namespace nnop {

static NormParams ln_params(const nnop_norm_desc& d) {
    NormParams p{};
    p.emb = d.emb; p.n = d.n; p.inv_emb = 1.0f / (float)d.emb;     // layer_norm.jl:160
    return p;
}

int launch_layer_norm(const nnop_norm_desc& d, void* y, float* mu, float* sigma, const void* x, const void* w,
                      const void* b, float eps, hipStream_t s) {
    NormParams p = ln_params(d);
    p.out = y; p.a = x; p.w = w; p.b = b; p.stat0 = mu; p.stat1 = sigma; p.eps = eps;
    const bool w32 = d.w_dtype == NNOP_F32;
    switch (d.dtype) {
        case NNOP_F32:  return launch_norm_fwd_t<float, float, true>(p, s);
        case NNOP_F16:  return w32 ? launch_norm_fwd_t<_Float16, float, true>(p, s) : launch_norm_fwd_t<_Float16, _Float16, true>(p, s);
        case NNOP_BF16: return w32 ? launch_norm_fwd_t<__bf16, float, true>(p, s) : launch_norm_fwd_t<__bf16, __bf16, true>(p, s);
    }
    return NNOP_ERR_DTYPE;
}

int launch_layer_norm_bwd(const nnop_norm_desc& d, void* dx, void* dw, void* db, const void* dy, const float* mu,
                          const float* sigma, const void* x, const void* w, void* ws, hipStream_t s) {
    NormParams p = ln_params(d);
    p.out = dx; p.a = dy; p.x = x; p.w = w; p.stat0 = const_cast<float*>(mu); p.stat1 = const_cast<float*>(sigma);
    p.part_w = (float*)ws;
    p.part_b = (float*)ws + (size_t)norm_bwd_max_parts(d.n) * (size_t)d.emb;
    const bool w32 = d.w_dtype == NNOP_F32;
    switch (d.dtype) {
        case NNOP_F32:  return launch_norm_bwd_t<float, float, float, true>(p, dw, db, s);
        case NNOP_F16:  return w32 ? launch_norm_bwd_t<_Float16, float, float, true>(p, dw, db, s)
                                   : launch_norm_bwd_t<_Float16, _Float16, _Float16, true>(p, dw, db, s);
        case NNOP_BF16: return w32 ? launch_norm_bwd_t<__bf16, float, float, true>(p, dw, db, s)
                                   : launch_norm_bwd_t<__bf16, __bf16, __bf16, true>(p, dw, db, s);
    }
    return NNOP_ERR_DTYPE;
}

}  // namespace nnop
