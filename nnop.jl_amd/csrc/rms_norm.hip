// rms_norm.hip -- RMSNorm forward and pullback (kernels in norm_common.hpp, LN = false).
#include "norm_common.hpp"

namespace nnop {

size_t norm_ws_bytes(const nnop_norm_desc& d, bool ln) { return norm_bwd_ws_bytes(d, ln); }

static NormParams rms_params(const nnop_norm_desc& d) {
    NormParams p{};
    p.emb = d.emb; p.n = d.n; p.inv_emb = 1.0f / (float)d.emb;     // rms_norm.jl:127
    return p;
}

int launch_rms_norm(const nnop_norm_desc& d, void* y, float* rms, const void* x, const void* w, float offset, float eps,
                    hipStream_t s) {
    NormParams p = rms_params(d);
    p.out = y; p.a = x; p.w = w; p.stat0 = rms; p.offset = offset; p.eps = eps;
    const bool w32 = d.w_dtype == NNOP_F32;
    switch (d.dtype) {
        case NNOP_F32:  return launch_norm_fwd_t<float, float, false>(p, s);
        case NNOP_F16:  return w32 ? launch_norm_fwd_t<_Float16, float, false>(p, s) : launch_norm_fwd_t<_Float16, _Float16, false>(p, s);
        case NNOP_BF16: return w32 ? launch_norm_fwd_t<__bf16, float, false>(p, s) : launch_norm_fwd_t<__bf16, __bf16, false>(p, s);
    }
    return NNOP_ERR_DTYPE;
}

int launch_rms_norm_bwd(const nnop_norm_desc& d, void* dx, float* dw, const void* dy, const float* rms, const void* x,
                        const void* w, float offset, void* ws, hipStream_t s) {
    NormParams p = rms_params(d);
    p.out = dx; p.a = dy; p.x = x; p.w = w; p.stat0 = const_cast<float*>(rms); p.offset = offset;
    p.part_w = (float*)ws;
    const bool w32 = d.w_dtype == NNOP_F32;
    switch (d.dtype) {
        case NNOP_F32:  return launch_norm_bwd_t<float, float, float, false>(p, dw, nullptr, s);
        case NNOP_F16:  return w32 ? launch_norm_bwd_t<_Float16, float, float, false>(p, dw, nullptr, s)
                                   : launch_norm_bwd_t<_Float16, _Float16, float, false>(p, dw, nullptr, s);
        case NNOP_BF16: return w32 ? launch_norm_bwd_t<__bf16, float, float, false>(p, dw, nullptr, s)
                                   : launch_norm_bwd_t<__bf16, __bf16, float, false>(p, dw, nullptr, s);
    }
    return NNOP_ERR_DTYPE;
}

}  // namespace nnop
