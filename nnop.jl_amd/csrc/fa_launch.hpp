// fa_launch.hpp -- internal (C++) launcher interface between the C ABI (nnop_capi.cpp) and the
// per-dtype kernel translation units.  Not part of the public boundary.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nnop_hip.h"

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
};

// One per dtype (fa_fwd_{f32,f16,bf16}.hip).  Return an nnop_status.
template <typename T> int launch_fwd(const nnop_fa_desc& d, const FwdArgs& a, hipStream_t s);
// One per dtype (fa_bwd_{f32,f16,bf16}.hip).
template <typename T> int launch_bwd(const nnop_fa_desc& d, const BwdArgs& a, hipStream_t s);

// Embedding dims the MFMA kernels are instantiated for.
inline bool emb_supported(int e) { return e == 16 || e == 32 || e == 64 || e == 128; }

// bytes of backward scratch: two fp32 per query row (folded log-sum-exp, delta), [2][B][QH][QL]
inline size_t bwd_workspace_bytes(const nnop_fa_desc& d) {
    return 2 * (size_t)d.batch * d.qh * d.ql * sizeof(float);
}

// Optional tuning override (read-only environment): workgroup waves for the forward.
int env_int(const char* name, int dflt);

}  // namespace nnop
