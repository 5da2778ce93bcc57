// fa_bwd_bf16.hip -- backward kernel instantiations for T = __bf16 (gfx950 only).
#include "fa_bwd_inst.hpp"
namespace nnop {
template int launch_bwd<__bf16>(const nnop_fa_desc&, const BwdArgs&, hipStream_t);
// the one out-of-line copy of the launcher's form rule (reported by nnop_debug_bwd_form; assumes a 16-byte aligned workspace)
int bwd_forms(const nnop_fa_desc& d, bool has_pair) { return emb_tiled(d.emb) ? bwd_w64_forms(d, has_pair, true) : 0; }
}
