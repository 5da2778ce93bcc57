// fa_fwd_bf16.hip -- forward kernel instantiations for T = __bf16 (gfx950 only).
#include "fa_fwd_inst.hpp"
namespace nnop {
template int launch_fwd<__bf16>(const nnop_fa_desc&, const FwdArgs&, hipStream_t);
// the one out-of-line copy of the launcher's form rule (reported by nnop_debug_fwd_form)
int fwd_form(const nnop_fa_desc& d, bool has_pair, bool has_mask) { return fwd_form_of(d, fwd_mode(d, has_pair, has_mask)); }
}
