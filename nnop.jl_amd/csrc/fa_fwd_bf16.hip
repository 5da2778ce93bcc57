// fa_fwd_bf16.hip -- forward kernel instantiations for T = __bf16 (gfx950 only).
#include "fa_fwd_inst.hpp"
namespace nnop {
template int launch_fwd<__bf16>(const nnop_fa_desc&, const FwdArgs&, hipStream_t);
}
