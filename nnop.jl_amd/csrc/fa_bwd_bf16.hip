// fa_bwd_bf16.hip -- backward kernel instantiations for T = __bf16 (gfx950 only).
#include "fa_bwd_inst.hpp"
namespace nnop {
template int launch_bwd<__bf16>(const nnop_fa_desc&, const BwdArgs&, hipStream_t);
}
