// fa_fwd_f32.hip -- forward kernel instantiations for T = float (gfx950 only).
#include "fa_fwd_inst.hpp"
namespace nnop {
template int launch_fwd<float>(const nnop_fa_desc&, const FwdArgs&, hipStream_t);
}
