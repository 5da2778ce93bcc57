// fa_fwd_f16.hip -- forward kernel instantiations for T = _Float16 (gfx950 only).
#include "fa_fwd_inst.hpp"
namespace nnop {
template int launch_fwd<_Float16>(const nnop_fa_desc&, const FwdArgs&, hipStream_t);
}
