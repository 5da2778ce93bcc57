// fa_bwd_f16.hip -- backward kernel instantiations for T = _Float16 (gfx950 only).
#include "fa_bwd_inst.hpp"
namespace nnop {
template int launch_bwd<_Float16>(const nnop_fa_desc&, const BwdArgs&, hipStream_t);
}
