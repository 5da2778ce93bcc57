// fa_bwd_f32.hip -- backward kernel instantiations for T = float (gfx950 only).
#include "fa_bwd_inst.hpp"
namespace nnop {
template int launch_bwd<float>(const nnop_fa_desc&, const BwdArgs&, hipStream_t);
}
