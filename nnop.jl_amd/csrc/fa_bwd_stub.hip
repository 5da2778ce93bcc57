// TEMPORARY: backward not written yet (replaced by fa_bwd_*.hip).
#include "fa_launch.hpp"
namespace nnop {
template <typename T> int launch_bwd(const nnop_fa_desc&, const BwdArgs&, hipStream_t) { return NNOP_ERR_HIP; }
template int launch_bwd<float>(const nnop_fa_desc&, const BwdArgs&, hipStream_t);
template int launch_bwd<_Float16>(const nnop_fa_desc&, const BwdArgs&, hipStream_t);
template int launch_bwd<__bf16>(const nnop_fa_desc&, const BwdArgs&, hipStream_t);
}
