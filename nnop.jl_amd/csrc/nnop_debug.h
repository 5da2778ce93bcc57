/*
 * nnop_debug.h -- test-only hook of libnnop_hip.so.  NOT part of the public boundary (include/nnop_hip.h):
 * no reference interface corresponds to it, the Julia extension never binds it, and its keys may change
 * without an ABI version bump.  The test-suite uses it to push one problem through the different kernel
 * forms the launchers choose between (workgroup shapes, split-KV on/off), which is how the
 * bitwise-reproducibility tests compare them.
 */
#ifndef NNOP_DEBUG_H
#define NNOP_DEBUG_H
#ifdef __cplusplus
extern "C" {
#endif
struct nnop_fa_desc;

/* Keys: the TuneKey enumerators of csrc/tuning.hpp (0 fwd_split, 1 fwd_nw, 2 fwd_w64, 3 bwd_big7,
 * 4 norm_bwd_cap, 5 bwd_nw, 6 fwd_exact_scale, 7 bwd_w64, 8 bwd_stages).  value -1 = automatic.  Returns the previous value, or INT_MIN for an
 * unknown key.  Process-wide; takes effect for launches issued after it returns. */
int nnop_debug_set(int key, int value);

/* Which forward kernel form the launcher picks for this problem (reporting only: bench.py names the kernel its roofline line is
 * about): 0 = fa_fwd_kernel (32-row waves), 1 = fa_fwd_split_kernel (split-KV), 2 = fa_fwd_w64_kernel (64-row waves),
 * 3 = fa_fwd_generic_kernel (plain HIP, embedding dims outside the tiled set),
 * negative = nnop_status of an invalid descriptor.  has_pair / has_mask: whether pair / kpad_mask would be non-NULL. */
int nnop_debug_fwd_form(const struct nnop_fa_desc* d, int has_pair, int has_mask);

/* Which backward kernels the launcher picks (reporting only): bit 0 set = dK/dV runs fa_bwd_w64_kernel, bit 1 set = dQ does;
 * otherwise the kernels of csrc/fa_bwd.hpp (or the plain-HIP ones for embedding dims outside the tiled set). */
int nnop_debug_bwd_form(const struct nnop_fa_desc* d, int has_pair, int has_mask);

/* 1 when the library was built with `make DEV=1` (timing ablations, experimental kernel bodies compiled in). */
int nnop_debug_dev_build(void);

#ifdef __cplusplus
}
#endif
#endif
