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

/* Keys: the TuneKey enumerators of csrc/tuning.hpp (0 fwd_split, 1 fwd_nw, 2 fwd_w64, 3 bwd_big7,
 * 4 norm_bwd_cap, 5 bwd_form, 6 fwd_exact_scale).  value -1 = automatic.  Returns the previous value, or INT_MIN for an
 * unknown key.  Process-wide; takes effect for launches issued after it returns. */
int nnop_debug_set(int key, int value);

/* 1 when the library was built with `make DEV=1` (timing ablations, experimental kernel bodies compiled in). */
int nnop_debug_dev_build(void);

#ifdef __cplusplus
}
#endif
#endif
