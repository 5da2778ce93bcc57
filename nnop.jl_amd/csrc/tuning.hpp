// tuning.hpp -- launch-shape overrides, parsed ONCE.
//
// The launchers never call getenv(): the NNOP_* environment variables below are read a single time
// (std::call_once, at the first launch of the process) into a small table of ints; after that a
// changed environment has no effect and a launch costs one relaxed atomic load per knob.  The
// table can also be written through `nnop_debug_set` (nnop_debug.h) -- a hook for the test-suite,
// which has to drive the same problem through different kernel forms (bitwise-reproducibility
// tests); it is not declared in the public header include/nnop_hip.h and is not part of the ABI.
//
// -1 always means "automatic" (the launcher's own measured heuristic).
#pragma once

namespace nnop {

enum TuneKey {
    kTuneFwdSplit = 0,   // NNOP_FWD_SPLIT   E<=64 plain forward: 0 = 8/4-wave form, 1 = 16-wave split-KV form
    kTuneFwdNW,          // NNOP_FWD_NW      waves per workgroup of the 32-row-per-wave forward (4 | 8)
    kTuneFwdW64,         // NNOP_FWD_W64     64-row-per-wave forward (fa_fwd_w64.hpp): 0 = never, 1 = wherever instantiated
    kTuneBwdBig7,        // NNOP_BWD_BIG7    workgroups from which the 7-wave E=128 backward form is used
    kTuneNormBwdCap,     // NNOP_NORM_BWD_CAP partial rows of the norm pullbacks
    kTuneBwdNW,          // NNOP_BWD_NW      16-bit E <= 64 backward: waves per workgroup (4 | 8)
    kTuneFwdExactScale,  // NNOP_FWD_EXACT_SCALE  64-row forward: 0 = fold scale*log2e into Q (rounded to T; opt-in, faster), 1 / auto = apply it in fp32 per logit
    kTuneBwdW64,         // NNOP_BWD_W64     one-wave-per-SIMD backward (fa_bwd_w64.hpp): 0 never, 1 both passes, 2 dK/dV only, 3 dQ only, 4 both + separate preprocess launch
    kTuneBwdStages,      // (no environment variable; nnop_debug_set only)  measurement only: which passes of the tiled backward run (bit mask: 1 preprocess, 2 dK/dV, 4 dQ)
    kTuneFwdPersist,     // NNOP_FWD_PERSIST 64-row forward as 256 persistent workgroups that walk a static block list: 0 never, 1 / auto wherever the list balances
    kTuneBwdPersist,     // NNOP_BWD_PERSIST the same for the one-wave-per-SIMD backward kernels
    kTuneFwdDuo,         // NNOP_FWD_DUO     two-waves-per-SIMD alternating-phase forward (fa_fwd_duo.hpp, 16-bit E = 64): 0 never, 1 wherever instantiated
    kTuneFwdPersistAsc,  // NNOP_FWD_PERSIST_ASC persistent forward: q-blocks of a column ascending (light first: 1) / descending (0)
    kTuneBwdNarrow,      // NNOP_BWD_NARROW  one-wave-per-SIMD backward with 32 stationary rows per wave (128-row workgroups): 0 never, 1 wherever instantiated
    kTuneFwdCausalAlt,   // NNOP_FWD_CAUSAL_ALT  32-row forward, causal: alternate the q-block direction of consecutive columns (0 never, 1 whenever the XCD remap allows)
    kTuneCount
};

// Current value of a knob (-1 = automatic).
int tune_get(int key);

}  // namespace nnop
