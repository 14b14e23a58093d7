// Test-only instantiation of the fused MLP kernel that takes its dropout decisions from injected mask tensors
// (sdy_mlp_args.keep_hidden / keep_out) instead of the Philox stream: the reference's own nn.Dropout masks, recorded in
// tests/golden, then drive exactly the code of mlp_h3_kernel<true> -- chain beside fc2's MFMAs, hidden tile in LDS, output
// dropout in the epilogue (reference: src/models/sfno/layers.py:76-78).  A translation unit of its own: a third
// instantiation compiled beside the two product ones changes THEIR register allocation (cdna_hip_programming.md rule 19; the
// dropout variant sits at exactly 512 registers), and this one may spill -- it is never on a timed path.
#define SDY_MLP_INJECT_TU 1
#include "mlp_h3.hip"
