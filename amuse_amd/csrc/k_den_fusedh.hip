// fp16 operand build of the fused pose-space denoiser step (AMUSE_PREC_F16): k_den_fused.hip compiled with fp16 instead of bf16 MFMA
// operands - see amuse_fused.hpp and amuse_dev.hpp PREC_F16.
#define AMUSE_OP_F16 1
#include "k_den_fused.hip"
