// fp16 operand build of the fused decode kernel (AMUSE_PREC_F16): k_vae_fused.hip compiled with fp16 instead of bf16 MFMA operands -
// see the note at the top of that file and amuse_dev.hpp PREC_F16.
#define AMUSE_OP_F16 1
#include "k_vae_fused.hip"
