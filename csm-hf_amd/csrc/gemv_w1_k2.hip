// generated stub: gemv kernels for weight dtype bf16_t, K-split 2 (see gemv_inst.inc)
#define GEMV_WT bf16_t
#define GEMV_KS 2
#define GEMV_FN launch_gemv_w1_k2
#include "gemv_inst.inc"
