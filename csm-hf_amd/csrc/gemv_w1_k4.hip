// generated stub: gemv kernels for weight dtype bf16_t, K-split 4 (see gemv_inst.inc)
#define GEMV_WT bf16_t
#define GEMV_KS 4
#define GEMV_FN launch_gemv_w1_k4
#include "gemv_inst.inc"
