// generated stub: gemv kernels for weight dtype float, K-split 4 (see gemv_inst.inc)
#define GEMV_WT float
#define GEMV_KS 4
#define GEMV_FN launch_gemv_w0_k4
#include "gemv_inst.inc"
