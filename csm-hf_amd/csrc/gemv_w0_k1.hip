// generated stub: gemv kernels for weight dtype float, K-split 1 (see gemv_inst.inc)
#define GEMV_WT float
#define GEMV_KS 1
#define GEMV_FN launch_gemv_w0_k1
#include "gemv_inst.inc"
