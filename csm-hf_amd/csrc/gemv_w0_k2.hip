// generated stub: gemv kernels for weight dtype float, K-split 2 (see gemv_inst.inc)
#define GEMV_WT float
#define GEMV_KS 2
#define GEMV_FN launch_gemv_w0_k2
#include "gemv_inst.inc"
