// generated stub: gemv kernels for weight dtype fp8_t (e4m3fn + row scales), K-split 4 (see gemv_inst.inc)
#define GEMV_WT fp8_t
#define GEMV_KS 4
#define GEMV_FN launch_gemv_w2_k4
#include "gemv_inst.inc"
