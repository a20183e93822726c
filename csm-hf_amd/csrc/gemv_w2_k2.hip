// generated stub: gemv kernels for weight dtype fp8_t (e4m3fn + row scales), K-split 2 (see gemv_inst.inc)
#define GEMV_WT fp8_t
#define GEMV_KS 2
#define GEMV_FN launch_gemv_w2_k2
#include "gemv_inst.inc"
