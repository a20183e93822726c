// Prefill (q_len > 1) flash attention kernels in their own translation unit: compiled with the MFMA accumulators in
// VGPRs (-mllvm --amdgpu-mfma-vgpr-form), because the softmax reads every score right after the product and the
// round trip through the accumulation registers (v_accvgpr_read / write per element per tile) was a third of the loop.
#define CSM_ATTN_PREFILL_KERNELS 1
#include "attn_prefill.h"

int launch_attn_prefill(hipStream_t st, int kvdtype, int B, int hd, const PrefillAttnArgs& a0, int bf16_math) {
  PrefillAttnArgs a = a0;
  if (hd != 64 || a.n_q % a.n_kv != 0 || a.n_q / a.n_kv > 4 || a.S < 1) return -2;
  const dim3 grid((a.S + 31) / 32, a.n_kv, B);
  if (bf16_math == 2) {   // exact on the bf16 pipe: K / V as three pieces in LDS (fp32 cache) or one (bf16 cache)
    const size_t lds = (size_t)(kvdtype == 1 ? 2 : 6) * 64 * 72 * sizeof(bf16_t);
    if (kvdtype == 1) hipLaunchKernelGGL((attn_prefill_x3_kernel<bf16_t>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((attn_prefill_x3_kernel<float>), grid, dim3(256), lds, st, a);
    return (int)hipGetLastError();
  }
  if (bf16_math) {
    // (round 4, profiles/r04_attn_prefill_knockout.txt: two or three K / V tiles in flight instead of one change nothing; the
    //  longest workgroup's chain of 32 key tiles sets the launch time at 2 048 frames)
    a.map = 1;
    if (a.n_kv != 8) a.kvfast = 0;
    const dim3 g2 = a.kvfast ? dim3(a.n_kv, (a.S + 31) / 32, B) : grid;
    if (kvdtype == 1) hipLaunchKernelGGL((attn_prefill_bf16_kernel<bf16_t>), g2, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((attn_prefill_bf16_kernel<float>), g2, dim3(256), 0, st, a);
    return (int)hipGetLastError();
  }
  if (kvdtype == 1) hipLaunchKernelGGL((attn_prefill_kernel<bf16_t>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((attn_prefill_kernel<float>), grid, dim3(256), 0, st, a);
  return (int)hipGetLastError();
}

