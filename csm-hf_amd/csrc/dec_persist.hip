// Translation unit of the persistent decoder engine (dec_persist.h).
#include <hip/hip_runtime.h>
#define CSM_DEC_PERSIST_KERNEL 1
#include "dec_persist.h"

int configure_dec_persist() { return dpk::configure(); }
int launch_dec_persist(hipStream_t st, const DecPersistArgs& a, int nt) { return dpk::launch(st, a, nt); }
