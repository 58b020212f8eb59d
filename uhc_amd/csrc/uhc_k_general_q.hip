// uhc_k_general_q.hip -- one translation unit of the fused step kernel: the general tier as a persistent consumer of an env queue (sticky tiers).
// Its workgroups have TWO waves (UHC_NW2): wave 0 runs the step like every other tier's workgroup, wave 1 serves every second group of support requests of the MPR
// rounds (uhc_mpr.h: mpr_wave_mw) -- two 79 KiB consumers share a CU, so two of its four SIMDs were idle, and the slowest general-tier env is what the headline's step waits for.
#define UHC_NW2
#include "uhc_physics_impl.h"

extern "C" hipError_t uhc_launch_m0_gen_q(const KernelArgs* A, const double* d_action, const double* d_tbase, const int* d_active, size_t lds_bytes, hipStream_t stream) {
    (void)d_active;
    hipLaunchKernelGGL((uhc_step_queue_kernel<0, 2, true>), dim3(A->grid), dim3(UHC_QUEUE_THREADS), lds_bytes, stream, *A, d_action, d_tbase);
    return hipGetLastError();
}
extern "C" hipError_t uhc_launch_m0_gen_q_lds(size_t lds_bytes) { return hipFuncSetAttribute((const void*)uhc_step_queue_kernel<0, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); }
