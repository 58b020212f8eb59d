// uhc_k_big_q.hip -- one translation unit of the fused step kernel: the large tier as a persistent consumer of an env queue (sticky tiers).
// (no tier 4 INSIDE this persistent consumer: an env beyond the large tier is flagged pend3 = 2 and appended to tier 4's own queue when that has consumers this step
//  -- uhc_k_huge_q.hip, four-wave workgroups --, else it stays flagged for the step's chained launch of uhc_k_big.hip, whose workgroup goes on with it as tier 4)
#include "uhc_physics_impl.h"

extern "C" hipError_t uhc_launch_m0_big_q(const KernelArgs* A, const double* d_action, const double* d_tbase, const int* d_active, size_t lds_bytes, hipStream_t stream) {
    (void)d_active;
    hipLaunchKernelGGL((uhc_step_queue_kernel<0, 3, true>), dim3(A->grid), dim3(UHC_WAVE), lds_bytes, stream, *A, d_action, d_tbase);
    return hipGetLastError();
}
extern "C" hipError_t uhc_launch_m0_big_q_lds(size_t lds_bytes) { return hipFuncSetAttribute((const void*)uhc_step_queue_kernel<0, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); }
