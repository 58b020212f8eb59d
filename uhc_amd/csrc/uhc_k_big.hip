// uhc_k_big.hip -- one translation unit of the fused step kernel (instantiations split across files so that they compile in parallel).
// the large tier's workgroups go on as tier 4 (Newton on the primal, uhc_primal.h) when an env does not fit or its working sets give up
#define UHC_WITH_TIER4
#include "uhc_physics_impl.h"

extern "C" hipError_t uhc_launch_m0_big(const KernelArgs* A, const double* d_action, const double* d_tbase, const int* d_active, size_t lds_bytes, hipStream_t stream) {
    hipLaunchKernelGGL((uhc_step_kernel<0, 3, true>), dim3(A->grid ? A->grid : A->n_env), dim3(UHC_WAVE), lds_bytes, stream, *A, d_action, d_tbase, d_active);
    return hipGetLastError();
}
extern "C" hipError_t uhc_launch_m0_big_lds(size_t lds_bytes) { return hipFuncSetAttribute((const void*)uhc_step_kernel<0, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); }
