// uhc_k_huge_q.hip -- one translation unit of the fused step kernel: tier 4 (Newton on the primal problem, uhc_primal.h) as a persistent consumer of
// the queue the large tier's consumers hand on to (sticky tiers): an env that turns out to be beyond the large tier early in the step is taken up at
// once, beside the fast tier's launch, instead of at the step's very end.
// Its workgroups have FOUR waves (UHC_NW4): wave 0 runs the step like every other tier's workgroup, waves 1-3 help with the Newton iteration's row
// passes, Hessian and factorisation (uhc_primal.h) -- the workgroup owns the whole CU's LDS anyway, and its env is what the step is waiting for.
#define UHC_WITH_TIER4
#define UHC_NW4
#include "uhc_physics_impl.h"

extern "C" hipError_t uhc_launch_m0_huge_q(const KernelArgs* A, const double* d_action, const double* d_tbase, const int* d_active, size_t lds_bytes, hipStream_t stream) {
    (void)d_active;
    hipLaunchKernelGGL((uhc_step_queue_kernel<0, 4, true>), dim3(A->grid), dim3(UHC_QUEUE_THREADS), lds_bytes, stream, *A, d_action, d_tbase);
    return hipGetLastError();
}
extern "C" hipError_t uhc_launch_m0_huge_q_lds(size_t lds_bytes) { return hipFuncSetAttribute((const void*)uhc_step_queue_kernel<0, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); }
