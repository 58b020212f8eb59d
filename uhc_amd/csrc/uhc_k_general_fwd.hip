// uhc_k_general_fwd.hip -- one translation unit of the fused step kernel (instantiations split across files so that they compile in parallel).
#include "uhc_physics_impl.h"

extern "C" hipError_t uhc_launch_m1_gen(const KernelArgs* A, const double* d_action, const double* d_tbase, const int* d_active, size_t lds_bytes, hipStream_t stream) {
    hipLaunchKernelGGL((uhc_step_kernel<1, 2, true>), dim3(A->grid ? A->grid : A->n_env), dim3(UHC_WAVE), lds_bytes, stream, *A, d_action, d_tbase, d_active);
    return hipGetLastError();
}
extern "C" hipError_t uhc_launch_m1_gen_lds(size_t lds_bytes) { return hipFuncSetAttribute((const void*)uhc_step_kernel<1, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); }
extern "C" hipError_t uhc_launch_m2_gen(const KernelArgs* A, const double* d_action, const double* d_tbase, const int* d_active, size_t lds_bytes, hipStream_t stream) {
    hipLaunchKernelGGL((uhc_step_kernel<2, 2, true>), dim3(A->grid ? A->grid : A->n_env), dim3(UHC_WAVE), lds_bytes, stream, *A, d_action, d_tbase, d_active);
    return hipGetLastError();
}
extern "C" hipError_t uhc_launch_m2_gen_lds(size_t lds_bytes) { return hipFuncSetAttribute((const void*)uhc_step_kernel<2, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); }

// set_state: scatter rows of (qpos, qvel) into the listed envs, clear warm start / flags
__global__ void uhc_set_state_kernel(DevState s, int nq, int nv, int nu, const int* env_ids, int n, const double* qpos,
                                     const double* qvel, int* mask) {
    const int r = blockIdx.x;
    if (r >= n) return;
    const int env = env_ids ? env_ids[r] : r;
    for (int i = threadIdx.x; i < nq; i += blockDim.x) s.qpos[(size_t)env * nq + i] = qpos[(size_t)r * nq + i];
    for (int i = threadIdx.x; i < nv; i += blockDim.x) {
        s.qvel[(size_t)env * nv + i] = qvel[(size_t)r * nv + i];
        s.qacc[(size_t)env * nv + i] = 0;
        s.qacc_ws[(size_t)env * nv + i] = 0;
        s.applied[(size_t)env * nv + i] = 0;
    }
    for (int i = threadIdx.x; i < nu; i += blockDim.x) s.ctrl[(size_t)env * nu + i] = 0;
    if (threadIdx.x == 0) { s.fail[env] = 0; s.overflow[env] = 0; s.fresh[env] = 0; mask[env] = 1; }
}

// set_state on every env whose select flag is set; row e of (qpos, qvel) belongs to env e
__global__ void uhc_set_state_masked_kernel(DevState s, int nq, int nv, int nu, int n_env, const int* select, const double* qpos,
                                            const double* qvel, int* mask) {
    const int env = blockIdx.x;
    if (env >= n_env) return;
    const int go = select[env] != 0;
    if (threadIdx.x == 0) mask[env] = go;
    if (!go) return;
    for (int i = threadIdx.x; i < nq; i += blockDim.x) s.qpos[(size_t)env * nq + i] = qpos[(size_t)env * nq + i];
    for (int i = threadIdx.x; i < nv; i += blockDim.x) {
        s.qvel[(size_t)env * nv + i] = qvel[(size_t)env * nv + i];
        s.qacc[(size_t)env * nv + i] = 0;
        s.qacc_ws[(size_t)env * nv + i] = 0;
        s.applied[(size_t)env * nv + i] = 0;
    }
    for (int i = threadIdx.x; i < nu; i += blockDim.x) s.ctrl[(size_t)env * nu + i] = 0;
    if (threadIdx.x == 0) { s.fail[env] = 0; s.overflow[env] = 0; s.fresh[env] = 1; }  // the forward pass of this reset runs at the head of the env's next step
}
extern "C" hipError_t uhc_launch_set_state_masked(const DevState* s, int nq, int nv, int nu, int n_env, const int* select, const double* qpos,
                                                  const double* qvel, int* mask, hipStream_t stream) {
    hipLaunchKernelGGL(uhc_set_state_masked_kernel, dim3(n_env), dim3(UHC_WAVE), 0, stream, *s, nq, nv, nu, n_env, select, qpos, qvel, mask);
    return hipGetLastError();
}

extern "C" hipError_t uhc_launch_set_state(const DevState* s, int nq, int nv, int nu, const int* env_ids, int n,
                                           const double* qpos, const double* qvel, int* mask, hipStream_t stream) {
    hipLaunchKernelGGL(uhc_set_state_kernel, dim3(n), dim3(UHC_WAVE), 0, stream, *s, nq, nv, nu, env_ids, n, qpos, qvel, mask);
    return hipGetLastError();
}

// sticky tiers, head of a control step: snapshot of the tier table + the queues of the general / large tier, which start with the active
// envs that begin the step there (lists[0 .. n_env) = tier 2, lists[n_env .. 2 n_env) = tier 3; counts[2], counts[3]; free slots = -1),
// the cursors the persistent launches share and the producers' exit counters (fin[1]: fast tier's workgroups, fin[2]: general tier's)
__global__ void uhc_tier_lists_kernel(const int* tier, const int* d_active, int n_env, int* tier_now, int* lists, int* counts, int* cursors, int* fin) {
    if (threadIdx.x < 4) { counts[threadIdx.x] = 0; cursors[threadIdx.x] = 0; fin[threadIdx.x] = 0; }
    for (int i = threadIdx.x; i < 2 * n_env; i += blockDim.x) lists[i] = -1;
    __syncthreads();
    for (int env = threadIdx.x; env < n_env; env += blockDim.x) {
        const int t = tier[env];
        tier_now[env] = t;
        if ((t == 2 || t == 3) && (!d_active || d_active[env])) lists[(t - 2) * n_env + atomicAdd(&counts[t], 1)] = env;
    }
}
extern "C" hipError_t uhc_launch_tier_lists(const int* tier, const int* d_active, int n_env, int* tier_now, int* lists, int* counts, int* cursors, int* fin, hipStream_t stream) {
    hipLaunchKernelGGL(uhc_tier_lists_kernel, dim3(1), dim3(256), 0, stream, tier, d_active, n_env, tier_now, lists, counts, cursors, fin);
    return hipGetLastError();
}
