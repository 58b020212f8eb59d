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
#define UHC_ORDER_BUCKETS 9
// (tier 4, `launch4` != 0: tier 4's queue consumers run this step; the envs whose last step ended in tier 4 (UHC_DEBUG bit 12 only) head their queue -- lists[2 n_env ..),
//  counts[6] -- and are flagged pend3 = 2, "straight to tier 4"; with launch4 == 0 they are the large tier's like any tier-3 env and the snapshot says 3)
__global__ void uhc_tier_lists_kernel(const int* tier, const int* d_active, int n_env, int* tier_now, int* lists, int* counts, int* cursors, int* fin,
                                      const int* cost, const int* fresh, int* order, int launch4, int* pend3) {
    __shared__ int nb[UHC_ORDER_BUCKETS + 1];
    if (threadIdx.x < 4) counts[threadIdx.x] = 0;
    if (threadIdx.x < 8) cursors[threadIdx.x] = 0;
    if (threadIdx.x >= 6 && threadIdx.x < 8) counts[threadIdx.x] = 0;
    if (threadIdx.x < 8) fin[threadIdx.x] = 0;  // exit counters of the fast / general tier's workgroups [1], [2]; spare seats taken [0]; consumers resident [3], [4]
    if (threadIdx.x <= UHC_ORDER_BUCKETS) nb[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < 3 * n_env; i += blockDim.x) lists[i] = -1;
    __syncthreads();
    // the fast tier's launch order: its envs from the costliest bucket down (cost = how close the env's last step came to the tier's
    // capacity, which is also what its step time grows with), then the envs that are not this launch's.  The launch does not fit the chip
    // at once; whatever starts in its second round is then cheap and far from the capacity -- the envs that may still be handed on are
    // handed on EARLY, while the consumers of the next tier have time left, and the launch's last workgroups are its shortest.
    auto bucket = [&](int env) {
        if (tier[env] != 1 || (d_active && !d_active[env])) return UHC_ORDER_BUCKETS;
        // (an env restarted since its last step has no history: its reset pose may not fit the tier at all -- many do not -- and it is handed
        //  on in its first forward pass; at the head of the launch that happens while the next tier's consumers still have the step ahead)
        if (fresh[env]) return 0;  // (a bucket of their own, ahead of everything: with the row storage in the cost, the next bucket holds hundreds of envs)
        // (round 6: the cost counts the packed row storage too, which on the ball-joint / object models puts most envs within an eighth of the capacity:
        //  the scale is fine where the hand-ons are -- an env that no longer fits in its first forward pass must not start in the launch's last round)
        const int c = cost[env];  // 0 .. 64+ (sixty-fourths of the capacity)
        return c >= 62 ? 1 : c >= 59 ? 2 : c >= 56 ? 3 : c >= 52 ? 4 : c >= 48 ? 5 : c >= 40 ? 6 : c >= 24 ? 7 : 8;
    };
    for (int env = threadIdx.x; env < n_env; env += blockDim.x) {
        int t = tier[env];
        const bool on = !d_active || d_active[env];
        if (t == 4 && fresh[env]) t = 2;  // (a restarted env has no history: its reset pose is the general tier's to look at, not a whole CU's)
        if (t == 4) {
            if (on) atomicAdd(&counts[7], 1);  // (counts[7]: the step's tier-4 envs -- these and every hand-on to tier 4 (KernelArgs::cnt4); the host sizes the NEXT steps' tier-4 consumers by it)
            if (launch4 && on) { lists[2 * n_env + atomicAdd(&counts[6], 1)] = env; pend3[env] = 2; }
            else if (!launch4) t = 3;
        }
        tier_now[env] = t;
        if ((t == 2 || t == 3) && on) lists[(t - 2) * n_env + atomicAdd(&counts[t], 1)] = env;
        if (order) atomicAdd(&nb[bucket(env)], 1);
    }
    __syncthreads();
    if (threadIdx.x < 2) counts[4 + threadIdx.x] = counts[2 + threadIdx.x];  // the queues as the step begins (the host compares with how they end)
    if (!order) return;
    if (threadIdx.x == 0) {
        int run = 0;
        for (int k = 0; k <= UHC_ORDER_BUCKETS; k++) { const int c = nb[k]; nb[k] = run; run += c; }
    }
    __syncthreads();
    for (int env = threadIdx.x; env < n_env; env += blockDim.x) order[atomicAdd(&nb[bucket(env)], 1)] = env;
}
extern "C" hipError_t uhc_launch_tier_lists(const int* tier, const int* d_active, int n_env, int* tier_now, int* lists, int* counts, int* cursors, int* fin,
                                            const int* cost, const int* fresh, int* order, int launch4, int* pend3, hipStream_t stream) {
    hipLaunchKernelGGL(uhc_tier_lists_kernel, dim3(1), dim3(256), 0, stream, tier, d_active, n_env, tier_now, lists, counts, cursors, fin, cost, fresh, order, launch4, pend3);
    return hipGetLastError();
}

// Holds a stream until `want` consumer workgroups have started (their LDS is then theirs), or 200 us have passed: what follows on the
// stream -- the fast tier's launch -- would otherwise take every CU's LDS first.  waited: 100 MHz ticks spent here (diagnostic).
__global__ void uhc_gate_kernel(const int* started, int want, int* waited, long long* trace) {
    const unsigned long long t0 = wall_clock64();
    const int s0 = __hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && wall_clock64() - t0 < 20000ull) __builtin_amdgcn_s_sleep(16);
    if (waited) *waited = (int)(wall_clock64() - t0);
    if (trace) { trace[0] = (long long)t0; trace[1] = (long long)wall_clock64(); trace[2] = s0; trace[3] = __hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); trace[4] = want; }
}
extern "C" hipError_t uhc_launch_gate(const int* started, int want, int* waited, long long* trace, hipStream_t stream) {
    hipLaunchKernelGGL(uhc_gate_kernel, dim3(1), dim3(1), 0, stream, started, want, waited, trace);
    return hipGetLastError();
}
