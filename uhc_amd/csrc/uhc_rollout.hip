// Rollout bookkeeping of the sampling loop (uhc/khrylib/rl/agents/agent.py:60-100: running_state(state), select_action, push to
// memory, reward / mask bookkeeping) as a handful of HIP kernels instead of ~80 framework launches per control step.  Everything is
// env-major float64 in HBM; the work per step is ~10 MB of streaming traffic, i.e. launch-latency bound: what matters is the NUMBER of
// launches inside the step's HIP graph, and that every reduction has a fixed order (bit-reproducible runs).
//
//   uhc_rollout_act      state -> states[:, t];  action = mean (mean_flag) | mean + exp(log_std) * noise;  -> actions[:, t], action
//   uhc_rollout_record   rewards[:, t] = reward + end * end_reward;  dones[:, t];  c_reward_sum, c_info_sum += column sums
//   uhc_filter_push      (n, M, S) <- Chan merge with the (optionally 0/1-weighted) rows of x          (ZFilter.rs.push, zfilter.py:17-27)
//   uhc_filter_apply     y = clip((x - M) / (std + 1e-8), +-clip)                                        (ZFilter.__call__, zfilter.py:55-64)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/uhc_amd.h"  // (declares the entry points below with default visibility: the library is built -fvisibility=hidden)

#include <string>

#define WAVE 64

extern "C" int uhc_internal_set_error(const char* msg);
#define HIP_OK(expr)                                                                                                          \
    do {                                                                                                                      \
        hipError_t e__ = (expr);                                                                                              \
        if (e__ != hipSuccess) return uhc_internal_set_error((std::string(#expr) + ": " + hipGetErrorString(e__)).c_str()); \
    } while (0)

// ------------------------------------------------------------------ policy output -> action, rollout buffers
// One thread per (env, j): j < obs_dim copies the filtered state into the pass buffer, the rest forms the action.  The sample is
// mean + std * noise with separate rounding of the product and the sum (no FMA contraction): bit-identical to the framework's
// `loc + scale * randn` on the same noise.
__global__ void __launch_bounds__(256) uhc_rollout_act_kernel(int n_env, int T, const long long* __restrict__ t_dev, int obs_dim, int act_dim,
                                                              const double* __restrict__ state, const double* __restrict__ mean,
                                                              const double* __restrict__ log_std, const double* __restrict__ noise,
                                                              const double* __restrict__ mean_flags /* [T][n_env], 1 = mean action */,
                                                              double* __restrict__ states, double* __restrict__ actions, double* __restrict__ action) {
    const int W = obs_dim + act_dim;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n_env * W) return;
    const int env = (int)(idx / W), j = (int)(idx % W);
    const long long t = t_dev[0];
    if (t < 0 || t >= T) return;  // more steps than the pass was sized for: nothing is written
    if (j < obs_dim) {
        states[((size_t)env * T + t) * obs_dim + j] = state[(size_t)env * obs_dim + j];
        return;
    }
    const int k = j - obs_dim;
    const double m = mean[(size_t)env * act_dim + k];
    double a = m;
    if (mean_flags[(size_t)t * n_env + env] == 0.0) a = __dadd_rn(m, __dmul_rn(exp(log_std[k]), noise[(size_t)env * act_dim + k]));
    actions[((size_t)env * T + t) * act_dim + k] = a;
    action[(size_t)env * act_dim + k] = a;
}

// ------------------------------------------------------------------ reward / done bookkeeping of one step
// One workgroup; the sums over the envs are tree reductions in a fixed order.
__global__ void __launch_bounds__(1024) uhc_rollout_record_kernel(int n_env, int T, const long long* __restrict__ t_dev, const double* __restrict__ reward,
                                                                  const int* __restrict__ done, const int* __restrict__ end, const double* __restrict__ end_reward,
                                                                  const double* __restrict__ parts, int parts_stride, int n_parts,
                                                                  double* __restrict__ rewards, double* __restrict__ dones,
                                                                  double* __restrict__ c_reward_sum, double* __restrict__ c_info_sum,
                                                                  const int* __restrict__ redo, long long* __restrict__ redo_counts) {
    __shared__ double red[1024];
    const int tid = threadIdx.x;
    const long long t = t_dev[0];
    if (t < 0 || t >= T) return;  // (uniform: every thread reads the same counter)
    const double er = end_reward[0];
    double acc[16];  // reward, up to 8 reward terms, seven env counts (exact in float64)
    const int nacc = n_parts + 8;
    for (int k = 0; k < nacc; k++) acc[k] = 0.0;
    for (int e = tid; e < n_env; e += blockDim.x) {
        const double r = reward[e];
        rewards[(size_t)e * T + t] = r + (double)end[e] * er;
        dones[(size_t)e * T + t] = (double)done[e];
        acc[0] += r;
        for (int k = 0; k < n_parts; k++) acc[1 + k] += parts[(size_t)e * parts_stride + k];
        if (redo) {  // UHC_F_REDO of the step: computed by the general kernel / its exact contact solve fell back to sweeps
            acc[n_parts + 1] += (redo[e] & 1) != 0;
            acc[n_parts + 2] += (redo[e] & 2) != 0;
            acc[n_parts + 3] += (redo[e] & 0x80) != 0;  // rows / contacts beyond the last tier's capacity were dropped in THIS step
            acc[n_parts + 4] += (redo[e] & 0x40) != 0;  // computed by the large tier
            acc[n_parts + 5] += (redo[e] & 8) != 0;     // an island with more than 64 force-carrying rows: solved exactly in windows of 64
            acc[n_parts + 6] += (redo[e] & (1 << 30)) != 0;  // (part of) the step solved by Newton on the primal: tier 4
            acc[n_parts + 7] += (redo[e] & (1 << 29)) != 0;  // ... and that iteration stopped at its cap
        }
    }
    for (int k = 0; k < nacc; k++) {
        red[tid] = acc[k];
        __syncthreads();
        for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        if (tid == 0) {
            if (k == 0) c_reward_sum[0] += red[0];
            else if (k <= n_parts) c_info_sum[k - 1] += red[0];
            else if (redo) redo_counts[k - n_parts - 1] += (long long)red[0];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ observation filter: batch push
// Stage 1: workgroup (column group of 64, row block of <= 64 rows), four waves, lane = column (coalesced row reads).  Every wave keeps
// its <= 16 rows in registers, takes their weighted mean and the squared deviations about THAT mean (two passes over registers: no
// cancellation), the four waves are merged (Chan et al.) in LDS in wave order.  Output per row block: count, mean[dim], S[dim].
#define FP_ROWS_PER_BLOCK 64
#define FP_ROWS_PER_WAVE 16
__device__ __forceinline__ void chan_merge(double& na, double& ma, double& Sa, double nb, double mb, double Sb) {
    const double tot = na + nb;
    if (nb == 0.0) return;
    const double d = mb - ma;
    Sa = Sa + Sb + d * d * (na * nb / tot);
    ma = ma + d * (nb / tot);
    na = tot;
}

__global__ void __launch_bounds__(256) uhc_filter_partial_kernel(const double* __restrict__ x, int n_rows, int dim, const int* __restrict__ weights,
                                                                 double* __restrict__ part /* [R][2][dim] */, double* __restrict__ cnt /* [R] */) {
    __shared__ double sm[4][2][WAVE];
    __shared__ double sc[4];
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const int col = blockIdx.x * WAVE + lane;
    const int row0 = blockIdx.y * FP_ROWS_PER_BLOCK + wave * FP_ROWS_PER_WAVE;
    double v[FP_ROWS_PER_WAVE], w[FP_ROWS_PER_WAVE];
    double c = 0.0, s = 0.0;
#pragma unroll
    for (int i = 0; i < FP_ROWS_PER_WAVE; i++) {
        const int r = row0 + i;
        const bool in = r < n_rows;
        w[i] = in ? (weights ? (weights[r] != 0 ? 1.0 : 0.0) : 1.0) : 0.0;
        v[i] = (in && col < dim) ? x[(size_t)r * dim + col] : 0.0;
        c += w[i];
        s += w[i] * v[i];
    }
    const double m = c > 0.0 ? s / c : 0.0;
    double S = 0.0;
#pragma unroll
    for (int i = 0; i < FP_ROWS_PER_WAVE; i++) {
        const double d = v[i] - m;
        S += w[i] * d * d;
    }
    sm[wave][0][lane] = m;
    sm[wave][1][lane] = S;
    if (lane == 0) sc[wave] = c;
    __syncthreads();
    if (wave == 0) {
        double n = sc[0], M = sm[0][0][lane], Sq = sm[0][1][lane];
        for (int k = 1; k < 4; k++) chan_merge(n, M, Sq, sc[k], sm[k][0][lane], sm[k][1][lane]);
        if (col < dim) {
            part[((size_t)blockIdx.y * 2 + 0) * dim + col] = M;
            part[((size_t)blockIdx.y * 2 + 1) * dim + col] = Sq;
        }
        if (blockIdx.x == 0 && lane == 0) cnt[blockIdx.y] = n;
    }
}

// Stage 2: one workgroup, thread = column: the row blocks' partials are merged in block order, then into the running (n, M, S).
// All threads read n before the barrier; thread 0 writes it after.
__global__ void __launch_bounds__(1024) uhc_filter_merge_kernel(int dim, int R, const double* __restrict__ part, const double* __restrict__ cnt,
                                                                double* __restrict__ n_run, double* __restrict__ M_run, double* __restrict__ S_run) {
    const double n0 = n_run[0];
    double ntot = n0;
    for (int col = threadIdx.x; col < dim; col += blockDim.x) {
        double nb = 0.0, mb = 0.0, Sb = 0.0;
        for (int r = 0; r < R; r++) chan_merge(nb, mb, Sb, cnt[r], part[((size_t)r * 2 + 0) * dim + col], part[((size_t)r * 2 + 1) * dim + col]);
        double n = n0, M = M_run[col], S = S_run[col];
        chan_merge(n, M, S, nb, mb, Sb);
        M_run[col] = M;
        S_run[col] = S;
        ntot = n;
    }
    __syncthreads();
    if (threadIdx.x == 0) n_run[0] = ntot;  // thread 0 owns column 0 (dim >= 1), so its ntot is the merged count
}

// ------------------------------------------------------------------ observation filter: normalise
__global__ void __launch_bounds__(256) uhc_filter_apply_kernel(const double* __restrict__ x, int n_rows, int dim, const double* __restrict__ n_run,
                                                               const double* __restrict__ M_run, const double* __restrict__ S_run, int demean, int destd,
                                                               double clip, double* __restrict__ out, long long* __restrict__ t_inc) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0 && t_inc) t_inc[0] += 1;  // the step counter of the pass: this launch is the last one of a control step
    if (idx >= (long long)n_rows * dim) return;
    const int col = (int)(idx % dim);
    const double n = n_run[0], M = M_run[col];
    double y = x[idx];
    if (demean) y = y - M;
    if (destd) {
        const double var = n > 1.0 ? S_run[col] / (n - 1.0) : M * M;
        y = y / (sqrt(var) + 1e-8);
    }
    if (clip != 0.0) y = fmin(fmax(y, -clip), clip);
    out[idx] = y;
}

// ------------------------------------------------------------------ C-ABI (include/uhc_amd.h)
extern "C" int32_t uhc_rollout_act(void* stream, int32_t n_env, int32_t T, const int64_t* d_t, int32_t obs_dim, int32_t act_dim, const double* d_state,
                                   const double* d_mean, const double* d_log_std, const double* d_noise, const double* d_mean_flags, double* d_states,
                                   double* d_actions, double* d_action) {
    if (n_env <= 0 || T <= 0 || obs_dim < 0 || act_dim <= 0 || !d_t || !d_mean || !d_log_std || !d_noise || !d_mean_flags || !d_actions || !d_action ||
        (obs_dim > 0 && (!d_state || !d_states)))
        return uhc_internal_set_error("uhc_rollout_act: bad argument");
    const long long total = (long long)n_env * (obs_dim + act_dim);
    hipLaunchKernelGGL(uhc_rollout_act_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n_env, T, (const long long*)d_t, obs_dim,
                       act_dim, d_state, d_mean, d_log_std, d_noise, d_mean_flags, d_states, d_actions, d_action);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int32_t uhc_rollout_record(void* stream, int32_t n_env, int32_t T, const int64_t* d_t, const double* d_reward, const int32_t* d_done,
                                      const int32_t* d_end, const double* d_end_reward, const double* d_parts, int32_t parts_stride, int32_t n_parts,
                                      double* d_rewards, double* d_dones, double* d_c_reward_sum, double* d_c_info_sum, const int32_t* d_redo,
                                      int64_t* d_redo_counts) {
    if ((d_redo && !d_redo_counts) || n_env <= 0 || T <= 0 || !d_t || !d_reward || !d_done || !d_end || !d_end_reward || !d_rewards || !d_dones || !d_c_reward_sum || n_parts < 0 ||
        n_parts > 8 || (n_parts > 0 && (!d_parts || !d_c_info_sum || parts_stride < n_parts)))
        return uhc_internal_set_error("uhc_rollout_record: bad argument");
    hipLaunchKernelGGL(uhc_rollout_record_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n_env, T, (const long long*)d_t, d_reward, d_done, d_end,
                       d_end_reward, d_parts, parts_stride, n_parts, d_rewards, d_dones, d_c_reward_sum, d_c_info_sum, d_redo, (long long*)d_redo_counts);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int64_t uhc_filter_scratch_doubles(int32_t n_rows, int32_t dim) {
    const int64_t R = (n_rows + FP_ROWS_PER_BLOCK - 1) / FP_ROWS_PER_BLOCK;
    return R * (2 * (int64_t)dim + 1);
}

extern "C" int32_t uhc_filter_push(void* stream, const double* d_x, int32_t n_rows, int32_t dim, const int32_t* d_weights, double* d_n, double* d_mean,
                                   double* d_S, double* d_scratch) {
    if (n_rows < 0 || dim <= 0 || !d_x || !d_n || !d_mean || !d_S || !d_scratch) return uhc_internal_set_error("uhc_filter_push: bad argument");
    if (n_rows == 0) return 0;
    const int R = (n_rows + FP_ROWS_PER_BLOCK - 1) / FP_ROWS_PER_BLOCK;
    double* part = d_scratch;
    double* cnt = d_scratch + (size_t)R * 2 * dim;
    hipLaunchKernelGGL(uhc_filter_partial_kernel, dim3((unsigned)((dim + WAVE - 1) / WAVE), (unsigned)R), dim3(256), 0, (hipStream_t)stream, d_x, n_rows, dim,
                       d_weights, part, cnt);
    HIP_OK(hipGetLastError());
    hipLaunchKernelGGL(uhc_filter_merge_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, dim, R, part, cnt, d_n, d_mean, d_S);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int32_t uhc_filter_apply(void* stream, const double* d_x, int32_t n_rows, int32_t dim, const double* d_n, const double* d_mean, const double* d_S,
                                    int32_t demean, int32_t destd, double clip, double* d_out, int64_t* d_t_inc) {
    if (n_rows < 0 || dim <= 0 || !d_x || !d_n || !d_mean || !d_S || !d_out) return uhc_internal_set_error("uhc_filter_apply: bad argument");
    const long long total = (long long)n_rows * dim;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(uhc_filter_apply_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, (hipStream_t)stream, d_x, n_rows, dim, d_n, d_mean, d_S, demean,
                       destd, clip, d_out, (long long*)d_t_inc);
    HIP_OK(hipGetLastError());
    return 0;
}
