// uhc_env.hip -- the Python-side env math of HumanoidEnv.step as device kernels, one env per wavefront:
//   pre : expert target pose for the PD controller (get_expert_kin_pose(delta_t=1), humanoid_im.py:1040)
//   post: cur_t += 1, termination (humanoid_im.py:1223-1243), imitation reward
//         (world_rfc_implicit_reward, uhc/losses/reward_function.py:12-88) and observation v2
//         (get_full_obs_v2, humanoid_im.py:419-503), written straight into the on-device rollout buffers.
// Expert clips live in HBM as a bank of per-frame records (uhc_device_env.h).
#include <hip/hip_runtime.h>

#include "uhc_device_env.h"

#define LANE ((int)threadIdx.x)
#define WAVE 64

namespace {

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ void qmul(double* r, const double* a, const double* b) {
    const double t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    const double t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    const double t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    const double t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
}
__device__ __forceinline__ void qinv(double* r, const double* q) {  // conj / |q|^2 (transformation.py:1509-1520)
    const double n = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    r[0] = q[0] / n; r[1] = -q[1] / n; r[2] = -q[2] / n; r[3] = -q[3] / n;
}
__device__ __forceinline__ void qinv_batch(double* r, const double* q) {  // conj / |q| (quaternion_inverse_batch, transformation.py:1523-1534: the norm, not its square)
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    r[0] = q[0] / n; r[1] = -q[1] / n; r[2] = -q[2] / n; r[3] = -q[3] / n;
}
// rotation of a (not necessarily unit) quaternion: normalises like quaternion_matrix (transformation.py:1344-1368)
__device__ __forceinline__ void qmat(double* m, const double* q) {
    const double n = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    const double s = sqrt(2.0 / n);
    const double w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
    m[0] = 1.0 - y * y - z * z; m[1] = x * y - z * w; m[2] = x * z + y * w;
    m[3] = x * y + z * w; m[4] = 1.0 - x * x - z * z; m[5] = y * z - x * w;
    m[6] = x * z - y * w; m[7] = y * z + x * w; m[8] = 1.0 - x * x - y * y;
}
__device__ __forceinline__ void rotT(double* r, const double* m, const double* v) {  // R^T v
    r[0] = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
    r[1] = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
    r[2] = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
}
__device__ __forceinline__ void heading_q(double* hq, const double* q) {  // math_utils.py:134-139
    const double n = sqrt(q[0] * q[0] + q[3] * q[3]);
    hq[0] = q[0] / n; hq[1] = 0; hq[2] = 0; hq[3] = q[3] / n;
}
__device__ __forceinline__ double heading(const double* q) {  // math_utils.py:177-184
    double w = q[0], z = q[3];
    if (z < 0) { w = -w; z = -z; }
    return 2 * acos(w / sqrt(w * w + z * z));
}
__device__ __forceinline__ double heading_new(const double* q) {  // math_utils.py:185-190
    return atan2(2 * (q[0] * q[3] + q[1] * q[2]), 1 - 2 * (q[2] * q[2] + q[3] * q[3]));
}
__device__ __forceinline__ void euler_rzyx(double* q, double az, double ay, double ax) {  // quaternion_from_euler(..., 'rzyx')
    double sz, cz, sy, cy, sx, cx;
    sincos(0.5 * az, &sz, &cz); sincos(0.5 * ay, &sy, &cy); sincos(0.5 * ax, &sx, &cx);
    q[0] = cx * cy * cz + sx * sy * sz; q[1] = sx * cy * cz - cx * sy * sz;
    q[2] = cx * sy * cz + sx * cy * sz; q[3] = cx * cy * sz - sx * sy * cz;
}
// local body quaternion b (0 = root) from a hinge-model qpos, or -- ball joints -- the qpos entries themselves (humanoid_im.py:925-947)
__device__ __forceinline__ void body_quat(double* q, const double* qpos, int b, int ball) {
    if (b == 0 || ball) { q[0] = qpos[3 + 4 * b]; q[1] = qpos[4 + 4 * b]; q[2] = qpos[5 + 4 * b]; q[3] = qpos[6 + 4 * b]; }
    else euler_rzyx(q, qpos[7 + 3 * (b - 1)], qpos[8 + 3 * (b - 1)], qpos[9 + 3 * (b - 1)]);
}
// entry i of the expert pose an env is reset to: the record's qpos, or -- ball joints -- root position, the quaternion pose's root
// quaternion (kept in the record's root slot) and the expert's local body quaternions (smpl_to_qpose(use_quat=True), smpl_mujoco.py:590-600)
__device__ __forceinline__ double expert_qpos(const double* fr, int i, int ball) {
    return (!ball || i < 7) ? fr[UHC_FR_QPOS + i] : fr[UHC_FR_BQUAT + 4 + (i - 7)];
}
__device__ __forceinline__ int expert_index(int t, int start_ind, int len) { return min(start_ind + t, len - 1); }

}  // namespace

// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WAVE) uhc_env_pre_kernel(EnvArgs E, const int* __restrict__ d_active) {
    const int env = blockIdx.x;
    if (env >= E.n_env || (d_active && !d_active[env])) return;
    const int len = E.e_len[env], ind = expert_index(E.cur_t[env] + 1, E.start_ind[env], len);
    const double* fr = E.bank + (size_t)(E.e_start[env] + ind) * UHC_FRAME_STRIDE;
    for (int a = LANE; a < E.nu; a += WAVE) E.target_base[(size_t)env * E.nu + a] = fr[UHC_FR_QPOS + 7 + a];
    for (int i = LANE; i < E.nq; i += WAVE) E.qpos_prev[(size_t)env * E.nq + i] = E.qpos[(size_t)env * E.nq + i];
}

// MODE 0: after do_simulation (bookkeeping + reward + termination + next observation)
// MODE 1: observation only (reset_model: humanoid_im.py:1245-1299 returns get_obs())
template <int MODE>
__global__ void __launch_bounds__(WAVE) uhc_env_post_kernel(EnvArgs E, const double* __restrict__ d_action, const int* __restrict__ d_active) {
    const int env = blockIdx.x;
    if (env >= E.n_env || (d_active && !d_active[env])) return;
    __shared__ double s_qpos[192], s_prev[192], s_q[4 * 32];
    const int nb = E.nbh - 1;  // the humanoid's bodies without the world (objects, if any, come after them: body_lim)
    const double* qpos_g = E.qpos + (size_t)env * E.nq;
    for (int i = LANE; i < E.nq; i += WAVE) { s_qpos[i] = qpos_g[i]; if (MODE == 0) s_prev[i] = E.qpos_prev[(size_t)env * E.nq + i]; }
    __syncthreads();
    const double* xpos = E.xpos + (size_t)env * 3 * E.nbody;
    const double* xquat = E.xquat + (size_t)env * 4 * E.nbody;
    const double* xipos = E.xipos + (size_t)env * 3 * E.nbody;
    const double* qvel = E.qvel + (size_t)env * E.nv;
    const int len = E.e_len[env], start_ind = E.start_ind[env];
    int cur_t = E.cur_t[env];
    const double* bank0 = E.bank + (size_t)E.e_start[env] * UHC_FRAME_STRIDE;

    if (MODE == 0) {
        cur_t += 1;  // humanoid_im.py:1213
        const double* fr = bank0 + (size_t)expert_index(cur_t, start_ind, len) * UHC_FRAME_STRIDE;
        const double* action = d_action + (size_t)env * E.action_dim;
        // ---- termination: mean weighted body distance (calc_body_diff, humanoid_im.py:1408-1415)
        double dist = 0, wcount = 0, pose2 = 0, vel2 = 0, wpose2 = 0, bcom2 = 0, jpos2 = 0;
        const bool expl = E.reward_v == 1 || E.reward_v == 3;  // explicit residual-force flavour
        const bool v23 = E.reward_v >= 4;                       // world_rfc_implicit_v2 / v3 (reward_function.py:643-820)
        if (LANE < nb) {
            const double w = E.jpos_diffw[LANE];
            double d[3];
            for (int k = 0; k < 3; k++) d[k] = (xpos[3 * (LANE + 1) + k] - fr[UHC_FR_WBPOS + 3 * LANE + k]) * w;
            if (w != 0) { dist = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]); wcount = 1; }
            // ---- reward terms per body (reward_function.py:44-63)
            double cq[4], pq[4], eq[4], ei[4], dq[4], pi[4];
            body_quat(cq, s_qpos, LANE, E.ball);
            body_quat(pq, s_prev, LANE, E.ball);
            for (int k = 0; k < 4; k++) eq[k] = fr[UHC_FR_BQUAT + 4 * LANE + k];
            qinv(ei, eq);
            qmul(dq, cq, ei);
            const double jw = v23 ? E.rjw[LANE] : 0.0;                // v2 / v3: reward_weights["jpos_diffw"] on every term
            const double wq = v23 ? jw : (LANE == 0 ? 1.0 : E.jpos_diffw[LANE]);  // pose_diff[1:] *= body_diffw
            const double pd = acos(fmin(fmax(dq[0], -1.0), 1.0)) * wq;  // multi_quat_norm: no abs
            pose2 = pd * pd;
            qinv(pi, pq);
            qmul(dq, cq, pi);  // get_angvel_fd: cur (x) prev^-1, axis-angle / dt (math_utils.py:92-100)
            double av[3] = {0, 0, 0};
            if (!(fabs(1.0 - dq[0]) < 1e-6 || fabs(1.0 + dq[0]) < 1e-6)) {
                const double ang = 2 * acos(dq[0]);
                const double sh = sin(0.5 * ang);
                double ax[3] = {dq[1] / sh, dq[2] / sh, dq[3] / sh};
                const double an = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
                for (int k = 0; k < 3; k++) av[k] = ax[k] / an * ang / E.dt;
            }
            // explicit variant: unweighted, expert velocity zero past the clip end (reward_function.py:300-301, 308)
            const double wv = (expl || v23) ? 1.0 : w;
            const bool past = expl && start_ind + cur_t >= len;
            for (int k = 0; k < 3; k++) {
                const double dv = (av[k] - (past ? 0.0 : fr[UHC_FR_BANGVEL + 3 * LANE + k])) * wv;
                vel2 += dv * dv;
            }
            if (v23) {
                double cw[4], ew[4];
                for (int k = 0; k < 4; k++) { cw[k] = xquat[4 * (LANE + 1) + k]; ew[k] = fr[UHC_FR_WBQUAT + 4 * LANE + k]; }
                qinv(ei, ew);
                qmul(dq, cw, ei);
                const double wp = acos(fmin(fmax(dq[0], -1.0), 1.0)) * jw;
                wpose2 = wp * wp;
                for (int k = 0; k < 3; k++) {
                    const double dc = (fr[UHC_FR_BCOM + 3 * LANE + k] - xipos[3 * (LANE + 1) + k]) * jw;
                    const double dj = (xpos[3 * (LANE + 1) + k] - fr[UHC_FR_WBPOS + 3 * LANE + k]) * jw;
                    bcom2 += dc * dc; jpos2 += dj * dj;
                }
            }
        }
        if (v23) { wpose2 = wsum(wpose2); bcom2 = wsum(bcom2); jpos2 = wsum(jpos2); }
        const double body_diff = wsum(dist) / fmax(wsum(wcount), 1.0);
        pose2 = wsum(pose2);
        vel2 = wsum(vel2);
        double ee2 = 0;
        if (LANE < 15) {
            const int b = E.ee_body[LANE / 3];
            const double d = xpos[3 * b + LANE % 3] - fr[UHC_FR_EE + LANE];
            ee2 = d * d;
        }
        ee2 = wsum(ee2);
        if (LANE == 0) {
            double com2 = 0, vf2 = 0;
            for (int k = 0; k < 3; k++) { const double d = xipos[3 + k] - fr[UHC_FR_COM + k]; com2 += d * d; }
            // implicit: |vf|^2 (:74-76); explicit: force + torque entries of every body, contact points skipped (:320-327)
            for (int k = 0; k < E.vf_dim; k++) { const double a = action[E.nu + k]; if (!expl || k % 9 >= 3) vf2 += a * a; }
            const double* W = E.rw;  // w_p w_v w_e w_c w_vf k_p k_v k_e k_c k_vf | w_wp w_j k_wp k_j
            double* rp_out = E.reward_parts + (size_t)env * 6;
            double r;
            if (v23) {  // means over bodies (over 3 nb entries for the velocity); v2 multiplies the six terms, v3 sums them with its weights
                const double p0 = exp(-W[5] * pose2 / nb), p1 = exp(-W[12] * wpose2 / nb), p2 = exp(-W[8] * bcom2 / nb), p3 = exp(-W[13] * jpos2 / nb);
                const double p4 = exp(-W[6] * vel2 / (3 * nb)), p5 = exp(-W[9] * vf2);
                r = E.reward_v == 5 ? W[0] * p0 + W[10] * p1 + W[3] * p2 + W[11] * p3 + W[1] * p4 + W[4] * p5 : p0 * p1 * p2 * p3 * p4 * p5;
                rp_out[0] = p0; rp_out[1] = p1; rp_out[2] = p2; rp_out[3] = p3; rp_out[4] = p4; rp_out[5] = p5;
            } else {
                const double rp = exp(-W[5] * pose2), rv = exp(-W[6] * vel2), re = exp(-W[7] * ee2), rc = exp(-W[8] * com2);
                const double rf = (E.vf_dim > 0 || expl) ? exp(-W[9] * vf2) : 0.0;
                if (E.reward_v >= 2) r = rp * rv * re * rc * ((expl || W[4] != 0.0) ? rf : 1.0);  // _v1_mul (:243-245) / explicit_mul (:424-426)
                else r = (W[0] * rp + W[1] * rv + W[2] * re + W[3] * rc + W[4] * rf) / (W[0] + W[1] + W[2] + W[3] + W[4]);
                rp_out[0] = rp; rp_out[1] = rv; rp_out[2] = re; rp_out[3] = rc; rp_out[4] = rf; rp_out[5] = 0.0;
            }
            E.reward[env] = r;
            E.episode[env] += 1.0;
            // env_term_body "body": mean body distance; "root": root height below the window's lowest expert height - 0.1 (:1225-1230)
            const bool body_fail = E.term_body == 1 ? s_qpos[2] < E.height_lb[env] - 0.1 : body_diff > E.body_diff_thresh;
            const int fail = (E.sim_fail[env] != 0) || body_fail;
            const int end = (cur_t >= E.env_episode_len) || (cur_t + start_ind >= len + E.expert_trail_steps - 1);
            E.fail[env] = fail; E.end[env] = end; E.done[env] = fail || end;
            E.episode[E.n_env + env] += r + (end ? E.end_reward : 0.0);
            E.percent[env] = (double)cur_t / (double)(len - 1);
            E.body_diff[env] = body_diff;
            E.cur_t[env] = cur_t;
        }
    }

    // ------------------------------------------------------------------ observation
    // v2: get_full_obs_v2 (humanoid_im.py:419-503); v1: get_full_obs_v1 (:323-417) = v2 + body-COM blocks;
    // v6: get_full_obs_v6 (:596-666); v3: get_full_obs_v3 (:758-767) = v2 at fut_frames look-aheads of `skip` frames, concatenated
    const int nfut = E.obs_v == 3 ? E.fut_frames : 1;
    const int blk = E.obs_dim / nfut;
    for (int fut = 0; fut < nfut; fut++) {
    double* obs = E.obs + (size_t)env * E.obs_dim + (size_t)fut * blk;
    const double* fr = bank0 + (size_t)expert_index(cur_t + 1 + fut * E.fut_skip, start_ind, len) * UHC_FRAME_STRIDE;
    double rootq[4] = {s_qpos[3], s_qpos[4], s_qpos[5], s_qpos[6]}, crq[4], hq[4], hqi[4], Rr[9], Rc[9], trq[4], tq[4];
    qmul(crq, rootq, E.base_rot_inv);            // remove_base_rot (:263-264)
    for (int k = 0; k < 4; k++) tq[k] = fr[UHC_FR_QPOS + 3 + k];
    qmul(trq, tq, E.base_rot_inv);
    int shape_base = 0;
    if (E.obs_v == 0) {
        // get_full_obs (:290-317): raw root quaternion (no base-rotation removal), expert joint angles of the CURRENT frame, phase
        const double* fr0 = bank0 + (size_t)expert_index(cur_t, start_ind, len) * UHC_FRAME_STRIDE;
        const int oh = E.obs_flags & 1, nvel = (E.obs_flags & 8) ? 6 : E.nvh;
        const int oq = oh, ov = oq + E.nqh - 2, oe = ov + nvel, op = oe + E.nu;
        if (LANE == 0) {
            double v0[3], dh[4];
            if (oh) obs[0] = heading(rootq);
            obs[oq] = s_qpos[2];
            if (E.obs_flags & 2) { heading_q(hq, rootq); qinv(hqi, hq); qmul(dh, hqi, rootq); } else { for (int k = 0; k < 4; k++) dh[k] = rootq[k]; }
            for (int k = 0; k < 4; k++) obs[oq + 1 + k] = dh[k];
            qmat(Rr, rootq);
            const double qv[3] = {qvel[0], qvel[1], qvel[2]};
            rotT(v0, Rr, qv);
            for (int k = 0; k < 3; k++) obs[ov + k] = v0[k];
            if (E.obs_flags & 4) obs[op] = (double)cur_t / (double)len;
        }
        for (int i = LANE; i < E.nu; i += WAVE) { obs[oq + 5 + i] = s_qpos[7 + i]; obs[oe + i] = fr0[UHC_FR_QPOS + 7 + i]; }
        for (int i = LANE + 3; i < nvel; i += WAVE) obs[ov + i] = qvel[i];
    } else if (E.obs_v == 6) {
        const double yaw = heading_new(crq);
        sincos(0.5 * yaw, &hq[3], &hq[0]); hq[1] = 0; hq[2] = 0;  // quaternion_about_axis(yaw, z) (math_utils.py:169-172)
        qmat(Rc, hq);
        if (LANE == 0) {
            double rel[3], rl[3], ci[4], dr[4], v1[3];
            for (int k = 0; k < 3; k++) rel[k] = fr[UHC_FR_QPOS + k] - s_qpos[k];
            rotT(rl, Rc, rel);
            for (int k = 0; k < 3; k++) obs[k] = rl[k];
            double rel_h = heading_new(trq) - yaw;
            if (rel_h > M_PI) rel_h -= 2 * M_PI;
            if (rel_h < -M_PI) rel_h += 2 * M_PI;
            obs[3] = rel_h;
            qinv(ci, crq);
            qmul(dr, trq, ci);
            for (int k = 0; k < 4; k++) obs[4 + k] = dr[k];
            const double qv[3] = {qvel[0], qvel[1], qvel[2]};
            rotT(v1, Rc, qv);
            for (int k = 0; k < 3; k++) obs[8 + k] = v1[k];
        }
        for (int i = LANE + 3; i < E.nvh; i += WAVE) obs[8 + i] = qvel[i];
        const int o1 = 8 + E.nvh, o2 = o1 + 2 * nb, o3 = o2 + 3 * (nb - 1), o4 = o3 + 4 * (nb - 1);
        if (LANE < nb) {
            const int b = LANE;
            double d[3], r[3];
            for (int k = 0; k < 3; k++) d[k] = xpos[3 * (b + 1) + k] - s_qpos[k];
            rotT(r, Rc, d);
            obs[o1 + b] = r[1]; obs[o1 + nb + b] = r[2];   // transform_vec_batch_new(...)[1:] slices the (3, N) result: y and z rows (:645)
            if (b >= 1) {
                double cq[4], tb[4], ci[4], o[4];
                for (int k = 0; k < 3; k++) d[k] = fr[UHC_FR_WBPOS + 3 * b + k] - xpos[3 * (b + 1) + k];
                rotT(r, Rc, d);
                for (int k = 0; k < 3; k++) obs[o2 + k * (nb - 1) + (b - 1)] = r[k];
                body_quat(cq, s_qpos, b, E.ball);
                for (int k = 0; k < 4; k++) { tb[k] = fr[UHC_FR_BQUAT + 4 * b + k]; obs[o3 + 4 * (b - 1) + k] = cq[k]; }
                qinv(ci, cq);
                qmul(o, ci, tb);
                for (int k = 0; k < 4; k++) obs[o4 + 4 * (b - 1) + k] = o[k];
            }
        }
        shape_base = o4 + 4 * (nb - 1);
    } else if (E.obs_v == 4) {
        // get_full_obs_v4 (:769-861): the v2 quantities with the global part first -- heading quaternion | target z + raw target root quaternion | z + de-headed
        // root quaternion | dz + root quaternion difference | qvel[:6] (the first three rotated twice, as in v2) | heading difference | relative root position
        // (here the real offset target_body_qpos[:3] - qpos[:3]) | shape -- and then one row of 26 per non-root body: target / current / difference of its three
        // joint angles, its joint velocities, its position and position error in the root frame, hq^-1 x its world quaternion, its quaternion error.
        const int LB = 28 + (E.has_shape ? 17 : 0);
        heading_q(hq, crq);
        qinv(hqi, hq);
        qmat(Rr, rootq);
        qmat(Rc, crq);
        if (LANE == 0) {
            double dh[4], ci[4], dr[4], v0[3], v1[3], rel[3], rl[3];
            for (int k = 0; k < 4; k++) obs[k] = hq[k];
            qmul(dh, hqi, crq);                      // de_heading(curr_root_quat)
            qinv(ci, crq);
            qmul(dr, trq, ci);
            obs[4] = fr[UHC_FR_QPOS + 2]; obs[9] = s_qpos[2]; obs[14] = fr[UHC_FR_QPOS + 2] - s_qpos[2];
            for (int k = 0; k < 4; k++) { obs[5 + k] = tq[k]; obs[10 + k] = dh[k]; obs[15 + k] = dr[k]; }
            const double qv[3] = {qvel[0], qvel[1], qvel[2]};
            rotT(v0, Rr, qv);
            rotT(v1, Rc, v0);
            for (int k = 0; k < 3; k++) { obs[19 + k] = v1[k]; obs[22 + k] = qvel[3 + k]; }
            double rel_h = heading(trq) - heading(crq);
            if (rel_h > M_PI) rel_h -= 2 * M_PI;
            if (rel_h < -M_PI) rel_h += 2 * M_PI;
            obs[25] = rel_h;
            for (int k = 0; k < 3; k++) rel[k] = fr[UHC_FR_QPOS + k] - s_qpos[k];
            rotT(rl, Rc, rel);
            obs[26] = rl[0]; obs[27] = rl[1];
        }
        if (LANE >= 1 && LANE < nb) {
            const int b = LANE;
            double* row = obs + LB + 26 * (b - 1);
            double d[3], r[3], cq[4], tb[4], o[4], ci[4];
            for (int k = 0; k < 3; k++) {
                const double t = fr[UHC_FR_QPOS + 7 + 3 * (b - 1) + k], c = s_qpos[7 + 3 * (b - 1) + k];
                row[k] = t; row[3 + k] = c; row[6 + k] = t - c; row[9 + k] = qvel[6 + 3 * (b - 1) + k];
            }
            for (int k = 0; k < 3; k++) d[k] = xpos[3 * (b + 1) + k] - s_qpos[k];
            rotT(r, Rc, d);
            for (int k = 0; k < 3; k++) row[12 + k] = r[k];
            for (int k = 0; k < 3; k++) d[k] = fr[UHC_FR_WBPOS + 3 * b + k] - xpos[3 * (b + 1) + k];
            rotT(r, Rc, d);
            for (int k = 0; k < 3; k++) row[15 + k] = r[k];
            const bool unset = xquat[4] == 0.0;                                  // cur_quat[0, 0] == 0 (:839-840)
            for (int k = 0; k < 4; k++) { tb[k] = fr[UHC_FR_WBQUAT + 4 * b + k]; cq[k] = unset ? tb[k] : xquat[4 * (b + 1) + k]; }
            qmul(o, hqi, cq);
            for (int k = 0; k < 4; k++) row[18 + k] = o[k];
            qinv_batch(ci, cq);
            qmul(o, ci, tb);
            for (int k = 0; k < 4; k++) row[22 + k] = o[k];
        }
        shape_base = 28;
    } else if (E.ball) {
        // get_full_obs_v2_quat (:668-756), fed with the quaternion expert pose (the reference hands it the Euler one and raises):
        // [0:4] heading quaternion | target z, z, dz | inv(current quaternions, root without base rotation) x target quaternions (4 nb) |
        // qvel (first three rotated twice, as in v2) | heading difference | "relative position" (bug-compatible, :717) | body positions
        // and differences in the root frame (component-major) | hq^-1 x world quaternions | world quaternions^-1 x expert's | shape
        const int nvh = E.nu + 6;
        const int oq = 7, ov = oq + 4 * nb, oh = ov + nvh, oj = oh + 3, oqw = oj + 6 * nb;
        heading_q(hq, crq);
        qinv(hqi, hq);
        qmat(Rr, rootq);
        qmat(Rc, crq);
        if (LANE == 0) {
            double v0[3], v1[3], rel[3], rl[3];
            for (int k = 0; k < 4; k++) obs[k] = hq[k];
            obs[4] = fr[UHC_FR_QPOS + 2]; obs[5] = s_qpos[2]; obs[6] = fr[UHC_FR_QPOS + 2] - s_qpos[2];
            const double qv[3] = {qvel[0], qvel[1], qvel[2]};
            rotT(v0, Rr, qv);
            rotT(v1, Rc, v0);
            for (int k = 0; k < 3; k++) obs[ov + k] = v1[k];
            double rel_h = heading(trq) - heading(crq);
            if (rel_h > M_PI) rel_h -= 2 * M_PI;
            if (rel_h < -M_PI) rel_h += 2 * M_PI;
            obs[oh] = rel_h;
            for (int k = 0; k < 3; k++) rel[k] = trq[k] - s_qpos[k];
            rotT(rl, Rc, rel);
            obs[oh + 1] = rl[0]; obs[oh + 2] = rl[1];
        }
        for (int i = LANE + 3; i < nvh; i += WAVE) obs[ov + i] = qvel[i];
        if (LANE < nb) {
            const int b = LANE;
            double d[3], r[3], cq[4], tb[4], o[4], ci[4];
            // pose difference: current quaternion b (root: base rotation removed) inverted, times the target's
            if (b == 0) { for (int k = 0; k < 4; k++) { cq[k] = crq[k]; tb[k] = trq[k]; } }
            else { for (int k = 0; k < 4; k++) { cq[k] = s_qpos[3 + 4 * b + k]; tb[k] = fr[UHC_FR_BQUAT + 4 * b + k]; } }
            qinv_batch(ci, cq);  // (the root quaternion is off the unit sphere by the rounded base rotation 0.7071: the two inverses differ at 1e-5)
            qmul(o, ci, tb);
            for (int k = 0; k < 4; k++) obs[oq + 4 * b + k] = o[k];
            for (int k = 0; k < 3; k++) d[k] = xpos[3 * (b + 1) + k] - s_qpos[k];
            rotT(r, Rc, d);
            for (int k = 0; k < 3; k++) obs[oj + k * nb + b] = r[k];
            for (int k = 0; k < 3; k++) d[k] = fr[UHC_FR_WBPOS + 3 * b + k] - xpos[3 * (b + 1) + k];
            rotT(r, Rc, d);
            for (int k = 0; k < 3; k++) obs[oj + 3 * nb + k * nb + b] = r[k];
            const bool unset = xquat[4] == 0.0;
            for (int k = 0; k < 4; k++) { tb[k] = fr[UHC_FR_WBQUAT + 4 * b + k]; cq[k] = unset ? tb[k] : xquat[4 * (b + 1) + k]; }
            qmul(o, hqi, cq);
            for (int k = 0; k < 4; k++) obs[oqw + 4 * b + k] = o[k];
            qinv_batch(ci, cq);  // quaternion_inverse_batch(cur_quat)
            qmul(o, ci, tb);
            for (int k = 0; k < 4; k++) obs[oqw + 4 * nb + 4 * b + k] = o[k];
        }
        shape_base = oqw + 8 * nb;
    } else {
        // v5 (:505-594) = v2 without the leading heading quaternion, with the atan2 yaw, one velocity rotation and a real root offset
        const bool v5 = E.obs_v == 5;
        const double yaw5 = v5 ? heading_new(crq) : 0.0;
        if (v5) { sincos(0.5 * yaw5, &hq[3], &hq[0]); hq[1] = 0; hq[2] = 0; } else heading_q(hq, crq);
        qinv(hqi, hq);
        qmat(Rr, rootq);
        qmat(Rc, crq);
        if (v5) obs -= 4;  // every v2 offset below shifts down by the missing heading block (nothing is written below obs + 4)
        if (LANE == 0) {
            double dh[4], ci[4], dr[4], v0[3], v1[3], rel[3], rl[3];
            if (!v5) for (int k = 0; k < 4; k++) obs[k] = hq[k];
            qmul(dh, hqi, crq);                      // de_heading(curr_root_quat)
            qinv(ci, crq);
            qmul(dr, trq, ci);
            obs[4] = fr[UHC_FR_QPOS + 2]; obs[78] = s_qpos[2]; obs[152] = fr[UHC_FR_QPOS + 2] - s_qpos[2];
            for (int k = 0; k < 4; k++) { obs[5 + k] = tq[k]; obs[79 + k] = dh[k]; obs[153 + k] = dr[k]; }
            const double qv[3] = {qvel[0], qvel[1], qvel[2]};
            rotT(v0, Rr, qv);
            if (v5) rotT(v1, Rc, qv); else rotT(v1, Rc, v0);  // v2 rotates twice, as the reference does (:425, :451)
            for (int k = 0; k < 3; k++) obs[226 + k] = v1[k];
            double rel_h = v5 ? heading_new(trq) - yaw5 : heading(trq) - heading(crq);
            if (rel_h > M_PI) rel_h -= 2 * M_PI;
            if (rel_h < -M_PI) rel_h += 2 * M_PI;
            obs[301] = rel_h;
            for (int k = 0; k < 3; k++) rel[k] = (v5 ? fr[UHC_FR_QPOS + k] : trq[k]) - s_qpos[k];  // v2: target_root_quat[:3] - qpos[:3], bug-compatible (:466)
            rotT(rl, Rc, rel);
            obs[302] = rl[0]; obs[303] = rl[1];
        }
        for (int i = LANE; i < E.nu; i += WAVE) {   // joint angles: target, current, difference
            const double t = fr[UHC_FR_QPOS + 7 + i], c = s_qpos[7 + i];
            obs[9 + i] = t; obs[83 + i] = c; obs[157 + i] = t - c;
        }
        for (int i = LANE + 3; i < E.nvh; i += WAVE) obs[226 + i] = qvel[i];
        const int qb = 304 + (E.obs_v == 1 ? 12 : 6) * nb;   // v1 inserts the two body-COM blocks before the quaternions
        if (LANE < nb) {
            const int b = LANE;
            double d[3], r[3], cq[4], tb[4], o[4], ci[4];
            for (int k = 0; k < 3; k++) d[k] = xpos[3 * (b + 1) + k] - s_qpos[k];
            rotT(r, Rc, d);
            for (int k = 0; k < 3; k++) obs[304 + k * nb + b] = r[k];            // (3, N) raveled: component-major
            for (int k = 0; k < 3; k++) d[k] = fr[UHC_FR_WBPOS + 3 * b + k] - xpos[3 * (b + 1) + k];
            rotT(r, Rc, d);
            for (int k = 0; k < 3; k++) obs[304 + 3 * nb + k * nb + b] = r[k];
            if (E.obs_v == 1) {
                for (int k = 0; k < 3; k++) d[k] = xipos[3 * (b + 1) + k] - s_qpos[k];
                rotT(r, Rc, d);
                for (int k = 0; k < 3; k++) obs[304 + 6 * nb + k * nb + b] = r[k];
                for (int k = 0; k < 3; k++) d[k] = fr[UHC_FR_BCOM + 3 * b + k] - xipos[3 * (b + 1) + k];
                rotT(r, Rc, d);
                for (int k = 0; k < 3; k++) obs[304 + 9 * nb + k * nb + b] = r[k];
            }
            const bool unset = xquat[4] == 0.0;                                  // cur_quat[0, 0] == 0 (:485-486)
            for (int k = 0; k < 4; k++) { tb[k] = fr[UHC_FR_WBQUAT + 4 * b + k]; cq[k] = unset ? tb[k] : xquat[4 * (b + 1) + k]; }
            qmul(o, hqi, cq);
            for (int k = 0; k < 4; k++) obs[qb + 4 * b + k] = o[k];
            qinv_batch(ci, cq);  // quaternion_inverse_batch(cur_quat)
            qmul(o, ci, tb);
            for (int k = 0; k < 4; k++) obs[qb + 4 * nb + 4 * b + k] = o[k];
        }
        shape_base = qb + 8 * nb;
        if (v5) { obs += 4; shape_base -= 4; }
    }
    if (E.has_shape) {
        const double* cb = E.clip_beta + (size_t)E.clip_id[env] * 17;
        if (LANE < 17) obs[shape_base + LANE] = cb[LANE];  // beta(16), gender
    }
    }
    (void)s_q;
}

extern "C" hipError_t uhc_launch_env_pre(const EnvArgs* E, const int* d_active, hipStream_t s) {
    hipLaunchKernelGGL(uhc_env_pre_kernel, dim3(E->n_env), dim3(WAVE), 0, s, *E, d_active);
    return hipGetLastError();
}
extern "C" hipError_t uhc_launch_env_post(int mode, const EnvArgs* E, const double* d_action, const int* d_active, hipStream_t s) {
    if (mode == 0) hipLaunchKernelGGL(uhc_env_post_kernel<0>, dim3(E->n_env), dim3(WAVE), 0, s, *E, d_action, d_active);
    else hipLaunchKernelGGL(uhc_env_post_kernel<1>, dim3(E->n_env), dim3(WAVE), 0, s, *E, d_action, d_active);
    return hipGetLastError();
}

// reset_model (humanoid_im.py:1245-1299): qpos/qvel staging rows <- expert frame 0 (+ joint noise)
__global__ void uhc_env_reset_stage_kernel(EnvArgs E, const int* env_ids, int n, const double* noise, double* out_qpos, double* out_qvel) {
    const int r = blockIdx.x;
    if (r >= n) return;
    const int env = env_ids[r];
    const double* fr = E.bank + (size_t)E.e_start[env] * UHC_FRAME_STRIDE;  // ind = 0
    // the expert velocity of a window's first frame is a copy of its second frame's finite difference
    // (torch_smpl_humanoid.py:202-207: qvel = cat(qvel[0:1], qvel)); the bank stores whole clips, so read frame 1
    const double* frv = fr + (E.e_len[env] > 1 ? UHC_FRAME_STRIDE : 0);
    // objects: init_pose = concat(expert pose, obj_pose[ind]), init_vel = concat(expert velocity, zeros) (humanoid_im.py:1284-1287)
    const double* op = E.obj_pose + (size_t)E.e_start[env] * 7 * E.n_obj;
    for (int i = threadIdx.x; i < E.nq; i += blockDim.x) {
        double v = i < E.nqh ? expert_qpos(fr, i, E.ball) : op[i - E.nqh];
        if (noise && i >= 7 && i < E.nqh && !E.ball) v += noise[(size_t)r * E.nu + (i - 7)];  // (the init noise is defined on joint angles)
        out_qpos[(size_t)r * E.nq + i] = v;
    }
    for (int i = threadIdx.x; i < E.nv; i += blockDim.x) out_qvel[(size_t)r * E.nv + i] = i < E.nvh ? frv[UHC_FR_QVEL + i] : 0.0;
    if (threadIdx.x == 0) { E.cur_t[env] = 0; E.start_ind[env] = 0; E.done[env] = 0; E.fail[env] = 0; E.end[env] = 0; E.episode[env] = 0.0; E.episode[E.n_env + env] = 0.0; }
}
extern "C" hipError_t uhc_launch_env_reset_stage(const EnvArgs* E, const int* env_ids, int n, const double* noise, double* out_qpos,
                                                 double* out_qvel, hipStream_t s) {
    hipLaunchKernelGGL(uhc_env_reset_stage_kernel, dim3(n), dim3(WAVE), 0, s, *E, env_ids, n, noise, out_qpos, out_qvel);
    return hipGetLastError();
}
// expert["height_lb"] = min root height over the window (torch_smpl_humanoid.py:250), for env_term_body "root"
__device__ __forceinline__ double window_height_lb(const EnvArgs& E, int env) {
    const double* fr = E.bank + (size_t)E.e_start[env] * UHC_FRAME_STRIDE + UHC_FR_QPOS + 2;
    double lb = fr[0];
    for (int t = 1; t < E.e_len[env]; t++) lb = fmin(lb, fr[(size_t)t * UHC_FRAME_STRIDE]);
    return lb;
}
// load_expert bookkeeping (humanoid_im.py:182-215): env -> (clip, window start, window length)
__global__ void uhc_env_assign_kernel(EnvArgs E, const int* env_ids, int n, const int* clip_ids, const int* fr_start, const int* fr_len) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int env = env_ids[r], c = clip_ids[r];
    E.clip_id[env] = c;
    E.e_start[env] = E.clip_start[c] + fr_start[r];
    E.e_len[env] = fr_len[r];
    if (E.clip_model) E.env_model[env] = E.clip_model[c];
    if (E.term_body == 1) E.height_lb[env] = window_height_lb(E, env);
}
extern "C" hipError_t uhc_launch_env_assign(const EnvArgs* E, const int* env_ids, int n, const int* clip_ids, const int* fr_start,
                                            const int* fr_len, hipStream_t s) {
    hipLaunchKernelGGL(uhc_env_assign_kernel, dim3((n + 63) / 64), dim3(64), 0, s, *E, env_ids, n, clip_ids, fr_start, fr_len);
    return hipGetLastError();
}

// queue of one next window per env
__global__ void uhc_env_set_next_kernel(EnvArgs E, const int* env_ids, int n, const int* clip_ids, const int* fr_start, const int* fr_len,
                                        const double* noise) {
    const int r = blockIdx.x;
    if (r >= n) return;
    const int env = env_ids[r];
    for (int a = threadIdx.x; a < E.nu; a += blockDim.x) E.next_noise[(size_t)env * E.nu + a] = noise ? noise[(size_t)r * E.nu + a] : 0.0;
    if (threadIdx.x == 0) { E.next_clip[env] = clip_ids[r]; E.next_start[env] = fr_start[r]; E.next_len[env] = fr_len[r]; E.has_next[env] = 1; }
}
extern "C" hipError_t uhc_launch_env_set_next(const EnvArgs* E, const int* env_ids, int n, const int* clip_ids, const int* fr_start,
                                              const int* fr_len, const double* noise, hipStream_t s) {
    hipLaunchKernelGGL(uhc_env_set_next_kernel, dim3(n), dim3(WAVE), 0, s, *E, env_ids, n, clip_ids, fr_start, fr_len, noise);
    return hipGetLastError();
}
// every done env: take the queued window (or restart the current one), stage its reset state (frame 0 pose + noise, frame 1
// velocity: see uhc_env_reset_stage_kernel) and raise select[env] for the masked set_state + forward + observation that follow
__global__ void uhc_env_auto_stage_kernel(EnvArgs E, double* out_qpos, double* out_qvel, int* select) {
    const int env = blockIdx.x;
    if (env >= E.n_env) return;
    const bool go = E.done[env] != 0;
    const bool had = go && E.has_next[env] != 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        select[env] = go;
        E.consumed[env] = had;
        const size_t N = E.n_env;
        E.snapshot[env] = go; E.snapshot[N + env] = E.episode[env]; E.snapshot[2 * N + env] = E.episode[N + env];
        E.snapshot[3 * N + env] = E.percent[env]; E.snapshot[4 * N + env] = had;
        if (go) { E.episode[env] = 0.0; E.episode[N + env] = 0.0; }
        if (had) {
            const int c = E.next_clip[env];
            E.clip_id[env] = c; E.e_start[env] = E.clip_start[c] + E.next_start[env]; E.e_len[env] = E.next_len[env]; E.has_next[env] = 0;
            if (E.clip_model) E.env_model[env] = E.clip_model[c];
            if (E.term_body == 1) E.height_lb[env] = window_height_lb(E, env);
        }
        if (go) { E.cur_t[env] = 0; E.start_ind[env] = 0; }
    }
    if (!go) return;
    __syncthreads();
    const double* fr = E.bank + (size_t)E.e_start[env] * UHC_FRAME_STRIDE;
    const double* frv = fr + (E.e_len[env] > 1 ? UHC_FRAME_STRIDE : 0);
    const double* op = E.obj_pose + (size_t)E.e_start[env] * 7 * E.n_obj;
    for (int i = threadIdx.x; i < E.nq; i += blockDim.x) {
        double v = i < E.nqh ? expert_qpos(fr, i, E.ball) : op[i - E.nqh];
        if (had && i >= 7 && i < E.nqh && !E.ball) v += E.next_noise[(size_t)env * E.nu + (i - 7)];
        out_qpos[(size_t)env * E.nq + i] = v;
    }
    for (int i = threadIdx.x; i < E.nv; i += blockDim.x) out_qvel[(size_t)env * E.nv + i] = i < E.nvh ? frv[UHC_FR_QVEL + i] : 0.0;
}
extern "C" hipError_t uhc_launch_env_auto_stage(const EnvArgs* E, double* out_qpos, double* out_qvel, int* select, hipStream_t s) {
    hipLaunchKernelGGL(uhc_env_auto_stage_kernel, dim3(E->n_env), dim3(WAVE), 0, s, *E, out_qpos, out_qvel, select);
    return hipGetLastError();
}
