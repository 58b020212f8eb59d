// uhc_device_env.h -- device-side layout of the env layer (expert clip bank, per-env episode state).
#pragma once
#include <stdint.h>

// one expert frame = UHC_FRAME_STRIDE doubles (features of Humanoid.qpos_fk, torch_smpl_humanoid.py:234-261)
#define UHC_FRAME_STRIDE 584
#define UHC_FR_QPOS 0      // 76
#define UHC_FR_QVEL 76     // 75
#define UHC_FR_WBPOS 151   // 72
#define UHC_FR_WBQUAT 223  // 96
#define UHC_FR_BQUAT 319   // 96
#define UHC_FR_BANGVEL 415 // 72
#define UHC_FR_EE 487      // 15
#define UHC_FR_COM 502     // 3
#define UHC_FR_BCOM 512    // 72 (body_com, used by observation v1)

struct EnvArgs {
    int n_env, nq, nv, nu, nbody, action_dim, vf_dim, obs_dim, obs_v, reward_v, has_shape, env_episode_len, expert_trail_steps, fut_frames, fut_skip, obs_flags, term_body;
    int ball;  // the humanoid has ball joints (robot.ball, `use_quat` in the reference): qpos = root position + nbody - 1 quaternions
    // free objects behind the humanoid (expert["obj_pose"], humanoid_im.py:1284-1287): n_obj free bodies at the END of the model; the
    // humanoid's own extents -- the reference's qpos_lim / qvel_lim / body_lim (humanoid_im.py:113-115) -- are nqh / nvh / nbh
    int n_obj, nqh, nvh, nbh;
    const double* obj_pose;   // [n_frames][7 n_obj], one row per frame of the bank (uhc_env_set_obj_pose), or NULL
    int ee_body[5];
    double dt, body_diff_thresh;
    double rw[16];            // w_p w_v w_e w_c w_vf k_p k_v k_e k_c k_vf | w_wp w_j k_wp k_j
    double base_rot_inv[4];
    // clip bank
    const double* bank;       // [n_frames][UHC_FRAME_STRIDE]
    const int* clip_start;    // [n_clips] first frame of each clip
    const double* clip_beta;  // [n_clips][17]: beta(16), gender
    const double* jpos_diffw; // [nbody-1]
    const double* rjw;        // [nbody-1] reward_weights["jpos_diffw"] of the v2 / v3 rewards
    // simulation state (owned by the UhcBatch)
    const double *qpos, *qvel, *xpos, *xquat, *xipos;
    const int* sim_fail;
    // per-env episode state and outputs (owned by the UhcEnv)
    int *clip_id, *e_start, *e_len, *cur_t, *start_ind;
    double *target_base, *qpos_prev, *obs, *reward, *reward_parts, *percent, *body_diff, *height_lb;
    int *done, *fail, *end;
    // queued next window per env (uhc_env_set_next / uhc_env_auto_reset)
    int *next_clip, *next_start, *next_len, *has_next, *consumed;
    double* next_noise;  // [n_env][nu]
    double *episode, *snapshot;  // [2][n_env] length, return; [5][n_env] done, length, return, percent, consumed (auto_reset)
    double end_reward;
    // per-clip body shape: model index of every clip, and the physics batch's per-env model selector it drives
    const int* clip_model;
    int* env_model;
};
