// uhc_env_capi.cpp -- host side of the env layer C-ABI (include/uhc_amd.h, "Env layer").
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/uhc_amd.h"
#include "uhc_device_env.h"

extern "C" hipError_t uhc_launch_env_pre(const EnvArgs* E, const int* d_active, hipStream_t s);
extern "C" hipError_t uhc_launch_env_post(int mode, const EnvArgs* E, const double* d_action, const int* d_active, hipStream_t s);
extern "C" hipError_t uhc_launch_env_reset_stage(const EnvArgs* E, const int* env_ids, int n, const double* noise, double* out_qpos,
                                                 double* out_qvel, hipStream_t s);
extern "C" hipError_t uhc_launch_env_set_next(const EnvArgs* E, const int* env_ids, int n, const int* clip_ids, const int* fr_start,
                                              const int* fr_len, const double* noise, hipStream_t s);
extern "C" hipError_t uhc_launch_env_auto_stage(const EnvArgs* E, double* out_qpos, double* out_qvel, int* select, hipStream_t s);
extern "C" int* uhc_internal_env_model(UhcBatch* b, int* n_models);
extern "C" int uhc_internal_set_state_masked(UhcBatch* b, const int* d_select, const double* d_qpos, const double* d_qvel);
extern "C" hipError_t uhc_launch_env_assign(const EnvArgs* E, const int* env_ids, int n, const int* clip_ids, const int* fr_start,
                                            const int* fr_len, hipStream_t s);
// accessors implemented in uhc_capi.cpp
extern "C" int uhc_internal_batch_info(UhcBatch* b, int* n_env, int* nq, int* nv, int* nu, int* nbody, int* action_dim, int* vf_dim,
                                       double* dt, double* base_rot_inv, void** stream, int** reset_mask);
extern "C" int uhc_internal_set_error(const char* msg);
extern "C" int uhc_internal_trailing_free(UhcBatch* b);

#define HIP_OK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess) return uhc_internal_set_error((std::string(#expr) + ": " + hipGetErrorString(e__)).c_str()); \
    } while (0)

struct UhcEnv {
    UhcBatch* b = nullptr;
    EnvArgs E;
    std::vector<void*> allocs;
    double *stage_qpos = nullptr, *stage_qvel = nullptr;
    int* select = nullptr;
    int n_clips = 0;
    int64_t n_frames = 0;
};

template <class T>
static int dalloc(UhcEnv* e, size_t n, T** p) {
    void* q = nullptr;
    if (hipMalloc(&q, (n ? n : 1) * sizeof(T)) != hipSuccess) return uhc_internal_set_error("uhc_env: hipMalloc failed");
    hipMemset(q, 0, (n ? n : 1) * sizeof(T));
    e->allocs.push_back(q);
    *p = (T*)q;
    return 0;
}

extern "C" int32_t uhc_env_create(UhcBatch* b, const UhcEnvDesc* d, UhcEnv** out) {
    if (!b || !d || !out) return uhc_internal_set_error("uhc_env_create: null argument");
    if (d->obs_v < 0 || d->obs_v > 6) return uhc_internal_set_error("uhc_env_create: obs_v must be 0 .. 6");
    if (d->obs_v == 3 && (d->fut_frames < 1 || d->fut_frames > 64 || d->fut_skip < 0)) return uhc_internal_set_error("uhc_env_create: obs_v 3 needs 1 <= fut_frames <= 64, skip >= 0");
    if (d->term_body != 0 && d->term_body != 1) return uhc_internal_set_error("uhc_env_create: term_body must be 0 (cfg.env_term_body 'body') or 1 ('root')");
    if (d->reward_v < 0 || d->reward_v > 5) return uhc_internal_set_error("uhc_env_create: reward_v must be 0 .. 5 (implicit, explicit, implicit_v1_mul, explicit_mul, implicit_v2, implicit_v3)");
    UhcEnv* e = new UhcEnv();
    e->b = b;
    EnvArgs& E = e->E;
    memset(&E, 0, sizeof E);
    void* stream;
    int* mask;
    uhc_internal_batch_info(b, &E.n_env, &E.nq, &E.nv, &E.nu, &E.nbody, &E.action_dim, &E.vf_dim, &E.dt, E.base_rot_inv, &stream, &mask);
    // free objects (expert["obj_pose"], humanoid_im.py:1284-1287): the LAST num_obj bodies of the model, one free joint each; what is in front of
    // them is the humanoid -- the reference's qpos_lim / qvel_lim / body_lim (humanoid_im.py:113-115), which its observations, rewards and
    // termination stop at
    E.n_obj = d->num_obj;
    if (E.n_obj < 0 || E.n_obj > uhc_internal_trailing_free(b)) { delete e; return uhc_internal_set_error("uhc_env_create: num_obj exceeds the free bodies at the end of the model"); }
    E.nqh = E.nq - 7 * E.n_obj; E.nvh = E.nv - 6 * E.n_obj; E.nbh = E.nbody - E.n_obj;
    const int nb = E.nbh - 1;
    // ball-joint humanoid (robot.ball): free root + one ball joint per further body
    E.ball = E.nqh == 7 + 4 * (nb - 1) && E.nvh == 6 + 3 * (nb - 1) && nb > 1 && E.nqh != E.nvh + 1;
    if ((E.nqh != E.nvh + 1 && !E.ball) || E.nq > 192 || nb < 1 || nb > 64) { delete e; return uhc_internal_set_error("uhc_env_create: hinge humanoid (free root + scalar joints) or ball-joint humanoid (free root + one ball joint per body), followed by num_obj free bodies, nq <= 192, expected"); }
    if (E.ball && (d->obs_v != 2 || d->reward_v != 0)) { delete e; return uhc_internal_set_error("uhc_env_create: the ball-joint env has observation v2 (get_full_obs_v2_quat) and reward 0 (world_rfc_implicit_quat)"); }
    E.has_shape = d->has_shape;
    E.obs_v = d->obs_v;
    E.reward_v = d->reward_v;
    E.obs_flags = d->obs_flags;
    E.term_body = d->term_body;
    if (d->obs_v == 0)  // [heading] qpos[2:] velocities expert joint angles [phase]; get_full_obs never appends the shape
        E.obs_dim = (d->obs_flags & 1) + (E.nqh - 2) + ((d->obs_flags & 8) ? 6 : E.nvh) + E.nu + ((d->obs_flags >> 2) & 1);
    else
        E.obs_dim = (d->obs_v == 6 ? 8 + E.nvh + 2 * nb + 11 * (nb - 1) : (d->obs_v == 5 ? 300 : 304) + (d->obs_v == 1 ? 20 : 14) * nb) + (d->has_shape ? 17 : 0);
    if (d->obs_v == 4) E.obs_dim = 28 + 26 * (nb - 1) + (d->has_shape ? 17 : 0);  // get_full_obs_v4 (:769-861): global block | shape | one row of 26 per non-root body
    if (E.ball) E.obs_dim = 7 + 4 * nb + (E.nu + 6) + 3 + 6 * nb + 8 * nb + (d->has_shape ? 17 : 0);  // get_full_obs_v2_quat (:668-756)
    if (d->obs_v == 0) E.has_shape = 0;
    E.fut_frames = d->obs_v == 3 ? d->fut_frames : 1;
    E.fut_skip = d->obs_v == 3 ? d->fut_skip : 0;
    E.obs_dim *= E.fut_frames;
    E.env_episode_len = d->env_episode_len;
    E.expert_trail_steps = d->env_expert_trail_steps;
    for (int k = 0; k < 5; k++) E.ee_body[k] = d->ee_body[k];
    E.body_diff_thresh = d->body_diff_thresh;
    for (int k = 0; k < 16; k++) E.rw[k] = d->reward_weights[k];
    double *w, *rjw;
    if (dalloc(e, nb, &w) || dalloc(e, nb, &rjw)) { delete e; return 1; }
    HIP_OK(hipMemcpy(w, d->jpos_diffw, nb * sizeof(double), hipMemcpyHostToDevice));
    E.jpos_diffw = w;
    {
        std::vector<double> ones(nb, 1.0);
        HIP_OK(hipMemcpy(rjw, d->reward_jpos_diffw ? d->reward_jpos_diffw : ones.data(), nb * sizeof(double), hipMemcpyHostToDevice));
        E.rjw = rjw;
    }
    void* p; int64_t cnt;
    uhc_batch_field(b, UHC_F_QPOS, &p, &cnt); E.qpos = (const double*)p;
    uhc_batch_field(b, UHC_F_QVEL, &p, &cnt); E.qvel = (const double*)p;
    uhc_batch_field(b, UHC_F_XPOS, &p, &cnt); E.xpos = (const double*)p;
    uhc_batch_field(b, UHC_F_XQUAT, &p, &cnt); E.xquat = (const double*)p;
    uhc_batch_field(b, UHC_F_XIPOS, &p, &cnt); E.xipos = (const double*)p;
    uhc_batch_field(b, UHC_F_FAIL, &p, &cnt); E.sim_fail = (const int*)p;
    const size_t N = E.n_env;
    if (dalloc(e, N, &E.clip_id) || dalloc(e, N, &E.e_start) || dalloc(e, N, &E.e_len) || dalloc(e, N, &E.cur_t) || dalloc(e, N, &E.start_ind) ||
        dalloc(e, N * E.nu, &E.target_base) || dalloc(e, N * E.nq, &E.qpos_prev) || dalloc(e, N * E.obs_dim, &E.obs) || dalloc(e, N, &E.reward) ||
        dalloc(e, N * 6, &E.reward_parts) || dalloc(e, N, &E.height_lb) || dalloc(e, N, &E.percent) || dalloc(e, N, &E.body_diff) || dalloc(e, N, &E.done) || dalloc(e, N, &E.fail) ||
        dalloc(e, N, &E.end) || dalloc(e, N * E.nq, &e->stage_qpos) || dalloc(e, N * E.nv, &e->stage_qvel) || dalloc(e, N, &e->select) ||
        dalloc(e, N, &E.next_clip) || dalloc(e, N, &E.next_start) || dalloc(e, N, &E.next_len) || dalloc(e, N, &E.has_next) || dalloc(e, N, &E.consumed) ||
        dalloc(e, N * E.nu, &E.next_noise) || dalloc(e, 2 * N, &E.episode) || dalloc(e, 5 * N, &E.snapshot)) { delete e; return 1; }
    *out = e;
    return 0;
}
extern "C" void uhc_env_free(UhcEnv* e) {
    if (!e) return;
    (void)hipDeviceSynchronize();
    for (void* p : e->allocs) (void)hipFree(p);
    delete e;
}
extern "C" int32_t uhc_env_obs_dim(const UhcEnv* e) { return e ? e->E.obs_dim : -1; }
extern "C" int32_t uhc_env_field(UhcEnv* e, int32_t f, void** p, int64_t* n) {
    if (!e) return uhc_internal_set_error("uhc_env_field: null env");
    const EnvArgs& E = e->E;
    const int64_t N = E.n_env;
    void* ptr = nullptr; int64_t cnt = 0;
    switch (f) {
        case UHC_E_OBS: ptr = E.obs; cnt = N * E.obs_dim; break;
        case UHC_E_REWARD: ptr = E.reward; cnt = N; break;
        case UHC_E_REWARD_PARTS: ptr = E.reward_parts; cnt = N * 6; break;
        case UHC_E_DONE: ptr = E.done; cnt = N; break;
        case UHC_E_FAIL: ptr = E.fail; cnt = N; break;
        case UHC_E_END: ptr = E.end; cnt = N; break;
        case UHC_E_PERCENT: ptr = E.percent; cnt = N; break;
        case UHC_E_CUR_T: ptr = E.cur_t; cnt = N; break;
        case UHC_E_BODY_DIFF: ptr = E.body_diff; cnt = N; break;
        case UHC_E_TARGET_BASE: ptr = E.target_base; cnt = N * E.nu; break;
        case UHC_E_CONSUMED: ptr = E.consumed; cnt = N; break;
        case UHC_E_EPISODE: ptr = E.episode; cnt = 2 * N; break;
        case UHC_E_SNAPSHOT: ptr = E.snapshot; cnt = 5 * N; break;
        default: return uhc_internal_set_error("uhc_env_field: unknown field");
    }
    if (p) *p = ptr;
    if (n) *n = cnt;
    return 0;
}
extern "C" int32_t uhc_env_set_bank(UhcEnv* e, const double* d_frames, int64_t n_frames, const int32_t* d_clip_start,
                                    const double* d_clip_beta, int32_t n_clips) {
    if (!e || !d_frames || !d_clip_start || !d_clip_beta || n_frames < 1 || n_clips < 1) return uhc_internal_set_error("uhc_env_set_bank: bad argument");
    e->E.bank = d_frames; e->E.clip_start = d_clip_start; e->E.clip_beta = d_clip_beta;
    e->n_clips = n_clips; e->n_frames = n_frames;
    e->E.obj_pose = nullptr;  // rows of another bank
    return 0;
}
extern "C" int32_t uhc_env_set_obj_pose(UhcEnv* e, const double* d_obj_pose, int64_t n_frames) {
    if (!e) return uhc_internal_set_error("uhc_env_set_obj_pose: null env");
    if (e->E.n_obj == 0) return d_obj_pose ? uhc_internal_set_error("uhc_env_set_obj_pose: the env was created with num_obj = 0") : 0;
    if (!d_obj_pose || !e->E.bank || n_frames != e->n_frames) return uhc_internal_set_error("uhc_env_set_obj_pose: one row of 7 num_obj numbers per frame of the clip bank (set the bank first)");
    e->E.obj_pose = d_obj_pose;
    return 0;
}
extern "C" int32_t uhc_env_set_clip_models(UhcEnv* e, const int32_t* d_clip_model) {
    if (!e) return uhc_internal_set_error("uhc_env_set_clip_models: null env");
    int n_models = 1;
    int* em = uhc_internal_env_model(e->b, &n_models);
    if (d_clip_model && (!em || n_models < 2)) return uhc_internal_set_error("uhc_env_set_clip_models: the batch was created with a single model");
    e->E.clip_model = d_clip_model;
    e->E.env_model = em;
    return 0;
}
static hipStream_t stream_of(UhcEnv* e) {
    void* s; int* m; int a; double d; double br[4];
    uhc_internal_batch_info(e->b, &a, &a, &a, &a, &a, &a, &a, &d, br, &s, &m);
    return (hipStream_t)s;
}
extern "C" int32_t uhc_env_assign(UhcEnv* e, const int32_t* ids, int32_t n, const int32_t* clip_ids, const int32_t* fr_start, const int32_t* fr_len) {
    if (!e || !ids || !clip_ids || !fr_start || !fr_len || n < 1) return uhc_internal_set_error("uhc_env_assign: bad argument");
    if (!e->E.bank) return uhc_internal_set_error("uhc_env_assign: no clip bank set");
    HIP_OK(uhc_launch_env_assign(&e->E, ids, n, clip_ids, fr_start, fr_len, stream_of(e)));
    return 0;
}
extern "C" int32_t uhc_env_reset(UhcEnv* e, const int32_t* ids, int32_t n, const double* d_noise) {
    if (!e || !ids || n < 1 || n > e->E.n_env) return uhc_internal_set_error("uhc_env_reset: bad argument");
    if (!e->E.bank) return uhc_internal_set_error("uhc_env_reset: no clip bank set");
    if (e->E.n_obj > 0 && !e->E.obj_pose) return uhc_internal_set_error("uhc_env_reset: the model has objects but no obj_pose rows are set (uhc_env_set_obj_pose)");
    hipStream_t s = stream_of(e);
    HIP_OK(uhc_launch_env_reset_stage(&e->E, ids, n, d_noise, e->stage_qpos, e->stage_qvel, s));
    if (uhc_batch_set_state(e->b, ids, n, e->stage_qpos, e->stage_qvel)) return 1;  // set_state + forward on those envs
    void* st; int* mask; int a; double d; double br[4];
    uhc_internal_batch_info(e->b, &a, &a, &a, &a, &a, &a, &a, &d, br, &st, &mask);
    HIP_OK(uhc_launch_env_post(1, &e->E, nullptr, mask, s));  // observation of the reset envs (mask = envs just reset)
    return 0;
}
extern "C" int32_t uhc_env_set_next(UhcEnv* e, const int32_t* ids, int32_t n, const int32_t* clip_ids, const int32_t* fr_start, const int32_t* fr_len,
                                    const double* d_noise) {
    if (!e || !ids || !clip_ids || !fr_start || !fr_len || n < 1 || n > e->E.n_env) return uhc_internal_set_error("uhc_env_set_next: bad argument");
    if (!e->E.bank) return uhc_internal_set_error("uhc_env_set_next: no clip bank set");
    HIP_OK(uhc_launch_env_set_next(&e->E, ids, n, clip_ids, fr_start, fr_len, d_noise, stream_of(e)));
    return 0;
}
extern "C" int32_t uhc_env_set_end_reward(UhcEnv* e, double end_reward) {
    if (!e) return uhc_internal_set_error("uhc_env_set_end_reward: null env");
    e->E.end_reward = end_reward;
    return 0;
}
extern "C" int32_t uhc_env_auto_reset(UhcEnv* e) {
    if (!e) return uhc_internal_set_error("uhc_env_auto_reset: null env");
    if (!e->E.bank) return uhc_internal_set_error("uhc_env_auto_reset: no clip bank set");
    if (e->E.n_obj > 0 && !e->E.obj_pose) return uhc_internal_set_error("uhc_env_auto_reset: the model has objects but no obj_pose rows are set (uhc_env_set_obj_pose)");
    hipStream_t s = stream_of(e);
    HIP_OK(uhc_launch_env_auto_stage(&e->E, e->stage_qpos, e->stage_qvel, e->select, s));
    if (uhc_internal_set_state_masked(e->b, e->select, e->stage_qpos, e->stage_qvel)) return 1;
    HIP_OK(uhc_launch_env_post(1, &e->E, nullptr, e->select, s));
    return 0;
}
extern "C" int32_t uhc_env_step(UhcEnv* e, const double* d_action, const int32_t* d_active) {
    if (!e || !d_action) return uhc_internal_set_error("uhc_env_step: bad argument");
    if (!e->E.bank) return uhc_internal_set_error("uhc_env_step: no clip bank set");
    hipStream_t s = stream_of(e);
    HIP_OK(uhc_launch_env_pre(&e->E, d_active, s));
    if (uhc_batch_simulate(e->b, d_action, e->E.target_base, d_active)) return 1;
    HIP_OK(uhc_launch_env_post(0, &e->E, d_action, d_active, s));
    return 0;
}
