/*
 * uhc_amd.h -- C-ABI of libuhc_amd.so: batched SMPL-humanoid rigid-body simulation on MI355X.
 *
 * This is the drop-in boundary for the hot path named in BASELINE.json (north_star).  The
 * reference (ZhengyiLuo/UHC) reaches its physics through the mujoco-py object API; each entry
 * point below names the reference call site it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; uhc_last_error() gives the message
 *     (thread-local).  No exception crosses this boundary.  A physics blow-up is NOT an error: it
 *     raises the per-env `fail` flag, mirroring uhc/envs/humanoid_im.py:1207-1211.
 *   - handles are opaque; the library owns all device buffers it allocates.
 *   - `double* d_*` / `int* d_*` arguments are DEVICE pointers (HBM resident) unless the name
 *     starts with `h_`; all arrays are dense row-major, env-major ([n_env][dim]).
 *   - a UhcBatch is bound to one device and one HIP stream; calls on one batch are not
 *     re-entrant, different batches are independent (one per GPU / rank).
 *   - all arithmetic is float64, like the reference (scripts/train_uhc.py:80-81).
 */
#ifndef UHC_AMD_H
#define UHC_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: what this header declares is its whole export table */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define UHC_ABI_VERSION 10  /* 10: uhc_build_flags; 9: UHC_F_REDO bits 29 / 30 (tier 4), swept-substep bits 8 .. 28; uhc_rollout_record counts int64 [7] */

/* joint / geom type codes (MuJoCo numbering) */
enum { UHC_JNT_FREE = 0, UHC_JNT_BALL = 1, UHC_JNT_SLIDE = 2, UHC_JNT_HINGE = 3 };
enum { UHC_GEOM_PLANE = 0, UHC_GEOM_SPHERE = 2, UHC_GEOM_CAPSULE = 3, UHC_GEOM_BOX = 6, UHC_GEOM_MESH = 7 };
/* Geoms that collide: the plane against CONVEX HULLS, and hulls against each other (MPR).  A hull is a mesh geom or a ROUNDED hull: a sphere (one core vertex, its
 * centre) or a capsule (two core vertices, the ends of its segment: geom_pos +- half length along the geom's z axis, in that order), whose surface lies geom_size[0]
 * (the radius) beyond the core along the query direction -- [MJ-ext] mjc_PlaneSphere / mjc_PlaneCapsule for the plane, mjc_Convex's support callbacks for the rest.
 * The core vertices live in the mesh tables (geom_vertadr / geom_vertnum / mesh_vert / mesh_adj) like a mesh's hull vertices, in the BODY frame; boxes carry mass only. */

/*
 * Flat description of one compiled model (host arrays, copied by uhc_model_create).
 * Produced by uhc_amd/model/mjcf.py; replaces mujoco_py.load_model_from_xml
 * (uhc/khrylib/rl/envs/common/mujoco_env.py:18, uhc/envs/humanoid_im.py:1448).
 */
typedef struct UhcModelDesc {
    int32_t nq, nv, nu, nbody, njnt, ngeom, nmeshvert, nmeshadj, nexclude;
    int32_t iterations;        /* solver sweeps cap (MuJoCo opt.iterations, default 100) */
    int32_t plane_mesh_maxcon; /* max contacts a plane-mesh pair may emit */
    int32_t solver;            /* contact solve of the dual QP  min 1/2 f'Af + f'b, f >= 0:
                                * 0 = projected Gauss-Seidel sweeps up to `iterations` / `tolerance` ([MJ-ext] mj_solPGS);
                                * 1 = the same QP solved to its exact optimum by active-set iterations (block principal pivoting:
                                *     factor the free block, flip every infeasible row); envs with friction-loss rows use 0 */
    double timestep, tolerance, meaninertia;
    double gravity[3];
    /* bodies [nbody] */
    const int32_t *body_parentid, *body_jntadr, *body_jntnum, *body_dofadr, *body_dofnum;
    const double *body_pos /*[nbody][3]*/, *body_quat /*[nbody][4]*/, *body_ipos, *body_iquat;
    const double *body_mass, *body_inertia /*[nbody][3]*/, *body_invweight0 /*[nbody][2]*/;
    /* joints [njnt] */
    const int32_t *jnt_type, *jnt_bodyid, *jnt_qposadr, *jnt_dofadr, *jnt_limited;
    const double *jnt_pos, *jnt_axis /*[njnt][3]*/, *jnt_range /*[njnt][2]*/, *jnt_stiffness, *jnt_margin;
    const double *qpos0, *qpos_spring; /*[nq]*/
    /* dofs [nv] */
    const int32_t *dof_bodyid, *dof_jntid, *dof_parentid, *dof_madr /*[nv+1]*/;
    const double *dof_armature, *dof_damping, *dof_frictionloss, *dof_invweight0;
    /* geoms [ngeom] */
    const int32_t *geom_type, *geom_bodyid, *geom_contype, *geom_conaffinity, *geom_condim;
    const int32_t *geom_vertadr, *geom_vertnum;
    const double *geom_pos, *geom_quat, *geom_size /*[ngeom][3]*/, *geom_friction /*[ngeom][3]*/;
    const double *geom_margin, *geom_gap, *geom_solref /*[ngeom][2]*/, *geom_solimp /*[ngeom][5]*/;
    const double *geom_rbound, *geom_center /*[ngeom][3] body frame*/;
    const double *mesh_vert;       /* [nmeshvert][3], body frame */
    const int32_t *mesh_adjadr;    /* [nmeshvert+1] CSR */
    const int32_t *mesh_adj;       /* [nmeshadj] global vertex ids */
    const int32_t *exclude_pair;   /* [nexclude][2] body ids */
    /* actuators [nu]: motors on joints ([MJ-ext] joint transmission).  actuator_dofid = first dof of the joint; actuator_gear [nu][3]:
     * gear[0] scales the control on a hinge / slide dof, on a ball joint the generalised force on its three dofs is gear[0..2] * ctrl
     * (a torque vector in the child body's frame: uhc/khrylib/mocap/skeleton_mesh_v2.py:183-193 writes one such motor per bone axis) */
    const int32_t *actuator_dofid;
    const double *actuator_gear;
} UhcModelDesc;

/*
 * Controller constants of HumanoidEnv.do_simulation / compute_torque / rfc_implicit
 * (uhc/envs/humanoid_im.py:1014-1076, 1136-1190).
 */
typedef struct UhcCtrlDesc {
    int32_t n_substeps;   /* frame_skip, 15 (humanoid_im.py:64) */
    int32_t action_type;  /* 0 = "position" (stable PD), 1 = "torque" (humanoid_im.py:1157-1160) */
    int32_t meta_pd;      /* 0 none, 1 per-substep gains (30 numbers), 2 per-joint (humanoid_im.py:1054-1067) */
    int32_t rfc_mode;     /* 0 none, 1 implicit root wrench (humanoid_im.py:1136-1143), 2 explicit per-body wrenches (:1080-1132) */
    int32_t action_dim;   /* nu + vf_dim + meta_pd_dim (humanoid_im.py:250) */
    int32_t body_vf_dim;  /* explicit: 6 + 3 * residual_force_torque (humanoid_im.py:242) */
    double rfc_scale;     /* residual_force_scale * rfc_rate */
    double rfc_lim;       /* residual_force_lim */
    double base_rot[4];   /* data_specs.base_rot (humanoid_im.py:85) */
    const double *jkp, *jkd, *torque_lim, *a_scale; /* [nu] host arrays */
    const int32_t* vf_body; /* explicit: model body id of each residual-force body, in action order (humanoid_im.py:236-241) */
    int32_t n_vf_body;
    int32_t _pad;
} UhcCtrlDesc;

typedef struct UhcModel UhcModel;
typedef struct UhcBatch UhcBatch;

/* per-env state fields addressable through uhc_batch_field() */
enum UhcField {
    UHC_F_QPOS = 0,      /* [n_env][nq]   data.qpos */
    UHC_F_QVEL = 1,      /* [n_env][nv]   data.qvel */
    UHC_F_XPOS = 2,      /* [n_env][nbody][3] data.body_xpos (as left by the last forward pass) */
    UHC_F_XQUAT = 3,     /* [n_env][nbody][4] data.body_xquat */
    UHC_F_XIPOS = 4,     /* [n_env][nbody][3] data.xipos */
    UHC_F_QM = 5,        /* [n_env][nM]   data.qM (tree-sparse) */
    UHC_F_QFRC_BIAS = 6, /* [n_env][nv]   data.qfrc_bias */
    UHC_F_QACC = 7,      /* [n_env][nv]   data.qacc of the last forward pass (the warm start is kept separately) */
    UHC_F_CTRL = 8,      /* [n_env][nu]   data.ctrl of the last substep */
    UHC_F_NCON = 9,      /* int32 [n_env] data.ncon of the last forward pass */
    UHC_F_NEFC = 10,     /* int32 [n_env] data.nefc */
    UHC_F_FAIL = 11,     /* int32 [n_env] sticky physics-failure flag (NaN / huge qacc, qpos, qvel).  An env that fails at uhc_batch_set_state
                          * (a pose that is not a pose) keeps the qpos / qvel it was handed and runs NO forward pass: its derived fields
                          * (xpos, xquat, xipos, qM, qfrc_bias, qacc, ncon, nefc) are those of its PREVIOUS state and must not be read
                          * until the env has been given a valid state (the reference raises fail from do_simulation only,
                          * uhc/envs/humanoid_im.py:1207-1211; mj_forward on garbage returns garbage there) */
    UHC_F_SOLVER_ITER = 12, /* int32 [n_env] PGS sweeps used by the last solve */
    UHC_F_QFRC_APPLIED = 13, /* [n_env][nv] data.qfrc_applied of the last substep */
    UHC_F_EFC_OVERFLOW = 14, /* int32 [n_env] sticky: constraint rows were dropped (nefc cap) */
    UHC_F_STAGE_PROF = 15,   /* int64 [n_env][40] per-stage shader-cycle counters (profiling builds only) */
    UHC_F_REDO = 16,         /* int32 [n_env] bit 0: the env's last step / forward pass exceeded the fast tier's capacity (64 rows, 16 contacts, packed
                              * row storage, 12 body-body rows) and was computed by the general tier (128 rows, 64 contacts, 20 body-body rows), the
                              * large one (256 / 128 / 32; bit 6) or tier 4 (up to 1024 rows / 192-320 contacts / 128 body-body rows, by what the LDS holds beside the Hessian; bits 6 and 30), all of
                              * which solve the problem exactly: working sets of <= 64 rows on the dual QP in the general / large tier; Newton's method
                              * on the primal problem -- the reference's MuJoCo default -- in tier 4, which takes whatever the large tier cannot hold
                              * or whose working sets it cannot finish (an island with more than 64 force-carrying rows).  Bit 30: (part of) the step
                              * was solved by tier 4; bit 29: one of its Newton iterations stopped at the cap of 100 (the best iterate is used).
                              * Batches whose last tier is the large one (UHC_TIERS=3, or a model whose tier-4 layout does not fit 160 KiB of LDS)
                              * keep rounds 3-4's behaviour: such an island in windows of 64 rows to a KKT residual of 1e-9 (1 + max |b|) (bit 3: it
                              * happened, it is not a fallback); bit 1: in at least one substep the solve fell back to solver 0 (sweeps to tolerance;
                              * with tier 4 only for friction-loss rows); bits 2, 4, 5, diagnostic: why (friction-loss rows / no convergence of the
                              * working sets / a working set the pivoting could not solve); bit 8 + k: substep k (< 21) of the step was one of those
                              * (a checker that follows the same path needs to know which); bit 7: constraint rows / contacts beyond the LAST tier's
                              * capacity were DROPPED in this step (UHC_F_EFC_OVERFLOW is the sticky version, cleared by the env's next set_state);
                              * without tier 4 a forward pass that dropped rows gets a bounded exact attempt and at most 32 sweeps (bits 1 and 7) */
    UHC_F_TIER = 17,         /* int32 [n_env] 1 | 2 | 3 | 4: the tier the env's next step starts in under uhc_batch_set_kernel_path(2) */
    UHC_F_HANDON_WHY = 18    /* int32 [n_env] diagnostic of the last step: bits 0-7 why the fast tier handed the env on, bits 8-15 why the general tier did
                              * (1 contacts, 2 constraint rows, 4 body-body row slots, 8 packed row storage, 16 MPR candidate list beyond the tier's
                              * capacity, 32 the working sets did not finish: more force-carrying rows in an island than a working set holds), bits 16-23 the
                              * substep of the last hand-on, bits 24-31 why the LARGE tier handed the env on to tier 4 (the same reason bits);
                              * 0 = the env stayed in the tier it started in */
};

const char* uhc_last_error(void);
int32_t uhc_abi_version(void);
/* Which kind of build the library is: bit 0 = the solver's measurement switches (UHC_DEBUG bits 8-12) are compiled in (-DUHC_EXPERIMENTS, tools/ builds only),
 * bit 1 = per-stage cycle counters (-DUHC_STAGE_PROF), bit 2 = poisoned LDS, bit 3 = LDS guard words.  0 for the shipped library: a stray UHC_DEBUG cannot
 * change which solver path an env takes (tests/test_capi_symbols.py). */
int32_t uhc_build_flags(void);

/* model lifetime (host side) */
int32_t uhc_model_create(const UhcModelDesc* desc, UhcModel** out);
void uhc_model_free(UhcModel* m);
int32_t uhc_model_nM(const UhcModel* m);

/*
 * Create a batch of n_env environments on `device_id`.
 *   models[n_models]: distinct models; env_model[n_env] (host, may be NULL => all envs use
 *   models[0]) selects each env's model.  All models must share topology (sizes, tree, types);
 *   bodies may differ in geometry/inertia (the reference rebuilds the model per episode from
 *   SMPL shape: uhc/envs/humanoid_im.py:154-180).
 */
int32_t uhc_batch_create(const UhcModel* const* models, int32_t n_models, const int32_t* h_env_model,
                         int32_t n_env, int32_t device_id, const UhcCtrlDesc* ctrl, UhcBatch** out);
void uhc_batch_free(UhcBatch* b);
/* bind to an existing hipStream_t (e.g. torch's current stream); NULL = the device's null stream.
 * A new batch starts on a private non-blocking stream. */
int32_t uhc_batch_set_stream(UhcBatch* b, void* hip_stream);
int32_t uhc_batch_sync(UhcBatch* b);
/* change rfc_scale between iterations (rfc_decay: uhc/agents/agent_copycat.py:283-290) */
int32_t uhc_batch_set_rfc_scale(UhcBatch* b, double rfc_scale);

/* Which kernel tier computes a step.  The fused step kernel exists in four tiers: fast (<= 64 constraint rows / 16 contacts / 12
 * body-body rows per env; Delassus matrix in registers), general (<= 128 / 64 / 20; working sets; two workgroups per CU), large
 * (<= 256 / 128 / 32; a whole CU's LDS) and tier 4 (<= 1024 / 192-320 / 128, rows in HBM, the Hessian of MuJoCo's primal problem in LDS,
 * Newton's method -- the reference's default solver, whose cost does not depend on the number of rows).  A tier that cannot hold an env
 * leaves it untouched and hands it to the next one, from the substep that did not fit; in the chained launches tier 4 has no launch of its own (the large
 * tier's workgroup goes on with it); under sticky tiers (mode 2) it has queue consumers like the general and the large tier.  What exceeds the LAST tier is dropped and flagged (UHC_F_EFC_OVERFLOW; the reference's models ask
 * MuJoCo for njmax 2500 / nconmax 500, uhc/khrylib/mocap/skeleton_mesh.py:46).  UHC_TIERS=2 | 3 in the environment of uhc_batch_create
 * ends the chain at the general / large tier (rounds 2-4's behaviour, kept for A/B measurements).
 *   0 (default) = chain: the fast tier on every env, then the general tier on the envs it handed on, then the large tier;
 *   1 = the general tier first (then the large one): for scenes where nearly every env exceeds the fast tier;
 *   2 = sticky tiers: every env starts a step in the tier that computed its previous step (it comes down a tier only with room to
 *       spare), and the general / large tiers' own envs run on a side stream BESIDE the fast tier's -- their launches last several times
 *       longer per env, in a chain behind the fast tier the whole step would wait for them.  Results do not depend on the mode beyond
 *       rounding (every tier solves the same QP exactly), and a rerun of the same calls takes the same tiers.  Not capture-safe across
 *       uhc_batch_set_stream changes: the side stream forks from and joins the batch's stream with events.
 * UHC_F_REDO of a step: bit 0 = computed by the general or large tier (or tier 4), bit 1 = its exact solve swept in some substep (bits
 * 2-5 why, bits 8+ which substeps), bit 6 = computed by the large tier or tier 4, bit 30 = tier 4 (Newton on the primal).  Timing (uhc_batch_set_timing) brackets the fast tier's launch (modes
 * 0, 2) or the general tier's (mode 1). */
int32_t uhc_batch_set_kernel_path(UhcBatch* b, int32_t mode);

/* switch the contact solver of the dual QP between launches: solver 0 / 1 and the sweep cap, as in UhcModelDesc (MuJoCo's opt.solver /
 * opt.iterations are run-time options too); iterations <= 0 keeps the current cap */
int32_t uhc_batch_set_solver(UhcBatch* b, int32_t solver, int32_t iterations);

/* device pointer + element count of a state field (valid until uhc_batch_free) */
int32_t uhc_batch_field(UhcBatch* b, int32_t field, void** d_ptr, int64_t* count);

/*
 * MujocoEnv.set_state + sim.forward() (uhc/khrylib/rl/envs/common/mujoco_env.py:106-113) for the
 * envs listed in d_env_ids[n] (device int32; NULL => all n_env envs in order, n must be n_env).
 * d_qpos [n][nq], d_qvel [n][nv] are indexed by position in the list.  Clears the fail flag and
 * the warm start (sim.reset(): mujoco_env.py:96) of those envs.
 */
int32_t uhc_batch_set_state(UhcBatch* b, const int32_t* d_env_ids, int32_t n, const double* d_qpos,
                            const double* d_qvel);

/*
 * HumanoidEnv.do_simulation(action, frame_skip) (uhc/envs/humanoid_im.py:1145-1190) for every env:
 * n_substeps x { compute_torque -> clip -> ctrl; rfc_implicit -> qfrc_applied; sim.step() }.
 *   d_action      [n_env][action_dim]  policy output (joint residuals | residual force | meta-PD)
 *   d_target_base [n_env][nu]          expert joint pose of the next frame (get_expert_kin_pose(delta_t=1))
 *   d_active      int32 [n_env] or NULL: envs with 0 are left untouched (finished episodes)
 */
int32_t uhc_batch_simulate(UhcBatch* b, const double* d_action, const double* d_target_base,
                           const int32_t* d_active);

/* Measurement hook: while enabled, every uhc_batch_simulate brackets its fused step kernel (the fast variant
 * of uhc_step_kernel) with HIP events on the batch's stream; uhc_batch_kernel_time drains them (synchronising
 * on the events) and returns the summed kernel time and the number of launches since the last call. */
int32_t uhc_batch_set_timing(UhcBatch* b, int32_t enable);
int32_t uhc_batch_kernel_time(UhcBatch* b, double* total_ms, int32_t* launches);

/* What the fast kernel does with an env that has more than 16 contacts or 64 constraint rows:
 * 0 (default) = leave it to the general kernel (exact, costs a second pass whenever one env overflows);
 * 1 = keep the first 16 contacts / the rows of the contacts that fit completely and carry on (MuJoCo itself drops
 *     contacts beyond nconmax); such envs are reported in UHC_F_EFC_OVERFLOW.  Rows that do not fit the packed
 *     row storage still go to the general kernel. */
int32_t uhc_batch_set_overflow_mode(UhcBatch* b, int32_t truncate);
/* mj_forward only (no control, no integration) on all envs: refreshes xpos/xquat/xipos/qM/qfrc_bias */
int32_t uhc_batch_forward(UhcBatch* b);

/* ------------------------------------------------------------------------------------------------
 * Env layer: what HumanoidEnv.step / reset / load_expert do around the physics, on device.
 * ---------------------------------------------------------------------------------------------- */
typedef struct UhcEnv UhcEnv;

/* constants of HumanoidEnv (uhc/envs/humanoid_im.py) and of the reward (uhc/losses/reward_function.py:12-36) */
typedef struct UhcEnvDesc {
    int32_t obs_v;               /* 2: get_full_obs_v2 (humanoid_im.py:419-503); 1: get_full_obs_v1 (:323-417); 6: get_full_obs_v6 (:596-666);
                                  * 3: get_full_obs_v3 (:758-767) = fut_frames v2 blocks, look-ahead 0, skip, 2 skip, ...;
                                  * 5: get_full_obs_v5 (:505-594); 0: get_full_obs (:290-317), shaped by obs_flags;
                                  * 4: get_full_obs_v4 (:769-861): its obs_full = global block (28) | shape (17) | one row of 26 per non-root body (hinge models) */
    int32_t has_shape;           /* append beta(16) + gender to the observation (humanoid_im.py:1390-1406) */
    int32_t env_episode_len;     /* cfg.env_episode_len */
    int32_t env_expert_trail_steps;
    int32_t ee_body[5];          /* model body ids of SMPL_EE_NAMES (smpl_parser.py:228) */
    int32_t reward_v;            /* 0: world_rfc_implicit_reward (reward_function.py:12-88) = world_rfc_implicit_reward_quat (:92-171);
                                  * 1: world_rfc_explicit_reward (:253-341); 2: world_rfc_implicit_v1_mul (:174-250);
                                  * 3: world_rfc_explicit_mul_reward (:346-430); 4: world_rfc_implicit_v2 (:643-723); 5: world_rfc_implicit_v3 (:726-820) */
    double body_diff_thresh;     /* humanoid_im.py:88-89 */
    double reward_weights[16];   /* w_p w_v w_e w_c w_vf k_p k_v k_e k_c k_vf | w_wp w_j k_wp k_j (reward_v 4, 5) | 0 0 */
    const double* jpos_diffw;    /* [nbody-1] SMPLConverter.get_new_diff_weight() (host) */
    int32_t fut_frames;          /* obs_v 3: cfg.fut_frames (default 10) */
    int32_t fut_skip;            /* obs_v 3: cfg.skip (default 10) */
    int32_t obs_flags;           /* obs_v 0: bit 0 cfg.obs_heading, bit 1 cfg.root_deheading, bit 2 cfg.obs_phase, bit 3 cfg.obs_vel == "root" */
    const double* reward_jpos_diffw; /* reward_v 4, 5: reward_weights["jpos_diffw"] [nbody-1] (host); NULL = ones */
    int32_t term_body;           /* cfg.env_term_body (humanoid_im.py:1223-1230): 0 "body" (mean body distance > body_diff_thresh), 1 "root" (root height
                                  * below the window's lowest expert root height - 0.1); "Head" reads a key the reference never sets */
    int32_t num_obj;             /* expert["num_obj"] (humanoid_im.py:1284-1287): free objects, the LAST num_obj bodies of the model (one free joint
                                  * each).  Observation, reward and termination read the humanoid in front of them -- qpos[:qpos_lim], qvel[:qvel_lim],
                                  * body_xpos[1:body_lim] (humanoid_im.py:113-115, 421-422) --, a reset puts the objects at obj_pose[ind] with zero velocity */
} UhcEnvDesc;

/* expert frame record layout of the clip bank (doubles; see uhc_amd/csrc/uhc_device_env.h) */
#define UHC_FRAME_STRIDE 584

enum UhcEnvField {
    UHC_E_OBS = 0,          /* [n_env][obs_dim] */
    UHC_E_REWARD = 1,       /* [n_env] */
    UHC_E_REWARD_PARTS = 2, /* [n_env][6] pose, vel, ee, com, vf, 0 (reward_v 4, 5: pose, world pose, body com, joint pos, vel, vf) */
    UHC_E_DONE = 3,         /* int32 [n_env] */
    UHC_E_FAIL = 4,         /* int32 [n_env] info["fail"] */
    UHC_E_END = 5,          /* int32 [n_env] info["end"] */
    UHC_E_PERCENT = 6,      /* [n_env] info["percent"] */
    UHC_E_CUR_T = 7,        /* int32 [n_env] */
    UHC_E_BODY_DIFF = 8,    /* [n_env] calc_body_diff() of the last step */
    UHC_E_TARGET_BASE = 9,  /* [n_env][nu] expert joint pose handed to the PD controller */
    UHC_E_CONSUMED = 10,    /* int32 [n_env] 1 where the last uhc_env_auto_reset() started the queued window, 0 elsewhere */
    UHC_E_EPISODE = 11,     /* [2][n_env] running episode length and return (reward + end * end_reward), kept by step / auto_reset */
    UHC_E_SNAPSHOT = 12     /* [5][n_env] written by uhc_env_auto_reset: done, episode length, episode return, percent, consumed */
};

int32_t uhc_env_create(UhcBatch* b, const UhcEnvDesc* desc, UhcEnv** out);
void uhc_env_free(UhcEnv* e);
int32_t uhc_env_obs_dim(const UhcEnv* e);
int32_t uhc_env_field(UhcEnv* e, int32_t field, void** d_ptr, int64_t* count);
/* clip bank (device pointers, borrowed: must outlive the env or the next set_bank):
 * d_frames [n_frames][UHC_FRAME_STRIDE], d_clip_start int32 [n_clips], d_clip_beta [n_clips][17] */
int32_t uhc_env_set_bank(UhcEnv* e, const double* d_frames, int64_t n_frames, const int32_t* d_clip_start,
                         const double* d_clip_beta, int32_t n_clips);
/* expert["obj_pose"] of every frame of the bank (device pointer, borrowed): d_obj_pose [n_frames][7 num_obj], object k at columns
 * 7k .. 7k + 6 (position, quaternion wxyz); reset / auto_reset read the row of the window's first frame (reset_model, humanoid_im.py:
 * 1284-1287: init_pose = concat(expert pose, obj_pose[0]), init_vel = concat(expert velocity, zeros)).  Call after uhc_env_set_bank. */
int32_t uhc_env_set_obj_pose(UhcEnv* e, const double* d_obj_pose, int64_t n_frames);
/* Per-clip body shape (the reference rebuilds the MuJoCo model from the clip's beta in load_expert -> reset_robot,
 * humanoid_im.py:154-180,204): d_clip_model int32 [n_clips] (borrowed, NULL to switch off) names, for every clip of the
 * bank, which of the batch's models (uhc_batch_create) its episodes run on; assign / auto_reset switch the env's model. */
int32_t uhc_env_set_clip_models(UhcEnv* e, const int32_t* d_clip_model);
/* load_expert (humanoid_im.py:182-215): env d_env_ids[i] tracks frames [fr_start, fr_start+fr_len) of clip d_clip_ids[i] */
int32_t uhc_env_assign(UhcEnv* e, const int32_t* d_env_ids, int32_t n, const int32_t* d_clip_ids,
                       const int32_t* d_fr_start, const int32_t* d_fr_len);
/* Episode turnover without a host round trip (the reference's sampler does `load_expert; reset` on the worker, agent_copycat.py:
 * 526-531, right after `done`): uhc_env_set_next queues, per env, the window its NEXT episode will track (+ the reset noise,
 * d_noise [n][nu] or NULL); uhc_env_auto_reset then does, for every env whose done flag is set, load_expert + reset_model on
 * the device (queued window if there is one, else the current window again) and refreshes its observation row.  The done
 * flags are left as the step wrote them; UHC_E_CONSUMED tells which envs took their queued window.  Of the reset's sim.forward()
 * only the kinematics run here (UHC_F_XPOS / XQUAT / XIPOS, the observation); its dynamics half (qM, qfrc_bias, qacc of the reset
 * state) runs at the head of the env's next uhc_env_step, so UHC_F_QM / QFRC_BIAS / QACC / NCON / NEFC of a restarted env are those
 * of its previous episode until then.  The steps that follow are bit-identical to those after uhc_env_assign + uhc_env_reset. */
int32_t uhc_env_set_next(UhcEnv* e, const int32_t* d_env_ids, int32_t n, const int32_t* d_clip_ids, const int32_t* d_fr_start,
                         const int32_t* d_fr_len, const double* d_noise);
int32_t uhc_env_auto_reset(UhcEnv* e);
/* bonus added to the reward of a step that ends its clip (Agent: `if end_reward and info["end"]: reward += env.end_reward`,
 * uhc/khrylib/rl/agents/agent.py:84-85); UHC_E_REWARD stays the plain imitation reward, the episode return includes the bonus */
int32_t uhc_env_set_end_reward(UhcEnv* e, double end_reward);
/* MujocoEnv.reset -> reset_model (humanoid_im.py:1245-1299): state <- expert frame 0 (+ d_noise [n][nu] on the
 * joint angles, may be NULL), forward pass, observation written to the obs rows of those envs */
int32_t uhc_env_reset(UhcEnv* e, const int32_t* d_env_ids, int32_t n, const double* d_noise);
/* HumanoidEnv.step for every env with d_active != 0 (NULL = all): PD target gather, do_simulation,
 * cur_t += 1, termination, reward, next observation */
int32_t uhc_env_step(UhcEnv* e, const double* d_action, const int32_t* d_active);

/* ------------------------------------------------------------------ rollout bookkeeping (one control step of the sampling loop)
 * The reference's sample_worker (uhc/khrylib/rl/agents/agent.py:60-100) does, per env step on the host: running_state(state),
 * select_action, memory.push, reward / mask bookkeeping.  Batched on the device these are launch-latency bound (~80 framework launches
 * per step); the entries below do them in a handful.  Free functions: all pointers are device pointers, `stream` is a hipStream_t
 * (NULL = the default stream), every reduction has a fixed order (bit-reproducible).  Layouts: states [n_env][T][obs_dim], actions
 * [n_env][T][act_dim], rewards / dones [n_env][T], mean_flags [T][n_env] (1.0 = take the mean action), d_t = the pass's step counter. */

/* memory.push(state, action, ...) + PolicyGaussian.select_action's sampling (policy_gaussian.py:22-31, distributions.py:6-25):
 * states[:, t] = state; action = mean where mean_flags[t] != 0 else mean + exp(log_std) * noise; actions[:, t] = action = d_action */
int32_t uhc_rollout_act(void* stream, int32_t n_env, int32_t T, const int64_t* d_t, int32_t obs_dim, int32_t act_dim, const double* d_state,
                        const double* d_mean, const double* d_log_std, const double* d_noise, const double* d_mean_flags, double* d_states,
                        double* d_actions, double* d_action);
/* agent.py:80-92: rewards[:, t] = reward + end * end_reward, dones[:, t] = done, c_reward_sum += sum(reward),
 * c_info_sum[k] += sum(parts[:, k]) (LoggerRL.step, logger_rl.py:29-33); n_parts <= 8.  d_redo (may be NULL): UHC_F_REDO of the step;
 * d_redo_counts (int64 [7]): [0] += envs the general / large tier or tier 4 computed, [1] += envs whose exact contact solve fell back to
 * sweeps, [2] += envs that lost constraint rows beyond the last tier's capacity in this step, [3] += envs the large tier or tier 4 computed,
 * [4] += envs with an island of more than 64 force-carrying rows solved exactly in windows of 64 rows (three-tier batches), [5] += envs
 * solved by Newton on the primal in tier 4, [6] += envs in which that iteration stopped at its cap (diagnostics) */
int32_t uhc_rollout_record(void* stream, int32_t n_env, int32_t T, const int64_t* d_t, const double* d_reward, const int32_t* d_done,
                           const int32_t* d_end, const double* d_end_reward, const double* d_parts, int32_t parts_stride, int32_t n_parts,
                           double* d_rewards, double* d_dones, double* d_c_reward_sum, double* d_c_info_sum, const int32_t* d_redo,
                           int64_t* d_redo_counts);
/* RunningStat.push for a batch (zfilter.py:17-27; Chan et al. merge, mathematically the rows pushed one by one): (n, mean, S) <-
 * merged with the rows of d_x [n_rows][dim] whose d_weights entry is non-zero (NULL = all rows).  d_scratch: uhc_filter_scratch_doubles */
int64_t uhc_filter_scratch_doubles(int32_t n_rows, int32_t dim);
int32_t uhc_filter_push(void* stream, const double* d_x, int32_t n_rows, int32_t dim, const int32_t* d_weights, double* d_n, double* d_mean,
                        double* d_S, double* d_scratch);
/* ZFilter.__call__ (zfilter.py:55-64): out = clip((x - mean) / (std + 1e-8), +-clip); std = sqrt(S / (n - 1)) (n <= 1: |mean|);
 * clip == 0: no clipping.  d_t_inc (may be NULL): incremented by one -- the step counter, this being the last launch of a step */
int32_t uhc_filter_apply(void* stream, const double* d_x, int32_t n_rows, int32_t dim, const double* d_n, const double* d_mean, const double* d_S,
                         int32_t demean, int32_t destd, double clip, double* d_out, int64_t* d_t_inc);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* UHC_AMD_H */
