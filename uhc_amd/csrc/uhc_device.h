// uhc_device.h -- device-side model description shared by the HIP kernels and the C-ABI host code.
// gfx950 only.  One environment per 64-lane wavefront; all per-env working state lives in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define UHC_WAVE 64
// capacities per env of the kernel tiers (fast / general / large / huge): constraint rows, contacts, rows between two moving bodies (kept
// as dense dof vectors).  The reference's models ask MuJoCo for njmax 2500 / nconmax 500 (uhc/khrylib/mocap/skeleton_mesh.py:46); what
// exceeds the last tier is dropped and flagged (UHC_F_EFC_OVERFLOW).  The first three tiers solve the DUAL problem (Delassus blocks of <= 64
// rows in registers) and keep their rows in LDS; the fourth -- rows in HBM / L2, the LDS for an nv x nv Hessian -- solves the PRIMAL problem by
// Newton's method, as the reference's MuJoCo does, at a cost that does not depend on the row count (uhc_primal.h).
#define UHC_FAST_MAXEFC 64   // one row per lane, Delassus matrix in registers
#define UHC_FAST_MAXCON 16
#define UHC_FAST_MAXCON_DENSE 24  // models with body-body contacts (1 row each at condim 1) reach a 16-contact cap long before the row cap: on the
                                 // self-colliding rollout 98 % of the fast tier's hand-ons named the contacts (tools/tier_trace.py, UHC_F_HANDON_WHY);
                                 // 24 leaves a third of them (profiles/r04_ab_layout.txt; 32 buys nothing more and costs packed-row storage)
#define UHC_FAST_MAXTWO 12
#define UHC_GEN_MAXEFC 128   // two rows per lane, working sets of <= 64 rows
#define UHC_GEN_MAXCON 64
#define UHC_GEN_MAXTWO 20
#define UHC_BIG_MAXEFC 256   // four rows per lane
#define UHC_BIG_MAXCON 128
#define UHC_BIG_MAXTWO 32
#define UHC_MAXTWO 32        // dense-row slots of the largest of the first three tiers (sizes the slot tables of their layouts)
#define UHC_HUGE_MAXEFC 1024 // tier 4: at most 16 rows per lane; the layout takes the largest multiple of 128 its LDS holds beside the Hessian
#define UHC_HUGE_MAXCON 192  // (a humanoid of 24 hulls lying among four boxes: <= 4 floor contacts per hull = 112, + body-body contacts)
#define UHC_HUGE_MAXCON_MOST 320  // ... and up to this many where the LDS has room left beside the rows and the Hessian (uhc_capi.cpp)
#define UHC_HUGE_MAXTWO 128
#define UHC_DOF_MAXACT 4
#define UHC_CON_STRIDE 24
// why a tier could not hold an env (bits 16+ of the forward pass's overflow word; UHC_F_HANDON_WHY = the word >> 16 of the env's last hand-on)
#define UHC_WHY_CONTACTS (1 << 16)
#define UHC_WHY_ROWS (1 << 17)
#define UHC_WHY_DENSE_SLOTS (1 << 18)
#define UHC_WHY_ROW_STORAGE (1 << 19)
#define UHC_WHY_CANDIDATES (1 << 20)
#define UHC_WHY_SOLVER (1 << 21)  // the working sets of the general / large tier did not finish (more than 64 force-carrying rows in an island, no convergence): Newton on the primal takes over
#define UHC_MINVAL 1e-15
#define UHC_MAXVAL 1e10

// topology + options: identical for every env of a batch
struct DevTopo {
    int nq, nv, nu, nbody, njnt, ngeom, nM, maxdepth, nmeshvert, npair, iterations, plane_mesh_maxcon, solver;
    double timestep, tolerance;
    double gravity[3];
    const int *body_parentid, *body_jntadr, *body_jntnum, *body_dofadr, *body_dofnum, *body_rootid, *body_nsub,
        *body_lastdof, *body_depth;
    const int *jnt_type, *jnt_bodyid, *jnt_qposadr, *jnt_dofadr, *jnt_limited;
    const int *dof_bodyid, *dof_jntid, *dof_parentid, *dof_madr, *dof_depth, *dof_ndesc;
    const short* dof_anc;  // [nv][maxdepth+1]: ancestor of dof i at depth q (q <= depth(i)), anc[i][depth(i)] = i
    const short *m_row, *m_col;  // [nM] sparse-M entry -> (i, j)
    const unsigned short* m_ij;  // [nM (+pad)] the same, packed i << 8 | j (nv <= 128)
    const unsigned char* dof_ncommon;  // [nv][nv] number of common chain entries of two dofs (depth of LCA + 1, 0 if none)
    const int *geom_type, *geom_bodyid, *geom_condim, *geom_vertadr, *geom_vertnum;
    const int *pair_g1, *pair_g2;  // statically filtered candidate geom pairs (g1 = plane, g2 = mesh)
    const int *cpair_g1, *cpair_g2;  // statically filtered convex-convex candidate pairs (mesh, mesh), g1 < g2, in (g1, g2) order
    int ncpair;
    const int* dof_rootid;  // [nv] body id of the kinematic-tree root the dof belongs to (its cdof are taken about that tree's COM)
    const int* actuator_dofid;
    int body_maxdepth;
    // static schedules of the tree-sparse factorisation / substitutions (uhc_capi.cpp).  Entries are LDS byte addresses
    // inside the LD buffer, whose slots nM (always 0.0) and nM+1 (write-only dump) serve idle lanes.
    const unsigned int* fac_prog;  // [fac_nslot (+2)][64][6] groups of three update slots of one step: {af_q | ar_q << 16 (q = 0..2), ao0 | ao1 << 16, ao2 | norm << 16, D_k}:  LD[ao] -= (LD[af] / D_k) * LD[ar]
    const unsigned int* sol_back;  // [nv-1][64] step s <-> i = nv-1-s : address of L[i][j] for j = lane (low 16) and lane+64 (high 16)
    const unsigned int* sol_fwd;   // [nv-1][64] step s <-> j = s      : address of L[i][j] for i = lane (low 16) and lane+64 (high 16)
    const unsigned int* chain;     // [nv][32] position q on the chain of dof i: (q-th dof from the root | LDS byte address of its L row << 16)
    int fac_nslot;                 // number of groups
    const int* dof_act;            // [nv][UHC_DOF_MAXACT] motors acting on the joint of each dof (-1 = none)
    int has_damping;               // some model of the batch has dof_damping > 0: mj_Euler integrates the damping implicitly
};

// offsets (in doubles) of the numeric arrays inside one model blob
struct DevNumOff {
    int body_pos, body_quat, body_ipos, body_iquat, body_mass, body_inertia, body_invweight0;
    int jnt_pos, jnt_axis, jnt_range, jnt_stiffness, jnt_margin, qpos0, qpos_spring;
    int dof_armature, dof_damping, dof_frictionloss, dof_invweight0;
    int geom_pos, geom_quat, geom_friction, geom_margin, geom_gap, geom_solref, geom_solimp, geom_rbound, geom_center, geom_box;
    int geom_radius;  // [ngeom] inflation radius of a ROUNDED hull (sphere, capsule: geom_size[0]), 0 for a mesh (include/uhc_amd.h)
    int mesh_vert, actuator_gear, meaninertia;
    int mesh_adj;  // int32 [nmeshvert][adjdeg] neighbours of every hull vertex (global vertex ids, -1 = none), packed two per double
    int stride;  // doubles per model
};

// LDS carve (offsets in doubles from the dynamic-LDS base)
struct DevLds {
    int qpos, qvel, qacc, ctrl, applied;
    int xpos, xquat, xmat, xipos, ximat, rootcom, cinert, crb, cvel, cacc, cfrc;
    int xanchor, xaxis, cdof, cdofdot;
    int M, LD, dinv, sdinv, bias, smooth, vec, z, zero;
    int mij;  // 16-bit (row << 8 | col) of every sparse-M entry, loaded once per kernel (k_crb)
    int con, Y, rowR, rowAref, rowB, rowF, rowDa, rowMisc /* ints: type,last,len,yoff */, ncon_nefc;
    int rowW;   // general / large tiers: the warm-start forces, kept for the sweeps fallback of the working-set solve
    int rowY;   // general / large tiers: ints [maxefc] offset of every row's packed Yhat entries inside Y
    int dense;  // [ndense][nvp] dense Yhat rows of the constraints that touch two moving bodies (self-collision, objects)
    int dcol;   // [ndense][64] column of the Delassus matrix of every dense row (A is symmetric: the row's lane reads it back); in the general /
                // large layouts it shares the contacts' storage, which nothing reads once the rows are built
    int dsc;    // general kernel: [ndense][4] J.qvel, J.qacc_smooth, J.qacc_warmstart, |Yhat|^2 of every dense row
    int H;      // tier 4: the Hessian of the primal problem, packed lower triangle column by column (nv (nv + 1) / 2 doubles)
    int total;  // doubles
};

struct DevCtrl {
    int n_substeps, action_type, meta_pd, rfc_mode, action_dim, n_vf_body, body_vf_dim;
    const int* vf_body;  // explicit RFC: body id per residual-force slot
    double rfc_scale, rfc_lim;
    double base_rot_inv[4];  // quaternion_inverse(base_rot) = conj / |q|^2
    const double *jkp, *jkd, *torque_lim, *a_scale;
};

struct DevState {  // HBM, env-major
    double *qpos, *qvel, *qacc, *qacc_ws, *xpos, *xquat, *xipos, *qM, *bias, *ctrl, *applied;
    double *cdof, *rootcom;  // explicit RFC only: kinematics of the last forward pass carried between launches
    int *ncon, *nefc, *fail, *solver_iter, *overflow, *redo;
    int *pend2, *pend3;  // envs handed on to the general / large tier this step and not yet taken (the chained launches' active masks)
    int* why;            // diagnostic (UHC_F_HANDON_WHY): bits 0-7 why the fast tier handed the env on in this step, bits 8-15 why the general tier did
                         // (1 contacts, 2 rows, 4 body-body row slots, 8 packed row storage, 16 MPR candidate list), bits 16+ the substep of the last hand-on
    int* resume;         // substep at which the tier below handed the env on (its state then is in qpos / qvel / qacc_ws / ctrl / applied); 0: from the start
    int* q_abort;  // consumers that gave up waiting for their producers (queue_claim)
    int* tier;   // per env: the tier that computed its last control step (minus hysteresis): where its next step starts (kernel path 2)
    const int* tier_now;  // snapshot of `tier` taken at the head of the step: what the tier filter of a launch reads
    int* cost;   // per env: how close its last step came to the fast tier's capacity, in sixty-fourths (orders the fast tier's launch)
    int* fresh;  // 1: the env was restarted on the device (set_state done, kinematics refreshed); its mj_forward runs at the head of its next step
    const int* env_model;
    long long* prof;  // [n_env][16] stage cycle accumulators (only written by -DUHC_STAGE_PROF builds)
    unsigned long long* path_stats;  // running counts of control steps (MODE 0): [0] envs the fast kernel handed on, [1] envs the general kernel
                                     // computed that would have fitted the fast one, [2] envs the general kernel computed (uhc_batch_set_kernel_path 2)
    const double* model_blob;
};

// Tier 4's Newton iteration (uhc_primal.h) re-uses the contacts' storage once the rows are built: offsets in doubles from DevLds::con, the same on the
// host (which sizes the region) and in every wave of the workgroup (the helper waves of the four-wave consumer build their own view of it).
#define UHC_PRIMAL_WAVES 4       // waves of a tier-4 queue consumer (uhc_k_huge_q.hip); the one-workgroup-per-env kernels run the same stages on one wave
#define UHC_PRIMAL_CLS_STRIDE 144  // entries of one ownership class of the pair table: sum_{q < 32} (q / 4 + 1)
struct PrimalScratch { int anc, stY, dstage, pair_all, pair_cls, pair_cnt, cw, mbx, part, runs, total; };
__host__ __device__ inline PrimalScratch primal_scratch(int nv, int YS) {
    PrimalScratch p;
    int o = 0;
    auto take = [&](int n) { const int at = o; o += (n + 1) & ~1; return at; };
    p.anc = take((nv * YS + 3) / 4 + 1);                  // shorts [nv][YS]: the dof chains
    p.stY = take(UHC_PRIMAL_WAVES * 16 * 32);             // per wave [16][32]: the run of chain rows being added
    p.dstage = take(nv * 8);                              // [nv][8]: D y of the dense rows being added, dof-major
    p.pair_all = take(528 / 4);                           // shorts [528]: pair p = q (q + 1) / 2 + q2 -> q << 8 | q2, q2 <= q < 32
    p.pair_cls = take(4 * UHC_PRIMAL_CLS_STRIDE / 4);     // shorts [4][144]: the same pairs by ownership class q2 & 3, q-major
    p.pair_cnt = take((4 * 33 + 3) / 4);                  // shorts [4][33]: pairs of a class with q < len
    p.cw = take(UHC_PRIMAL_WAVES * 32);                   // per wave [2][16]: coefficients and weights of the run
    p.mbx = take(4);                                      // ints [8]: the command the helper waves read behind the barrier
    p.part = take(UHC_PRIMAL_WAVES * 128);                // per wave [128]: partial sums of a scatter whose rows are dealt out to the waves
    p.runs = take(UHC_HUGE_MAXEFC / 2);                   // ints [maxefc]: the runs of chain rows (primal_run_table)
    p.total = o;
    return p;
}

// capacities of one tier's LDS layout
struct TierCap {
    int maxefc, maxcon;
    int ndense;    // dense-row slots (0: the model has no two-body contacts)
    int ycap;      // doubles available for the packed Yhat rows
    int vstage;    // LDS offset (doubles) where the hull vertices are staged for the MPR pass of every substep, or -1: they do not fit the
                   // region that is free at collision time (the not-yet-written constraint rows) and MPR reads them from L2
    int ld_delta;  // bytes to add to the schedule tables' LDS addresses (they are built for the fast layout's LD buffer)
};
struct KernelArgs {
    DevTopo t;
    DevNumOff o;
    // all three layouts are phase-aliased: a persistent part (state, factor, body poses, cdof) + one region shared by the dynamics
    // temporaries (first half of a forward pass) and the constraint data (second half); M itself is parked in registers
    DevLds lf;  // fast tier: 40 KiB (53 with dense-row slots): 4 (3) workgroups per CU
    DevLds l;   // general tier: <= 79 KiB: 2 workgroups per CU
    DevLds lh;  // large tier: <= 160 KiB
    DevLds lx;  // tier 4 (huge): <= 160 KiB -- persistent part, contacts, per-row scalars, the Hessian; Yhat rows and dense rows in HBM (gY, gD)
    TierCap cf, cg, ch, cx;
    double* gY;  // tier 4: [n_env][gy_stride] packed Yhat rows of the env (L2-resident while its workgroup runs)
    double* gD;  // tier 4: [n_env][gd_stride] dense Yhat rows (slot-major, nvp doubles each)
    int gy_stride, gd_stride;
    int last_tier;  // 2, 3 or 4: the tier that drops what exceeds it instead of handing the env on.  In the chained launches tier 4 has no launch of its own: the large
                    // tier's workgroup goes on with it (same LDS allocation, other carve) when its env does not fit or its working sets give up; under sticky tiers
                    // it has queue consumers (uhc_k_huge_q.hip) that take what the large tier's consumers hand on
    const int* order;  // launch order: workgroup k works on env order[k] (null: env k)
    int tier_want;  // 0: the launch works on every active env; else (sticky tiers) it leaves out the envs whose tier_now differs AND has its own launch
    int sticky_mask;  // bit t: tier t has its own (list) launch this step
    int* list;        // != NULL: persistent launch over this env queue (list_count entries so far, shared cursor; slots beyond hold -1)
    int* list_count;
    int* list_cursor;
    const int* prod_fin;  // the queue's producers: exit only when this counter has reached prod_total (NULL: the queue does not grow)
    int prod_total;
    int n_wait;           // queue consumers: so many workgroups that find the queue empty on entry stay and wait for the producers' hand-ons
    int* spares;          // ... the seats taken so far (NULL: every such workgroup waits)
    int* started;         // queue consumers: bumped once per workgroup on entry (the gate before the fast tier's launch waits for it), or NULL
    int* fin;             // this launch's own exit counter (bumped once per workgroup), or NULL
    int* q_next;          // hand-on target: the next tier's queue (NULL: flag the env in redo / redo2 for a chained launch)
    int* q_next_count;
    const int* guard_tab;  // debug (UHC_GUARD_LDS=1 at batch creation, a library built with -DUHC_GUARD_LDS): per tier 64 ints -- [0] / [1] number of guard
                           // words after the persistent / the constraint-phase LDS regions, [2 ..] / [32 ..] their offsets (doubles); NULL otherwise
    int* guard_hits;       // ... [0] guard words found overwritten, [1] the first one: tier << 16 | kind << 8 | index, [2] its env, [3] its offset
    int* cnt4;            // sticky tiers: bumped once per env the large tier hands on to tier 4 (the host sizes the tier-4 consumers of the next steps by it), or NULL
    int grid;  // workgroups of a list launch (0: one per env)
    int marks[8];  // sticky tiers: when an env starts its next step a tier up / down (uhc_step_env; UHC_TIER_MARKS)
    int t4_rows;   // sticky tiers: an env of the general / large tier whose step peaked at this many rows or more starts its NEXT step in tier 4, at the head of the
                   // four-wave consumers' queue, and stays there while it peaks above 3/4 of the mark (0: never; UHC_T4_ROWS)
    int truncate;  // fast kernel: drop contacts / rows beyond its capacity instead of handing the env to the general kernel
    int ball_limits;  // the model has limited ball joints: the fast tier launches its DENSE instantiation (which carries the ball-limit rows)
    int dbg;                 // debug switches (UHC_DEBUG env var): bit 0 = working sets never merge islands, bit 1 = MPR vertices not staged in LDS, bit 2 = a handed-on env restarts its step instead of resuming at the substep, bit 3 = sticky fast tier launches in env order, bit 4 = tier trace in the stage-profile record, bit 5 = no gate before the fast tier's launch, bit 6 = log the queue lengths and the gate's wait per step, bit 7 = no box cull of the convex pairs, bit 8 (256) = the general tier's first working set is NOT filled by rank (k_as_general; A/B of DESIGN 2), bits 9 / 10 (512 / 1024) = fill 48 / 56 lanes instead of 64, bit 11 (2048) = fixed cap UHC_Q2_MAX on the general tier's consumers (launch()).  bit 12 (4096) = sticky tier 4: an env whose step ended in tier 4 starts its next step there, at the head of the tier-4 consumers' queue (uhc_capi.cpp launch(); measured and not the default, see uhc_step_env).  Bits 8-12 change which envs report windows / sweeps in UHC_F_REDO: measurement switches that exist only in libraries built with -DUHC_EXPERIMENTS (tools/ A/B builds) -- the shipped library compiles their reads out (UHC_EXP in uhc_physics_impl.h) and masks the bits (uhc_capi.cpp); uhc_build_flags() bit 0 tells which kind a library is
    int nvp;                 // stride of a dense row (nv rounded up to 2 doubles)
    int adjdeg;              // stride of the per-model hull adjacency table (largest vertex degree over the batch's models)
    DevCtrl c;
    DevState s;
    int n_env;
};
