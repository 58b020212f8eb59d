#pragma once
// uhc_physics_impl.h -- fused rigid-body step for batched SMPL humanoids on MI355X (gfx950).
//
// One environment per 64-lane wavefront (workgroup = 1 wave), every per-env intermediate in LDS or
// registers, one launch per control step (n_substeps x {stable-PD, residual force, forward dynamics,
// PGS contact solve, semi-implicit Euler}).  HBM traffic per env-step is the state only
// (qpos, qvel, warm start, qM, qfrc_bias in; the same plus body poses out).
//
// Stage map: P1 k_kinematics, P2 k_com_pos, P3 k_crb + k_factor, P4 k_collision,
// P5 k_rows, P6/P9 k_pgs, P7 k_com_vel + k_rne, P8 k_smooth, P10 k_euler, E3/E4 k_pd_torque,
// E5 k_rfc_implicit.  The reference call sites these replace: uhc/envs/humanoid_im.py:1145-1190
// (do_simulation), :1014-1076 (PD), :1136-1143 (RFC), and MuJoCo's mj_step behind self.sim.step().
#include <type_traits>
#include "../../include/uhc_amd.h"
#include "uhc_device.h"

extern __shared__ __attribute__((aligned(16))) double smem[];

// Multi-wave workgroups (queue consumers only).  UHC_NW4 (uhc_k_huge_q.hip): four waves -- wave 0 runs everything below as if it were alone, waves 1-3 are helpers of
// tier 4's Newton iteration (uhc_primal.h).  UHC_NW2 (uhc_k_general_q.hip): two waves -- the helper serves half of the support requests of every MPR round
// (uhc_mpr.h: mpr_wave_mw): the general tier's slowest env is what a control step of the headline waits for, a quarter of its cycles are MPR, and with two
// 79 KiB consumers per CU two of the CU's four SIMDs idle.  In both, helpers see nothing else of this file.
#if defined(UHC_NW4)
#define UHC_NWG 4
#elif defined(UHC_NW2)
#define UHC_NWG 2
#else
#define UHC_NWG 1
#endif
#if UHC_NWG > 1
#define LANE ((int)(threadIdx.x & 63u))
#else
#define LANE ((int)threadIdx.x)
#endif
// optional per-stage cycle accounting (build with -DUHC_STAGE_PROF; see tools/stage_profile.py)
#define UHC_NPROF 40  // int64 words per env of the stage-profile record (UHC_F_STAGE_PROF)
#ifdef UHC_STAGE_PROF
#define PROF_DECL long long pt_[UHC_NPROF] = {}; long long pt_last_ = __builtin_readcyclecounter();
#define PROF_ARGS , long long* pt_, long long& pt_last_
#define PROF_PASS , pt_, pt_last_
#define PROF(i) { const long long now_ = __builtin_readcyclecounter(); pt_[i] += now_ - pt_last_; pt_last_ = now_; }
#else
#define PROF_DECL
#define PROF_ARGS
#define PROF_PASS
#define PROF(i)
#endif
// Measurement switches of the solver (UHC_DEBUG bits 8-10 and 12: working-set fill, sticky tier 4) exist only in libraries built with -DUHC_EXPERIMENTS
// (tools/ A/B builds); the shipped library compiles them out -- a stray environment variable cannot change which envs report windows / sweeps
// (uhc_build_flags() bit 0 says which kind a library is; tests/test_capi_symbols.py asserts the shipped one is not).
#ifdef UHC_EXPERIMENTS
#define UHC_EXP(bit) ((A.dbg & (bit)) != 0)
#else
#define UHC_EXP(bit) false
#endif
// Three tiers of one kernel: TIER 1 = fast (compact LDS, <= 64 rows, Delassus matrix in registers), TIER 2 = general (<= 128 rows, working
// sets, two workgroups per CU), TIER 3 = large (<= 256 rows, a whole CU's LDS).  A tier that cannot hold an env leaves it untouched and
// hands it to the next one; only the last tier of a batch (KernelArgs::last_tier) drops what exceeds it and flags the env.
template <int TIER> __device__ __forceinline__ const DevLds& lds_of(const KernelArgs& A) {
    if constexpr (TIER == 1) return A.lf; else if constexpr (TIER == 2) return A.l; else if constexpr (TIER == 3) return A.lh; else return A.lx;
}
template <int TIER> __device__ __forceinline__ const TierCap& cap_of(const KernelArgs& A) {
    if constexpr (TIER == 1) return A.cf; else if constexpr (TIER == 2) return A.cg; else if constexpr (TIER == 3) return A.ch; else return A.cx;
}
// does this tier hand an env it cannot hold to the next one (true), or drop the excess and flag it (false)?
template <int TIER> __device__ __forceinline__ bool hands_on(const KernelArgs& A) { return TIER == 1 ? !A.truncate : TIER < A.last_tier; }
// wsync: the LANES OF ONE WAVE have exchanged data through LDS (or global memory).  With one wave per workgroup that is __syncthreads() (whose s_barrier the
// backend drops); in a four-wave workgroup the same fences without the barrier -- the other waves are not coming
#if UHC_NWG > 1
__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
#else
__device__ __forceinline__ void wsync() { __syncthreads(); }
#endif
// where a tier keeps its constraint rows: LDS (tiers 1-3), or the env's slice of the HBM row store (tier 4: the LDS is the Hessian's, uhc_primal.h)
template <int TIER> __device__ __forceinline__ double* y_store(const KernelArgs& A, double* S, int env) {
    if constexpr (TIER == 4) return A.gY + (size_t)env * A.gy_stride; else return S + lds_of<TIER>(A).Y;
}
template <int TIER> __device__ __forceinline__ double* d_store(const KernelArgs& A, double* S, int env) {
    if constexpr (TIER == 4) return A.gD + (size_t)env * A.gd_stride; else return S + lds_of<TIER>(A).dense;
}

// ------------------------------------------------------------------ lane helpers
__device__ __forceinline__ double bcast(double v, int src) {  // src must be wave-uniform
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
// Cross-lane sums: four DPP steps (quad swaps, half-row and row mirrors: VALU-rate, no LDS crossbar round trip as
// ds_bpermute-based shuffles have) leave each 16-lane row's sum in all of its lanes; the four rows are combined
// through readlane.  ~5x cheaper than a shfl_xor butterfly for one wave per SIMD (tools/ubench/lat.hip).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
#define DPP_QUAD_1032 0xB1
#define DPP_QUAD_2301 0x4E
#define DPP_ROW_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR 0x140
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<DPP_QUAD_1032>(v);
    v += dpp_f64<DPP_QUAD_2301>(v);
    v += dpp_f64<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_f64<DPP_ROW_MIRROR>(v);
    return (bcast(v, 0) + bcast(v, 16)) + (bcast(v, 32) + bcast(v, 48));
}
__device__ __forceinline__ double wave_min(double v) {
    v = fmin(v, dpp_f64<DPP_QUAD_1032>(v));
    v = fmin(v, dpp_f64<DPP_QUAD_2301>(v));
    v = fmin(v, dpp_f64<DPP_ROW_HALF_MIRROR>(v));
    v = fmin(v, dpp_f64<DPP_ROW_MIRROR>(v));
    return fmin(fmin(bcast(v, 0), bcast(v, 16)), fmin(bcast(v, 32), bcast(v, 48)));
}
__device__ __forceinline__ int wave_or(int v) { return __builtin_amdgcn_ballot_w64(v != 0) != 0; }  // used as "any lane set"
template <int TIER> __device__ __forceinline__ void k_row_one(const KernelArgs& A, const double* mb, double* S, const int r, double* Yb);  // (the helper wave of a two-wave consumer builds rows too: uhc_mpr.h mpr_helper)
struct LaneConst { int d0, d1, n0, n1, m0, m1, pk0, pk1, r0, r1; bool v0, v1; };  // r0, r1: tree-root body of the lane's dofs (dense rows only)
template <int TIER> __device__ __forceinline__ void k_dense_groups(const KernelArgs& A, double* S, const LaneConst& LC, double* Db, const int ntwo, const int first, const int stride);
template <bool DENSE> __device__ __forceinline__ LaneConst lane_const(const DevTopo& T);
#include "uhc_mpr.h"  // (uses the lane helpers above)

// ------------------------------------------------------------------ small math (registers)
__device__ __forceinline__ void cross3(double* r, const double* a, const double* b) {
    double r0 = a[1] * b[2] - a[2] * b[1], r1 = a[2] * b[0] - a[0] * b[2], r2 = a[0] * b[1] - a[1] * b[0];
    r[0] = r0; r[1] = r1; r[2] = r2;
}
__device__ __forceinline__ double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void quat_mul(double* r, const double* a, const double* b) {
    double t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    double t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    double t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    double t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
}
__device__ __forceinline__ void quat_normalize(double* q) {
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n < UHC_MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
    double inv = 1.0 / n;
    q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
}
__device__ __forceinline__ void quat_to_mat(double* m, const double* q) {
    double w = q[0], x = q[1], y = q[2], z = q[3];
    m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
    m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
    m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
__device__ __forceinline__ void mat_vec(double* r, const double* m, const double* v) {
    double t0 = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
    double t1 = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
    double t2 = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
    r[0] = t0; r[1] = t1; r[2] = t2;
}
__device__ __forceinline__ void axis_angle_quat(double* q, const double* axis, double angle) {
    double s, c;
    sincos(0.5 * angle, &s, &c);
    q[0] = c; q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
__device__ __forceinline__ void cross_motion(double* r, const double* v, const double* m) {
    double a[3], b[3];
    cross3(r, v, m);
    cross3(a, v, m + 3);
    cross3(b, v + 3, m);
    r[3] = a[0] + b[0]; r[4] = a[1] + b[1]; r[5] = a[2] + b[2];
}
__device__ __forceinline__ void cross_force(double* r, const double* v, const double* f) {
    double a[3], b[3];
    cross3(a, v, f);
    cross3(b, v + 3, f + 3);
    r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2];
    cross3(r + 3, v, f + 3);
}
__device__ __forceinline__ void inert_mul(double* r, const double* I, const double* v) {
    const double *w = v, *l = v + 3, *h = I + 6;
    double hxl[3], wxh[3];
    cross3(hxl, h, l);
    cross3(wxh, w, h);
    r[0] = I[0] * w[0] + I[3] * w[1] + I[4] * w[2] + hxl[0];
    r[1] = I[3] * w[0] + I[1] * w[1] + I[5] * w[2] + hxl[1];
    r[2] = I[4] * w[0] + I[5] * w[1] + I[2] * w[2] + hxl[2];
    r[3] = I[9] * l[0] + wxh[0];
    r[4] = I[9] * l[1] + wxh[1];
    r[5] = I[9] * l[2] + wxh[2];
}
__device__ __forceinline__ double dot6(const double* a, const double* b) {
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
__device__ __forceinline__ double clampd(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }
__device__ __forceinline__ double max0(double x) {  // max(x, 0) without the canonicalising self-max clang adds to fmax
    double r;
    asm("v_max_f64 %0, %1, 0" : "=v"(r) : "v"(x));
    return r;
}

// compile-time loop: register arrays indexed by the loop variable stay in VGPRs
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// FAST variant: the tree-sparse mass matrix of the last forward pass lives in registers between substeps
// (entry e = LANE + 64 m), because the PD controller of the NEXT substep needs it (humanoid_im.py:1019-1022)
// and the 40 KiB LDS budget only holds its factor.
// Explicit parking of values in the accumulation registers (AGPRs).  Only 256 of the 512 registers of a full-file
// wave are directly addressable by VALU instructions; what the compiler cannot fit it shuttles through AGPRs (or
// scratch) at its own discretion.  Parking long-lived data here by hand keeps the hot loops inside the VGPRs: the
// finished Delassus row while it is being built, and the mass matrix between substeps (MPark).
__device__ __forceinline__ void agpr_put(int& a, int v) { asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v)); }
__device__ __forceinline__ int agpr_get(int a) { int v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a)); return v; }
#define UHC_MREG 24
// the joint-space inertia of the last forward pass (entries LANE + 64 m), parked in AGPRs: the PD solve of the next substep
// reads it back (the fast layout factorises M in place, so LDS does not keep it)
struct MPark { int lo[UHC_MREG], hi[UHC_MREG]; };

// Per-lane body constants, loaded once per kernel (lane = body): the level-synchronous tree passes of every substep
// would otherwise fetch them from global tables inside their per-level branches and wait for L2 at each of the 9 levels.
// jt packs the joint types of the body's joints, 2 bits each (<= 8 joints per body, checked by the host).
struct BodyConst { int p, ja, jn, depth, da, dn, nsub, root, jt; bool act; };
__device__ __forceinline__ BodyConst body_const(const DevTopo& T) {
    BodyConst c;
    const int b = LANE;
    c.act = b > 0 && b < T.nbody;
    c.p = c.act ? T.body_parentid[b] : 0;
    c.ja = c.act ? T.body_jntadr[b] : 0;
    c.jn = c.act ? T.body_jntnum[b] : 0;
    c.depth = c.act ? T.body_depth[b] : -1;
    c.da = c.act ? T.body_dofadr[b] : 0;
    c.dn = c.act ? T.body_dofnum[b] : 0;
    c.nsub = c.act ? T.body_nsub[b] : 0;
    c.root = c.act ? T.body_rootid[b] : -1;
    c.jt = 0;
    for (int q = 0; q < c.jn && q < 8; q++) c.jt |= T.jnt_type[c.ja + q] << (2 * q);
    return c;
}
__device__ __forceinline__ int jt_of(const BodyConst& c, int q) { return (c.jt >> (2 * q)) & 3; }
__device__ __forceinline__ int jdofs(int jt) { return jt == UHC_JNT_FREE ? 6 : jt == UHC_JNT_BALL ? 3 : 1; }

// ------------------------------------------------------------------ P1 kinematics
// Pass 1 (all bodies in parallel): pose of each body relative to its parent frame, including its own
// joint rotations (the expensive sincos work).  Pass 2 (level-synchronous): compose with the parent.
// Pass 3 (all joints in parallel): joint anchors/axes to the world frame.
template <int TIER>
__device__ __forceinline__ void k_kinematics(const KernelArgs& A, const double* mb, double* S, const BodyConst& BC PROF_ARGS) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    const int b = LANE;
    double *xpos = S + L.xpos, *xquat = S + L.xquat, *xmat = S + L.xmat, *xipos = S + L.xipos, *ximat = S + L.ximat;
    if (b == 0) {
        for (int k = 0; k < 3; k++) { xpos[k] = 0; xipos[k] = 0; }
        xquat[0] = 1; xquat[1] = xquat[2] = xquat[3] = 0;
        for (int k = 0; k < 9; k++) { xmat[k] = (k % 4 == 0); ximat[k] = (k % 4 == 0); }
    }
    const bool act = BC.act;
    const int depth = BC.depth;
    double lpos[3] = {0, 0, 0}, lquat[4] = {1, 0, 0, 0}, ip[3] = {0, 0, 0}, iq[4] = {1, 0, 0, 0};
    bool is_free = false;
    const int p = BC.p, ja = BC.ja, jn = BC.jn;
    if (act) {
        is_free = jn == 1 && jt_of(BC, 0) == UHC_JNT_FREE;
        for (int k = 0; k < 3; k++) ip[k] = mb[A.o.body_ipos + 3 * b + k];
        for (int k = 0; k < 4; k++) iq[k] = mb[A.o.body_iquat + 4 * b + k];
        double R[9], t[3];
        if (is_free) {
            const double* q = S + L.qpos + T.jnt_qposadr[ja];
            for (int k = 0; k < 3; k++) lpos[k] = q[k];
            for (int k = 0; k < 4; k++) lquat[k] = q[3 + k];
            quat_normalize(lquat);
            for (int k = 0; k < 3; k++) { S[L.xanchor + 3 * ja + k] = lpos[k]; S[L.xaxis + 3 * ja + k] = (k == 2); }
        } else {
            for (int k = 0; k < 3; k++) lpos[k] = mb[A.o.body_pos + 3 * b + k];
            for (int k = 0; k < 4; k++) lquat[k] = mb[A.o.body_quat + 4 * b + k];
            for (int j = ja; j < ja + jn; j++) {
                const int qa = T.jnt_qposadr[j], jt = jt_of(BC, j - ja);
                double qloc[4], jp[3], jx[3], ax[3];
                for (int k = 0; k < 3; k++) { jp[k] = mb[A.o.jnt_pos + 3 * j + k]; jx[k] = mb[A.o.jnt_axis + 3 * j + k]; }
                quat_to_mat(R, lquat);
                mat_vec(t, R, jp);
                double anc[3];
                for (int k = 0; k < 3; k++) { anc[k] = lpos[k] + t[k]; S[L.xanchor + 3 * j + k] = anc[k]; }  // parent frame for now
                mat_vec(ax, R, jx);
                for (int k = 0; k < 3; k++) S[L.xaxis + 3 * j + k] = ax[k];
                if (jt == UHC_JNT_SLIDE) {
                    const double dq = S[L.qpos + qa] - mb[A.o.qpos0 + qa];
                    for (int k = 0; k < 3; k++) lpos[k] += ax[k] * dq;
                    continue;
                } else if (jt == UHC_JNT_HINGE) {
                    axis_angle_quat(qloc, jx, S[L.qpos + qa] - mb[A.o.qpos0 + qa]);
                    quat_mul(lquat, lquat, qloc);
                } else if (jt == UHC_JNT_BALL) {
                    for (int k = 0; k < 4; k++) qloc[k] = S[L.qpos + qa + k];
                    quat_normalize(qloc);
                    quat_mul(lquat, lquat, qloc);
                }
                quat_to_mat(R, lquat);
                mat_vec(t, R, jp);
                for (int k = 0; k < 3; k++) lpos[k] = anc[k] - t[k];
            }
        }
    }
    wsync();
    PROF(19)
    for (int level = 1; level <= T.body_maxdepth; level++) {
        if (depth == level) {
            double pos[3], quat[4], R[9], t[3];
            if (is_free) {
                for (int k = 0; k < 3; k++) pos[k] = lpos[k];
                for (int k = 0; k < 4; k++) quat[k] = lquat[k];
            } else {
                double pm[9], pq[4];
                for (int k = 0; k < 9; k++) pm[k] = xmat[9 * p + k];
                for (int k = 0; k < 4; k++) pq[k] = xquat[4 * p + k];
                mat_vec(t, pm, lpos);
                for (int k = 0; k < 3; k++) pos[k] = xpos[3 * p + k] + t[k];
                quat_mul(quat, pq, lquat);
                quat_normalize(quat);
            }
            quat_to_mat(R, quat);
            double qi[4], Ri[9];
            mat_vec(t, R, ip);
            quat_mul(qi, quat, iq);
            quat_to_mat(Ri, qi);
            for (int k = 0; k < 3; k++) { xpos[3 * b + k] = pos[k]; xipos[3 * b + k] = pos[k] + t[k]; }
            for (int k = 0; k < 4; k++) xquat[4 * b + k] = quat[k];
            for (int k = 0; k < 9; k++) { xmat[9 * b + k] = R[k]; ximat[9 * b + k] = Ri[k]; }
        }
        wsync();
    }
    PROF(20)
    // joint anchors / axes: parent frame -> world
    for (int j = LANE; j < T.njnt; j += UHC_WAVE) {
        if (T.jnt_type[j] == UHC_JNT_FREE) continue;
        const int pb = T.body_parentid[T.jnt_bodyid[j]];
        double pm[9], a[3], x[3], t[3];
        for (int k = 0; k < 9; k++) pm[k] = xmat[9 * pb + k];
        for (int k = 0; k < 3; k++) { a[k] = S[L.xanchor + 3 * j + k]; x[k] = S[L.xaxis + 3 * j + k]; }
        mat_vec(t, pm, a);
        for (int k = 0; k < 3; k++) S[L.xanchor + 3 * j + k] = xpos[3 * pb + k] + t[k];
        mat_vec(t, pm, x);
        for (int k = 0; k < 3; k++) S[L.xaxis + 3 * j + k] = t[k];
    }
    wsync();
}

// ------------------------------------------------------------------ P2 comPos: tree COM, cinert (body/lane), cdof (joint/lane)
template <int TIER>
__device__ __forceinline__ void k_com_pos(const KernelArgs& A, const double* mb, double* S, const BodyConst& BC) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    const int b = LANE;
    const bool act = BC.act;
    const double mass = act ? mb[A.o.body_mass + b] : 0.0;
    const int root = BC.root;
    double xi[3] = {0, 0, 0};
    if (act) for (int k = 0; k < 3; k++) xi[k] = S[L.xipos + 3 * b + k];
    // one reduction per kinematic tree (bodies whose parent is the world)
    for (int r = 1; r < T.nbody; r++) {
        if (T.body_parentid[r] != 0) continue;
        const double w = root == r ? mass : 0.0;
        double m = wave_sum(w), cx = wave_sum(w * xi[0]), cy = wave_sum(w * xi[1]), cz = wave_sum(w * xi[2]);
        if (LANE == 0) {
            const double inv = m < UHC_MINVAL ? 0.0 : 1.0 / m;
            S[L.rootcom + 3 * r] = cx * inv; S[L.rootcom + 3 * r + 1] = cy * inv; S[L.rootcom + 3 * r + 2] = cz * inv;
        }
    }
    wsync();
    if (b == 0) for (int k = 0; k < 10; k++) S[L.cinert + k] = 0;
    if (act) {
        double c[3], R[9], I[3], J[9];
        for (int k = 0; k < 3; k++) { c[k] = xi[k] - S[L.rootcom + 3 * root + k]; I[k] = mb[A.o.body_inertia + 3 * b + k]; }
        for (int k = 0; k < 9; k++) R[k] = S[L.ximat + 9 * b + k];
        for (int r = 0; r < 3; r++)
            for (int s = 0; s < 3; s++)
                J[3 * r + s] = R[3 * r] * I[0] * R[3 * s] + R[3 * r + 1] * I[1] * R[3 * s + 1] + R[3 * r + 2] * I[2] * R[3 * s + 2];
        const double cc = dot3(c, c);
        double* ci = S + L.cinert + 10 * b;
        ci[0] = J[0] + mass * (cc - c[0] * c[0]);
        ci[1] = J[4] + mass * (cc - c[1] * c[1]);
        ci[2] = J[8] + mass * (cc - c[2] * c[2]);
        ci[3] = J[1] - mass * c[0] * c[1];
        ci[4] = J[2] - mass * c[0] * c[2];
        ci[5] = J[5] - mass * c[1] * c[2];
        ci[6] = mass * c[0]; ci[7] = mass * c[1]; ci[8] = mass * c[2]; ci[9] = mass;
    }
    for (int j = LANE; j < T.njnt; j += UHC_WAVE) {
        const int bj = T.jnt_bodyid[j], da = T.jnt_dofadr[j], jt = T.jnt_type[j];
        const int rj = T.body_rootid[bj];
        double off[3], ax[3];
        for (int k = 0; k < 3; k++) { off[k] = S[L.rootcom + 3 * rj + k] - S[L.xanchor + 3 * j + k]; ax[k] = S[L.xaxis + 3 * j + k]; }
        double* cd = S + L.cdof + 6 * da;
        if (jt == UHC_JNT_FREE) {
            for (int k = 0; k < 18; k++) cd[k] = 0;
            cd[3] = 1; cd[6 + 4] = 1; cd[12 + 5] = 1;
            cd += 18;
        }
        if (jt == UHC_JNT_FREE || jt == UHC_JNT_BALL) {
            for (int k = 0; k < 3; k++) {
                double a3[3] = {S[L.xmat + 9 * bj + k], S[L.xmat + 9 * bj + 3 + k], S[L.xmat + 9 * bj + 6 + k]}, cr[3];
                cross3(cr, a3, off);
                for (int s = 0; s < 3; s++) { cd[6 * k + s] = a3[s]; cd[6 * k + 3 + s] = cr[s]; }
            }
        } else if (jt == UHC_JNT_SLIDE) {
            for (int s = 0; s < 3; s++) { cd[s] = 0; cd[3 + s] = ax[s]; }
        } else {
            double cr[3];
            cross3(cr, ax, off);
            for (int s = 0; s < 3; s++) { cd[s] = ax[s]; cd[3 + s] = cr[s]; }
        }
    }
    wsync();
}

// ------------------------------------------------------------------ P3 composite inertias + sparse M
template <int TIER, bool DENSE>
__device__ __forceinline__ void k_crb(const KernelArgs& A, const double* mb, double* S, MPark& MP, const BodyConst& BC PROF_ARGS) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    const int b = LANE;
    // composite inertia = sum of cinert over the subtree, which is the contiguous range [b, b + nsub) in DFS order.
    // One work item per (body, component): consecutive lanes read consecutive components, the loop length is the subtree
    // size of the item's body (a body-per-lane loop made the root lane walk 24 x 10 dependent LDS loads alone).
    for (int item = LANE; item < 10 * T.nbody; item += UHC_WAVE) {
        const int bb = item / 10, k = item - 10 * bb;
        double acc = 0.0;
        if (bb > 0) {
            const int n = T.body_nsub[bb];
            for (int c = bb + n - 1; c >= bb; c--) acc += S[L.cinert + 10 * c + k];
        }
        S[L.crb + item] = acc;
    }
    wsync();
    PROF(21)
    // buf[i] = crb[body(i)] * cdof[i]   (kept in the cdofdot area until k_com_vel overwrites it)
    for (int i = LANE; i < T.nv; i += UHC_WAVE) {
        double I[10], v[6], r[6];
        const int bi = T.dof_bodyid[i];
        for (int k = 0; k < 10; k++) I[k] = S[L.crb + 10 * bi + k];
        for (int k = 0; k < 6; k++) v[k] = S[L.cdof + 6 * i + k];
        inert_mul(r, I, v);
        for (int k = 0; k < 6; k++) S[L.cdofdot + 6 * i + k] = r[k];
    }
    wsync();
    PROF(22)
#pragma unroll
    for (int m = 0; m < UHC_MREG; m++) {
        const int e = LANE + UHC_WAVE * m;
        double v = 0.0;
        if (e < T.nM) {
            // (the larger tiers, and the fast tier of models with body-body contacts, keep their LDS for rows: the table comes from L2)
            const int ij = (TIER == 1 && !DENSE) ? ((const unsigned short*)(S + L.mij))[e] : T.m_ij[e], i = ij >> 8, j = ij & 0xff;
            double a[6], c[6];
            for (int k = 0; k < 6; k++) { a[k] = S[L.cdof + 6 * j + k]; c[k] = S[L.cdofdot + 6 * i + k]; }
            v = dot6(a, c);
            if (i == j) v += mb[A.o.dof_armature + i];
            S[L.M + e] = v;
        }
        agpr_put(MP.lo[m], __double2loint(v)); agpr_put(MP.hi[m], __double2hiint(v));
    }
    wsync();
}

// Per-lane topology constants, loaded once per kernel: the lane owns dofs LANE and LANE+64.
// pk packs (madr | depth << 16 | ndesc << 24) so that a wave-uniform dof index i can fetch its row
// address / depth / descendant count with one v_readlane instead of a table load.
template <bool DENSE>
__device__ __forceinline__ LaneConst lane_const(const DevTopo& T) {
    LaneConst c;
    const int i0 = LANE, i1 = LANE + UHC_WAVE;
    c.v0 = i0 < T.nv; c.v1 = i1 < T.nv;
    c.d0 = c.v0 ? T.dof_depth[i0] : 0; c.d1 = c.v1 ? T.dof_depth[i1] : 0;
    c.n0 = c.v0 ? T.dof_ndesc[i0] : -1; c.n1 = c.v1 ? T.dof_ndesc[i1] : -1;
    c.m0 = c.v0 ? T.dof_madr[i0] : 0; c.m1 = c.v1 ? T.dof_madr[i1] : 0;
    c.r0 = (DENSE && c.v0) ? T.dof_rootid[i0] : 0; c.r1 = (DENSE && c.v1) ? T.dof_rootid[i1] : 0;
    c.pk0 = c.m0 | (c.d0 << 16) | ((c.v0 ? c.n0 : 0) << 24);
    c.pk1 = c.m1 | (c.d1 << 16) | ((c.v1 ? c.n1 : 0) << 24);
    return c;
}

// 1 / x to full double precision: v_rcp_f64 + two Newton steps (~35 cycles; the IEEE division sequence with its scale / fixup costs ~77)
__device__ __forceinline__ double rcp_newton(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}
// In-place L^T D L of the tree-sparse matrix in the LD buffer (same elimination order as MuJoCo's mj_factorM [MJ-ext]);
// also dinv[i] = 1/D[i].  The updates of elimination step k,  LD[row(anc_a) + t] -= (LD[kk+a] / D_k) * LD[kk+a+t]  for
// every ancestor a and offset t, depend on the tree only, so the host lays them out once as a flat program of groups
// (DevTopo::fac_prog): per group a lane gets one 24-byte record with the ready LDS byte addresses of three updates of one step,
// of that step's D_k and -- in the step's last group -- of its entry of row k to normalise.  The loop body is the same for every
// group: no step bookkeeping, predicates, address arithmetic or branches (a wave-uniform branch costs 25-60 cycles with one wave
// per SIMD; the earlier per-step version spent most of its ~1100 cycles per step on them).  Idle lanes read the zero slot and
// write the dump slot.  The LDS queue of one wave is in order, so consecutive groups and steps need no barrier; the record of
// the next group streams in from L2 meanwhile.
__device__ __forceinline__ double lds_at(const char* SB, unsigned int byte_off) { return *(const double*)(SB + byte_off); }
template <int TIER>
__device__ __forceinline__ void k_factor(const KernelArgs& A, double* S, int ld, const LaneConst& LC) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    double* LD = S + ld;
    char* SB = (char*)S + cap_of<TIER>(A).ld_delta;  // schedule addresses are byte offsets in the fast layout
    const unsigned int zero_adr = (unsigned)(A.lf.LD + T.nM) * 8u, dump_adr = zero_adr + 8u;
    struct FacRec { unsigned int fr[3], o01, o2n, dk; };
    const FacRec* pg = (const FacRec*)T.fac_prog + LANE;
    FacRec cur = pg[0];
    for (int g = 0; g < T.fac_nslot; g++) {
        const FacRec nxt = pg[(size_t)(g + 1) * UHC_WAVE];  // the table carries two groups of slack
        const unsigned int nr = cur.o2n >> 16;
        const unsigned int oa[3] = {cur.o01 & 0xffffu, cur.o01 >> 16, cur.o2n & 0xffffu};
        const double Dk = lds_at(SB, cur.dk);
        double f[3], r[3], o[3];
#pragma unroll
        for (int q = 0; q < 3; q++) { f[q] = lds_at(SB, cur.fr[q] & 0xffffu); r[q] = lds_at(SB, cur.fr[q] >> 16); o[q] = lds_at(SB, oa[q]); }
        const double fraw = lds_at(SB, nr);
        const double inv = rcp_newton(Dk);  // on the group-to-group critical path: 1 / D_k feeds every store
#pragma unroll
        for (int q = 0; q < 3; q++) *(double*)(SB + oa[q]) = fma(-(f[q] * inv), r[q], o[q]);
        *(double*)(SB + (nr == zero_adr ? dump_adr : nr)) = fraw * inv;
        cur = nxt;
    }
    wsync();
    if (LC.v0) { const double di = 1.0 / LD[LC.m0]; S[L.dinv + LANE] = di; S[L.sdinv + LANE] = sqrt(di); }
    if (LC.v1) { const double di = 1.0 / LD[LC.m1]; S[L.dinv + LANE + UHC_WAVE] = di; S[L.sdinv + LANE + UHC_WAVE] = sqrt(di); }
    wsync();
}

// x = M^-1 x for one right-hand side held in registers (lane owns dofs LANE and LANE+64).
// half: 0 = full solve, 1 = only  L^-1  (used after the constraint solve: qacc += L^-1 D^-1/2 z).
// The serial chain is register-only (v_readlane + FMA).  Which L entry a lane needs at each step is static: the
// host tables sol_back / sol_fwd hold its LDS address (or the zero slot) for both of the lane's dofs; the words of
// the next block of U steps are fetched while the current block runs, the block's L entries are read up front.
struct DofVec { double a, b; };
// dv_get without control flow: both halves are read and the words selected with scalar logic (a uniform branch costs 25-60 cycles
// here, and this sits on the step-to-step chain of the substitutions)
__device__ __forceinline__ double dv_get_nb(const DofVec& x, int i) {  // i wave-uniform, 0 <= i < 128
    const int l = i & (UHC_WAVE - 1), m = -(i >> 6);
    const int alo = __builtin_amdgcn_readlane(__double2loint(x.a), l), ahi = __builtin_amdgcn_readlane(__double2hiint(x.a), l);
    const int blo = __builtin_amdgcn_readlane(__double2loint(x.b), l), bhi = __builtin_amdgcn_readlane(__double2hiint(x.b), l);
    return __hiloint2double(ahi ^ ((ahi ^ bhi) & m), alo ^ ((alo ^ blo) & m));
}
// The tables are padded by the host to a multiple of U steps (+ one block of look-ahead) with no-op steps (zero-slot
// addresses), so the loop body has no tests at all.
template <bool BACK>
__device__ __forceinline__ void solve_sweep(const unsigned int* tab, const char* SB, int n, DofVec& x) {
    constexpr int U = 8;
    const int np = (n + U - 1) & ~(U - 1);
    unsigned int wn[U];
#pragma unroll
    for (int u = 0; u < U; u++) wn[u] = tab[(size_t)u * UHC_WAVE];
    for (int s0 = 0; s0 < np; s0 += U) {
        double l0[U], l1[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const unsigned int ww = wn[u];
            wn[u] = tab[(size_t)(s0 + U + u) * UHC_WAVE];
            l0[u] = lds_at(SB, ww & 0xffffu);
            l1[u] = lds_at(SB, ww >> 16);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the 16 LDS reads of the block ahead of its serial chain (the scheduler sinks them to their uses otherwise)
        // one test per block: when all eight pivots are dofs < 64 (8 of the humanoid's 10 blocks) the broadcast reads x.a only
        if (BACK ? n - s0 < UHC_WAVE : s0 + U <= UHC_WAVE) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const double xs = bcast(x.a, BACK ? max(n - s0 - u, 0) : s0 + u);
                x.a = fma(-l0[u], xs, x.a);
                x.b = fma(-l1[u], xs, x.b);
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int s = s0 + u;
                const double xs = dv_get_nb(x, BACK ? max(n - s, 0) : s);
                x.a = fma(-l0[u], xs, x.a);
                x.b = fma(-l1[u], xs, x.b);
            }
        }
    }
}
// The same sweep for NR right-hand sides at once: the L entries are read once, and the NR serial chains (readlane -> FMA) interleave --
// the sweep is latency-bound, so four right-hand sides cost little more than one (the dense rows of a contact pyramid go through together).
template <bool BACK, int NR>
__device__ __forceinline__ void solve_sweep_n(const unsigned int* tab, const char* SB, int n, DofVec (&x)[NR]) {
    constexpr int U = 8;
    const int np = (n + U - 1) & ~(U - 1);
    unsigned int wn[U];
#pragma unroll
    for (int u = 0; u < U; u++) wn[u] = tab[(size_t)u * UHC_WAVE];
    for (int s0 = 0; s0 < np; s0 += U) {
        double l0[U], l1[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const unsigned int ww = wn[u];
            wn[u] = tab[(size_t)(s0 + U + u) * UHC_WAVE];
            l0[u] = lds_at(SB, ww & 0xffffu);
            l1[u] = lds_at(SB, ww >> 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (BACK ? n - s0 < UHC_WAVE : s0 + U <= UHC_WAVE) {
#pragma unroll
            for (int u = 0; u < U; u++) {
#pragma unroll
                for (int j = 0; j < NR; j++) {
                    const double xs = bcast(x[j].a, BACK ? max(n - s0 - u, 0) : s0 + u);
                    x[j].a = fma(-l0[u], xs, x[j].a);
                    x[j].b = fma(-l1[u], xs, x[j].b);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int s = s0 + u;
#pragma unroll
                for (int j = 0; j < NR; j++) {
                    const double xs = dv_get_nb(x[j], BACK ? max(n - s, 0) : s);
                    x[j].a = fma(-l0[u], xs, x[j].a);
                    x[j].b = fma(-l1[u], xs, x[j].b);
                }
            }
        }
    }
}
template <int TIER>
__device__ __forceinline__ void k_solve(const KernelArgs& A, const double* S, int ld, DofVec& x, int half, const LaneConst& LC) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    const char* SB = (const char*)S + cap_of<TIER>(A).ld_delta;
    if (T.nv < 2) { if (!half && LC.v0) x.a *= S[L.dinv + LANE]; return; }
    if (!half) {
        // x <- L^-T x : for i descending, every ancestor j of i:  x[j] -= L[i][j] x[i]
        solve_sweep<true>(T.sol_back + LANE, SB, T.nv - 1, x);
        if (LC.v0) x.a *= S[L.dinv + LANE];
        if (LC.v1) x.b *= S[L.dinv + LANE + UHC_WAVE];
    }
    // x <- L^-1 x : for j ascending, every descendant i of j:  x[i] -= L[i][j] x[j]
    solve_sweep<false>(T.sol_fwd + LANE, SB, T.nv - 1, x);
}

// ------------------------------------------------------------------ P7 velocities + bias forces
template <int TIER>
__device__ __forceinline__ void k_com_vel(const KernelArgs& A, double* S, const BodyConst& BC) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    const int b = LANE;
    if (b == 0) for (int k = 0; k < 6; k++) S[L.cvel + k] = 0;
    wsync();
    const int depth = BC.depth;
    for (int level = 1; level <= T.body_maxdepth; level++) {
        if (depth == level) {
            double cvel[6], t[6], cd[6];
            const int p = BC.p;
            for (int k = 0; k < 6; k++) cvel[k] = S[L.cvel + 6 * p + k];
            const int jn = BC.jn;
            int dnext = BC.da;
            for (int q = 0; q < jn; q++) {
                const int jt = jt_of(BC, q);
                int da = dnext;
                dnext += jdofs(jt);
                if (jt == UHC_JNT_FREE) {
                    for (int k = 0; k < 18; k++) S[L.cdofdot + 6 * da + k] = 0;
                    for (int k = 0; k < 3; k++) {
                        const double qv = S[L.qvel + da + k];
                        for (int s = 0; s < 6; s++) cvel[s] += S[L.cdof + 6 * (da + k) + s] * qv;
                    }
                    da += 3;
                }
                if (jt == UHC_JNT_FREE || jt == UHC_JNT_BALL) {
                    for (int k = 0; k < 3; k++) {
                        for (int s = 0; s < 6; s++) cd[s] = S[L.cdof + 6 * (da + k) + s];
                        cross_motion(t, cvel, cd);
                        for (int s = 0; s < 6; s++) S[L.cdofdot + 6 * (da + k) + s] = t[s];
                    }
                    for (int k = 0; k < 3; k++) {
                        const double qv = S[L.qvel + da + k];
                        for (int s = 0; s < 6; s++) cvel[s] += S[L.cdof + 6 * (da + k) + s] * qv;
                    }
                } else {
                    for (int s = 0; s < 6; s++) cd[s] = S[L.cdof + 6 * da + s];
                    cross_motion(t, cvel, cd);
                    const double qv = S[L.qvel + da];
                    for (int s = 0; s < 6; s++) { S[L.cdofdot + 6 * da + s] = t[s]; cvel[s] += cd[s] * qv; }
                }
            }
            for (int k = 0; k < 6; k++) S[L.cvel + 6 * b + k] = cvel[k];
        }
        wsync();
    }
}
template <int TIER>
__device__ __forceinline__ void k_rne(const KernelArgs& A, double* S, const BodyConst& BC PROF_ARGS) {  // qfrc_bias = RNE(qacc = 0)
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    const int b = LANE;
    if (b == 0) for (int k = 0; k < 6; k++) { S[L.cacc + k] = k < 3 ? 0.0 : -T.gravity[k - 3]; S[L.cfrc + k] = 0; }
    wsync();
    const int depth = BC.depth;
    for (int level = 1; level <= T.body_maxdepth; level++) {
        if (depth == level) {
            double cacc[6], cvel[6], I[10], t[6], u[6], f[6];
            const int p = BC.p;
            for (int k = 0; k < 6; k++) { cacc[k] = S[L.cacc + 6 * p + k]; cvel[k] = S[L.cvel + 6 * b + k]; }
            const int da = BC.da, dn = BC.dn;
            for (int i = da; i < da + dn; i++) {
                const double qv = S[L.qvel + i];
                for (int s = 0; s < 6; s++) cacc[s] += S[L.cdofdot + 6 * i + s] * qv;
            }
            for (int k = 0; k < 10; k++) I[k] = S[L.cinert + 10 * b + k];
            inert_mul(t, I, cacc);
            inert_mul(u, I, cvel);
            cross_force(f, cvel, u);
            for (int k = 0; k < 6; k++) { S[L.cacc + 6 * b + k] = cacc[k]; S[L.cfrc + 6 * b + k] = f[k] + t[k]; }
        }
        wsync();
    }
    PROF(23)
    // subtree sums of cfrc (DFS order), one work item per (body, component); the sums replace cfrc after all reads
    double sub[3] = {0, 0, 0};
    for (int pass = 0; pass < 3; pass++) {
        const int item = LANE + UHC_WAVE * pass;
        if (item < 6 * T.nbody) {
            const int bb = item / 6, k = item - 6 * bb;
            if (bb > 0) {
                const int n = T.body_nsub[bb];
                for (int c = bb + n - 1; c >= bb; c--) sub[pass] += S[L.cfrc + 6 * c + k];
            }
        }
    }
    wsync();
    for (int pass = 0; pass < 3; pass++) {
        const int item = LANE + UHC_WAVE * pass;
        if (item >= 6 && item < 6 * T.nbody) S[L.cfrc + item] = sub[pass];
    }
    wsync();
    for (int i = LANE; i < T.nv; i += UHC_WAVE) {
        double a[6], c[6];
        const int bi = T.dof_bodyid[i];
        for (int k = 0; k < 6; k++) { a[k] = S[L.cdof + 6 * i + k]; c[k] = S[L.cfrc + 6 * bi + k]; }
        S[L.bias + i] = dot6(a, c);
    }
    wsync();
}

// ------------------------------------------------------------------ P8 smooth forces / acceleration
template <int TIER>
__device__ __forceinline__ void k_smooth(const KernelArgs& A, const double* mb, double* S, const LaneConst& LC) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    // qfrc_smooth = passive - bias + applied + actuator  -> S.smooth
    for (int i = LANE; i < T.nv; i += UHC_WAVE) {
        double f = -mb[A.o.dof_damping + i] * S[L.qvel + i] - S[L.bias + i] + S[L.applied + i];
        const int j = T.dof_jntid[i], jt = T.jnt_type[j];
        const double k = mb[A.o.jnt_stiffness + j];
        if (k != 0 && (jt == UHC_JNT_HINGE || jt == UHC_JNT_SLIDE)) {
            const int qa = T.jnt_qposadr[j];
            f -= k * (S[L.qpos + qa] - mb[A.o.qpos_spring + qa]);
        }
        // actuation: the motors of the dof's joint, gear component = the dof's index inside the joint (0 on scalar joints)
        const int comp = i - T.jnt_dofadr[j];
#pragma unroll
        for (int q = 0; q < UHC_DOF_MAXACT; q++) {
            const int a = T.dof_act[i * UHC_DOF_MAXACT + q];
            if (a >= 0) f += mb[A.o.actuator_gear + 3 * a + comp] * S[L.ctrl + a];
        }
        S[L.smooth + i] = f;
    }
    wsync();
    DofVec x;
    x.a = LANE < T.nv ? S[L.smooth + LANE] : 0.0;
    x.b = LANE + UHC_WAVE < T.nv ? S[L.smooth + LANE + UHC_WAVE] : 0.0;
    k_solve<TIER>(A, S, L.LD, x, 0, LC);
    if (LANE < T.nv) S[L.smooth + LANE] = x.a;  // now qacc_smooth
    if (LANE + UHC_WAVE < T.nv) S[L.smooth + LANE + UHC_WAVE] = x.b;
    wsync();
}

// ------------------------------------------------------------------ P4 collision: plane vs convex mesh
// [MJ-ext] mju_makeFrame: x given; y = the supplied axis when there is one (mjc_PlaneCapsule lays the contact frame's second axis along the capsule) and it is not
// (nearly) parallel to x, else picked -- (0, 1, 0), or (0, 0, 1) where x is close to it; y made orthogonal to x and normalised, z = x cross y.  One orthogonalisation
// for both cases: without a hint the arithmetic is what it always was.
__device__ __forceinline__ void make_frame(double* f, const double* yh = nullptr) {
    double n = sqrt(dot3(f, f));
    f[0] /= n; f[1] /= n; f[2] /= n;
    f[3] = f[4] = f[5] = 0;
    if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1;
    if (yh) {
        const double dh = dot3(f, yh);
        const double t0 = yh[0] - f[0] * dh, t1 = yh[1] - f[1] * dh, t2 = yh[2] - f[2] * dh;
        if (sqrt(t0 * t0 + t1 * t1 + t2 * t2) >= 1e-8) { f[3] = yh[0]; f[4] = yh[1]; f[5] = yh[2]; }
    }
    const double dp = dot3(f, f + 3);
    for (int k = 0; k < 3; k++) f[3 + k] -= f[k] * dp;
    n = sqrt(dot3(f + 3, f + 3));
    f[3] /= n; f[4] /= n; f[5] /= n;
    cross3(f + 6, f, f + 3);
}
__device__ __forceinline__ double impedance(const double* si, double pos, double margin) {
    double dmin = clampd(si[0], 0.0001, 0.9999), dmax = clampd(si[1], 0.0001, 0.9999), width = si[2];
    double mid = clampd(si[3], 0.0001, 0.9999), power = si[4] < 1 ? 1 : si[4];
    if (width < UHC_MINVAL) return 0.5 * (dmin + dmax);
    const double x = fabs(pos - margin) / width;
    double y;
    if (x >= 1) return dmax;
    if (x <= 0) return dmin;
    if (power == 1) y = x;
    else if (x <= mid) y = pow(x / mid, power) * mid;
    else y = 1 - pow((1 - x) / (1 - mid), power) * (1 - mid);
    return dmin + y * (dmax - dmin);
}
// Per-lane constants of the statically filtered collision pairs (lane = pair, first 64 pairs): geom / body ids, hull vertex
// range and contact dimension, loaded once per kernel; a wave-uniform pair index fetches them with v_readlane instead of a
// chain of dependent global table loads every substep.
struct PairConst { int g1, g2, b1, b2, va, vn, dim; };  // dim: bit 8 set = the hull is a capsule (its contact frames lie along its axis)
__device__ __forceinline__ int pair_dim(const DevTopo& T, int g1, int g2) {
    return max(T.geom_condim[g1], T.geom_condim[g2]) | ((T.geom_type[g2] == UHC_GEOM_CAPSULE && T.geom_vertnum[g2] == 2) ? 256 : 0);
}
__device__ __forceinline__ PairConst pair_const(const DevTopo& T) {
    PairConst c = {0, 0, 0, 0, 0, 0, 0};
    if (LANE < T.npair) {
        c.g1 = T.pair_g1[LANE]; c.g2 = T.pair_g2[LANE];
        c.b1 = T.geom_bodyid[c.g1]; c.b2 = T.geom_bodyid[c.g2];
        c.va = T.geom_vertadr[c.g2]; c.vn = T.geom_vertnum[c.g2];
        c.dim = pair_dim(T, c.g1, c.g2);
    }
    return c;
}
__device__ __forceinline__ PairConst pair_of(const DevTopo& T, const PairConst& mine, int pi) {  // pi wave-uniform
    PairConst c;
    if (pi < UHC_WAVE) {
        c.g1 = __builtin_amdgcn_readlane(mine.g1, pi); c.g2 = __builtin_amdgcn_readlane(mine.g2, pi);
        c.b1 = __builtin_amdgcn_readlane(mine.b1, pi); c.b2 = __builtin_amdgcn_readlane(mine.b2, pi);
        c.va = __builtin_amdgcn_readlane(mine.va, pi); c.vn = __builtin_amdgcn_readlane(mine.vn, pi);
        c.dim = __builtin_amdgcn_readlane(mine.dim, pi);
    } else {
        c.g1 = T.pair_g1[pi]; c.g2 = T.pair_g2[pi];
        c.b1 = T.geom_bodyid[c.g1]; c.b2 = T.geom_bodyid[c.g2];
        c.va = T.geom_vertadr[c.g2]; c.vn = T.geom_vertnum[c.g2];
        c.dim = pair_dim(T, c.g1, c.g2);
    }
    return c;
}
template <int TIER>
__device__ __forceinline__ void k_write_contact(const KernelArgs& A, const double* mb, double* S, int c, int g1, int g2, int b1, int b2, int dim,
                                                const double* pos, const double* n, double dist, double margin, double gap, const double* yhint = nullptr) {
    const DevTopo& T = A.t;
    double* C = S + lds_of<TIER>(A).con + c * UHC_CON_STRIDE;
    double fr[9];
    for (int k = 0; k < 3; k++) { C[k] = pos[k]; fr[k] = n[k]; }
    make_frame(fr, yhint);
    for (int k = 0; k < 9; k++) C[3 + k] = fr[k];
    const double inc = margin - gap;
    double solref[2], solimp[5];
    for (int k = 0; k < 2; k++) solref[k] = 0.5 * (mb[A.o.geom_solref + 2 * g1 + k] + mb[A.o.geom_solref + 2 * g2 + k]);
    for (int k = 0; k < 5; k++) solimp[k] = 0.5 * (mb[A.o.geom_solimp + 5 * g1 + k] + mb[A.o.geom_solimp + 5 * g2 + k]);
    const double timeconst = fmax(solref[0], 2 * T.timestep), dampratio = solref[1];
    const double dmax = clampd(solimp[1], 0.0001, 0.9999);
    C[12] = dist; C[13] = inc;
    C[14] = fmax(mb[A.o.geom_friction + 3 * g1], mb[A.o.geom_friction + 3 * g2]);
    C[15] = 1.0 / (dmax * dmax * timeconst * timeconst * dampratio * dampratio);
    C[16] = 2.0 / (dmax * timeconst);
    C[17] = impedance(solimp, dist, inc);
    C[18] = mb[A.o.body_invweight0 + 2 * b1] + mb[A.o.body_invweight0 + 2 * b2];
    C[19] = b1; C[20] = b2; C[21] = dim;
}
// [MJ-ext] mj_instantiateLimit, ball joint: the joint's rotation as angle * axis (mju_quat2Vel with dt = 1: axis = the unit quaternion's
// vector part normalised, angle = 2 atan2(|vector part|, w) taken into (-pi, pi]); value = |angle|, dist = max(range) - value.  Returns dist
// and the row's Jacobian on the joint's three dofs, -axis of the rotation as it is signed (jac).
__device__ __forceinline__ double ball_limit(const double* qp, const double* range, double* jac) {
    double q[4] = {qp[0], qp[1], qp[2], qp[3]}, ax[3] = {1, 0, 0};
    quat_normalize(q);
    double sn = sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (sn < UHC_MINVAL) sn = 0; else { ax[0] = q[1] / sn; ax[1] = q[2] / sn; ax[2] = q[3] / sn; }
    double ang = 2 * atan2(sn, q[0]);
    if (ang > M_PI) ang -= 2 * M_PI;
    double value = fabs(ang), sg = ang < 0 ? -1.0 : 1.0;
    if (value < UHC_MINVAL) { ax[0] = 1; ax[1] = ax[2] = 0; sg = 1; value = 0; }
    for (int k = 0; k < 3; k++) jac[k] = -sg * ax[k];
    return fmax(range[0], range[1]) - value;
}
// returns ncon (wave-uniform)
template <int TIER, bool DENSE>
__device__ __forceinline__ int k_collision(const KernelArgs& A, const double* mb, double* S, int* overflow, const PairConst& PC PROF_ARGS) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    int ncon = 0;
    for (int p0 = 0; p0 < T.npair; p0 += UHC_WAVE) {
        const int p = p0 + LANE;
        bool keep = false;
        if (p < T.npair) {  // bounding-sphere cull, one pair per lane
            PairConst P = PC;  // the lane's own pair when p0 == 0; further blocks of 64 pairs read the tables
            if (p0 > 0) {
                P.g1 = T.pair_g1[p]; P.g2 = T.pair_g2[p];
                P.b1 = T.geom_bodyid[P.g1]; P.b2 = T.geom_bodyid[P.g2];
            }
            const int g1 = P.g1, g2 = P.g2, b1 = P.b1, b2 = P.b2;
            double pq[4], gq[4], xq[4], R[9], t[3], gp[3], ce[3];
            for (int k = 0; k < 4; k++) { gq[k] = mb[A.o.geom_quat + 4 * g1 + k]; xq[k] = S[L.xquat + 4 * b1 + k]; }
            quat_mul(pq, xq, gq);
            quat_to_mat(R, pq);
            double m1[9], m2[9];
            for (int k = 0; k < 9; k++) { m1[k] = S[L.xmat + 9 * b1 + k]; m2[k] = S[L.xmat + 9 * b2 + k]; }
            for (int k = 0; k < 3; k++) { gp[k] = mb[A.o.geom_pos + 3 * g1 + k]; ce[k] = mb[A.o.geom_center + 3 * g2 + k]; }
            mat_vec(t, m1, gp);
            double ppos[3], n[3] = {R[2], R[5], R[8]};
            for (int k = 0; k < 3; k++) ppos[k] = S[L.xpos + 3 * b1 + k] + t[k];
            mat_vec(t, m2, ce);
            double cd = 0;
            for (int k = 0; k < 3; k++) cd += n[k] * (S[L.xpos + 3 * b2 + k] + t[k] - ppos[k]);
            const double margin = fmax(mb[A.o.geom_margin + g1], mb[A.o.geom_margin + g2]);
            keep = !(cd - mb[A.o.geom_rbound + g2] > margin);
        }
        unsigned long long mask = __ballot(keep);
        while (mask) {
            const int pi = p0 + __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const PairConst P = pair_of(T, PC, pi);
            const int g1 = P.g1, g2 = P.g2, b1 = P.b1, b2 = P.b2;
            double pq[4], gq[4], xq[4], R[9], t[3], gp[3], m1[9], m2[9], xp2[3];
            for (int k = 0; k < 4; k++) { gq[k] = mb[A.o.geom_quat + 4 * g1 + k]; xq[k] = S[L.xquat + 4 * b1 + k]; }
            quat_mul(pq, xq, gq);
            quat_to_mat(R, pq);
            for (int k = 0; k < 9; k++) { m1[k] = S[L.xmat + 9 * b1 + k]; m2[k] = S[L.xmat + 9 * b2 + k]; }
            for (int k = 0; k < 3; k++) { gp[k] = mb[A.o.geom_pos + 3 * g1 + k]; xp2[k] = S[L.xpos + 3 * b2 + k]; }
            mat_vec(t, m1, gp);
            double ppos[3], n[3] = {R[2], R[5], R[8]};
            for (int k = 0; k < 3; k++) ppos[k] = S[L.xpos + 3 * b1 + k] + t[k];
            const double margin = fmax(mb[A.o.geom_margin + g1], mb[A.o.geom_margin + g2]);
            const double gap = fmax(mb[A.o.geom_gap + g1], mb[A.o.geom_gap + g2]);
            const double rr = mb[A.o.geom_radius + g2];  // a rounded hull's radius (0 for a mesh); loaded here, with the margin and the gap: nothing waits for it before the arg-min is through
            const int va = P.va, vn = P.vn;
            // support vertex along -normal: lane-local min then wave arg-min (ties -> lowest vertex id)
            double bd = 1e300;
            int bv = 0x7fffffff;
            for (int v = va + LANE; v < va + vn; v += UHC_WAVE) {
                double lv[3] = {mb[A.o.mesh_vert + 3 * v], mb[A.o.mesh_vert + 3 * v + 1], mb[A.o.mesh_vert + 3 * v + 2]}, w[3];
                mat_vec(w, m2, lv);
                double dist = 0;
                for (int k = 0; k < 3; k++) dist += n[k] * (w[k] + xp2[k] - ppos[k]);
                if (dist < bd) { bd = dist; bv = v; }
            }
            {   // wave arg-min: DPP min, then the lowest vertex id among the lanes that hold it
                const double mn = wave_min(bd);
                const unsigned long long tie = __builtin_amdgcn_ballot_w64(bd == mn);
                int cand = bd == mn ? bv : 0x7fffffff;
                if (__popcll(tie) == 1) cand = __builtin_amdgcn_readlane(bv, __ffsll((long long)tie) - 1);
                else {
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o));
                }
                bd = mn; bv = cand;
            }
            // a ROUNDED hull (sphere: one core vertex, capsule: the two ends of its segment) is its core lowered by the radius: [MJ-ext] _PlaneSphere at the centre / at
            // both segment ends (mjc_PlaneCapsule, which also lays the contact frame's second axis along the capsule); 0 for a mesh
            bd -= rr;
            if (bd > margin || bv == 0x7fffffff) continue;
            double cax[3] = {0, 0, 0};
            const bool capsule = (P.dim & 256) != 0;
            if (capsule) {  // the capsule's axis in the world: from its second core vertex to its first
                double dl[3];
                for (int k = 0; k < 3; k++) dl[k] = mb[A.o.mesh_vert + 3 * va + k] - mb[A.o.mesh_vert + 3 * (va + 1) + k];
                mat_vec(cax, m2, dl);
            }
            // candidate 0 = support vertex, candidates 1.. = its hull neighbours (adjacency order)
            // (the hull graph belongs to the env's model: body shapes generated from different betas have different hulls.  Fixed-stride
            //  table of neighbour ids in the model blob, -1 past a vertex's own degree)
            const int* adj = (const int*)(mb + A.o.mesh_adj) + (size_t)bv * A.adjdeg;
            bool ok = false;
            double w[3] = {0, 0, 0}, dist = 0;
            const int v = LANE == 0 ? bv : (LANE <= A.adjdeg ? adj[LANE - 1] : -1);
            if (v >= 0) {
                double lv[3] = {mb[A.o.mesh_vert + 3 * v], mb[A.o.mesh_vert + 3 * v + 1], mb[A.o.mesh_vert + 3 * v + 2]};
                mat_vec(w, m2, lv);
                for (int k = 0; k < 3; k++) { w[k] += xp2[k]; dist += n[k] * (w[k] - ppos[k]); }
                dist -= rr;
                ok = LANE == 0 || dist <= margin;
            }
            const unsigned long long cm = __ballot(ok);
            const int rank = __popcll(cm & ((1ull << LANE) - 1ull));
            if (ok && rank < T.plane_mesh_maxcon && ncon + rank < cap_of<TIER>(A).maxcon) {
                const double push = rr + 0.5 * dist;
                const double cp[3] = {w[0] - push * n[0], w[1] - push * n[1], w[2] - push * n[2]};
                k_write_contact<TIER>(A, mb, S, ncon + rank, P.g1, P.g2, P.b1, P.b2, P.dim & 255, cp, n, dist, margin, gap, capsule ? cax : nullptr);
            }
            const int want = ncon + min((int)__popcll(cm), T.plane_mesh_maxcon);
            if (want > cap_of<TIER>(A).maxcon) *overflow |= (hands_on<TIER>(A) ? 1 : 2) | UHC_WHY_CONTACTS;  // 1: needs the next tier, 2: dropped
            ncon = min(cap_of<TIER>(A).maxcon, want);
        }
    }
    // ---- convex-convex pairs (hull vs hull): one candidate pair per lane through bounding-sphere cull and MPR (uhc_mpr.h); the hits
    //      become contacts in pair order.  Contact = (pos, normal from geom 1 to geom 2, dist = margin - depth) [MJ-ext mjc_Convex].
    PROF(32)
    if constexpr (DENSE) {
        // phase 1: bounding-sphere cull of every pair, survivors compacted (in pair order) into a list -- phase 2 then needs one
        // MPR pass for the 10-40 candidates of a typical pose instead of one per block of 64 pairs
        int* cand = (int*)(S + L.rowMisc);  // free until the rows are enumerated
        constexpr int CAND_CAP = 256;
        int ncand = 0;
        for (int p0 = 0; p0 < T.ncpair; p0 += UHC_WAVE) {
            const int p = p0 + LANE;
            bool c = false;
            if (p < T.ncpair) {
                const int g1 = T.cpair_g1[p], g2 = T.cpair_g2[p], b1 = T.geom_bodyid[g1], b2 = T.geom_bodyid[g2];
                double ce1[3], ce2[3], t1[3], t2[3], R1[9], R2[9];
                for (int k = 0; k < 9; k++) { R1[k] = S[L.xmat + 9 * b1 + k]; R2[k] = S[L.xmat + 9 * b2 + k]; }
                for (int k = 0; k < 3; k++) { ce1[k] = mb[A.o.geom_center + 3 * g1 + k]; ce2[k] = mb[A.o.geom_center + 3 * g2 + k]; }
                mat_vec(t1, R1, ce1); mat_vec(t2, R2, ce2);
                double d2 = 0;
                for (int k = 0; k < 3; k++) { const double d = (t1[k] + S[L.xpos + 3 * b1 + k]) - (t2[k] + S[L.xpos + 3 * b2 + k]); d2 += d * d; }
                const double margin = fmax(mb[A.o.geom_margin + g1], mb[A.o.geom_margin + g2]);
                const double bound = mb[A.o.geom_rbound + g1] + mb[A.o.geom_rbound + g2] + margin;
                c = !(d2 > bound * bound);
                // second cull: the hulls' boxes in their body frames (model constants), separating-axis test on the six face normals.  A
                // limb's bounding sphere reaches far beyond its sides; side by side (thighs, arm along the torso) the spheres overlap in
                // every pose and the pair went through the whole portal search only to be found apart.  Conservative: a box holds its
                // hull, and boxes further apart than the margin along an axis mean hulls further apart than the margin.
                if (c && !(A.dbg & 128)) {
                    const double* B1 = mb + A.o.geom_box + 6 * g1;
                    const double* B2 = mb + A.o.geom_box + 6 * g2;
                    double c1[3], c2[3], dw[3];
                    mat_vec(c1, R1, B1); mat_vec(c2, R2, B2);
                    for (int k = 0; k < 3; k++) dw[k] = (c2[k] + S[L.xpos + 3 * b2 + k]) - (c1[k] + S[L.xpos + 3 * b1 + k]);
                    // columns of R are the body axes in the world: C[i][k] = axis_i(1) . axis_k(2)
                    double C[3][3], ta[3], tb[3];
                    for (int i = 0; i < 3; i++) {
                        ta[i] = dw[0] * R1[i] + dw[1] * R1[3 + i] + dw[2] * R1[6 + i];
                        tb[i] = dw[0] * R2[i] + dw[1] * R2[3 + i] + dw[2] * R2[6 + i];
                        for (int k = 0; k < 3; k++) C[i][k] = fabs(R1[i] * R2[k] + R1[3 + i] * R2[3 + k] + R1[6 + i] * R2[6 + k]);
                    }
                    bool apart = false;
                    for (int i = 0; i < 3; i++) {
                        apart |= fabs(ta[i]) > B1[3 + i] + B2[3] * C[i][0] + B2[4] * C[i][1] + B2[5] * C[i][2] + margin;
                        apart |= fabs(tb[i]) > B2[3 + i] + B1[3] * C[0][i] + B1[4] * C[1][i] + B1[5] * C[2][i] + margin;
                    }
                    c = !apart;
                }
            }
            const unsigned long long cm = __ballot(c);
            const int rank = ncand + __popcll(cm & ((1ull << LANE) - 1ull));
            if (c && rank < CAND_CAP) cand[rank] = p;
            ncand += (int)__popcll(cm);
        }
        if (ncand > CAND_CAP) { *overflow |= (hands_on<TIER>(A) ? 1 : 2) | UHC_WHY_CANDIDATES; ncand = CAND_CAP; }
        PROF(33)
        // the hull vertices (body frame, model constants) into the LDS region the constraint rows will use after this pass
        const int vstage = cap_of<TIER>(A).vstage;
        if (vstage >= 0 && ncand > 0)
            for (int i = LANE; i < 3 * T.nmeshvert; i += UHC_WAVE) S[vstage + i] = mb[A.o.mesh_vert + i];
        wsync();
        PROF(34)
        // phase 2: MPR -- lane = candidate pair for the portal logic, the whole wave for every support query (uhc_mpr.h: mpr_wave)
        for (int c0 = 0; c0 < ncand; c0 += UHC_WAVE) {
            const int ci = c0 + LANE;
            const bool act = ci < ncand;
            int g1 = 0, g2 = 0;
            double gap = 0;
            MprLane M;
            M.b1 = M.b2 = M.voff1 = M.vn1 = M.voff2 = M.vn2 = 0;
            M.hm1 = M.hm2 = 0; M.c1 = M.c2 = v3(0, 0, 0);
            double margin = 0;
            if (act) {
                const int p = cand[ci];
                g1 = T.cpair_g1[p]; g2 = T.cpair_g2[p];
                M.b1 = T.geom_bodyid[g1]; M.b2 = T.geom_bodyid[g2];
                M.voff1 = 3 * T.geom_vertadr[g1]; M.vn1 = T.geom_vertnum[g1];
                M.voff2 = 3 * T.geom_vertadr[g2]; M.vn2 = T.geom_vertnum[g2];
                double ce1[3], ce2[3], t1[3], t2[3], R1[9], R2[9];
                for (int k = 0; k < 9; k++) { R1[k] = S[L.xmat + 9 * M.b1 + k]; R2[k] = S[L.xmat + 9 * M.b2 + k]; }
                for (int k = 0; k < 3; k++) { ce1[k] = mb[A.o.geom_center + 3 * g1 + k]; ce2[k] = mb[A.o.geom_center + 3 * g2 + k]; }
                mat_vec(t1, R1, ce1); mat_vec(t2, R2, ce2);
                M.c1 = v3(t1[0] + S[L.xpos + 3 * M.b1], t1[1] + S[L.xpos + 3 * M.b1 + 1], t1[2] + S[L.xpos + 3 * M.b1 + 2]);
                M.c2 = v3(t2[0] + S[L.xpos + 3 * M.b2], t2[1] + S[L.xpos + 3 * M.b2 + 1], t2[2] + S[L.xpos + 3 * M.b2 + 2]);
                margin = fmax(mb[A.o.geom_margin + g1], mb[A.o.geom_margin + g2]);
                M.hm1 = 0.5 * margin + mb[A.o.geom_radius + g1]; M.hm2 = 0.5 * margin + mb[A.o.geom_radius + g2];  // how far each hull's surface lies beyond its (core) vertices
                gap = fmax(mb[A.o.geom_gap + g1], mb[A.o.geom_gap + g2]);
            }
#if defined(UHC_NW2)
            if constexpr (TIER == 2) {  // the two-wave consumer: the helper wave serves every second pair of requests (mailbox on the rows' scalars, free until the rows are enumerated)
                if (vstage >= 0) mpr_wave_mw<true>(S + vstage, S + L.xmat, S + L.xpos, act, M, mpr_mb<TIER>(A, S), vstage, mb + A.o.mesh_vert);
                else mpr_wave_mw<false>(mb + A.o.mesh_vert, S + L.xmat, S + L.xpos, act, M, mpr_mb<TIER>(A, S), -1, mb + A.o.mesh_vert);
            } else
#endif
            // two copies of the refinement so that each knows its address space at compile time (ds_read vs global_load)
            if (vstage >= 0) mpr_wave(S + vstage, S + L.xmat, S + L.xpos, act, M);
            else mpr_wave(mb + A.o.mesh_vert, S + L.xmat, S + L.xpos, act, M);
            const bool hit = act && M.hit && !(M.dir.x == 0 && M.dir.y == 0 && M.dir.z == 0);
            const unsigned long long hm = __ballot(hit);
            const int rank = __popcll(hm & ((1ull << LANE) - 1ull));
            if (hit && ncon + rank < cap_of<TIER>(A).maxcon) {
                const double cp[3] = {M.pos.x, M.pos.y, M.pos.z}, nn[3] = {M.dir.x, M.dir.y, M.dir.z};
                k_write_contact<TIER>(A, mb, S, ncon + rank, g1, g2, M.b1, M.b2, max(T.geom_condim[g1], T.geom_condim[g2]), cp, nn, margin - M.depth, margin, gap);
            }
            const int want = ncon + (int)__popcll(hm);
            if (want > cap_of<TIER>(A).maxcon) *overflow |= (hands_on<TIER>(A) ? 1 : 2) | UHC_WHY_CONTACTS;
            ncon = min(cap_of<TIER>(A).maxcon, want);
        }
        PROF(35)
    }
    wsync();
    return ncon;
}

// ------------------------------------------------------------------ P5 constraint rows (row-per-lane) + half-solved Jacobian
// row types
#define ROW_FRICTION 1
#define ROW_LIMIT 2
#define ROW_CONTACT 3
#define ROW_PYR 4
#define ROW_TWO 0x10   // contact between two moving bodies: the row is kept as a dense dof vector (slot = type >> 8), not along one chain
#define ROW_NEG 0x20   // single-chain contact row whose moving body is geom 1's (geom 2's is static): Jacobian sign -1
#define RTYPE(t) ((t) & 0xf)
struct RowMisc { int type, last, aux, edge; };  // aux: contact id | dof ; edge: pyramid edge | sign

template <int TIER, bool DENSE>
__device__ __forceinline__ int k_enumerate_rows(const KernelArgs& A, const double* mb, double* S, int ncon, int* overflow) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    RowMisc* RM = (RowMisc*)(S + L.rowMisc);
    int nefc = 0;
    // (1) friction loss
    for (int i0 = 0; i0 < T.nv; i0 += UHC_WAVE) {
        const int i = i0 + LANE;
        const bool has = i < T.nv && mb[A.o.dof_frictionloss + i] > 0;
        const unsigned long long m = __ballot(has);
        const int r = nefc + __popcll(m & ((1ull << LANE) - 1ull));
        if (has && r < cap_of<TIER>(A).maxefc) { RM[r].type = ROW_FRICTION; RM[r].last = i; RM[r].aux = i; RM[r].edge = 1; }
        nefc += __popcll(m);
    }
    // (2) joint limits: lower side then upper side of each joint, joints in order
    for (int j0 = 0; j0 < T.njnt; j0 += UHC_WAVE) {
        const int j = j0 + LANE;
        bool lo = false, hi = false, ball = false;
        if (j < T.njnt && T.jnt_limited[j]) {
            const int jt = T.jnt_type[j];
            if (jt == UHC_JNT_HINGE || jt == UHC_JNT_SLIDE) {
                const double v = S[L.qpos + T.jnt_qposadr[j]], mg = mb[A.o.jnt_margin + j];
                lo = (v - mb[A.o.jnt_range + 2 * j]) < mg;
                hi = (mb[A.o.jnt_range + 2 * j + 1] - v) < mg;
            } else if (DENSE && jt == UHC_JNT_BALL) {  // one row: the rotation angle against max(range); the row ends at the joint's third dof
                double jac[3];                         // (DENSE instantiations only: a model with limited ball joints runs on them in every tier)
                ball = true;
                lo = ball_limit(S + L.qpos + T.jnt_qposadr[j], mb + A.o.jnt_range + 2 * j, jac) < mb[A.o.jnt_margin + j];
            }
        }
        const unsigned long long ml = __ballot(lo), mh = __ballot(hi), below = (1ull << LANE) - 1ull;
        int r = nefc + __popcll(ml & below) + __popcll(mh & below);
        if (lo) { if (r < cap_of<TIER>(A).maxefc) { RM[r].type = ROW_LIMIT; RM[r].last = T.jnt_dofadr[j] + (ball ? 2 : 0); RM[r].aux = j; RM[r].edge = ball ? 2 : -1; } r++; }
        if (hi) { if (r < cap_of<TIER>(A).maxefc) { RM[r].type = ROW_LIMIT; RM[r].last = T.jnt_dofadr[j]; RM[r].aux = j; RM[r].edge = 1; } }
        nefc += __popcll(ml) + __popcll(mh);
    }
    wsync();
    // (3) contacts
    if (LANE == 0) {
        int r = nefc, trunc = 0, ntwo = 0, twofull = 0;
        int* NI = (int*)(S + L.ncon_nefc);
        const int maxtwo = !DENSE ? 0 : cap_of<TIER>(A).ndense;
        for (int c = 0; c < ncon; c++) {
            const double* C = S + L.con + c * UHC_CON_STRIDE;
            if (C[12] >= C[13]) continue;
            const int dim = (int)C[21], b1 = (int)C[19], b2 = (int)C[20];
            const int nr = dim == 1 ? 1 : 4, l1 = T.body_lastdof[b1], l2 = T.body_lastdof[b2];
            const bool two = DENSE && l1 >= 0 && l2 >= 0;
            if (TIER == 1 && A.truncate && r + nr > cap_of<TIER>(A).maxefc) { trunc = 1; break; }  // whole contacts only
            if (two && ntwo + nr > maxtwo) { twofull = 1; if (hands_on<TIER>(A)) break; continue; }  // no dense slot left
            for (int e = 0; e < nr; e++, r++)
                if (r < cap_of<TIER>(A).maxefc) {
                    int ty = dim == 1 ? ROW_CONTACT : ROW_PYR;
                    if (two) { ty |= ROW_TWO | (ntwo << 8); NI[4 + ntwo] = r; ntwo++; }
                    else if (l2 < 0) ty |= ROW_NEG;
                    RM[r].type = ty; RM[r].last = two ? -1 : (l2 >= 0 ? l2 : l1); RM[r].aux = c; RM[r].edge = e;
                }
        }
        NI[1] = r; NI[0] = trunc; NI[2] = ntwo; NI[3] = twofull;
    }
    wsync();
    nefc = ((int*)(S + L.ncon_nefc))[1];
    if (((int*)(S + L.ncon_nefc))[0]) *overflow |= 2;
    if (((int*)(S + L.ncon_nefc))[3]) *overflow |= (hands_on<TIER>(A) ? 1 : 2) | UHC_WHY_DENSE_SLOTS;  // the next tier has more dense slots; the last one dropped the rows
    if (nefc > cap_of<TIER>(A).maxefc) { *overflow |= (hands_on<TIER>(A) ? 1 : 2) | UHC_WHY_ROWS; nefc = cap_of<TIER>(A).maxefc; }
    return nefc;
}

// A contact between two moving bodies (hull vs hull of one humanoid, object vs body) has J = J_b2(p) - J_b1(p): non-zero on the union
// of two dof chains (up to 45 dofs for two arms; two disjoint trees for an object), which the chain-packed rows cannot hold.  Such rows
// are kept DENSE: Yhat = D^-1/2 L^-T J^T as an nv-vector in LDS (slot `slot`), built wave-cooperatively with lane = dof: the Jacobian
// entry of dof i is +-dv . (cdof_lin + cdof_ang x (p - c0(i))) if i lies on the chain of body 2 / body 1 (both: the contributions
// cancel exactly), the back substitution is the register-resident sweep of k_solve.  Returns the row's
// J.qvel, J.qacc_smooth, J.qacc_warmstart and |Yhat|^2 (wave-uniform).
struct DenseOut { double vel, jas, jaw, yy; };
#define UHC_DENSE_GROUP 4  // dense rows per back substitution (a contact pyramid)
// rows[j] < 0: no row in position j of the group.  The rows' Yhat go to the dense slots slot0 + j.
template <int TIER>
__device__ __forceinline__ void k_dense_rows(const KernelArgs& A, double* S, const int (&rows)[UHC_DENSE_GROUP], int slot0, const LaneConst& LC,
                                             DenseOut (&o)[UHC_DENSE_GROUP], double* Db) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    DofVec x[UHC_DENSE_GROUP];
    double qv[2] = {0, 0}, qs[2] = {0, 0}, qw[2] = {0, 0};
    for (int h = 0; h < 2; h++) {
        const int i = LANE + h * UHC_WAVE;
        if (!(h ? LC.v1 : LC.v0)) continue;
        qv[h] = S[L.qvel + i]; qs[h] = S[L.smooth + i]; qw[h] = S[L.qacc + i];
    }
#pragma unroll
    for (int j = 0; j < UHC_DENSE_GROUP; j++) {
        x[j].a = 0.0; x[j].b = 0.0;
        o[j].vel = o[j].jas = o[j].jaw = o[j].yy = 0.0;
        if (rows[j] < 0) continue;  // (wave-uniform)
        const RowMisc rm = ((const RowMisc*)(S + L.rowMisc))[rows[j]];
        const double* C = S + L.con + rm.aux * UHC_CON_STRIDE;
        const int l1 = T.body_lastdof[(int)C[19]], l2 = T.body_lastdof[(int)C[20]];
        double dv[3];
        if (RTYPE(rm.type) == ROW_CONTACT) for (int k = 0; k < 3; k++) dv[k] = C[3 + k];
        else {
            const double sgn = (rm.edge & 1) ? -1.0 : 1.0, mu = C[14];
            const int td = 1 + rm.edge / 2;
            for (int k = 0; k < 3; k++) dv[k] = C[3 + k] + sgn * mu * C[3 + 3 * td + k];
        }
        for (int h = 0; h < 2; h++) {
            const int i = LANE + h * UHC_WAVE, nd = h ? LC.n1 : LC.n0, root = h ? LC.r1 : LC.r0;
            if (!(h ? LC.v1 : LC.v0)) continue;
            const int sg = (int)(i <= l2 && l2 <= i + nd) - (int)(i <= l1 && l1 <= i + nd);
            double jv = 0.0;
            if (sg != 0) {
                double cd[6], off[3], cr[3];
                for (int t = 0; t < 6; t++) cd[t] = S[L.cdof + 6 * i + t];
                for (int k = 0; k < 3; k++) off[k] = C[k] - S[L.rootcom + 3 * root + k];
                cross3(cr, cd, off);
                jv = (double)sg * (dv[0] * (cd[3] + cr[0]) + dv[1] * (cd[4] + cr[1]) + dv[2] * (cd[5] + cr[2]));
            }
            if (h) x[j].b = jv; else x[j].a = jv;
        }
        o[j].vel = wave_sum(x[j].a * qv[0] + x[j].b * qv[1]);
        o[j].jas = wave_sum(x[j].a * qs[0] + x[j].b * qs[1]);
        o[j].jaw = wave_sum(x[j].a * qw[0] + x[j].b * qw[1]);
    }
    if (T.nv >= 2) solve_sweep_n<true, UHC_DENSE_GROUP>(T.sol_back + LANE, (const char*)S + cap_of<TIER>(A).ld_delta, T.nv - 1, x);  // x <- L^-T x
    const double sd0 = LC.v0 ? S[L.sdinv + LANE] : 0.0, sd1 = LC.v1 ? S[L.sdinv + LANE + UHC_WAVE] : 0.0;
#pragma unroll
    for (int j = 0; j < UHC_DENSE_GROUP; j++) {
        if (rows[j] < 0) continue;
        double* D = Db + (slot0 + j) * A.nvp;
        const double y0 = x[j].a * sd0, y1 = x[j].b * sd1;
        if (LC.v0) D[LANE] = y0;
        if (LC.v1) D[LANE + UHC_WAVE] = y1;
        o[j].yy = wave_sum(y0 * y0 + y1 * y1);
    }
}

__device__ __forceinline__ int wave_excl_scan(int v, int* total) {
    int x = v;
#pragma unroll
    for (int o = 1; o < UHC_WAVE; o <<= 1) {
        const int y = __shfl_up(x, o);
        if (LANE >= o) x += y;
    }
    *total = __builtin_amdgcn_readlane(x, UHC_WAVE - 1);
    return x - v;
}

// Per row (lane r and r+64): J over the dof chain of the row, reference acceleration, R, warm-start
// force, and Yhat = D^-1/2 L^-T J^T stored chain-sparse (index = depth of the dof).
// Returns 0, or 1 when the packed rows do not fit this tier's Yhat storage (-> the env goes to the next tier; the last tier's storage holds
// maxefc full-length rows, so it cannot happen there).
// The dense rows (contacts between two moving bodies) of the groups first, first + stride, ...: a group = UHC_DENSE_GROUP rows through one back substitution, the whole
// wave on it (lane = dof); the rows' Yhat go to their dense slots, their scalars to dsc for the lane that owns the row.
template <int TIER>
__device__ __forceinline__ void k_dense_groups(const KernelArgs& A, double* S, const LaneConst& LC, double* Db, const int ntwo, const int first, const int stride) {
    const DevLds& L = lds_of<TIER>(A);
    const int* NI = (const int*)(S + L.ncon_nefc);
    for (int k0 = first * UHC_DENSE_GROUP; k0 < ntwo; k0 += stride * UHC_DENSE_GROUP) {
        int rr[UHC_DENSE_GROUP];
#pragma unroll
        for (int j = 0; j < UHC_DENSE_GROUP; j++) rr[j] = k0 + j < ntwo ? __builtin_amdgcn_readfirstlane(NI[4 + k0 + j]) : -1;
        DenseOut o[UHC_DENSE_GROUP];
        k_dense_rows<TIER>(A, S, rr, k0, LC, o, Db);
#pragma unroll
        for (int j = 0; j < UHC_DENSE_GROUP; j++)
            if (rr[j] >= 0 && LANE == 0) { const int k = k0 + j; S[L.dsc + 4 * k] = o[j].vel; S[L.dsc + 4 * k + 1] = o[j].jas; S[L.dsc + 4 * k + 2] = o[j].jaw; S[L.dsc + 4 * k + 3] = o[j].yy; }
    }
}

// One constraint row: Jacobian along the dof chain, reference acceleration, the half-solved row Yhat = D^-1/2 L^-T J' and its scalars.  Reads state, contacts,
// factor and the rows' enumeration (RM, RY) from LDS, writes the row's own packed entries and its own slots of the scalar arrays: rows are independent of each other,
// which is what lets the helper wave of a two-wave consumer take every second block of 64 rows (k_rows below).
template <int TIER>
__device__ __forceinline__ void k_row_one(const KernelArgs& A, const double* mb, double* S, const int r, double* Yb) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    const RowMisc* RM = (const RowMisc*)(S + L.rowMisc);
    const int* RY = (const int*)(S + L.rowY);
    {
        const RowMisc rm = RM[r];
        const int rt = RTYPE(rm.type);
        const bool two = (rm.type & ROW_TWO) != 0;
        const int last = rm.last, len = two ? 0 : T.dof_depth[last] + 1;
        const short* anc = T.dof_anc + (two ? 0 : last) * (T.maxdepth + 1);
        // (tier 4 keeps its rows in HBM: the row is built -- Jacobian, three dot products, the back substitution's read-modify-write double loop --
        //  in the lane's own 33-double strip of LDS, where the Hessian will be, and written out once; through L2 the double loop cost 390 k cycles
        //  per forward pass)
        double* Yg = Yb + RY[r];
        double* Y = (TIER == 4 && T.nv * (T.nv + 1) / 2 >= UHC_WAVE * 33) ? S + L.H + LANE * 33 : Yg;
        double pos = 0, margin = 0, diagApprox = 0, K, B, imp, floss = 0;
        if (rt == ROW_FRICTION || rt == ROW_LIMIT) {
            const double dsolimp[5] = {0.9, 0.95, 0.001, 0.5, 2.0};
            const double timeconst = fmax(0.02, 2 * T.timestep), dmax = 0.95;
            K = 1.0 / (dmax * dmax * timeconst * timeconst);
            B = 2.0 / (dmax * timeconst);
            int inv_dof = last;
            if (rt == ROW_LIMIT && rm.edge == 2) {  // ball joint: Jacobian -axis on its three dofs (the last three positions of the chain)
                const int j = rm.aux;
                double jac[3];
                margin = mb[A.o.jnt_margin + j];
                pos = ball_limit(S + L.qpos + T.jnt_qposadr[j], mb + A.o.jnt_range + 2 * j, jac);
                for (int q = 0; q < len; q++) Y[q] = 0;
                Y[len - 3] = jac[0]; Y[len - 2] = jac[1]; Y[len - 1] = jac[2];
                inv_dof = T.jnt_dofadr[j];
            } else if (rt == ROW_LIMIT) {
                const int j = rm.aux;
                const double v = S[L.qpos + T.jnt_qposadr[j]];
                margin = mb[A.o.jnt_margin + j];
                pos = rm.edge < 0 ? v - mb[A.o.jnt_range + 2 * j] : mb[A.o.jnt_range + 2 * j + 1] - v;
                for (int q = 0; q < len; q++) Y[q] = 0;
                Y[len - 1] = -(double)rm.edge;
            } else {
                floss = mb[A.o.dof_frictionloss + last];
                for (int q = 0; q < len; q++) Y[q] = 0;
                Y[len - 1] = 1;
            }
            diagApprox = mb[A.o.dof_invweight0 + inv_dof];
            imp = impedance(dsolimp, pos, margin);
        } else {
            const double* C = S + L.con + rm.aux * UHC_CON_STRIDE;
            const int bb = (rm.type & ROW_NEG) ? (int)C[19] : (int)C[20];  // the moving body of a single-chain row
            const double jsg = (rm.type & ROW_NEG) ? -1.0 : 1.0;
            const int root = T.body_rootid[bb];
            double off[3], dv[3];
            const double mu = C[14];
            for (int k = 0; k < 3; k++) off[k] = C[k] - S[L.rootcom + 3 * root + k];
            if (rt == ROW_CONTACT) for (int k = 0; k < 3; k++) dv[k] = jsg * C[3 + k];
            else {
                const double sgn = (rm.edge & 1) ? -1.0 : 1.0;
                const int td = 1 + rm.edge / 2;
                for (int k = 0; k < 3; k++) dv[k] = jsg * (C[3 + k] + sgn * mu * C[3 + 3 * td + k]);
            }
            for (int q = 0; q < len; q++) {
                const int i = anc[q];
                double cd[6], cr[3];
                for (int s = 0; s < 6; s++) cd[s] = S[L.cdof + 6 * i + s];
                cross3(cr, cd, off);
                Y[q] = dv[0] * (cd[3] + cr[0]) + dv[1] * (cd[4] + cr[1]) + dv[2] * (cd[5] + cr[2]);
            }
            pos = C[12]; margin = C[13]; K = C[15]; B = C[16]; imp = C[17];
            diagApprox = rt == ROW_CONTACT ? C[18] : C[18] + mu * mu * C[18];
            if (rt == ROW_PYR) {
                // all edges of a pyramid share R = 2 mu^2 R(first edge); first edge uses friction[0] = mu
                const double R0 = fmax(UHC_MINVAL, (1 - imp) * (C[18] + mu * mu * C[18]) / imp);
                diagApprox = -2 * mu * mu * R0;  // negative => final R given directly
            }
        }
        double vel = 0, jas = 0, jaw = 0;
        for (int q = 0; q < len; q++) {
            const int i = anc[q];
            const double j = Y[q];
            vel += j * S[L.qvel + i];
            jas += j * S[L.smooth + i];
            jaw += j * S[L.qacc + i];
        }
        double yy = 0;
        if (two) { const double* o = S + L.dsc + 4 * (rm.type >> 8); vel = o[0]; jas = o[1]; jaw = o[2]; yy = o[3]; }
        const double R = diagApprox < 0 ? -diagApprox : fmax(UHC_MINVAL, (1 - imp) * diagApprox / imp);
        const double aref = -B * vel - K * imp * (pos - margin);
        const double jar = jaw - aref, D = 1.0 / R;
        double f;
        if (rt == ROW_FRICTION) f = clampd(-D * jar, -floss, floss);
        else f = jar < 0 ? -D * jar : 0.0;
        // Y <- L^-T Y restricted to the chain, then scale by sqrt(1/D_i)
        for (int q = len - 1; q >= 1; q--) {
            const int i = anc[q];
            const double xi = Y[q];
            const int mi = T.dof_madr[i];
            for (int q2 = q - 1; q2 >= 0; q2--) Y[q2] -= S[L.LD + mi + (q - q2)] * xi;
        }
        double da = R + yy;
        for (int q = 0; q < len; q++) {
            const double y = Y[q] * S[L.sdinv + anc[q]];
            Yg[q] = y;
            da += y * y;
        }
        S[L.rowR + r] = R;
        // (aref itself is folded into b; the slot keeps the friction-loss bound of a friction-loss row, and for the unilateral rows how far the
        //  warm-start acceleration leaves them from carrying a force, in force units: D jar < 0 = carries one.  k_as_general fills its first
        //  working set with the rows that are closest.)
        S[L.rowAref + r] = rt == ROW_FRICTION ? floss : D * jar;
        S[L.rowB + r] = jas - aref;
        S[L.rowF + r] = f;
        S[L.rowDa + r] = da;             // diagonal of A + R
    }
}

template <int TIER>
__device__ __forceinline__ int k_rows(const KernelArgs& A, const double* mb, double* S, int nefc, const LaneConst& LC, double* Yb, double* Db) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    const RowMisc* RM = (const RowMisc*)(S + L.rowMisc);
    const int* NI = (const int*)(S + L.ncon_nefc);
    // packed storage: row r starts at RY[r] and is as long as its dof chain (0 for dense rows); offsets by a scan in row order
    int* RY = (int*)(S + L.rowY);
    int ytot = 0;
    for (int r0 = 0; r0 < nefc; r0 += UHC_WAVE) {
        const int r = r0 + LANE;
        int len = 0;
        if (r < nefc) { const RowMisc rm = RM[r]; len = (rm.type & ROW_TWO) ? 0 : T.dof_depth[rm.last] + 1; }
        int tot;
        const int off = wave_excl_scan(len, &tot);
        if (r < nefc) RY[r] = ytot + off;
        ytot += tot;
    }
    if (ytot + 8 > cap_of<TIER>(A).ycap) return 1;
    if (LANE == 0) RY[nefc] = ytot;  // (so that a row's length is RY[r + 1] - RY[r])
    const int ntwo = cap_of<TIER>(A).ndense > 0 ? NI[2] : 0;
#if defined(UHC_NW2)
    if constexpr (TIER == 2) {
        // The two-wave consumer (uhc_k_general_q.hip): the helper wave -- asleep in a barrier outside MPR's rounds -- takes every second GROUP of dense rows (a group is one
        // back substitution over all dofs for four rows) and the rows 64 .. 127 (uhc_mpr.h: mpr_helper, MCMD_ROWS).  Rows and groups are independent of each other: they
        // read state, contacts, factor and the enumeration from LDS and write their own slots.  The command sits in the mailbox's header on rowR, which the ROWS overwrite:
        // the helper has it in registers behind the second barrier, the dense phase leaves rowR alone, a third barrier separates the phases (a row of two bodies reads its
        // scalars from dsc), a fourth ends the pass.
        const bool split_dense = ntwo > UHC_DENSE_GROUP, split_rows = nefc > UHC_WAVE;
        if (split_dense || split_rows) {
            int* mbi = mpr_mb<TIER>(A, S).hdr;
            if (LANE == 0) {
                mbi[0] = MCMD_ROWS; mbi[1] = nefc; mbi[2] = split_dense ? ntwo : 0;
                mbi[4] = (int)((unsigned long long)mb & 0xffffffffull); mbi[5] = (int)((unsigned long long)mb >> 32);
            }
            __syncthreads();  // the helper reads the command
            __syncthreads();  // ... and has it in registers
            k_dense_groups<TIER>(A, S, LC, Db, ntwo, 0, split_dense ? 2 : 1);
            __syncthreads();  // every dense row and its scalars are there
            if (LANE < nefc) k_row_one<TIER>(A, mb, S, LANE, Yb);
            __syncthreads();  // every row is there
        } else {
            k_dense_groups<TIER>(A, S, LC, Db, ntwo, 0, 1);
            wsync();
            if (LANE < nefc) k_row_one<TIER>(A, mb, S, LANE, Yb);
        }
    } else
#endif
    {
        k_dense_groups<TIER>(A, S, LC, Db, ntwo, 0, 1);  // dense rows first (wave-cooperative); their scalars wait in dsc for the lane that owns the row
        wsync();
        for (int r = LANE; r < nefc; r += UHC_WAVE) k_row_one<TIER>(A, mb, S, r, Yb);
    }
    // the register-resident solves of the working sets (k_as_general) read rows in chunks of 8 entries, past the row's own length and into
    // the next row: everything there must be finite (it meets a zero multiplier) -- the rows are, and so is the slack after the last one
    if (LANE < 8) Yb[ytot + LANE] = 0.0;
    wsync();
    return 0;
}

// ------------------------------------------------------------------ P9 projected Gauss-Seidel on the dual (matrix-free)
// z = sum_r f_r Yhat_r  (nv vector in LDS);  (A f)_r = Yhat_r . z[chain_r].
template <int TIER>
__device__ __forceinline__ int k_pgs(const KernelArgs& A, const double* mb, double* S, int nefc, int max_sweeps, const double* Yb, const double* Db) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    const RowMisc* RM = (const RowMisc*)(S + L.rowMisc);
    const int YS = T.maxdepth + 1;
    const int* RY = (const int*)(S + L.rowY);
    double* z = S + L.z;
    const int* NI = (const int*)(S + L.ncon_nefc);
    const int ntwo = cap_of<TIER>(A).ndense > 0 ? NI[2] : 0;
    // the dof chains (which dof sits at position q of the chain that ends in dof i) are read once per row and sweep: from L2 that is a
    // ~600-cycle round trip on the row-to-row critical path (hundreds of rows x up to `iterations` sweeps).  The contacts' storage is
    // dead by now (the rows are built, the working sets -- if any -- have given up): the table is staged there.
    const short* anc_tab = T.dof_anc;
    if constexpr (TIER != 1) {
        short* dst = (short*)(S + L.con);
        if ((size_t)T.nv * YS * sizeof(short) <= (size_t)cap_of<TIER>(A).maxcon * UHC_CON_STRIDE * sizeof(double)) {
            for (int i = LANE; i < T.nv * YS; i += UHC_WAVE) dst[i] = T.dof_anc[i];
            anc_tab = dst;
            wsync();
        }
    }
    // z from the warm-start forces: dof-per-lane pull over all rows (dense rows carry last = -1 and are added from their slots)
    for (int i = LANE; i < T.nv; i += UHC_WAVE) {
        const int di = T.dof_depth[i], nd = T.dof_ndesc[i];
        double acc = 0;
        for (int r = 0; r < nefc; r++) {
            const int last = RM[r].last;
            if (last >= i && last <= i + nd) acc += S[L.rowF + r] * Yb[RY[r] + di];
        }
        for (int k = 0; k < ntwo; k++) acc += S[L.rowF + NI[4 + k]] * Db[k * A.nvp + i];
        z[i] = acc;
    }
    wsync();
    // dual cost of the warm start; fall back to zero forces if it is not an improvement
    double cost = 0;
    for (int r = LANE; r < nefc; r += UHC_WAVE) {
        const RowMisc rm = RM[r];
        const double f = S[L.rowF + r];
        double af = S[L.rowR + r] * f;
        if (rm.type & ROW_TWO) {
            const double* D = Db + (rm.type >> 8) * A.nvp;
            for (int i = 0; i < T.nv; i++) af += D[i] * z[i];
        } else {
            const int last = rm.last, len = T.dof_depth[last] + 1;
            const short* anc = anc_tab + last * YS;
            for (int q = 0; q < len; q++) af += Yb[RY[r] + q] * z[anc[q]];
        }
        cost += f * (0.5 * af + S[L.rowB + r]);
    }
    cost = wave_sum(cost);
    if (cost > 0) {
        for (int r = LANE; r < nefc; r += UHC_WAVE) S[L.rowF + r] = 0;
        for (int i = LANE; i < T.nv; i += UHC_WAVE) z[i] = 0;
    }
    wsync();
    const double scale = 1.0 / (mb[A.o.meaninertia] * (T.nv > 1 ? T.nv : 1));
    int iters = 0;
    for (int it = 0; it < max_sweeps; it++) {
        double improvement = 0;
        for (int r = 0; r < nefc; r++) {
            const RowMisc rm = RM[r];
            const bool two = (rm.type & ROW_TWO) != 0;
            const int len = RY[r + 1] - RY[r];  // 0 for dense rows
            int dof = 0;
            double y = 0, part = 0, y1 = 0;
            if (two) {  // dense row: lane = dof (two per lane)
                const double* D = Db + (rm.type >> 8) * A.nvp;
                if (LANE < T.nv) { y = D[LANE]; part = y * z[LANE]; }
                if (LANE + UHC_WAVE < T.nv) { y1 = D[LANE + UHC_WAVE]; part += y1 * z[LANE + UHC_WAVE]; }
            } else if (LANE < len) {
                dof = anc_tab[rm.last * YS + LANE];
                y = Yb[RY[r] + LANE];
                part = y * z[dof];
            }
            const double old = S[L.rowF + r], Rr = S[L.rowR + r], Arr = S[L.rowDa + r];
            const double res = wave_sum(part) + Rr * old + S[L.rowB + r];
            double f = old - res / Arr;
            if (RTYPE(rm.type) == ROW_FRICTION) { const double fl = S[L.rowAref + r]; f = clampd(f, -fl, fl); }
            else f = f < 0 ? 0.0 : f;
            double delta = f - old;
            double change = 0.5 * delta * delta * Arr + delta * res;
            if (change > 1e-10) { f = old; delta = 0; change = 0; }
            improvement -= change;
            if (delta != 0) {
                if (two) {
                    if (LANE < T.nv) z[LANE] += delta * y;
                    if (LANE + UHC_WAVE < T.nv) z[LANE + UHC_WAVE] += delta * y1;
                } else if (LANE < len) z[dof] += delta * y;
                if (LANE == 0) S[L.rowF + r] = f;
            }
            wsync();
        }
        iters = it + 1;
        if (improvement * scale < T.tolerance) break;
    }
    return iters;
}


// ------------------------------------------------------------------ FAST path (nefc <= 64): row-per-lane, A in registers
// Lane r owns constraint row r: its scalars live in registers, its Yhat row in LDS (variable length,
// packed), and row r of the Delassus matrix A = Yhat Yhat^T + diag(R) in 64 VGPR pairs.  A PGS row
// update is then: lane i computes its own step, one broadcast of delta, one FMA per lane on the
// residual vector -- no reduction and no LDS traffic inside the sweep.
struct FastRow { int type, last, len, yoff, two /* dense slot or -1 */; double R, b, f, floss, diag; };


// returns 0 on success, 1 if the packed Yhat rows do not fit (-> the env is redone by the general kernel)
// The row's Yhat = D^-1/2 L^-T J^T entries along its dof chain are built in REGISTERS (Y[q], q = position on the chain,
// compile-time indices): Jacobian, the three J.v products, the back substitution and the D^-1/2 scaling never round-trip
// through LDS (a lane-serial read-modify-write chain through LDS costs ~100 cycles per update with one wave per SIMD).
// The unrolled loops skip, uniformly, the chain positions beyond the longest row of the wave.  T.chain[dof][q] packs
// (q-th dof of the chain | LDS byte address of that dof's L row << 16); positions past the chain end point at safe
// finite data and meet Y = 0.  The finished rows are also stored to LDS (packed) for the A build of the other lanes.
#define UHC_YM 32
template <bool DENSE>
__device__ __forceinline__ int k_rows_fast(const KernelArgs& A, const double* mb, double* S, int& nefc, FastRow& row, double (&Y)[UHC_YM], const LaneConst& LC, int* ytot) {
    const DevTopo& T = A.t;
    const DevLds& L = A.lf;
    const RowMisc* RM = (const RowMisc*)(S + L.rowMisc);
    const char* SB = (const char*)S;
    const int r = LANE;
    bool valid = r < nefc;
    RowMisc rm = {0, 0, 0, 0};
    if (valid) rm = RM[r];
    const bool two = DENSE && (rm.type & ROW_TWO) != 0;  // dense row (two moving bodies): no chain, built wave-cooperatively below
    const bool jneg = (rm.type & ROW_NEG) != 0;
    row.two = two ? (rm.type >> 8) : -1;
    rm.type = RTYPE(rm.type);
    if (two) rm.last = 0;
    row.type = rm.type; row.last = two ? -1 : rm.last;
    row.len = (valid && !two) ? T.dof_depth[rm.last] + 1 : 0;
    int total, status = 0;
    row.yoff = wave_excl_scan(row.len, &total);
    *ytot = total;
    row.R = 1; row.b = 0; row.f = 0; row.floss = 0; row.diag = 1;
    if (total + 8 > A.cf.ycap) {
        if (!A.truncate) return 1;
        // keep the leading rows whose packed entries fit, cut back to the start of a pyramid so its edges stay together
        const unsigned long long fit = __builtin_amdgcn_ballot_w64(valid && row.yoff + row.len + 8 <= A.cf.ycap);
        int nfit = ~fit == 0ull ? UHC_WAVE : __ffsll((long long)~fit) - 1;
        if (nfit < nefc && nfit > 0) {
            const unsigned long long starts = __builtin_amdgcn_ballot_w64(valid && !(rm.type == ROW_PYR && rm.edge != 0));
            const unsigned long long upto = starts & (nfit >= 63 ? ~0ull : ((2ull << nfit) - 1ull));
            nfit = upto ? 63 - __builtin_clzll(upto) : 0;
        }
        nefc = nfit;
        status = 2;
        valid = r < nefc;
        if (!valid) { rm.type = 0; row.type = 0; row.len = 0; }
        if (nefc == 0) return 2;
    }
    const int len = row.len;
    int maxlen = len;  // wave maximum (uniform)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o));
    maxlen = __builtin_amdgcn_readfirstlane(maxlen);
    unsigned int ch[UHC_YM];
    {
        const unsigned int* chain = T.chain + (size_t)rm.last * UHC_YM;
        static_for<0, UHC_YM>([&](auto qc) __attribute__((always_inline)) { constexpr int q = decltype(qc)::value; ch[q] = chain[q]; });
    }
    const bool is_con = valid && (rm.type == ROW_CONTACT || rm.type == ROW_PYR);
    double pos = 0, margin = 0, diagApprox = 0, K = 0, B = 0, imp = 1, floss = 0, unit = 0;
    double unit1 = 0, unit2 = 0;  // ball-joint limit rows: the Jacobian entries of chain positions len - 2 and len - 3 (DENSE instantiations only:
                                  // a model with limited ball joints is launched on them, KernelArgs::ball_limits)
    double off[3] = {0, 0, 0}, dv[3] = {0, 0, 0};
    if (valid && !is_con) {
        const double dsolimp[5] = {0.9, 0.95, 0.001, 0.5, 2.0};
        const double timeconst = fmax(0.02, 2 * T.timestep), dmax = 0.95;
        K = 1.0 / (dmax * dmax * timeconst * timeconst);
        B = 2.0 / (dmax * timeconst);
        int inv_dof = rm.last;
        if (DENSE && rm.type == ROW_LIMIT && rm.edge == 2) {
            const int j = rm.aux;
            double jac[3];
            margin = mb[A.o.jnt_margin + j];
            pos = ball_limit(S + L.qpos + T.jnt_qposadr[j], mb + A.o.jnt_range + 2 * j, jac);
            unit2 = jac[0]; unit1 = jac[1]; unit = jac[2];
            inv_dof = T.jnt_dofadr[j];
        } else if (rm.type == ROW_LIMIT) {
            const int j = rm.aux;
            const double v = S[L.qpos + T.jnt_qposadr[j]];
            margin = mb[A.o.jnt_margin + j];
            pos = rm.edge < 0 ? v - mb[A.o.jnt_range + 2 * j] : mb[A.o.jnt_range + 2 * j + 1] - v;
            unit = -(double)rm.edge;
        } else {
            floss = mb[A.o.dof_frictionloss + rm.last];
            unit = 1;
        }
        diagApprox = mb[A.o.dof_invweight0 + inv_dof];
        imp = impedance(dsolimp, pos, margin);
    } else if (is_con) {
        const double* C = S + L.con + rm.aux * UHC_CON_STRIDE;
        const int bb = jneg ? (int)C[19] : (int)C[20];  // the moving body of a single-chain row
        const double jsg = jneg ? -1.0 : 1.0;
        const int root = T.body_rootid[bb];
        const double mu = C[14];
        for (int k = 0; k < 3; k++) off[k] = C[k] - S[L.rootcom + 3 * root + k];
        if (rm.type == ROW_CONTACT) for (int k = 0; k < 3; k++) dv[k] = jsg * C[3 + k];
        else {
            const double sgn = (rm.edge & 1) ? -1.0 : 1.0;
            const int td = 1 + rm.edge / 2;
            for (int k = 0; k < 3; k++) dv[k] = jsg * (C[3 + k] + sgn * mu * C[3 + 3 * td + k]);
        }
        pos = C[12]; margin = C[13]; K = C[15]; B = C[16]; imp = C[17];
        diagApprox = rm.type == ROW_CONTACT ? C[18] : C[18] + mu * mu * C[18];
        if (rm.type == ROW_PYR) {
            // all edges of a pyramid share R = 2 mu^2 R(first edge); first edge uses friction[0] = mu
            const double R0 = fmax(UHC_MINVAL, (1 - imp) * (C[18] + mu * mu * C[18]) / imp);
            diagApprox = -2 * mu * mu * R0;  // negative => final R given directly
        }
    }
    // ---- J along the chain, and J.qvel, J.qacc_smooth, J.qacc_warmstart
    double vel = 0, jas = 0, jaw = 0;
    if constexpr (DENSE) {  // dense rows: wave-cooperative (lane = dof), their scalars go to the lane that owns the row
        const int* NI = (const int*)(S + L.ncon_nefc);
        const int ntwo = NI[2];
        for (int k0 = 0; k0 < ntwo; k0 += UHC_DENSE_GROUP) {
            int rr[UHC_DENSE_GROUP];
#pragma unroll
            for (int j = 0; j < UHC_DENSE_GROUP; j++) {
                rr[j] = k0 + j < ntwo ? __builtin_amdgcn_readfirstlane(NI[4 + k0 + j]) : -1;
                if (rr[j] >= nefc) rr[j] = -1;  // dropped by the truncation above
            }
            DenseOut o[UHC_DENSE_GROUP];
            k_dense_rows<1>(A, S, rr, k0, LC, o, S + L.dense);
#pragma unroll
            for (int j = 0; j < UHC_DENSE_GROUP; j++) if (LANE == rr[j]) { vel = o[j].vel; jas = o[j].jas; jaw = o[j].jaw; }
        }
    }
    // chain positions in groups of four: one uniform test per group (a branch costs 25-60 cycles here) and the LDS reads of four
    // positions in flight together; positions past a row's own length yield y = 0, chain slots past the chain hold dof 0
    static_for<0, UHC_YM / 4>([&](auto gc) __attribute__((always_inline)) {
      constexpr int g4 = decltype(gc)::value;
      static_for<0, 4>([&](auto hc) __attribute__((always_inline)) { Y[4 * g4 + decltype(hc)::value] = 0.0; });
      if (4 * g4 < maxlen) static_for<0, 4>([&](auto hc) __attribute__((always_inline)) {
        constexpr int q = 4 * g4 + decltype(hc)::value;
        {
            const int i = ch[q] & 0xffff;
            double cd[6], cr[3];
            for (int t = 0; t < 6; t++) cd[t] = S[L.cdof + 6 * i + t];
            cross3(cr, cd, off);
            const double yc = dv[0] * (cd[3] + cr[0]) + dv[1] * (cd[4] + cr[1]) + dv[2] * (cd[5] + cr[2]);
            double yu = q == len - 1 ? unit : 0.0;
            if constexpr (DENSE) yu = q == len - 2 ? unit1 : (q == len - 3 ? unit2 : yu);
            const double y = q < len ? (is_con ? yc : yu) : 0.0;
            Y[q] = y;
            vel = fma(y, S[L.qvel + i], vel);
            jas = fma(y, S[L.smooth + i], jas);
            jaw = fma(y, S[L.qacc + i], jaw);
        }
      });
    });
    if (valid) {
        const double R = diagApprox < 0 ? -diagApprox : fmax(UHC_MINVAL, (1 - imp) * diagApprox / imp);
        const double aref = -B * vel - K * imp * (pos - margin);
        const double jar = jaw - aref, D = 1.0 / R;
        row.f = rm.type == ROW_FRICTION ? clampd(-D * jar, -floss, floss) : (jar < 0 ? -D * jar : 0.0);
        row.R = R; row.b = jas - aref; row.floss = floss;
    }
    // ---- Y <- L^-T Y along the chain: for q descending, every earlier position q2: Y[q2] -= L[anc_q][anc_q2] Y[q]
    static_for<1, UHC_YM>([&](auto qc) __attribute__((always_inline)) {
        constexpr int q = UHC_YM - decltype(qc)::value;  // UHC_YM-1 .. 1
        if (q < maxlen) {
            const double xi = Y[q];
            const unsigned int mi = ch[q] >> 16;
            static_for<0, q>([&](auto pc) __attribute__((always_inline)) {
                constexpr int q2 = decltype(pc)::value;
                Y[q2] = fma(-lds_at(SB, mi + 8u * (unsigned)(q - q2)), xi, Y[q2]);
            });
        }
    });
    if (LANE < 8) S[L.Y + total + LANE] = 0.0;  // the A build reads rows in chunks of 8: finite slack after the last row
    double* Yst = S + L.Y + row.yoff;
    static_for<0, UHC_YM / 4>([&](auto gc) __attribute__((always_inline)) {
        constexpr int g4 = decltype(gc)::value;
        if (4 * g4 < maxlen) static_for<0, 4>([&](auto hc) __attribute__((always_inline)) {
            constexpr int q = 4 * g4 + decltype(hc)::value;
            Y[q] *= S[L.sdinv + (ch[q] & 0xffff)];
            if (q < len) Yst[q] = Y[q];
        });
    });
    wsync();
    return status;
}

__device__ __forceinline__ double max_neg(double a, double b) {  // max(a, -b): one VOP3 with a source modifier
    double r;
    asm("v_max_f64 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// N = rows swept (multiple of 8 >= nefc; padding rows have f = w = 0 and zero off-diagonals, so their steps are no-ops).
// State per lane (= per row r): f_r and w_r = -res_r / A_rr, the step to the unconstrained 1-D minimiser of the row.
// Brow[i] = A[r][i] / A[r][r] (unit diagonal).  Row step i: delta_i = max(w_i, -f_i) (the step clipped at f >= 0) is
// computed by every lane for ITS row, lane i's value is broadcast (two v_readlane), every w moves by -delta_i * Brow[i]
// (one FMA; lane i's own w drops by delta_i through the unit diagonal) and lane i alone adds delta_i to its f:
// f += delta * e_i with a one-hot (1.0 in lane i) that moves up one lane per step by a DPP wave shift -- no exec
// mask writes (an s_mov to exec between dependent VALU ops costs ~20 cycles each way with one wave per SIMD).
// Serial chain per row: max -> readlane -> FMA; no reduction and no LDS traffic inside the sweep.
// The dual cost 1/2 f'Af + f'b = sum_r 1/2 f_r (b_r + res_r) is evaluated once per sweep; the sweep's improvement is
// the difference of two consecutive costs (the reference adds up the per-row decreases, which telescope to the same
// number).  Its "cost went up by > 1e-10 -> revert the step" guard is omitted: in exact arithmetic a projected 1-D
// minimisation never raises the cost (the general kernel and the oracle keep the guard).
template <int N, bool FRIC, bool FIXED = false>  // FIXED: exactly `iterations` sweeps, no cost evaluation (pre-sweeps of the active-set solver)
__device__ __forceinline__ int pgs_sweeps(double (&Brow)[UHC_WAVE], double& f, double& w, double diag, double b,
                                          bool fric, double floss, int iterations, double scale, double tolerance) {
    int iters = 0;
    double cost_prev = FIXED ? 0.0 : wave_sum(0.5 * f * fma(-w, diag, b));
    const int hot0 = LANE == 0 ? 0x3FF00000 : 0;  // high word of 1.0
    for (int it = 0; it < iterations; it++) {
        int hot = hot0;
        // a lane's own force changes only at its own step, so the force it needs there is the one it had when the sweep
        // began: the steps read this snapshot and the commits into f stay off the max -> readlane -> FMA chain
        const double fs = f;
        static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            double delta;
            if (FRIC) {
                const double lo = fric ? -floss - fs : -fs, hi = fric ? floss - fs : 1e300;
                delta = fmin(fmax(w, lo), hi);
            } else {
                delta = max_neg(w, fs);
            }
            const double di = bcast(delta, i);
            w = fma(-di, Brow[i], w);
            f = fma(delta, __hiloint2double(hot, 0), f);
            hot = __builtin_amdgcn_update_dpp(0, hot, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
        });
        iters = it + 1;
        if (FIXED) continue;
        const double cost = wave_sum(0.5 * f * fma(-w, diag, b));
        const double improvement = cost_prev - cost;
        cost_prev = cost;
        if (improvement * scale < tolerance) break;
    }
    return iters;
}

// Exact solve of the dual QP  min 1/2 f'Af + f'b, f >= 0  by block principal pivoting (UhcModelDesc.solver == 1; the oracle's
// orc_solve_active_set is the same algorithm).  F = rows allowed a positive force, start F = the rows that UHC_AS_PRESWEEPS Gauss-Seidel
// sweeps from f = 0 leave with a force (k_pgs_fast).  One iteration:
//   1. W <- A (from the AGPR-parked rows), c <- -b; Gaussian elimination over the steps k in F, lane = row: the pivot row k is
//      broadcast entry by entry (two v_readlane each), every row i > k of F subtracts l_ik times it.  Rows outside F keep l = 0,
//      i.e. behave as identity rows; their columns are updated but never read.  No pivoting (A_FF is positive definite).
//   2. back substitution with the pivots' reciprocals kept by their lanes: f_k = (c_k - sum_{j>k in F} U_kj f_j) / U_kk, f = 0 outside F;
//   3. y = A f + b; a row is infeasible if f < 0 (inside F) or y < 0 (outside); none -> optimum (KKT holds exactly);
//   4. flip all infeasible rows while their count keeps falling (three grace rounds), else only the highest one (finite).
// Everything is unrolled over static register indices; a step whose row is not in F, and column chunks beyond nefc, are skipped
// by uniform branches.  Returns the number of factorisations, or -1 (pivot breakdown / no convergence: the caller runs the sweeps).
#define UHC_AS_MAXIT 64
#define UHC_AS_PRESWEEPS 16  // measured on the bench workload: 2 / 4 / 6 / 8 / 12 / 16 / 20 / 24 / 32 sweeps -> 3.80 / 3.69 / 3.64 / 3.62 / 3.58 / 3.55 / 3.56 / 3.57 / 3.59 ms per launch
                             // (2.57 ... 1.39 factorisations per solve: a sweep is cheaper than the factorisation rounds it saves, up to a point)
// NC = nefc rounded up to 8: the unrolled loops stop there.  (Testing the column range at run time instead -- a uniform branch per
// chunk of 8 columns -- costs ~70 cycles per branch with one wave per SIMD, 40% of the elimination: tools/ubench/elim.hip.)
template <int NC>
__device__ __forceinline__ int as_solve(const int (&Alo)[UHC_WAVE], const int (&Ahi)[UHC_WAVE], int nefc, double b, bool inF0, double& f_out PROF_ARGS) {
    const bool valid = LANE < nefc;
    bool inF = valid && inF0;
    int best = nefc + 1, grace = 3;
    double f = 0.0;
    for (int it = 1; it <= UHC_AS_MAXIT; it++) {
        const unsigned long long Fm = __builtin_amdgcn_ballot_w64(inF);
        double W[NC];
        static_for<0, NC>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            W[s] = __hiloint2double(agpr_get(Ahi[s]), agpr_get(Alo[s]));
        });
        PROF(25)
        double c = -b, mypinv = 0.0;
        bool broke = false;
        int ln = LANE;
        asm volatile("" : "+v"(ln));  // opaque per iteration: keeps the (lane > k) masks from being hoisted into (spilled) SGPR pairs
        // the membership tests are scalar bit tests of a mask copy that is opaque per loop: shared across the three loops the compiler
        // turns them into lane-mask booleans, spills those to VGPR lanes and reloads one per step
        unsigned long long Fe = Fm;
        asm volatile("" : "+s"(Fe));
        static_for<0, NC>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            if ((Fe >> k) & 1ull) {
                const double pk = bcast(W[k], k);
                broke |= !(pk > UHC_MINVAL);
                const double pinv = rcp_newton(pk);
                mypinv = ln == k ? pinv : mypinv;
                const double l = (ln > k && inF) ? W[k] * pinv : 0.0;
                c = fma(-l, bcast(c, k), c);
                static_for<k + 1, NC>([&](auto jc) __attribute__((always_inline)) {
                    constexpr int j = decltype(jc)::value;
                    W[j] = fma(-l, bcast(W[j], k), W[j]);  // (an LDS round trip of the pivot row instead of the readlanes is 2x slower)
                });
            }
        });
        if (broke) return -1;
        PROF(26)
        double acc = c;
        f = 0.0;
        unsigned long long Fb = Fm;
        asm volatile("" : "+s"(Fb));
        static_for<0, NC>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = NC - 1 - decltype(kc)::value;
            if ((Fb >> k) & 1ull) {
                const double xk = bcast(acc * mypinv, k);
                f = ln == k ? xk : f;
                acc = fma(-W[k], xk, acc);
            }
        });
        PROF(27)
        double y = b;
        unsigned long long Fy = Fm;
        asm volatile("" : "+s"(Fy));
        static_for<0, NC>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            if ((Fy >> s) & 1ull) y = fma(__hiloint2double(agpr_get(Ahi[s]), agpr_get(Alo[s])), bcast(f, s), y);
        });
        PROF(28)
        const bool bad = valid && (inF ? f < 0.0 : y < 0.0);
        const unsigned long long Bm = __builtin_amdgcn_ballot_w64(bad);
        if (Bm == 0ull) { f_out = f; return it; }
        const int nbad = __builtin_popcountll(Bm);
        bool all = true;
        if (nbad < best) { best = nbad; grace = 3; }
        else if (grace > 0) grace--;
        else all = false;
        const int last = 63 - __builtin_clzll(Bm);
        if (bad && (all || LANE == last)) inF = !inF;
    }
    return -1;
}

// PGS with A in registers; returns the sweep count.  On exit S[L.z] = sum_r f_r Yhat_r.
// LT: the tier whose LDS layout the rows live in (the general / large tiers solve working sets of <= 64 of their rows with the same code).
// SL[0 .. nslot): lane that holds the dense row of slot k (an entry outside [0, nefc) = that row is not part of this solve).
template <int LT, bool DENSE>
__device__ __forceinline__ int k_pgs_fast(const KernelArgs& A, const double* mb, double* S, int nefc, FastRow& row, const double (&Y)[UHC_YM], const LaneConst& LC,
                                          const int* SL, int nslot PROF_ARGS) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<LT>(A);
    const bool valid = LANE < nefc;
    int ntwo = 0;  // dense rows taking part (wave-uniform)
    if constexpr (DENSE) {
        for (int k = 0; k < nslot; k++) ntwo += (unsigned)SL[k] < (unsigned)nefc;
        ntwo = __builtin_amdgcn_readfirstlane(ntwo);
    }
    // ---- Delassus columns of the dense rows: A[l][c] = Yhat_l . Yhat_c for every lane l, kept in LDS (dcol[slot][lane]): lane l reads its
    //      entry when column c comes up below, and -- A being symmetric -- the lane that owns dense row c reads its whole ROW from there.
    if (DENSE && ntwo > 0) {
        unsigned int cdq[UHC_YM];
        const unsigned int* chain = T.chain + (size_t)(row.last >= 0 ? row.last : 0) * UHC_YM;
        static_for<0, UHC_YM>([&](auto qc) __attribute__((always_inline)) { constexpr int q = decltype(qc)::value; cdq[q] = chain[q] & 0xffffu; });
        for (int k = 0; k < nslot; k++) {
            if ((unsigned)__builtin_amdgcn_readfirstlane(SL[k]) >= (unsigned)nefc) continue;
            const double* Dk = S + L.dense + k * A.nvp;
            double acc = 0.0;
            static_for<0, UHC_YM>([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value;
                acc = fma(Y[q], Dk[cdq[q]], acc);  // Y is zero past the row's own chain (and for dense rows)
            });
            S[L.dcol + k * UHC_WAVE + LANE] = valid ? acc : 0.0;  // (dense lanes: 0 for now)
        }
        wsync();
        // dense x dense entries: one wave reduction per pair of slots (lane = dof), written to both lanes' columns
        for (int k = 0; k < nslot; k++) {
            const int lk = __builtin_amdgcn_readfirstlane(SL[k]);
            if ((unsigned)lk >= (unsigned)nefc) continue;
            const double* Dk = S + L.dense + k * A.nvp;
            const double dk0 = LC.v0 ? Dk[LANE] : 0.0, dk1 = LC.v1 ? Dk[LANE + UHC_WAVE] : 0.0;
            for (int m = 0; m <= k; m++) {
                const int lm = __builtin_amdgcn_readfirstlane(SL[m]);
                if ((unsigned)lm >= (unsigned)nefc) continue;
                const double* Dm = S + L.dense + m * A.nvp;
                const double v = wave_sum(dk0 * (LC.v0 ? Dm[LANE] : 0.0) + dk1 * (LC.v1 ? Dm[LANE + UHC_WAVE] : 0.0));
                if (LANE == 0) { S[L.dcol + k * UHC_WAVE + lm] = v; S[L.dcol + m * UHC_WAVE + lk] = v; }
            }
        }
        wsync();
    }
    // ---- A[r][s] = sum over the common part of the two dof chains, entirely in registers: row s is broadcast from lane
    //      the packed LDS rows (same address in all lanes), the own row is pre-masked to the common chain prefix.  Rows of one body are
    //      adjacent and share that prefix length, so the masked copy is rebuilt only when the body changes.
    unsigned int ncp[UHC_WAVE / 4];  // common-prefix length of this lane's chain with every row's chain, 4 per register
    {
        int nraw[UHC_WAVE];
        static_for<0, UHC_WAVE>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            nraw[s] = 0;
            if (s < nefc) {
                const int ls = __builtin_amdgcn_readlane(row.last, s);
                nraw[s] = (valid && row.last >= 0 && ls >= 0) ? (int)A.t.dof_ncommon[row.last * T.nv + ls] : 0;
            }
        });
        static_for<0, UHC_WAVE / 4>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            ncp[j] = (unsigned)nraw[4 * j] | ((unsigned)nraw[4 * j + 1] << 8) | ((unsigned)nraw[4 * j + 2] << 16) | ((unsigned)nraw[4 * j + 3] << 24);
        });
    }
    double Ym[UHC_YM];
    int Alo[UHC_WAVE], Ahi[UHC_WAVE];  // AGPR-parked A[r][s]
    double diag = 1.0;
    int prev = -1;
    static_for<0, UHC_YM>([&](auto qc) __attribute__((always_inline)) { Ym[decltype(qc)::value] = 0.0; });
    auto build_A = [&](auto dense_c) __attribute__((always_inline)) {
    constexpr bool DCOL = decltype(dense_c)::value;  // some row of this env is dense
    static_for<0, UHC_WAVE>([&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        double acc = 0.0;
        if (s < nefc) {
            const int ls = __builtin_amdgcn_readlane(row.last, s);
            const int lens = __builtin_amdgcn_readlane(row.len, s);
            const int two_s = DCOL ? __builtin_amdgcn_readlane(row.two, s) : -1;
            if (DCOL && two_s >= 0) acc = S[L.dcol + two_s * UHC_WAVE + LANE];
            else {
            const double* Ys = S + L.Y + __builtin_amdgcn_readlane(row.yoff, s);  // row s in LDS: one broadcast read per entry
            if (ls != prev) {
                prev = ls;
                const int nc = (ncp[s / 4] >> (8 * (s % 4))) & 0xff;
                static_for<0, UHC_YM>([&](auto qc) __attribute__((always_inline)) {
                    constexpr int q = decltype(qc)::value;
                    Ym[q] = q < nc ? Y[q] : 0.0;
                });
            }
            // chunks of 8 chain positions, skipped uniformly beyond row s's length; inside a chunk no tests are needed:
            // Ym is zero past the common prefix (<= lens) and every row's registers are zero past its own length
            // (all loads of a row up front in two halves of 16 with two accumulators: A-build -8 %, but the rest of the kernel +1.5 %
            //  from the changed register allocation: not kept)
            static_for<0, UHC_YM / 8>([&](auto cc) __attribute__((always_inline)) {
                constexpr int c = decltype(cc)::value;
                if (8 * c < lens) {
                    double y8[8];
                    static_for<0, 8>([&](auto qc) __attribute__((always_inline)) { constexpr int j = decltype(qc)::value; y8[j] = Ys[8 * c + j]; });
                    static_for<0, 8>([&](auto qc) __attribute__((always_inline)) {
                        constexpr int j = decltype(qc)::value;
                        acc = fma(Ym[8 * c + j], y8[j], acc);
                    });
                }
            });
            if (DCOL && row.two >= 0) acc = S[L.dcol + row.two * UHC_WAVE + s];  // the dense row's own lane: its row of A by symmetry
            }
            if (DCOL && !valid) acc = 0.0;
            if (s == LANE) { acc += row.R; diag = acc; }
        }
        agpr_put(Alo[s], __double2loint(acc));
        agpr_put(Ahi[s], __double2hiint(acc));
    });
    };
    if (DENSE && ntwo > 0) build_A(std::true_type{}); else build_A(std::false_type{});
    if (!valid) diag = 1.0;
    PROF(10)
    const bool any_fric = wave_or(row.type == ROW_FRICTION ? 1 : 0) != 0;
    int iters = -1;
    if (T.solver == 1 && !any_fric) {
        double fx = 0.0;
        const double bq = valid ? row.b : 0.0;
        // initial guess of the free set: the rows UHC_AS_PRESWEEPS Gauss-Seidel sweeps from f = 0 leave with a force (cuts the
        // factorisation rounds from ~3.2 to ~1.7)
        // (starting from the warm-start forces' support instead was measured: 2.14 factorisations instead of 2.01 and costlier
        //  rounds, 3.60 -> 3.88 ms per launch)
        bool f0;
        {
            double Br[UHC_WAVE];
            const double dinv0 = 1.0 / diag;
            static_for<0, UHC_WAVE>([&](auto sc) __attribute__((always_inline)) {
                constexpr int s = decltype(sc)::value;
                Br[s] = (s == LANE) ? 1.0 : __hiloint2double(agpr_get(Ahi[s]), agpr_get(Alo[s])) * dinv0;
            });
            double fp = 0.0, wp = valid ? -bq * dinv0 : 0.0;
            switch ((nefc + 7) >> 3) {
                case 1: pgs_sweeps<8, false, true>(Br, fp, wp, diag, bq, false, 0.0, UHC_AS_PRESWEEPS, 0.0, 0.0); break;
                case 2: pgs_sweeps<16, false, true>(Br, fp, wp, diag, bq, false, 0.0, UHC_AS_PRESWEEPS, 0.0, 0.0); break;
                case 3: pgs_sweeps<24, false, true>(Br, fp, wp, diag, bq, false, 0.0, UHC_AS_PRESWEEPS, 0.0, 0.0); break;
                case 4: pgs_sweeps<32, false, true>(Br, fp, wp, diag, bq, false, 0.0, UHC_AS_PRESWEEPS, 0.0, 0.0); break;
                case 5: pgs_sweeps<40, false, true>(Br, fp, wp, diag, bq, false, 0.0, UHC_AS_PRESWEEPS, 0.0, 0.0); break;
                case 6: pgs_sweeps<48, false, true>(Br, fp, wp, diag, bq, false, 0.0, UHC_AS_PRESWEEPS, 0.0, 0.0); break;
                case 7: pgs_sweeps<56, false, true>(Br, fp, wp, diag, bq, false, 0.0, UHC_AS_PRESWEEPS, 0.0, 0.0); break;
                default: pgs_sweeps<64, false, true>(Br, fp, wp, diag, bq, false, 0.0, UHC_AS_PRESWEEPS, 0.0, 0.0); break;
            }
            f0 = fp > 0.0;
        }
        PROF(29)
        switch ((nefc + 7) >> 3) {
            case 1: iters = as_solve<8>(Alo, Ahi, nefc, bq, f0, fx PROF_PASS); break;
            case 2: iters = as_solve<16>(Alo, Ahi, nefc, bq, f0, fx PROF_PASS); break;
            case 3: iters = as_solve<24>(Alo, Ahi, nefc, bq, f0, fx PROF_PASS); break;
            case 4: iters = as_solve<32>(Alo, Ahi, nefc, bq, f0, fx PROF_PASS); break;
            case 5: iters = as_solve<40>(Alo, Ahi, nefc, bq, f0, fx PROF_PASS); break;
            case 6: iters = as_solve<48>(Alo, Ahi, nefc, bq, f0, fx PROF_PASS); break;
            case 7: iters = as_solve<56>(Alo, Ahi, nefc, bq, f0, fx PROF_PASS); break;
            default: iters = as_solve<64>(Alo, Ahi, nefc, bq, f0, fx PROF_PASS); break;
        }
        if (iters > 0) row.f = fx;
    }
    if (iters < 0) {
    double Arow[UHC_WAVE];
    static_for<0, UHC_WAVE>([&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        Arow[s] = __hiloint2double(agpr_get(Ahi[s]), agpr_get(Alo[s]));
    });
    const double dinvA = 1.0 / diag;
    // ---- residual of the warm start, dual cost test
    double f = row.f, res = row.b;
    static_for<0, UHC_WAVE>([&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        if (s < nefc) res += Arow[s] * bcast(f, s);
    });
    double cost = valid ? f * (0.5 * (res - row.b) + row.b) : 0.0;
    cost = wave_sum(cost);
    if (cost > 0) { f = 0; res = row.b; }
    double w = valid ? -res * dinvA : 0.0;
    if (!valid) f = 0.0;
    static_for<0, UHC_WAVE>([&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        Arow[s] = (s == LANE) ? 1.0 : Arow[s] * dinvA;  // Arow now holds Brow (unit diagonal)
    });
    const double rb = valid ? row.b : 0.0;
    const double scale = 1.0 / (mb[A.o.meaninertia] * (T.nv > 1 ? T.nv : 1));
    const bool fric = row.type == ROW_FRICTION;
    if (any_fric) iters = pgs_sweeps<UHC_WAVE, true>(Arow, f, w, diag, rb, fric, row.floss, T.iterations, scale, T.tolerance);
    else switch ((nefc + 7) >> 3) {
        case 1: iters = pgs_sweeps<8, false>(Arow, f, w, diag, rb, false, 0.0, T.iterations, scale, T.tolerance); break;
        case 2: iters = pgs_sweeps<16, false>(Arow, f, w, diag, rb, false, 0.0, T.iterations, scale, T.tolerance); break;
        case 3: iters = pgs_sweeps<24, false>(Arow, f, w, diag, rb, false, 0.0, T.iterations, scale, T.tolerance); break;
        case 4: iters = pgs_sweeps<32, false>(Arow, f, w, diag, rb, false, 0.0, T.iterations, scale, T.tolerance); break;
        case 5: iters = pgs_sweeps<40, false>(Arow, f, w, diag, rb, false, 0.0, T.iterations, scale, T.tolerance); break;
        case 6: iters = pgs_sweeps<48, false>(Arow, f, w, diag, rb, false, 0.0, T.iterations, scale, T.tolerance); break;
        case 7: iters = pgs_sweeps<56, false>(Arow, f, w, diag, rb, false, 0.0, T.iterations, scale, T.tolerance); break;
        default: iters = pgs_sweeps<64, false>(Arow, f, w, diag, rb, false, 0.0, T.iterations, scale, T.tolerance); break;
    }
    row.f = f;
    if (LT != 1 && T.solver == 1 && !any_fric) return -1;  // a working set of the general kernel that is not solved exactly is of no use to it
    }
    const double f = row.f;
    PROF(11)
    // ---- z = sum_r f_r Yhat_r for qacc = qacc_smooth + L^-1 D^-1/2 z: every row scatters its <= 31 chain entries into the
    //      per-dof accumulators with LDS float64 atomics (rows of one body hit the same addresses; the LDS unit serialises them)
    for (int i = LANE; i < T.nv; i += UHC_WAVE) S[L.z + i] = 0.0;
    wsync();
    if (valid && row.last >= 0) {
        const unsigned int* chain = T.chain + (size_t)row.last * UHC_YM;
        const double* Yr = S + L.Y + row.yoff;
        static_for<0, UHC_YM>([&](auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
            if (q < row.len) __hip_atomic_fetch_add(S + L.z + (chain[q] & 0xffffu), f * Yr[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        });
    }
    wsync();
    if (DENSE && ntwo > 0) {  // dense rows: lane = dof
        for (int k = 0; k < nslot; k++) {
            const int ln = __builtin_amdgcn_readfirstlane(SL[k]);
            if ((unsigned)ln >= (unsigned)nefc) continue;
            const double fk = bcast(f, ln);
            const double* Dk = S + L.dense + k * A.nvp;
            if (LC.v0) S[L.z + LANE] += fk * Dk[LANE];
            if (LC.v1) S[L.z + LANE + UHC_WAVE] += fk * Dk[LANE + UHC_WAVE];
        }
        wsync();
    }
    return iters;
}

// ------------------------------------------------------------------ general kernel: exact solve by working sets
// Up to 128 rows do not fit the register-resident Delassus matrix (lane = row), but the rows that carry a force at the optimum are
// rarely more than 64.  Solve the QP restricted to a working set C of <= 64 rows exactly (compacted into the lanes, same code as the
// fast kernel: A_CC in registers, block principal pivoting), evaluate y = A f + b on the rows outside C (matrix-free: Yhat_r . z + b_r),
// add the violated ones (y < 0), drop the rows of C that ended without a force, repeat.  Every new working set contains the support of
// the current f, so the dual cost falls strictly from one round to the next: no cycling.  C starts as the rows with a positive
// warm-start force.
// An island with 64 force-carrying rows and more that want in is beyond one register-resident solve: it is taken in WINDOWS of 64 candidate
// rows (block coordinate descent on the same QP).  The island's force-carrying rows outside the window keep their force and enter the
// window's sub-QP through its right-hand side (b_C + A_C,fixed f_fixed = b_C + Yhat_C . zfix), the window is solved exactly, and the next
// window starts at the row after the last one taken, cyclically.  Every window solve minimises the dual cost over its rows with the others
// held, so the cost falls monotonically and the iteration converges to the QP's optimum; it ends when every row of the island satisfies
// KKT to 1e-9 (1 + max |b|): y >= -tol on rows without a force, |y| <= tol on rows with one.  (Rounds measured on Delassus matrices from
// CPU roll-outs of the checker, tools/proto_block_cd.py: 10 in the median and 30 at most while up to ~90 rows carry a force; an island of 110 rows that
// ALL carry one -- seven boxes pushed into each other -- needs 30-120: two windows that couple through every box-box row.  Past
// UHC_WS_BLOCK_MAXIT rounds the sweeps take over, as before.)
// Returns the number of factorisations (bit 16 set: some island went through windows), or a negative reason (friction-loss rows, -2: `lost`
// and an island needs windows,
// no convergence in UHC_WS_MAXIT / UHC_WS_BLOCK_MAXIT rounds, a working set the pivoting cannot solve: -1 .. -4): the caller then runs the sweeps.
#define UHC_WS_MAXIT 16
#define UHC_WS_BLOCK_MAXIT 80
#define UHC_WS_LOST_MAXIT 6   // working-set rounds of a pass that has dropped rows (k_forward)
#define UHC_WS_WINDOWED 0x10000
#define UHC_WS_FILL 64  // lanes of an island's first working set that are filled by rank (k_as_general); 48 / 56 / 64 measured: 95.2 / 96.1 / 98.3 k env-steps/s on the headline
#define UHC_LOST_SWEEPS 32  // sweeps of an env-step that lost constraint rows beyond the last tier's capacity (k_forward)
// NRL = rows per lane: 2 in the general tier (<= 128 rows), 4 in the large tier (<= 256 rows); row r lives in lane r % 64, slot r / 64.
template <int TIER, bool DENSE>
__device__ __forceinline__ int k_as_general(const KernelArgs& A, const double* mb, double* S, int nefc, const LaneConst& LC, bool lost PROF_ARGS) {
    constexpr int NRL = TIER == 3 ? 4 : 2;
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    const RowMisc* RM = (const RowMisc*)(S + L.rowMisc);
    const int* RY = (const int*)(S + L.rowY);
    const int YS = T.maxdepth + 1;
    int* NI = (int*)(S + L.ncon_nefc);
    const int nslot = (DENSE && cap_of<TIER>(A).ndense > 0) ? NI[2] : 0;
    int* SLg = NI + 4 + UHC_MAXTWO;           // slot -> lane of the current working set
    int* list = (int*)(S + L.rowAref);        // working-set row ids (rowAref only holds friction-loss bounds, unused here)
    double* z = S + L.z;
    const unsigned long long below = (1ull << LANE) - 1ull;
    int rr[NRL];
    bool vv[NRL];
    bool fric = false;
#pragma unroll
    for (int h = 0; h < NRL; h++) {
        rr[h] = LANE + h * UHC_WAVE;
        vv[h] = rr[h] < nefc;
        fric = fric || (vv[h] && RTYPE(RM[rr[h]].type) == ROW_FRICTION);
    }
    if (wave_or(fric)) return -1;
    // (D jar of every row, k_rows: how close the warm-start acceleration leaves it to carrying a force -- read before `list` / `label` reuse the storage)
    double pr[NRL];
#pragma unroll
    for (int h = 0; h < NRL; h++) pr[h] = vv[h] ? S[L.rowAref + rr[h]] : 1e300;
    wsync();
    // the warm-start forces survive for the sweeps fallback (which must start where the reference's PGS starts)
#pragma unroll
    for (int h = 0; h < NRL; h++) if (vv[h]) S[L.rowW + rr[h]] = S[L.rowF + rr[h]];
    for (int i = LANE; i < T.nv; i += UHC_WAVE) z[i] = 0.0;
    wsync();
    // ---- islands: kinematic trees that share no contact row have independent QPs (A is block diagonal), so each island gets its own
    //      working set of <= 64 rows: a humanoid and four boxes resting beside it are five small solves, not one of 130 rows.
    //      Tree label = root body id; a dense row (two bodies) merges the labels of its two trees (lane 0, a handful of rows).
    int* label = list + UHC_WAVE;  // [nbody] island label of every tree root
    if (LANE < T.nbody) label[LANE] = LANE;
    wsync();
    if (LANE == 0)
        for (int k = 0; k < nslot; k++) {
            const RowMisc rm = RM[NI[4 + k]];
            const double* C = S + L.con + rm.aux * UHC_CON_STRIDE;
            int a = T.body_rootid[(int)C[19]], b = T.body_rootid[(int)C[20]];
            while (label[a] != a) a = label[a];
            while (label[b] != b) b = label[b];
            if (a != b) label[max(a, b)] = min(a, b);
        }
    wsync();
    int isl[NRL];
    bool fpos[NRL];
    // ---- first candidates: the rows whose warm-start force (k_rows: the force the previous substep's acceleration implies) is positive.
    //      Contacts persist from substep to substep, so this is nearly the final active set, and it is free.  (A cold start -- the rows
    //      that 8 matrix-free Gauss-Seidel sweeps from f = 0 leave with a force -- costs as much as three register-resident solves and
    //      was slower on every workload measured, also when the warm start marks more rows than lanes: those are capped below.)
#pragma unroll
    for (int h = 0; h < NRL; h++) {
        isl[h] = -1;
        fpos[h] = vv[h] && S[L.rowF + rr[h]] > 0.0;
        if (!vv[h]) continue;
        const RowMisc rm = RM[rr[h]];
        int a;
        if (rm.type & ROW_TWO) a = T.body_rootid[(int)S[L.con + rm.aux * UHC_CON_STRIDE + 20]];
        else a = T.dof_rootid[rm.last];
        while (label[a] != a) a = label[a];
        isl[h] = a;
    }
    // ---- rank of every row inside its island by D jar.  The first working set of an island is filled up to UHC_WS_FILL lanes with the rows that
    //      are closest to carrying a force instead of taking only those the warm start marks: on roll-outs of the self-colliding humanoid the
    //      working sets then end after 1.4 rounds instead of 2.5 (all rows at once: 1.0; a round = row load, Delassus build, pre-sweeps,
    //      factorisations, y on the other rows), and an island of <= UHC_WS_FILL rows -- a box on the floor -- is solved in one go.
    int rk[NRL];
#pragma unroll
    for (int h = 0; h < NRL; h++) rk[h] = 0;
    if (!UHC_EXP(256)) {
#pragma unroll
        for (int h2 = 0; h2 < NRL; h2++) {
            const int n2 = min(UHC_WAVE, nefc - h2 * UHC_WAVE);
            for (int l = 0; l < n2; l++) {
                const double ps = bcast(pr[h2], l);
                const int is = __builtin_amdgcn_readlane(isl[h2], l);
                const int s = h2 * UHC_WAVE + l;
#pragma unroll
                for (int h = 0; h < NRL; h++) rk[h] += (is == isl[h] && (ps < pr[h] || (ps == pr[h] && s < rr[h]))) ? 1 : 0;
            }
        }
    } else {
#pragma unroll
        for (int h = 0; h < NRL; h++) rk[h] = 1 << 20;
    }
    const int nfill = UHC_EXP(512) ? 48 : UHC_EXP(1024) ? 56 : UHC_WS_FILL;  // (UHC_DEBUG bits 9 / 10: experiments)
#pragma unroll
    for (int h = 0; h < NRL; h++) fpos[h] = fpos[h] || (vv[h] && rk[h] < nfill);  // from here on: the first candidates
    wsync();  // (the dense rows' Delassus columns may live where the contacts were: nothing reads the contacts from here on)
    double* ztot = S + L.vec;
    double* zfix = S + L.qacc;  // (the warm-start acceleration was last read when the rows were built; k_forward writes qacc after the solve)
    for (int i = LANE; i < T.nv; i += UHC_WAVE) ztot[i] = 0.0;
    int iters = 0;
    bool windowed = false;
    unsigned long long todo = 0ull;  // island labels present (nbody <= 64)
#pragma unroll
    for (int h = 0; h < NRL; h++) {
#pragma unroll 1
        for (int b = 1; b < T.nbody; b++) todo |= __builtin_amdgcn_ballot_w64(isl[h] == b) ? (1ull << b) : 0ull;
    }
    bool nomerge = (A.dbg & 1) != 0;
    while (todo) {
        // ---- next group: islands are independent, but every solve pays the fixed cost of a register-resident build, so small islands
        //      share one (A is block diagonal across them by itself: rows of different trees have no common dofs).  Islands are packed
        //      while their candidates fit 48 of the 64 lanes (room for violated rows to join); a group that overflows is split again.
        unsigned long long G = 0ull, rest = todo;
        int gcount = 0;
        while (rest) {
            const int I = __ffsll((long long)rest) - 1;
            rest &= rest - 1;
            int c = 0;
#pragma unroll
            for (int h = 0; h < NRL; h++) c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(isl[h] == I && fpos[h]));
            if (G == 0ull || (!nomerge && gcount + c <= 48)) { G |= 1ull << I; gcount += c; }
        }
        todo &= ~G;
        bool in[NRL], c[NRL], p[NRL];  // in the group; candidate of the working set; carried a force after the last solve (must stay)
#pragma unroll
        for (int h = 0; h < NRL; h++) { in[h] = isl[h] >= 0 && ((G >> isl[h]) & 1ull); c[h] = in[h] && fpos[h]; p[h] = false; }
        bool done = false, split = false;
        bool block = false;   // this island is taken in windows of 64 rows
        int cursor = 0;       // first row of the next window
        double tol = 0.0;     // KKT tolerance (0 while every force-carrying row is inside the working set: the solve is direct)
        for (int outer = 0; !done; outer++) {
            if (outer >= (block ? UHC_WS_BLOCK_MAXIT : lost ? UHC_WS_LOST_MAXIT : UHC_WS_MAXIT)) break;
            // ---- compact the group's working set into the lanes (row order kept)
            unsigned long long m[NRL];
            int nC = 0;
#pragma unroll
            for (int h = 0; h < NRL; h++) { m[h] = __builtin_amdgcn_ballot_w64(c[h]); nC += __builtin_popcountll(m[h]); }
            if (nC > UHC_WAVE) {
                if (__builtin_popcountll(G) > 1) {
                    split = true;  // too many candidates for one solve: take the group's islands one at a time
                    break;
                }
                // one island, more candidates than lanes: the rows with a force stay, the others join in row order while there is room
                // (the rest wait for a later round: any violated row that joins lowers the dual cost, so this still terminates)
                unsigned long long q[NRL];
                int nq = 0;
#pragma unroll
                for (int h = 0; h < NRL; h++) { q[h] = __builtin_amdgcn_ballot_w64(c[h] && !p[h]); nq += __builtin_popcountll(q[h]); }
                const int room = UHC_WAVE - (nC - nq);
                if (room <= 0 && lost) return -2;  // (a pass that dropped rows is not worth the windows: k_forward)
                if (room <= 0 && A.last_tier == 4) return -5;  // (an island with more force-carrying rows than lanes: Newton on the primal takes it, tier 4)
                if (room <= 0 || block) {
                    // 64 rows carry a force and more want in: the next window of 64 candidates, cyclically from the cursor
                    if (!block) {
                        block = true; windowed = true;
                        double bmax = 0.0;
#pragma unroll
                        for (int h = 0; h < NRL; h++) if (in[h]) bmax = fmax(bmax, fabs(S[L.rowB + rr[h]]));
                        bmax = -wave_min(-bmax);
                        tol = 1e-9 * (1.0 + bmax);
                    }
                    const int hc = cursor >> 6;
                    const unsigned long long lowc = (1ull << (cursor & 63)) - 1ull;
                    int K = 0;  // candidates ahead of the cursor
#pragma unroll
                    for (int h = 0; h < NRL; h++) K += h < hc ? __builtin_popcountll(m[h]) : (h == hc ? __builtin_popcountll(m[h] & lowc) : 0);
                    int before = 0, last = -1;
                    const int nAll = nC;
#pragma unroll
                    for (int h = 0; h < NRL; h++) {
                        int rank = before + __builtin_popcountll(m[h] & below) - K;
                        if (rank < 0) rank += nAll;
                        before += __builtin_popcountll(m[h]);
                        const unsigned long long e = __builtin_amdgcn_ballot_w64(c[h] && rank == UHC_WAVE - 1);
                        if (e) last = h * UHC_WAVE + __ffsll((long long)e) - 1;
                        c[h] = c[h] && rank < UHC_WAVE;
                    }
                    cursor = (last + 1) % (NRL * UHC_WAVE);
                    nC = 0;
#pragma unroll
                    for (int h = 0; h < NRL; h++) { m[h] = __builtin_amdgcn_ballot_w64(c[h]); nC += __builtin_popcountll(m[h]); }
                } else {
                int before = 0;
                nC = 0;
#pragma unroll
                for (int h = 0; h < NRL; h++) {
                    c[h] = c[h] && (p[h] || before + __builtin_popcountll(q[h] & below) < room);
                    before += __builtin_popcountll(q[h]);
                    m[h] = __builtin_amdgcn_ballot_w64(c[h]);
                    nC += __builtin_popcountll(m[h]);
                }
                }
            }
            if (nC == 0) {  // no candidate: f = 0 is optimal on this group iff b >= 0 on its rows
                bool any = false;
#pragma unroll
                for (int h = 0; h < NRL; h++) { c[h] = in[h] && S[L.rowB + rr[h]] < 0.0; any = any || c[h]; }
                if (!wave_or(any)) {
#pragma unroll
                    for (int h = 0; h < NRL; h++) if (in[h]) S[L.rowF + rr[h]] = 0.0;
                    wsync();
                    done = true;
                }
                continue;
            }
            {
                int off = 0;
#pragma unroll
                for (int h = 0; h < NRL; h++) {
                    if (c[h]) list[off + __builtin_popcountll(m[h] & below)] = rr[h];
                    off += __builtin_popcountll(m[h]);
                }
            }
            if (LANE < nslot) SLg[LANE] = -1;
            wsync();
            const bool valid = LANE < nC;
            const int r = valid ? list[LANE] : 0;
            FastRow row;
            RowMisc rm = {0, 0, 0, 0};
            if (valid) rm = RM[r];
            const bool two = valid && (rm.type & ROW_TWO) != 0;
            row.two = two ? (rm.type >> 8) : -1;
            row.type = valid ? RTYPE(rm.type) : 0;
            row.last = (valid && !two) ? rm.last : (valid ? -1 : 0);
            row.len = (valid && !two) ? T.dof_depth[rm.last] + 1 : 0;
            row.yoff = valid ? RY[r] : 0;
            row.R = valid ? S[L.rowR + r] : 1.0; row.b = valid ? S[L.rowB + r] : 0.0; row.f = 0.0; row.floss = 0.0; row.diag = 1.0;
            if (two) SLg[row.two] = LANE;
            double Y[UHC_YM];
            static_for<0, UHC_YM>([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value;
                Y[q] = q < row.len ? S[L.Y + row.yoff + q] : 0.0;
            });
            wsync();
            // ---- windows: the island's force-carrying rows outside C keep their force; zfix = sum of f Yhat over them, b_C += Yhat_C . zfix
            bool fx[NRL];
#pragma unroll
            for (int h = 0; h < NRL; h++) fx[h] = false;
            if (block) {
                unsigned long long fxm[NRL];
                for (int i = LANE; i < T.nv; i += UHC_WAVE) zfix[i] = 0.0;
                wsync();
#pragma unroll
                for (int h = 0; h < NRL; h++) {
                    const double fr = (in[h] && !c[h]) ? S[L.rowF + rr[h]] : 0.0;
                    fx[h] = fr > 0.0;
                    fxm[h] = __builtin_amdgcn_ballot_w64(fx[h]);
                    if (fx[h]) {
                        const RowMisc q = RM[rr[h]];
                        if (!(q.type & ROW_TWO)) {
                            const int len = T.dof_depth[q.last] + 1;
                            const short* anc = T.dof_anc + q.last * YS;
                            const double* Yr = S + L.Y + RY[rr[h]];
                            for (int k = 0; k < len; k++) __hip_atomic_fetch_add(zfix + anc[k], fr * Yr[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
                wsync();
                if constexpr (DENSE) {
                    for (int k = 0; k < nslot; k++) {  // dense rows: lane = dof
                        const int rid = __builtin_amdgcn_readfirstlane(NI[4 + k]);
                        unsigned long long bits = 0ull;
#pragma unroll
                        for (int h = 0; h < NRL; h++) if ((rid >> 6) == h) bits = fxm[h];
                        if (!((bits >> (rid & 63)) & 1ull)) continue;
                        const double fk = S[L.rowF + rid];
                        const double* Dk = S + L.dense + k * A.nvp;
                        if (LC.v0) zfix[LANE] += fk * Dk[LANE];
                        if (LC.v1) zfix[LANE + UHC_WAVE] += fk * Dk[LANE + UHC_WAVE];
                    }
                    wsync();
                }
                if (valid) {
                    double yb = 0.0;
                    if (two) {
                        const double* D = S + L.dense + row.two * A.nvp;
                        for (int i = 0; i < T.nv; i++) yb = fma(D[i], zfix[i], yb);
                    } else {
                        const short* anc = T.dof_anc + rm.last * YS;
                        const double* Yr = S + L.Y + row.yoff;
                        for (int k = 0; k < row.len; k++) yb = fma(Yr[k], zfix[anc[k]], yb);
                    }
                    row.b += yb;
                }
            }
            PROF(30)
            const int it = k_pgs_fast<TIER, DENSE>(A, mb, S, nC, row, Y, LC, SLg, nslot PROF_PASS);  // exact on C (or its sweeps to tolerance); z = sum_C f Yhat
            if (it < 0) return -4;  // pivot breakdown / no convergence of the pivoting on this working set
            iters += it > 0 ? it : 1;
            if (block) {  // z of the whole island
                for (int i = LANE; i < T.nv; i += UHC_WAVE) z[i] += zfix[i];
                wsync();
            }
            // ---- forces back to the island's rows, y on its rows outside C
#pragma unroll
            for (int h = 0; h < NRL; h++) if (in[h] && !fx[h]) S[L.rowF + rr[h]] = 0.0;
            wsync();
            if (valid) S[L.rowF + r] = row.f;
            wsync();
            bool viol[NRL], keep[NRL], anyv = false;
#pragma unroll
            for (int h = 0; h < NRL; h++) {
                viol[h] = keep[h] = false;
                if (!in[h]) continue;
                if (c[h]) { keep[h] = S[L.rowF + rr[h]] > 0.0; continue; }
                const RowMisc q = RM[rr[h]];
                double y = S[L.rowB + rr[h]];
                if (q.type & ROW_TWO) {
                    const double* D = S + L.dense + (q.type >> 8) * A.nvp;
                    for (int i = 0; i < T.nv; i++) y = fma(D[i], z[i], y);
                } else {
                    const int len = T.dof_depth[q.last] + 1;
                    const short* anc = T.dof_anc + q.last * YS;
                    const double* Yr = S + L.Y + RY[rr[h]];
                    for (int k = 0; k < len; k++) y = fma(Yr[k], z[anc[k]], y);
                }
                if (fx[h]) y = fma(S[L.rowR + rr[h]], S[L.rowF + rr[h]], y);  // (A_rr = |Yhat_r|^2 + R_r: a held row's own force)
                viol[h] = fx[h] ? fabs(y) > tol : y < -tol;  // (a held row must end with y = 0: it stays a candidate until it does)
                anyv = anyv || viol[h];
            }
            PROF(31)
            if (!wave_or(anyv)) {  // KKT holds on every row of the island: its optimum
                for (int i = LANE; i < T.nv; i += UHC_WAVE) ztot[i] += z[i];
                wsync();
                done = true;
            } else {
#pragma unroll
                for (int h = 0; h < NRL; h++) { c[h] = keep[h] || viol[h] || fx[h]; p[h] = keep[h] || fx[h]; }
                wsync();
            }
        }
        if (split) { todo |= G; nomerge = true; continue; }
        if (!done) return -3;
    }
    for (int i = LANE; i < T.nv; i += UHC_WAVE) z[i] = ztot[i];
    wsync();
    return iters | (windowed ? UHC_WS_WINDOWED : 0);
}

// ------------------------------------------------------------------ tier 4's solver: Newton on the primal problem (uhc_primal.h; the translation units
// of the large tier, whose workgroups go on as tier 4, define UHC_WITH_TIER4 -- the others never instantiate it)
template <int TIER>
__device__ __forceinline__ int k_primal(const KernelArgs& A, const double* mb, double* S, int nefc, const LaneConst& LC, const double* Yb, const double* Db, int* nact PROF_ARGS);
#ifdef UHC_WITH_TIER4
#include "uhc_primal.h"
#endif

// ------------------------------------------------------------------ mj_forward
struct FwdOut { int ncon, nefc, iters, overflow, nact, ytot; };  // ytot (fast tier): packed Yhat entries the pass's rows take  // nact (tier 4): rows that carry a force at the optimum  // overflow bit 0: the env does not fit this tier => redone by the next one
// LDS guard words (debug builds, -DUHC_GUARD_LDS on top of -DUHC_POISON_LDS; the host lays them out with UHC_GUARD_LDS=1 -- uhc_capi.cpp): two doubles after
// every region, holding the poison pattern; kind 0 = after the persistent regions (poisoned once per env with the rest of the LDS), kind 1 = after the
// regions of the constraint phase (armed when the dynamics temporaries that overlay them are dead, checked when the forward pass returns)
#define UHC_POISON_BITS 0x7ff8dead7fffbeefll
template <int TIER>
__device__ __forceinline__ void guard_arm(const KernelArgs& A, double* S) {
#ifdef UHC_GUARD_LDS
    if (!A.guard_tab) return;
    const int* g = A.guard_tab + 64 * (TIER - 1);
    wsync();
    if (LANE < g[1]) { S[g[32 + LANE]] = __longlong_as_double(UHC_POISON_BITS); S[g[32 + LANE] + 1] = __longlong_as_double(UHC_POISON_BITS); }
    wsync();
#endif
}
template <int TIER>
__device__ __forceinline__ void guard_check(const KernelArgs& A, const double* S, int env, int kind) {
#ifdef UHC_GUARD_LDS
    if (!A.guard_tab) return;
    const int* g = A.guard_tab + 64 * (TIER - 1);
    const int* o = g + (kind ? 32 : 2);
    wsync();
    if (LANE < g[kind]) {
        if (__double_as_longlong(S[o[LANE]]) != UHC_POISON_BITS || __double_as_longlong(S[o[LANE] + 1]) != UHC_POISON_BITS) {
            if (atomicAdd(A.guard_hits, 1) == 0) { A.guard_hits[1] = (TIER << 16) | (kind << 8) | LANE; A.guard_hits[2] = env; A.guard_hits[3] = o[LANE]; }
        }
    }
#endif
}

template <int TIER, bool DENSE>
__device__ __forceinline__ FwdOut k_forward(const KernelArgs& A, const double* mb, double* S, const LaneConst& LC, const BodyConst& BC, const PairConst& PC, MPark& MP, const int env PROF_ARGS) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    FwdOut out = {0, 0, 0, 0, 0, 0};
    k_kinematics<TIER>(A, mb, S, BC PROF_PASS);
    PROF(1)
    k_com_pos<TIER>(A, mb, S, BC);
    PROF(2)
    k_crb<TIER, DENSE>(A, mb, S, MP, BC PROF_PASS);
    PROF(3)
    k_factor<TIER>(A, S, L.LD, LC);
    PROF(4)
    k_com_vel<TIER>(A, S, BC);
    PROF(5)
    k_rne<TIER>(A, S, BC PROF_PASS);
    PROF(6)
    k_smooth<TIER>(A, mb, S, LC);
    PROF(7)
    guard_arm<TIER>(A, S);
    out.ncon = k_collision<TIER, DENSE>(A, mb, S, &out.overflow, PC PROF_PASS);
    PROF(8)
    out.nefc = k_enumerate_rows<TIER, DENSE>(A, mb, S, out.ncon, &out.overflow);
    DofVec x = {0.0, 0.0};
    if (out.overflow & 1) return out;  // (bit 0 is only raised by a tier that hands the env on)
    if (out.nefc > 0) {
        if constexpr (TIER == 1) {
            FastRow row;
            double Yreg[UHC_YM];
            const int st = k_rows_fast<DENSE>(A, mb, S, out.nefc, row, Yreg, LC, &out.ytot);  // 1: needs the general kernel, 2: rows dropped (truncate mode)
            out.overflow |= st | (st == 1 ? UHC_WHY_ROW_STORAGE : 0);
            if (st == 1) return out;
            PROF(9)
            const int* NI = (const int*)(S + L.ncon_nefc);
            out.iters = k_pgs_fast<1, DENSE>(A, mb, S, out.nefc, row, Yreg, LC, NI + 4, DENSE ? NI[2] : 0 PROF_PASS);
            PROF(12)
        } else {
            double* Yb = y_store<TIER>(A, S, env);
            double* Db = d_store<TIER>(A, S, env);
            if (k_rows<TIER>(A, mb, S, out.nefc, LC, Yb, Db)) { out.overflow |= 1 | UHC_WHY_ROW_STORAGE; return out; }  // the packed rows need the next tier's storage
            PROF(9)
            int it = -1;
            if constexpr (TIER == 4) {
                // the last tier: Newton on the primal (exact, whatever the number of rows or of force-carrying rows: uhc_primal.h); friction-loss
                // rows (box constraints) and solver 0 go to the sweeps as in the other tiers
                bool fric = false;
                for (int r = LANE; r < out.nefc; r += UHC_WAVE) fric = fric || RTYPE(((const RowMisc*)(S + L.rowMisc))[r].type) == ROW_FRICTION;
                fric = wave_or(fric);
                if (T.solver == 1 && !fric) {
                    it = k_primal<TIER>(A, mb, S, out.nefc, LC, Yb, Db, &out.nact PROF_PASS);
                    out.overflow |= 128;                            // UHC_F_REDO bit 30: solved by Newton on the primal
                    if (it < 0) { out.overflow |= 256; it = -it; }  // bit 29: it stopped at its iteration cap (the best iterate is used)
                } else {
                    if (T.solver == 1) out.overflow |= 4 | 8;
                    it = k_pgs<TIER>(A, mb, S, out.nefc, T.iterations, Yb, Db);
                }
                out.iters = it;
                PROF(11)
            } else {
            // An env that has just lost rows or contacts beyond the last tier's capacity (overflow bit 1; UHC_F_REDO bit 7) is no longer
            // solving the reference's QP.  Most such passes are ordinary pile-ups and their truncated QP is solved exactly like any other.
            // One in twelve is a simulation on its way to the bad-value flag (tools/diag_redo.py on the configs[4] probe, replayed on
            // the checker: joint speeds of 160 - 20 000 rad/s at the head of the step, |b| of 1e7 - 5e16 in the substep, one island of 250
            // rows that the working sets cannot finish), and used to end in `iterations` sweeps: 34 ms per substep in the large tier, a
            // third of that probe's time.  A pass that has dropped rows therefore gets a bounded attempt -- UHC_WS_LOST_MAXIT working-set
            // rounds, no windows -- and, when that gives up, UHC_LOST_SWEEPS sweeps from the warm start.  (Batches whose last tier is the
            // large one, UHC_TIERS=3: with tier 4 behind it the large tier hands such an env on instead.)
            const bool lost = T.solver == 1 && (out.overflow & 2) != 0;
            if (T.solver == 1) it = k_as_general<TIER, DENSE>(A, mb, S, out.nefc, LC, lost PROF_PASS);
            if (it >= 0 && (it & UHC_WS_WINDOWED)) { it &= UHC_WS_WINDOWED - 1; out.overflow |= 16; }  // (UHC_F_REDO bit 3: an island was solved in windows)
            if (it < -1 && T.solver == 1 && A.last_tier == 4) {
                // the working sets did not finish (an island with more force-carrying rows than lanes, no convergence, a pivot breakdown): the env
                // goes on to tier 4, whose Newton iteration on the primal has no such limit -- instead of the sweeps of rounds 2-4
                out.overflow |= 1 | UHC_WHY_SOLVER;
                return out;
            }
            if (it < 0) {
                if (T.solver == 1) {  // the working sets gave up: sweep from the warm start, as the reference's PGS does
                    for (int r = LANE; r < out.nefc; r += UHC_WAVE) S[L.rowF + r] = S[L.rowW + r];
                    wsync();
                }
                out.overflow |= 4 | ((T.solver == 1 && it != -2) ? (4 << (-it)) : 0);  // (-2: a pass that dropped rows met an island that needs windows)
                it = k_pgs<TIER>(A, mb, S, out.nefc, lost ? min(T.iterations, UHC_LOST_SWEEPS) : T.iterations, Yb, Db);
            }  // 4: solved by sweeps (to tolerance), reported in UHC_F_REDO bit 1; 8 / 32 / 64: why the working sets gave up (bits 2, 4, 5); 16: not a fallback (above)
            out.iters = it;
            PROF(11)
            }
        }
        // qacc = qacc_smooth + L^-1 D^-1/2 z
        if (LANE < T.nv) x.a = S[L.z + LANE] * S[L.sdinv + LANE];
        if (LANE + UHC_WAVE < T.nv) x.b = S[L.z + LANE + UHC_WAVE] * S[L.sdinv + LANE + UHC_WAVE];
        k_solve<TIER>(A, S, L.LD, x, 1, LC);
    }
    if (LANE < T.nv) S[L.qacc + LANE] = S[L.smooth + LANE] + x.a;
    if (LANE + UHC_WAVE < T.nv) S[L.qacc + LANE + UHC_WAVE] = S[L.smooth + LANE + UHC_WAVE] + x.b;
    wsync();
    return out;
}

// ------------------------------------------------------------------ P10 semi-implicit Euler
// acc: LDS offset of the acceleration the velocity update uses (qacc, or the implicitly damped one left by k_damped_accel)
template <int TIER>
__device__ __forceinline__ void k_euler(const KernelArgs& A, double* S, int acc) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    const double h = T.timestep;
    for (int i = LANE; i < T.nv; i += UHC_WAVE) S[L.qvel + i] += h * S[acc + i];
    wsync();
    for (int j = LANE; j < T.njnt; j += UHC_WAVE) {
        int qa = T.jnt_qposadr[j], da = T.jnt_dofadr[j];
        const int jt = T.jnt_type[j];
        if (jt == UHC_JNT_FREE) {
            for (int k = 0; k < 3; k++) S[L.qpos + qa + k] += h * S[L.qvel + da + k];
            qa += 3; da += 3;
        }
        if (jt == UHC_JNT_FREE || jt == UHC_JNT_BALL) {
            double w[3] = {S[L.qvel + da], S[L.qvel + da + 1], S[L.qvel + da + 2]}, qr[4], q[4];
            const double n = sqrt(dot3(w, w));
            if (n < UHC_MINVAL) { w[0] = 1; w[1] = w[2] = 0; } else { w[0] /= n; w[1] /= n; w[2] /= n; }
            axis_angle_quat(qr, w, h * n);
            for (int k = 0; k < 4; k++) q[k] = S[L.qpos + qa + k];
            quat_mul(q, q, qr);
            quat_normalize(q);
            for (int k = 0; k < 4; k++) S[L.qpos + qa + k] = q[k];
        } else {
            S[L.qpos + qa] += h * S[L.qvel + da];
        }
    }
    wsync();
}
__device__ __forceinline__ bool bad(double x) { return isnan(x) || x > UHC_MAXVAL || x < -UHC_MAXVAL; }

// [MJ-ext] mj_Euler with joint damping: (M + h diag(B)) a = qfrc_smooth + qfrc_constraint = M qacc, i.e.
// a = qacc - (M + h B)^-1 (h B qacc) -- the right-hand side is diagonal, so neither force vector has to be kept.  M comes from
// the parked copy of this forward pass (FAST) or S.M; the factor overwrites LD (rebuilt by the next forward pass / PD solve).
// Result in S[L.smooth] (qacc_smooth is dead after k_forward); S[L.qacc] keeps the explicit acceleration (warm start, checks).
template <int TIER>
__device__ __forceinline__ void k_damped_accel(const KernelArgs& A, const double* mb, double* S, const MPark& MP, const LaneConst& LC) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    const double h = T.timestep;
#pragma unroll
    for (int m = 0; m < UHC_MREG; m++) {
        const int e = LANE + UHC_WAVE * m;
        if (e < T.nM) S[L.LD + e] = __hiloint2double(agpr_get(MP.hi[m]), agpr_get(MP.lo[m]));
    }
    wsync();
    DofVec x = {0.0, 0.0};
    if (LC.v0) { const double hb = h * mb[A.o.dof_damping + LANE]; S[L.LD + LC.m0] += hb; x.a = hb * S[L.qacc + LANE]; }
    if (LC.v1) { const double hb = h * mb[A.o.dof_damping + LANE + UHC_WAVE]; S[L.LD + LC.m1] += hb; x.b = hb * S[L.qacc + LANE + UHC_WAVE]; }
    wsync();
    k_factor<TIER>(A, S, L.LD, LC);
    k_solve<TIER>(A, S, L.LD, x, 0, LC);
    if (LC.v0) S[L.smooth + LANE] = S[L.qacc + LANE] - x.a;
    if (LC.v1) S[L.smooth + LANE + UHC_WAVE] = S[L.qacc + LANE + UHC_WAVE] - x.b;
    wsync();
}

// ------------------------------------------------------------------ E3/E4 stable PD, E5 implicit residual force
// compute_torque + compute_desired_accel (humanoid_im.py:1014-1076): uses the M and bias left by the
// previous forward pass (S.M, S.bias); factorises M + diag(kd) dt into S.LD (overwritten later by P3).
template <int TIER>
__device__ __forceinline__ void k_pd_torque(const KernelArgs& A, double* S, const double* action, const double* tbase, int it, const MPark& MP, const LaneConst& LC PROF_ARGS) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    const DevCtrl& C = A.c;
    const double dt = T.timestep;
    const int nu = T.nu, vf = C.rfc_mode == 1 ? 6 : C.rfc_mode == 2 ? C.n_vf_body * C.body_vf_dim : 0;
    double skp = 1, skd = 1;
    if (C.meta_pd == 1) {
        skp = clampd(action[nu + vf + it] + 1, 0, 10);
        skd = clampd(action[nu + vf + it + C.n_substeps] + 1, 0, 10);
    }
#pragma unroll
    for (int m = 0; m < UHC_MREG; m++) {
        const int e = LANE + UHC_WAVE * m;
        if (e < T.nM) S[L.LD + e] = __hiloint2double(agpr_get(MP.hi[m]), agpr_get(MP.lo[m]));
    }
    wsync();
    PROF(24)
    // lane owns dofs LANE and LANE+64; actuator a drives dof 6+a (free root first)
    double kp[2] = {0, 0}, kd[2] = {0, 0}, qe[2] = {0, 0}, qv[2] = {0, 0};
    for (int h = 0; h < 2; h++) {
        const int i = LANE + h * UHC_WAVE, a = i - 6;
        if (i < T.nv) qv[h] = S[L.qvel + i];
        if (a >= 0 && a < nu) {
            const double cur = S[L.qpos + 7 + a];
            double base = tbase[a];
            // (the reference's unwrapping loops; a joint angle that has run away -- a physics blow-up on its way to the failure check --
            //  would keep them turning for seconds, so anything beyond 32 turns is taken off in one go first)
            if (!(fabs(base - cur) <= 64 * M_PI)) { const double turns = rint((base - cur) / (2 * M_PI)); base = fabs(turns) < 1e300 ? base - 2 * M_PI * turns : cur; }
            while (base - cur > M_PI) base -= 2 * M_PI;
            while (base - cur < -M_PI) base += 2 * M_PI;
            double gkp = C.jkp[a], gkd = C.jkd[a];
            if (C.meta_pd == 1) { gkp *= skp; gkd *= skd; }
            else if (C.meta_pd == 2) {
                gkp *= clampd(action[nu + vf + a] + 1, 0, 10);
                gkd *= clampd(action[nu + vf + nu + a] + 1, 0, 10);
            }
            kp[h] = gkp; kd[h] = gkd;
            qe[h] = cur + qv[h] * dt - (base + action[a]);
            S[L.LD + T.dof_madr[i]] += gkd * dt;
        }
    }
    wsync();
    PROF(16)
    k_factor<TIER>(A, S, L.LD, LC);
    PROF(17)
    DofVec x;
    x.a = LANE < T.nv ? -S[L.bias + LANE] - kp[0] * qe[0] - kd[0] * qv[0] : 0.0;
    x.b = LANE + UHC_WAVE < T.nv ? -S[L.bias + LANE + UHC_WAVE] - kp[1] * qe[1] - kd[1] * qv[1] : 0.0;
    k_solve<TIER>(A, S, L.LD, x, 0, LC);
    PROF(18)
    for (int h = 0; h < 2; h++) {
        const int a = LANE + h * UHC_WAVE - 6;
        if (a >= 0 && a < nu) {
            const double qdd = h ? x.b : x.a;
            const double tq = -kp[h] * qe[h] - kd[h] * (qv[h] + qdd * dt);
            S[L.ctrl + a] = clampd(tq, -C.torque_lim[a], C.torque_lim[a]);
        }
    }
    wsync();
}
template <int TIER>
__device__ __forceinline__ void k_rfc_implicit(const KernelArgs& A, double* S, const double* action) {
    const DevLds& L = lds_of<TIER>(A);
    const DevCtrl& C = A.c;
    if (LANE == 0) {
        double vf[6], q[4], rq[4], hq[4], R[9], r[3];
        for (int k = 0; k < 6; k++) vf[k] = action[A.t.nu + k] * C.rfc_scale;
        for (int k = 0; k < 4; k++) rq[k] = S[L.qpos + 3 + k];
        quat_mul(q, rq, C.base_rot_inv);
        const double n = sqrt(q[0] * q[0] + q[3] * q[3]);  // get_heading_q: math_utils.py:134-139
        hq[0] = q[0] / n; hq[1] = 0; hq[2] = 0; hq[3] = q[3] / n;
        quat_to_mat(R, hq);
        mat_vec(r, R, vf);
        for (int k = 0; k < 3; k++) vf[k] = r[k];
        for (int k = 0; k < 6; k++) S[L.applied + k] = clampd(vf[k], -C.rfc_lim, C.rfc_lim);
    }
    wsync();
}

// rfc_explicit (humanoid_im.py:1080-1132) + mj_applyFT [MJ-ext]: one (contact point, force, torque) per listed body, given in
// the body frame; qfrc_applied = sum_b J_b(point)^T [f; tau].  With cdof about the root's subtree COM c0 this is
// qfrc_i = cdof_i . W_sub(body(i)), W_b = [tau + (p - c0) x f ; f] summed over the subtree of the dof's body.
// Uses the kinematics left by the previous forward pass (xpos, xmat, cdof, subtree COM), as the reference does.
template <int TIER>
__device__ __forceinline__ void k_rfc_explicit(const KernelArgs& A, double* S, const double* action) {
    const DevLds& L = lds_of<TIER>(A);
    const DevTopo& T = A.t;
    const DevCtrl& C = A.c;
    double* W = S + L.cfrc;  // free between substeps
    for (int i = LANE; i < 6 * T.nbody; i += UHC_WAVE) W[i] = 0;
    wsync();
    if (LANE < C.n_vf_body) {
        const int body = C.vf_body[LANE];
        if (body > 0) {
            const double* vf = action + T.nu + LANE * C.body_vf_dim;
            const double* R = S + L.xmat + 9 * body;
            const double* c0 = S + L.rootcom + 3 * T.body_rootid[body];
            double cpl[3] = {vf[0], vf[1], vf[2]}, fl[3], tl[3] = {0, 0, 0}, off[3], f[3], tq[3], n[3];
            for (int k = 0; k < 3; k++) { fl[k] = vf[3 + k] * C.rfc_scale; if (C.body_vf_dim >= 9) tl[k] = vf[6 + k] * C.rfc_scale; }
            mat_vec(off, R, cpl);
            mat_vec(f, R, fl);
            mat_vec(tq, R, tl);
            for (int k = 0; k < 3; k++) off[k] += S[L.xpos + 3 * body + k] - c0[k];
            cross3(n, off, f);
            for (int k = 0; k < 3; k++) { W[6 * body + k] = n[k] + tq[k]; W[6 * body + 3 + k] = f[k]; }
        }
    }
    wsync();
    if (LANE < 6)  // bodies are numbered depth first (parent < child): one backward sweep sums every subtree
        for (int b = T.nbody - 1; b >= 1; b--) {
            const int p = T.body_parentid[b];
            if (p > 0) W[6 * p + LANE] += W[6 * b + LANE];
        }
    wsync();
    for (int i = LANE; i < T.nv; i += UHC_WAVE) {
        const double* cd = S + L.cdof + 6 * i;
        const double* w = W + 6 * T.dof_bodyid[i];
        S[L.applied + i] = cd[0] * w[0] + cd[1] * w[1] + cd[2] * w[2] + cd[3] * w[3] + cd[4] * w[4] + cd[5] * w[5];
    }
    wsync();
}

// ------------------------------------------------------------------ the kernels
// MODE 0: do_simulation (n_substeps of control + step);  MODE 1: forward only (after set_state).
// FAST: compact LDS (4 workgroups per CU), <= 64 constraint rows, A in registers.  An env that does not
// fit (rows, contacts or Yhat storage) leaves its state untouched and raises redo[env]; the host then
// launches the general variant on exactly those envs.
// DENSE: the model has contacts between two moving bodies (convex-convex pairs): MPR narrow phase + dense rows are compiled in.  The
// floor-only stock model runs the DENSE = false instantiation, whose code and register allocation are those of the kernel without them.
// returns 1 when the env was handed on to the next tier (nothing committed; the large tier's workgroup then goes on with it as tier 4), else 0
template <int MODE, int TIER, bool DENSE>
__device__ __forceinline__ int uhc_step_env(const KernelArgs& A, const double* __restrict__ d_action, const double* __restrict__ d_tbase, const int env) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    double* S = smem;
#ifdef UHC_POISON_LDS  // debug builds (tools/poison_build.py): any read of LDS the kernel did not write first meets a NaN / a huge int
    for (int i = LANE; i < L.total; i += UHC_WAVE) S[i] = __longlong_as_double(UHC_POISON_BITS);
    wsync();
#endif
    const double* mb = A.s.model_blob + (size_t)(A.s.env_model ? A.s.env_model[env] : 0) * A.o.stride;
    // tier trace (UHC_DEBUG bit 4, product builds; tools/tier_trace.py): when each tier took the env up and let it go (100 MHz wall clock)
    // and the substep of a hand-on, in the first words of the env's stage-profile record
#ifndef UHC_STAGE_PROF
#define TRACE(slot, v) if (MODE == 0 && (A.dbg & 16) && LANE == 0) A.s.prof[(size_t)env * UHC_NPROF + (TIER < 4 ? (slot) : 21 + ((slot) & 1))] = (long long)(v);  // (tier 4's stamps: words 21 / 22; the substep at which the large tier handed on: word 23)
#else
#define TRACE(slot, v)
#endif
    TRACE(2 * (TIER - 1), wall_clock64())
    int fail = MODE == 0 ? A.s.fail[env] : 0;
    // A tier that finds an env too big in substep k >= 1 hands it on WITH the substeps it has done: the state at the start of substep k
    // (qpos, qvel, warm start) and that substep's controls (ctrl, applied: PD torque and residual force are already computed) are in
    // the state arrays, and this tier goes on from substep k instead of repeating the step -- the work below is not lost, and the step
    // ends (15 - k) / 15 of an env-step after the hand-on instead of a whole one.
    // (read past the scalar cache: `env` is wave-uniform, and a consumer workgroup that lives through many envs of a launch must not meet a
    //  line another wave fetched before the tier below wrote its word)
    const int res = (MODE == 0 && TIER != 1) ? __hip_atomic_load(A.s.resume + env, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    const bool fresh = MODE == 0 && res == 0 && A.s.fresh[env] != 0;  // restarted by uhc_env_auto_reset: sim.forward() of the reset is still due
    // ---- load state (coalesced: consecutive lanes, consecutive doubles)
    for (int i = LANE; i < T.nq; i += UHC_WAVE) S[L.qpos + i] = A.s.qpos[(size_t)env * T.nq + i];
    for (int i = LANE; i < T.nv; i += UHC_WAVE) {
        S[L.qvel + i] = A.s.qvel[(size_t)env * T.nv + i];
        S[L.qacc + i] = A.s.qacc_ws[(size_t)env * T.nv + i];  // warm start (MuJoCo qacc_warmstart)
        S[L.applied + i] = A.s.applied[(size_t)env * T.nv + i];
        if (MODE == 0) S[L.bias + i] = A.s.bias[(size_t)env * T.nv + i];
    }
    for (int i = LANE; i < T.nu; i += UHC_WAVE) S[L.ctrl + i] = A.s.ctrl[(size_t)env * T.nu + i];
    if (TIER == 1 && !DENSE && L.mij > 0) {   // (row, col) of the sparse mass-matrix entries, two per 32-bit word (the kinematics-only launch
                                              // <2, 1, false> also serves layouts without the table: models with body-body contacts)
        unsigned int* dst = (unsigned int*)(S + L.mij);
        const unsigned int* src = (const unsigned int*)T.m_ij;
        for (int w = LANE; w < (T.nM + 1) / 2; w += UHC_WAVE) dst[w] = src[w];
    }
    if (LANE == 0) { S[L.zero] = 0.0; S[L.LD + T.nM] = 0.0; S[L.LD + T.nM + 1] = 0.0; }  // slots idle lanes read / write instead of branching
    if (MODE == 0 && A.c.rfc_mode == 2) {  // explicit RFC reads the kinematics of the previous forward pass
        for (int i = LANE; i < 6 * T.nv; i += UHC_WAVE) S[L.cdof + i] = A.s.cdof[(size_t)env * 6 * T.nv + i];
        for (int i = LANE; i < 3 * T.nbody; i += UHC_WAVE) {
            S[L.rootcom + i] = A.s.rootcom[(size_t)env * 3 * T.nbody + i];
            S[L.xpos + i] = A.s.xpos[(size_t)env * 3 * T.nbody + i];
        }
        for (int b = LANE; b < T.nbody; b += UHC_WAVE) {
            double q[4], R[9];
            for (int k = 0; k < 4; k++) q[k] = A.s.xquat[(size_t)env * 4 * T.nbody + 4 * b + k];
            quat_to_mat(R, q);
            for (int k = 0; k < 9; k++) S[L.xmat + 9 * b + k] = R[k];
        }
    }
    const LaneConst LC = lane_const<DENSE>(T);
    const BodyConst BC = body_const(T);
    const PairConst PC = pair_const(T);
    // joint-space inertia between substeps: the PD solve of substep t+1 uses M of substep t's forward pass
    MPark MP;
#pragma unroll
    for (int m = 0; m < UHC_MREG; m++) {
        const int e = LANE + UHC_WAVE * m;
        const double v = (MODE == 0 && e < T.nM) ? A.s.qM[(size_t)env * T.nM + e] : 0.0;
        agpr_put(MP.lo[m], __double2loint(v)); agpr_put(MP.hi[m], __double2hiint(v));
    }
    wsync();
    FwdOut fo = {0, 0, 0, 0, 0, 0};
    int it = 0;
    int overflow = 0, swept = 0;  // swept: bit 8 + k = substep k of this step was solved by the sweeps (general kernel, solver 1)
    bool ran = false, fits = true;  // fits (general / large tier): every substep of this step was within the fast kernel's capacity
    int pk_nefc = 0, pk_ncon = 0, pk_ntwo = 0, pk_y = 0, pk_act = 0;  // the step's peaks over its substeps: rows, contacts, body-body rows, packed Yhat entries (sticky tiers)
    PROF_DECL
    if (MODE == 2) {  // kinematics of a device-side restart: what the reset observation reads; the rest of sim.forward() is deferred
        k_kinematics<TIER>(A, mb, S, BC PROF_PASS);
        for (int i = LANE; i < 3 * T.nbody; i += UHC_WAVE) {
            A.s.xpos[(size_t)env * 3 * T.nbody + i] = S[L.xpos + i];
            A.s.xipos[(size_t)env * 3 * T.nbody + i] = S[L.xipos + i];
        }
        for (int i = LANE; i < 4 * T.nbody; i += UHC_WAVE) A.s.xquat[(size_t)env * 4 * T.nbody + i] = S[L.xquat + i];
        return 0;
    }
    if (MODE == 1) {
        // set_state + sim.forward(): a pose that is not a pose (NaN, beyond +-1e10: what mj_checkPos / mj_checkVel reject at the head of
        // mj_step [MJ-ext]) never enters the forward pass -- the env is flagged `fail` and left as it was handed over (the reference turns
        // MuJoCo's warning into `fail`, uhc/envs/humanoid_im.py:1207-1211); every stage behind this line sees finite inputs
        int b = 0;
        for (int i = LANE; i < T.nq; i += UHC_WAVE) b |= bad(S[L.qpos + i]);
        for (int i = LANE; i < T.nv; i += UHC_WAVE) b |= bad(S[L.qvel + i]);
        if (wave_or(b)) fail = 1;
        else {
            fo = k_forward<TIER, DENSE>(A, mb, S, LC, BC, PC, MP, env PROF_PASS);
            guard_check<TIER>(A, S, env, 1);
            overflow |= fo.overflow;
            ran = true;
        }
    } else if (!fail) {
        const double* action = d_action + (size_t)env * A.c.action_dim;
        const double* tbase = d_tbase + (size_t)env * T.nu;
        // a freshly restarted env first runs the forward pass of its reset (it = -1: no control, no integration), through the
        // same inlined k_forward as the substeps
        for (it = fresh ? -1 : res; it < A.c.n_substeps; it++) {
            int b = 0;
            if (it >= 0) {
            if (!(res > 0 && it == res)) {  // (the controls of the substep an env was handed on in came with it)
            if (A.c.action_type == 0) k_pd_torque<TIER>(A, S, action, tbase, it, MP, LC PROF_PASS);
            else {
                for (int a = LANE; a < T.nu; a += UHC_WAVE)
                    S[L.ctrl + a] = clampd(action[a] * A.c.a_scale[a] * 100, -A.c.torque_lim[a], A.c.torque_lim[a]);
                wsync();
            }
            if (A.c.rfc_mode == 1) k_rfc_implicit<TIER>(A, S, action);
            else if (A.c.rfc_mode == 2) k_rfc_explicit<TIER>(A, S, action);
            }
            }
            // mj_step: checkPos / checkVel -> forward -> checkAcc -> Euler  (also ahead of the forward pass a device-side restart still owes)
            for (int i = LANE; i < T.nq; i += UHC_WAVE) b |= bad(S[L.qpos + i]);
            for (int i = LANE; i < T.nv; i += UHC_WAVE) b |= bad(S[L.qvel + i]);
            if (wave_or(b)) { fail = 1; break; }
            PROF(0)
            fo = k_forward<TIER, DENSE>(A, mb, S, LC, BC, PC, MP, env PROF_PASS);
            guard_check<TIER>(A, S, env, 1);
            PROF(13)
            overflow |= fo.overflow;
            if (overflow & 1) break;
            if (TIER != 1 && (fo.overflow & 4) && it >= 0 && it < 21) swept |= 1 << (8 + it);  // (bits 8 .. 28; 29 and 30 report the primal solver)
            if (TIER != 1) fits = fits && fo.nefc <= UHC_WAVE && fo.ncon <= A.cf.maxcon &&
                              (!(DENSE && cap_of<TIER>(A).ndense > 0) || ((const int*)(S + L.ncon_nefc))[2] <= A.cf.ndense);
            {
                const int ntwo = (DENSE && cap_of<TIER>(A).ndense > 0) ? ((const int*)(S + L.ncon_nefc))[2] : 0;
                pk_nefc = max(pk_nefc, fo.nefc); pk_ncon = max(pk_ncon, fo.ncon); pk_ntwo = max(pk_ntwo, ntwo); pk_act = max(pk_act, fo.nact);
                if (TIER != 1 && fo.nefc > 0) pk_y = max(pk_y, ((const int*)(S + L.rowY))[fo.nefc]);
                if (TIER == 1) pk_y = max(pk_y, fo.ytot);
            }
            ran = true;
            if (it < 0) {  // mj_forward alone leaves qacc_warmstart (zero after the reset) for the first real substep
                wsync();
                for (int i = LANE; i < T.nv; i += UHC_WAVE) S[L.qacc + i] = A.s.qacc_ws[(size_t)env * T.nv + i];
                wsync();
                continue;
            }
            b = 0;
            for (int i = LANE; i < T.nv; i += UHC_WAVE) b |= bad(S[L.qacc + i]);
            if (wave_or(b)) { fail = 1; break; }
            if (T.has_damping) { k_damped_accel<TIER>(A, mb, S, MP, LC); k_euler<TIER>(A, S, L.smooth); }
            else k_euler<TIER>(A, S, L.qacc);
            PROF(14)
        }
    }
    if (overflow & 1) {  // nothing committed: the next tier takes the env over -- from the substep that did not fit, or from the same inputs
        if (MODE == 0 && it >= 1 && it > res && !(A.dbg & 4)) {  // (UHC_DEBUG bit 2: hand on from the start of the step, for A/B measurements)
            for (int i = LANE; i < T.nq; i += UHC_WAVE) A.s.qpos[(size_t)env * T.nq + i] = S[L.qpos + i];
            for (int i = LANE; i < T.nv; i += UHC_WAVE) {
                A.s.qvel[(size_t)env * T.nv + i] = S[L.qvel + i];
                A.s.qacc_ws[(size_t)env * T.nv + i] = S[L.qacc + i];
                A.s.applied[(size_t)env * T.nv + i] = S[L.applied + i];
            }
            for (int i = LANE; i < T.nu; i += UHC_WAVE) A.s.ctrl[(size_t)env * T.nu + i] = S[L.ctrl + i];
            if (LANE == 0) A.s.resume[env] = it;
            TRACE(TIER < 3 ? 5 + TIER : 23, it)
            __threadfence();  // (the queue append below publishes the env: its state must be visible first)
        }
        if (LANE == 0) {
            // flagged for the chained launch of the next tier, which takes whatever no consumer took (none running, or given up) -- BEFORE the
            // env is published: a consumer that claims and finishes it clears the flag, and that clear must be the last write
            // (the large tier hands on to tier 4: pend3 = 2 -- in a one-workgroup-per-env launch the same workgroup goes on with it at once; a queue
            //  consumer leaves the env flagged for the step's chained large-tier launch, which takes a `2` straight to tier 4)
            if (TIER == 1) A.s.pend2[env] = 1; else if (TIER == 2) { A.s.pend2[env] = 0; A.s.pend3[env] = 1; } else A.s.pend3[env] = 2;
            if (TIER <= 2) A.s.why[env] = (A.s.why[env] & (TIER == 1 ? 0xff00 : 0x00ff)) | (((overflow >> 16) & 0xff) << (TIER == 1 ? 0 : 8)) | ((it & 0xff) << 16);  // diagnostic: why, and at which substep
            else A.s.why[env] |= ((overflow >> 16) & 0xff) << 24;  // bits 24+: why the large tier handed the env on to tier 4
            __threadfence();
            if (TIER == 3 && A.cnt4) atomicAdd(A.cnt4, 1);
            if (A.q_next) {  // the next tier's consumers are running beside this launch: straight into their queue
                const int k = atomicAdd(A.q_next_count, 1);
                __hip_atomic_store(A.q_next + k, env, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (TIER == 1 && MODE == 0) atomicAdd(A.s.path_stats, 1ull);
        }
        TRACE(2 * (TIER - 1) + 1, wall_clock64())
        guard_check<TIER>(A, S, env, 0);
        return 1;
    }
    guard_check<TIER>(A, S, env, 0);
    // ---- store state
    for (int i = LANE; i < T.nq; i += UHC_WAVE) A.s.qpos[(size_t)env * T.nq + i] = S[L.qpos + i];
    for (int i = LANE; i < T.nv; i += UHC_WAVE) {
        A.s.qvel[(size_t)env * T.nv + i] = S[L.qvel + i];
        A.s.qacc[(size_t)env * T.nv + i] = S[L.qacc + i];
        if (MODE == 0 && ran) A.s.qacc_ws[(size_t)env * T.nv + i] = S[L.qacc + i];  // mj_forward alone leaves the warm start
        A.s.applied[(size_t)env * T.nv + i] = S[L.applied + i];
    }
    for (int i = LANE; i < T.nu; i += UHC_WAVE) A.s.ctrl[(size_t)env * T.nu + i] = S[L.ctrl + i];
    if (ran) {
        for (int i = LANE; i < T.nv; i += UHC_WAVE) A.s.bias[(size_t)env * T.nv + i] = S[L.bias + i];
#pragma unroll
        for (int m = 0; m < UHC_MREG; m++) {
            const int e = LANE + UHC_WAVE * m;
            if (e < T.nM) A.s.qM[(size_t)env * T.nM + e] = __hiloint2double(agpr_get(MP.hi[m]), agpr_get(MP.lo[m]));
        }
        for (int i = LANE; i < 3 * T.nbody; i += UHC_WAVE) {
            A.s.xpos[(size_t)env * 3 * T.nbody + i] = S[L.xpos + i];
            A.s.xipos[(size_t)env * 3 * T.nbody + i] = S[L.xipos + i];
        }
        for (int i = LANE; i < 4 * T.nbody; i += UHC_WAVE) A.s.xquat[(size_t)env * 4 * T.nbody + i] = S[L.xquat + i];
        if (A.c.rfc_mode == 2) {
            for (int i = LANE; i < 6 * T.nv; i += UHC_WAVE) A.s.cdof[(size_t)env * 6 * T.nv + i] = S[L.cdof + i];
            for (int i = LANE; i < 3 * T.nbody; i += UHC_WAVE) A.s.rootcom[(size_t)env * 3 * T.nbody + i] = S[L.rootcom + i];
        }
    }
#ifdef UHC_STAGE_PROF
    PROF(15)
    if (A.s.prof) {
        long long mine = 0;
#pragma unroll
        for (int k = 0; k < UHC_NPROF; k++) if (LANE == k) mine = pt_[k];
        if (LANE < UHC_NPROF) A.s.prof[(size_t)env * UHC_NPROF + LANE] += mine;
    }
#endif
    TRACE(2 * (TIER - 1) + 1, wall_clock64())
    double vmax_end = 0.0;  // largest |qvel| at the end of the step (the launch order of the next one)
    if (MODE == 0) {
        for (int i = LANE; i < T.nv; i += UHC_WAVE) vmax_end = fmax(vmax_end, fabs(S[L.qvel + i]));
        vmax_end = -wave_min(-vmax_end);
    }
    if (LANE == 0) {
        if (ran) { A.s.ncon[env] = fo.ncon; A.s.nefc[env] = fo.nefc; A.s.solver_iter[env] = fo.iters; }
        A.s.fail[env] = fail;
        if (overflow & 3) A.s.overflow[env] = 1;
        A.s.fresh[env] = 0;
        if (TIER == 2) A.s.pend2[env] = 0;  // taken: the chained launch of this tier has nothing left to do for the env
        if (TIER >= 3) A.s.pend3[env] = 0;
        // where the env's next step starts (uhc_batch_set_kernel_path 2): an env comes down a tier only with room to spare
        // An env that comes CLOSE to a tier's capacity starts its next step one tier up: finding out in the middle of a step that it no
        // longer fits costs that tier's work so far, and the step then ends a whole general- (or large-) tier env-step after the moment
        // of the hand-on -- with a thousand envs some env does that every step, and every step lasts fast + general + large.  It comes
        // down again only well below the mark (hysteresis).  The marks are the batch's (KernelArgs::marks, UHC_TIER_MARKS).
        if (MODE == 0 && ran) {
            const int* mk = A.marks;  // up2: rows, contacts, body-body rows | dn1: the same | up3, dn2: eighths of the general tier's capacities
            const bool up2 = pk_nefc > mk[0] || pk_ncon > mk[1] || pk_ntwo > mk[2];   // of 64 rows / 16 contacts / 12 body-body rows
            // (... and only if its packed rows fit the fast tier's storage: with objects in the model that storage is small, and an env that
            //  fits by rows and contacts alone came down every step only to be handed on in its first forward pass)
            const bool dn1 = pk_nefc <= mk[3] && pk_ncon <= mk[4] && pk_ntwo <= mk[5] && (TIER == 1 || pk_y + 8 <= (7 * A.cf.ycap) / 8);
            const bool up3 = pk_nefc > (mk[6] * A.cg.maxefc) / 8 || pk_ncon > (mk[6] * A.cg.maxcon) / 8 || pk_ntwo > (mk[6] * A.cg.ndense) / 8 || pk_y > (mk[6] * A.cg.ycap) / 8;
            const bool dn2 = pk_nefc <= (mk[7] * A.cg.maxefc) / 8 && pk_ncon <= (mk[7] * A.cg.maxcon) / 8 && pk_ntwo <= (mk[7] * A.cg.ndense) / 8 && pk_y <= (mk[7] * A.cg.ycap) / 8;
            const int big = A.last_tier >= 3 ? 3 : 2;
            int next;
            if (TIER == 1) next = up2 ? 2 : 1;
            else if (TIER == 2) next = (up3 && big == 3) ? 3 : (dn1 ? 1 : 2);
            else next = !dn2 ? 3 : (dn1 ? 1 : 2);
            // an env that needed tier 4 (beyond the large tier's capacities, or more force-carrying rows than its working sets finish) starts its next
            // step THERE: at the head of the queue of tier 4's consumers (uhc_capi.cpp) instead of an abandoned large-tier attempt first.  It comes down with room to spare: 3/4 of the large tier's capacities and <= 48 rows with a force.
            // OPT-IN (UHC_DEBUG bit 12): measured on the random-policy ball-joint rollouts it LOSES -- configs[4] 41.6 k -> 27.1 k env-steps/s: the envs that
            // reach tier 4 there are diverging ones that reset within two or three steps, a reset env then runs its first step in tier 4 too, and a whole step of
            // primal solves costs more than the large-tier attempt it saves -- so by default such an env starts its next step in the large tier.
            // Heavy envs go STRAIGHT to tier 4 (round 6): the four-wave Newton iteration on the primal costs the same whatever the number of rows, while a
            // general- or large-tier env-step grows with every working-set round -- and the step's slowest env is what a control step waits for.  An env
            // whose step peaked at KernelArgs::t4_rows rows or more starts its next step at the head of tier 4's queue and stays while it peaks above 3/4 of that.
            if (A.last_tier == 4 && A.t4_rows > 0 && ((TIER == 2 || TIER == 3) ? pk_nefc >= A.t4_rows : (TIER == 4 && 4 * pk_nefc >= 3 * A.t4_rows))) next = 4;
            // (Sending SLOW envs there too -- by the duration of their last general-tier step -- was built and measured in round 6 and is gone again: a tier-4 env-step of a
            //  60-80-row env is no shorter than its general-tier one, and a whole CU per env is what the fast tier's second round needs: profiles/r06_q_t4ticks_sweep.txt.)
            if (TIER == 4 && UHC_EXP(4096) && A.last_tier == 4 && !(pk_nefc <= (3 * A.ch.maxefc) / 4 && pk_ncon <= (3 * A.ch.maxcon) / 4 && pk_ntwo <= (3 * A.ch.ndense) / 4 && pk_act <= 48)) next = 4;
            A.s.tier[env] = next;
            // the fast tier's launch order (uhc_tier_lists_kernel): how close the step came to ANY of the fast tier's capacities, in sixty-fourths -- rows, contacts,
            // body-body rows and (round 6) the packed row storage, which names 2 of 3 hand-ons of the ball-joint rollouts (tools/tier_trace.py) -- and at the top of
            // the scale an env whose joints spin faster than any tracked motion's: a simulation on its way to the bad-value flag piles up rows within a few
            // substeps, goes through every tier and ends the step if it starts in the launch's last round (profiles/r06_b_tier_trace_configs4.txt)
            int cost = max(pk_nefc, max((pk_ncon * UHC_FAST_MAXEFC) / max(A.cf.maxcon, 1), (pk_ntwo * UHC_FAST_MAXEFC) / max(A.cf.ndense, 1)));
            cost = max(cost, (int)(((long long)pk_y * UHC_FAST_MAXEFC) / max(A.cf.ycap, 1)));
            if (vmax_end > 60.0) cost = max(cost, UHC_FAST_MAXEFC); else if (vmax_end > 30.0) cost = max(cost, 40);
            A.s.cost[env] = cost;
        }
        if (TIER != 1) A.s.redo[env] = 1 | ((overflow & 4) ? 2 : 0) | ((overflow >> 1) & 0x3c) | (TIER >= 3 ? 0x40 : 0) | ((overflow & 2) ? 0x80 : 0) | swept |
                                       ((overflow & 128) ? (1 << 30) : 0) | ((overflow & 256) ? (1 << 29) : 0);
        else if (overflow & 2) A.s.redo[env] = 0x80;  // (fast tier in truncate mode; bit 7 = rows / contacts were dropped in this step)
        if (TIER != 1 && MODE == 0) {
            atomicAdd(A.s.path_stats + 2, 1ull);
            if (fits) atomicAdd(A.s.path_stats + 1, 1ull);
        }  // UHC_F_REDO: bit 0 = computed by the general kernel, bit 1 = its contact solve ran the sweeps
    }
    return 0;
}

// The launch forms.  (1) one workgroup per env of the batch (blockIdx = env), filtered by the active mask and -- under sticky tiers -- by
// the tier the env starts its step in: the fast tier's launch skips the envs that have a launch of their own this step.  (2) a PERSISTENT
// launch over an env QUEUE (sticky general / large tiers): the queue starts with the envs that begin the step in this tier and grows by
// the envs the tier below hands on WHILE both launches run; every workgroup takes envs off the queue until it is empty AND all its
// producers have finished (a counter every producer workgroup bumps on exit).  The grid is as large as the host expects the queue to
// get: an underestimate costs time, never an env -- and no workgroup is started (and has to be given its 79 / 160 KiB of LDS) only to
// find that its env belongs to another tier.  A handed-on env is picked up a fraction of a step after the fast tier found it too big,
// instead of after the fast tier's whole launch.
// Not every consumer WAITS when the queue is empty.  A workgroup that found an env when it came in is a WORKER: it works the queue off
// and leaves when there is nothing left -- its 79 KiB go back to the launch beside it, whose second round can use them.  A workgroup
// that found the queue empty when it came in (the host sizes the launch for the envs the step begins with PLUS the hand-ons it expects)
// is a SPARE: it takes one of n_wait seats and waits for the producers' hand-ons until they have all finished; with no seat left it
// leaves too (with one consumer per expected env and all of them waiting, an overestimate starved the fast tier to the point that the
// idle consumers ran into their time-out).
__device__ __forceinline__ int queue_claim(const KernelArgs& A, int& role, const bool quiet) {  // lane 0 only; role: 0 undecided, 1 worker, 2 spare
    const unsigned long long t0 = wall_clock64();  // 100 MHz
    for (;;) {
        const int f = A.prod_fin ? __hip_atomic_load(A.prod_fin, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : 0;
        const int c = __hip_atomic_load(A.list_count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        const int cur = __hip_atomic_load(A.list_cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur < c) {
            if (atomicCAS(A.list_cursor, cur, cur + 1) != cur) continue;
            int env;  // (the producer bumps the count before it fills the slot: slots start at -1)
            while ((env = __hip_atomic_load(A.list + cur, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < 0) __builtin_amdgcn_s_sleep(8);
            if (role == 0) role = 1;
            return env;
        }
        if (!A.prod_fin || role == 1 || f >= A.prod_total) return -1;  // (f was read before the count: every append of a finished producer was seen)
        if (role == 0) {
            if (A.spares && atomicAdd(A.spares, 1) >= A.n_wait) return -1;
            role = 2;
        }
        // Never wait for ever: if the producers' launch cannot run beside this one (streams that share a hardware queue run in order) the
        // wait would not end.  After 50 ms with nothing to do the consumer leaves; what is handed on later stays flagged for the chained
        // launches, and the host stops starting consumers when it sees the count (DevState::q_abort).
        // (quiet: tier 4's consumers -- they wait behind the large tier's, which in a step of the general tier's majority regime may last longer than that
        //  without anything being wrong; when one leaves, what is handed on later stays flagged for the chained launch as well)
        if (wall_clock64() - t0 > 5000000ull) { if (!quiet) atomicAdd(A.s.q_abort, 1); return -1; }
        __builtin_amdgcn_s_sleep(64);
    }
}
template <int MODE, int TIER, bool DENSE>
__global__ void __launch_bounds__(UHC_WAVE) uhc_step_kernel(KernelArgs A, const double* __restrict__ d_action,
                                                            const double* __restrict__ d_tbase, const int* __restrict__ d_active) {
    const int env = (A.order && (int)blockIdx.x < A.n_env) ? A.order[blockIdx.x] : (int)blockIdx.x;
    bool go = env >= 0 && env < A.n_env && !(d_active && !d_active[env]);  // (env < 0: a free slot of a list launch, A.order)
    if (go && A.tier_want) {  // sticky tiers: the envs whose tier has its own launch this step are not this launch's
        const int t = A.s.tier_now[env];
        go = t == A.tier_want || !((A.sticky_mask >> t) & 1);
    }
    if (go) {
#ifdef UHC_WITH_TIER4
        if constexpr (TIER == 3 && MODE != 2) {
            // an env a large-tier queue consumer has already found too big (pend3 = 2) skips the large tier's attempt
            int handed = (A.last_tier == 4 && d_active == A.s.pend3 && A.s.pend3[env] == 2) ? 1 : 0;
            if (!handed) handed = uhc_step_env<MODE, TIER, DENSE>(A, d_action, d_tbase, env);
            if (handed && A.last_tier == 4) {  // the same workgroup, the same LDS allocation carved for tier 4: the env's state is where the hand-on left it
                wsync();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                uhc_step_env<MODE, 4, true>(A, d_action, d_tbase, env);
            }
        } else uhc_step_env<MODE, TIER, DENSE>(A, d_action, d_tbase, env);
#else
        uhc_step_env<MODE, TIER, DENSE>(A, d_action, d_tbase, env);
#endif
    }
    if (A.fin && LANE == 0) { __threadfence(); atomicAdd(A.fin, 1); }  // producer bookkeeping of the queues
}
// (2): its own entry point, so that the one-workgroup-per-env kernels keep the register allocation of a straight-line body
#define UHC_QUEUE_THREADS (UHC_WAVE * UHC_NWG)
template <int MODE, int TIER, bool DENSE>
__global__ void __launch_bounds__(UHC_QUEUE_THREADS) uhc_step_queue_kernel(KernelArgs A, const double* __restrict__ d_action, const double* __restrict__ d_tbase) {
#if defined(UHC_NW4)
    static_assert(TIER == 4, "the four-wave form is tier 4's");
    if (threadIdx.x >= UHC_WAVE) { primal_helper<TIER>(A, smem); return; }  // (before anything that says LANE == 0: the helpers' lanes count from 0 too)
#elif defined(UHC_NW2)
    static_assert(TIER == 2, "the two-wave form is the general tier's");
    if (threadIdx.x >= UHC_WAVE) { mpr_helper<TIER>(A, smem); return; }
#endif
    if (A.started && LANE == 0) atomicAdd(A.started, 1);  // resident: holds its LDS from here on
    // (tier trace, UHC_DEBUG bit 4: the consumer's own record -- entry, first env claimed, exit, envs processed -- in words 8 .. 11 (general
    //  tier) / 12 .. 15 (large tier) of the stage-profile record of env blockIdx.x)
    long long* tr = (TIER < 4 && (A.dbg & 16) && (int)blockIdx.x < A.n_env) ? A.s.prof + (size_t)blockIdx.x * UHC_NPROF + 8 + 4 * (TIER - 2) : nullptr;
    if (tr && LANE == 0) { tr[0] = (long long)wall_clock64(); tr[1] = 0; tr[3] = 0; }
    int role = 0;
    for (;;) {
        int env = -1;
        if (LANE == 0) env = queue_claim(A, role, TIER == 4);
        env = __builtin_amdgcn_readfirstlane(env);
        if (env < 0) break;
        if (tr && LANE == 0) { if (tr[1] == 0) tr[1] = (long long)wall_clock64(); tr[3]++; }
        // (a large-tier consumer that finds its env too big flags it, pend3 = 2, and appends it to TIER 4's OWN queue when that has consumers this step
        //  (uhc_k_huge_q.hip: this kernel instantiated for tier 4 alone); without them the step's chained large-tier launch takes it to tier 4.  Tier 4
        //  INLINE in the large tier's persistent consumer hung the rollout at its first use -- the 619-spill instantiation, never explained; and it
        //  would hold the large tier's queue up for a 10-40 ms env-step)
        uhc_step_env<MODE, TIER, DENSE>(A, d_action, d_tbase, env);
        wsync();
    }
    if (tr && LANE == 0) tr[2] = (long long)wall_clock64();
    if (A.fin && LANE == 0) { __threadfence(); atomicAdd(A.fin, 1); }  // consumer bookkeeping (the next tier's consumers wait for it)
#if defined(UHC_NW4)
    primal_release_helpers<TIER>(A, smem);
#elif defined(UHC_NW2)
    mpr_release_helpers<TIER>(A, smem);
#endif
}
