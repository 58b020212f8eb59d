/*
 * physics_oracle.c -- TEST INFRASTRUCTURE.  CPU float64 restatement, one environment, of the
 * rigid-body step the reference obtains from MuJoCo 2.1.0 (`self.sim.step()`,
 * uhc/envs/humanoid_im.py:1177; `sim.forward()`, uhc/khrylib/rl/envs/common/mujoco_env.py:113)
 * plus the Python-side controller wrapped around it (uhc/envs/humanoid_im.py:1014-1190).
 *
 * PARITY UNPINNED for the MuJoCo stages: MuJoCo 2.1.0 (mujoco-py>=2.1,<2.2, requirements.txt:11)
 * is an un-vendored binary dependency that is absent from /root/reference and from this image, and
 * the reference ships no physics golden vectors.  Every stage tagged [MJ-ext]
 * restates MuJoCo's published algorithm (engine_core_smooth / engine_core_constraint /
 * engine_collision_convex / engine_solver / engine_forward) from its documentation; what pins it
 * here are analytic known-answer tests (tests/test_oracle_physics.py) and the reference's own
 * independent FK (uhc/smpllib/torch_smpl_humanoid.py:303-362, golden fixture).
 * The controller part IS pinned against the imported reference (tests/golden/g5_pd_rfc.npz).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object.  The product path (uhc_amd/csrc) never links or calls it.
 *
 * Written for readability, not speed: dense loops over nv where MuJoCo uses chains, no SIMD.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "../include/uhc_amd.h"
#include "physics_oracle.h"

#define MINVAL 1e-15   /* [MJ-ext] mjMINVAL */
#define MAXVAL 1e10    /* [MJ-ext] mjMAXVAL */

/* ------------------------------------------------------------------ small math */
static void cross3(double r[3], const double a[3], const double b[3]) {
    r[0] = a[1] * b[2] - a[2] * b[1];
    r[1] = a[2] * b[0] - a[0] * b[2];
    r[2] = a[0] * b[1] - a[1] * b[0];
}
static double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double dot6(const double a[6], const double b[6]) {
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
static void quat_mul(double r[4], const double a[4], const double b[4]) {
    double t[4];
    t[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    t[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    t[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    t[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    memcpy(r, t, sizeof t);
}
static void quat_normalize(double q[4]) {
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static void quat_to_mat(double m[9], const double q[4]) {
    double w = q[0], x = q[1], y = q[2], z = q[3];
    m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
    m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
    m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
static void mat_vec(double r[3], const double m[9], const double v[3]) {
    double t0 = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
    double t1 = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
    double t2 = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
    r[0] = t0; r[1] = t1; r[2] = t2;
}
static void axis_angle_quat(double q[4], const double axis[3], double angle) {
    double s = sin(0.5 * angle);
    q[0] = cos(0.5 * angle); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
/* spatial vectors are [angular(3); linear(3)] */
static void cross_motion(double r[6], const double v[6], const double m[6]) { /* [MJ-ext] mju_crossMotion */
    double a[3], b[3];
    cross3(r, v, m);
    cross3(a, v, m + 3);
    cross3(b, v + 3, m);
    r[3] = a[0] + b[0]; r[4] = a[1] + b[1]; r[5] = a[2] + b[2];
}
static void cross_force(double r[6], const double v[6], const double f[6]) { /* [MJ-ext] mju_crossForce */
    double a[3], b[3];
    cross3(a, v, f);
    cross3(b, v + 3, f + 3);
    r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2];
    cross3(r + 3, v, f + 3);
}
/* 10-number spatial inertia about a reference point: Ixx Iyy Izz Ixy Ixz Iyz | m*c (3) | m */
static void inert_mul(double r[6], const double I[10], const double v[6]) { /* [MJ-ext] mju_mulInertVec */
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

/* ------------------------------------------------------------------ data */
OrcData* orc_data_create(const UhcModelDesc* m) {
    OrcData* d = (OrcData*)calloc(1, sizeof(OrcData));
    int nb = m->nbody, nv = m->nv, nM = m->dof_madr[nv];
    d->nM = nM;
    d->qpos = calloc(m->nq, 8); d->qvel = calloc(nv, 8); d->qacc = calloc(nv, 8);
    d->qacc_warmstart = calloc(nv, 8); d->ctrl = calloc(m->nu > 0 ? m->nu : 1, 8);
    d->qfrc_applied = calloc(nv, 8);
    d->xpos = calloc(nb * 3, 8); d->xquat = calloc(nb * 4, 8); d->xmat = calloc(nb * 9, 8);
    d->xipos = calloc(nb * 3, 8); d->ximat = calloc(nb * 9, 8);
    d->xanchor = calloc(m->njnt * 3, 8); d->xaxis = calloc(m->njnt * 3, 8);
    d->subtree_com = calloc(nb * 3, 8); d->cinert = calloc(nb * 10, 8); d->crb = calloc(nb * 10, 8);
    d->cdof = calloc(nv * 6, 8); d->cdof_dot = calloc(nv * 6, 8); d->cvel = calloc(nb * 6, 8);
    d->qM = calloc(nM, 8); d->qLD = calloc(nM, 8);
    d->qfrc_bias = calloc(nv, 8); d->qfrc_passive = calloc(nv, 8); d->qfrc_actuator = calloc(nv, 8);
    d->qfrc_smooth = calloc(nv, 8); d->qacc_smooth = calloc(nv, 8); d->qfrc_constraint = calloc(nv, 8);
    d->con_pos = calloc(ORC_MAXCON * 3, 8); d->con_frame = calloc(ORC_MAXCON * 9, 8);
    d->con_dist = calloc(ORC_MAXCON, 8); d->con_margin = calloc(ORC_MAXCON, 8);
    d->con_friction = calloc(ORC_MAXCON * 3, 8); d->con_solref = calloc(ORC_MAXCON * 2, 8);
    d->con_solimp = calloc(ORC_MAXCON * 5, 8);
    d->con_geom1 = calloc(ORC_MAXCON, 4); d->con_geom2 = calloc(ORC_MAXCON, 4); d->con_dim = calloc(ORC_MAXCON, 4);
    d->efc_J = calloc((size_t)ORC_MAXEFC * nv, 8); d->efc_pos = calloc(ORC_MAXEFC, 8);
    d->efc_margin = calloc(ORC_MAXEFC, 8); d->efc_R = calloc(ORC_MAXEFC, 8); d->efc_D = calloc(ORC_MAXEFC, 8);
    d->efc_aref = calloc(ORC_MAXEFC, 8); d->efc_b = calloc(ORC_MAXEFC, 8); d->efc_force = calloc(ORC_MAXEFC, 8);
    d->efc_vel = calloc(ORC_MAXEFC, 8); d->efc_diagApprox = calloc(ORC_MAXEFC, 8);
    d->efc_floss = calloc(ORC_MAXEFC, 8);
    d->efc_type = calloc(ORC_MAXEFC, 4); d->efc_id = calloc(ORC_MAXEFC, 4); d->efc_edge = calloc(ORC_MAXEFC, 4);
    d->ar_cap = 128; d->efc_AR = calloc((size_t)d->ar_cap * d->ar_cap, 8);
    d->work = calloc((size_t)ORC_MAXEFC * nv + 16 * nv, 8);
    memcpy(d->qpos, m->qpos0, m->nq * 8);
    return d;
}
void orc_data_free(OrcData* d) {
    if (!d) return;
    free(d->qpos); free(d->qvel); free(d->qacc); free(d->qacc_warmstart); free(d->ctrl); free(d->qfrc_applied);
    free(d->xpos); free(d->xquat); free(d->xmat); free(d->xipos); free(d->ximat); free(d->xanchor); free(d->xaxis);
    free(d->subtree_com); free(d->cinert); free(d->crb); free(d->cdof); free(d->cdof_dot); free(d->cvel);
    free(d->qM); free(d->qLD); free(d->qfrc_bias); free(d->qfrc_passive); free(d->qfrc_actuator);
    free(d->qfrc_smooth); free(d->qacc_smooth); free(d->qfrc_constraint);
    free(d->con_pos); free(d->con_frame); free(d->con_dist); free(d->con_margin); free(d->con_friction);
    free(d->con_solref); free(d->con_solimp); free(d->con_geom1); free(d->con_geom2); free(d->con_dim);
    free(d->efc_J); free(d->efc_pos); free(d->efc_margin); free(d->efc_R); free(d->efc_D); free(d->efc_aref);
    free(d->efc_b); free(d->efc_force); free(d->efc_vel); free(d->efc_diagApprox); free(d->efc_floss);
    free(d->efc_type); free(d->efc_id); free(d->efc_edge); free(d->efc_AR); free(d->work);
    free(d);
}

/* ------------------------------------------------------------------ P1: mj_kinematics [MJ-ext] */
void orc_kinematics(const UhcModelDesc* m, OrcData* d) {
    d->xpos[0] = d->xpos[1] = d->xpos[2] = 0;
    d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0;
    quat_to_mat(d->xmat, d->xquat);
    memset(d->xipos, 0, 24); quat_to_mat(d->ximat, d->xquat);
    for (int b = 1; b < m->nbody; b++) {
        int p = m->body_parentid[b], ja = m->body_jntadr[b], jn = m->body_jntnum[b];
        double pos[3], quat[4], R[9], t[3];
        if (jn == 1 && m->jnt_type[ja] == UHC_JNT_FREE) {
            const double* q = d->qpos + m->jnt_qposadr[ja];
            memcpy(pos, q, 24); memcpy(quat, q + 3, 32);
            quat_normalize(quat);
            memcpy(d->xanchor + 3 * ja, pos, 24);
            d->xaxis[3 * ja] = 0; d->xaxis[3 * ja + 1] = 0; d->xaxis[3 * ja + 2] = 1;
        } else {
            mat_vec(t, d->xmat + 9 * p, m->body_pos + 3 * b);
            for (int k = 0; k < 3; k++) pos[k] = d->xpos[3 * p + k] + t[k];
            quat_mul(quat, d->xquat + 4 * p, m->body_quat + 4 * b);
            for (int j = ja; j < ja + jn; j++) {
                int qa = m->jnt_qposadr[j];
                double qloc[4];
                quat_to_mat(R, quat);
                mat_vec(t, R, m->jnt_pos + 3 * j);
                for (int k = 0; k < 3; k++) d->xanchor[3 * j + k] = pos[k] + t[k];
                mat_vec(d->xaxis + 3 * j, R, m->jnt_axis + 3 * j);
                switch (m->jnt_type[j]) {
                case UHC_JNT_SLIDE:
                    for (int k = 0; k < 3; k++) pos[k] += d->xaxis[3 * j + k] * (d->qpos[qa] - m->qpos0[qa]);
                    continue;
                case UHC_JNT_HINGE:
                    axis_angle_quat(qloc, m->jnt_axis + 3 * j, d->qpos[qa] - m->qpos0[qa]);
                    quat_mul(quat, quat, qloc);
                    break;
                case UHC_JNT_BALL:
                    memcpy(qloc, d->qpos + qa, 32);
                    quat_normalize(qloc);
                    quat_mul(quat, quat, qloc);
                    break;
                default: break;
                }
                /* keep the anchor fixed under the joint's rotation */
                quat_to_mat(R, quat);
                mat_vec(t, R, m->jnt_pos + 3 * j);
                for (int k = 0; k < 3; k++) pos[k] = d->xanchor[3 * j + k] - t[k];
            }
        }
        quat_normalize(quat);
        memcpy(d->xpos + 3 * b, pos, 24); memcpy(d->xquat + 4 * b, quat, 32);
        quat_to_mat(d->xmat + 9 * b, quat);
        mat_vec(t, d->xmat + 9 * b, m->body_ipos + 3 * b);
        for (int k = 0; k < 3; k++) d->xipos[3 * b + k] = pos[k] + t[k];
        double qi[4];
        quat_mul(qi, quat, m->body_iquat + 4 * b);
        quat_to_mat(d->ximat + 9 * b, qi);
    }
}

/* ------------------------------------------------------------------ P2: mj_comPos [MJ-ext] */
static int body_rootid(const UhcModelDesc* m, int b) {
    while (b > 0 && m->body_parentid[b] > 0) b = m->body_parentid[b];
    return b;
}
void orc_com_pos(const UhcModelDesc* m, OrcData* d) {
    int nb = m->nbody;
    double* mass = d->work; /* subtree masses */
    for (int b = 0; b < nb; b++) {
        mass[b] = m->body_mass[b];
        for (int k = 0; k < 3; k++) d->subtree_com[3 * b + k] = m->body_mass[b] * d->xipos[3 * b + k];
    }
    for (int b = nb - 1; b > 0; b--) {
        int p = m->body_parentid[b];
        mass[p] += mass[b];
        for (int k = 0; k < 3; k++) d->subtree_com[3 * p + k] += d->subtree_com[3 * b + k];
    }
    for (int b = 0; b < nb; b++)
        for (int k = 0; k < 3; k++)
            d->subtree_com[3 * b + k] = mass[b] < MINVAL ? d->xipos[3 * b + k] : d->subtree_com[3 * b + k] / mass[b];
    /* body inertias about the subtree COM of the kinematic-tree root, world-aligned */
    memset(d->cinert, 0, 80);
    for (int b = 1; b < nb; b++) {
        const double* c0 = d->subtree_com + 3 * body_rootid(m, b);
        const double* R = d->ximat + 9 * b;
        const double* I = m->body_inertia + 3 * b;
        double ms = m->body_mass[b], c[3], *ci = d->cinert + 10 * b;
        for (int k = 0; k < 3; k++) c[k] = d->xipos[3 * b + k] - c0[k];
        /* R diag(I) R^T */
        double J[9];
        for (int r = 0; r < 3; r++)
            for (int s = 0; s < 3; s++)
                J[3 * r + s] = R[3 * r] * I[0] * R[3 * s] + R[3 * r + 1] * I[1] * R[3 * s + 1] + R[3 * r + 2] * I[2] * R[3 * s + 2];
        double cc = dot3(c, c);
        ci[0] = J[0] + ms * (cc - c[0] * c[0]);
        ci[1] = J[4] + ms * (cc - c[1] * c[1]);
        ci[2] = J[8] + ms * (cc - c[2] * c[2]);
        ci[3] = J[1] - ms * c[0] * c[1];
        ci[4] = J[2] - ms * c[0] * c[2];
        ci[5] = J[5] - ms * c[1] * c[2];
        ci[6] = ms * c[0]; ci[7] = ms * c[1]; ci[8] = ms * c[2]; ci[9] = ms;
    }
    /* motion axes of the dofs, expressed at that same point */
    for (int j = 0; j < m->njnt; j++) {
        int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j];
        const double* c0 = d->subtree_com + 3 * body_rootid(m, b);
        double off[3], *cd = d->cdof + 6 * da;
        for (int k = 0; k < 3; k++) off[k] = c0[k] - d->xanchor[3 * j + k];
        switch (m->jnt_type[j]) {
        case UHC_JNT_FREE:
            memset(cd, 0, 18 * 8);
            cd[3] = 1; cd[6 + 4] = 1; cd[12 + 5] = 1;
            cd += 18; /* then rotations about the body axes */
            /* fall through */
        case UHC_JNT_BALL:
            for (int k = 0; k < 3; k++) {
                double ax[3] = {d->xmat[9 * b + k], d->xmat[9 * b + 3 + k], d->xmat[9 * b + 6 + k]};
                memcpy(cd + 6 * k, ax, 24);
                cross3(cd + 6 * k + 3, ax, off);
            }
            break;
        case UHC_JNT_SLIDE:
            memset(cd, 0, 24); memcpy(cd + 3, d->xaxis + 3 * j, 24);
            break;
        case UHC_JNT_HINGE:
            memcpy(cd, d->xaxis + 3 * j, 24);
            cross3(cd + 3, d->xaxis + 3 * j, off);
            break;
        }
    }
}

/* ------------------------------------------------------------------ P3: mj_crb + mj_factorM [MJ-ext] */
void orc_crb(const UhcModelDesc* m, OrcData* d) {
    int nb = m->nbody, nv = m->nv;
    memcpy(d->crb, d->cinert, nb * 80);
    for (int b = nb - 1; b > 0; b--) {
        int p = m->body_parentid[b];
        if (p > 0) for (int k = 0; k < 10; k++) d->crb[10 * p + k] += d->crb[10 * b + k];
    }
    for (int i = 0; i < nv; i++) {
        double buf[6];
        int adr = m->dof_madr[i];
        inert_mul(buf, d->crb + 10 * m->dof_bodyid[i], d->cdof + 6 * i);
        for (int j = i; j >= 0; j = m->dof_parentid[j]) d->qM[adr++] = dot6(d->cdof + 6 * j, buf);
        d->qM[m->dof_madr[i]] += m->dof_armature[i];
    }
}
/* in-place L^T D L of a tree-sparse matrix given in qM layout */
void orc_factor_sparse(const UhcModelDesc* m, double* LD) {
    for (int k = m->nv - 1; k >= 0; k--) {
        int kk = m->dof_madr[k], ki = kk + 1;
        for (int i = m->dof_parentid[k]; i >= 0; i = m->dof_parentid[i], ki++) {
            double a = LD[ki] / LD[kk];
            int n = m->dof_madr[i + 1] - m->dof_madr[i]; /* entries of row i == tail of row k from ki */
            for (int t = 0; t < n; t++) LD[m->dof_madr[i] + t] -= a * LD[ki + t];
            LD[ki] = a;
        }
    }
}
void orc_solve_sparse(const UhcModelDesc* m, const double* LD, double* x) {
    int nv = m->nv;
    for (int i = nv - 1; i >= 0; i--) {
        int adr = m->dof_madr[i] + 1;
        for (int j = m->dof_parentid[i]; j >= 0; j = m->dof_parentid[j]) x[j] -= LD[adr++] * x[i];
    }
    for (int i = 0; i < nv; i++) x[i] /= LD[m->dof_madr[i]];
    for (int i = 0; i < nv; i++) {
        int adr = m->dof_madr[i] + 1;
        for (int j = m->dof_parentid[i]; j >= 0; j = m->dof_parentid[j]) x[i] -= LD[adr++] * x[j];
    }
}
void orc_factor_m(const UhcModelDesc* m, OrcData* d) {
    memcpy(d->qLD, d->qM, d->nM * 8);
    orc_factor_sparse(m, d->qLD);
}
void orc_full_m(const UhcModelDesc* m, const double* qM, double* dense) { /* [MJ-ext] mj_fullM */
    int nv = m->nv;
    memset(dense, 0, (size_t)nv * nv * 8);
    for (int i = 0; i < nv; i++) {
        int adr = m->dof_madr[i];
        for (int j = i; j >= 0; j = m->dof_parentid[j], adr++) dense[i * nv + j] = dense[j * nv + i] = qM[adr];
    }
}

/* ------------------------------------------------------------------ Jacobian of a world point on a body */
static void jac_point(const UhcModelDesc* m, const OrcData* d, int body, const double p[3], double* jacp /*3*nv*/) {
    int nv = m->nv;
    memset(jacp, 0, 3 * nv * 8);
    if (body <= 0) return;
    const double* c0 = d->subtree_com + 3 * body_rootid(m, body);
    double off[3] = {p[0] - c0[0], p[1] - c0[1], p[2] - c0[2]};
    /* last dof of the nearest ancestor that has dofs, then walk the dof chain */
    int b = body;
    while (b > 0 && m->body_dofnum[b] == 0) b = m->body_parentid[b];
    if (b <= 0) return;
    for (int i = m->body_dofadr[b] + m->body_dofnum[b] - 1; i >= 0; i = m->dof_parentid[i]) {
        double t[3];
        cross3(t, d->cdof + 6 * i, off);
        for (int k = 0; k < 3; k++) jacp[k * nv + i] = d->cdof[6 * i + 3 + k] + t[k];
    }
}

/* ------------------------------------------------------------------ P4: mj_collision [MJ-ext] */
static void make_frame(double f[9]) { /* [MJ-ext] mju_makeFrame: x given, y picked, z = x cross y */
    double n = sqrt(dot3(f, f));
    f[0] /= n; f[1] /= n; f[2] /= n;
    f[3] = f[4] = f[5] = 0;
    if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1;
    double dp = dot3(f, f + 3);
    for (int k = 0; k < 3; k++) f[3 + k] -= f[k] * dp;
    n = sqrt(dot3(f + 3, f + 3));
    f[3] /= n; f[4] /= n; f[5] /= n;
    cross3(f + 6, f, f + 3);
}
static int bodies_filtered(const UhcModelDesc* m, int b1, int b2) {
    if (b1 == b2) return 1;
    /* parent-child filter applies only when neither body is the world */
    if (b1 != 0 && b2 != 0 && (m->body_parentid[b1] == b2 || m->body_parentid[b2] == b1)) return 1;
    for (int e = 0; e < m->nexclude; e++) {
        int x = m->exclude_pair[2 * e], y = m->exclude_pair[2 * e + 1];
        if ((x == b1 && y == b2) || (x == b2 && y == b1)) return 1;
    }
    return 0;
}
/* Geoms that collide as convex hulls: meshes and the ROUNDED hulls -- a sphere (one core vertex) and a capsule (the two end points of its segment), pushed out by
 * their radius geom_size[0] along the query direction ([MJ-ext] mjc_Convex's support callbacks for mjGEOM_SPHERE / mjGEOM_CAPSULE: centre or segment end + radius * dir).
 * The model compiler (uhc_amd/model/mjcf.py) puts the core vertices into the mesh tables. */
static int is_hull(int t) { return t == UHC_GEOM_MESH || t == UHC_GEOM_SPHERE || t == UHC_GEOM_CAPSULE; }
static double hull_radius(const UhcModelDesc* m, int g) {
    int t = m->geom_type[g];
    return (t == UHC_GEOM_SPHERE || t == UHC_GEOM_CAPSULE) ? m->geom_size[3 * g] : 0.0;
}
/* [MJ-ext] mju_makeFrame with a y axis supplied (mjc_PlaneCapsule aligns the contact frame with the capsule's axis): y made orthogonal to x and
 * normalised, z = x cross y; a y (nearly) parallel to x falls back to the picked axis of make_frame */
static void make_frame_hint(double f[9], const double y[3]) {
    double n = sqrt(dot3(f, f));
    f[0] /= n; f[1] /= n; f[2] /= n;
    double dp = dot3(f, y), t[3];
    for (int k = 0; k < 3; k++) t[k] = y[k] - f[k] * dp;
    n = sqrt(dot3(t, t));
    if (n < 1e-8) { make_frame(f); return; }
    for (int k = 0; k < 3; k++) f[3 + k] = t[k] / n;
    cross3(f + 6, f, f + 3);
}
static void add_contact_hint(const UhcModelDesc* m, OrcData* d, int g1, int g2, const double pos[3],
                             const double normal[3], double dist, double margin, const double* yhint) {
    if (d->ncon >= ORC_MAXCON) return;
    int c = d->ncon++;
    memcpy(d->con_pos + 3 * c, pos, 24);
    memcpy(d->con_frame + 9 * c, normal, 24);
    if (yhint) make_frame_hint(d->con_frame + 9 * c, yhint);
    else make_frame(d->con_frame + 9 * c);
    d->con_dist[c] = dist;
    d->con_margin[c] = margin; /* includemargin = margin - gap; gap handled by caller */
    d->con_geom1[c] = g1; d->con_geom2[c] = g2;
    d->con_dim[c] = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
    for (int k = 0; k < 3; k++)
        d->con_friction[3 * c + k] = fmax(m->geom_friction[3 * g1 + k], m->geom_friction[3 * g2 + k]);
    /* equal solmix weights: plain average [MJ-ext] */
    for (int k = 0; k < 2; k++) d->con_solref[2 * c + k] = 0.5 * (m->geom_solref[2 * g1 + k] + m->geom_solref[2 * g2 + k]);
    for (int k = 0; k < 5; k++) d->con_solimp[5 * c + k] = 0.5 * (m->geom_solimp[5 * g1 + k] + m->geom_solimp[5 * g2 + k]);
}
static void add_contact(const UhcModelDesc* m, OrcData* d, int g1, int g2, const double pos[3],
                        const double normal[3], double dist, double margin) {
    add_contact_hint(m, d, g1, g2, pos, normal, dist, margin, NULL);
}
/* plane (g1) vs convex hull (g2): support vertex, then hull-graph neighbours within margin.  A rounded hull (sphere, capsule) is its core vertices lowered by the
 * radius: [MJ-ext] _PlaneSphere -- dist = n . (centre - plane) - radius, kept while dist <= margin, pos = centre - n (radius + dist / 2) -- at the sphere's centre or
 * the capsule's two segment ends (mjc_PlaneCapsule, which also aligns the contact frame's second axis with the capsule's axis).  The support vertex (the lower end) is
 * emitted first, the other end -- its only neighbour -- after it. */
static void collide_plane_mesh(const UhcModelDesc* m, OrcData* d, int g1, int g2, double margin, double gap) {
    int b1 = m->geom_bodyid[g1], b2 = m->geom_bodyid[g2];
    double pR[9], pq[4], ppos[3], t[3];
    /* plane pose */
    quat_mul(pq, d->xquat + 4 * b1, m->geom_quat + 4 * g1);
    quat_to_mat(pR, pq);
    mat_vec(t, d->xmat + 9 * b1, m->geom_pos + 3 * g1);
    for (int k = 0; k < 3; k++) ppos[k] = d->xpos[3 * b1 + k] + t[k];
    double n[3] = {pR[2], pR[5], pR[8]};
    /* bounding-sphere cull */
    mat_vec(t, d->xmat + 9 * b2, m->geom_center + 3 * g2);
    double cdist = 0;
    for (int k = 0; k < 3; k++) cdist += n[k] * (d->xpos[3 * b2 + k] + t[k] - ppos[k]);
    if (cdist - m->geom_rbound[g2] > margin) return;
    /* support vertex along -normal (first minimum wins) */
    int va = m->geom_vertadr[g2], vn = m->geom_vertnum[g2], best = -1;
    double bestd = 0, bw[3] = {0, 0, 0};
    for (int v = va; v < va + vn; v++) {
        double w[3];
        mat_vec(w, d->xmat + 9 * b2, m->mesh_vert + 3 * v);
        double dist = 0;
        for (int k = 0; k < 3; k++) { w[k] += d->xpos[3 * b2 + k]; dist += n[k] * (w[k] - ppos[k]); }
        if (best < 0 || dist < bestd) { best = v; bestd = dist; memcpy(bw, w, 24); }
    }
    const double rr = hull_radius(m, g2);
    bestd -= rr;
    if (best < 0 || bestd > margin) return;
    double cp[3], axis[3];
    const double* yh = NULL;
    if (m->geom_type[g2] == UHC_GEOM_CAPSULE && vn == 2) { /* the capsule's axis in the world: from its second core vertex to its first */
        double dl[3];
        for (int k = 0; k < 3; k++) dl[k] = m->mesh_vert[3 * va + k] - m->mesh_vert[3 * (va + 1) + k];
        mat_vec(axis, d->xmat + 9 * b2, dl);
        yh = axis;
    }
    for (int k = 0; k < 3; k++) cp[k] = bw[k] - (rr + 0.5 * bestd) * n[k];
    add_contact_hint(m, d, g1, g2, cp, n, bestd, margin - gap, yh);
    int cnt = 1;
    for (int e = m->mesh_adjadr[best]; e < m->mesh_adjadr[best + 1] && cnt < m->plane_mesh_maxcon; e++) {
        int v = m->mesh_adj[e];
        double w[3], dist = 0;
        mat_vec(w, d->xmat + 9 * b2, m->mesh_vert + 3 * v);
        for (int k = 0; k < 3; k++) { w[k] += d->xpos[3 * b2 + k]; dist += n[k] * (w[k] - ppos[k]); }
        dist -= rr;
        if (dist <= margin) {
            for (int k = 0; k < 3; k++) cp[k] = w[k] - (rr + 0.5 * dist) * n[k];
            add_contact_hint(m, d, g1, g2, cp, n, dist, margin - gap, yh);
            cnt++;
        }
    }
}
/* ---- convex-convex narrow phase: Minkowski Portal Refinement.
 * [MJ-ext] MuJoCo 2.1.0 sends mesh-mesh (and every other pair without an analytic routine) through mjc_Convex, i.e. libccd's
 * ccdMPRPenetration (libccd src/mpr.c, double precision build: CCD_EPS = DBL_EPSILON) with its own support / centre callbacks:
 *   centre  = geom_xpos (for a mesh: the hull's centre of mass),
 *   support = the hull vertex with the largest dot product with the direction (first maximum wins) + dir * margin / 2,
 *   tolerance = opt.mpr_tolerance (1e-6), iteration cap = opt.mpr_iterations (50),
 * and turns (depth, dir, pos) into ONE contact: dist = margin - depth, normal = dir (from geom 1 to geom 2), pos = the mid point of
 * the two witness points.  Restated from libccd's published algorithm (G. Snethen, "XenoCollide", Game Programming Gems 7); the
 * structure (discoverPortal / refinePortal / findPenetr / findPos, the zero and equality tests) follows libccd so that the same
 * portal is found, but nothing here is pinned against MuJoCo. */
#define CCD_EPS 2.220446049250313e-16
#define MPR_TOLERANCE 1e-6
#define MPR_MAXIT 50
#define MPR_MAXSUP 256 /* support points a pair may ask for before it counts as apart (uhc_mpr.h: UHC_MPR_MAXSUP) */
typedef struct { double v[3], v1[3], v2[3]; } CcdSup; /* a point of the Minkowski difference and its two witnesses */
typedef struct { const UhcModelDesc* m; const OrcData* d; int g1, g2; double margin; int idx1, idx2; /* hill-climb caches (vertex of the last support call) */ } CcdPair;
/* Sensitivity switch (tools/sensitivity.py; NOT part of the checked path): how a hull's support vertex is found.
 *   0 (default, what the device kernels do): scan all vertices, first maximum wins;
 *   1: hill-climbing on the hull graph from the vertex the previous support call of the same MPR run ended at (vertex 0 at first), to a
 *      vertex none of whose neighbours projects further -- [MJ-ext] mjc_MeshSupport takes this route when the mesh has a graph.  On a convex
 *      hull both find a maximum; they differ where several vertices project equally far (a face or edge normal to the direction). */
static int g_support_mode = 0;
void orc_set_support_mode(int mode) { g_support_mode = mode; }
static int ccd_zero(double x) { return fabs(x) < CCD_EPS; }
static int ccd_eq(double a, double b) {
    double ab = fabs(a - b);
    if (ab < CCD_EPS) return 1;
    a = fabs(a); b = fabs(b);
    return b > a ? ab < CCD_EPS * b : ab < CCD_EPS * a;
}
static void v3sub(double r[3], const double a[3], const double b[3]) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
static void v3norm(double a[3]) { double n = sqrt(dot3(a, a)); a[0] /= n; a[1] /= n; a[2] /= n; }
static void mesh_support(const UhcModelDesc* m, const OrcData* d, int g, const double dir[3], double margin, double out[3], int* cache) {
    int b = m->geom_bodyid[g], va = m->geom_vertadr[g], vn = m->geom_vertnum[g], best = va;
    const double* R = d->xmat + 9 * b;
    double loc[3] = {R[0] * dir[0] + R[3] * dir[1] + R[6] * dir[2], R[1] * dir[0] + R[4] * dir[1] + R[7] * dir[2],
                     R[2] * dir[0] + R[5] * dir[1] + R[8] * dir[2]}; /* R^T dir */
    double bd = -1e300;
    if (g_support_mode == 1 && cache && m->mesh_adjadr) {
        best = (*cache >= va && *cache < va + vn) ? *cache : va;
        bd = dot3(loc, m->mesh_vert + 3 * best);
        for (int moved = 1; moved;) {
            moved = 0;
            int from = best;
            for (int e = m->mesh_adjadr[from]; e < m->mesh_adjadr[from + 1]; e++) {
                int v = m->mesh_adj[e];
                double s = dot3(loc, m->mesh_vert + 3 * v);
                if (s > bd) { bd = s; best = v; moved = 1; }
            }
        }
        *cache = best;
    } else
    for (int v = va; v < va + vn; v++) {
        double s = dot3(loc, m->mesh_vert + 3 * v);
        if (s > bd) { bd = s; best = v; }
    }
    mat_vec(out, R, m->mesh_vert + 3 * best);
    const double hm = 0.5 * margin + hull_radius(m, g); /* a rounded hull's surface lies its radius beyond the core vertex */
    for (int k = 0; k < 3; k++) out[k] += d->xpos[3 * b + k] + dir[k] * hm;
}
static void geom_centre(const UhcModelDesc* m, const OrcData* d, int g, double out[3]) {
    int b = m->geom_bodyid[g];
    mat_vec(out, d->xmat + 9 * b, m->geom_center + 3 * g);
    for (int k = 0; k < 3; k++) out[k] += d->xpos[3 * b + k];
}
static void ccd_support(const CcdPair* P, const double dir[3], CcdSup* s) {
    double nd[3] = {-dir[0], -dir[1], -dir[2]};
    mesh_support(P->m, P->d, P->g1, dir, P->margin, s->v1, (int*)&P->idx1);
    mesh_support(P->m, P->d, P->g2, nd, P->margin, s->v2, (int*)&P->idx2);
    v3sub(s->v, s->v1, s->v2);
}
static void portal_dir(const CcdSup p[4], double dir[3]) {
    double a[3], b[3];
    v3sub(a, p[2].v, p[1].v); v3sub(b, p[3].v, p[1].v);
    cross3(dir, a, b);
    v3norm(dir);
}
static int portal_reach_tolerance(const CcdSup p[4], const CcdSup* v4, const double dir[3]) {
    double dv1 = dot3(p[1].v, dir), dv2 = dot3(p[2].v, dir), dv3 = dot3(p[3].v, dir), dv4 = dot3(v4->v, dir);
    double d1 = dv4 - dv1, d2 = dv4 - dv2, d3 = dv4 - dv3;
    d1 = fmin(d1, d2); d1 = fmin(d1, d3);
    return ccd_eq(d1, MPR_TOLERANCE) || d1 < MPR_TOLERANCE;
}
static void expand_portal(CcdSup p[4], const CcdSup* v4) {
    double v4v0[3];
    cross3(v4v0, v4->v, p[0].v);
    if (dot3(p[1].v, v4v0) > 0) {
        if (dot3(p[2].v, v4v0) > 0) p[1] = *v4; else p[3] = *v4;
    } else {
        if (dot3(p[3].v, v4v0) > 0) p[2] = *v4; else p[1] = *v4;
    }
}
static double point_seg_dist2(const double x0[3], const double b[3], double w[3]) { /* from the origin */
    double dd[3], t;
    v3sub(dd, b, x0);
    t = -dot3(x0, dd) / dot3(dd, dd);
    if (t < 0 || ccd_zero(t)) memcpy(w, x0, 24);
    else if (t > 1 || ccd_eq(t, 1)) memcpy(w, b, 24);
    else for (int k = 0; k < 3; k++) w[k] = dd[k] * t + x0[k];
    return dot3(w, w);
}
static double point_tri_dist2(const double x0[3], const double B[3], const double C[3], double w[3]) { /* from the origin */
    double d1[3], d2[3], v, ww, p, q, r, s, t, dist, dist2, w2[3];
    v3sub(d1, B, x0); v3sub(d2, C, x0);
    v = dot3(d1, d1); ww = dot3(d2, d2); p = dot3(x0, d1); q = dot3(x0, d2); r = dot3(d1, d2);
    s = (q * r - ww * p) / (ww * v - r * r);
    t = (-s * r - q) / ww;
    if ((ccd_zero(s) || s > 0) && (ccd_eq(s, 1) || s < 1) && (ccd_zero(t) || t > 0) && (ccd_eq(t, 1) || t < 1) && (ccd_eq(t + s, 1) || t + s < 1)) {
        for (int k = 0; k < 3; k++) w[k] = x0[k] + d1[k] * s + d2[k] * t;
        return dot3(w, w);
    }
    dist = point_seg_dist2(x0, B, w);
    dist2 = point_seg_dist2(x0, C, w2);
    if (dist2 < dist) { dist = dist2; memcpy(w, w2, 24); }
    dist2 = point_seg_dist2(B, C, w2);
    if (dist2 < dist) { dist = dist2; memcpy(w, w2, 24); }
    return dist;
}
static void find_pos(const CcdSup p[4], double pos[3]) {
    double dir[3], b[4], vec[3], sum, inv, p1[3] = {0, 0, 0}, p2[3] = {0, 0, 0};
    portal_dir(p, dir);
    cross3(vec, p[1].v, p[2].v); b[0] = dot3(vec, p[3].v);
    cross3(vec, p[3].v, p[2].v); b[1] = dot3(vec, p[0].v);
    cross3(vec, p[0].v, p[1].v); b[2] = dot3(vec, p[3].v);
    cross3(vec, p[2].v, p[1].v); b[3] = dot3(vec, p[0].v);
    sum = b[0] + b[1] + b[2] + b[3];
    if (ccd_zero(sum) || sum < 0) {
        b[0] = 0;
        cross3(vec, p[2].v, p[3].v); b[1] = dot3(vec, dir);
        cross3(vec, p[3].v, p[1].v); b[2] = dot3(vec, dir);
        cross3(vec, p[1].v, p[2].v); b[3] = dot3(vec, dir);
        sum = b[1] + b[2] + b[3];
    }
    inv = 1.0 / sum;
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 3; k++) { p1[k] += p[i].v1[k] * b[i]; p2[k] += p[i].v2[k] * b[i]; }
    for (int k = 0; k < 3; k++) pos[k] = (p1[k] * inv + p2[k] * inv) * 0.5;
}
/* 0: penetration found (depth, dir, pos);  -1: the (inflated) hulls are apart */
static int mpr_penetration(const CcdPair* P, double* depth, double dir[3], double pos[3]) {
    CcdSup p[4], v4;
    double va[3], vb[3], dt;
    int size, nsup = 0;
    /* ---- discoverPortal */
    geom_centre(P->m, P->d, P->g1, p[0].v1); geom_centre(P->m, P->d, P->g2, p[0].v2);
    v3sub(p[0].v, p[0].v1, p[0].v2);
    if (ccd_eq(p[0].v[0], 0) && ccd_eq(p[0].v[1], 0) && ccd_eq(p[0].v[2], 0)) p[0].v[0] += CCD_EPS * 10;
    for (int k = 0; k < 3; k++) dir[k] = -p[0].v[k];
    v3norm(dir);
    if (++nsup > MPR_MAXSUP) return -1;
    ccd_support(P, dir, &p[1]);
    dt = dot3(p[1].v, dir);
    if (ccd_zero(dt) || dt < 0) return -1;
    cross3(dir, p[0].v, p[1].v);
    if (ccd_zero(dot3(dir, dir))) {
        if (ccd_eq(p[1].v[0], 0) && ccd_eq(p[1].v[1], 0) && ccd_eq(p[1].v[2], 0)) { /* origin on v1: touching contact */
            *depth = 0; dir[0] = dir[1] = dir[2] = 0;
        } else { /* origin on the segment v0-v1 */
            *depth = sqrt(dot3(p[1].v, p[1].v));
            memcpy(dir, p[1].v, 24); v3norm(dir);
        }
        for (int k = 0; k < 3; k++) pos[k] = 0.5 * (p[1].v1[k] + p[1].v2[k]);
        return 0;
    }
    v3norm(dir);
    if (++nsup > MPR_MAXSUP) return -1;
    ccd_support(P, dir, &p[2]);
    dt = dot3(p[2].v, dir);
    if (ccd_zero(dt) || dt < 0) return -1;
    v3sub(va, p[1].v, p[0].v); v3sub(vb, p[2].v, p[0].v);
    cross3(dir, va, vb); v3norm(dir);
    if (dot3(dir, p[0].v) > 0) { CcdSup t = p[1]; p[1] = p[2]; p[2] = t; dir[0] = -dir[0]; dir[1] = -dir[1]; dir[2] = -dir[2]; }
    size = 3;
    while (size < 4) {
        int cont = 0;
        if (++nsup > MPR_MAXSUP) return -1;
        ccd_support(P, dir, &v4);
        dt = dot3(v4.v, dir);
        if (ccd_zero(dt) || dt < 0) return -1;
        cross3(va, p[1].v, v4.v);
        dt = dot3(va, p[0].v);
        if (dt < 0 && !ccd_zero(dt)) { p[2] = v4; cont = 1; }
        if (!cont) {
            cross3(va, v4.v, p[2].v);
            dt = dot3(va, p[0].v);
            if (dt < 0 && !ccd_zero(dt)) { p[1] = v4; cont = 1; }
        }
        if (cont) {
            v3sub(va, p[1].v, p[0].v); v3sub(vb, p[2].v, p[0].v);
            cross3(dir, va, vb); v3norm(dir);
        } else { p[3] = v4; size = 4; }
    }
    /* ---- refinePortal */
    for (;;) {
        portal_dir(p, dir);
        dt = dot3(dir, p[1].v);
        if (ccd_zero(dt) || dt > 0) break; /* the portal encapsulates the origin */
        if (++nsup > MPR_MAXSUP) return -1;
        ccd_support(P, dir, &v4);
        dt = dot3(v4.v, dir);
        if (!(ccd_zero(dt) || dt > 0) || portal_reach_tolerance(p, &v4, dir)) return -1;
        expand_portal(p, &v4);
    }
    /* ---- findPenetr */
    for (unsigned long it = 0;; it++) {
        portal_dir(p, dir);
        if (++nsup > MPR_MAXSUP) return -1;
        ccd_support(P, dir, &v4);
        if (portal_reach_tolerance(p, &v4, dir) || it > MPR_MAXIT) {
            double pd[3];
            *depth = sqrt(point_tri_dist2(p[1].v, p[2].v, p[3].v, pd));
            if (ccd_zero(pd[0]) && ccd_zero(pd[1]) && ccd_zero(pd[2])) memcpy(pd, dir, 24);
            v3norm(pd);
            memcpy(dir, pd, 24);
            find_pos(p, pos);
            return 0;
        }
        expand_portal(p, &v4);
    }
}
static void collide_mesh_mesh(const UhcModelDesc* m, OrcData* d, int g1, int g2, double margin, double gap) {
    double c1[3], c2[3], dc[3], depth, dir[3], pos[3];
    geom_centre(m, d, g1, c1); geom_centre(m, d, g2, c2);
    v3sub(dc, c1, c2);
    double bound = m->geom_rbound[g1] + m->geom_rbound[g2] + margin;
    if (dot3(dc, dc) > bound * bound) return; /* [MJ-ext] mj_collideGeoms bounding-sphere filter */
    CcdPair P = {m, d, g1, g2, margin, -1, -1};
    if (mpr_penetration(&P, &depth, dir, pos)) return;
    if (dir[0] == 0 && dir[1] == 0 && dir[2] == 0) return; /* normal undefined (mjc_MPRIteration) */
    add_contact(m, d, g1, g2, pos, dir, margin - depth, margin - gap);
}
void orc_collision(const UhcModelDesc* m, OrcData* d) {
    /* Contact order: plane pairs first, convex-convex pairs after them, each in (g1 < g2) order.  [MJ-ext] MuJoCo orders contacts by
     * body pair and then geom pair; with the floor as geom 0 of the world body (every model of the reference) that is the same order.
     * The device kernel uses this two-pass order, so both sides enumerate constraint rows identically. */
    d->ncon = 0;
    for (int pass = 0; pass < 2; pass++)
        for (int g1 = 0; g1 < m->ngeom; g1++)
            for (int g2 = g1 + 1; g2 < m->ngeom; g2++) {
                int b1 = m->geom_bodyid[g1], b2 = m->geom_bodyid[g2];
                if (!((m->geom_contype[g1] & m->geom_conaffinity[g2]) || (m->geom_contype[g2] & m->geom_conaffinity[g1]))) continue;
                if (bodies_filtered(m, b1, b2)) continue;
                double margin = fmax(m->geom_margin[g1], m->geom_margin[g2]);
                double gap = fmax(m->geom_gap[g1], m->geom_gap[g2]);
                int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
                if (pass == 0) {
                    if (t1 == UHC_GEOM_PLANE && is_hull(t2)) collide_plane_mesh(m, d, g1, g2, margin, gap);
                    else if (t2 == UHC_GEOM_PLANE && is_hull(t1)) collide_plane_mesh(m, d, g2, g1, margin, gap);
                } else if (is_hull(t1) && is_hull(t2)) collide_mesh_mesh(m, d, g1, g2, margin, gap);
            }
}

/* ------------------------------------------------------------------ P5: mj_makeConstraint [MJ-ext] */
static double impedance(const double solimp[5], double pos, double margin) { /* [MJ-ext] getimpedance */
    double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
    dmin = fmin(fmax(dmin, 0.0001), 0.9999); dmax = fmin(fmax(dmax, 0.0001), 0.9999);
    if (width < MINVAL) return 0.5 * (dmin + dmax);
    mid = fmin(fmax(mid, 0.0001), 0.9999);
    if (power < 1) power = 1;
    double x = fabs(pos - margin) / width, y;
    if (x >= 1) return dmax;
    if (x <= 0) return dmin;
    if (power == 1) y = x;
    else if (x <= mid) y = pow(x / mid, power) * mid;          /* a*x^p, a = 1/mid^(p-1) */
    else y = 1 - pow((1 - x) / (1 - mid), power) * (1 - mid);  /* 1 - b*(1-x)^p */
    return dmin + y * (dmax - dmin);
}
static int add_row(const UhcModelDesc* m, OrcData* d, int type, double pos, double margin, double diagApprox,
                   const double solref[2], const double solimp[5], double floss) {
    if (d->nefc >= ORC_MAXEFC) { d->efc_overflow = 1; return -1; }
    int r = d->nefc++;
    memset(d->efc_J + (size_t)r * m->nv, 0, m->nv * 8);
    d->efc_type[r] = type; d->efc_pos[r] = pos; d->efc_margin[r] = margin; d->efc_diagApprox[r] = diagApprox;
    d->efc_floss[r] = floss; d->efc_id[r] = -1; d->efc_edge[r] = 0;
    /* reference acceleration parameters, stored temporarily in aref (K) / b (B) / D (imp) */
    double timeconst = fmax(solref[0], 2 * m->timestep), dampratio = solref[1];
    double dmax = fmin(fmax(solimp[1], 0.0001), 0.9999);
    d->efc_aref[r] = 1.0 / (dmax * dmax * timeconst * timeconst * dampratio * dampratio); /* K */
    d->efc_b[r] = 2.0 / (dmax * timeconst);                                                 /* B */
    d->efc_D[r] = impedance(solimp, pos, margin);                                           /* imp */
    return r;
}
void orc_make_constraint(const UhcModelDesc* m, OrcData* d) {
    static const double dsolref[2] = {0.02, 1.0}, dsolimp[5] = {0.9, 0.95, 0.001, 0.5, 2.0};
    int nv = m->nv;
    d->nefc = 0;
    /* (1) dof friction loss */
    for (int i = 0; i < nv; i++)
        if (m->dof_frictionloss[i] > 0) {
            int r = add_row(m, d, ORC_EFC_FRICTION, 0, 0, m->dof_invweight0[i], dsolref, dsolimp, m->dof_frictionloss[i]);
            if (r >= 0) d->efc_J[(size_t)r * nv + i] = 1;
        }
    /* (2) joint limits */
    for (int j = 0; j < m->njnt; j++) {
        if (!m->jnt_limited[j]) continue;
        int t = m->jnt_type[j];
        if (t == UHC_JNT_BALL) {
            /* [MJ-ext] mj_instantiateLimit, ball joint: the joint's rotation as angle * axis (mju_quat2Vel with dt = 1: unit quaternion ->
             * axis = vector part / |vector part|, angle = 2 atan2(|vector part|, w) taken into (-pi, pi]); value = |angle|,
             * dist = max(range[0], range[1]) - value; ONE row whose Jacobian is -axis (the rotation's own direction) on the joint's three
             * dofs (the angular velocity of a ball joint lives in the child body's frame, which is the quaternion's) */
            double q[4], ax[3], margin = m->jnt_margin[j];
            memcpy(q, d->qpos + m->jnt_qposadr[j], 32);
            quat_normalize(q);
            double sn = sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            if (sn < MINVAL) { ax[0] = 1; ax[1] = ax[2] = 0; sn = 0; } else { ax[0] = q[1] / sn; ax[1] = q[2] / sn; ax[2] = q[3] / sn; }
            double ang = 2 * atan2(sn, q[0]);
            if (ang > M_PI) ang -= 2 * M_PI;
            double value = fabs(ang), sg = ang < 0 ? -1.0 : 1.0;
            if (value < MINVAL) { ax[0] = 1; ax[1] = ax[2] = 0; sg = 1; value = 0; }  /* mju_normalize3 of a zero vector: (1, 0, 0), length 0 */
            double dist = fmax(m->jnt_range[2 * j], m->jnt_range[2 * j + 1]) - value;
            if (dist < margin) {
                int r = add_row(m, d, ORC_EFC_LIMIT, dist, margin, m->dof_invweight0[m->jnt_dofadr[j]], dsolref, dsolimp, 0);
                if (r >= 0) for (int k = 0; k < 3; k++) d->efc_J[(size_t)r * nv + m->jnt_dofadr[j] + k] = -sg * ax[k];
            }
            continue;
        }
        if (t != UHC_JNT_HINGE && t != UHC_JNT_SLIDE) continue;
        double value = d->qpos[m->jnt_qposadr[j]], margin = m->jnt_margin[j];
        for (int side = -1; side <= 1; side += 2) {
            double dist = side * (m->jnt_range[2 * j + (side + 1) / 2] - value);
            if (dist < margin) {
                int r = add_row(m, d, ORC_EFC_LIMIT, dist, margin, m->dof_invweight0[m->jnt_dofadr[j]], dsolref, dsolimp, 0);
                if (r >= 0) d->efc_J[(size_t)r * nv + m->jnt_dofadr[j]] = -side;
            }
        }
    }
    /* (3) contacts, pyramidal friction cones */
    double* jp1 = d->work; double* jp2 = d->work + 3 * nv; double* jc = d->work + 6 * nv;
    for (int c = 0; c < d->ncon; c++) {
        if (d->con_dist[c] >= d->con_margin[c]) continue;
        int b1 = m->geom_bodyid[d->con_geom1[c]], b2 = m->geom_bodyid[d->con_geom2[c]], dim = d->con_dim[c];
        const double* f = d->con_frame + 9 * c;
        jac_point(m, d, b1, d->con_pos + 3 * c, jp1);
        jac_point(m, d, b2, d->con_pos + 3 * c, jp2);
        int ndir = dim >= 3 ? 3 : 1;
        for (int r = 0; r < ndir; r++)
            for (int i = 0; i < nv; i++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += f[3 * r + k] * (jp2[k * nv + i] - jp1[k * nv + i]);
                jc[r * nv + i] = s;
            }
        double tran = m->body_invweight0[2 * b1] + m->body_invweight0[2 * b2];
        if (dim == 1) {
            int r = add_row(m, d, ORC_EFC_CONTACT, d->con_dist[c], d->con_margin[c], tran, d->con_solref + 2 * c, d->con_solimp + 5 * c, 0);
            if (r >= 0) { memcpy(d->efc_J + (size_t)r * nv, jc, nv * 8); d->efc_id[r] = c; }
        } else {
            /* condim 3 only (torsional/rolling pyramid edges are not produced by these models) */
            for (int e = 0; e < 4; e++) {
                double mu = d->con_friction[3 * c] /* both tangents slide with friction[0] */, sgn = (e & 1) ? -1.0 : 1.0;
                int r = add_row(m, d, ORC_EFC_CONTACT_PYR, d->con_dist[c], d->con_margin[c], tran + mu * mu * tran,
                                d->con_solref + 2 * c, d->con_solimp + 5 * c, 0);
                if (r < 0) break;
                d->efc_id[r] = c; d->efc_edge[r] = e;
                for (int i = 0; i < nv; i++) d->efc_J[(size_t)r * nv + i] = jc[i] + sgn * mu * jc[(1 + e / 2) * nv + i];
            }
        }
    }
    /* impedance -> R, D, aref  ([MJ-ext] mj_makeImpedance) */
    for (int r = 0; r < d->nefc; r++) {
        double vel = 0, K = d->efc_aref[r], B = d->efc_b[r], imp = d->efc_D[r];
        for (int i = 0; i < nv; i++) vel += d->efc_J[(size_t)r * nv + i] * d->qvel[i];
        d->efc_vel[r] = vel;
        d->efc_R[r] = fmax(MINVAL, (1 - imp) * d->efc_diagApprox[r] / imp);
        d->efc_aref[r] = -B * vel - K * imp * (d->efc_pos[r] - d->efc_margin[r]);
    }
    /* pyramidal contacts: all edges share R = 2 mu^2 R(first edge) */
    for (int r = 0; r < d->nefc; r++)
        if (d->efc_type[r] == ORC_EFC_CONTACT_PYR && d->efc_edge[r] == 0) {
            double mu = d->con_friction[3 * d->efc_id[r]], Rpy = 2 * mu * mu * d->efc_R[r];
            for (int e = 0; e < 4 && r + e < d->nefc && d->efc_id[r + e] == d->efc_id[r]; e++) d->efc_R[r + e] = Rpy;
        }
    for (int r = 0; r < d->nefc; r++) d->efc_D[r] = 1.0 / d->efc_R[r];
}

/* ------------------------------------------------------------------ P7: mj_fwdVelocity [MJ-ext] */
void orc_com_vel(const UhcModelDesc* m, OrcData* d) {
    memset(d->cvel, 0, 48);
    for (int b = 1; b < m->nbody; b++) {
        double cvel[6], t[6];
        memcpy(cvel, d->cvel + 6 * m->body_parentid[b], 48);
        for (int j = m->body_jntadr[b]; j >= 0 && j < m->body_jntadr[b] + m->body_jntnum[b]; j++) {
            int da = m->jnt_dofadr[j];
            switch (m->jnt_type[j]) {
            case UHC_JNT_FREE:
                memset(d->cdof_dot + 6 * da, 0, 18 * 8);
                for (int k = 0; k < 3; k++)
                    for (int s = 0; s < 6; s++) cvel[s] += d->cdof[6 * (da + k) + s] * d->qvel[da + k];
                da += 3;
                /* fall through */
            case UHC_JNT_BALL:
                for (int k = 0; k < 3; k++) cross_motion(d->cdof_dot + 6 * (da + k), cvel, d->cdof + 6 * (da + k));
                for (int k = 0; k < 3; k++)
                    for (int s = 0; s < 6; s++) cvel[s] += d->cdof[6 * (da + k) + s] * d->qvel[da + k];
                break;
            default:
                cross_motion(t, cvel, d->cdof + 6 * da);
                memcpy(d->cdof_dot + 6 * da, t, 48);
                for (int s = 0; s < 6; s++) cvel[s] += d->cdof[6 * da + s] * d->qvel[da];
            }
        }
        memcpy(d->cvel + 6 * b, cvel, 48);
    }
}
void orc_passive(const UhcModelDesc* m, OrcData* d) {
    memset(d->qfrc_passive, 0, m->nv * 8);
    for (int j = 0; j < m->njnt; j++) {
        double k = m->jnt_stiffness[j];
        int t = m->jnt_type[j];
        if (k == 0 || (t != UHC_JNT_HINGE && t != UHC_JNT_SLIDE)) continue;
        d->qfrc_passive[m->jnt_dofadr[j]] -= k * (d->qpos[m->jnt_qposadr[j]] - m->qpos_spring[m->jnt_qposadr[j]]);
    }
    for (int i = 0; i < m->nv; i++) d->qfrc_passive[i] -= m->dof_damping[i] * d->qvel[i];
}
void orc_rne_bias(const UhcModelDesc* m, OrcData* d) { /* mj_rne(flg_acc=0) */
    int nb = m->nbody;
    double* cacc = d->work;          /* nb*6 */
    double* cfrc = d->work + 6 * nb; /* nb*6 */
    memset(cacc, 0, 48);
    for (int k = 0; k < 3; k++) cacc[3 + k] = -m->gravity[k];
    memset(cfrc, 0, 48);
    for (int b = 1; b < nb; b++) {
        double t[6], u[6];
        memcpy(cacc + 6 * b, cacc + 6 * m->body_parentid[b], 48);
        for (int i = m->body_dofadr[b]; i >= 0 && i < m->body_dofadr[b] + m->body_dofnum[b]; i++)
            for (int s = 0; s < 6; s++) cacc[6 * b + s] += d->cdof_dot[6 * i + s] * d->qvel[i];
        inert_mul(t, d->cinert + 10 * b, cacc + 6 * b);
        inert_mul(u, d->cinert + 10 * b, d->cvel + 6 * b);
        cross_force(cfrc + 6 * b, d->cvel + 6 * b, u);
        for (int s = 0; s < 6; s++) cfrc[6 * b + s] += t[s];
    }
    for (int b = nb - 1; b > 0; b--) {
        int p = m->body_parentid[b];
        if (p > 0) for (int s = 0; s < 6; s++) cfrc[6 * p + s] += cfrc[6 * b + s];
    }
    for (int i = 0; i < m->nv; i++) d->qfrc_bias[i] = dot6(d->cdof + 6 * i, cfrc + 6 * m->dof_bodyid[i]);
}

/* ------------------------------------------------------------------ P8: actuation + smooth acceleration */
void orc_fwd_acceleration(const UhcModelDesc* m, OrcData* d) {
    int nv = m->nv;
    memset(d->qfrc_actuator, 0, nv * 8);
    for (int a = 0; a < m->nu; a++) { /* [MJ-ext] joint transmission: gear[0] on a scalar joint, gear[0..2] on the three dofs of a ball joint */
        int da = m->actuator_dofid[a], nd = m->jnt_type[m->dof_jntid[da]] == UHC_JNT_BALL ? 3 : 1;
        for (int k = 0; k < nd; k++) d->qfrc_actuator[da + k] += m->actuator_gear[3 * a + k] * d->ctrl[a];
    }
    for (int i = 0; i < nv; i++) {
        d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_applied[i] + d->qfrc_actuator[i];
        d->qacc_smooth[i] = d->qfrc_smooth[i];
    }
    orc_solve_sparse(m, d->qLD, d->qacc_smooth);
}

/* ------------------------------------------------------------------ P6 + P9: dual problem and PGS [MJ-ext] */
#define ORC_AS_MAXROWS 256  /* rows up to which solver 1 pivots on the dual (the device's first three tiers); beyond: Newton on the primal */
void orc_project_constraint(const UhcModelDesc* m, OrcData* d) {
    int nv = m->nv, n = d->nefc;
    for (int r = 0; r < n; r++) {
        double a = 0;
        for (int i = 0; i < nv; i++) a += d->efc_J[(size_t)r * nv + i] * d->qacc_smooth[i];
        d->efc_b[r] = a - d->efc_aref[r];
    }
    /* the Delassus matrix, for the solvers that work on the dual (the primal Newton solver does not: it is what takes the big problems) */
    int fric = 0;
    for (int r = 0; r < n; r++) fric |= d->efc_type[r] == ORC_EFC_FRICTION;
    if (m->solver >= 1 && !fric && (m->solver == 2 || n > ORC_AS_MAXROWS)) return;
    if (n > d->ar_cap) {
        free(d->efc_AR);
        d->ar_cap = n + 64;
        d->efc_AR = malloc((size_t)d->ar_cap * d->ar_cap * 8);
    }
    double* MinvJt = d->work; /* n*nv: row r = M^-1 J_r^T */
    for (int r = 0; r < n; r++) {
        memcpy(MinvJt + (size_t)r * nv, d->efc_J + (size_t)r * nv, nv * 8);
        orc_solve_sparse(m, d->qLD, MinvJt + (size_t)r * nv);
    }
    for (int r = 0; r < n; r++)
        for (int s = 0; s < n; s++) {
            double a = 0;
            for (int i = 0; i < nv; i++) a += d->efc_J[(size_t)r * nv + i] * MinvJt[(size_t)s * nv + i];
            d->efc_AR[(size_t)r * n + s] = a;
        }
    for (int r = 0; r < n; r++) d->efc_AR[(size_t)r * n + r] += d->efc_R[r];
}
static double project_force(const OrcData* d, int r, double f) {
    switch (d->efc_type[r]) {
    case ORC_EFC_FRICTION: return fmin(fmax(f, -d->efc_floss[r]), d->efc_floss[r]);
    default: return f < 0 ? 0 : f; /* limits, frictionless and pyramidal contacts: f >= 0 */
    }
}
void orc_solve_pgs(const UhcModelDesc* m, OrcData* d) {
    int nv = m->nv, n = d->nefc;
    const double* AR = d->efc_AR;
    double* f = d->efc_force;
    d->solver_iter = 0;
    if (n == 0) {
        memset(d->qfrc_constraint, 0, nv * 8);
        memcpy(d->qacc, d->qacc_smooth, nv * 8);
        return;
    }
    /* warm start: forces implied by qacc_warmstart ([MJ-ext] warmstart + mj_constraintUpdate) */
    for (int r = 0; r < n; r++) {
        double jar = -d->efc_aref[r];
        for (int i = 0; i < nv; i++) jar += d->efc_J[(size_t)r * nv + i] * d->qacc_warmstart[i];
        if (d->efc_type[r] == ORC_EFC_FRICTION) f[r] = fmin(fmax(-d->efc_D[r] * jar, -d->efc_floss[r]), d->efc_floss[r]);
        else f[r] = jar < 0 ? -d->efc_D[r] * jar : 0;
    }
    double cost = 0;
    for (int r = 0; r < n; r++) {
        double a = 0;
        for (int s = 0; s < n; s++) a += AR[(size_t)r * n + s] * f[s];
        cost += f[r] * (0.5 * a + d->efc_b[r]);
    }
    if (cost > 0) memset(f, 0, n * 8);
    double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
    for (int it = 0; it < m->iterations; it++) {
        double improvement = 0;
        for (int r = 0; r < n; r++) {
            double res = d->efc_b[r], old = f[r];
            for (int s = 0; s < n; s++) res += AR[(size_t)r * n + s] * f[s];
            f[r] = project_force(d, r, old - res * (1.0 / AR[(size_t)r * n + r])); /* [MJ-ext] uses a precomputed ARinv */
            double delta = f[r] - old;
            double change = 0.5 * delta * delta * AR[(size_t)r * n + r] + delta * res;
            if (change > 1e-10) { f[r] = old; change = 0; }
            improvement -= change;
        }
        d->solver_iter = it + 1;
        if (improvement * scale < m->tolerance) break;
    }
    for (int i = 0; i < nv; i++) {
        double a = 0;
        for (int r = 0; r < n; r++) a += d->efc_J[(size_t)r * nv + i] * f[r];
        d->qfrc_constraint[i] = a;
        d->qacc[i] = a;
    }
    orc_solve_sparse(m, d->qLD, d->qacc);
    for (int i = 0; i < nv; i++) d->qacc[i] += d->qacc_smooth[i];
}

/* Exact solve of the same dual QP (UhcModelDesc.solver == 1; NOT a MuJoCo algorithm -- MuJoCo's exact solver is Newton on the
 * primal, which reaches the same optimum): block principal pivoting (Judice & Pires 1994; Kim & Park 2011 for NNLS).
 * F = rows allowed a positive force.  Solve A_FF f_F = -b_F, f = 0 elsewhere, y = A f + b; a row is infeasible when
 * f < 0 (in F) or y < 0 (outside F).  Flip all infeasible rows while their number keeps falling (3 grace rounds), otherwise
 * only the highest-index one (finite for symmetric positive definite A).  Start: F = the rows with a force after ORC_AS_PRESWEEPS sweeps.
 * The elimination works on the whole nefc x nefc matrix with the rows outside F replaced by identity rows, in the order
 * k = 0 .. nefc-1 and without pivoting -- the same order the device kernel uses. */
#define ORC_AS_MAXIT 64
#define ORC_AS_PRESWEEPS 16
static int solve_masked(int n, const double* A, const double* b, const unsigned char* F, double* W, double* f) {
    /* Gaussian elimination on [A_FF | -b_F], rows / columns outside F skipped; back substitution */
    double* c = W + (size_t)n * n;
    for (int i = 0; i < n; i++) { memcpy(W + (size_t)i * n, A + (size_t)i * n, n * 8); c[i] = -b[i]; }
    for (int k = 0; k < n; k++) {
        if (!F[k]) continue;
        double piv = W[(size_t)k * n + k];
        if (!(piv > MINVAL)) return 1;
        for (int i = k + 1; i < n; i++) {
            if (!F[i]) continue;
            double l = W[(size_t)i * n + k] / piv;
            for (int j = k + 1; j < n; j++) W[(size_t)i * n + j] -= l * W[(size_t)k * n + j];
            c[i] -= l * c[k];
        }
    }
    for (int k = n - 1; k >= 0; k--) {
        if (!F[k]) { f[k] = 0; continue; }
        double a = c[k];
        for (int j = k + 1; j < n; j++) if (F[j]) a -= W[(size_t)k * n + j] * f[j];
        f[k] = a / W[(size_t)k * n + k];
    }
    return 0;
}
int orc_solve_active_set(const UhcModelDesc* m, OrcData* d) {
    int nv = m->nv, n = d->nefc;
    const double* A = d->efc_AR;
    double* f = d->efc_force;
    unsigned char F[ORC_MAXEFC];
    double* W = (double*)malloc(((size_t)n * n + n) * 8);
    double* y = (double*)malloc((size_t)n * 8);
    /* initial guess: the rows ORC_AS_PRESWEEPS plain Gauss-Seidel sweeps from f = 0 leave with a force */
    memset(f, 0, n * 8);
    for (int s = 0; s < ORC_AS_PRESWEEPS; s++)
        for (int r = 0; r < n; r++) {
            double res = d->efc_b[r];
            for (int c = 0; c < n; c++) res += A[(size_t)r * n + c] * f[c];
            double fn = f[r] - res / A[(size_t)r * n + r];
            f[r] = fn < 0 ? 0 : fn;
        }
    for (int r = 0; r < n; r++) F[r] = f[r] > 0;
    int grace = 3, best = n + 1, it = 0, ok = 0;
    for (; it < ORC_AS_MAXIT; it++) {
        if (solve_masked(n, A, d->efc_b, F, W, f)) break;
        int nbad = 0, last = -1;
        for (int r = 0; r < n; r++) {
            double a = d->efc_b[r];
            for (int s = 0; s < n; s++) a += A[(size_t)r * n + s] * f[s];
            y[r] = a;
            int badr = F[r] ? f[r] < 0 : a < 0;
            if (badr) { nbad++; last = r; }
        }
        if (nbad == 0) { ok = 1; it++; break; }
        int all = 1;
        if (nbad < best) { best = nbad; grace = 3; }
        else if (grace > 0) grace--;
        else all = 0;
        for (int r = 0; r < n; r++) {
            int badr = F[r] ? f[r] < 0 : y[r] < 0;
            if (badr && (all || r == last)) F[r] = !F[r];
        }
    }
    free(W); free(y);
    d->solver_iter = it;
    if (!ok) return 1;
    for (int i = 0; i < nv; i++) {
        double a = 0;
        for (int r = 0; r < n; r++) a += d->efc_J[(size_t)r * nv + i] * f[r];
        d->qfrc_constraint[i] = a;
        d->qacc[i] = a;
    }
    orc_solve_sparse(m, d->qLD, d->qacc);
    for (int i = 0; i < nv; i++) d->qacc[i] += d->qacc_smooth[i];
    return 0;
}
/* Exact solve of the same problem in its PRIMAL form (UhcModelDesc.solver == 2, and whatever the active set above cannot finish or is too
 * big for): [MJ-ext] MuJoCo's default solver is Newton on
 *     min_a  1/2 (a - a_s)^T M (a - a_s) + sum_r s_r(J_r a - aref_r),      s_r(x) = 1/2 D_r x^2 for x < 0, else 0
 * (unilateral rows: limits, frictionless contacts, pyramid edges; D = 1/R) -- strictly convex, its minimiser is qacc_smooth + M^-1 J^T f*
 * with f* the optimum of the dual QP the two solvers above work on (f_r = -D_r min(0, J_r a - aref_r)).  The cost of an iteration does not
 * depend on how many rows carry a force (an nv x nv Hessian, H = M + sum_active D_r J_r^T J_r), which is why the device's last tier uses it for
 * environments with hundreds of rows (k_primal, uhc_amd/csrc/uhc_physics_impl.h).  Restated here in a-space with dense algebra; the device works
 * in u = D^1/2 L (a - a_s), where M becomes the identity -- Newton's method with an exact line search is invariant under that change of variables.
 *   start: a_s + M^-1 J^T f_ws (f_ws = the forces the warm-start acceleration implies, as the PGS above starts) if its cost is below cost(a_s), else a_s
 *   step:  active = {J_r a - aref_r < 0};  Cholesky of H;  dir = -H^-1 grad;  exact line search along dir (the derivative is piecewise linear and
 *          increasing: safeguarded Newton on it);  a += alpha dir
 *   stop:  a full step (alpha = 1 to 1e-12) that leaves the active set as it was -- the minimiser of that set's quadratic, KKT holds --, a gradient
 *          below 1e-14 of its first norm, or a Newton step below 1e-13 of the iterate (M-norm); ORC_PRIMAL_MAXIT iterations otherwise (returns 1:
 *          not converged, the result is still the best iterate). */
#define ORC_PRIMAL_MAXIT 100
#define ORC_PRIMAL_LS_MAXIT 60
static double primal_cost(int nv, int n, const double* Mx, const double* a, const double* as, const double* J, const double* D, const double* aref, double* jar) {
    double c = 0;
    for (int i = 0; i < nv; i++) {
        double mi = 0;
        for (int j = 0; j < nv; j++) mi += Mx[(size_t)i * nv + j] * (a[j] - as[j]);
        c += 0.5 * (a[i] - as[i]) * mi;
    }
    for (int r = 0; r < n; r++) {
        double x = -aref[r];
        for (int i = 0; i < nv; i++) x += J[(size_t)r * nv + i] * a[i];
        jar[r] = x;
        if (x < 0) c += 0.5 * D[r] * x * x;
    }
    return c;
}
int orc_solve_primal(const UhcModelDesc* m, OrcData* d) {
    int nv = m->nv, n = d->nefc;
    const double *J = d->efc_J, *D = d->efc_D, *aref = d->efc_aref, *as = d->qacc_smooth;
    double* Mx = (double*)malloc((size_t)nv * nv * 8), *H = (double*)malloc((size_t)nv * nv * 8);
    double* a = (double*)malloc((size_t)nv * 8), *g = (double*)malloc((size_t)nv * 8), *dir = (double*)malloc((size_t)nv * 8), *Md = (double*)malloc((size_t)nv * 8);
    double* jar = (double*)malloc((size_t)(n + 1) * 8), *p = (double*)malloc((size_t)(n + 1) * 8);
    unsigned char* act = (unsigned char*)malloc(n + 1);
    orc_full_m(m, d->qM, Mx);
    /* start point */
    for (int i = 0; i < nv; i++) a[i] = 0;
    for (int r = 0; r < n; r++) {
        double x = -aref[r];
        for (int i = 0; i < nv; i++) x += J[(size_t)r * nv + i] * d->qacc_warmstart[i];
        const double f = x < 0 ? -D[r] * x : 0;
        for (int i = 0; i < nv; i++) a[i] += J[(size_t)r * nv + i] * f;
    }
    orc_solve_sparse(m, d->qLD, a);
    for (int i = 0; i < nv; i++) a[i] += as[i];
    if (!(primal_cost(nv, n, Mx, a, as, J, D, aref, jar) < primal_cost(nv, n, Mx, as, as, J, D, aref, p))) memcpy(a, as, nv * 8);
    int it = 0, ok = 0;
    double g0 = -1;
    for (; it < ORC_PRIMAL_MAXIT; it++) {
        for (int r = 0; r < n; r++) {
            double x = -aref[r];
            for (int i = 0; i < nv; i++) x += J[(size_t)r * nv + i] * a[i];
            jar[r] = x; act[r] = x < 0;
        }
        memcpy(H, Mx, (size_t)nv * nv * 8);
        double gn = 0;
        for (int i = 0; i < nv; i++) {
            double mi = 0;
            for (int j = 0; j < nv; j++) mi += Mx[(size_t)i * nv + j] * (a[j] - as[j]);
            g[i] = mi;
        }
        for (int r = 0; r < n; r++) {
            if (!act[r]) continue;
            const double* Jr = J + (size_t)r * nv;
            for (int i = 0; i < nv; i++) {
                if (Jr[i] == 0) continue;
                g[i] += D[r] * jar[r] * Jr[i];
                for (int j = 0; j < nv; j++) H[(size_t)i * nv + j] += D[r] * Jr[i] * Jr[j];
            }
        }
        for (int i = 0; i < nv; i++) gn += g[i] * g[i];
        gn = sqrt(gn);
        if (g0 < 0) g0 = gn;
        if (gn <= 1e-14 * g0 || gn == 0) { ok = 1; break; }
        /* Cholesky H = C C^T (lower, in place), dir = -H^-1 g */
        int bad_h = 0;
        for (int k = 0; k < nv && !bad_h; k++) {
            double s = H[(size_t)k * nv + k];
            for (int q = 0; q < k; q++) s -= H[(size_t)k * nv + q] * H[(size_t)k * nv + q];
            if (!(s > 0)) { bad_h = 1; break; }
            const double c = sqrt(s);
            H[(size_t)k * nv + k] = c;
            for (int i = k + 1; i < nv; i++) {
                double t = H[(size_t)i * nv + k];
                for (int q = 0; q < k; q++) t -= H[(size_t)i * nv + q] * H[(size_t)k * nv + q];
                H[(size_t)i * nv + k] = t / c;
            }
        }
        if (bad_h) break;
        for (int i = 0; i < nv; i++) {
            double t = -g[i];
            for (int q = 0; q < i; q++) t -= H[(size_t)i * nv + q] * dir[q];
            dir[i] = t / H[(size_t)i * nv + i];
        }
        for (int i = nv - 1; i >= 0; i--) {
            double t = dir[i];
            for (int q = i + 1; q < nv; q++) t -= H[(size_t)q * nv + i] * dir[q];
            dir[i] = t / H[(size_t)i * nv + i];
        }
        /* exact line search: phi'(alpha) = (a - a_s + alpha dir)^T M dir + sum_r D_r min(0, jar_r + alpha p_r) p_r */
        double lin0 = 0, quad = 0;
        for (int i = 0; i < nv; i++) {
            double mi = 0;
            for (int j = 0; j < nv; j++) mi += Mx[(size_t)i * nv + j] * dir[j];
            Md[i] = mi;
            lin0 += (a[i] - as[i]) * mi;
            quad += dir[i] * mi;
        }
        for (int r = 0; r < n; r++) {
            double x = 0;
            for (int i = 0; i < nv; i++) x += J[(size_t)r * nv + i] * dir[i];
            p[r] = x;
        }
        {   /* a step below the rounding of the iterate (both measured in the M-norm, which is what the device's u-coordinates measure): converged */
            double un = 0;
            for (int i = 0; i < nv; i++) {
                double mi = 0;
                for (int j = 0; j < nv; j++) mi += Mx[(size_t)i * nv + j] * (a[j] - as[j]);
                un += (a[i] - as[i]) * mi;
            }
            if (quad <= 1e-26 * (1.0 + un)) { ok = 1; it++; break; }
        }
        double alpha = 1.0, lo = 0.0, hi = -1.0;  /* phi'(lo) < 0; hi < 0: no upper bracket yet */
        for (int ls = 0; ls < ORC_PRIMAL_LS_MAXIT; ls++) {
            double d1 = lin0 + alpha * quad, d2 = quad;
            for (int r = 0; r < n; r++) {
                const double x = jar[r] + alpha * p[r];
                if (x < 0) { d1 += D[r] * x * p[r]; d2 += D[r] * p[r] * p[r]; }
            }
            if (fabs(d1) <= 1e-15 * (fabs(lin0) + 1e-300)) break;
            if (d1 < 0) lo = alpha; else hi = alpha;
            double nx = alpha - d1 / d2;
            if (!(nx > lo) || (hi > 0 && !(nx < hi))) nx = hi > 0 ? 0.5 * (lo + hi) : 2 * alpha;
            if (nx == alpha) break;
            alpha = nx;
        }
        for (int i = 0; i < nv; i++) a[i] += alpha * dir[i];
        int same = fabs(alpha - 1.0) <= 1e-12;
        for (int r = 0; r < n && same; r++) same = ((jar[r] + alpha * p[r]) < 0) == act[r];
        if (same) { ok = 1; it++; break; }
    }
    /* forces, constraint force, acceleration */
    for (int i = 0; i < nv; i++) d->qfrc_constraint[i] = 0;
    for (int r = 0; r < n; r++) {
        double x = -aref[r];
        for (int i = 0; i < nv; i++) x += J[(size_t)r * nv + i] * a[i];
        const double f = x < 0 ? -D[r] * x : 0;
        d->efc_force[r] = f;
        if (f != 0) for (int i = 0; i < nv; i++) d->qfrc_constraint[i] += J[(size_t)r * nv + i] * f;
    }
    memcpy(d->qacc, d->qfrc_constraint, nv * 8);
    orc_solve_sparse(m, d->qLD, d->qacc);
    for (int i = 0; i < nv; i++) d->qacc[i] += as[i];
    d->solver_iter = it;
    free(Mx); free(H); free(a); free(g); free(dir); free(Md); free(jar); free(p); free(act);
    return !ok;
}
/* solver 0: PGS sweeps.  solver 1 (the package default): the exact optimum -- block pivoting on the dual while the problem is small enough for
 * it (ORC_AS_MAXROWS rows: what the device's first three tiers hold), Newton on the primal beyond that and wherever the pivoting does not
 * finish: the same optimum either way.  solver 2: the primal solver alone (tests).  Friction-loss rows (box constraints) go to the sweeps. */
static void orc_solve_constraints(const UhcModelDesc* m, OrcData* d) {
    int fric = 0;
    for (int r = 0; r < d->nefc; r++) fric |= d->efc_type[r] == ORC_EFC_FRICTION;
    if (m->solver >= 1 && d->nefc > 0 && !fric) {
        if (m->solver == 1 && d->nefc <= ORC_AS_MAXROWS && !orc_solve_active_set(m, d)) return;
        d->primal_solves++;
        if (!orc_solve_primal(m, d)) return;
        d->primal_unconverged++;
        return;  /* (the best iterate of a Newton run that hit its cap: an env with |b| of 1e16 on its way to the bad-value flag) */
    }
    orc_solve_pgs(m, d);  /* solver 0, or friction-loss rows */
}

/* ------------------------------------------------------------------ mj_forward / mj_step [MJ-ext] */
void orc_forward(const UhcModelDesc* m, OrcData* d) {
    orc_kinematics(m, d);
    orc_com_pos(m, d);
    orc_crb(m, d);
    orc_factor_m(m, d);
    orc_collision(m, d);
    orc_make_constraint(m, d);
    if (d->ncon > d->max_ncon) d->max_ncon = d->ncon;
    if (d->nefc > d->max_nefc) d->max_nefc = d->nefc;
    orc_com_vel(m, d);
    orc_passive(m, d);
    orc_rne_bias(m, d);
    orc_fwd_acceleration(m, d);
    orc_project_constraint(m, d);
    orc_solve_constraints(m, d);
}
static int bad(double x) { return isnan(x) || x > MAXVAL || x < -MAXVAL; }
void orc_euler(const UhcModelDesc* m, OrcData* d) {
    /* P10: [MJ-ext] mj_Euler.  Without joint damping the velocity update uses qacc as is.  With any dof_damping > 0 the damping
     * force is integrated implicitly: (M + h diag(B)) a = qfrc_smooth + qfrc_constraint (qfrc_smooth already holds the explicit
     * -B v of mj_passive), v += h a.  d->qacc itself -- what the next step's warm start and the acceleration check see -- stays
     * the explicit one.  The reference reaches this branch whenever a config sets joint damping (humanoid_im.py:221-224,
     * config/copycat_ball/copycat_ball_1.yml:104-113). */
    double h = m->timestep;
    int damped = 0;
    for (int i = 0; i < m->nv; i++) damped |= m->dof_damping[i] > 0;
    if (!damped) {
        for (int i = 0; i < m->nv; i++) d->qvel[i] += h * d->qacc[i];
    } else {
        double* MhB = (double*)malloc((size_t)d->nM * 8);
        double* a = (double*)malloc((size_t)m->nv * 8);
        memcpy(MhB, d->qM, (size_t)d->nM * 8);
        for (int i = 0; i < m->nv; i++) {
            MhB[m->dof_madr[i]] += h * m->dof_damping[i];
            a[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
        }
        orc_factor_sparse(m, MhB);
        orc_solve_sparse(m, MhB, a);
        for (int i = 0; i < m->nv; i++) d->qvel[i] += h * a[i];
        free(MhB); free(a);
    }
    for (int j = 0; j < m->njnt; j++) {
        int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
        switch (m->jnt_type[j]) {
        case UHC_JNT_FREE:
            for (int k = 0; k < 3; k++) d->qpos[qa + k] += h * d->qvel[da + k];
            qa += 3; da += 3;
                /* fall through */
        case UHC_JNT_BALL: { /* [MJ-ext] mju_quatIntegrate: q <- q * exp(h w / 2), w in body frame */
            double w[3] = {d->qvel[da], d->qvel[da + 1], d->qvel[da + 2]}, qr[4];
            double n = sqrt(dot3(w, w));
            if (n < MINVAL) { w[0] = 1; w[1] = w[2] = 0; } else { w[0] /= n; w[1] /= n; w[2] /= n; }
            axis_angle_quat(qr, w, h * n);
            quat_mul(d->qpos + qa, d->qpos + qa, qr);
            quat_normalize(d->qpos + qa);
            break; }
        default:
            d->qpos[qa] += h * d->qvel[da];
        }
    }
    memcpy(d->qacc_warmstart, d->qacc, m->nv * 8);
}
void orc_step(const UhcModelDesc* m, OrcData* d) {
    if (d->fail) return;
    for (int i = 0; i < m->nq; i++) if (bad(d->qpos[i])) d->fail = 1;  /* mj_checkPos */
    for (int i = 0; i < m->nv; i++) if (bad(d->qvel[i])) d->fail = 1;  /* mj_checkVel */
    if (d->fail) return;
    orc_forward(m, d);
    for (int i = 0; i < m->nv; i++) if (bad(d->qacc[i])) d->fail = 1;  /* mj_checkAcc */
    if (d->fail) return; /* mujoco-py raises -> HumanoidEnv.step sets fail (humanoid_im.py:1207-1211) */
    orc_euler(m, d);
}
void orc_set_state(const UhcModelDesc* m, OrcData* d, const double* qpos, const double* qvel) {
    /* sim.reset() + set_state + sim.forward(): mujoco_env.py:95-113 */
    memcpy(d->qpos, qpos, m->nq * 8);
    memcpy(d->qvel, qvel, m->nv * 8);
    memset(d->qacc_warmstart, 0, m->nv * 8);
    memset(d->qfrc_applied, 0, m->nv * 8);
    memset(d->ctrl, 0, (m->nu > 0 ? m->nu : 1) * 8);
    d->fail = 0; d->efc_overflow = 0; d->max_ncon = 0; d->max_nefc = 0; d->primal_solves = 0; d->primal_unconverged = 0;
    orc_forward(m, d);
}

/* ------------------------------------------------------------------ controller (pinned by the reference) */
/* get_heading_q: uhc/utils/math_utils.py:134-139 */
static void heading_q(double hq[4], const double q[4]) {
    hq[0] = q[0]; hq[1] = 0; hq[2] = 0; hq[3] = q[3];
    double n = sqrt(hq[0] * hq[0] + hq[3] * hq[3]);
    hq[0] /= n; hq[3] /= n;
}
/* quat_mul_vec (uhc/utils/transformation.py quat_mul_vec): v' = R(q) v */
static void quat_rot(double r[3], const double q[4], const double v[3]) {
    double R[9];
    quat_to_mat(R, q);
    mat_vec(r, R, v);
}
/* compute_torque + clip (humanoid_im.py:1033-1076, :1161) for substep `it`: writes d->ctrl */
void orc_pd_torque(const UhcModelDesc* m, const UhcCtrlDesc* c, OrcData* d, const double* action,
                   const double* target_base, int it) {
    int nv = m->nv, nu = m->nu;
    double dt = m->timestep;
    int vf_dim = c->rfc_mode == 1 ? 6 : c->rfc_mode == 2 ? c->n_vf_body * c->body_vf_dim : 0; /* humanoid_im.py:233-243 */
    double* kp = calloc(nv, 8); double* kd = calloc(nv, 8); double* qerr = calloc(nv, 8);
    double* rhs = calloc(nv, 8); double* MK = malloc(d->nM * 8);
    double skp = 1, skd = 1;
    if (c->meta_pd == 1) { /* :1054-1057 */
        skp = fmin(fmax(action[nu + vf_dim + it] + 1, 0), 10);
        skd = fmin(fmax(action[nu + vf_dim + it + c->n_substeps] + 1, 0), 10);
    }
    for (int a = 0; a < nu; a++) {
        double cur = d->qpos[7 + a], base = target_base[a];
        while (base - cur > M_PI) base -= 2 * M_PI;   /* :1042-1045 */
        while (base - cur < -M_PI) base += 2 * M_PI;
        double target = base + action[a];
        double gkp = c->jkp[a], gkd = c->jkd[a];
        if (c->meta_pd == 1) { gkp *= skp; gkd *= skd; }
        else if (c->meta_pd == 2) {
            gkp *= fmin(fmax(action[nu + vf_dim + a] + 1, 0), 10);
            gkd *= fmin(fmax(action[nu + vf_dim + nu + a] + 1, 0), 10);
        }
        kp[6 + a] = gkp; kd[6 + a] = gkd;
        qerr[6 + a] = cur + d->qvel[6 + a] * dt - target;  /* :1071 */
    }
    /* compute_desired_accel: (M + Kd dt) qdd = -C - Kp e_q - Kd e_v   (:1014-1031) */
    memcpy(MK, d->qM, d->nM * 8);
    for (int i = 0; i < nv; i++) {
        MK[m->dof_madr[i]] += kd[i] * dt;
        rhs[i] = -d->qfrc_bias[i] - kp[i] * qerr[i] - kd[i] * d->qvel[i];
    }
    orc_factor_sparse(m, MK);
    orc_solve_sparse(m, MK, rhs);
    for (int a = 0; a < nu; a++) {
        double tq = -kp[6 + a] * qerr[6 + a] - kd[6 + a] * (d->qvel[6 + a] + rhs[6 + a] * dt); /* :1073-1075 */
        d->ctrl[a] = fmin(fmax(tq, -c->torque_lim[a]), c->torque_lim[a]);                       /* :1161 */
    }
    free(kp); free(kd); free(qerr); free(rhs); free(MK);
}
/* rfc_implicit: humanoid_im.py:1136-1143 -> d->qfrc_applied[0:6] */
void orc_rfc_implicit(const UhcModelDesc* m, const UhcCtrlDesc* c, OrcData* d, const double* action) {
    double vf[6], q[4], binv[4] = {c->base_rot[0], -c->base_rot[1], -c->base_rot[2], -c->base_rot[3]}, hq[4], r[3];
    double bn = c->base_rot[0] * c->base_rot[0] + c->base_rot[1] * c->base_rot[1] + c->base_rot[2] * c->base_rot[2] + c->base_rot[3] * c->base_rot[3];
    for (int k = 0; k < 4; k++) binv[k] /= bn; /* quaternion_inverse divides by |q|^2 (transformation.py:1509-1520) */
    for (int k = 0; k < 6; k++) vf[k] = action[m->nu + k] * c->rfc_scale;
    quat_mul(q, d->qpos + 3, binv);
    heading_q(hq, q);
    quat_rot(r, hq, vf);
    memcpy(vf, r, 24);
    for (int k = 0; k < 6; k++) d->qfrc_applied[k] = fmin(fmax(vf[k], -c->rfc_lim), c->rfc_lim);
}
/* rfc_explicit: humanoid_im.py:1080-1132 with the default switches (residual_contact_only = False, no projection,
 * one wrench per body): per body a contact point, force and torque in the BODY frame -> world (mujoco_env.py:171-180),
 * accumulated into qfrc_applied by mj_applyFT [MJ-ext]: qfrc += jacp' f + jacr' tau with the Jacobian at the world
 * point, built from the cdof / subtree_com left by the previous forward pass. */
void orc_rfc_explicit(const UhcModelDesc* m, const UhcCtrlDesc* c, OrcData* d, const double* action) {
    const int nv = m->nv, bvd = c->body_vf_dim;
    double* qfrc = calloc(nv, 8);
    for (int i = 0; i < c->n_vf_body; i++) {
        const int body = c->vf_body[i];
        const double* vf = action + m->nu + i * bvd;
        const double* R = d->xmat + 9 * body;
        double cp[3], f[3], tq[3] = {0, 0, 0}, fl[3], tl[3] = {0, 0, 0};
        for (int k = 0; k < 3; k++) { fl[k] = vf[3 + k] * c->rfc_scale; if (bvd >= 9) tl[k] = vf[6 + k] * c->rfc_scale; }
        for (int r = 0; r < 3; r++) {
            cp[r] = R[3 * r] * vf[0] + R[3 * r + 1] * vf[1] + R[3 * r + 2] * vf[2] + d->xpos[3 * body + r];
            f[r] = R[3 * r] * fl[0] + R[3 * r + 1] * fl[1] + R[3 * r + 2] * fl[2];
            tq[r] = R[3 * r] * tl[0] + R[3 * r + 1] * tl[1] + R[3 * r + 2] * tl[2];
        }
        if (body <= 0) continue;
        const double* c0 = d->subtree_com + 3 * body_rootid(m, body);
        const double off[3] = {cp[0] - c0[0], cp[1] - c0[1], cp[2] - c0[2]};
        int b = body;
        while (b > 0 && m->body_dofnum[b] == 0) b = m->body_parentid[b];
        if (b <= 0) continue;
        for (int j = m->body_dofadr[b] + m->body_dofnum[b] - 1; j >= 0; j = m->dof_parentid[j]) {
            double t[3];
            cross3(t, d->cdof + 6 * j, off);
            const double* cd = d->cdof + 6 * j;
            qfrc[j] += (cd[3] + t[0]) * f[0] + (cd[4] + t[1]) * f[1] + (cd[5] + t[2]) * f[2] + cd[0] * tq[0] + cd[1] * tq[1] + cd[2] * tq[2];
        }
    }
    memcpy(d->qfrc_applied, qfrc, (size_t)nv * 8);
    free(qfrc);
}
/*
 * One HumanoidEnv.do_simulation (humanoid_im.py:1145-1190).  action layout: humanoid_im.py:226-255.
 * M and C used by the PD solve are the ones left in `d` by the previous forward pass
 * (humanoid_im.py:1019-1022 reads data.qM / data.qfrc_bias before sim.step()).
 */
void orc_do_simulation_mixed(const UhcModelDesc* m0, const UhcCtrlDesc* c, OrcData* d, const double* action,
                             const double* target_base, unsigned sweep_mask) {
    /* sweep_mask bit k: substep k is solved by the sweeps (solver 0) whatever the model says.  The device path reports the substeps in
     * which its exact solve gave up and swept (UHC_F_REDO bits 8+); the checker follows it substep by substep. */
    UhcModelDesc mm = *m0;
    const UhcModelDesc* m = &mm;
    for (int it = 0; it < c->n_substeps && !d->fail; it++) {
        mm.solver = (it < 32 && ((sweep_mask >> it) & 1u)) ? 0 : m0->solver;
        if (c->action_type == 0) orc_pd_torque(m, c, d, action, target_base, it);
        else
            for (int a = 0; a < m->nu; a++) /* :1160-1161 */
                d->ctrl[a] = fmin(fmax(action[a] * c->a_scale[a] * 100, -c->torque_lim[a]), c->torque_lim[a]);
        if (c->rfc_mode == 1) orc_rfc_implicit(m, c, d, action);
        else if (c->rfc_mode == 2) orc_rfc_explicit(m, c, d, action);
        orc_step(m, d);
    }
}
void orc_do_simulation(const UhcModelDesc* m, const UhcCtrlDesc* c, OrcData* d, const double* action,
                       const double* target_base) {
    orc_do_simulation_mixed(m, c, d, action, target_base, 0u);
}

/* ------------------------------------------------------------------ field access for tests */
int orc_get(const UhcModelDesc* m, const OrcData* d, const char* name, double* out, int max) {
    struct { const char* n; const double* p; int cnt; } tab[] = {
        {"qpos", d->qpos, m->nq}, {"qvel", d->qvel, m->nv}, {"qacc", d->qacc, m->nv},
        {"qacc_smooth", d->qacc_smooth, m->nv}, {"qacc_warmstart", d->qacc_warmstart, m->nv},
        {"xpos", d->xpos, 3 * m->nbody}, {"xquat", d->xquat, 4 * m->nbody}, {"xipos", d->xipos, 3 * m->nbody},
        {"xmat", d->xmat, 9 * m->nbody}, {"ximat", d->ximat, 9 * m->nbody},
        {"subtree_com", d->subtree_com, 3 * m->nbody}, {"cinert", d->cinert, 10 * m->nbody},
        {"cdof", d->cdof, 6 * m->nv}, {"cvel", d->cvel, 6 * m->nbody},
        {"qM", d->qM, d->nM}, {"qLD", d->qLD, d->nM}, {"qfrc_bias", d->qfrc_bias, m->nv},
        {"qfrc_smooth", d->qfrc_smooth, m->nv}, {"qfrc_constraint", d->qfrc_constraint, m->nv},
        {"qfrc_applied", d->qfrc_applied, m->nv}, {"qfrc_actuator", d->qfrc_actuator, m->nv},
        {"ctrl", d->ctrl, m->nu},
        {"con_pos", d->con_pos, 3 * d->ncon}, {"con_dist", d->con_dist, d->ncon}, {"con_frame", d->con_frame, 9 * d->ncon},
        {"efc_J", d->efc_J, d->nefc * m->nv}, {"efc_R", d->efc_R, d->nefc}, {"efc_aref", d->efc_aref, d->nefc},
        {"efc_b", d->efc_b, d->nefc}, {"efc_force", d->efc_force, d->nefc}, {"efc_pos", d->efc_pos, d->nefc},
        {"efc_AR", d->efc_AR, d->nefc * d->nefc}, {"efc_D", d->efc_D, d->nefc},
    };
    for (size_t i = 0; i < sizeof tab / sizeof tab[0]; i++)
        if (!strcmp(tab[i].n, name)) {
            int n = tab[i].cnt < max ? tab[i].cnt : max;
            memcpy(out, tab[i].p, (size_t)n * 8);
            return tab[i].cnt;
        }
    if (!strcmp(name, "con_geom1") || !strcmp(name, "con_geom2")) { /* the contacts' geom ids, as doubles */
        const int* g = name[8] == '1' ? d->con_geom1 : d->con_geom2;
        for (int c = 0; c < d->ncon && c < max; c++) out[c] = g[c];
        return d->ncon;
    }
    return -1;
}
int orc_get_int(const OrcData* d, const char* name) {
    if (!strcmp(name, "ncon")) return d->ncon;
    if (!strcmp(name, "nefc")) return d->nefc;
    if (!strcmp(name, "fail")) return d->fail;
    if (!strcmp(name, "solver_iter")) return d->solver_iter;
    if (!strcmp(name, "efc_overflow")) return d->efc_overflow;
    if (!strcmp(name, "nM")) return d->nM;
    if (!strcmp(name, "max_ncon")) return d->max_ncon;
    if (!strcmp(name, "max_nefc")) return d->max_nefc;
    if (!strcmp(name, "primal_solves")) return d->primal_solves;
    if (!strcmp(name, "primal_unconverged")) return d->primal_unconverged;
    return -1;
}
void orc_set(const UhcModelDesc* m, OrcData* d, const char* name, const double* in) {
    if (!strcmp(name, "qpos")) memcpy(d->qpos, in, m->nq * 8);
    else if (!strcmp(name, "qvel")) memcpy(d->qvel, in, m->nv * 8);
    else if (!strcmp(name, "ctrl")) memcpy(d->ctrl, in, m->nu * 8);
    else if (!strcmp(name, "qfrc_applied")) memcpy(d->qfrc_applied, in, m->nv * 8);
    else if (!strcmp(name, "qacc_warmstart")) memcpy(d->qacc_warmstart, in, m->nv * 8);
    else if (!strcmp(name, "qM")) memcpy(d->qM, in, d->nM * 8);
    else if (!strcmp(name, "qfrc_bias")) memcpy(d->qfrc_bias, in, m->nv * 8);
}

/* ------------------------------------------------------------------ batch driver for the CPU baseline */
void orc_set_threads(int n) { /* OpenMP threads of the batch driver (the CPU baseline is reported at 1 thread and at all host cores) */
#ifdef _OPENMP
    extern void omp_set_num_threads(int);
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}
void orc_batch_do_simulation(const UhcModelDesc* m, const UhcCtrlDesc* c, OrcData** ds, int n_env,
                             const double* actions, const double* target_base) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int e = 0; e < n_env; e++)
        orc_do_simulation(m, c, ds[e], actions + (size_t)e * c->action_dim, target_base + (size_t)e * m->nu);
}
