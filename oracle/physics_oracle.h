/* physics_oracle.h -- TEST INFRASTRUCTURE (see physics_oracle.c). */
#ifndef PHYSICS_ORACLE_H
#define PHYSICS_ORACLE_H
#include "../include/uhc_amd.h"

/* capacities: above what the device path's last tier holds (KernelArgs::cx: up to 1024 rows / 256 contacts), towards the reference's njmax 2500 / nconmax 500
 * (uhc/khrylib/mocap/skeleton_mesh.py:46) */
#define ORC_MAXCON 512
#define ORC_MAXEFC 1280

enum { ORC_EFC_FRICTION = 1, ORC_EFC_LIMIT = 2, ORC_EFC_CONTACT = 3, ORC_EFC_CONTACT_PYR = 4 };

typedef struct OrcData {
    int nM, ncon, nefc, fail, solver_iter, efc_overflow, max_ncon, max_nefc; /* max_*: running maxima since set_state */
    int primal_solves, primal_unconverged; /* forward passes solved by Newton on the primal since set_state, and how many of those hit the iteration cap */
    int ar_cap; /* rows the Delassus matrix efc_AR is allocated for (grown on demand: the primal solver does not need it) */
    double *qpos, *qvel, *qacc, *qacc_warmstart, *ctrl, *qfrc_applied;
    double *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis;
    double *subtree_com, *cinert, *crb, *cdof, *cdof_dot, *cvel;
    double *qM, *qLD;
    double *qfrc_bias, *qfrc_passive, *qfrc_actuator, *qfrc_smooth, *qacc_smooth, *qfrc_constraint;
    double *con_pos, *con_frame, *con_dist, *con_margin, *con_friction, *con_solref, *con_solimp;
    int *con_geom1, *con_geom2, *con_dim;
    double *efc_J, *efc_pos, *efc_margin, *efc_R, *efc_D, *efc_aref, *efc_b, *efc_force, *efc_vel;
    double *efc_diagApprox, *efc_floss, *efc_AR;
    int *efc_type, *efc_id; /* efc_id: contact index for contact rows, -1 otherwise; pyramid first row flagged by efc_edge==0 */
    int *efc_edge;
    double *work;
} OrcData;

OrcData* orc_data_create(const UhcModelDesc* m);
void orc_data_free(OrcData* d);
void orc_kinematics(const UhcModelDesc* m, OrcData* d);
void orc_com_pos(const UhcModelDesc* m, OrcData* d);
void orc_crb(const UhcModelDesc* m, OrcData* d);
void orc_factor_sparse(const UhcModelDesc* m, double* LD);
void orc_solve_sparse(const UhcModelDesc* m, const double* LD, double* x);
void orc_factor_m(const UhcModelDesc* m, OrcData* d);
void orc_full_m(const UhcModelDesc* m, const double* qM, double* dense);
void orc_collision(const UhcModelDesc* m, OrcData* d);
void orc_make_constraint(const UhcModelDesc* m, OrcData* d);
void orc_com_vel(const UhcModelDesc* m, OrcData* d);
void orc_passive(const UhcModelDesc* m, OrcData* d);
void orc_rne_bias(const UhcModelDesc* m, OrcData* d);
void orc_fwd_acceleration(const UhcModelDesc* m, OrcData* d);
void orc_project_constraint(const UhcModelDesc* m, OrcData* d);
void orc_solve_pgs(const UhcModelDesc* m, OrcData* d);
void orc_forward(const UhcModelDesc* m, OrcData* d);
void orc_euler(const UhcModelDesc* m, OrcData* d);
void orc_step(const UhcModelDesc* m, OrcData* d);
void orc_set_state(const UhcModelDesc* m, OrcData* d, const double* qpos, const double* qvel);
void orc_pd_torque(const UhcModelDesc* m, const UhcCtrlDesc* c, OrcData* d, const double* action,
                   const double* target_base, int it);
void orc_rfc_implicit(const UhcModelDesc* m, const UhcCtrlDesc* c, OrcData* d, const double* action);
void orc_rfc_explicit(const UhcModelDesc* m, const UhcCtrlDesc* c, OrcData* d, const double* action);
void orc_do_simulation(const UhcModelDesc* m, const UhcCtrlDesc* c, OrcData* d, const double* action,
                       const double* target_base);
/* the same with solver 0 forced in the substeps whose bit is set (follows the device path's per-substep fallback, UHC_F_REDO bits 8+) */
void orc_do_simulation_mixed(const UhcModelDesc* m, const UhcCtrlDesc* c, OrcData* d, const double* action,
                             const double* target_base, unsigned sweep_mask);
void orc_set_threads(int n);
void orc_batch_do_simulation(const UhcModelDesc* m, const UhcCtrlDesc* c, OrcData** ds, int n_env,
                             const double* actions, const double* target_base);
int orc_get(const UhcModelDesc* m, const OrcData* d, const char* name, double* out, int max);
int orc_get_int(const OrcData* d, const char* name);
int orc_solve_primal(const UhcModelDesc* m, OrcData* d);
int orc_solve_active_set(const UhcModelDesc* m, OrcData* d);
void orc_set(const UhcModelDesc* m, OrcData* d, const char* name, const double* in);
#endif
