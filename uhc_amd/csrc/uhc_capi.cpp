// uhc_capi.cpp -- host side of libuhc_amd.so: the C-ABI declared in include/uhc_amd.h.
// Owns model copies, derived topology tables, device buffers and kernel launches.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/uhc_amd.h"
#include "uhc_device.h"

// the instantiations of uhc_step_kernel<MODE, FAST, DENSE> live in separate translation units (uhc_k_*.hip) so that they compile in parallel
#define UHC_DECL_LAUNCH(fn)                                                                                                               \
    extern "C" hipError_t fn(const KernelArgs* A, const double* d_action, const double* d_tbase, const int* d_active, size_t lds_bytes, \
                             hipStream_t stream);                                                                                        \
    extern "C" hipError_t fn##_lds(size_t lds_bytes);
UHC_DECL_LAUNCH(uhc_launch_m0_fast) UHC_DECL_LAUNCH(uhc_launch_m0_fast_dense) UHC_DECL_LAUNCH(uhc_launch_m1_fast) UHC_DECL_LAUNCH(uhc_launch_m1_fast_dense)
UHC_DECL_LAUNCH(uhc_launch_m2_fast) UHC_DECL_LAUNCH(uhc_launch_m0_gen) UHC_DECL_LAUNCH(uhc_launch_m1_gen) UHC_DECL_LAUNCH(uhc_launch_m2_gen)
UHC_DECL_LAUNCH(uhc_launch_m0_big) UHC_DECL_LAUNCH(uhc_launch_m1_big) UHC_DECL_LAUNCH(uhc_launch_m0_gen_q) UHC_DECL_LAUNCH(uhc_launch_m0_big_q) UHC_DECL_LAUNCH(uhc_launch_m0_huge_q)
// mode 0: control step, 1: forward only, 2: kinematics only; tier 1: the fast kernel, 2: general, 3: large; dense: the model has body-body contacts
static hipError_t uhc_launch_step(int mode, int tier, const KernelArgs* A, const double* d_action, const double* d_tbase, const int* d_active,
                                  size_t lds_bytes, hipStream_t stream) {
    const bool dense = A->cf.ndense > 0 || A->ball_limits;  // (limited ball joints: the instantiation that carries their rows)
    const bool fast = tier == 1;
    if (A->list) return (tier == 4 ? uhc_launch_m0_huge_q : tier == 3 ? uhc_launch_m0_big_q : uhc_launch_m0_gen_q)(A, d_action, d_tbase, d_active, lds_bytes, stream);  // queue consumers (mode 0, tiers 2 / 3 / 4)
    if (tier == 3) return (mode == 0 ? uhc_launch_m0_big : uhc_launch_m1_big)(A, d_action, d_tbase, d_active, lds_bytes, stream);
    if (!fast) return (mode == 0 ? uhc_launch_m0_gen : mode == 1 ? uhc_launch_m1_gen : uhc_launch_m2_gen)(A, d_action, d_tbase, d_active, lds_bytes, stream);
    if (mode == 2) return uhc_launch_m2_fast(A, d_action, d_tbase, d_active, lds_bytes, stream);
    if (mode == 0) return (dense ? uhc_launch_m0_fast_dense : uhc_launch_m0_fast)(A, d_action, d_tbase, d_active, lds_bytes, stream);
    return (dense ? uhc_launch_m1_fast_dense : uhc_launch_m1_fast)(A, d_action, d_tbase, d_active, lds_bytes, stream);
}
static hipError_t uhc_set_lds_limit(size_t lds_bytes, size_t lds_bytes_fast, size_t lds_bytes_big) {
    hipError_t e;
    if (lds_bytes_big && ((e = uhc_launch_m0_big_lds(lds_bytes_big)) != hipSuccess || (e = uhc_launch_m1_big_lds(lds_bytes_big)) != hipSuccess || (e = uhc_launch_m0_big_q_lds(lds_bytes_big)) != hipSuccess || (e = uhc_launch_m0_huge_q_lds(lds_bytes_big)) != hipSuccess)) return e;
    if ((e = uhc_launch_m0_gen_q_lds(lds_bytes)) != hipSuccess) return e;
    if ((e = uhc_launch_m0_gen_lds(lds_bytes)) != hipSuccess || (e = uhc_launch_m1_gen_lds(lds_bytes)) != hipSuccess || (e = uhc_launch_m2_gen_lds(lds_bytes)) != hipSuccess) return e;
    if ((e = uhc_launch_m0_fast_lds(lds_bytes_fast)) != hipSuccess || (e = uhc_launch_m0_fast_dense_lds(lds_bytes_fast)) != hipSuccess) return e;
    if ((e = uhc_launch_m1_fast_lds(lds_bytes_fast)) != hipSuccess || (e = uhc_launch_m1_fast_dense_lds(lds_bytes_fast)) != hipSuccess) return e;
    return uhc_launch_m2_fast_lds(lds_bytes_fast);
}
extern "C" hipError_t uhc_launch_tier_lists(const int* tier, const int* d_active, int n_env, int* tier_now, int* lists, int* counts, int* cursors, int* fin,
                                            const int* cost, const int* fresh, int* order, int launch4, int* pend3, hipStream_t stream);
extern "C" hipError_t uhc_launch_gate(const int* started, int want, int* waited, long long* trace, hipStream_t stream);
extern "C" hipError_t uhc_launch_set_state(const DevState* s, int nq, int nv, int nu, const int* env_ids, int n,
                                           const double* qpos, const double* qvel, int* mask, hipStream_t stream);

static thread_local std::string g_err;
static int fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}
#define HIP_OK(expr)                                                                      \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess) return fail("%s: %s", #expr, hipGetErrorString(e__));      \
    } while (0)

extern "C" const char* uhc_last_error(void) { return g_err.c_str(); }
extern "C" int32_t uhc_abi_version(void) { return UHC_ABI_VERSION; }
// what kind of build this library is: bit 0 = the solver's measurement switches are compiled in (-DUHC_EXPERIMENTS: UHC_DEBUG bits 8-12 act), bit 1 = stage
// cycle counters (-DUHC_STAGE_PROF), bit 2 = poisoned LDS (-DUHC_POISON_LDS), bit 3 = LDS guard words (-DUHC_GUARD_LDS).  The shipped library returns 0.
extern "C" int32_t uhc_build_flags(void) {
    int32_t f = 0;
#ifdef UHC_EXPERIMENTS
    f |= 1;
#endif
#ifdef UHC_STAGE_PROF
    f |= 2;
#endif
#ifdef UHC_POISON_LDS
    f |= 4;
#endif
#ifdef UHC_GUARD_LDS
    f |= 8;
#endif
    return f;
}

// ------------------------------------------------------------------ model (host copy)
struct UhcModel {
    UhcModelDesc d;  // scalars; pointers re-targeted at the vectors below
    std::vector<int32_t> body_parentid, body_jntadr, body_jntnum, body_dofadr, body_dofnum;
    std::vector<double> body_pos, body_quat, body_ipos, body_iquat, body_mass, body_inertia, body_invweight0;
    std::vector<int32_t> jnt_type, jnt_bodyid, jnt_qposadr, jnt_dofadr, jnt_limited;
    std::vector<double> jnt_pos, jnt_axis, jnt_range, jnt_stiffness, jnt_margin, qpos0, qpos_spring;
    std::vector<int32_t> dof_bodyid, dof_jntid, dof_parentid, dof_madr;
    std::vector<double> dof_armature, dof_damping, dof_frictionloss, dof_invweight0;
    std::vector<int32_t> geom_type, geom_bodyid, geom_contype, geom_conaffinity, geom_condim, geom_vertadr, geom_vertnum;
    std::vector<double> geom_pos, geom_quat, geom_size, geom_friction, geom_margin, geom_gap, geom_solref, geom_solimp,
        geom_rbound, geom_center, mesh_vert;
    std::vector<int32_t> mesh_adjadr, mesh_adj, exclude_pair, actuator_dofid;
    std::vector<double> actuator_gear;
};
template <class T>
static void take(std::vector<T>& v, const T*& p, size_t n) {
    v.assign(p, p + n);
    if (v.empty()) v.push_back(T());
    p = v.data();
}

extern "C" int32_t uhc_model_create(const UhcModelDesc* in, UhcModel** out) {
    if (!in || !out) return fail("uhc_model_create: null argument");
    if (in->nbody < 1 || in->nbody > UHC_WAVE) return fail("uhc_model_create: nbody=%d outside [1,64]", in->nbody);
    if (in->nv < 1 || in->nv > 2 * UHC_WAVE) return fail("uhc_model_create: nv=%d outside [1,128]", in->nv);
    if (in->njnt > 2 * UHC_WAVE) return fail("uhc_model_create: njnt=%d > 128", in->njnt);
    UhcModel* m = new UhcModel();
    m->d = *in;
    UhcModelDesc& d = m->d;
    const size_t nb = d.nbody, nj = d.njnt, nv = d.nv, ng = d.ngeom;
    take(m->body_parentid, d.body_parentid, nb); take(m->body_jntadr, d.body_jntadr, nb);
    take(m->body_jntnum, d.body_jntnum, nb); take(m->body_dofadr, d.body_dofadr, nb); take(m->body_dofnum, d.body_dofnum, nb);
    take(m->body_pos, d.body_pos, 3 * nb); take(m->body_quat, d.body_quat, 4 * nb); take(m->body_ipos, d.body_ipos, 3 * nb);
    take(m->body_iquat, d.body_iquat, 4 * nb); take(m->body_mass, d.body_mass, nb); take(m->body_inertia, d.body_inertia, 3 * nb);
    take(m->body_invweight0, d.body_invweight0, 2 * nb);
    take(m->jnt_type, d.jnt_type, nj); take(m->jnt_bodyid, d.jnt_bodyid, nj); take(m->jnt_qposadr, d.jnt_qposadr, nj);
    take(m->jnt_dofadr, d.jnt_dofadr, nj); take(m->jnt_limited, d.jnt_limited, nj);
    take(m->jnt_pos, d.jnt_pos, 3 * nj); take(m->jnt_axis, d.jnt_axis, 3 * nj); take(m->jnt_range, d.jnt_range, 2 * nj);
    take(m->jnt_stiffness, d.jnt_stiffness, nj); take(m->jnt_margin, d.jnt_margin, nj);
    take(m->qpos0, d.qpos0, d.nq); take(m->qpos_spring, d.qpos_spring, d.nq);
    take(m->dof_bodyid, d.dof_bodyid, nv); take(m->dof_jntid, d.dof_jntid, nv); take(m->dof_parentid, d.dof_parentid, nv);
    take(m->dof_madr, d.dof_madr, nv + 1);
    take(m->dof_armature, d.dof_armature, nv); take(m->dof_damping, d.dof_damping, nv);
    take(m->dof_frictionloss, d.dof_frictionloss, nv); take(m->dof_invweight0, d.dof_invweight0, nv);
    take(m->geom_type, d.geom_type, ng); take(m->geom_bodyid, d.geom_bodyid, ng); take(m->geom_contype, d.geom_contype, ng);
    take(m->geom_conaffinity, d.geom_conaffinity, ng); take(m->geom_condim, d.geom_condim, ng);
    take(m->geom_vertadr, d.geom_vertadr, ng); take(m->geom_vertnum, d.geom_vertnum, ng);
    take(m->geom_pos, d.geom_pos, 3 * ng); take(m->geom_quat, d.geom_quat, 4 * ng); take(m->geom_size, d.geom_size, 3 * ng);
    take(m->geom_friction, d.geom_friction, 3 * ng); take(m->geom_margin, d.geom_margin, ng); take(m->geom_gap, d.geom_gap, ng);
    take(m->geom_solref, d.geom_solref, 2 * ng); take(m->geom_solimp, d.geom_solimp, 5 * ng);
    take(m->geom_rbound, d.geom_rbound, ng); take(m->geom_center, d.geom_center, 3 * ng);
    take(m->mesh_vert, d.mesh_vert, 3 * (size_t)d.nmeshvert);
    take(m->mesh_adjadr, d.mesh_adjadr, (size_t)d.nmeshvert + 1); take(m->mesh_adj, d.mesh_adj, d.nmeshadj);
    take(m->exclude_pair, d.exclude_pair, 2 * (size_t)d.nexclude);
    take(m->actuator_dofid, d.actuator_dofid, d.nu); take(m->actuator_gear, d.actuator_gear, 3 * (size_t)d.nu);
    // structural checks the kernels rely on
    for (size_t b = 1; b < nb; b++)
        if (d.body_parentid[b] >= (int)b) { delete m; return fail("uhc_model_create: bodies must be in depth-first order"); }
    for (size_t i = 0; i < nv; i++)
        if (d.dof_parentid[i] >= (int)i) { delete m; return fail("uhc_model_create: dofs must be in depth-first order"); }
    *out = m;
    return 0;
}
extern "C" void uhc_model_free(UhcModel* m) { delete m; }
extern "C" int32_t uhc_model_nM(const UhcModel* m) { return m ? m->d.dof_madr[m->d.nv] : -1; }

// ------------------------------------------------------------------ batch
struct UhcBatch {
    int n_env = 0, device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    KernelArgs A;
    size_t lds_bytes = 0, lds_bytes_fast = 0, lds_bytes_big = 0;
    bool use_fast = true;
    bool general_only = false;
    // uhc_batch_set_kernel_path(2): sticky tiers -- every env starts a step in the tier that computed its last one (DevState::tier)
    int path_mode = 0;
    hipStream_t side_stream = nullptr, side_stream3 = nullptr, side_stream4 = nullptr;  // kernel path 2: the general / large tiers' own envs run beside the fast tier's
    hipEvent_t ev_fork = nullptr, ev_side1 = nullptr, ev_side2 = nullptr, ev_side3 = nullptr;
    int* tier_now = nullptr;
    bool large_first = false;  // the large tier's consumers are launched (and resident) before the general tier's
    int n_cu = 256;
    std::vector<std::pair<char*, size_t>> fences;  // UHC_GUARD_LDS=1: (base, payload bytes) of every fenced device array
    int* d_guard_hits = nullptr;  // UHC_GUARD_LDS=1: the kernels' report (KernelArgs::guard_hits), printed by uhc_batch_sync / uhc_batch_free
    int guard_reported = 0;
    int* d_order = nullptr;  // launch order of the fast tier under sticky tiers (uhc_tier_lists_kernel)
    int aborts_seen = 0, abort_events = 0;
    long long queues_off_until = 0;
    int q2_wait_min = 16;    // at least so many general-tier consumers wait for hand-ons (UHC_Q2_WAIT)
    int q2_div = 1;          // waiting general-tier consumers per expected env: 1 / q2_div (UHC_Q2_DIV)
    int q2_max = 256;        // most general-tier consumers beside a fast tier that still has most of the envs (UHC_Q2_MAX)
    int q4_max = 16;         // most tier-4 consumers (UHC_Q4_MAX; 0: none -- what the large tier hands on waits for the chained launch at the end of the step)
    int q3_max = 32;         // most large-tier consumers in that regime (UHC_Q3_MAX)
    int *d_lists = nullptr, *d_counts = nullptr, *d_cursors = nullptr, *d_fin = nullptr;
    bool queues_off = false;
    int* h_counts = nullptr;  // pinned [8][8]: give-ups, gate wait, final queue lengths [2], [3], queue lengths at the head of the step [4], [5]; the last steps', copied back asynchronously
    hipEvent_t cnt_ev[8] = {};
    long long cnt_step = 0;
    std::vector<void*> allocs;
    int nM = 0;
    int* reset_mask = nullptr;
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_used, ev_free;
    int n_models = 1;
    int n_trailing_free = 0;  // free bodies at the end of the model (objects)
    // field table
    void* field_ptr[19] = {nullptr};
    int64_t field_count[19] = {0};
};

template <class T>
static int upload(UhcBatch* b, const std::vector<T>& h, const T** dptr) {
    void* p = nullptr;
    size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
    HIP_OK(hipMalloc(&p, bytes));
    b->allocs.push_back(p);
    if (!h.empty()) HIP_OK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *dptr = (const T*)p;
    return 0;
}
// UHC_GUARD_LDS=1 (debug, with the guard words in LDS): every zero-initialised device array of a batch -- the state the kernels WRITE -- sits between two
// 256-byte fences of 0xA5; uhc_batch_free checks them ("uhc guard: ... HBM ...")
static bool hbm_guard_on() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("UHC_GUARD_LDS"); v = (e && (e[0] == '1' || e[0] == '2')) ? 1 : 0; }
    return v == 1;
}
template <class T>
static int dalloc(UhcBatch* b, size_t n, T** dptr) {
    void* p = nullptr;
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    if (hbm_guard_on()) {
        HIP_OK(hipMalloc(&p, bytes + 512));
        HIP_OK(hipMemset(p, 0xA5, bytes + 512));
        HIP_OK(hipMemset((char*)p + 256, 0, bytes));
        b->allocs.push_back(p);
        b->fences.push_back({(char*)p, bytes});
        *dptr = (T*)((char*)p + 256);
        return 0;
    }
    HIP_OK(hipMalloc(&p, bytes));
    HIP_OK(hipMemset(p, 0, bytes));
    b->allocs.push_back(p);
    *dptr = (T*)p;
    return 0;
}
#define TRY(x) do { if (x) return 1; } while (0)

static bool same_topology(const UhcModelDesc& a, const UhcModelDesc& b) {
    if (a.nq != b.nq || a.nv != b.nv || a.nu != b.nu || a.nbody != b.nbody || a.njnt != b.njnt || a.ngeom != b.ngeom ||
        a.nmeshvert != b.nmeshvert || a.nexclude != b.nexclude)  // (hull graphs may differ: they live in the model blobs)
        return false;
    auto eq = [](const int32_t* x, const int32_t* y, size_t n) { return !memcmp(x, y, n * 4); };
    return eq(a.body_parentid, b.body_parentid, a.nbody) && eq(a.jnt_type, b.jnt_type, a.njnt) &&
           eq(a.jnt_bodyid, b.jnt_bodyid, a.njnt) && eq(a.dof_parentid, b.dof_parentid, a.nv) &&
           eq(a.geom_type, b.geom_type, a.ngeom) && eq(a.geom_bodyid, b.geom_bodyid, a.ngeom) &&
           eq(a.geom_vertadr, b.geom_vertadr, a.ngeom) && eq(a.geom_vertnum, b.geom_vertnum, a.ngeom) &&
           eq(a.geom_contype, b.geom_contype, a.ngeom) && eq(a.geom_conaffinity, b.geom_conaffinity, a.ngeom) &&
           eq(a.jnt_limited, b.jnt_limited, a.njnt) && eq(a.geom_condim, b.geom_condim, a.ngeom);
}

static void build_blob(const UhcModelDesc& d, DevNumOff& o, std::vector<double>& blob, int adjdeg) {
    blob.clear();
    auto put = [&](const double* p, size_t n) { int off = (int)blob.size(); blob.insert(blob.end(), p, p + n); return off; };
    o.body_pos = put(d.body_pos, 3 * d.nbody); o.body_quat = put(d.body_quat, 4 * d.nbody);
    o.body_ipos = put(d.body_ipos, 3 * d.nbody); o.body_iquat = put(d.body_iquat, 4 * d.nbody);
    o.body_mass = put(d.body_mass, d.nbody); o.body_inertia = put(d.body_inertia, 3 * d.nbody);
    o.body_invweight0 = put(d.body_invweight0, 2 * d.nbody);
    o.jnt_pos = put(d.jnt_pos, 3 * d.njnt); o.jnt_axis = put(d.jnt_axis, 3 * d.njnt); o.jnt_range = put(d.jnt_range, 2 * d.njnt);
    o.jnt_stiffness = put(d.jnt_stiffness, d.njnt); o.jnt_margin = put(d.jnt_margin, d.njnt);
    o.qpos0 = put(d.qpos0, d.nq); o.qpos_spring = put(d.qpos_spring, d.nq);
    o.dof_armature = put(d.dof_armature, d.nv); o.dof_damping = put(d.dof_damping, d.nv);
    o.dof_frictionloss = put(d.dof_frictionloss, d.nv); o.dof_invweight0 = put(d.dof_invweight0, d.nv);
    o.geom_pos = put(d.geom_pos, 3 * d.ngeom); o.geom_quat = put(d.geom_quat, 4 * d.ngeom);
    o.geom_friction = put(d.geom_friction, 3 * d.ngeom); o.geom_margin = put(d.geom_margin, d.ngeom);
    o.geom_gap = put(d.geom_gap, d.ngeom); o.geom_solref = put(d.geom_solref, 2 * d.ngeom);
    o.geom_solimp = put(d.geom_solimp, 5 * d.ngeom); o.geom_rbound = put(d.geom_rbound, d.ngeom);
    o.geom_center = put(d.geom_center, 3 * d.ngeom);
    o.mesh_vert = put(d.mesh_vert, 3 * (size_t)d.nmeshvert);
    {   // box around every hull in its body's frame (centre, half extents): the second cull of the convex pairs (after the bounding spheres)
        std::vector<double> box((size_t)6 * std::max(d.ngeom, 1), 0.0);
        for (int g = 0; g < d.ngeom; g++) {
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            const int v0 = d.geom_vertadr ? d.geom_vertadr[g] : 0, nvt = d.geom_vertnum ? d.geom_vertnum[g] : 0;
            for (int v = v0; v < v0 + nvt; v++)
                for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], d.mesh_vert[3 * v + k]); hi[k] = std::max(hi[k], d.mesh_vert[3 * v + k]); }
            const double rr = (d.geom_type[g] == UHC_GEOM_SPHERE || d.geom_type[g] == UHC_GEOM_CAPSULE) ? d.geom_size[3 * g] : 0.0;  // a rounded hull reaches its radius beyond the core
            for (int k = 0; k < 3; k++) {
                if (nvt > 0) { box[6 * g + k] = 0.5 * (lo[k] + hi[k]); box[6 * g + 3 + k] = 0.5 * (hi[k] - lo[k]) + rr; }
                else { box[6 * g + k] = d.geom_center[3 * g + k]; box[6 * g + 3 + k] = d.geom_rbound[g]; }  // (no hull: the bounding sphere's box)
            }
        }
        o.geom_box = put(box.data(), box.size());
        std::vector<double> rad((size_t)std::max(d.ngeom, 1), 0.0);
        for (int g = 0; g < d.ngeom; g++)
            if (d.geom_type[g] == UHC_GEOM_SPHERE || d.geom_type[g] == UHC_GEOM_CAPSULE) rad[g] = d.geom_size[3 * g];
        o.geom_radius = put(rad.data(), rad.size());
    }
    o.actuator_gear = put(d.actuator_gear, 3 * (size_t)d.nu);
    o.meaninertia = put(&d.meaninertia, 1);
    {   // hull graph: fixed stride, neighbour order = the CSR's order (the multi-contact rule takes neighbours in that order)
        std::vector<int32_t> adj((size_t)std::max(d.nmeshvert, 1) * adjdeg + 2, -1);
        for (int v = 0; v < d.nmeshvert; v++)
            for (int e = d.mesh_adjadr[v], k = 0; e < d.mesh_adjadr[v + 1]; e++, k++) adj[(size_t)v * adjdeg + k] = d.mesh_adj[e];
        std::vector<double> packed((adj.size() + 1) / 2, 0.0);
        memcpy(packed.data(), adj.data(), adj.size() / 2 * 2 * sizeof(int32_t));
        o.mesh_adj = put(packed.data(), packed.size());
    }
    while (blob.size() % 2) blob.push_back(0.0);
    o.stride = (int)blob.size();
}

extern "C" int32_t uhc_batch_create(const UhcModel* const* models, int32_t n_models, const int32_t* h_env_model, int32_t n_env,
                                    int32_t device_id, const UhcCtrlDesc* ctrl, UhcBatch** out) {
    if (!models || n_models < 1 || n_env < 1 || !ctrl || !out) return fail("uhc_batch_create: bad argument");
    const UhcModelDesc& d = models[0]->d;
    for (int k = 1; k < n_models; k++)
        if (!same_topology(d, models[k]->d)) return fail("uhc_batch_create: model %d differs in topology from model 0", k);
    if (h_env_model)
        for (int e = 0; e < n_env; e++)
            if (h_env_model[e] < 0 || h_env_model[e] >= n_models) return fail("uhc_batch_create: env_model[%d] out of range", e);
    int ndev = 0;
    HIP_OK(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail("uhc_batch_create: device %d not present (%d devices)", device_id, ndev);
    HIP_OK(hipSetDevice(device_id));
    // free bodies at the end of the model (objects: a body under the world with one free joint and no children, after the humanoid)
    int n_trail = 0;
    for (int j = d.njnt - 1; j >= 1 && d.jnt_type[j] == UHC_JNT_FREE && d.body_parentid[d.jnt_bodyid[j]] == 0 && d.jnt_bodyid[j] == d.nbody - 1 - n_trail; j--) n_trail++;
    // the stable-PD controller works on the humanoid's block of M (the reference cuts M to [:qvel_lim, :qvel_lim], humanoid_im.py:1021-1022;
    // objects are separate trees, so the solve over all dofs gives the same humanoid accelerations)
    if (ctrl->action_type == 0 && d.nq - 7 * n_trail != d.nv - 6 * n_trail + 1) return fail("uhc_batch_create: PD control expects a free root + scalar joints (+ free objects behind them)");
    if (ctrl->action_type == 0 && d.nu != d.nv - 6 * n_trail - 6) return fail("uhc_batch_create: PD control expects one motor per non-root dof of the humanoid");

    UhcBatch* b = new UhcBatch();
    b->n_env = n_env;
    b->device = device_id;
    b->n_trailing_free = n_trail;
    KernelArgs& A = b->A;
    memset(&A, 0, sizeof A);
    DevTopo& T = A.t;
    const int nb = d.nbody, nv = d.nv, nj = d.njnt, ng = d.ngeom;
    T.nq = d.nq; T.nv = nv; T.nu = d.nu; T.nbody = nb; T.njnt = nj; T.ngeom = ng; T.nmeshvert = d.nmeshvert;
    T.nM = d.dof_madr[nv];
    T.iterations = d.iterations; T.plane_mesh_maxcon = d.plane_mesh_maxcon; T.solver = d.solver;
    T.timestep = d.timestep; T.tolerance = d.tolerance;
    for (int k = 0; k < 3; k++) T.gravity[k] = d.gravity[k];
    b->nM = T.nM;
    T.has_damping = 0;
    for (int k = 0; k < n_models; k++)
        for (int i = 0; i < nv; i++) T.has_damping |= models[k]->d.dof_damping[i] > 0;
    for (int j = 0; j < d.njnt; j++) A.ball_limits |= d.jnt_type[j] == UHC_JNT_BALL && d.jnt_limited[j] != 0;

    // ---- derived topology tables
    std::vector<int> body_depth(nb, 0), body_rootid(nb, 0), body_nsub(nb, 1), body_lastdof(nb, -1);
    for (int i = 1; i < nb; i++) {
        int p = d.body_parentid[i];
        body_depth[i] = body_depth[p] + 1;
        body_rootid[i] = p == 0 ? i : body_rootid[p];
        body_lastdof[i] = d.body_dofnum[i] > 0 ? d.body_dofadr[i] + d.body_dofnum[i] - 1 : body_lastdof[p];
    }
    for (int i = nb - 1; i > 0; i--) body_nsub[d.body_parentid[i]] += body_nsub[i];
    for (int i = 1; i < nb; i++) {  // DFS order check: subtree must be the contiguous range [i, i+nsub)
        for (int c = i + 1; c < i + body_nsub[i]; c++) {
            int p = c;
            while (p > i) p = d.body_parentid[p];
            if (p != i) { delete b; return fail("uhc_batch_create: bodies are not in depth-first order"); }
        }
    }
    T.body_maxdepth = *std::max_element(body_depth.begin(), body_depth.end());
    if (nb > 32) { delete b; return fail("uhc_batch_create: %d bodies > 32 (subtree force sums use three 64-lane passes of 6 components)", nb); }
    for (int i = 1; i < nb; i++)
        if (d.body_jntnum[i] > 8) { delete b; return fail("uhc_batch_create: body %d has %d joints (> 8)", i, d.body_jntnum[i]); }
    std::vector<int> dof_depth(nv, 0), dof_ndesc(nv, 0);
    for (int i = 0; i < nv; i++) dof_depth[i] = d.dof_parentid[i] < 0 ? 0 : dof_depth[d.dof_parentid[i]] + 1;
    for (int i = nv - 1; i >= 0; i--)
        if (d.dof_parentid[i] >= 0) dof_ndesc[d.dof_parentid[i]] += dof_ndesc[i] + 1;
    for (int i = 0; i < nv; i++)
        for (int c = i + 1; c <= i + dof_ndesc[i]; c++) {
            int p = c;
            while (p > i) p = d.dof_parentid[p];
            if (p != i) { delete b; return fail("uhc_batch_create: dofs are not in depth-first order"); }
        }
    T.maxdepth = *std::max_element(dof_depth.begin(), dof_depth.end());
    if (T.maxdepth + 1 > 32) { delete b; return fail("uhc_batch_create: dof chain depth %d > 32 unsupported", T.maxdepth + 1); }
    const int YS = T.maxdepth + 1;
    std::vector<short> dof_anc((size_t)nv * YS, 0), m_row(T.nM), m_col(T.nM);
    for (int i = 0; i < nv; i++) {
        int k = i;
        for (int q = dof_depth[i]; q >= 0; q--) { dof_anc[(size_t)i * YS + q] = (short)k; k = d.dof_parentid[k]; }
        int adr = d.dof_madr[i];
        for (int j = i; j >= 0; j = d.dof_parentid[j], adr++) { m_row[adr] = (short)i; m_col[adr] = (short)j; }
    }
    std::vector<unsigned short> m_ij(T.nM + 4, 0);
    for (int e = 0; e < T.nM; e++) m_ij[e] = (unsigned short)((m_row[e] << 8) | m_col[e]);
    std::vector<unsigned char> ncommon((size_t)nv * nv, 0);
    for (int i = 0; i < nv; i++)
        for (int j = 0; j < nv; j++) {
            int q = 0;
            const int lim = std::min(dof_depth[i], dof_depth[j]);
            while (q <= lim && dof_anc[(size_t)i * YS + q] == dof_anc[(size_t)j * YS + q]) q++;
            ncommon[(size_t)i * nv + j] = (unsigned char)q;
        }
    // statically filtered collision pairs (plane, mesh)
    std::vector<int> pg1, pg2, cg1, cg2;
    int skipped_pairs = 0;
    for (int g1 = 0; g1 < ng; g1++)
        for (int g2 = g1 + 1; g2 < ng; g2++) {
            int b1 = d.geom_bodyid[g1], b2 = d.geom_bodyid[g2];
            if (!((d.geom_contype[g1] & d.geom_conaffinity[g2]) || (d.geom_contype[g2] & d.geom_conaffinity[g1]))) continue;
            if (b1 == b2) continue;
            if (b1 != 0 && b2 != 0 && (d.body_parentid[b1] == b2 || d.body_parentid[b2] == b1)) continue;
            bool ex = false;
            for (int e = 0; e < d.nexclude; e++) {
                int x = d.exclude_pair[2 * e], y = d.exclude_pair[2 * e + 1];
                if ((x == b1 && y == b2) || (x == b2 && y == b1)) ex = true;
            }
            if (ex) continue;
            int t1 = d.geom_type[g1], t2 = d.geom_type[g2];
            if (body_lastdof[b1] < 0 && body_lastdof[b2] < 0) continue;  // two static bodies never collide ([MJ-ext] same weld id)
            // hulls: meshes and the rounded hulls (sphere = one core vertex, capsule = two, + a radius: include/uhc_amd.h) -- with vertices in the mesh tables
            auto hull = [&](int g, int t) { return (t == UHC_GEOM_MESH || t == UHC_GEOM_SPHERE || t == UHC_GEOM_CAPSULE) && d.geom_vertnum[g] > 0; };
            if (t1 == UHC_GEOM_PLANE && hull(g2, t2) && b1 == 0) { pg1.push_back(g1); pg2.push_back(g2); }
            else if (t2 == UHC_GEOM_PLANE && hull(g1, t1) && b2 == 0) { pg1.push_back(g2); pg2.push_back(g1); }
            else if (hull(g1, t1) && hull(g2, t2)) { cg1.push_back(g1); cg2.push_back(g2); }
            else skipped_pairs++;
        }
    if (skipped_pairs) { delete b; return fail("uhc_batch_create: %d collision pairs of unsupported geom types (built: plane-hull, hull-hull; a hull is a mesh, a sphere or a capsule)", skipped_pairs); }
    T.npair = (int)pg1.size();
    T.ncpair = (int)cg1.size();
    std::vector<int> dof_rootid(nv, 0);
    for (int i = 0; i < nv; i++) dof_rootid[i] = body_rootid[d.dof_bodyid[i]];

    auto ivec = [](const int32_t* p, size_t n) { return std::vector<int>(p, p + n); };
    TRY(upload(b, ivec(d.body_parentid, nb), &T.body_parentid)); TRY(upload(b, ivec(d.body_jntadr, nb), &T.body_jntadr));
    TRY(upload(b, ivec(d.body_jntnum, nb), &T.body_jntnum)); TRY(upload(b, ivec(d.body_dofadr, nb), &T.body_dofadr));
    TRY(upload(b, ivec(d.body_dofnum, nb), &T.body_dofnum)); TRY(upload(b, body_rootid, &T.body_rootid));
    TRY(upload(b, body_nsub, &T.body_nsub)); TRY(upload(b, body_lastdof, &T.body_lastdof)); TRY(upload(b, body_depth, &T.body_depth));
    TRY(upload(b, ivec(d.jnt_type, nj), &T.jnt_type)); TRY(upload(b, ivec(d.jnt_bodyid, nj), &T.jnt_bodyid));
    TRY(upload(b, ivec(d.jnt_qposadr, nj), &T.jnt_qposadr)); TRY(upload(b, ivec(d.jnt_dofadr, nj), &T.jnt_dofadr));
    TRY(upload(b, ivec(d.jnt_limited, nj), &T.jnt_limited));
    TRY(upload(b, ivec(d.dof_bodyid, nv), &T.dof_bodyid)); TRY(upload(b, ivec(d.dof_jntid, nv), &T.dof_jntid));
    TRY(upload(b, ivec(d.dof_parentid, nv), &T.dof_parentid)); TRY(upload(b, ivec(d.dof_madr, nv + 1), &T.dof_madr));
    TRY(upload(b, dof_depth, &T.dof_depth)); TRY(upload(b, dof_ndesc, &T.dof_ndesc));
        TRY(upload(b, dof_anc, &T.dof_anc)); TRY(upload(b, ncommon, &T.dof_ncommon)); TRY(upload(b, m_row, &T.m_row)); TRY(upload(b, m_col, &T.m_col)); TRY(upload(b, m_ij, &T.m_ij));
    TRY(upload(b, ivec(d.geom_type, ng), &T.geom_type)); TRY(upload(b, ivec(d.geom_bodyid, ng), &T.geom_bodyid));
    TRY(upload(b, ivec(d.geom_condim, ng), &T.geom_condim)); TRY(upload(b, ivec(d.geom_vertadr, ng), &T.geom_vertadr));
    TRY(upload(b, ivec(d.geom_vertnum, ng), &T.geom_vertnum));
    TRY(upload(b, pg1, &T.pair_g1)); TRY(upload(b, pg2, &T.pair_g2));
    TRY(upload(b, cg1, &T.cpair_g1)); TRY(upload(b, cg2, &T.cpair_g2)); TRY(upload(b, dof_rootid, &T.dof_rootid));
    TRY(upload(b, ivec(d.actuator_dofid, d.nu), &T.actuator_dofid));

    // ---- numeric blobs, one per model
    std::vector<double> all, one;
    A.adjdeg = 1;
    for (int k = 0; k < n_models; k++)
        for (int v = 0; v < models[k]->d.nmeshvert; v++) A.adjdeg = std::max(A.adjdeg, models[k]->d.mesh_adjadr[v + 1] - models[k]->d.mesh_adjadr[v]);
    if (A.adjdeg > UHC_WAVE - 1) { delete b; return fail("uhc_batch_create: a hull vertex with %d neighbours (> 63)", A.adjdeg); }
    for (int k = 0; k < n_models; k++) {
        build_blob(models[k]->d, A.o, one, A.adjdeg);
        all.insert(all.end(), one.begin(), one.end());
    }
    TRY(upload(b, all, &A.s.model_blob));
    if (h_env_model) TRY(upload(b, std::vector<int>(h_env_model, h_env_model + n_env), &A.s.env_model));
    else if (n_models > 1) TRY(upload(b, std::vector<int>(n_env, 0), &A.s.env_model));  // selectable later (uhc_env_set_clip_models)
    b->n_models = n_models;

    // ---- controller
    DevCtrl& C = A.c;
    C.n_substeps = ctrl->n_substeps; C.action_type = ctrl->action_type; C.meta_pd = ctrl->meta_pd; C.rfc_mode = ctrl->rfc_mode;
    C.action_dim = ctrl->action_dim; C.rfc_scale = ctrl->rfc_scale; C.rfc_lim = ctrl->rfc_lim;
    {
        const double* q = ctrl->base_rot;
        double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
        if (n2 <= 0) { delete b; return fail("uhc_batch_create: zero base_rot"); }
        C.base_rot_inv[0] = q[0] / n2; C.base_rot_inv[1] = -q[1] / n2; C.base_rot_inv[2] = -q[2] / n2; C.base_rot_inv[3] = -q[3] / n2;
    }
    C.n_vf_body = 0; C.body_vf_dim = 0; C.vf_body = nullptr;
    if (C.rfc_mode == 2) {
        C.n_vf_body = ctrl->n_vf_body; C.body_vf_dim = ctrl->body_vf_dim;
        if (C.n_vf_body < 1 || C.n_vf_body > 64 || !ctrl->vf_body || (C.body_vf_dim != 6 && C.body_vf_dim != 9)) {
            delete b; return fail("uhc_batch_create: explicit RFC needs 1..64 vf bodies and body_vf_dim 6 or 9");
        }
        std::vector<int> vb(ctrl->vf_body, ctrl->vf_body + C.n_vf_body), seen(nb, 0);
        for (int v : vb) {
            if (v < 0 || v >= nb || seen[v]++) { delete b; return fail("uhc_batch_create: vf_body ids must be distinct model bodies"); }
        }
        TRY(upload(b, vb, &C.vf_body));
    } else if (C.rfc_mode != 0 && C.rfc_mode != 1) { delete b; return fail("uhc_batch_create: rfc_mode must be 0, 1 or 2"); }
    const int min_adim = d.nu + (C.rfc_mode == 1 ? 6 : C.rfc_mode == 2 ? C.n_vf_body * C.body_vf_dim : 0) + (C.meta_pd == 1 ? 2 * C.n_substeps : C.meta_pd == 2 ? 2 * d.nu : 0);
    if (C.action_dim < min_adim) { delete b; return fail("uhc_batch_create: action_dim %d < %d required by the controller", C.action_dim, min_adim); }
    auto dvec = [&](const double* p) { return std::vector<double>(p, p + d.nu); };
    TRY(upload(b, dvec(ctrl->jkp), &C.jkp)); TRY(upload(b, dvec(ctrl->jkd), &C.jkd));
    TRY(upload(b, dvec(ctrl->torque_lim), &C.torque_lim)); TRY(upload(b, dvec(ctrl->a_scale), &C.a_scale));

    // ---- LDS carve (doubles; every offset even => 16-byte aligned).  Every tier's layout has a persistent part (state, the factor of M,
    //      body poses, cdof) and ONE region shared by the dynamics temporaries (first half of a forward pass) and the constraint data
    //      (second half).  M itself is parked in registers between substeps (MPark), so no tier keeps a second copy in LDS.
    int off = 0;
    // Guard words (debug): with UHC_GUARD_LDS=1 every region carved below is followed by two doubles nobody owns.  A library built
    // with -DUHC_GUARD_LDS (tools/poison_build.py) fills them with the poison pattern -- the persistent regions' once per env, the constraint
    // phase's after the dynamics temporaries that overlay them are dead -- and reports any that changed (KernelArgs::guard_hits).  The regions
    // move by a few doubles; capacities and tiers stay as they are.  MPR's vertex staging, which runs across three regions on purpose, is off.
    const bool guard_selftest = getenv("UHC_GUARD_LDS") && getenv("UHC_GUARD_LDS")[0] == '2';  // (2: one "guard" sits ON qpos of the fast tier -- the report must fire: tests/test_gpu_poison.py)
    const bool guard_on = guard_selftest || (getenv("UHC_GUARD_LDS") && getenv("UHC_GUARD_LDS")[0] == '1');
    std::vector<int> g_persist[4], g_phase2[4];
    std::vector<int>* grec = nullptr;
    int gt = 0;  // the layout being carved: 0 fast, 1 general, 2 large, 3 tier 4
    auto carve = [&](int n) { int o = off; off += (n + 1) & ~1; if (guard_on && grec) { grec->push_back(off); off += 2; } return o; };
    auto end_guard = [&]() { if (guard_on) { g_phase2[gt].push_back(off); off += 2; } };
    if (T.nM > 64 * 24) { delete b; return fail("uhc_batch_create: nM %d > 1536 (the register tile that carries M between substeps)", T.nM); }
    A.nvp = (nv + 1) & ~1;
    { const char* dbg = getenv("UHC_DEBUG"); A.dbg = dbg ? atoi(dbg) : 0; if (guard_on) A.dbg |= 2; }
#ifndef UHC_EXPERIMENTS
    A.dbg &= ~0x1f00;  // bits 8-12 (working-set fill, consumer cap, sticky tier 4) are measurement switches of tools/ builds: the shipped library does not read them
#endif
    {   // sticky-tier marks: an env goes up a tier when it no longer fits (64 rows / 16 contacts / 12 body-body rows; the general tier's
        // capacities) and comes down again at 56 / 14 / 10 and at 7/8 of the general tier's.  Going up EARLIER (at 3/4 of a capacity, so
        // that no env finds out in mid-step) was measured on the self-colliding rollout: 51-55 k env-steps/s against 57 k -- the envs
        // parked a tier up cost more than the late hand-ons they avoid (profiles/r03_marks_sweep.txt)
        const int def[8] = {64, 16, 12, 56, 14, 10, 8, 7};
        for (int k = 0; k < 8; k++) A.marks[k] = def[k];
        A.marks[2] = -1;  // (body-body rows: set from the fast layout's dense slots below unless UHC_TIER_MARKS names them)
        if (const char* m = getenv("UHC_TIER_MARKS")) sscanf(m, "%d,%d,%d,%d,%d,%d,%d,%d", A.marks, A.marks + 1, A.marks + 2, A.marks + 3, A.marks + 4, A.marks + 5, A.marks + 6, A.marks + 7);
    }
    int end1 = 0;
    auto common = [&](DevLds& F, bool fast) {  // persistent part + phase 1; returns the offset where phase 2 starts (fast: with the (row, col) table of M in LDS)
        off = 0;
        g_persist[gt].clear(); g_phase2[gt].clear(); grec = &g_persist[gt];
        F.qpos = carve(d.nq); F.qvel = carve(nv); F.qacc = carve(nv); F.ctrl = carve(d.nu); F.applied = carve(nv);
        F.bias = carve(nv); F.smooth = carve(nv); F.z = carve(nv); F.dinv = carve(nv); F.sdinv = carve(nv);
        F.zero = carve(2);
        F.mij = fast ? carve((T.nM + 3) / 4 + 1) : 0;  // the larger tiers read the (row, col) table from L2 and keep the LDS for rows
        F.LD = carve(T.nM + 2); F.M = F.LD;
        F.cdof = carve(6 * nv);
        F.xpos = carve(3 * nb); F.xquat = carve(4 * nb); F.xmat = carve(9 * nb); F.xipos = carve(3 * nb); F.rootcom = carve(3 * nb);
        F.vec = fast ? F.z : carve(nv);  // the working sets accumulate z over islands in vec
        const int base = off;
        grec = nullptr;  // (phase 1: overlaid by the constraint data, no guards)
        // phase 1
        F.cinert = carve(10 * nb);
        const int r2 = off;
        F.ximat = carve(9 * nb); F.xanchor = carve(3 * nj); F.xaxis = carve(3 * nj);  // dead after k_com_pos
        const int endA = off;
        off = r2;
        F.cdofdot = carve(6 * nv);
        const int r3 = off;
        F.crb = carve(10 * nb);                                                        // k_crb only
        const int endB = off;
        off = r3;
        F.cvel = carve(6 * nb); F.cacc = carve(6 * nb); F.cfrc = carve(6 * nb);         // k_com_vel .. k_rne
        end1 = std::max(std::max(endA, endB), off);
        off = base;
        grec = &g_phase2[gt];
        return base;
    };
    // ---- fast tier: target 40 KiB per workgroup => 4 workgroups (one per SIMD) per CU
    {
        DevLds& F = A.lf;
        gt = 0;
        common(F, T.ncpair == 0);  // (the DENSE instantiations read the (row, col) table of M from L2, like the larger tiers: 2.4 KiB for rows)
        // UHC_FAST_DENSE = "KiB,dense rows[,contacts]": the dense fast tier's LDS budget, body-body row slots and contact capacity (experiments)
        int dense_kib = 52, fast_maxcon = UHC_FAST_MAXCON, fast_ndense = UHC_FAST_MAXTWO;
        if (T.ncpair > 0) fast_maxcon = UHC_FAST_MAXCON_DENSE;
        // A batch of four or more rounds of the 3-per-CU layout (>= 3072 envs on 256 CUs) is throughput-bound by how many fast-tier
        // workgroups a CU holds, not by the general tier's slowest env: it gets the 40 KiB layout with 6 body-body row slots -- 4 per CU, one
        // per SIMD; 12 % of the env-steps go through the general tier instead of 2.6 %.  Measured on the generated model class, env-steps/s
        // with 52 KiB / 12 slots -> 40 KiB / 6 slots: 1024 envs 88 k -> 88 k, 2048 envs 119 k -> 98 k (250 envs per step queue for the general
        // tier's consumers), 3072 envs 112 k -> 141 k, 4096 envs (configs[2]'s share of one GPU) 100 k -> 141 k.
        // Only for a humanoid of hinge joints without objects: a ball-joint humanoid folds into itself and boxes bring 16 rows each, their envs
        // need the 12 slots -- with 6, 95 % of configs[4]'s env-steps at 4096 envs went through the general tier (tier trace
        // r04_v6_tier_trace_configs4_4096.txt), `ball_rollout` took 53 ms per step against 37-43 ms with the wide layout.
        bool hinge_only = n_trail == 0;
        for (int j = 0; j < d.njnt; j++) hinge_only = hinge_only && d.jnt_type[j] != UHC_JNT_BALL;
        if (T.ncpair > 0 && n_env >= 3072 && hinge_only) { dense_kib = 40; fast_ndense = 6; }
        // A BALL-JOINT humanoid gets 9 body-body row slots instead of 12 (round 6): three dense rows less are 370 doubles more for the packed rows, whose storage names
        // two of three hand-ons of the ball-joint rollouts (long dof chains, three dofs per joint).  Measured on one box, 3 x 60 steps, env-steps/s with 12 / 9 / 8 / 7 slots:
        // `ball_rollout` 54.1 k / 65.8 k / 65.6 k / 62.4 k, `configs4` 57.3 k / 58.4 k / 57.5 k / 54.5 k (profiles/r06_fd_dense_slots.txt); the hinge models keep 12
        // (headline 96.5 k with 12, 96.4 k with 10, 95.4 k with 8; `shapes` 101.1 / 100.7 / 98.7 k).
        bool any_ball = false;
        for (int j = 0; j < d.njnt; j++) any_ball = any_ball || d.jnt_type[j] == UHC_JNT_BALL;
        if (T.ncpair > 0 && any_ball && dense_kib == 52) fast_ndense = 9;
        if (const char* fd = getenv("UHC_FAST_DENSE")) {
            int kib = 0, nd = 0, nc = 0;
            const int got = sscanf(fd, "%d,%d,%d", &kib, &nd, &nc);
            // (a model with convex pairs keeps at least one body-body slot: the layout drops the (row, col) table of M from LDS for such a model
            //  and the launcher picks the DENSE instantiation -- which reads the table from L2 -- by `cf.ndense > 0`: nd = 0 would launch
            //  <0, 1, false> on a layout without the table it indexes)
            if (got >= 2 && kib >= 32 && kib <= 160 && nd >= (T.ncpair > 0 ? 1 : 0) && nd <= UHC_FAST_MAXTWO) { dense_kib = kib; fast_ndense = nd; }
            if (got == 3 && nc >= 8 && nc <= UHC_GEN_MAXCON && T.ncpair > 0) fast_maxcon = nc;
        }
        F.con = carve(fast_maxcon * UHC_CON_STRIDE);
        F.rowMisc = carve(UHC_WAVE * 2);
        F.ncon_nefc = carve(2 + UHC_MAXTWO / 2);
        // models with body-body contacts (self-collision, objects) keep up to UHC_FAST_MAXTWO dense rows + their Delassus columns; they
        // get a third of the CU's LDS (3 workgroups per CU) instead of a quarter -- the stock floor-only model keeps its 40 KiB layout
        A.cf.maxefc = UHC_FAST_MAXEFC; A.cf.maxcon = fast_maxcon; A.cf.ld_delta = 0;
        A.cf.ndense = T.ncpair > 0 ? fast_ndense : 0;
        F.dense = carve(A.cf.ndense * A.nvp);
        // the dense rows' Delassus columns are written by the contact solve, when nothing reads the contacts any more (the rows are built):
        // they take the contacts' storage when they fit it (6 slots x 64 lanes = 16 contacts x 24 doubles), as in the larger tiers
        const bool dcol_alias = A.cf.ndense * UHC_WAVE <= fast_maxcon * UHC_CON_STRIDE && !(getenv("UHC_FAST_DCOL_OWN") && getenv("UHC_FAST_DCOL_OWN")[0] == '1');
        F.dcol = dcol_alias ? F.con : carve(A.cf.ndense * UHC_WAVE);
        F.Y = off;
        // (52 KiB, not 160 / 3 = 53.3: the LDS is handed out in granules, and 54 608 B rounded up no longer fits three times -- the tier
        //  trace showed 512 of 1 024 workgroups resident, two per CU; 52 KiB is a whole number of every granule up to 4 KiB)
        const int budget = T.ncpair > 0 ? (dense_kib * 1024) / 8 : 40 * 1024 / 8;
        int ycap = budget - off - (guard_on ? 2 : 0);
        const int need1 = end1 - off;  // phase 1 may need more than the constraint data
        if (ycap < need1) ycap = need1;
        if (ycap < 8 * YS) ycap = 8 * YS;
        A.cf.ycap = ycap;
        A.cf.vstage = ((A.dbg & 2) == 0 && T.ncpair > 0 && 3 * d.nmeshvert <= A.cf.ndense * A.nvp + (dcol_alias ? 0 : A.cf.ndense * UHC_WAVE) + ycap) ? F.dense : -1;  // dense, (dcol,) Y are contiguous
        off += ycap;
        end_guard();
        F.rowR = F.rowAref = F.rowB = F.rowF = F.rowDa = F.rowW = F.rowY = F.dsc = F.Y;  // unused by the fast kernel
        F.total = off;
        if (A.marks[2] < 0) {  // the marks follow the layout: up at the capacity, down with a little room to spare
            A.marks[2] = A.cf.ndense > 0 ? A.cf.ndense : UHC_FAST_MAXTWO; A.marks[5] = A.cf.ndense > 0 ? std::max(1, A.cf.ndense - (A.cf.ndense > 8 ? 2 : 1)) : 10;
            A.marks[1] = A.cf.maxcon; A.marks[4] = A.cf.maxcon - std::max(2, A.cf.maxcon / 8);
        }
        b->lds_bytes_fast = (size_t)off * sizeof(double);
        // occupancy experiments (tools/occupancy_sweep.sh): UHC_LDS_PAD_FAST = KiB the fast tier's launch ASKS for -- the layout
        // is unchanged, the workgroup merely occupies more of the CU's 160 KiB, so fewer of them share a CU (64: two, 100: one)
        if (const char* pad = getenv("UHC_LDS_PAD_FAST")) {
            const size_t want = (size_t)atoi(pad) * 1024;
            if (want > b->lds_bytes_fast && want <= 160 * 1024) b->lds_bytes_fast = want;
        }
        const char* env = getenv("UHC_FORCE_GENERAL");
        b->use_fast = !(env && env[0] == '1') && b->lds_bytes_fast <= 160 * 1024;
    }
    // ---- general / large tiers: rows two (four) per lane, Yhat packed row after row.  The general tier is sized for TWO workgroups per CU
    //      (<= 79 KiB): its Yhat storage is what that budget leaves, and an env whose packed rows (or contacts, rows, dense rows) do not
    //      fit goes on to the large tier, which owns a whole CU's LDS and holds maxefc rows of full length.
    auto rows_layout = [&](DevLds& F, TierCap& cp, int maxefc, int maxcon, int maxtwo, int budget_doubles, bool full_y) -> bool {
        common(F, false);
        cp.maxefc = maxefc; cp.maxcon = maxcon; cp.ndense = T.ncpair > 0 ? maxtwo : 0;
        cp.ld_delta = (F.LD - A.lf.LD) * 8;
        F.con = carve(std::max(maxcon * UHC_CON_STRIDE, cp.ndense * UHC_WAVE));
        F.dcol = F.con;  // the dense rows' Delassus columns are built when nothing reads the contacts any more (k_as_general)
        F.rowMisc = carve(maxefc * 2);  // 4 ints per row; the collision pass keeps its candidate-pair list here (256 ints)
        F.ncon_nefc = carve(2 + UHC_MAXTWO);  // ints: truncated flag, nefc, number of two-body rows, spare, their row ids, slot -> lane of the working set
        F.rowY = carve(maxefc / 2 + 1);  // + the end of the last row
        F.rowR = carve(maxefc); F.rowAref = carve(maxefc); F.rowB = carve(maxefc); F.rowF = carve(maxefc); F.rowDa = carve(maxefc); F.rowW = carve(maxefc);
        // MPR walks hull vertices once per support call and lane: staged in LDS they cost an LDS read instead of an L2 round trip.  The
        // dense rows, their scalars and Yhat (contiguous) are not written before the collision pass is over: the vertices borrow them.
        F.dense = carve(cp.ndense * A.nvp);
        F.dsc = carve(cp.ndense * 4);  // (vel, jas, jaw, |Yhat|^2) of every dense row
        F.Y = off;
        const int full = maxefc * YS + 8;
        int ycap = full_y ? full : std::min(budget_doubles - off - (guard_on ? 2 : 0), full);
        if (!full_y && ycap < 64 * 18) return false;  // too little left for rows: not worth a tier of its own
        if (ycap < end1 - off) ycap = end1 - off;     // (the region also holds the dynamics temporaries of phase 1)
        cp.ycap = ycap & ~1;
        cp.vstage = ((A.dbg & 2) == 0 && T.ncpair > 0 && 3 * d.nmeshvert <= cp.ndense * A.nvp + cp.ndense * 4 + cp.ycap) ? F.dense : -1;
        off += cp.ycap;
        end_guard();
        F.total = off;
        return off <= budget_doubles;
    };
    // ---- tier 4 (huge): what the large tier cannot hold (more than 256 rows / 128 contacts / 32 body-body rows) or cannot finish (islands with
    //      more force-carrying rows than a 64-row working set).  Persistent part + contacts + per-row scalars + the nv x nv Hessian of the
    //      primal problem (packed lower triangle) in LDS; the Yhat rows themselves -- chain rows packed, body-body rows as dense nv-vectors --
    //      in HBM (KernelArgs::gY / gD, one slice per env, L2-resident while the env's workgroup runs).  Rows: the largest multiple of 128 up
    //      to UHC_HUGE_MAXEFC that the LDS holds (nv 75: 1024; nv 99, the humanoid among four boxes: 768).
    auto huge_layout = [&](DevLds& F, TierCap& cp, int maxefc, int maxcon) -> bool {
        common(F, false);
        cp.maxefc = maxefc; cp.maxcon = maxcon; cp.ndense = T.ncpair > 0 ? UHC_HUGE_MAXTWO : 0;
        cp.ld_delta = (F.LD - A.lf.LD) * 8;
        // (after the rows are built the contacts' storage serves the Newton iteration: the dof-chain table, the run of chain rows the wave is adding
        //  to the Hessian, the dense group's D y, the pair table -- uhc_primal.h)
        const int scratch4 = primal_scratch(nv, YS).total;  // uhc_device.h: table | per-wave runs of rows | dense group | pair tables | run coefficients | mailbox
        F.con = carve(std::max(cp.maxcon * UHC_CON_STRIDE, scratch4));
        F.dcol = F.con;
        F.rowMisc = carve(std::max(maxefc * 2, 128));  // (the collision pass keeps its candidate-pair list here: 256 ints)
        F.ncon_nefc = carve(2 + (cp.ndense + 1) / 2 + 1);  // ints: truncated flag, nefc, number of two-body rows, spare, their row ids
        F.rowY = carve(maxefc / 2 + 1);
        F.rowR = carve(maxefc); F.rowAref = carve(maxefc); F.rowB = carve(maxefc); F.rowF = carve(maxefc); F.rowDa = carve(maxefc); F.rowW = carve(maxefc);
        F.dsc = carve(std::max(cp.ndense * 4, 2));
        F.H = carve((nv * (nv + 1)) / 2);
        F.Y = F.dense = F.H;  // (never addressed in this tier: the rows live in gY / gD)
        cp.ycap = maxefc * YS + 8;
        // MPR's hull vertices: staged where the Hessian will be (nothing of it exists during the collision pass) when they fit
        cp.vstage = ((A.dbg & 2) == 0 && T.ncpair > 0 && 3 * d.nmeshvert <= (nv * (nv + 1)) / 2) ? F.H : -1;
        if (off < end1) off = end1;  // (the region also holds the dynamics temporaries of phase 1)
        end_guard();
        F.total = off;
        return off <= 160 * 1024 / 8;
    };
    {
        const char* tv = getenv("UHC_TIERS");
        A.last_tier = (tv && tv[0] == '2') ? 2 : 3;
        if (A.last_tier == 3) {  // the large tier: as many rows (<= 256) as a CU's 160 KiB hold at full length
            bool ok = false;
            gt = 2;
            for (int me = UHC_BIG_MAXEFC; me > UHC_GEN_MAXEFC && !ok; me -= 32) {
                ok = rows_layout(A.lh, A.ch, me, me / 2, UHC_BIG_MAXTWO, 160 * 1024 / 8, true);
            }
            if (ok) b->lds_bytes_big = (size_t)A.lh.total * sizeof(double);
            else A.last_tier = 2;
        }
        const int two_per_cu = 79 * 1024 / 8;
        gt = 1;
        bool ok = A.last_tier == 3 && rows_layout(A.l, A.cg, UHC_GEN_MAXEFC, UHC_GEN_MAXCON, UHC_GEN_MAXTWO, two_per_cu, false);
        if (!ok) {  // the last tier must hold every row at full length: a whole CU's LDS if need be (32 dense slots as before)
            if (!rows_layout(A.l, A.cg, UHC_GEN_MAXEFC, UHC_GEN_MAXCON, A.last_tier == 2 ? UHC_MAXTWO : UHC_GEN_MAXTWO, 160 * 1024 / 8, true)) {
                delete b; return fail("uhc_batch_create: model needs %zu B of LDS per env (> 160 KiB)", (size_t)A.l.total * 8);
            }
        }
        b->lds_bytes = (size_t)A.l.total * sizeof(double);
        A.lx = A.lh; A.cx = A.ch; A.gY = A.gD = nullptr; A.gy_stride = A.gd_stride = 0;
        if (A.last_tier == 3 && !(tv && tv[0] == '3')) {  // (UHC_TIERS=3: the three-tier chain of rounds 3-4, windows and all)
            bool ok4 = false;
            gt = 3;
            // Rows and contacts share what the LDS has left beside the Hessian (128 rows = 8.7 KB = 45 contacts).  A contact brings at most four rows, a
            // body-body contact one; the extremes seen in 2.5 M env-steps of the ball-joint rollouts are 504 rows with 192+ contacts (2.6 rows per
            // contact) -- and 4 of those env-steps met a fixed 192-contact cap with 768 rows allotted.  So: for every row count that fits (multiples of
            // 128), the most contacts that fit beside it (>= UHC_HUGE_MAXCON, <= UHC_HUGE_MAXCON_MOST); the pair with the largest
            // min(rows / 2.75, contacts) wins (nv 99, the humanoid among four boxes: 640 rows / 237 contacts instead of 768 / 192).
            int best_me = 0, best_mc = 0;
            double best = -1.0;
            for (int me = UHC_HUGE_MAXEFC; me >= 384; me -= 128) {
                if (!huge_layout(A.lx, A.cx, me, UHC_HUGE_MAXCON)) continue;
                int mc = UHC_HUGE_MAXCON_MOST;
                while (mc > UHC_HUGE_MAXCON && !huge_layout(A.lx, A.cx, me, mc)) mc -= 8;
                const double score = std::min(me / 2.75, (double)mc);
                if (score > best) { best = score; best_me = me; best_mc = mc; }
            }
            ok4 = best_me > 0 && huge_layout(A.lx, A.cx, best_me, best_mc);
            if (ok4 && (A.dbg & 64)) fprintf(stderr, "uhc tier 4: %d rows / %d contacts / %d body-body rows, %d B of LDS\n", A.cx.maxefc, A.cx.maxcon, A.cx.ndense, A.lx.total * 8);
            if (ok4) {
                A.last_tier = 4;
                b->lds_bytes_big = std::max(b->lds_bytes_big, (size_t)A.lx.total * sizeof(double));
                A.gy_stride = A.cx.ycap;
                A.gd_stride = std::max(A.cx.ndense, 1) * A.nvp;
            } else { A.lx = A.lh; A.cx = A.ch; }
        }
    }
    A.guard_tab = nullptr; A.guard_hits = nullptr;
    if (guard_on) {
        std::vector<int> tab(4 * 64, 0);
        if (guard_selftest) g_persist[0].push_back(A.lf.qpos);
        for (int t = 0; t < 4; t++) {
            if (g_persist[t].size() > 30 || g_phase2[t].size() > 32) { delete b; return fail("uhc_batch_create: UHC_GUARD_LDS: more guard words than the table holds"); }
            tab[64 * t] = (int)g_persist[t].size(); tab[64 * t + 1] = (int)g_phase2[t].size();
            for (size_t k = 0; k < g_persist[t].size(); k++) tab[64 * t + 2 + k] = g_persist[t][k];
            for (size_t k = 0; k < g_phase2[t].size(); k++) tab[64 * t + 32 + k] = g_phase2[t][k];
        }
        int* d_tab = nullptr;
        TRY(dalloc(b, tab.size(), &d_tab)); TRY(dalloc(b, 4, &b->d_guard_hits));
        HIP_OK(hipMemcpy(d_tab, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
        HIP_OK(hipMemset(b->d_guard_hits, 0, 4 * sizeof(int)));
        A.guard_tab = d_tab; A.guard_hits = b->d_guard_hits;
        fprintf(stderr, "uhc guard: LDS guard words on -- fast %zu + %zu, general %zu + %zu, large %zu + %zu, tier 4 %zu + %zu (persistent + constraint phase)\n", g_persist[0].size(), g_phase2[0].size(),
                g_persist[1].size(), g_phase2[1].size(), g_persist[2].size(), g_phase2[2].size(), g_persist[3].size(), g_phase2[3].size());
    }
    // ---- static schedules for the factorisation and the triangular solves (see DevTopo).  Addresses are LDS byte
    //      addresses of the FAST layout's LD buffer (the general kernel adds its own LD offset, KernelArgs::ld_delta).
    {
        // substitution tables: nv-1 steps, padded with no-op steps (zero-slot addresses) to a multiple of 8 plus one block of look-ahead
        const size_t sol_steps = (size_t)((std::max(nv - 1, 1) + 7) / 8 * 8 + 8);
        std::vector<unsigned int> fac_prog, sol_back(sol_steps * 64, 0), sol_fwd(sol_steps * 64, 0);
        const unsigned ldb = (unsigned)A.lf.LD * 8u;
        auto adr = [&](int rel) { return ldb + 8u * (unsigned)rel; };
        const int ZERO = T.nM, DUMP = T.nM + 1;
        if (adr(DUMP) + 8 > 65536u || (unsigned)A.l.LD * 8u + 8u * (T.nM + 2) > 65536u) { delete b; return fail("uhc_batch_create: LD buffer beyond the 16-bit schedule addresses"); }
        // factorisation program: a flat list of groups, one 24-byte record per lane and group (DevTopo::fac_prog).  A group holds three
        // 64-lane slots of updates of ONE elimination step k,  LD[row(anc_a) + t] -= (LD[kk + a] / D_k) * LD[kk + a + t],  the address of
        // D_k, and -- in the step's last group -- the lane's entry of row k to normalise (LD[kk + 1 + lane] /= D_k).  Steps need no
        // boundary handling in the kernel: the LDS queue of a wave is in order.
        int ngroups = 0;
        for (int k = nv - 1; k >= 1; k--) {
            const int dk = dof_depth[k], kk = d.dof_madr[k];
            if (!dk) continue;
            std::vector<unsigned> E;  // (af | ar << 16, ao) per entry
            int anc = k;
            for (int a = 1; a <= dk; a++) {
                anc = d.dof_parentid[anc];
                for (int t = 0; t <= dk - a; t++) { E.push_back(adr(kk + a) | (adr(kk + a + t) << 16)); E.push_back(adr(d.dof_madr[anc] + t)); }
            }
            constexpr int G = 3;  // update slots per group (uhc_physics.hip k_factor): 99 groups for the SMPL tree (129 with 2, 88 with 4 but 36 % more padded slots)
            while ((E.size() / 2) % (64 * G)) { E.push_back(adr(ZERO) | (adr(ZERO) << 16)); E.push_back(adr(DUMP)); }
            const int ng = (int)(E.size() / 2 / (64 * G));
            for (int gi = 0; gi < ng; gi++)
                for (int l = 0; l < 64; l++) {
                    size_t e[G];
                    for (int q = 0; q < G; q++) e[q] = ((size_t)gi * 64 * G + 64 * q + l) * 2;
                    const unsigned nr = (gi == ng - 1 && l < dk) ? adr(kk + 1 + l) : adr(ZERO);
                    for (int q = 0; q < G; q++) fac_prog.push_back(E[e[q]]);
                    fac_prog.push_back(E[e[0] + 1] | (E[e[1] + 1] << 16));
                    fac_prog.push_back(E[e[2] + 1] | (nr << 16));
                    fac_prog.push_back(adr(kk));
                }
            ngroups += ng;
        }
        T.fac_nslot = ngroups;
        for (int q = 0; q < 2 * 64; q++) {  // two groups of look-ahead slack
            for (int w = 0; w < 3; w++) fac_prog.push_back(adr(ZERO) | (adr(ZERO) << 16));
            fac_prog.push_back(adr(DUMP) | (adr(DUMP) << 16)); fac_prog.push_back(adr(DUMP) | (adr(ZERO) << 16)); fac_prog.push_back(adr(ZERO));
        }
        auto entry = [&](int i, int j) -> unsigned {  // address of L[i][j] if j is a proper ancestor of i, else the zero slot
            if (i >= nv || j >= nv || j >= i || dof_depth[j] >= dof_depth[i]) return adr(ZERO);
            return dof_anc[(size_t)i * YS + dof_depth[j]] == j ? adr(d.dof_madr[i] + dof_depth[i] - dof_depth[j]) : adr(ZERO);
        };
        for (size_t q = 0; q < sol_steps * 64; q++) sol_back[q] = sol_fwd[q] = adr(ZERO) | (adr(ZERO) << 16);
        for (int s2 = 0; s2 < nv - 1; s2++)
            for (int l = 0; l < 64; l++) {
                const int i = nv - 1 - s2;
                sol_back[(size_t)s2 * 64 + l] = entry(i, l) | (entry(i, l + 64) << 16);
                sol_fwd[(size_t)s2 * 64 + l] = entry(l, s2) | (entry(l + 64, s2) << 16);
            }
        std::vector<unsigned int> chain((size_t)nv * 32, adr(0) << 16);
        for (int i = 0; i < nv; i++)
            for (int q = 0; q <= dof_depth[i]; q++) {
                const int a = dof_anc[(size_t)i * YS + q];
                chain[(size_t)i * 32 + q] = (unsigned)a | (adr(d.dof_madr[a]) << 16);
            }
        TRY(upload(b, chain, &T.chain));
        if (nv > 128) { delete b; return fail("uhc_batch_create: nv %d > 128 unsupported", nv); }
        // actuation is gathered per dof (lane = dof): up to UHC_DOF_MAXACT motors drive the joint of a dof (three on a ball joint, one per gear axis)
        std::vector<int> dof_act((size_t)nv * UHC_DOF_MAXACT, -1);
        for (int a = 0; a < d.nu; a++) {
            const int j = d.dof_jntid[d.actuator_dofid[a]], nd = d.jnt_type[j] == UHC_JNT_BALL ? 3 : d.jnt_type[j] == UHC_JNT_FREE ? 6 : 1;
            if (nd == 6) { delete b; return fail("uhc_batch_create: motors on free joints are not supported"); }
            for (int k = 0; k < nd; k++) {
                int q = 0;
                while (q < UHC_DOF_MAXACT && dof_act[(size_t)(d.jnt_dofadr[j] + k) * UHC_DOF_MAXACT + q] >= 0) q++;
                if (q == UHC_DOF_MAXACT) { delete b; return fail("uhc_batch_create: more than %d motors on one joint", UHC_DOF_MAXACT); }
                dof_act[(size_t)(d.jnt_dofadr[j] + k) * UHC_DOF_MAXACT + q] = a;
            }
        }
        TRY(upload(b, dof_act, &T.dof_act));
        TRY(upload(b, fac_prog, &T.fac_prog)); TRY(upload(b, sol_back, &T.sol_back)); TRY(upload(b, sol_fwd, &T.sol_fwd));
    }
    HIP_OK(uhc_set_lds_limit(b->lds_bytes, b->lds_bytes_fast, b->lds_bytes_big));

    // ---- state
    DevState& S = A.s;
    const size_t E = n_env;
    TRY(dalloc(b, E * d.nq, &S.qpos)); TRY(dalloc(b, E * nv, &S.qvel)); TRY(dalloc(b, E * nv, &S.qacc)); TRY(dalloc(b, E * nv, &S.qacc_ws));
    TRY(dalloc(b, E * 3 * nb, &S.xpos)); TRY(dalloc(b, E * 4 * nb, &S.xquat)); TRY(dalloc(b, E * 3 * nb, &S.xipos));
    TRY(dalloc(b, 4, &S.path_stats));
    TRY(dalloc(b, E * T.nM, &S.qM)); TRY(dalloc(b, 5 * E, &S.redo)); S.pend2 = S.redo + E; S.pend3 = S.redo + 2 * E; S.resume = S.redo + 3 * E; S.why = S.redo + 4 * E; TRY(dalloc(b, 1, &S.q_abort));
    TRY(dalloc(b, E, &S.tier)); TRY(dalloc(b, E, &b->tier_now)); S.tier_now = b->tier_now; TRY(dalloc(b, E, &S.cost));
    if (!(A.dbg & 8)) TRY(dalloc(b, E, &b->d_order));  // (UHC_DEBUG bit 3: the fast tier launches in env order)
    if (const char* q = getenv("UHC_Q2_DIV")) b->q2_div = std::max(1, atoi(q));
    if (const char* q = getenv("UHC_Q2_WAIT")) b->q2_wait_min = std::max(1, atoi(q));
    if (const char* q = getenv("UHC_Q2_MAX")) b->q2_max = std::max(16, atoi(q));
    if (const char* q = getenv("UHC_Q3_MAX")) b->q3_max = std::max(2, atoi(q));
    if (const char* q = getenv("UHC_Q4_MAX")) b->q4_max = std::max(0, atoi(q));
    A.t4_rows = 0;
    if (const char* q = getenv("UHC_T4_ROWS")) A.t4_rows = std::max(0, atoi(q));
    TRY(dalloc(b, 3 * E, &b->d_lists)); TRY(dalloc(b, 8, &b->d_counts)); TRY(dalloc(b, 8, &b->d_cursors)); TRY(dalloc(b, 8, &b->d_fin));
    { std::vector<int> one(E, 1); HIP_OK(hipMemcpy(S.tier, one.data(), E * sizeof(int), hipMemcpyHostToDevice)); } TRY(dalloc(b, E, &S.fresh)); TRY(dalloc(b, E * 40, &S.prof)); TRY(dalloc(b, E * nv, &S.bias)); TRY(dalloc(b, E * d.nu, &S.ctrl));
    TRY(dalloc(b, E * nv, &S.applied));
    if (A.c.rfc_mode == 2) { TRY(dalloc(b, E * 6 * nv, &S.cdof)); TRY(dalloc(b, E * 3 * nb, &S.rootcom)); }
    TRY(dalloc(b, E, &S.ncon)); TRY(dalloc(b, E, &S.nefc)); TRY(dalloc(b, E, &S.fail)); TRY(dalloc(b, E, &S.solver_iter));
    TRY(dalloc(b, E, &S.overflow));
    if (A.last_tier == 4) { TRY(dalloc(b, E * (size_t)A.gy_stride, &A.gY)); TRY(dalloc(b, E * (size_t)A.gd_stride, &A.gD)); }
    TRY(dalloc(b, E, &b->reset_mask));
    A.n_env = n_env;
    // qpos <- qpos0 of each env's model
    {
        std::vector<double> q0(E * d.nq);
        for (size_t e = 0; e < E; e++) {
            const UhcModelDesc& md = models[h_env_model ? h_env_model[e] : 0]->d;
            memcpy(&q0[e * d.nq], md.qpos0, d.nq * sizeof(double));
        }
        HIP_OK(hipMemcpy(S.qpos, q0.data(), q0.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    void* fp[] = {S.qpos, S.qvel, S.xpos, S.xquat, S.xipos, S.qM, S.bias, S.qacc, S.ctrl, S.ncon, S.nefc, S.fail,
                  S.solver_iter, S.applied, S.overflow, S.prof, S.redo, S.tier, S.why};
    int64_t fc[] = {(int64_t)E * d.nq, (int64_t)E * nv, (int64_t)E * 3 * nb, (int64_t)E * 4 * nb, (int64_t)E * 3 * nb,
                    (int64_t)E * T.nM, (int64_t)E * nv, (int64_t)E * nv, (int64_t)E * d.nu, (int64_t)E, (int64_t)E, (int64_t)E,
                    (int64_t)E, (int64_t)E * nv, (int64_t)E, (int64_t)E * 40, (int64_t)E, (int64_t)E, (int64_t)E};
    for (int k = 0; k < 19; k++) { b->field_ptr[k] = fp[k]; b->field_count[k] = fc[k]; }
    HIP_OK(hipStreamCreateWithFlags(&b->own_stream, hipStreamNonBlocking));
    b->stream = b->own_stream;
    *out = b;
    return 0;
}

static void guard_report(UhcBatch* b) {  // UHC_GUARD_LDS=1: what the kernels found (once per new finding)
    if (!b->d_guard_hits) return;
    int h[4] = {0, 0, 0, 0};
    if (hipMemcpy(h, b->d_guard_hits, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return;
    if (h[0] > b->guard_reported) {
        b->guard_reported = h[0];
        fprintf(stderr, "uhc guard: %d LDS guard words OVERWRITTEN so far; the first: tier %d, %s region %d (offset %d doubles), env %d\n", h[0], h[1] >> 16,
                ((h[1] >> 8) & 0xff) ? "constraint-phase" : "persistent", h[1] & 0xff, h[3], h[2]);
    }
}
extern "C" void uhc_batch_free(UhcBatch* b) {
    if (!b) return;
    hipSetDevice(b->device);
    hipDeviceSynchronize();
    guard_report(b);
    if (!b->fences.empty()) {
        int bad = 0;
        unsigned char h[512];
        for (size_t k = 0; k < b->fences.size(); k++) {
            const auto& f = b->fences[k];
            bool hit = false;
            if (hipMemcpy(h, f.first, 256, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(h + 256, f.first + 256 + f.second, 256, hipMemcpyDeviceToHost) != hipSuccess) continue;
            for (int i = 0; i < 512; i++) hit = hit || h[i] != 0xA5;
            if (hit) { if (!bad) fprintf(stderr, "uhc guard: HBM array %zu (%zu bytes) has an OVERWRITTEN fence\n", k, f.second); bad++; }
        }
        fprintf(stderr, "uhc guard: %zu fenced HBM arrays checked, %d with an overwritten fence\n", b->fences.size(), bad);
    }
    if (b->d_guard_hits) fprintf(stderr, "uhc guard: batch of %d envs freed, %d guard words overwritten in its lifetime\n", b->n_env, b->guard_reported);
    for (void* p : b->allocs) hipFree(p);
    for (auto& ev : b->ev_used) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
    for (auto& ev : b->ev_free) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
    for (hipEvent_t e : {b->ev_fork, b->ev_side1, b->ev_side2, b->ev_side3}) if (e) hipEventDestroy(e);
    for (hipEvent_t e : b->cnt_ev) if (e) hipEventDestroy(e);
    if (b->h_counts) hipHostFree(b->h_counts);
    if (b->side_stream) hipStreamDestroy(b->side_stream);
    if (b->side_stream3) hipStreamDestroy(b->side_stream3);
    if (b->side_stream4) hipStreamDestroy(b->side_stream4);
    if (b->own_stream) hipStreamDestroy(b->own_stream);
    delete b;
}
extern "C" int32_t uhc_batch_set_stream(UhcBatch* b, void* s) {
    if (!b) return fail("uhc_batch_set_stream: null batch");
    b->stream = (hipStream_t)s;  // NULL is the device's null (default) stream
    return 0;
}
extern "C" int32_t uhc_batch_sync(UhcBatch* b) {
    if (!b) return fail("uhc_batch_sync: null batch");
    HIP_OK(hipStreamSynchronize(b->stream));
    guard_report(b);
    return 0;
}
extern "C" int32_t uhc_batch_set_rfc_scale(UhcBatch* b, double s) {
    if (!b) return fail("uhc_batch_set_rfc_scale: null batch");
    b->A.c.rfc_scale = s;
    return 0;
}
extern "C" int32_t uhc_batch_set_kernel_path(UhcBatch* b, int32_t mode) {
    if (!b) return fail("uhc_batch_set_kernel_path: null batch");
    if (mode < 0 || mode > 2) return fail("uhc_batch_set_kernel_path: mode %d (0 fast then general, 1 general only, 2 adaptive)", mode);
    b->general_only = mode == 1;
    b->path_mode = mode;
    if (mode == 2 && !b->side_stream) {
        HIP_OK(hipSetDevice(b->device));
        // The consumers WAIT for launches of the batch's own stream, so they must never sit behind them in one hardware queue (streams beyond
        // the runtime's pool of hardware queues share one and then run in order).  Streams of another priority get their queues from
        // another pool.  The two side streams may still share a queue with each other: the general tier's consumers are launched first,
        // the large tier's (which wait for them) second, so that in-order execution is merely slower.
        int least = 0, greatest = 0;
        HIP_OK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_OK(hipStreamCreateWithPriority(&b->side_stream, hipStreamNonBlocking, greatest));
        // (the large tier's stream at the LOWEST priority when the device has three levels: a pool of its own, so that its consumers can be
        //  launched BEFORE the general tier's -- they need whole CUs, which they only find while nothing else is resident -- without ever
        //  sharing a queue with the launch they wait for)
        HIP_OK(hipStreamCreateWithPriority(&b->side_stream3, hipStreamNonBlocking, (least != greatest && least != 0) ? least : greatest));
        b->large_first = least != greatest && least != 0;
        { hipDeviceProp_t pr; HIP_OK(hipGetDeviceProperties(&pr, b->device)); b->n_cu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
        HIP_OK(hipHostMalloc((void**)&b->h_counts, sizeof(int) * 8 * 8, hipHostMallocDefault));
        memset(b->h_counts, 0, sizeof(int) * 8 * 8);
        for (int k = 0; k < 8; k++) HIP_OK(hipEventCreateWithFlags(&b->cnt_ev[k], hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&b->ev_side1, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&b->ev_side2, hipEventDisableTiming));
        // tier 4's own launch (envs whose last step ended there): whole CUs, like the large tier's -- same priority class, launched first of all
        HIP_OK(hipStreamCreateWithPriority(&b->side_stream4, hipStreamNonBlocking, (least != greatest && least != 0) ? least : greatest));
        HIP_OK(hipEventCreateWithFlags(&b->ev_side3, hipEventDisableTiming));
    }
    return 0;
}
extern "C" int32_t uhc_batch_set_solver(UhcBatch* b, int32_t solver, int32_t iterations) {
    if (!b || (solver != 0 && solver != 1)) return fail("uhc_batch_set_solver: solver must be 0 (sweeps) or 1 (active set)");
    b->A.t.solver = solver;
    if (iterations > 0) b->A.t.iterations = iterations;
    return 0;
}
extern "C" int32_t uhc_batch_field(UhcBatch* b, int32_t f, void** p, int64_t* n) {
    if (!b || f < 0 || f > 18 || !b->field_ptr[f]) return fail("uhc_batch_field: unknown field %d", f);
    if (p) *p = b->field_ptr[f];
    if (n) *n = b->field_count[f];
    return 0;
}
// fast kernel on every (active) env, then the general kernel on the envs that raised redo
static int launch(UhcBatch* b, int mode, const double* d_action, const double* d_tbase, const int* d_active) {
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    const bool timed = b->timing && mode == 0;  // HIP events around the kernel that does the work of a control step
    if (timed) {
        if (b->ev_free.empty()) { HIP_OK(hipEventCreate(&ev.first)); HIP_OK(hipEventCreate(&ev.second)); }
        else { ev = b->ev_free.back(); b->ev_free.pop_back(); }
    }
    const bool general = b->general_only;
    const bool big = b->A.last_tier >= 3;  // (tier 4 has no chained launch of its own: the large tier's workgroups go on with it; under sticky tiers it has queue consumers, below)
    // tier chain: every tier works on the envs the previous one flagged (redo / redo2) and left untouched
    HIP_OK(hipMemsetAsync(b->A.s.redo, 0, sizeof(int) * b->n_env * 5, b->stream));  // redo (the step's UHC_F_REDO words), pend2, pend3, resume, why: one allocation
    // (inside a stream capture the sticky launch cannot be used: it sizes its consumer launches from counts the host reads between steps
    //  -- event queries and a wait that are not allowed while capturing, and a replay would repeat the capture step's sizes anyway.  A
    //  captured step takes the plain tier chain, which computes the same step.)
    bool capturing = false;
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(b->stream, &cs) == hipSuccess) capturing = cs == hipStreamCaptureStatusActive;
        else (void)hipGetLastError();
    }
    if (mode == 0 && b->path_mode == 2 && b->use_fast && !general && !capturing) {
        // sticky tiers: an env starts in the tier that computed its last step.  The general / large tiers' own envs run on a side stream
        // BESIDE the fast tier (their launches last several times longer per env; in a chain behind it the step would wait for them);
        // only the envs a tier hands on this very step go through the chain.  All launches filter on one snapshot of the tier table.
        KernelArgs K = b->A;
        // tier 4's own consumers: when the newest counts seen say that envs went through tier 4 (counts[7]: hand-ons of the large tier + envs that start there),
        // a few persistent workgroups wait on a queue of their own (d_lists + 2 n_env) beside everything else, and the large tier's consumers append what
        // they find too big instead of leaving it for the chained launch at the very end of the step -- where a 10 ms env-step of one env used to be added
        // to every step in which any env needed tier 4 (configs[4]: 64 k -> 42 k env-steps/s when tier 4 came in).  Envs that START in tier 4
        // (UHC_DEBUG bit 12) are put at the head of that queue by the list kernel.
        int est4 = 0, est2_then = 0;
        if (b->A.last_tier == 4 && b->q4_max > 0)
            for (long long k = b->cnt_step - 1; k >= 0 && k > b->cnt_step - 8; k--)
                if (hipEventQuery(b->cnt_ev[k % 8]) == hipSuccess) { est4 = b->h_counts[8 * (k % 8) + 7]; est2_then = b->h_counts[8 * (k % 8) + 2]; break; }
        (void)hipGetLastError();
        const bool launch4 = b->A.last_tier == 4 && est4 > 0 && est2_then > 0 && big && !b->queues_off;  // (the list kernel's view; the consumers need the large tier's beside them: q4 below)
        K.cnt4 = b->d_counts + 7;
        HIP_OK(uhc_launch_tier_lists(b->A.s.tier, d_active, b->n_env, b->tier_now, b->d_lists, b->d_counts, b->d_cursors, b->d_fin, b->A.s.cost, b->A.s.fresh, b->d_order,
                                     launch4 ? 1 : 0, b->A.s.pend3, b->stream));
        HIP_OK(hipEventRecord(b->ev_fork, b->stream));
        // how long the queues got is known on the host with a lag (asynchronous copies of the final counts, never waited for): the newest
        // copy that has landed sizes this step's consumer launches.  While the general tier's queue was empty when last seen there are no
        // consumers at all: an env the fast tier hands on is flagged and goes through the chained launches like in mode 0.
        // (the host may not run more than two steps ahead of the device here: a launch sized for a queue of five that meets nine hundred
        //  envs works them off five at a time)
        if (b->cnt_step >= 2) HIP_OK(hipEventSynchronize(b->cnt_ev[(b->cnt_step - 2) % 8]));
        int est2 = 0, est3 = 0, handed2 = 0;  // queue lengths at the end of the newest step seen, and how many of the general tier's came in during the step
        for (long long k = b->cnt_step - 1; k >= 0 && k > b->cnt_step - 8; k--)
            if (hipEventQuery(b->cnt_ev[k % 8]) == hipSuccess) {
                const int* hc = b->h_counts + 8 * (k % 8);
                est2 = hc[2]; est3 = hc[3]; handed2 = std::max(0, hc[2] - hc[4]);
                if (b->A.dbg & 64) fprintf(stderr, "uhc step %lld: queues %d / %d envs (%d handed on), gate waited %.1f us, gave up %d, queues_off %d\n", k, est2, est3, handed2, 0.01 * hc[1], hc[0], (int)b->queues_off);
                if (hc[0] > b->aborts_seen) {  // a consumer gave up waiting: its producers did not run beside it (or too slowly).  No waiting
                    b->aborts_seen = hc[0];     // consumers for the next 32 steps; after the third time, for good
                    b->queues_off_until = ++b->abort_events >= 3 ? (long long)1 << 62 : b->cnt_step + 32;
                }
                b->queues_off = b->cnt_step < b->queues_off_until;
                break;
            }
        // Three regimes.  No env in the general tier when last seen: no side launches, plain chain.  Up to three quarters of the batch
        // there: its launch is a CONSUMER that also waits for what the fast tier hands on while both run -- and is kept small enough that
        // the fast tier's workgroups always find LDS beside it (a consumer that holds all LDS while it waits for a launch that cannot
        // start would only end by its time-out).  The majority there: the fast tier is the side show; the general tier's launch takes
        // its list at full width and does not wait, what the fast tier hands on goes through the chained launch.
        const bool queues = est2 > 0;
        const bool waiting = queues && est2 <= (3 * b->n_env) / 4 && !b->queues_off;
        const bool q3 = queues && big;  // (a large-tier consumer waits beside the general tier's launch whenever there is one: what that hands on is rare and slow)
        // (one waiting consumer per env expected in the queue: its own envs are done within one general-tier env-step and the consumers are
        //  free when the fast tier hands envs on.  Half as many -- more LDS for the fast tier, two envs in a row per consumer -- was
        //  measured on the self-colliding rollout: 59 k env-steps/s against 66 k)
        // The large tier's share of the chip: each of its workgroups holds a whole CU, two of the general tier's fit one.  With g3 CUs for
        // the large tier both queues take equally long when est3 / g3 = est2 / (2 (n_cu - g3)) (their env-steps last about as long); never
        // more consumers than envs expected, never fewer than 64 when there are that many envs.  (A fixed 64 was measured on the
        // ball_objects scene late in its cycle: 370 envs in the large tier's queue, six in a row per consumer, the step 61 ms of which the
        // general tier's launch took the first 20 -- tools/tier_trace.py --workload ball_objects.)
        const int share3 = est3 > 0 ? (int)((2ll * b->n_cu * est3) / std::max(1, est2 + 2 * est3)) : 0;
        // (beside a fast tier that still has most of the envs -- `waiting` -- the consumers are kept to a third of the chip: at most 32 CUs
        //  for the large tier, 256 workgroups of the general tier, 64 waiting spares.  Sized for their queues alone they left the fast tier 28
        //  CUs in the ball-joint rollout's first steps, and consumers that waited for it ran into their time-out.)
        const int grid3 = waiting ? std::min(est3 + est3 / 4 + 2, b->q3_max) : std::min(est3 + est3 / 4 + 2, std::max(64, std::min(share3, (3 * b->n_cu) / 4)));
        const int room2 = 2 * (b->n_cu - (b->large_first && q3 ? std::min(grid3, std::max(est3, 1)) : 0));  // general-tier workgroups beside the large tier's
        // (the cap grows with the general tier's share of the step's LDS-time: x consumers and s1 fast-tier workgroups on each of the
        //  remaining n_cu - x / 2 CUs take equally long when x = est2 s1 n_cu / (est1 + est2 s1 / 2) -- env-steps of the two tiers last
        //  about as long.  7/8 of that, never below UHC_Q2_MAX, never above 3/4 of the chip: 300 for configs[4] at 1024 envs, where 256 / 320 /
        //  384 consumers were measured at 65.5 / 66.5 k env-steps/s / worse.)
        const int s1 = std::max(1, std::min(4, (int)(160 * 1024 / std::max<size_t>(b->lds_bytes_fast, 1))));
        const int est1 = std::max(0, b->n_env - est2 - est3);
        const int bal = (int)(((long long)est2 * s1 * b->n_cu) / std::max(1, est1 + (est2 * s1) / 2));
#ifdef UHC_EXPERIMENTS
        const bool fixed_cap2 = (b->A.dbg & 2048) != 0;  // (measurement switch: a fixed cap UHC_Q2_MAX on the general tier's consumers)
#else
        const bool fixed_cap2 = false;
#endif
        const int cap2 = fixed_cap2 ? b->q2_max : std::max(b->q2_max, std::min((7 * bal) / 8, (3 * b->n_cu) / 2));
        const int grid2 = waiting ? std::min(est2 / b->q2_div + 8, cap2) : std::min(est2 + est2 / 4 + 8, std::min(b->n_env, std::max(64, room2)));
        K.sticky_mask = (queues ? 4 : 0) | (q3 ? 8 : 0) | (launch4 ? 16 : 0);
        const bool q4 = launch4 && q3;
        const int grid4 = std::min(b->q4_max, est4 + est4 / 2 + 2);
        if (q4) {  // (first of the side launches: a whole CU's LDS each, only to be had before the fast tier's launch has filled the chip)
            HIP_OK(hipStreamWaitEvent(b->side_stream4, b->ev_fork, 0));
            K.tier_want = 0; K.list = b->d_lists + 2 * b->n_env; K.list_count = b->d_counts + 6; K.list_cursor = b->d_cursors + 4;
            K.grid = grid4; K.n_wait = grid4; K.spares = nullptr; K.started = nullptr;
            K.prod_fin = b->d_fin + 5; K.prod_total = grid3;  // the large tier's consumers
            K.fin = nullptr; K.q_next = nullptr; K.q_next_count = nullptr;
            if (b->A.dbg & 64) fprintf(stderr, "uhc step %lld: %d tier-4 consumers (%d env-steps went through tier 4 when last seen)\n", b->cnt_step, grid4, est4);
            HIP_OK(uhc_launch_step(mode, 4, &K, d_action, d_tbase, nullptr, b->lds_bytes_big, b->side_stream4));
            HIP_OK(hipEventRecord(b->ev_side3, b->side_stream4));
        }
        auto launch_large = [&]() -> int {
            HIP_OK(hipStreamWaitEvent(b->side_stream3, b->ev_fork, 0));
            K.tier_want = 0; K.list = b->d_lists + b->n_env; K.list_count = b->d_counts + 3; K.list_cursor = b->d_cursors + 3;
            K.grid = grid3; K.n_wait = grid3; K.spares = nullptr; K.started = b->d_fin + 4;
            K.prod_fin = b->d_fin + 2; K.prod_total = grid2;  // the general tier's workgroups: they never wait for this launch
            K.fin = q4 ? b->d_fin + 5 : nullptr; K.q_next = q4 ? b->d_lists + 2 * b->n_env : nullptr; K.q_next_count = q4 ? b->d_counts + 6 : nullptr;
            HIP_OK(uhc_launch_step(mode, 3, &K, d_action, d_tbase, nullptr, b->lds_bytes_big, b->side_stream3));
            HIP_OK(hipEventRecord(b->ev_side2, b->side_stream3));
            return 0;
        };
        // (the large tier's consumers first where their stream has a queue pool of its own: whole CUs are only free while nothing else is
        //  resident, so the general tier's launch waits behind a gate until they have reported in)
        if (q3 && b->large_first) { if (launch_large()) return -1; }
        if (queues) {
            HIP_OK(hipStreamWaitEvent(b->side_stream, b->ev_fork, 0));
            if (q3 && b->large_first && !(b->A.dbg & 32)) HIP_OK(uhc_launch_gate(b->d_fin + 4, grid3, nullptr, nullptr, b->side_stream));
            K.tier_want = 0; K.list = b->d_lists; K.list_count = b->d_counts + 2; K.list_cursor = b->d_cursors + 2;
            K.grid = grid2;
            K.prod_fin = waiting ? b->d_fin + 1 : nullptr; K.prod_total = b->n_env;  // every workgroup of the fast tier's launch below
            K.fin = b->d_fin + 2; K.started = waiting ? b->d_fin + 3 : nullptr;
            // (seats for twice the hand-ons last seen: when the queue shrinks from step to step -- after a restart of many envs -- the launch
            //  is sized for a queue that is no longer there, and idle consumers that stay hold LDS the fast tier is waiting for: 150 of
            //  them made its last workgroups start 25 ms into the step on the ball-joint rollout's first steps)
            K.n_wait = std::min(64, std::max(b->q2_wait_min, 2 * handed2 + 8)); K.spares = b->d_fin;
            K.q_next = q3 ? b->d_lists + b->n_env : nullptr; K.q_next_count = q3 ? b->d_counts + 3 : nullptr;
            HIP_OK(uhc_launch_step(mode, 2, &K, d_action, d_tbase, nullptr, b->lds_bytes, b->side_stream));
            HIP_OK(hipEventRecord(b->ev_side1, b->side_stream));
        }
        if (q3 && !b->large_first) { if (launch_large()) return -1; }
        K.list = nullptr; K.list_count = nullptr; K.list_cursor = nullptr; K.grid = 0; K.prod_fin = nullptr; K.prod_total = 0; K.started = nullptr; K.spares = nullptr;
        // The fast tier's launch fills every CU's LDS the moment it starts; consumers that are not resident by then get theirs only when
        // its first workgroups leave (the tier trace showed them starting 3.7 ms into the step).  A one-thread gate on this stream holds
        // the launch back until every consumer workgroup has reported in (or 200 us have passed).
        if (waiting && !(b->A.dbg & 32)) HIP_OK(uhc_launch_gate(b->d_fin + 3, grid2, b->d_counts + 1,
                                                                     (b->A.dbg & 16) ? b->A.s.prof + (size_t)(b->n_env - 1) * 40 + 16 : nullptr, b->stream));
        K.tier_want = 1;
        K.order = b->d_order;  // (costliest envs first; null with UHC_DEBUG bit 3: env order)
        K.fin = waiting ? b->d_fin + 1 : nullptr;
        K.q_next = waiting ? b->d_lists : nullptr; K.q_next_count = waiting ? b->d_counts + 2 : nullptr;
        if (timed) HIP_OK(hipEventRecord(ev.first, b->stream));
        HIP_OK(uhc_launch_step(mode, 1, &K, d_action, d_tbase, d_active, b->lds_bytes_fast, b->stream));
        if (timed) { HIP_OK(hipEventRecord(ev.second, b->stream)); b->ev_used.push_back(ev); }
        K.tier_want = 0; K.sticky_mask = 0; K.fin = nullptr; K.q_next = nullptr; K.q_next_count = nullptr; K.order = nullptr;
        // chained launches on what is still flagged: everything handed on when no consumers run, nothing (two empty launches) when they do
        if (queues) HIP_OK(hipStreamWaitEvent(b->stream, b->ev_side1, 0));
        HIP_OK(uhc_launch_step(mode, 2, &K, d_action, d_tbase, b->A.s.pend2, b->lds_bytes, b->stream));
        if (q3) HIP_OK(hipStreamWaitEvent(b->stream, b->ev_side2, 0));
        if (q4) HIP_OK(hipStreamWaitEvent(b->stream, b->ev_side3, 0));
        if (big) HIP_OK(uhc_launch_step(mode, 3, &K, d_action, d_tbase, b->A.s.pend3, b->lds_bytes_big, b->stream));
        // the final queue lengths of this step, for the steps to come
        const int slot = (int)(b->cnt_step % 8);
        HIP_OK(hipMemcpyAsync(b->d_counts, b->A.s.q_abort, sizeof(int), hipMemcpyDeviceToDevice, b->stream));  // counts[0] carries the give-up count
        HIP_OK(hipMemcpyAsync(b->h_counts + 8 * slot, b->d_counts, 8 * sizeof(int), hipMemcpyDeviceToHost, b->stream));
        HIP_OK(hipEventRecord(b->cnt_ev[slot], b->stream));
        b->cnt_step++;
        return 0;
    }
    if (b->use_fast && !general) {
        if (timed) HIP_OK(hipEventRecord(ev.first, b->stream));
        HIP_OK(uhc_launch_step(mode, 1, &b->A, d_action, d_tbase, d_active, b->lds_bytes_fast, b->stream));
        if (timed) { HIP_OK(hipEventRecord(ev.second, b->stream)); b->ev_used.push_back(ev); }
        HIP_OK(uhc_launch_step(mode, 2, &b->A, d_action, d_tbase, b->A.s.pend2, b->lds_bytes, b->stream));
    } else {
        if (timed) HIP_OK(hipEventRecord(ev.first, b->stream));
        HIP_OK(uhc_launch_step(mode, 2, &b->A, d_action, d_tbase, d_active, b->lds_bytes, b->stream));
        if (timed) { HIP_OK(hipEventRecord(ev.second, b->stream)); b->ev_used.push_back(ev); }
    }
    if (big) HIP_OK(uhc_launch_step(mode, 3, &b->A, d_action, d_tbase, b->A.s.pend3, b->lds_bytes_big, b->stream));
    return 0;

    return 0;
}

extern "C" int32_t uhc_batch_set_overflow_mode(UhcBatch* b, int32_t truncate) {
    if (!b) return fail("uhc_batch_set_overflow_mode: null batch");
    b->A.truncate = truncate != 0;
    return 0;
}
extern "C" int32_t uhc_batch_set_timing(UhcBatch* b, int32_t enable) {
    if (!b) return fail("uhc_batch_set_timing: null batch");
    b->timing = enable != 0;
    return 0;
}
extern "C" int32_t uhc_batch_kernel_time(UhcBatch* b, double* total_ms, int32_t* launches) {
    if (!b || !total_ms || !launches) return fail("uhc_batch_kernel_time: null argument");
    double tot = 0;
    for (auto& ev : b->ev_used) {
        float ms = 0;
        HIP_OK(hipEventSynchronize(ev.second));
        HIP_OK(hipEventElapsedTime(&ms, ev.first, ev.second));
        tot += ms;
        b->ev_free.push_back(ev);
    }
    *total_ms = tot;
    *launches = (int32_t)b->ev_used.size();
    b->ev_used.clear();
    return 0;
}

extern "C" int32_t uhc_batch_forward(UhcBatch* b) {
    if (!b) return fail("uhc_batch_forward: null batch");
    HIP_OK(hipSetDevice(b->device));
    return launch(b, 1, nullptr, nullptr, nullptr);
}
extern "C" int32_t uhc_batch_set_state(UhcBatch* b, const int32_t* d_env_ids, int32_t n, const double* d_qpos, const double* d_qvel) {
    if (!b || !d_qpos || !d_qvel) return fail("uhc_batch_set_state: null argument");
    if (n < 1 || n > b->n_env) return fail("uhc_batch_set_state: n=%d outside [1,%d]", n, b->n_env);
    if (!d_env_ids && n != b->n_env) return fail("uhc_batch_set_state: env_ids==NULL requires n == n_env");
    HIP_OK(hipSetDevice(b->device));
    HIP_OK(hipMemsetAsync(b->reset_mask, 0, sizeof(int) * b->n_env, b->stream));
    HIP_OK(uhc_launch_set_state(&b->A.s, b->A.t.nq, b->A.t.nv, b->A.t.nu, d_env_ids, n, d_qpos, d_qvel, b->reset_mask, b->stream));
    // sim.forward() on the listed envs only (the others keep their one-substep-stale qM / qfrc_bias)
    return launch(b, 1, nullptr, nullptr, b->reset_mask);
}
extern "C" int32_t uhc_batch_simulate(UhcBatch* b, const double* d_action, const double* d_target_base, const int32_t* d_active) {
    if (!b || !d_action || !d_target_base) return fail("uhc_batch_simulate: null argument");
    HIP_OK(hipSetDevice(b->device));
    return launch(b, 0, d_action, d_target_base, d_active);
}

extern "C" hipError_t uhc_launch_set_state_masked(const DevState* s, int nq, int nv, int nu, int n_env, const int* select, const double* qpos,
                                                  const double* qvel, int* mask, hipStream_t stream);
// env layer: set_state + forward on the envs flagged in d_select (rows of d_qpos / d_qvel are indexed by env)
extern "C" int uhc_internal_set_state_masked(UhcBatch* b, const int* d_select, const double* d_qpos, const double* d_qvel) {
    HIP_OK(hipSetDevice(b->device));
    HIP_OK(uhc_launch_set_state_masked(&b->A.s, b->A.t.nq, b->A.t.nv, b->A.t.nu, b->n_env, d_select, d_qpos, d_qvel, b->reset_mask, b->stream));
    // only the kinematics now (the reset observation reads body poses); the dynamics part of sim.forward() runs at the head of the
    // env's next step kernel (DevState::fresh), which saves a forward-pass-long launch per control step
    const bool kf = b->use_fast && !b->general_only;
    HIP_OK(uhc_launch_step(2, kf ? 1 : 2, &b->A, nullptr, nullptr, b->reset_mask, kf ? b->lds_bytes_fast : b->lds_bytes, b->stream));
    return 0;
}

extern "C" int uhc_internal_trailing_free(UhcBatch* b) { return b->n_trailing_free; }
extern "C" int* uhc_internal_env_model(UhcBatch* b, int* n_models) { *n_models = b->n_models; return const_cast<int*>(b->A.s.env_model); }
// ------------------------------------------------------------------ internal accessors for the env layer (uhc_env_capi.cpp)
extern "C" int uhc_internal_set_error(const char* msg) { return fail("%s", msg); }
extern "C" int uhc_internal_batch_info(UhcBatch* b, int* n_env, int* nq, int* nv, int* nu, int* nbody, int* action_dim, int* vf_dim,
                                       double* dt, double* base_rot_inv, void** stream, int** reset_mask) {
    *n_env = b->n_env; *nq = b->A.t.nq; *nv = b->A.t.nv; *nu = b->A.t.nu; *nbody = b->A.t.nbody;
    *action_dim = b->A.c.action_dim; *vf_dim = b->A.c.rfc_mode == 1 ? 6 : b->A.c.rfc_mode == 2 ? b->A.c.n_vf_body * b->A.c.body_vf_dim : 0;
    *dt = b->A.t.timestep * b->A.c.n_substeps;
    for (int k = 0; k < 4; k++) base_rot_inv[k] = b->A.c.base_rot_inv[k];
    *stream = (void*)b->stream;
    *reset_mask = b->reset_mask;
    return 0;
}
