#pragma once
// uhc_primal.h -- Newton on the PRIMAL form of the contact problem, for the last tier (TIER 4) of the fused step kernel.
//
// The reference's models leave MuJoCo's solver at its default (assets/mujoco_models/humanoid_template.xml: no <option solver>): Newton on
//     min_a  1/2 (a - a_s)^T M (a - a_s) + sum_r s_r(J_r a - aref_r),     s_r(x) = 1/2 D_r x^2 for x < 0, else 0      [MJ-ext]
// whose minimiser is the optimum of the dual QP the other tiers work on (PGS sweeps / active sets on A = J M^-1 J^T + R).  The dual solvers
// pay per FORCE-CARRYING ROW (a 64 x 64 Delassus block in registers, windows beyond that); the primal pays per DOF: an nv x nv Hessian whatever
// the number of rows -- which is what an environment with hundreds of rows needs (a humanoid lying among boxes: 400-500 rows, 100-250 of
// them carrying a force; the reference asks MuJoCo for njmax 2500, uhc/khrylib/mocap/skeleton_mesh.py:46).
//
// In the coordinates the rows are stored in, u = D^1/2 L (a - a_s) with M = L^T D L, the mass matrix is the identity:
//     min_u  1/2 |u|^2 + sum_r 1/2 D_r min(0, Yhat_r . u + b_r)^2,      Yhat_r = D^-1/2 L^-T J_r^T (k_rows),  b_r = J_r a_s - aref_r
//     gradient  g = u + sum_active D_r jar_r Yhat_r,     Hessian  H = I + sum_active D_r Yhat_r Yhat_r^T   (>= I: the Cholesky cannot break down)
// and at the optimum u = sum_r f_r Yhat_r = z, f_r = -D_r min(0, jar_r): exactly what k_forward turns into qacc afterwards.
// Newton's method with an exact line search is invariant under the change of variables, so the test suite's checker (dense algebra in a-space)
// runs the same iteration:
//   start   u0 = sum f_ws Yhat (the forces the warm-start acceleration implies, k_rows) if its cost is below cost(0), else 0
//   step    active = {jar < 0};  H (packed lower, column-major, LDS) = I + sum_active D Yhat Yhat^T;  blocked right-looking Cholesky;
//           dir = -H^-1 g;  p = Yhat dir;  exact line search on the piecewise-quadratic cost (safeguarded Newton on its derivative);
//           u += alpha dir, jar += alpha p
//   stop    a full step (|alpha - 1| <= 1e-12) that leaves the active set as it was, or |g| <= 1e-14 |g_0|;  UHC_PRIMAL_MAXIT otherwise.
// Rows: chain rows lie packed in Yb (offsets rowY, dofs T.dof_anc), rows between two moving bodies as dense nv-vectors in Db (slot = type >> 8).
// Both live in HBM / L2 for this tier (KernelArgs::gY, gD): the LDS is the Hessian's.
//
// FOUR WAVES (round 6).  A tier-4 env-step is what ends a control step of the ball-joint rollouts, and on one wave it was 14 M cycles of which the
// row passes (jar, gradient, Hessian) took 58 % and the factorisation 18 %.  The queue consumers of this tier (uhc_k_huge_q.hip, -DUHC_NW4) are
// workgroups of UHC_PRIMAL_WAVES waves, one per SIMD of the CU whose LDS the workgroup owns anyway: wave 0 runs the step as every other tier does
// (all of uhc_physics_impl.h assumes workgroup = wave) and the three HELPERS sleep in s_barrier until wave 0 posts a command in the LDS mailbox:
//   * row dots (jar = Yhat u + b, p = Yhat dir): rows dealt out in blocks of 64, dense rows in groups of four;
//   * gradient + Hessian: every dof -- an entry of the gradient, a COLUMN of the Hessian -- is OWNED by wave (depth of the dof) & 3.  A chain row's
//     entry at chain position q belongs to the dof of depth q, so the pair (q, q2) of its outer product goes to wave q2 & 3 whichever row it comes
//     from: every wave walks ALL runs of rows (its own staging strip, no barrier between runs) and adds its own pairs -- no two waves ever touch the
//     same word, no atomics; the dense rows' columns are dealt out by the same rule;
//   * Cholesky: right-looking in panels of UHC_PRIMAL_NB columns: wave 0 factorises the panel in registers (lane = row), all waves share the
//     trailing update column by column.
// The one-workgroup-per-env kernels (set_state's forward pass, the chained launches) run the SAME stage functions on their one wave (NW = 1: it owns
// everything); every word of H receives the same additions in the same order whatever NW is; a gradient is summed per wave over the rows dealt to it and the
// partial sums are added in wave order, so the two forms agree to rounding (1e-16 relative), not to the bit.
#define UHC_PRIMAL_MAXIT 100
#define UHC_PRIMAL_LS_MAXIT 60
#define UHC_PRIMAL_DGROUP 8  // dense rows per pass over the Hessian
#define UHC_PRIMAL_MAXFLIP 10  // rows that may change sides between two iterations for the factor to be updated instead of rebuilt
#define UHC_PRIMAL_NB 16  // columns per panel of the factorisation
#ifdef UHC_NW4
#define UHC_PRIMAL_NW UHC_PRIMAL_WAVES
#else
#define UHC_PRIMAL_NW 1
#endif
enum { PCMD_EXIT = 0, PCMD_SCATTER_U, PCMD_DOTS_U_JAR, PCMD_GRAD, PCMD_GRAD_HESS, PCMD_CHOL, PCMD_DOTS_DIR_P };

// 1 / sqrt(x) to full double precision without the IEEE sqrt + division sequences (~170 cycles together, on the column-to-column critical path of the
// factorisation and of the rank-one updates): v_rsq_f64 + three Newton steps in fused form (e = 1 - x r^2; r += r e / 2)
__device__ __forceinline__ double rsqrt_newton(double x) {
    double r = __builtin_amdgcn_rsq(x);
#pragma unroll
    for (int k = 0; k < 3; k++) { const double e = fma(-x * r, r, 1.0); r = fma(0.5 * r, e, r); }
    return r;
}
// packed lower triangle, column by column: column j holds rows j .. n-1
__device__ __forceinline__ int hcol(int j, int n) { return j * n - (j * (j - 1)) / 2; }
// all waves of the workgroup (NW > 1: a real barrier; the one-wave kernels: the wave-local fence every other stage uses)
template <int NW> __device__ __forceinline__ void mw_barrier() { if constexpr (NW > 1) __syncthreads(); else wsync(); }

struct PrimalCtx {
    const RowMisc* RM; const int* RY; const int* NI; const int* dof_depth;
    int YS, n, nvp, nefc, nslot, nruns;
    double *u, *vec, *jar, *pp, *Dr, *cf, *H, *dsc;
    const double* bb;
    short* anc_tab; double *stY, *dstage, *cw, *part;
    int* runs;
    unsigned short *pair_all, *pair_cls, *pair_cnt;
    int* mbx;
    const double *Yb, *Db;
};
template <int TIER>
__device__ __forceinline__ PrimalCtx primal_ctx(const KernelArgs& A, double* S, int nefc, const double* Yb, const double* Db) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    PrimalCtx C;
    C.RM = (const RowMisc*)(S + L.rowMisc); C.RY = (const int*)(S + L.rowY); C.NI = (const int*)(S + L.ncon_nefc); C.dof_depth = T.dof_depth;
    C.YS = T.maxdepth + 1; C.n = T.nv; C.nvp = A.nvp; C.nefc = nefc;
    C.nslot = cap_of<TIER>(A).ndense > 0 ? __builtin_amdgcn_readfirstlane(C.NI[2]) : 0;
    C.u = S + L.z;
    C.vec = S + L.vec;       // gradient, then the Newton direction
    C.jar = S + L.rowAref;   // Yhat_r . u + b_r  (the slot held D jar of the warm start for the working sets' ranking: not used in this tier)
    C.pp = S + L.rowDa;      // the rows' weights in the Hessian, then Yhat_r . dir  (the slot held diag(A): only the sweeps read it, and they do not run after this)
    C.Dr = S + L.rowW;       // 1 / R_r
    C.cf = S + L.rowF;       // per-row coefficient of the current scatter; the forces at the end
    C.bb = S + L.rowB;
    C.H = S + L.H; C.dsc = S + L.dsc;
    // the contacts' storage: nothing reads the contacts once the rows are built (the host sizes it for all of this: uhc_device.h primal_scratch)
    const PrimalScratch ps = primal_scratch(T.nv, C.YS);
    double* X = S + L.con;
    C.anc_tab = (short*)(X + ps.anc); C.stY = X + ps.stY; C.dstage = X + ps.dstage; C.cw = X + ps.cw;
    C.pair_all = (unsigned short*)(X + ps.pair_all); C.pair_cls = (unsigned short*)(X + ps.pair_cls); C.pair_cnt = (unsigned short*)(X + ps.pair_cnt);
    C.mbx = (int*)(X + ps.mbx);
    C.part = X + ps.part; C.runs = (int*)(X + ps.runs);
    C.nruns = 0;  // (k_primal: after primal_run_table; the helpers: from the mailbox)
    C.Yb = Yb; C.Db = Db;
    return C;
}

// y = Yhat_r . v for every row (v: an nv-vector in LDS), into out[r] (+ add[r] when add != nullptr); dense rows wave-cooperatively first (four rows
// per round: their loads and reductions overlap), then lane = row: its packed entries stream from L2 four independent loads per round -- a loop of
// dependent single loads pays the L2 latency once per entry -- and the dof indices come from the LDS copy of the chain table.
template <int NW>
__device__ __forceinline__ void primal_row_dots(const PrimalCtx& C, int wid, const double* v, double* out, const double* add) {
    const bool v0 = LANE < C.n, v1 = LANE + UHC_WAVE < C.n;
    const double va = v0 ? v[LANE] : 0.0, vb = v1 ? v[LANE + UHC_WAVE] : 0.0;
    for (int k0 = 4 * wid; k0 < C.nslot; k0 += 4 * NW) {
        double s[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const double* Dk = C.Db + (size_t)min(k0 + j, C.nslot - 1) * C.nvp;
            s[j] = (v0 ? Dk[LANE] * va : 0.0) + (v1 ? Dk[LANE + UHC_WAVE] * vb : 0.0);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) s[j] = wave_sum(s[j]);
#pragma unroll
        for (int j = 0; j < 4; j++) if (LANE == 0 && k0 + j < C.nslot) C.dsc[4 * (k0 + j)] = s[j];
    }
    mw_barrier<NW>();
    for (int r = wid * UHC_WAVE + LANE; r < C.nefc; r += UHC_WAVE * NW) {
        const RowMisc rm = C.RM[r];
        double y = add ? add[r] : 0.0;
        if (rm.type & ROW_TWO) y += C.dsc[4 * (rm.type >> 8)];
        else {
            const int len = C.RY[r + 1] - C.RY[r];
            const short* anc = C.anc_tab + rm.last * C.YS;
            const double* Yr = C.Yb + C.RY[r];
            double y1 = 0.0;
            int q = 0;
            for (; q + 4 <= len; q += 4) {
                const double a0 = Yr[q], a1 = Yr[q + 1], a2 = Yr[q + 2], a3 = Yr[q + 3];
                y = fma(a0, v[anc[q]], y); y1 = fma(a1, v[anc[q + 1]], y1);
                y = fma(a2, v[anc[q + 2]], y); y1 = fma(a3, v[anc[q + 3]], y1);
            }
            for (; q < len; q++) y = fma(Yr[q], v[anc[q]], y);
            y += y1;
        }
        out[r] = y;
    }
    mw_barrier<NW>();
}

// The chain rows' share of a scatter (vec += sum_r c_r Yhat_r) and of the Hessian (H += sum_r w_r Yhat_r Yhat_r^T), WITHOUT atomics.  (A first version
// let every lane push its own row's len^2 / 2 products into H with LDS float64 atomics: measured at ~12 cycles per lane and instruction, 1.3 M cycles per
// Hessian.)  Rows that share a dof chain -- the 4 pyramid edges of a contact, all contacts of one hull: RUNS of up to 16 consecutive rows with the same
// last dof -- are taken TOGETHER: their entries are staged in the wave's own LDS strip ([16][32]), then lane = chain position q adds sum_t c_t y_t[q] to
// vec[dof(q)], and lane = pair (q, q2) adds sum_t w_t y_t[q] y_t[q2] to H[dof(q)][dof(q2)] -- distinct addresses inside a run, and the LDS queue of the
// wave keeps runs in order: plain read-modify-write.
// The runs are a property of the rows, not of the iterate: k_primal tabulates them once per solve (primal_run_table).  A pass first marks the runs with a
// non-zero coefficient or weight (lane = run, one ballot per 64 runs), then walks the marked ones with the NEXT run's entries already on their way from L2
// into registers while the current one is added -- the first version paid one exposed L2 round trip per run, and that, not the arithmetic, was its cost
// (four waves that each walked every run were no faster than one: profiles/r06_c_diag_tier4_configs4_*.txt).
//   WITH_H (four waves): every wave walks every marked run and takes the positions q with q & 3 == wid and the pairs with q2 & 3 == wid (the class
//     tables): the dof at chain position q has depth q, so these are exactly the words the wave owns -- no two waves touch the same word.
//   gradient alone (four waves): the marked runs are DEALT OUT to the waves, every wave adds its runs into a partial vector of its own, and the stage
//     function sums the partials in wave order behind a barrier.
#define PRUN_R0(pk) ((pk) & 1023)
#define PRUN_NB(pk) (((pk) >> 10) & 31)
#define PRUN_LEN(pk) (((pk) >> 15) & 63)
#define PRUN_LAST(pk) (((pk) >> 21) & 255)
// wave 0, once per solve: runs[k] = r0 | nb << 10 | len << 15 | last dof << 21 for every run of chain rows; returns their number
__device__ __forceinline__ int primal_run_table(const PrimalCtx& C) {
    int nr = 0, r0 = 0;
    while (r0 < C.nefc) {
        const int rr = r0 + LANE;
        int last_l = -2, two_l = 1;
        if (LANE < 16 && rr < C.nefc) { const RowMisc rm = C.RM[rr]; last_l = rm.last; two_l = (rm.type & ROW_TWO) ? 1 : 0; }
        const int last0 = __builtin_amdgcn_readfirstlane(last_l);
        if (__builtin_amdgcn_readfirstlane(two_l)) { r0++; continue; }  // (dense rows: the stage function's own loops)
        const unsigned long long same = __builtin_amdgcn_ballot_w64(LANE < 16 && rr < C.nefc && !two_l && last_l == last0);
        const int nb = __builtin_ctzll(~same);  // consecutive rows from r0 on with this chain (>= 1, <= 16)
        const int len = __builtin_amdgcn_readfirstlane(C.RY[r0 + 1] - C.RY[r0]);
        if (LANE == 0) C.runs[nr] = r0 | (nb << 10) | (len << 15) | (last0 << 21);
        nr++;
        r0 += nb;
    }
    return nr;
}
template <int NW, bool WITH_H>
__device__ __forceinline__ void primal_chain_pass(const PrimalCtx& C, int wid, const double* c, const double* w, double* vec) {
    double* stY = C.stY + wid * (16 * 32);
    double* cw = C.cw + wid * 32;
    const unsigned short* ptab = NW == 1 ? C.pair_all : C.pair_cls + wid * UHC_PRIMAL_CLS_STRIDE;
    const int n = C.n, nr = C.nruns;
    int dealt = 0;  // gradient alone, NW > 1: marked runs seen so far (run k of them is wave k % NW's)
    for (int rb = 0; rb < nr; rb += UHC_WAVE) {
        // ---- which of these 64 runs contribute
        int pk = 0;
        bool any = false;
        if (rb + LANE < nr) {
            pk = C.runs[rb + LANE];
            const int r0 = PRUN_R0(pk), nb = PRUN_NB(pk);
            for (int t = 0; t < nb; t++) any = any || c[r0 + t] != 0.0 || (WITH_H && w[r0 + t] != 0.0);
        }
        unsigned long long m = __builtin_amdgcn_ballot_w64(any);
        if (!WITH_H && NW > 1) {  // this wave's share of the marked runs
            unsigned long long mine = 0ull, mm = m;
            while (mm) { const unsigned long long bit = mm & (0ull - mm); mm ^= bit; if ((dealt++ % NW) == wid) mine |= bit; }
            m = mine;
        }
        if (!m) continue;
        // ---- the marked runs in turn, the next one's entries in flight
        double y[8];  // entries idx = LANE + 64 j of the run's [nb][32] strip
        auto fetch = [&](int pkr) __attribute__((always_inline)) {
            const int r0 = PRUN_R0(pkr), nb = PRUN_NB(pkr), len = PRUN_LEN(pkr);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                y[j] = 0.0;
                if (2 * j < nb) {  // (wave-uniform: rows 2 j and 2 j + 1 of the run)
                    const int t = 2 * j + (LANE >> 5), q = LANE & 31;
                    if (t < nb && q < len) y[j] = C.Yb[C.RY[r0 + t] + q];
                }
            }
        };
        int cur = __builtin_ctzll(m);
        m &= m - 1;
        int pkc = __builtin_amdgcn_readlane(pk, cur);
        fetch(pkc);
        for (;;) {
            const int r0 = PRUN_R0(pkc), nb = PRUN_NB(pkc), len = PRUN_LEN(pkc), last0 = PRUN_LAST(pkc);
#pragma unroll
            for (int j = 0; j < 8; j++) if (2 * j < nb) stY[LANE + UHC_WAVE * j] = y[j];
            // (the run's coefficients go through LDS, not v_readlane: the readers below sit in divergent branches -- lanes < len, lanes < np -- and a
            //  register the compiler spilled and reloads INSIDE such a branch holds garbage in the lanes the branch switched off, which are exactly the
            //  lanes a readlane of row t >= len would read.  The 552-spill instantiation <0, 3> of round 5 did that and produced NaNs.)
            if (LANE < 16) { cw[LANE] = LANE < nb ? c[r0 + LANE] : 0.0; cw[16 + LANE] = (WITH_H && LANE < nb) ? w[r0 + LANE] : 0.0; }
            const bool more = m != 0ull;
            int pkn = 0;
            if (more) { const int nx = __builtin_ctzll(m); m &= m - 1; pkn = __builtin_amdgcn_readlane(pk, nx); fetch(pkn); }  // (y is free again: its values are on their way to LDS)
            wsync();
            const short* anc = C.anc_tab + last0 * C.YS;
            if (LANE < len && (!WITH_H || NW == 1 || (LANE & 3) == wid)) {
                double sg = 0.0;
                for (int t = 0; t < nb; t++) sg = fma(cw[t], stY[t * 32 + LANE], sg);
                vec[anc[LANE]] += sg;
            }
            if (WITH_H) {
                const int np = NW == 1 ? (len * (len + 1)) / 2 : (int)C.pair_cnt[wid * 33 + len];
                for (int p0 = 0; p0 < np; p0 += UHC_WAVE) {
                    const int pi = p0 + LANE;
                    if (pi < np) {
                        const int qq = ptab[pi], q = qq >> 8, q2 = qq & 0xff;
                        double sh = 0.0;
                        for (int t = 0; t < nb; t++) sh = fma(cw[16 + t] * stY[t * 32 + q], stY[t * 32 + q2], sh);
                        const int i = anc[q], jc = anc[q2];
                        C.H[hcol(jc, n) - jc + i] += sh;
                    }
                }
            }
            wsync();
            if (!more) break;
            pkc = pkn;
        }
    }
}

// One stage of a Newton iteration, run by every wave of the workgroup:
//     target = init (or 0) + sum_r cf_r Yhat_r          and, WITH_H,        H = I + sum_r pp_r Yhat_r Yhat_r^T
// cf: the rows' coefficients (gradient: D jar of the active rows; start point: the warm-start forces), pp: their weights (D of the active rows, else 0).
// WITH_H: every wave works on the dofs it OWNS (NW == 1: all of them) -- entries of target, columns of H.  Without: the rows are dealt out, every wave sums
// its rows into its own partial vector, and the partials are added up in wave order.
template <int NW, bool WITH_H>
__device__ __forceinline__ void primal_grad_hess(const PrimalCtx& C, int wid, double* target, const double* init) {
    const int n = C.n;
    const bool v0 = LANE < n, v1 = LANE + UHC_WAVE < n;
    const int ia = min(LANE, n - 1), ib = min(LANE + UHC_WAVE, n - 1);
    if constexpr (!WITH_H) {
        double* part = C.part + wid * 128;
        if (v0) part[LANE] = 0.0;
        if (v1) part[LANE + UHC_WAVE] = 0.0;
        wsync();
        primal_chain_pass<NW, false>(C, wid, C.cf, C.pp, part);
        {   // the dense rows' share: four rows per wave and round, lane = dof
            double ga = 0.0, gb = 0.0;
            for (int k0 = 4 * wid; k0 < C.nslot; k0 += 4 * NW) {
                double ck[4], da[4], db[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int k = min(k0 + j, C.nslot - 1);
                    ck[j] = k0 + j < C.nslot ? C.cf[__builtin_amdgcn_readfirstlane(C.NI[4 + k])] : 0.0;
                    const double* Dk = C.Db + (size_t)k * C.nvp;
                    da[j] = v0 ? Dk[LANE] : 0.0; db[j] = v1 ? Dk[LANE + UHC_WAVE] : 0.0;
                }
#pragma unroll
                for (int j = 0; j < 4; j++) { ga = fma(ck[j], da[j], ga); gb = fma(ck[j], db[j], gb); }
            }
            if (v0) part[LANE] += ga;
            if (v1) part[LANE + UHC_WAVE] += gb;
        }
        mw_barrier<NW>();
        for (int i = wid * 32 + (LANE & 31); i < n; i += 32 * NW) {  // (lanes 32-63 repeat lanes 0-31: the same value to the same word)
            double sum = init ? init[i] : 0.0;
#pragma unroll
            for (int w2 = 0; w2 < NW; w2++) sum += C.part[w2 * 128 + i];
            target[i] = sum;
        }
        mw_barrier<NW>();
        return;
    } else {
    const bool own0 = v0 && (NW == 1 || (C.dof_depth[ia] & 3) == wid), own1 = v1 && (NW == 1 || (C.dof_depth[ib] & 3) == wid);
    const unsigned long long own_m[2] = {__builtin_amdgcn_ballot_w64(own0), __builtin_amdgcn_ballot_w64(own1)};
    if (own0) target[LANE] = init ? init[LANE] : 0.0;
    if (own1) target[LANE + UHC_WAVE] = init ? init[LANE + UHC_WAVE] : 0.0;
    // the owned columns: zero below the diagonal, one on it
    for (int h = 0; h < 2; h++) {
        unsigned long long m = own_m[h];
        while (m) {
            const int j = UHC_WAVE * h + __builtin_ctzll(m);
            m &= m - 1;
            double* Hj = C.H + hcol(j, n) - j;
            if (LANE >= j && v0) Hj[LANE] = LANE == j ? 1.0 : 0.0;
            if (LANE + UHC_WAVE >= j && v1) Hj[LANE + UHC_WAVE] = LANE + UHC_WAVE == j ? 1.0 : 0.0;
        }
    }
    wsync();
    primal_chain_pass<NW, true>(C, wid, C.cf, C.pp, target);
    {   // the dense rows' share of the scatter: lane = owned dof, four rows' loads in flight
        double ga = 0.0, gb = 0.0;
        for (int k0 = 0; k0 < C.nslot; k0 += 4) {
            double ck[4], da[4], db[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int k = min(k0 + j, C.nslot - 1);
                ck[j] = k0 + j < C.nslot ? C.cf[__builtin_amdgcn_readfirstlane(C.NI[4 + k])] : 0.0;
                const double* Dk = C.Db + (size_t)k * C.nvp;
                da[j] = own0 ? Dk[LANE] : 0.0; db[j] = own1 ? Dk[LANE + UHC_WAVE] : 0.0;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) { ga = fma(ck[j], da[j], ga); gb = fma(ck[j], db[j], gb); }
        }
        if (own0) target[LANE] += ga;
        if (own1) target[LANE + UHC_WAVE] += gb;
    }
    {
        // dense rows, UHC_PRIMAL_DGROUP at a time: lane = row index i of the Hessian (its own y_i of the group's rows in registers), the wave's columns j
        // in turn: the column is contiguous in i, D y_j of every row of the group comes by one broadcast LDS read from the staged copy (of which every
        // wave writes and reads the dofs it owns)
        int k = 0;
        while (k < C.nslot) {
            DofVec y[UHC_PRIMAL_DGROUP];
            int got = 0;
#pragma unroll
            for (int s2 = 0; s2 < UHC_PRIMAL_DGROUP; s2++) { y[s2].a = y[s2].b = 0.0; }
            while (k < C.nslot && got < UHC_PRIMAL_DGROUP) {
                const int rid = __builtin_amdgcn_readfirstlane(C.NI[4 + k]);
                const double dd = C.pp[rid];  // (wave-uniform address: D of an active row, else 0)
                if (__builtin_amdgcn_readfirstlane(__double2hiint(dd)) != 0 || __builtin_amdgcn_readfirstlane(__double2loint(dd)) != 0) {
                    const double* Dk = C.Db + (size_t)k * C.nvp;
                    const double ya = v0 ? Dk[LANE] : 0.0, yb = v1 ? Dk[LANE + UHC_WAVE] : 0.0;
#pragma unroll
                    for (int s2 = 0; s2 < UHC_PRIMAL_DGROUP; s2++) if (s2 == got) { y[s2].a = ya; y[s2].b = yb; }
                    if (own0) C.dstage[LANE * UHC_PRIMAL_DGROUP + got] = dd * ya;   // [dof][row of the group]: one 64-byte line per column j
                    if (own1) C.dstage[(LANE + UHC_WAVE) * UHC_PRIMAL_DGROUP + got] = dd * yb;
                    got++;
                }
                k++;
            }
            if (got == 0) break;
            for (int g = got; g < UHC_PRIMAL_DGROUP; g++) {  // (unused places of the last group: zero multipliers)
                if (own0) C.dstage[LANE * UHC_PRIMAL_DGROUP + g] = 0.0;
                if (own1) C.dstage[(LANE + UHC_WAVE) * UHC_PRIMAL_DGROUP + g] = 0.0;
            }
            wsync();
            for (int h = 0; h < 2; h++) {
                unsigned long long m = own_m[h];
                while (m) {
                    const int j = UHC_WAVE * h + __builtin_ctzll(m);
                    m &= m - 1;
                    double* Hj = C.H + hcol(j, n) - j;
                    double yj[UHC_PRIMAL_DGROUP];
#pragma unroll
                    for (int s2 = 0; s2 < UHC_PRIMAL_DGROUP; s2++) yj[s2] = C.dstage[j * UHC_PRIMAL_DGROUP + s2];
                    double ha = Hj[ia], hb = Hj[ib];  // (rows above the diagonal: in-range garbage, never stored)
#pragma unroll
                    for (int s2 = 0; s2 < UHC_PRIMAL_DGROUP; s2++) { ha = fma(yj[s2], y[s2].a, ha); hb = fma(yj[s2], y[s2].b, hb); }
                    if (LANE >= j && v0) Hj[LANE] = ha;
                    if (LANE + UHC_WAVE >= j && v1) Hj[LANE + UHC_WAVE] = hb;
                }
            }
            wsync();
        }
    }
    mw_barrier<NW>();
    }
}

// Cholesky H = C C^T of the packed Hessian, right-looking in panels of UHC_PRIMAL_NB columns; the diagonal keeps 1 / C_jj (H >= I: the pivots are >= 1 up
// to rounding).  Panel: wave 0, lane = row (rows LANE and LANE + 64), the panel's NB entries of either row in registers; a column is scaled by the
// reciprocal root of its diagonal entry and taken off the later columns of the panel, the multipliers by v_readlane (wave-uniform control flow
// throughout).  Trailing update: the columns behind the panel are dealt out to the waves; per column NB broadcast reads (row j of the panel), the lane's
// two entries read-modify-written.  Rows above the diagonal compute garbage from in-range reads and are not stored.
// (Round 5's left-looking factorisation by one wave: 150 k cycles at nv = 117, 18-44 % of a tier-4 env-step.)
template <int NW>
__device__ __forceinline__ void primal_chol(double* H, int n, int wid) {
    const int ia = min(LANE, n - 1), ib = min(LANE + UHC_WAVE, n - 1);
    for (int jb = 0; jb < n; jb += UHC_PRIMAL_NB) {
        const int nbk = min(UHC_PRIMAL_NB, n - jb);
        if (wid == 0) {
            double Pa[UHC_PRIMAL_NB], Pb[UHC_PRIMAL_NB];
#pragma unroll
            for (int p = 0; p < UHC_PRIMAL_NB; p++) {
                const int j = min(jb + p, n - 1);
                const double* Hj = H + hcol(j, n) - j;
                Pa[p] = Hj[ia]; Pb[p] = Hj[ib];
            }
#pragma unroll
            for (int p = 0; p < UHC_PRIMAL_NB; p++) {
                if (p < nbk) {
                    const int j = jb + p;
                    DofVec col = {Pa[p], Pb[p]};
                    const double rc = rsqrt_newton(dv_get_nb(col, j));
                    col.a *= rc; col.b *= rc;
                    Pa[p] = col.a; Pb[p] = col.b;
#pragma unroll
                    for (int p2 = p + 1; p2 < UHC_PRIMAL_NB; p2++) {
                        if (p2 < nbk) {
                            const double l = dv_get_nb(col, jb + p2);  // C[jb + p2][j]
                            Pa[p2] = fma(-col.a, l, Pa[p2]); Pb[p2] = fma(-col.b, l, Pb[p2]);
                        }
                    }
                    double* Hj = H + hcol(j, n) - j;
                    if (LANE >= j && LANE < n) Hj[LANE] = LANE == j ? rc : col.a;
                    if (LANE + UHC_WAVE >= j && LANE + UHC_WAVE < n) Hj[LANE + UHC_WAVE] = LANE + UHC_WAVE == j ? rc : col.b;
                }
            }
        }
        mw_barrier<NW>();
        const int j0 = jb + nbk;
        if (j0 < n) {
            double La[UHC_PRIMAL_NB], Lb[UHC_PRIMAL_NB];  // the lane's rows of the panel (zero beyond the panel's width)
#pragma unroll
            for (int p = 0; p < UHC_PRIMAL_NB; p++) {
                const int j = min(jb + p, n - 1);
                const double* Hj = H + hcol(j, n) - j;
                La[p] = p < nbk ? Hj[ia] : 0.0; Lb[p] = p < nbk ? Hj[ib] : 0.0;
            }
            for (int j = j0 + wid; j < n; j += NW) {
                double* Hj = H + hcol(j, n) - j;
                double lj[UHC_PRIMAL_NB];
#pragma unroll
                for (int p = 0; p < UHC_PRIMAL_NB; p++) { const int jp = min(jb + p, n - 1); lj[p] = H[hcol(jp, n) - jp + j]; }  // C[j][jb + p]
                double ha = Hj[ia], hb = Hj[ib];
#pragma unroll
                for (int p = 0; p < UHC_PRIMAL_NB; p++) { ha = fma(-La[p], lj[p], ha); hb = fma(-Lb[p], lj[p], hb); }
                if (LANE >= j && LANE < n) Hj[LANE] = ha;
                if (LANE + UHC_WAVE >= j && LANE + UHC_WAVE < n) Hj[LANE + UHC_WAVE] = hb;
            }
        }
        mw_barrier<NW>();
    }
}

// Rank-one update (sigma = +1) or downdate (-1) of the packed Cholesky factor, C C^T <- C C^T + sigma x x^T, column by column (the LINPACK rotation
// scheme): lane = row, x in registers, one column read and written per step.  ~150 cycles per column: when a Newton iteration moves a handful of rows
// across jar = 0, the factor follows them instead of being rebuilt.  Returns false when a downdate loses positive definiteness to rounding
// (r^2 <= 0): the caller rebuilds.  The diagonal holds 1 / C_kk throughout.  (Wave 0 alone.)
__device__ __forceinline__ bool primal_chol_rank1(double* H, int n, DofVec x, double sigma) {
    // With rd = 1 / C_kk (what the diagonal stores) and s = x_k rd:  r^2 = C_kk^2 q,  q = 1 + sigma s^2;  w = 1 / sqrt(q);  the rotation is c = q w, s;  the new
    // column (C_ik + sigma s x_i) w, the new diagonal's reciprocal rd w, x_i <- c x_i - s C_ik(new): one reciprocal root per column and no division.  The next
    // column is fetched while this one's chain (read -> root -> write) runs.
    const int ib = min(LANE + UHC_WAVE, n - 1);
    bool good = true;
    int base = 0;  // hcol(k) - k
    double ca = H[base + LANE], cb = H[base + ib], rd = H[base + 0];  // column k (rows above the diagonal: in-range garbage, never stored)
    for (int k = 0; k < n; k++) {
        const int nbase = k + 1 < n ? base + n - k - 1 : base, kn = min(k + 1, n - 1);
        const double na_ = H[nbase + LANE], nb_ = H[nbase + ib], nrd = H[nbase + kn];
        const double sk = dv_get_nb(x, k) * rd;
        const double q = fma(sigma * sk, sk, 1.0);
        good = good && (q > 0.0);
        const double w = rsqrt_newton(q > 0.0 ? q : 1.0);
        const double c = q * w, sg = sigma * sk;
        const double na = fma(sg, x.a, ca) * w, nb = fma(sg, x.b, cb) * w;
        if (LANE > k && LANE < n) H[base + LANE] = na;
        if (LANE + UHC_WAVE > k && LANE + UHC_WAVE < n) H[base + LANE + UHC_WAVE] = nb;
        if (LANE == (k & 63)) { if (k < UHC_WAVE) H[base + k] = rd * w; }
        if (k >= UHC_WAVE && LANE + UHC_WAVE == k) H[base + k] = rd * w;
        x.a = fma(c, x.a, -sk * na);
        x.b = fma(c, x.b, -sk * nb);
        ca = na_; cb = nb_; rd = nrd; base = nbase;
    }
    wsync();
    return good;
}

// wave 0 hands a stage to the helper waves (NW > 1): the command goes into the mailbox, the barrier releases them; the stage function itself -- which
// wave 0 runs too, as wid 0 -- ends in the barrier that collects them again
template <int NW>
__device__ __forceinline__ void primal_post(const PrimalCtx& C, int cmd) {
    if constexpr (NW > 1) {
        if (LANE == 0) C.mbx[0] = cmd;
        __syncthreads();
    }
}
#ifdef UHC_NW4
// The helper waves of a four-wave consumer: asleep in the barrier until wave 0 posts a command; they know nothing of the env but what the mailbox says.
template <int TIER>
__device__ __forceinline__ void primal_helper(const KernelArgs& A, double* S) {
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const PrimalScratch ps = primal_scratch(A.t.nv, A.t.maxdepth + 1);
    const int* mbx = (const int*)(S + lds_of<TIER>(A).con + ps.mbx);
    for (;;) {
        __syncthreads();
        const int cmd = __builtin_amdgcn_readfirstlane(mbx[0]);
        if (cmd == PCMD_EXIT) return;
        const int nefc = __builtin_amdgcn_readfirstlane(mbx[1]);
        const unsigned long long yb = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(mbx[3]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(mbx[2]);
        const unsigned long long db = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(mbx[5]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(mbx[4]);
        PrimalCtx C = primal_ctx<TIER>(A, S, nefc, (const double*)yb, (const double*)db);
        C.nruns = __builtin_amdgcn_readfirstlane(mbx[6]);
        if (cmd == PCMD_SCATTER_U) primal_grad_hess<UHC_PRIMAL_NW, false>(C, wid, C.u, nullptr);
        else if (cmd == PCMD_DOTS_U_JAR) primal_row_dots<UHC_PRIMAL_NW>(C, wid, C.u, C.jar, C.bb);
        else if (cmd == PCMD_GRAD) primal_grad_hess<UHC_PRIMAL_NW, false>(C, wid, C.vec, C.u);
        else if (cmd == PCMD_GRAD_HESS) primal_grad_hess<UHC_PRIMAL_NW, true>(C, wid, C.vec, C.u);
        else if (cmd == PCMD_CHOL) primal_chol<UHC_PRIMAL_NW>(C.H, C.n, wid);
        else if (cmd == PCMD_DOTS_DIR_P) primal_row_dots<UHC_PRIMAL_NW>(C, wid, C.vec, C.pp, nullptr);
    }
}
// wave 0, at the end of the kernel: the helpers leave
template <int TIER>
__device__ __forceinline__ void primal_release_helpers(const KernelArgs& A, double* S) {
    const PrimalScratch ps = primal_scratch(A.t.nv, A.t.maxdepth + 1);
    int* mbx = (int*)(S + lds_of<TIER>(A).con + ps.mbx);
    if (LANE == 0) mbx[0] = PCMD_EXIT;
    __syncthreads();
}
#endif

// returns the Newton iterations taken (>= 1), negated when the iteration cap was reached; z = u in S[L.z], the forces in S[L.rowF]
template <int TIER>
__device__ __forceinline__ int k_primal(const KernelArgs& A, const double* mb, double* S, int nefc, const LaneConst& LC, const double* Yb, const double* Db, int* nact PROF_ARGS) {
    constexpr int NW = UHC_PRIMAL_NW;
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    PrimalCtx C = primal_ctx<TIER>(A, S, nefc, Yb, Db);
    const RowMisc* RM = C.RM;
    const int* RY = C.RY;
    const int YS = C.YS, n = C.n;
    double *u = C.u, *vec = C.vec, *jar = C.jar, *pp = C.pp, *Dr = C.Dr, *cf = C.cf, *H = C.H;
    // the dof chains (which dof sits at position q of the chain that ends in dof i), from L2 into the contacts' storage: nothing reads the contacts
    // once the rows are built, and every row pass of every Newton iteration walks this table; the pair tables of the chain pass
    for (int i = LANE; i < n * YS; i += UHC_WAVE) C.anc_tab[i] = T.dof_anc[i];
    if (LANE < 32) {
        for (int q2 = 0; q2 <= LANE; q2++) C.pair_all[(LANE * (LANE + 1)) / 2 + q2] = (unsigned short)((LANE << 8) | q2);
        for (int c = 0; c < 4; c++) {  // class c: the pairs with q2 & 3 == c, q-major; lane = q
            int o = 0;
            for (int q = c; q < LANE; q++) o += (q - c) / 4 + 1;
            for (int q2 = c; q2 <= LANE; q2 += 4) C.pair_cls[c * UHC_PRIMAL_CLS_STRIDE + o++] = (unsigned short)((LANE << 8) | q2);
        }
    }
    if (LANE < 33)
        for (int c = 0; c < 4; c++) {  // pairs of class c with q < LANE
            int o = 0;
            for (int q = c; q < LANE; q++) o += (q - c) / 4 + 1;
            C.pair_cnt[c * 33 + LANE] = (unsigned short)o;
        }
    C.nruns = primal_run_table(C);
    if (NW > 1 && LANE == 0) {
        C.mbx[1] = nefc; C.mbx[6] = C.nruns;
        C.mbx[2] = (int)((unsigned long long)Yb & 0xffffffffull); C.mbx[3] = (int)((unsigned long long)Yb >> 32);
        C.mbx[4] = (int)((unsigned long long)Db & 0xffffffffull); C.mbx[5] = (int)((unsigned long long)Db >> 32);
    }
    // ---- start point: u0 = sum f_ws Yhat, kept if its cost is below the cost of u = 0
    for (int r = LANE; r < nefc; r += UHC_WAVE) Dr[r] = 1.0 / S[L.rowR + r];
    wsync();
    primal_post<NW>(C, PCMD_SCATTER_U);
    primal_grad_hess<NW, false>(C, 0, u, nullptr);  // (cf = the warm-start forces k_rows left in rowF)
    primal_post<NW>(C, PCMD_DOTS_U_JAR);
    primal_row_dots<NW>(C, 0, u, jar, C.bb);
    {
        double c1 = 0.0, c0 = 0.0;
        for (int r = LANE; r < nefc; r += UHC_WAVE) {
            const double x = jar[r], b = C.bb[r], d = Dr[r];
            if (x < 0) c1 = fma(0.5 * d * x, x, c1);
            if (b < 0) c0 = fma(0.5 * d * b, b, c0);
        }
        for (int i = LANE; i < n; i += UHC_WAVE) c1 = fma(0.5 * u[i], u[i], c1);
        c1 = wave_sum(c1); c0 = wave_sum(c0);
        if (!(c1 < c0)) {
            for (int i = LANE; i < n; i += UHC_WAVE) u[i] = 0.0;
            for (int r = LANE; r < nefc; r += UHC_WAVE) jar[r] = C.bb[r];
        }
        wsync();
    }
    PROF(30)
    int it = 0;
    bool ok = false;
    double g0 = -1.0;
    bool have_factor = false;  // H holds the Cholesky factor of the Hessian of act_prev
    unsigned act_prev = 0u;
    for (; it < UHC_PRIMAL_MAXIT; it++) {
        // ---- jar from u itself in every iteration (not jar += alpha p): a row that sits at jar = 0 -- touching, no force -- would otherwise carry the
        //      rounding noise of the updates, change sides from one iteration to the next and keep the "same active set" test from ever holding
        if (it > 0) { primal_post<NW>(C, PCMD_DOTS_U_JAR); primal_row_dots<NW>(C, 0, u, jar, C.bb); }
        PROF(36)
        // ---- active set, gradient coefficients, Hessian weights
        unsigned act = 0u;  // bit h: row LANE + 64 h is active
        for (int r = LANE; r < nefc; r += UHC_WAVE) {
            const double x = jar[r];
            const bool a = x < 0;
            act |= a ? (1u << (r >> 6)) : 0u;
            cf[r] = a ? Dr[r] * x : 0.0;   // gradient coefficient
            pp[r] = a ? Dr[r] : 0.0;       // weight of the row's outer product in the Hessian (pp holds p only after the factorisation)
        }
        // ---- a few rows changed sides since the last factorisation: the factor follows them by rank-one updates (rows that joined) and downdates
        //      (rows that left) instead of a rebuild -- typical of the second and later iterations of a warm-started solve
        if (have_factor) {
            const unsigned flips = act ^ act_prev;
            int nflip = 0;
            for (int h = 0; h * UHC_WAVE < nefc; h++) nflip += __builtin_popcountll(__builtin_amdgcn_ballot_w64((flips >> h) & 1u));
            if (nflip > UHC_PRIMAL_MAXFLIP) have_factor = false;
            double* stY = C.stY;  // (wave 0's strip: a dense copy of the chain row)
            for (int pass = 0; pass < 2 && have_factor; pass++) {  // joins first: the matrix only grows before it shrinks
                for (int h = 0; h * UHC_WAVE < nefc && have_factor; h++) {
                    unsigned long long m = __builtin_amdgcn_ballot_w64(((flips >> h) & 1u) && ((((act >> h) & 1u) != 0u) == (pass == 0)));
                    while (m && have_factor) {
                        const int r = h * UHC_WAVE + __ffsll((long long)m) - 1;
                        m &= m - 1;
                        const RowMisc rm = RM[r];
                        const double sq = sqrt(Dr[r]);
                        DofVec xv = {0.0, 0.0};
                        if (rm.type & ROW_TWO) {
                            const double* Dk = Db + (size_t)(rm.type >> 8) * A.nvp;
                            if (LC.v0) xv.a = sq * Dk[LANE];
                            if (LC.v1) xv.b = sq * Dk[LANE + UHC_WAVE];
                        } else {
                            const int len = RY[r + 1] - RY[r];
                            const short* anc = C.anc_tab + rm.last * YS;
                            for (int i = LANE; i < n; i += UHC_WAVE) stY[i] = 0.0;
                            wsync();
                            if (LANE < len) stY[anc[LANE]] = sq * Yb[RY[r] + LANE];
                            wsync();
                            if (LC.v0) xv.a = stY[LANE];
                            if (LC.v1) xv.b = stY[LANE + UHC_WAVE];
                            wsync();
                        }
                        have_factor = primal_chol_rank1(H, n, xv, pass == 0 ? 1.0 : -1.0);
                    }
                }
            }
        }
        wsync();
        PROF(37)
        // ---- gradient u + sum_active D jar Yhat -- and, without a factor to keep, the Hessian I + sum_active D Yhat Yhat^T in the same pass over the rows
        if (have_factor) { primal_post<NW>(C, PCMD_GRAD); primal_grad_hess<NW, false>(C, 0, vec, u); }
        else { primal_post<NW>(C, PCMD_GRAD_HESS); primal_grad_hess<NW, true>(C, 0, vec, u); }
        PROF(38)
        DofVec x;
        x.a = LC.v0 ? vec[LANE] : 0.0; x.b = LC.v1 ? vec[LANE + UHC_WAVE] : 0.0;
        const double gn = sqrt(wave_sum(x.a * x.a + x.b * x.b));
        if (g0 < 0) g0 = gn;
        PROF(31)
        if (gn <= 1e-14 * g0 || gn == 0.0) { ok = true; break; }
        // ---- Cholesky H = C C^T (primal_chol); the diagonal keeps 1 / C_jj
        if (!have_factor) { primal_post<NW>(C, PCMD_CHOL); primal_chol<NW>(H, n, 0); }
        have_factor = true;
        act_prev = act;
        PROF(26)
        // ---- dir = -H^-1 g: forward substitution column by column (the next column's entries are fetched while this one's step runs), back
        //      substitution row by row (x_j -= C[k][j] x_k for j < k: no reduction on the serial chain); x in registers
        x.a = -x.a; x.b = -x.b;
        {
            const int ib = min(LANE + UHC_WAVE, n - 1);
            int base = 0;  // hcol(k) - k
            double ca = H[base + LANE], cb = H[base + ib], dg = H[base + 0];
            for (int k = 0; k < n; k++) {
                const int nb = base + n - k - 1;  // column k + 1 (one column of slack after the last: reads of the final round stay inside H)
                const int kn = min(k + 1, n - 1), nbase = k + 1 < n ? nb : base;
                const double na = H[nbase + LANE], nbv = H[nbase + ib], ndg = H[nbase + kn];
                const double xk = dv_get_nb(x, k) * dg;
                x.a = LANE == k ? xk : (LANE > k ? fma(-ca, xk, x.a) : x.a);
                x.b = LANE + UHC_WAVE == k ? xk : (LANE + UHC_WAVE > k ? fma(-cb, xk, x.b) : x.b);
                ca = na; cb = nbv; dg = ndg; base = nbase;
            }
            // back substitution: C^T x = y.  Row k of C, entries (k, j), j < k, sits at hcol(j) - j + k: lane = j
            const int ca_adr = hcol(min(LANE, n - 1), n) - min(LANE, n - 1), cb_adr = hcol(ib, n) - ib;  // + k = the lane's entry of row k
            double ra = H[ca_adr + n - 1], rb = H[cb_adr + n - 1], dg2 = H[hcol(n - 1, n)];
            for (int k = n - 1; k >= 0; k--) {
                const int kp = max(k - 1, 0);  // the next step's row, fetched now
                const double na = H[ca_adr + kp], nbv = H[cb_adr + kp], ndg = H[hcol(kp, n)];
                const double xk = dv_get_nb(x, k) * dg2;
                x.a = LANE == k ? xk : (LANE < k ? fma(-ra, xk, x.a) : x.a);
                x.b = LANE + UHC_WAVE == k ? xk : (LANE + UHC_WAVE < k ? fma(-rb, xk, x.b) : x.b);
                ra = na; rb = nbv; dg2 = ndg;
            }
            if (!LC.v0) x.a = 0.0;
            if (!LC.v1) x.b = 0.0;
        }
        if (LC.v0) vec[LANE] = x.a;
        if (LC.v1) vec[LANE + UHC_WAVE] = x.b;
        wsync();
        PROF(27)
        // ---- exact line search along dir: phi'(alpha) = u . dir + alpha |dir|^2 + sum_r D_r min(0, jar_r + alpha p_r) p_r
        primal_post<NW>(C, PCMD_DOTS_DIR_P);
        primal_row_dots<NW>(C, 0, vec, pp, nullptr);
        const double ua = LC.v0 ? u[LANE] : 0.0, ub = LC.v1 ? u[LANE + UHC_WAVE] : 0.0;
        const double lin0 = wave_sum(ua * x.a + ub * x.b), quad = wave_sum(x.a * x.a + x.b * x.b);
        if (quad <= 1e-26 * (1.0 + wave_sum(ua * ua + ub * ub))) { ok = true; it++; break; }  // a step below the rounding of u: converged
        double alpha = 1.0, lo = 0.0, hi = -1.0;
        for (int ls = 0; ls < UHC_PRIMAL_LS_MAXIT; ls++) {
            double d1 = 0.0, d2 = 0.0;
            for (int r = LANE; r < nefc; r += UHC_WAVE) {
                const double p = pp[r], xr = fma(alpha, p, jar[r]), dp = Dr[r] * p;
                if (xr < 0) { d1 = fma(dp, xr, d1); d2 = fma(dp, p, d2); }
            }
            d1 = wave_sum(d1) + fma(alpha, quad, lin0); d2 = wave_sum(d2) + quad;
            if (fabs(d1) <= 1e-15 * (fabs(lin0) + 1e-300)) break;
            if (d1 < 0) lo = alpha; else hi = alpha;
            double nx = alpha - d1 / d2;
            if (!(nx > lo) || (hi > 0 && !(nx < hi))) nx = hi > 0 ? 0.5 * (lo + hi) : 2 * alpha;
            if (nx == alpha) break;
            alpha = nx;
        }
        if (LC.v0) u[LANE] = fma(alpha, x.a, ua);
        if (LC.v1) u[LANE + UHC_WAVE] = fma(alpha, x.b, ub);
        bool same = fabs(alpha - 1.0) <= 1e-12;
        bool flip = false;
        for (int r = LANE; r < nefc; r += UHC_WAVE) {
            const double xr = fma(alpha, pp[r], jar[r]);
            jar[r] = xr;  // (what the final forces are read from when this was the last step; recomputed from u otherwise)
            flip = flip || ((xr < 0) != (((act >> (r >> 6)) & 1u) != 0u));
        }
        wsync();
        PROF(28)
        if (same && !wave_or(flip)) { ok = true; it++; break; }
    }
    // ---- forces (z = u is in place)
    int na = 0;
    for (int r = LANE; r < nefc; r += UHC_WAVE) { const double x = jar[r]; cf[r] = x < 0 ? -Dr[r] * x : 0.0; na += x < 0 ? 1 : 0; }
    for (int o = 32; o > 0; o >>= 1) na += __shfl_xor(na, o);
    *nact = na;
    wsync();
    return ok ? max(it, 1) : -max(it, 1);
}
