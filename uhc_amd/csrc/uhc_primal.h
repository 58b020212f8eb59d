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
//   step    active = {jar < 0};  H (packed lower, column-major, LDS) = I + sum_active D Yhat Yhat^T;  left-looking Cholesky by one wave;
//           dir = -H^-1 g;  p = Yhat dir;  exact line search on the piecewise-quadratic cost (safeguarded Newton on its derivative);
//           u += alpha dir, jar += alpha p
//   stop    a full step (|alpha - 1| <= 1e-12) that leaves the active set as it was, or |g| <= 1e-14 |g_0|;  UHC_PRIMAL_MAXIT otherwise.
// Rows: chain rows lie packed in Yb (offsets rowY, dofs T.dof_anc), rows between two moving bodies as dense nv-vectors in Db (slot = type >> 8).
// Both live in HBM / L2 for this tier (KernelArgs::gY, gD): the LDS is the Hessian's.
#define UHC_PRIMAL_MAXIT 100
#define UHC_PRIMAL_LS_MAXIT 60
#define UHC_PRIMAL_DGROUP 4  // dense rows per pass over the Hessian

// packed lower triangle, column by column: column j holds rows j .. n-1
__device__ __forceinline__ int hcol(int j, int n) { return j * n - (j * (j - 1)) / 2; }

// y = Yhat_r . v for every row (v: an nv-vector in LDS), into out[r] (+ add[r] when add != nullptr); dense rows wave-cooperatively first
template <int TIER>
__device__ __forceinline__ void primal_row_dots(const KernelArgs& A, double* S, int nefc, const LaneConst& LC, const double* Yb, const double* Db,
                                                const double* v, double* out, const double* add) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    const RowMisc* RM = (const RowMisc*)(S + L.rowMisc);
    const int* RY = (const int*)(S + L.rowY);
    const int* NI = (const int*)(S + L.ncon_nefc);
    const int YS = T.maxdepth + 1;
    const int nslot = cap_of<TIER>(A).ndense > 0 ? NI[2] : 0;
    const double va = LC.v0 ? v[LANE] : 0.0, vb = LC.v1 ? v[LANE + UHC_WAVE] : 0.0;
    for (int k = 0; k < nslot; k++) {
        const double* Dk = Db + (size_t)k * A.nvp;
        const double s = wave_sum((LC.v0 ? Dk[LANE] * va : 0.0) + (LC.v1 ? Dk[LANE + UHC_WAVE] * vb : 0.0));
        if (LANE == 0) S[L.dsc + 4 * k] = s;
    }
    wsync();
    for (int r = LANE; r < nefc; r += UHC_WAVE) {
        const RowMisc rm = RM[r];
        double y = add ? add[r] : 0.0;
        if (rm.type & ROW_TWO) y += S[L.dsc + 4 * (rm.type >> 8)];
        else {
            const int len = RY[r + 1] - RY[r];
            const short* anc = T.dof_anc + rm.last * YS;
            const double* Yr = Yb + RY[r];
            for (int q = 0; q < len; q++) y = fma(Yr[q], v[anc[q]], y);
        }
        out[r] = y;
    }
    wsync();
}

// vec += sum_r c_r Yhat_r over the rows with c_r != 0 (c: per-row coefficients in LDS)
template <int TIER>
__device__ __forceinline__ void primal_scatter(const KernelArgs& A, double* S, int nefc, const LaneConst& LC, const double* Yb, const double* Db,
                                               const double* c, double* vec) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    const RowMisc* RM = (const RowMisc*)(S + L.rowMisc);
    const int* RY = (const int*)(S + L.rowY);
    const int* NI = (const int*)(S + L.ncon_nefc);
    const int YS = T.maxdepth + 1;
    const int nslot = cap_of<TIER>(A).ndense > 0 ? NI[2] : 0;
    for (int r = LANE; r < nefc; r += UHC_WAVE) {
        const double cr = c[r];
        if (cr == 0.0) continue;
        const RowMisc rm = RM[r];
        if (rm.type & ROW_TWO) continue;
        const int len = RY[r + 1] - RY[r];
        const short* anc = T.dof_anc + rm.last * YS;
        const double* Yr = Yb + RY[r];
        for (int q = 0; q < len; q++) __hip_atomic_fetch_add(vec + anc[q], cr * Yr[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    wsync();
    double ga = 0.0, gb = 0.0;
    for (int k = 0; k < nslot; k++) {
        const double ck = c[__builtin_amdgcn_readfirstlane(NI[4 + k])];
        if (ck == 0.0) continue;
        const double* Dk = Db + (size_t)k * A.nvp;
        if (LC.v0) ga = fma(ck, Dk[LANE], ga);
        if (LC.v1) gb = fma(ck, Dk[LANE + UHC_WAVE], gb);
    }
    if (LC.v0) vec[LANE] += ga;
    if (LC.v1) vec[LANE + UHC_WAVE] += gb;
    wsync();
}

// returns the Newton iterations taken (>= 1), negated when the iteration cap was reached; z = u in S[L.z], the forces in S[L.rowF]
template <int TIER>
__device__ __forceinline__ int k_primal(const KernelArgs& A, const double* mb, double* S, int nefc, const LaneConst& LC, const double* Yb, const double* Db PROF_ARGS) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    const RowMisc* RM = (const RowMisc*)(S + L.rowMisc);
    const int* RY = (const int*)(S + L.rowY);
    const int* NI = (const int*)(S + L.ncon_nefc);
    const int YS = T.maxdepth + 1, n = T.nv;
    const int nslot = cap_of<TIER>(A).ndense > 0 ? NI[2] : 0;
    double* u = S + L.z;
    double* vec = S + L.vec;     // gradient, then the Newton direction
    double* jar = S + L.rowAref;  // Yhat_r . u + b_r  (the slot held D jar of the warm start for the working sets' ranking: not used in this tier)
    double* pp = S + L.rowDa;     // Yhat_r . dir      (the slot held diag(A): only the sweeps read it, and they do not run after this)
    double* Dr = S + L.rowW;      // 1 / R_r
    double* cf = S + L.rowF;      // per-row coefficient of the current scatter; the forces at the end
    double* H = S + L.H;
    // ---- start point: u0 = sum f_ws Yhat, kept if its cost is below the cost of u = 0
    for (int i = LANE; i < n; i += UHC_WAVE) u[i] = 0.0;
    for (int r = LANE; r < nefc; r += UHC_WAVE) Dr[r] = 1.0 / S[L.rowR + r];
    wsync();
    primal_scatter<TIER>(A, S, nefc, LC, Yb, Db, cf, u);  // (cf = the warm-start forces k_rows left in rowF)
    primal_row_dots<TIER>(A, S, nefc, LC, Yb, Db, u, jar, S + L.rowB);
    {
        double c1 = 0.0, c0 = 0.0;
        for (int r = LANE; r < nefc; r += UHC_WAVE) {
            const double x = jar[r], b = S[L.rowB + r], d = Dr[r];
            if (x < 0) c1 = fma(0.5 * d * x, x, c1);
            if (b < 0) c0 = fma(0.5 * d * b, b, c0);
        }
        for (int i = LANE; i < n; i += UHC_WAVE) c1 = fma(0.5 * u[i], u[i], c1);
        c1 = wave_sum(c1); c0 = wave_sum(c0);
        if (!(c1 < c0)) {
            for (int i = LANE; i < n; i += UHC_WAVE) u[i] = 0.0;
            for (int r = LANE; r < nefc; r += UHC_WAVE) jar[r] = S[L.rowB + r];
        }
        wsync();
    }
    PROF(30)
    int it = 0;
    bool ok = false;
    double g0 = -1.0;
    for (; it < UHC_PRIMAL_MAXIT; it++) {
        // ---- jar from u itself in every iteration (not jar += alpha p): a row that sits at jar = 0 -- touching, no force -- would otherwise carry the
        //      rounding noise of the updates, change sides from one iteration to the next and keep the "same active set" test from ever holding
        if (it > 0) primal_row_dots<TIER>(A, S, nefc, LC, Yb, Db, u, jar, S + L.rowB);
        // ---- active set, gradient
        unsigned act = 0u;  // bit h: row LANE + 64 h is active
        for (int r = LANE; r < nefc; r += UHC_WAVE) {
            const double x = jar[r];
            const bool a = x < 0;
            act |= a ? (1u << (r >> 6)) : 0u;
            cf[r] = a ? Dr[r] * x : 0.0;
        }
        for (int i = LANE; i < n; i += UHC_WAVE) vec[i] = u[i];
        wsync();
        primal_scatter<TIER>(A, S, nefc, LC, Yb, Db, cf, vec);
        DofVec x;
        x.a = LC.v0 ? vec[LANE] : 0.0; x.b = LC.v1 ? vec[LANE + UHC_WAVE] : 0.0;
        const double gn = sqrt(wave_sum(x.a * x.a + x.b * x.b));
        if (g0 < 0) g0 = gn;
        if (gn <= 1e-14 * g0 || gn == 0.0) { ok = true; break; }
        // ---- Hessian: identity + the active rows' outer products
        const int nH = (n * (n + 1)) / 2;
        for (int e = LANE; e < nH; e += UHC_WAVE) H[e] = 0.0;
        wsync();
        for (int j = LANE; j < n; j += UHC_WAVE) H[hcol(j, n)] = 1.0;
        wsync();
        for (int r = LANE; r < nefc; r += UHC_WAVE) {  // chain rows: lane = row; entries (anc[q], anc[q2]), q2 <= q, of column anc[q2]
            if (!((act >> (r >> 6)) & 1u)) continue;
            const RowMisc rm = RM[r];
            if (rm.type & ROW_TWO) continue;
            const int len = RY[r + 1] - RY[r];
            const short* anc = T.dof_anc + rm.last * YS;
            const double* Yr = Yb + RY[r];
            const double d = Dr[r];
            for (int q2 = 0; q2 < len; q2++) {
                const int j = anc[q2];
                const double dj = d * Yr[q2];
                double* Hc = H + hcol(j, n) - j;
                for (int q = q2; q < len; q++) __hip_atomic_fetch_add(Hc + anc[q], dj * Yr[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        wsync();
        {   // dense rows, UHC_PRIMAL_DGROUP at a time: lane = row index i of the Hessian, columns j in turn (y_j by readlane, the column contiguous in i)
            int k = 0;
            while (k < nslot) {
                DofVec y[UHC_PRIMAL_DGROUP];
                double dk[UHC_PRIMAL_DGROUP];
                int got = 0;
#pragma unroll
                for (int s = 0; s < UHC_PRIMAL_DGROUP; s++) { y[s].a = y[s].b = 0.0; dk[s] = 0.0; }
                while (k < nslot && got < UHC_PRIMAL_DGROUP) {
                    const int rid = __builtin_amdgcn_readfirstlane(NI[4 + k]);
                    const unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)act, rid & 63);
                    if ((bits >> (rid >> 6)) & 1u) {
                        const double* Dk = Db + (size_t)k * A.nvp;
                        const double ya = LC.v0 ? Dk[LANE] : 0.0, yb = LC.v1 ? Dk[LANE + UHC_WAVE] : 0.0, dd = Dr[rid];
#pragma unroll
                        for (int s = 0; s < UHC_PRIMAL_DGROUP; s++) if (s == got) { y[s].a = ya; y[s].b = yb; dk[s] = dd; }
                        got++;
                    }
                    k++;
                }
                if (got == 0) break;
                for (int j = 0; j < n; j++) {
                    double yj[UHC_PRIMAL_DGROUP];
#pragma unroll
                    for (int s = 0; s < UHC_PRIMAL_DGROUP; s++) yj[s] = dk[s] * dv_get_nb(y[s], j);
                    double* Hc = H + hcol(j, n) - j;
                    if (LANE >= j && LC.v0) {
                        double h = Hc[LANE];
#pragma unroll
                        for (int s = 0; s < UHC_PRIMAL_DGROUP; s++) h = fma(yj[s], y[s].a, h);
                        Hc[LANE] = h;
                    }
                    if (LANE + UHC_WAVE >= j && LC.v1) {
                        double h = Hc[LANE + UHC_WAVE];
#pragma unroll
                        for (int s = 0; s < UHC_PRIMAL_DGROUP; s++) h = fma(yj[s], y[s].b, h);
                        Hc[LANE + UHC_WAVE] = h;
                    }
                }
            }
            wsync();
        }
        PROF(31)
        // ---- left-looking Cholesky H = C C^T, column by column; the diagonal keeps 1 / C_jj
        for (int j = 0; j < n; j++) {
            const double* Hj = H + hcol(j, n) - j;
            double va = (LC.v0 && LANE >= j) ? Hj[LANE] : 0.0;
            double vb = (LC.v1 && LANE + UHC_WAVE >= j) ? Hj[LANE + UHC_WAVE] : 0.0;
            double wa = 0.0, wb = 0.0;  // second accumulator: two independent FMA chains
            int k = 0;
            for (; k + 1 < j; k += 2) {
                const double* C0 = H + hcol(k, n) - k;
                const double* C1 = H + hcol(k + 1, n) - (k + 1);
                const double c0 = C0[j], c1 = C1[j];
                if (LC.v0 && LANE >= j) { va = fma(-C0[LANE], c0, va); wa = fma(-C1[LANE], c1, wa); }
                if (LC.v1 && LANE + UHC_WAVE >= j) { vb = fma(-C0[LANE + UHC_WAVE], c0, vb); wb = fma(-C1[LANE + UHC_WAVE], c1, wb); }
            }
            if (k < j) {
                const double* C0 = H + hcol(k, n) - k;
                const double c0 = C0[j];
                if (LC.v0 && LANE >= j) va = fma(-C0[LANE], c0, va);
                if (LC.v1 && LANE + UHC_WAVE >= j) vb = fma(-C0[LANE + UHC_WAVE], c0, vb);
            }
            va += wa; vb += wb;
            DofVec col = {va, vb};
            const double djj = dv_get_nb(col, j);
            const double rc = 1.0 / sqrt(djj);  // (H >= I: djj >= 1 up to rounding)
            double* Hw = H + hcol(j, n) - j;
            if (LC.v0 && LANE >= j) Hw[LANE] = LANE == j ? rc : va * rc;
            if (LC.v1 && LANE + UHC_WAVE >= j) Hw[LANE + UHC_WAVE] = LANE + UHC_WAVE == j ? rc : vb * rc;
            wsync();  // (the next column reads what other lanes wrote here)
        }
        wsync();
        PROF(26)
        // ---- dir = -H^-1 g: forward substitution column by column, back substitution with a wave reduction per column; x in registers
        x.a = -x.a; x.b = -x.b;
        for (int k = 0; k < n; k++) {
            const double* Ck = H + hcol(k, n) - k;
            const double xk = dv_get_nb(x, k) * Ck[k];
            if (LC.v0) x.a = LANE == k ? xk : (LANE > k ? fma(-Ck[LANE], xk, x.a) : x.a);
            if (LC.v1) x.b = LANE + UHC_WAVE == k ? xk : (LANE + UHC_WAVE > k ? fma(-Ck[LANE + UHC_WAVE], xk, x.b) : x.b);
        }
        for (int k = n - 1; k >= 0; k--) {
            const double* Ck = H + hcol(k, n) - k;
            const double s = wave_sum(((LC.v0 && LANE > k) ? Ck[LANE] * x.a : 0.0) + ((LC.v1 && LANE + UHC_WAVE > k) ? Ck[LANE + UHC_WAVE] * x.b : 0.0));
            const double xk = (dv_get_nb(x, k) - s) * Ck[k];
            if (LANE == k) x.a = xk;
            if (LANE + UHC_WAVE == k) x.b = xk;
        }
        if (LC.v0) vec[LANE] = x.a;
        if (LC.v1) vec[LANE + UHC_WAVE] = x.b;
        wsync();
        PROF(27)
        // ---- exact line search along dir: phi'(alpha) = u . dir + alpha |dir|^2 + sum_r D_r min(0, jar_r + alpha p_r) p_r
        primal_row_dots<TIER>(A, S, nefc, LC, Yb, Db, vec, pp, nullptr);
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
    for (int r = LANE; r < nefc; r += UHC_WAVE) { const double x = jar[r]; cf[r] = x < 0 ? -Dr[r] * x : 0.0; }
    wsync();
    return ok ? max(it, 1) : -max(it, 1);
}
