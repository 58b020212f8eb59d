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
#define UHC_PRIMAL_DGROUP 8  // dense rows per pass over the Hessian
#define UHC_PRIMAL_MAXFLIP 10  // rows that may change sides between two iterations for the factor to be updated instead of rebuilt

// packed lower triangle, column by column: column j holds rows j .. n-1
__device__ __forceinline__ int hcol(int j, int n) { return j * n - (j * (j - 1)) / 2; }

// y = Yhat_r . v for every row (v: an nv-vector in LDS), into out[r] (+ add[r] when add != nullptr); dense rows wave-cooperatively first.
// anc: the dof-chain table ([nv][YS] shorts), staged in LDS by k_primal (the rows' own entries stream from L2: four independent loads per round --
// a loop of dependent single loads pays the L2 latency once per entry)
template <int TIER>
__device__ __forceinline__ void primal_row_dots(const KernelArgs& A, double* S, int nefc, const LaneConst& LC, const double* Yb, const double* Db,
                                                const short* anc_tab, const double* v, double* out, const double* add) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    const RowMisc* RM = (const RowMisc*)(S + L.rowMisc);
    const int* RY = (const int*)(S + L.rowY);
    const int* NI = (const int*)(S + L.ncon_nefc);
    const int YS = T.maxdepth + 1;
    const int nslot = cap_of<TIER>(A).ndense > 0 ? NI[2] : 0;
    const double va = LC.v0 ? v[LANE] : 0.0, vb = LC.v1 ? v[LANE + UHC_WAVE] : 0.0;
    for (int k0 = 0; k0 < nslot; k0 += 4) {  // four dense rows per round: their loads and reductions overlap
        double s[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const double* Dk = Db + (size_t)min(k0 + j, nslot - 1) * A.nvp;
            s[j] = (LC.v0 ? Dk[LANE] * va : 0.0) + (LC.v1 ? Dk[LANE + UHC_WAVE] * vb : 0.0);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) s[j] = wave_sum(s[j]);
#pragma unroll
        for (int j = 0; j < 4; j++) if (LANE == 0 && k0 + j < nslot) S[L.dsc + 4 * (k0 + j)] = s[j];
    }
    wsync();
    for (int r = LANE; r < nefc; r += UHC_WAVE) {
        const RowMisc rm = RM[r];
        double y = add ? add[r] : 0.0;
        if (rm.type & ROW_TWO) y += S[L.dsc + 4 * (rm.type >> 8)];
        else {
            const int len = RY[r + 1] - RY[r];
            const short* anc = anc_tab + rm.last * YS;
            const double* Yr = Yb + RY[r];
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
    wsync();
}

// The chain rows' share of the gradient and of the Hessian, WITHOUT atomics.  (A first version let every lane push its own row's len^2 / 2 products
// into H with LDS float64 atomics: measured at ~12 cycles per lane and instruction, 1.3 M cycles per Hessian -- hidden in the profile behind the
// Cholesky that had to wait for the LDS queue to drain.)  Rows that share a dof chain -- the 4 pyramid edges of a contact, all contacts of one
// hull: runs of up to 16 consecutive rows with the same last dof -- are taken TOGETHER by the whole wave: their entries are staged in LDS
// ([16][32]), then lane = chain position q adds sum_t c_t y_t[q] to vec[dof(q)], and lane = pair (q, q2) of the chain's len (len + 1) / 2 pairs adds
// sum_t w_t y_t[q] y_t[q2] to H[dof(q)][dof(q2)] -- distinct addresses inside a run, and the LDS queue of the wave keeps runs in order: plain
// read-modify-write.  c: per-row coefficients (0 = the row does not contribute), w: per-row weights of the outer products (WITH_H).
template <int TIER, bool WITH_H>
__device__ __forceinline__ void primal_chain_pass(const KernelArgs& A, double* S, int nefc, const double* Yb, const short* anc_tab, const unsigned short* pair_tab,
                                                  double* stY, double* cw, const double* c, const double* w, double* vec, double* H, int n) {
    const DevTopo& T = A.t;
    const DevLds& L = lds_of<TIER>(A);
    const RowMisc* RM = (const RowMisc*)(S + L.rowMisc);
    const int* RY = (const int*)(S + L.rowY);
    const int YS = T.maxdepth + 1;
    int r0 = 0;
#ifdef UHC_PRIMAL_GUARD
    int guard = 0;
#endif
    while (r0 < nefc) {
#ifdef UHC_PRIMAL_GUARD
        if (++guard > 4096) { if (LANE == 0) printf("primal_chain_pass: no progress at row %d of %d\n", r0, nefc); break; }
#endif
        const int rr = r0 + LANE;
        int last_l = -2, two_l = 1;
        if (LANE < 16 && rr < nefc) { const RowMisc rm = RM[rr]; last_l = rm.last; two_l = (rm.type & ROW_TWO) ? 1 : 0; }
        const int last0 = __builtin_amdgcn_readfirstlane(last_l);
        if (__builtin_amdgcn_readfirstlane(two_l)) { r0++; continue; }  // (dense rows: primal_dense_*)
        const unsigned long long same = __builtin_amdgcn_ballot_w64(LANE < 16 && rr < nefc && !two_l && last_l == last0);
        const int nb = __builtin_ctzll(~same);  // the run's length: consecutive rows from r0 on with this chain (>= 1, <= 16)
        double c_l = 0.0, w_l = 0.0;
        if (LANE < nb) { c_l = c[rr]; if (WITH_H) w_l = w[rr]; }
        if (__builtin_amdgcn_ballot_w64(c_l != 0.0 || w_l != 0.0)) {
            const int len = __builtin_amdgcn_readfirstlane(RY[r0 + 1] - RY[r0]);
            for (int idx = LANE; idx < nb * 32; idx += UHC_WAVE) {
                const int t = idx >> 5, q = idx & 31;
                if (q < len) stY[idx] = Yb[RY[r0 + t] + q];
            }
            // (the run's coefficients go through LDS, not v_readlane: the readers below sit in divergent branches -- lanes < len, lanes < np -- and a
            //  register the compiler spilled and reloads INSIDE such a branch holds garbage in the lanes the branch switched off, which are exactly the
            //  lanes a readlane of row t >= len would read.  The 552-spill instantiation <0, 3> did that and produced NaNs; <1, 3>, 6 spills, did not.)
            if (LANE < 16) { cw[LANE] = c_l; cw[16 + LANE] = w_l; }
            wsync();
            const short* anc = anc_tab + last0 * YS;
            if (LANE < len) {
                double sg = 0.0;
                for (int t = 0; t < nb; t++) sg = fma(cw[t], stY[t * 32 + LANE], sg);
                vec[anc[LANE]] += sg;
            }
            if (WITH_H) {
                const int np = (len * (len + 1)) / 2;
                for (int p0 = 0; p0 < np; p0 += UHC_WAVE) {
                    const int pi = p0 + LANE;
                    if (pi < np) {
                        const int qq = pair_tab[pi], q = qq >> 8, q2 = qq & 0xff;
                        double sh = 0.0;
                        for (int t = 0; t < nb; t++) sh = fma(cw[16 + t] * stY[t * 32 + q], stY[t * 32 + q2], sh);
                        const int i = anc[q], jc = anc[q2];
                        H[hcol(jc, n) - jc + i] += sh;
                    }
                }
            }
            wsync();
        }
        r0 += nb;
    }
}
// the dense rows' share of a scatter: vec += sum_k c_k Yhat_k, lane = dof
template <int TIER>
__device__ __forceinline__ void primal_dense_scatter(const KernelArgs& A, double* S, const LaneConst& LC, const double* Db, const double* c, double* vec) {
    const DevLds& L = lds_of<TIER>(A);
    const int* NI = (const int*)(S + L.ncon_nefc);
    const int nslot = cap_of<TIER>(A).ndense > 0 ? NI[2] : 0;
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

// Two columns (j, j + 1) of the left-looking Cholesky factorisation of the packed Hessian: lane = row (rows LANE and LANE + 64), the two columns'
// running values in registers, every earlier column k read ONCE for both (its rows: two vector reads; its entries (j, k), (j + 1, k): two broadcast
// reads), four columns per round so that eight reads are in flight before the first FMA needs one -- a loop of one column per round is bound by the
// LDS round trip (measured: 190 cycles per column and (j, k) pair, 1.3 M cycles per factorisation at nv = 117: 64 % of a tier-4 env-step).  LO: the
// pair still has rows below 64 (j < 64); later pairs skip that half.  Rows above the diagonal compute garbage from in-range reads and are not stored.
template <bool LO>
__device__ __forceinline__ void primal_chol2(double* H, int n, int j) {
    const int ia = LANE, ib = min(LANE + UHC_WAVE, n - 1), j1 = min(j + 1, n - 1);
    const bool two = j + 1 < n;
    double* Hj = H + hcol(j, n) - j;
    double* Hj1 = H + hcol(j1, n) - j1;
    double a0 = LO ? Hj[ia] : 0.0, b0 = Hj[ib], a1 = LO ? Hj1[ia] : 0.0, b1 = Hj1[ib];
    double a0x = 0.0, b0x = 0.0, a1x = 0.0, b1x = 0.0;  // second accumulators: independent FMA chains
    int base = 0, k = 0;  // base = hcol(k) - k
    for (; k + 4 <= j; k += 4) {
        const int o0 = base, o1 = o0 + n - k - 1, o2 = o1 + n - k - 2, o3 = o2 + n - k - 3;
        base = o3 + n - k - 4;
        const double c0 = H[o0 + j], c1 = H[o1 + j], c2 = H[o2 + j], c3 = H[o3 + j];
        const double d0 = H[o0 + j1], d1 = H[o1 + j1], d2 = H[o2 + j1], d3 = H[o3 + j1];
        const double q0 = H[o0 + ib], q1 = H[o1 + ib], q2 = H[o2 + ib], q3 = H[o3 + ib];
        if (LO) {
            const double p0 = H[o0 + ia], p1 = H[o1 + ia], p2 = H[o2 + ia], p3 = H[o3 + ia];
            a0 = fma(-p0, c0, a0); a0x = fma(-p1, c1, a0x); a0 = fma(-p2, c2, a0); a0x = fma(-p3, c3, a0x);
            a1 = fma(-p0, d0, a1); a1x = fma(-p1, d1, a1x); a1 = fma(-p2, d2, a1); a1x = fma(-p3, d3, a1x);
        }
        b0 = fma(-q0, c0, b0); b0x = fma(-q1, c1, b0x); b0 = fma(-q2, c2, b0); b0x = fma(-q3, c3, b0x);
        b1 = fma(-q0, d0, b1); b1x = fma(-q1, d1, b1x); b1 = fma(-q2, d2, b1); b1x = fma(-q3, d3, b1x);
    }
    for (; k < j; k++) {
        const int o0 = base;
        base = o0 + n - k - 1;
        const double c0 = H[o0 + j], d0 = H[o0 + j1], q0 = H[o0 + ib];
        if (LO) { const double p0 = H[o0 + ia]; a0 = fma(-p0, c0, a0); a1 = fma(-p0, d0, a1); }
        b0 = fma(-q0, c0, b0); b1 = fma(-q0, d0, b1);
    }
    a0 += a0x; b0 += b0x; a1 += a1x; b1 += b1x;
    // column j: scale by 1 / sqrt of its diagonal; column j + 1: one more update by the finished column j, then the same
    DofVec v0 = {a0, b0};
    const double rc0 = 1.0 / sqrt(dv_get_nb(v0, j));  // (H >= I: the pivot is >= 1 up to rounding)
    v0.a *= rc0; v0.b *= rc0;
    const double cj1 = dv_get_nb(v0, j1);  // C[j + 1][j]
    DofVec v1 = {fma(-v0.a, cj1, a1), fma(-v0.b, cj1, b1)};
    const double rc1 = 1.0 / sqrt(dv_get_nb(v1, j1));
    v1.a *= rc1; v1.b *= rc1;
    if (LO && ia >= j && ia < n) Hj[ia] = ia == j ? rc0 : v0.a;
    if (LANE + UHC_WAVE >= j && LANE + UHC_WAVE < n) Hj[LANE + UHC_WAVE] = LANE + UHC_WAVE == j ? rc0 : v0.b;
    if (two) {
        if (LO && ia >= j1 && ia < n) Hj1[ia] = ia == j1 ? rc1 : v1.a;
        if (LANE + UHC_WAVE >= j1 && LANE + UHC_WAVE < n) Hj1[LANE + UHC_WAVE] = LANE + UHC_WAVE == j1 ? rc1 : v1.b;
    }
}

// Rank-one update (sigma = +1) or downdate (-1) of the packed Cholesky factor, C C^T <- C C^T + sigma x x^T, column by column (the LINPACK rotation
// scheme): lane = row, x in registers, one column read and written per step.  ~150 cycles per column against a fresh factorisation's 3 000: when a
// Newton iteration moves a handful of rows across jar = 0, the factor follows them instead of being rebuilt.  Returns false when a downdate
// loses positive definiteness to rounding (r^2 <= 0): the caller rebuilds.  The diagonal holds 1 / C_kk throughout.
__device__ __forceinline__ bool primal_chol_rank1(double* H, int n, DofVec x, double sigma) {
    const int ib = min(LANE + UHC_WAVE, n - 1);
    bool good = true;
    int base = 0;  // hcol(k) - k
    for (int k = 0; k < n; k++) {
        const double ca = H[base + LANE], cb = H[base + ib], rd = H[base + k];  // column k (rows above the diagonal: in-range garbage, never stored)
        const double xk = dv_get_nb(x, k);
        const double ckk = 1.0 / rd;
        const double r2 = fma(sigma * xk, xk, ckk * ckk);
        good = good && (r2 > 0.0);
        const double r = sqrt(r2 > 0.0 ? r2 : 1.0);
        const double ic = ckk / r, sg = sigma * xk * rd;  // 1 / c,  sigma s
        const double sx = xk * rd, c = r * rd;            // s, c
        const double na = (ca + sg * x.a) * ic, nb = (cb + sg * x.b) * ic;
        if (LANE > k && LANE < n) H[base + LANE] = na;
        if (LANE + UHC_WAVE > k && LANE + UHC_WAVE < n) H[base + LANE + UHC_WAVE] = nb;
        if (LANE == (k & 63)) { if (k < UHC_WAVE) H[base + k] = 1.0 / r; }
        if (k >= UHC_WAVE && LANE + UHC_WAVE == k) H[base + k] = 1.0 / r;
        x.a = fma(c, x.a, -sx * na);
        x.b = fma(c, x.b, -sx * nb);
        base += n - k - 1;
    }
    wsync();
    return good;
}

// returns the Newton iterations taken (>= 1), negated when the iteration cap was reached; z = u in S[L.z], the forces in S[L.rowF]
template <int TIER>
__device__ __forceinline__ int k_primal(const KernelArgs& A, const double* mb, double* S, int nefc, const LaneConst& LC, const double* Yb, const double* Db, int* nact PROF_ARGS) {
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
    // the dof chains (which dof sits at position q of the chain that ends in dof i), from L2 into the contacts' storage: nothing reads the contacts
    // once the rows are built, and every row pass of every Newton iteration walks this table
    // (the host sizes the contacts' storage for all three: table, the lanes' row strips, the dense group -- uhc_capi.cpp huge_layout)
    short* anc_l = (short*)(S + L.con);
    for (int i = LANE; i < n * YS; i += UHC_WAVE) anc_l[i] = T.dof_anc[i];
    const short* anc_tab = anc_l;
    double* stY = S + L.con + ((n * YS + 3) / 4 + 1);                   // [16][32]: the run of chain rows being added (primal_chain_pass)
    double* dstage = stY + 16 * 32;                                     // [nv][UHC_PRIMAL_DGROUP]: D y of the dense rows being added, dof-major
    unsigned short* pair_tab = (unsigned short*)(dstage + n * UHC_PRIMAL_DGROUP + 1);  // [496] pair p = q (q + 1) / 2 + q2 -> (q << 8 | q2), q2 <= q < 32
    double* cw = (double*)(pair_tab + 496);                             // [2][16]: coefficients and weights of the run of rows being added
    if (LANE < 32) for (int q2 = 0; q2 <= LANE; q2++) pair_tab[(LANE * (LANE + 1)) / 2 + q2] = (unsigned short)((LANE << 8) | q2);
    wsync();
    // ---- start point: u0 = sum f_ws Yhat, kept if its cost is below the cost of u = 0
    for (int i = LANE; i < n; i += UHC_WAVE) u[i] = 0.0;
    for (int r = LANE; r < nefc; r += UHC_WAVE) Dr[r] = 1.0 / S[L.rowR + r];
    wsync();
    primal_chain_pass<TIER, false>(A, S, nefc, Yb, anc_tab, pair_tab, stY, cw, cf, nullptr, u, H, n);  // (cf = the warm-start forces k_rows left in rowF)
    primal_dense_scatter<TIER>(A, S, LC, Db, cf, u);
    primal_row_dots<TIER>(A, S, nefc, LC, Yb, Db, anc_tab, u, jar, S + L.rowB);
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
    bool have_factor = false;  // H holds the Cholesky factor of the Hessian of act_prev
    unsigned act_prev = 0u;
    for (; it < UHC_PRIMAL_MAXIT; it++) {
        // ---- jar from u itself in every iteration (not jar += alpha p): a row that sits at jar = 0 -- touching, no force -- would otherwise carry the
        //      rounding noise of the updates, change sides from one iteration to the next and keep the "same active set" test from ever holding
        if (it > 0) primal_row_dots<TIER>(A, S, nefc, LC, Yb, Db, anc_tab, u, jar, S + L.rowB);
        // ---- active set, gradient
        unsigned act = 0u;  // bit h: row LANE + 64 h is active
        for (int r = LANE; r < nefc; r += UHC_WAVE) {
            const double x = jar[r];
            const bool a = x < 0;
            act |= a ? (1u << (r >> 6)) : 0u;
            cf[r] = a ? Dr[r] * x : 0.0;   // gradient coefficient
            pp[r] = a ? Dr[r] : 0.0;       // weight of the row's outer product in the Hessian (pp holds p only after the factorisation)
        }
        for (int i = LANE; i < n; i += UHC_WAVE) vec[i] = u[i];
        // ---- a few rows changed sides since the last factorisation: the factor follows them by rank-one updates (rows that joined) and downdates
        //      (rows that left) instead of a rebuild -- typical of the second and later iterations of a warm-started solve
        if (have_factor) {
            const unsigned flips = act ^ act_prev;
            int nflip = 0;
            for (int h = 0; h * UHC_WAVE < nefc; h++) nflip += __builtin_popcountll(__builtin_amdgcn_ballot_w64((flips >> h) & 1u));
            if (nflip > UHC_PRIMAL_MAXFLIP) have_factor = false;
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
                            const short* anc = anc_tab + rm.last * YS;
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
        if (have_factor) {  // gradient alone
            primal_chain_pass<TIER, false>(A, S, nefc, Yb, anc_tab, pair_tab, stY, cw, cf, nullptr, vec, H, n);
            primal_dense_scatter<TIER>(A, S, LC, Db, cf, vec);
        } else {
        // ---- Hessian: identity + the active rows' outer products; gradient: u + sum_active D jar Yhat -- one pass over the rows for both
        const int nH = (n * (n + 1)) / 2;
        for (int e = LANE; e < nH; e += UHC_WAVE) H[e] = 0.0;
        wsync();
        for (int j = LANE; j < n; j += UHC_WAVE) H[hcol(j, n)] = 1.0;
        wsync();
        primal_chain_pass<TIER, true>(A, S, nefc, Yb, anc_tab, pair_tab, stY, cw, cf, pp, vec, H, n);
        primal_dense_scatter<TIER>(A, S, LC, Db, cf, vec);
        {   // dense rows, UHC_PRIMAL_DGROUP at a time: lane = row index i of the Hessian (its own y_i of the group's rows in registers), columns j in
            // turn: the column is contiguous in i, D y_j of every row of the group comes by one broadcast LDS read from the staged copy
            int k = 0;
#ifdef UHC_PRIMAL_GUARD
            int guard2 = 0;
#endif
            while (k < nslot) {
#ifdef UHC_PRIMAL_GUARD
                if (++guard2 > 1024) { if (LANE == 0) printf("k_primal: dense loop stuck at slot %d of %d\n", k, nslot); break; }
#endif
                DofVec y[UHC_PRIMAL_DGROUP];
                int got = 0;
#pragma unroll
                for (int s2 = 0; s2 < UHC_PRIMAL_DGROUP; s2++) { y[s2].a = y[s2].b = 0.0; }
                while (k < nslot && got < UHC_PRIMAL_DGROUP) {
                    const int rid = __builtin_amdgcn_readfirstlane(NI[4 + k]);
                    const unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)act, rid & 63);
                    if ((bits >> (rid >> 6)) & 1u) {
                        const double* Dk = Db + (size_t)k * A.nvp;
                        const double ya = LC.v0 ? Dk[LANE] : 0.0, yb = LC.v1 ? Dk[LANE + UHC_WAVE] : 0.0, dd = Dr[rid];
#pragma unroll
                        for (int s2 = 0; s2 < UHC_PRIMAL_DGROUP; s2++) if (s2 == got) { y[s2].a = ya; y[s2].b = yb; }
                        if (LC.v0) dstage[LANE * UHC_PRIMAL_DGROUP + got] = dd * ya;   // [dof][row of the group]: one 64-byte line per column j
                        if (LC.v1) dstage[(LANE + UHC_WAVE) * UHC_PRIMAL_DGROUP + got] = dd * yb;
                        got++;
                    }
                    k++;
                }
                if (got == 0) break;
                for (int g = got; g < UHC_PRIMAL_DGROUP; g++) {  // (unused places of the last group: zero multipliers)
                    if (LC.v0) dstage[LANE * UHC_PRIMAL_DGROUP + g] = 0.0;
                    if (LC.v1) dstage[(LANE + UHC_WAVE) * UHC_PRIMAL_DGROUP + g] = 0.0;
                }
                wsync();
                const int ib = min(LANE + UHC_WAVE, n - 1);
                int base = 0;  // hcol(j) - j
                for (int j = 0; j < n; j++) {
                    double yj[UHC_PRIMAL_DGROUP];
#pragma unroll
                    for (int s2 = 0; s2 < UHC_PRIMAL_DGROUP; s2++) yj[s2] = dstage[j * UHC_PRIMAL_DGROUP + s2];
                    double ha = H[base + LANE], hb = H[base + ib];
#pragma unroll
                    for (int s2 = 0; s2 < UHC_PRIMAL_DGROUP; s2++) { ha = fma(yj[s2], y[s2].a, ha); hb = fma(yj[s2], y[s2].b, hb); }
                    if (LANE >= j && LC.v0) H[base + LANE] = ha;
                    if (LANE + UHC_WAVE >= j && LC.v1) H[base + LANE + UHC_WAVE] = hb;
                    base += n - j - 1;
                }
                wsync();
            }
        }
        }
        DofVec x;
        x.a = LC.v0 ? vec[LANE] : 0.0; x.b = LC.v1 ? vec[LANE + UHC_WAVE] : 0.0;
        const double gn = sqrt(wave_sum(x.a * x.a + x.b * x.b));
        if (g0 < 0) g0 = gn;
        PROF(31)
        if (gn <= 1e-14 * g0 || gn == 0.0) { ok = true; break; }
        // ---- left-looking Cholesky H = C C^T, two columns per pass (primal_chol2); the diagonal keeps 1 / C_jj
        if (!have_factor)
            for (int j = 0; j < n; j += 2) {
                if (j < UHC_WAVE) primal_chol2<true>(H, n, j); else primal_chol2<false>(H, n, j);
                wsync();  // (the next pair reads what other lanes wrote here)
            }
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
        primal_row_dots<TIER>(A, S, nefc, LC, Yb, Db, anc_tab, vec, pp, nullptr);
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
#ifdef UHC_PRIMAL_GUARD
    if (LANE == 0 && (it >= 20 || !ok)) printf("k_primal: env block %d nefc %d nslot %d: %d iterations, ok %d, g0 %.3e\n", (int)blockIdx.x, nefc, nslot, it, (int)ok, g0);
#endif
    // ---- forces (z = u is in place)
    int na = 0;
    for (int r = LANE; r < nefc; r += UHC_WAVE) { const double x = jar[r]; cf[r] = x < 0 ? -Dr[r] * x : 0.0; na += x < 0 ? 1 : 0; }
    for (int o = 32; o > 0; o >>= 1) na += __shfl_xor(na, o);
    *nact = na;
    wsync();
    return ok ? max(it, 1) : -max(it, 1);
}
