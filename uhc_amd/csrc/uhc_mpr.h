// uhc_mpr.h -- convex-convex narrow phase for the fused step kernel: Minkowski Portal Refinement, ONE CANDIDATE PAIR PER LANE.
//
// What it replaces: the mesh-mesh branch of MuJoCo's mj_collision behind self.sim.step() (uhc/envs/humanoid_im.py:1177): generated
// SMPL models collide every body hull with every other (uhc/smpllib/smpl_parser.py:327-328, excludes at uhc/smpllib/smpl_robot.py:
// 1177-1198), which MuJoCo 2.1 resolves with libccd's ccdMPRPenetration [MJ-ext].  the tests check this file against a CPU restatement of the same algorithm; both follow libccd's structure (discoverPortal / refinePortal / findPenetr / findPos)
// and its zero / equality tests so that the same portal is found.
//
// Mapping to the hardware: the algorithm is a short, branchy, strictly serial refinement whose only wide operation is the support
// function (arg-max of <= 64 dot products per hull).  A wave-cooperative support costs two DPP reductions per call and leaves the
// serial part replicated in 64 lanes; with ~250 statically possible hull pairs per humanoid and 10-40 of them passing the
// bounding-sphere cull every substep, lane = pair is the better use of the wave: every lane walks its own two hulls (L2-resident
// vertices, loads of consecutive vertices are independent and pipeline), divergence costs only the longest refinement (~10 support
// calls), and the whole self-collision pass costs about as much as two wave-cooperative pairs did.  The portal (4 points x 3
// vectors) lives in registers; portal updates are selects, never dynamic register indexing (which would go to scratch).
#pragma once

#define UHC_CCD_EPS 2.220446049250313e-16
#define UHC_MPR_TOLERANCE 1e-6
#define UHC_MPR_MAXIT 50
// libccd's discoverPortal and refinePortal have no iteration limit of their own (they end geometrically), and on degenerate input they
// have been seen not to end.  A pair that has asked for this many support points without an answer counts as apart.  (A typical pair
// asks for 5-20, findPenetr for at most UHC_MPR_MAXIT + 2 more.)  On a GPU a loop that does not end is a hung queue and a dead process.
#define UHC_MPR_MAXSUP 256

struct V3 { double x, y, z; };
__device__ __forceinline__ V3 v3(double x, double y, double z) { V3 r = {x, y, z}; return r; }
__device__ __forceinline__ V3 operator-(const V3& a, const V3& b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator+(const V3& a, const V3& b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator*(const V3& a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 neg(const V3& a) { return v3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ double vdot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 vcross(const V3& a, const V3& b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ V3 vnorm(const V3& a) { const double n = sqrt(vdot(a, a)); return v3(a.x / n, a.y / n, a.z / n); }
__device__ __forceinline__ V3 vsel(bool c, const V3& a, const V3& b) { return v3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }
__device__ __forceinline__ bool ccd_zero(double x) { return fabs(x) < UHC_CCD_EPS; }
__device__ __forceinline__ bool ccd_eq(double a, double b) {
    const double ab = fabs(a - b);
    if (ab < UHC_CCD_EPS) return true;
    a = fabs(a); b = fabs(b);
    return b > a ? ab < UHC_CCD_EPS * b : ab < UHC_CCD_EPS * a;
}

// A point of the Minkowski difference, v = v1 - v2, and the SUM s = v1 + v2 of its witnesses on hull 1 / hull 2: the contact position is
// the only consumer of the witnesses and it reads them as (v1 + v2) / 2 (findPos; the margin push along +-dir cancels in the sum, the radii of two rounded hulls leave their difference there), so the
// portal carries 6 doubles per point instead of libccd's 9 -- 15 doubles less live state through the refinement loops of every lane.
struct CcdSup { V3 v, s; };
__device__ __forceinline__ CcdSup sup_sel(bool c, const CcdSup& a, const CcdSup& b) {
    CcdSup r = {vsel(c, a.v, b.v), vsel(c, a.s, b.s)};
    return r;
}

// one convex hull posed in the world: rotation (row major), position, vertex range of the model blob
struct CcdHull { double R[9]; V3 p; int voff, vn; };  // voff: offset (doubles) of the hull's first vertex in the vertex array

// hull vertex with the largest projection on `dir` (first maximum wins), in the world, pushed out by hm = margin / 2 (+ the radius of a ROUNDED hull -- a
// sphere is one core vertex, a capsule the two ends of its segment: include/uhc_amd.h) along dir
__device__ __forceinline__ V3 hull_support(const double* __restrict__ VB, const CcdHull& H, const V3& dir, double hm) {
    const double* vert = VB + H.voff;
    const double lx = H.R[0] * dir.x + H.R[3] * dir.y + H.R[6] * dir.z;  // R^T dir
    const double ly = H.R[1] * dir.x + H.R[4] * dir.y + H.R[7] * dir.z;
    const double lz = H.R[2] * dir.x + H.R[5] * dir.y + H.R[8] * dir.z;
    double bd = -1e300, bx = 0, by = 0, bz = 0;
    // four vertices per trip: their twelve loads are issued together (the vertices sit in L2 / L1, a load costs hundreds of cycles and a
    // one-vertex loop would pay that per vertex); the tail repeats the last vertex, which a strict `>` never selects again
    const int last = H.vn - 1;
    for (int v = 0; v < H.vn; v += 4) {
        double c[4][3];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = 3 * min(v + k, last);
            c[k][0] = vert[i]; c[k][1] = vert[i + 1]; c[k][2] = vert[i + 2];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double s = lx * c[k][0] + ly * c[k][1] + lz * c[k][2];
            if (s > bd) { bd = s; bx = c[k][0]; by = c[k][1]; bz = c[k][2]; }
        }
    }
    return v3((H.R[0] * bx + H.R[1] * by + H.R[2] * bz) + (H.p.x + dir.x * hm), (H.R[3] * bx + H.R[4] * by + H.R[5] * bz) + (H.p.y + dir.y * hm),
              (H.R[6] * bx + H.R[7] * by + H.R[8] * bz) + (H.p.z + dir.z * hm));
}
__device__ __forceinline__ CcdSup ccd_support(const double* __restrict__ VB, const CcdHull& H1, const CcdHull& H2, const V3& dir, double hm1, double hm2) {
    CcdSup s;
    const V3 v1 = hull_support(VB, H1, dir, hm1), v2 = hull_support(VB, H2, neg(dir), hm2);
    s.v = v1 - v2;
    s.s = v1 + v2;
    return s;
}
__device__ __forceinline__ V3 portal_dir(const CcdSup& p1, const CcdSup& p2, const CcdSup& p3) {
    return vnorm(vcross(p2.v - p1.v, p3.v - p1.v));
}
__device__ __forceinline__ bool portal_reach_tolerance(const CcdSup& p1, const CcdSup& p2, const CcdSup& p3, const CcdSup& v4, const V3& dir) {
    const double dv1 = vdot(p1.v, dir), dv2 = vdot(p2.v, dir), dv3 = vdot(p3.v, dir), dv4 = vdot(v4.v, dir);
    double d1 = dv4 - dv1;
    const double d2 = dv4 - dv2, d3 = dv4 - dv3;
    d1 = fmin(d1, d2); d1 = fmin(d1, d3);
    return ccd_eq(d1, UHC_MPR_TOLERANCE) || d1 < UHC_MPR_TOLERANCE;
}
__device__ __forceinline__ void expand_portal(const CcdSup& p0, CcdSup& p1, CcdSup& p2, CcdSup& p3, const CcdSup& v4) {
    const V3 v4v0 = vcross(v4.v, p0.v);
    int k;  // which of p1..p3 the new point replaces
    if (vdot(p1.v, v4v0) > 0) k = vdot(p2.v, v4v0) > 0 ? 1 : 3;
    else k = vdot(p3.v, v4v0) > 0 ? 2 : 1;
    p1 = sup_sel(k == 1, v4, p1); p2 = sup_sel(k == 2, v4, p2); p3 = sup_sel(k == 3, v4, p3);
}
__device__ __forceinline__ double point_seg_dist2(const V3& x0, const V3& b, V3& w) {  // from the origin
    const V3 dd = b - x0;
    const double t = -vdot(x0, dd) / vdot(dd, dd);
    if (t < 0 || ccd_zero(t)) w = x0;
    else if (t > 1 || ccd_eq(t, 1)) w = b;
    else w = v3(dd.x * t + x0.x, dd.y * t + x0.y, dd.z * t + x0.z);
    return vdot(w, w);
}
__device__ __forceinline__ double point_tri_dist2(const V3& x0, const V3& B, const V3& C, V3& w) {  // from the origin
    const V3 d1 = B - x0, d2 = C - x0;
    const double v = vdot(d1, d1), ww = vdot(d2, d2), p = vdot(x0, d1), q = vdot(x0, d2), r = vdot(d1, d2);
    const double s = (q * r - ww * p) / (ww * v - r * r);
    const double t = (-s * r - q) / ww;
    if ((ccd_zero(s) || s > 0) && (ccd_eq(s, 1) || s < 1) && (ccd_zero(t) || t > 0) && (ccd_eq(t, 1) || t < 1) && (ccd_eq(t + s, 1) || t + s < 1)) {
        w = v3(x0.x + d1.x * s + d2.x * t, x0.y + d1.y * s + d2.y * t, x0.z + d1.z * s + d2.z * t);
        return vdot(w, w);
    }
    V3 w2;
    double dist = point_seg_dist2(x0, B, w);
    double dist2 = point_seg_dist2(x0, C, w2);
    if (dist2 < dist) { dist = dist2; w = w2; }
    dist2 = point_seg_dist2(B, C, w2);
    if (dist2 < dist) { dist = dist2; w = w2; }
    return dist;
}
__device__ __forceinline__ V3 find_pos(const CcdSup& p0, const CcdSup& p1, const CcdSup& p2, const CcdSup& p3) {
    const V3 dir = portal_dir(p1, p2, p3);
    double b0 = vdot(vcross(p1.v, p2.v), p3.v), b1 = vdot(vcross(p3.v, p2.v), p0.v), b2 = vdot(vcross(p0.v, p1.v), p3.v),
           b3 = vdot(vcross(p2.v, p1.v), p0.v);
    double sum = b0 + b1 + b2 + b3;
    if (ccd_zero(sum) || sum < 0) {
        b0 = 0;
        b1 = vdot(vcross(p2.v, p3.v), dir);
        b2 = vdot(vcross(p3.v, p1.v), dir);
        b3 = vdot(vcross(p1.v, p2.v), dir);
        sum = b1 + b2 + b3;
    }
    const double inv = 1.0 / sum;
    V3 a = v3(0, 0, 0);  // sum_k b_k (v1_k + v2_k): libccd accumulates the two witnesses apart and adds at the end (rounding aside, the same)
    a = a + p0.s * b0;
    a = a + p1.s * b1;
    a = a + p2.s * b2;
    a = a + p3.s * b3;
    return v3(a.x * inv * 0.5, a.y * inv * 0.5, a.z * inv * 0.5);
}

// true: the hulls (each inflated by hm = margin / 2 + its radius) penetrate; depth, dir (from hull 1 to hull 2, zero if undefined) and pos are set.
// c1, c2: the hulls' centres (MuJoCo: geom_xpos = the mesh's centre of mass).
__device__ __forceinline__ bool mpr_penetration(const double* __restrict__ VB, const CcdHull& H1, const CcdHull& H2, const V3& c1, const V3& c2, double hm1, double hm2, double& depth,
                                                V3& dir, V3& pos) {
    CcdSup p0, p1, p2, p3, v4;
    double dt;
    // ---- discoverPortal
    p0.s = c1 + c2; p0.v = c1 - c2;
    if (ccd_eq(p0.v.x, 0) && ccd_eq(p0.v.y, 0) && ccd_eq(p0.v.z, 0)) p0.v.x += UHC_CCD_EPS * 10;
    dir = vnorm(neg(p0.v));
    p1 = ccd_support(VB, H1, H2, dir, hm1, hm2);
    dt = vdot(p1.v, dir);
    if (ccd_zero(dt) || dt < 0) return false;
    dir = vcross(p0.v, p1.v);
    if (ccd_zero(vdot(dir, dir))) {
        if (ccd_eq(p1.v.x, 0) && ccd_eq(p1.v.y, 0) && ccd_eq(p1.v.z, 0)) { depth = 0; dir = v3(0, 0, 0); }   // origin on v1: touching
        else { depth = sqrt(vdot(p1.v, p1.v)); dir = vnorm(p1.v); }                                             // origin on the segment v0-v1
        pos = v3(0.5 * p1.s.x, 0.5 * p1.s.y, 0.5 * p1.s.z);
        return true;
    }
    dir = vnorm(dir);
    p2 = ccd_support(VB, H1, H2, dir, hm1, hm2);
    dt = vdot(p2.v, dir);
    if (ccd_zero(dt) || dt < 0) return false;
    dir = vnorm(vcross(p1.v - p0.v, p2.v - p0.v));
    if (vdot(dir, p0.v) > 0) { const CcdSup t = p1; p1 = p2; p2 = t; dir = neg(dir); }
    for (;;) {
        v4 = ccd_support(VB, H1, H2, dir, hm1, hm2);
        dt = vdot(v4.v, dir);
        if (ccd_zero(dt) || dt < 0) return false;
        bool cont = false;
        dt = vdot(vcross(p1.v, v4.v), p0.v);
        if (dt < 0 && !ccd_zero(dt)) { p2 = v4; cont = true; }
        if (!cont) {
            dt = vdot(vcross(v4.v, p2.v), p0.v);
            if (dt < 0 && !ccd_zero(dt)) { p1 = v4; cont = true; }
        }
        if (!cont) { p3 = v4; break; }
        dir = vnorm(vcross(p1.v - p0.v, p2.v - p0.v));
    }
    // ---- refinePortal
    for (;;) {
        dir = portal_dir(p1, p2, p3);
        dt = vdot(dir, p1.v);
        if (ccd_zero(dt) || dt > 0) break;  // the portal encapsulates the origin
        v4 = ccd_support(VB, H1, H2, dir, hm1, hm2);
        dt = vdot(v4.v, dir);
        if (!(ccd_zero(dt) || dt > 0) || portal_reach_tolerance(p1, p2, p3, v4, dir)) return false;
        expand_portal(p0, p1, p2, p3, v4);
    }
    // ---- findPenetr
    for (int it = 0;; it++) {
        dir = portal_dir(p1, p2, p3);
        v4 = ccd_support(VB, H1, H2, dir, hm1, hm2);
        if (portal_reach_tolerance(p1, p2, p3, v4, dir) || it > UHC_MPR_MAXIT) {
            V3 pd;
            depth = sqrt(point_tri_dist2(p1.v, p2.v, p3.v, pd));
            if (ccd_zero(pd.x) && ccd_zero(pd.y) && ccd_zero(pd.z)) pd = dir;
            dir = vnorm(pd);
            pos = find_pos(p0, p1, p2, p3);
            return true;
        }
        expand_portal(p0, p1, p2, p3, v4);
    }
}

// ------------------------------------------------------------------ the same refinement, supports computed by the whole wave
// Lane = candidate pair for the PORTAL LOGIC (short, branchy, serial: lanes diverge freely), but the support function -- the only wide
// operation and, with one pair per lane walking its own two hulls, ~85 % of the pass: 64 lanes reading 24-byte vertices at unrelated LDS
// addresses meet ~7-way bank conflicts, ~115 cycles per vertex and lane -- is taken out of the lanes: the refinement runs in ROUNDS.
// In a round every live pair asks for one support point (a direction); the wave then serves the requests one pair at a time with
// lane = hull vertex (one coalesced, conflict-free read of <= 64 vertices per hull, dot product, DPP arg-max, lowest lane among equal
// maxima = the scan's "first maximum wins"), hands the point to the pair's lane, and all lanes advance their own state machine to the
// next request.  Same arithmetic as mpr_penetration, same portal, same result bits; the cost of a round is ~300 cycles per live pair
// instead of ~11 000 for the slowest lane's two hull walks.
enum { MPR_DONE = 0, MPR_D1, MPR_D2, MPR_D3, MPR_REFINE, MPR_PEN };
struct MprLane {     // one candidate pair (valid in lanes whose `st` is not MPR_DONE at entry)
    int b1, b2;      // bodies of the two hulls (poses are read from LDS: xmat, xpos)
    int voff1, vn1, voff2, vn2;  // vertex ranges (doubles offset into VB, count)
    V3 c1, c2;       // hull centres in the world
    double hm1, hm2; // how far each hull's surface lies beyond its vertices along a query direction: margin / 2, + the radius of a rounded hull (sphere, capsule)
    // results
    bool hit;
    double depth;
    V3 dir, pos;
};
__device__ __forceinline__ double wave_max_f64(double v) {
    v = fmax(v, dpp_f64<DPP_QUAD_1032>(v));
    v = fmax(v, dpp_f64<DPP_QUAD_2301>(v));
    v = fmax(v, dpp_f64<DPP_ROW_HALF_MIRROR>(v));
    v = fmax(v, dpp_f64<DPP_ROW_MIRROR>(v));
    return fmax(fmax(bcast(v, 0), bcast(v, 16)), fmax(bcast(v, 32), bcast(v, 48)));
}
// support point of one hull for the pair owned by lane p (all arguments wave-uniform): the hull vertex with the largest projection on
// dir (first maximum), in the world, pushed out by hm (margin / 2 + the hull's radius) along dir -- bit for bit hull_support's value
__device__ __forceinline__ V3 hull_support_wave(const double* __restrict__ VB, const double* __restrict__ R, const double* __restrict__ P, int voff, int vn, const V3& dir,
                                                double hm) {
    const double lx = R[0] * dir.x + R[3] * dir.y + R[6] * dir.z;  // R^T dir
    const double ly = R[1] * dir.x + R[4] * dir.y + R[7] * dir.z;
    const double lz = R[2] * dir.x + R[5] * dir.y + R[8] * dir.z;
    double bd = -1e300, bx = 0, by = 0, bz = 0;
    for (int v0 = 0; v0 < vn; v0 += UHC_WAVE) {  // (hulls of the generated models have <= 50 vertices: one trip)
        const int v = v0 + LANE;
        const bool in = v < vn;
        const double* c = VB + voff + 3 * (in ? v : 0);
        const double cx = c[0], cy = c[1], cz = c[2];
        const double s = in ? lx * cx + ly * cy + lz * cz : -1e300;
        const double mx = wave_max_f64(s);
        if (mx > bd) {  // strict: an equal maximum in a later chunk does not replace the first
            const int k = __ffsll((long long)__builtin_amdgcn_ballot_w64(s == mx)) - 1;
            bd = mx; bx = bcast(cx, k); by = bcast(cy, k); bz = bcast(cz, k);
        }
    }
    return v3((R[0] * bx + R[1] * by + R[2] * bz) + (P[0] + dir.x * hm), (R[3] * bx + R[4] * by + R[5] * bz) + (P[1] + dir.y * hm),
              (R[6] * bx + R[7] * by + R[8] * bz) + (P[2] + dir.z * hm));
}
// Both hulls of a pair at once when each fits one trip (<= 64 vertices: every generated model): straight-line code, the two loads, dot
// products and DPP reductions interleave instead of running one after the other through a loop branch.
__device__ __forceinline__ void pair_support_wave(const double* __restrict__ VB, const double* __restrict__ R1, const double* __restrict__ P1, int voff1, int vn1,
                                                  const double* __restrict__ R2, const double* __restrict__ P2, int voff2, int vn2, const V3& d, double hm1, double hm2, V3& a, V3& b) {
    if (vn1 > UHC_WAVE || vn2 > UHC_WAVE) {
        a = hull_support_wave(VB, R1, P1, voff1, vn1, d, hm1);
        b = hull_support_wave(VB, R2, P2, voff2, vn2, neg(d), hm2);
        return;
    }
    const bool in1 = LANE < vn1, in2 = LANE < vn2;
    const double* c1 = VB + voff1 + 3 * (in1 ? LANE : 0);
    const double* c2 = VB + voff2 + 3 * (in2 ? LANE : 0);
    const double x1 = c1[0], y1 = c1[1], z1 = c1[2], x2 = c2[0], y2 = c2[1], z2 = c2[2];
    const V3 e = neg(d);
    const double l1x = R1[0] * d.x + R1[3] * d.y + R1[6] * d.z, l1y = R1[1] * d.x + R1[4] * d.y + R1[7] * d.z, l1z = R1[2] * d.x + R1[5] * d.y + R1[8] * d.z;
    const double l2x = R2[0] * e.x + R2[3] * e.y + R2[6] * e.z, l2y = R2[1] * e.x + R2[4] * e.y + R2[7] * e.z, l2z = R2[2] * e.x + R2[5] * e.y + R2[8] * e.z;
    const double s1 = in1 ? l1x * x1 + l1y * y1 + l1z * z1 : -1e300;
    const double s2 = in2 ? l2x * x2 + l2y * y2 + l2z * z2 : -1e300;
    double m1 = s1, m2 = s2;  // two independent reduction chains, issued alternately
    m1 = fmax(m1, dpp_f64<DPP_QUAD_1032>(m1)); m2 = fmax(m2, dpp_f64<DPP_QUAD_1032>(m2));
    m1 = fmax(m1, dpp_f64<DPP_QUAD_2301>(m1)); m2 = fmax(m2, dpp_f64<DPP_QUAD_2301>(m2));
    m1 = fmax(m1, dpp_f64<DPP_ROW_HALF_MIRROR>(m1)); m2 = fmax(m2, dpp_f64<DPP_ROW_HALF_MIRROR>(m2));
    m1 = fmax(m1, dpp_f64<DPP_ROW_MIRROR>(m1)); m2 = fmax(m2, dpp_f64<DPP_ROW_MIRROR>(m2));
    m1 = fmax(fmax(bcast(m1, 0), bcast(m1, 16)), fmax(bcast(m1, 32), bcast(m1, 48)));
    m2 = fmax(fmax(bcast(m2, 0), bcast(m2, 16)), fmax(bcast(m2, 32), bcast(m2, 48)));
    const int k1 = __ffsll((long long)__builtin_amdgcn_ballot_w64(s1 == m1)) - 1, k2 = __ffsll((long long)__builtin_amdgcn_ballot_w64(s2 == m2)) - 1;
    const double b1x = bcast(x1, k1), b1y = bcast(y1, k1), b1z = bcast(z1, k1), b2x = bcast(x2, k2), b2y = bcast(y2, k2), b2z = bcast(z2, k2);
    a = v3((R1[0] * b1x + R1[1] * b1y + R1[2] * b1z) + (P1[0] + d.x * hm1), (R1[3] * b1x + R1[4] * b1y + R1[5] * b1z) + (P1[1] + d.y * hm1),
           (R1[6] * b1x + R1[7] * b1y + R1[8] * b1z) + (P1[2] + d.z * hm1));
    b = v3((R2[0] * b2x + R2[1] * b2y + R2[2] * b2z) + (P2[0] + e.x * hm2), (R2[3] * b2x + R2[4] * b2y + R2[5] * b2z) + (P2[1] + e.y * hm2),
           (R2[6] * b2x + R2[7] * b2y + R2[8] * b2z) + (P2[2] + e.z * hm2));
}
// Two pairs' requests at once (every hull <= 64 vertices): four independent load / dot / reduction chains in straight-line code.  A support
// query is a dependent chain of ~300 cycles that keeps the VALU busy a third of the time; the second pair's chain fills the gaps.  Per pair
// the arithmetic is pair_support_wave's, operation for operation: same bits.
#ifndef UHC_MPR_PAIRWISE
#define UHC_MPR_PAIRWISE 1
#endif
struct SupArgs { const double *R1, *P1, *R2, *P2; int voff1, vn1, voff2, vn2; V3 d; double hm1, hm2; };
__device__ __forceinline__ void pair_support_wave2(const double* __restrict__ VB, const SupArgs& A, const SupArgs& B, V3& aA, V3& bA, V3& aB, V3& bB) {
    const bool iA1 = LANE < A.vn1, iA2 = LANE < A.vn2, iB1 = LANE < B.vn1, iB2 = LANE < B.vn2;
    const double* cA1 = VB + A.voff1 + 3 * (iA1 ? LANE : 0);
    const double* cA2 = VB + A.voff2 + 3 * (iA2 ? LANE : 0);
    const double* cB1 = VB + B.voff1 + 3 * (iB1 ? LANE : 0);
    const double* cB2 = VB + B.voff2 + 3 * (iB2 ? LANE : 0);
    const double xA1 = cA1[0], yA1 = cA1[1], zA1 = cA1[2], xA2 = cA2[0], yA2 = cA2[1], zA2 = cA2[2];
    const double xB1 = cB1[0], yB1 = cB1[1], zB1 = cB1[2], xB2 = cB2[0], yB2 = cB2[1], zB2 = cB2[2];
    const V3 dA = A.d, eA = neg(A.d), dB = B.d, eB = neg(B.d);
    const double* R;
    R = A.R1; const double lA1x = R[0] * dA.x + R[3] * dA.y + R[6] * dA.z, lA1y = R[1] * dA.x + R[4] * dA.y + R[7] * dA.z, lA1z = R[2] * dA.x + R[5] * dA.y + R[8] * dA.z;
    R = A.R2; const double lA2x = R[0] * eA.x + R[3] * eA.y + R[6] * eA.z, lA2y = R[1] * eA.x + R[4] * eA.y + R[7] * eA.z, lA2z = R[2] * eA.x + R[5] * eA.y + R[8] * eA.z;
    R = B.R1; const double lB1x = R[0] * dB.x + R[3] * dB.y + R[6] * dB.z, lB1y = R[1] * dB.x + R[4] * dB.y + R[7] * dB.z, lB1z = R[2] * dB.x + R[5] * dB.y + R[8] * dB.z;
    R = B.R2; const double lB2x = R[0] * eB.x + R[3] * eB.y + R[6] * eB.z, lB2y = R[1] * eB.x + R[4] * eB.y + R[7] * eB.z, lB2z = R[2] * eB.x + R[5] * eB.y + R[8] * eB.z;
    const double sA1 = iA1 ? lA1x * xA1 + lA1y * yA1 + lA1z * zA1 : -1e300;
    const double sA2 = iA2 ? lA2x * xA2 + lA2y * yA2 + lA2z * zA2 : -1e300;
    const double sB1 = iB1 ? lB1x * xB1 + lB1y * yB1 + lB1z * zB1 : -1e300;
    const double sB2 = iB2 ? lB2x * xB2 + lB2y * yB2 + lB2z * zB2 : -1e300;
    double mA1 = sA1, mA2 = sA2, mB1 = sB1, mB2 = sB2;
    mA1 = fmax(mA1, dpp_f64<DPP_QUAD_1032>(mA1)); mA2 = fmax(mA2, dpp_f64<DPP_QUAD_1032>(mA2)); mB1 = fmax(mB1, dpp_f64<DPP_QUAD_1032>(mB1)); mB2 = fmax(mB2, dpp_f64<DPP_QUAD_1032>(mB2));
    mA1 = fmax(mA1, dpp_f64<DPP_QUAD_2301>(mA1)); mA2 = fmax(mA2, dpp_f64<DPP_QUAD_2301>(mA2)); mB1 = fmax(mB1, dpp_f64<DPP_QUAD_2301>(mB1)); mB2 = fmax(mB2, dpp_f64<DPP_QUAD_2301>(mB2));
    mA1 = fmax(mA1, dpp_f64<DPP_ROW_HALF_MIRROR>(mA1)); mA2 = fmax(mA2, dpp_f64<DPP_ROW_HALF_MIRROR>(mA2)); mB1 = fmax(mB1, dpp_f64<DPP_ROW_HALF_MIRROR>(mB1)); mB2 = fmax(mB2, dpp_f64<DPP_ROW_HALF_MIRROR>(mB2));
    mA1 = fmax(mA1, dpp_f64<DPP_ROW_MIRROR>(mA1)); mA2 = fmax(mA2, dpp_f64<DPP_ROW_MIRROR>(mA2)); mB1 = fmax(mB1, dpp_f64<DPP_ROW_MIRROR>(mB1)); mB2 = fmax(mB2, dpp_f64<DPP_ROW_MIRROR>(mB2));
    mA1 = fmax(fmax(bcast(mA1, 0), bcast(mA1, 16)), fmax(bcast(mA1, 32), bcast(mA1, 48)));
    mA2 = fmax(fmax(bcast(mA2, 0), bcast(mA2, 16)), fmax(bcast(mA2, 32), bcast(mA2, 48)));
    mB1 = fmax(fmax(bcast(mB1, 0), bcast(mB1, 16)), fmax(bcast(mB1, 32), bcast(mB1, 48)));
    mB2 = fmax(fmax(bcast(mB2, 0), bcast(mB2, 16)), fmax(bcast(mB2, 32), bcast(mB2, 48)));
    const int kA1 = __ffsll((long long)__builtin_amdgcn_ballot_w64(sA1 == mA1)) - 1, kA2 = __ffsll((long long)__builtin_amdgcn_ballot_w64(sA2 == mA2)) - 1;
    const int kB1 = __ffsll((long long)__builtin_amdgcn_ballot_w64(sB1 == mB1)) - 1, kB2 = __ffsll((long long)__builtin_amdgcn_ballot_w64(sB2 == mB2)) - 1;
    double bx, by, bz, hm;
    bx = bcast(xA1, kA1); by = bcast(yA1, kA1); bz = bcast(zA1, kA1); hm = A.hm1; R = A.R1;
    aA = v3((R[0] * bx + R[1] * by + R[2] * bz) + (A.P1[0] + dA.x * hm), (R[3] * bx + R[4] * by + R[5] * bz) + (A.P1[1] + dA.y * hm), (R[6] * bx + R[7] * by + R[8] * bz) + (A.P1[2] + dA.z * hm));
    bx = bcast(xA2, kA2); by = bcast(yA2, kA2); bz = bcast(zA2, kA2); hm = A.hm2; R = A.R2;
    bA = v3((R[0] * bx + R[1] * by + R[2] * bz) + (A.P2[0] + eA.x * hm), (R[3] * bx + R[4] * by + R[5] * bz) + (A.P2[1] + eA.y * hm), (R[6] * bx + R[7] * by + R[8] * bz) + (A.P2[2] + eA.z * hm));
    bx = bcast(xB1, kB1); by = bcast(yB1, kB1); bz = bcast(zB1, kB1); hm = B.hm1; R = B.R1;
    aB = v3((R[0] * bx + R[1] * by + R[2] * bz) + (B.P1[0] + dB.x * hm), (R[3] * bx + R[4] * by + R[5] * bz) + (B.P1[1] + dB.y * hm), (R[6] * bx + R[7] * by + R[8] * bz) + (B.P1[2] + dB.z * hm));
    bx = bcast(xB2, kB2); by = bcast(yB2, kB2); bz = bcast(zB2, kB2); hm = B.hm2; R = B.R2;
    bB = v3((R[0] * bx + R[1] * by + R[2] * bz) + (B.P2[0] + eB.x * hm), (R[3] * bx + R[4] * by + R[5] * bz) + (B.P2[1] + eB.y * hm), (R[6] * bx + R[7] * by + R[8] * bz) + (B.P2[2] + eB.z * hm));
}
// xmat / xpos: the bodies' poses in LDS ([nbody][9], [nbody][3])
__device__ __forceinline__ void mpr_wave(const double* __restrict__ VB, const double* __restrict__ xmat, const double* __restrict__ xpos, bool active, MprLane& M) {
    CcdSup p0, p1, p2, p3, v4;
    V3 dir = v3(0, 0, 0);
    double dt;
    int st = MPR_DONE, it = 0;
    M.hit = false; M.depth = 0; M.dir = v3(0, 0, 0); M.pos = v3(0, 0, 0);
    p0.v = p0.s = p1.v = p1.s = p2.v = p2.s = p3.v = p3.s = v4.v = v4.s = v3(0, 0, 0);
    if (active) {
        p0.s = M.c1 + M.c2; p0.v = M.c1 - M.c2;
        if (ccd_eq(p0.v.x, 0) && ccd_eq(p0.v.y, 0) && ccd_eq(p0.v.z, 0)) p0.v.x += UHC_CCD_EPS * 10;
        dir = vnorm(neg(p0.v));
        st = MPR_D1;
    }
    for (int round = 0;; round++) {
        unsigned long long live = __builtin_amdgcn_ballot_w64(st != MPR_DONE);
        if (!live || round == UHC_MPR_MAXSUP) break;
        // ---- serve this round's support requests, one pair at a time, lane = hull vertex
        while (live) {
            const int p = __ffsll((long long)live) - 1;
            live &= live - 1;
            SupArgs Ap;
            {
                const int b1 = __builtin_amdgcn_readlane(M.b1, p), b2 = __builtin_amdgcn_readlane(M.b2, p);
                Ap.R1 = xmat + 9 * b1; Ap.P1 = xpos + 3 * b1; Ap.R2 = xmat + 9 * b2; Ap.P2 = xpos + 3 * b2;
                Ap.voff1 = __builtin_amdgcn_readlane(M.voff1, p); Ap.vn1 = __builtin_amdgcn_readlane(M.vn1, p);
                Ap.voff2 = __builtin_amdgcn_readlane(M.voff2, p); Ap.vn2 = __builtin_amdgcn_readlane(M.vn2, p);
                Ap.d = v3(bcast(dir.x, p), bcast(dir.y, p), bcast(dir.z, p)); Ap.hm1 = bcast(M.hm1, p); Ap.hm2 = bcast(M.hm2, p);
            }
            V3 a, b;
            if (live && UHC_MPR_PAIRWISE) {  // a second request of this round: served together with the first
                const int q = __ffsll((long long)live) - 1;
                SupArgs Aq;
                const int b1 = __builtin_amdgcn_readlane(M.b1, q), b2 = __builtin_amdgcn_readlane(M.b2, q);
                Aq.R1 = xmat + 9 * b1; Aq.P1 = xpos + 3 * b1; Aq.R2 = xmat + 9 * b2; Aq.P2 = xpos + 3 * b2;
                Aq.voff1 = __builtin_amdgcn_readlane(M.voff1, q); Aq.vn1 = __builtin_amdgcn_readlane(M.vn1, q);
                Aq.voff2 = __builtin_amdgcn_readlane(M.voff2, q); Aq.vn2 = __builtin_amdgcn_readlane(M.vn2, q);
                Aq.d = v3(bcast(dir.x, q), bcast(dir.y, q), bcast(dir.z, q)); Aq.hm1 = bcast(M.hm1, q); Aq.hm2 = bcast(M.hm2, q);
                if (max(max(Ap.vn1, Ap.vn2), max(Aq.vn1, Aq.vn2)) <= UHC_WAVE) {
                    live &= live - 1;
                    V3 a2, b2v;
                    pair_support_wave2(VB, Ap, Aq, a, b, a2, b2v);
                    const bool mq = LANE == q;
                    v4.v = vsel(mq, a2 - b2v, v4.v);
                    v4.s = vsel(mq, a2 + b2v, v4.s);
                    const bool mine = LANE == p;
                    v4.v = vsel(mine, a - b, v4.v);
                    v4.s = vsel(mine, a + b, v4.s);
                    continue;
                }
            }
            pair_support_wave(VB, Ap.R1, Ap.P1, Ap.voff1, Ap.vn1, Ap.R2, Ap.P2, Ap.voff2, Ap.vn2, Ap.d, Ap.hm1, Ap.hm2, a, b);
            const bool mine = LANE == p;
            v4.v = vsel(mine, a - b, v4.v);
            v4.s = vsel(mine, a + b, v4.s);
        }
        // ---- every live pair advances to its next request (or finishes)
        bool to_refine = false, to_pen = false;
        if (st == MPR_D1) {
            p1 = v4;
            dt = vdot(p1.v, dir);
            if (ccd_zero(dt) || dt < 0) st = MPR_DONE;
            else {
                dir = vcross(p0.v, p1.v);
                if (ccd_zero(vdot(dir, dir))) {
                    if (ccd_eq(p1.v.x, 0) && ccd_eq(p1.v.y, 0) && ccd_eq(p1.v.z, 0)) { M.depth = 0; M.dir = v3(0, 0, 0); }  // origin on v1: touching
                    else { M.depth = sqrt(vdot(p1.v, p1.v)); M.dir = vnorm(p1.v); }                                          // origin on the segment v0-v1
                    M.pos = v3(0.5 * p1.s.x, 0.5 * p1.s.y, 0.5 * p1.s.z);
                    M.hit = true;
                    st = MPR_DONE;
                } else { dir = vnorm(dir); st = MPR_D2; }
            }
        } else if (st == MPR_D2) {
            p2 = v4;
            dt = vdot(p2.v, dir);
            if (ccd_zero(dt) || dt < 0) st = MPR_DONE;
            else {
                dir = vnorm(vcross(p1.v - p0.v, p2.v - p0.v));
                if (vdot(dir, p0.v) > 0) { const CcdSup t = p1; p1 = p2; p2 = t; dir = neg(dir); }
                st = MPR_D3;
            }
        } else if (st == MPR_D3) {
            dt = vdot(v4.v, dir);
            if (ccd_zero(dt) || dt < 0) st = MPR_DONE;
            else {
                bool cont = false;
                dt = vdot(vcross(p1.v, v4.v), p0.v);
                if (dt < 0 && !ccd_zero(dt)) { p2 = v4; cont = true; }
                if (!cont) {
                    dt = vdot(vcross(v4.v, p2.v), p0.v);
                    if (dt < 0 && !ccd_zero(dt)) { p1 = v4; cont = true; }
                }
                if (cont) dir = vnorm(vcross(p1.v - p0.v, p2.v - p0.v));
                else { p3 = v4; to_refine = true; }
            }
        } else if (st == MPR_REFINE) {
            dt = vdot(v4.v, dir);
            if (!(ccd_zero(dt) || dt > 0) || portal_reach_tolerance(p1, p2, p3, v4, dir)) st = MPR_DONE;
            else { expand_portal(p0, p1, p2, p3, v4); to_refine = true; }
        } else if (st == MPR_PEN) {
            if (portal_reach_tolerance(p1, p2, p3, v4, dir) || it > UHC_MPR_MAXIT) {
                V3 pd;
                M.depth = sqrt(point_tri_dist2(p1.v, p2.v, p3.v, pd));
                if (ccd_zero(pd.x) && ccd_zero(pd.y) && ccd_zero(pd.z)) pd = dir;
                M.dir = vnorm(pd);
                M.pos = find_pos(p0, p1, p2, p3);
                M.hit = true;
                st = MPR_DONE;
            } else {
                expand_portal(p0, p1, p2, p3, v4);
                it++;
                dir = portal_dir(p1, p2, p3);
            }
        }
        if (to_refine) {  // head of refinePortal: does the portal already enclose the origin?
            dir = portal_dir(p1, p2, p3);
            dt = vdot(dir, p1.v);
            if (ccd_zero(dt) || dt > 0) to_pen = true;
            else st = MPR_REFINE;
        }
        if (to_pen) { dir = portal_dir(p1, p2, p3); it = 0; st = MPR_PEN; }  // head of findPenetr
    }
}

#if defined(UHC_NW2)
// ------------------------------------------------------------------ the same rounds, the requests of a round served by TWO waves (round 6)
// The general tier's queue consumers are two-wave workgroups (uhc_k_general_q.hip, -DUHC_NW2): two 79 KiB consumers share a CU whose other two SIMDs idled, and a
// quarter of the cycles of the step's slowest env are support requests.  Wave 0 owns the pairs' state machines as in mpr_wave; per round it writes every live pair's
// direction into an LDS mailbox, posts the round (barrier), BOTH waves walk the live mask in the order mpr_wave does and serve alternate groups of two requests
// (pair_support_wave2 / pair_support_wave: the per-pair arithmetic of mpr_wave, operation for operation -- same bits), lane 0 of the serving wave writes the point
// (v1 - v2, v1 + v2) back, barrier, the pairs' lanes pick their points up and advance.  The helper sleeps in s_barrier between rounds and between MPR passes.
enum { MCMD_EXIT = 0, MCMD_ROUND = 1, MCMD_ROWS = 2 };  // ROWS: the helper builds every second group of dense rows and the constraint rows 64 .. nefc - 1 (uhc_physics_impl.h: k_rows), header ints: 1 = nefc, 2 = number of dense rows (0: they are wave 0's alone), 4 / 5 = the env's model blob
// The mailbox sits on the rows' scalar arrays (free until the rows are enumerated), every piece INSIDE one of them (the debug layout puts guard words between the arrays):
//   rowR: ints cmd, live lo, live hi, vertex base (LDS offset in doubles, or -1: the global pointer in ints 4, 5) | x of the direction in / of v1 - v2 out, per pair
//   rowAref: y, z | rowB: x, y of v1 + v2 | rowF: its z, the pair's hm1 | rowDa: ints b1, b2, voff1, vn1 per pair | rowW: ints voff2, vn2 per pair, then the pair's hm2
struct MprMB { int* hdr; double* io[6]; double* hm1; double* hm2; int* st4; int* st2; };
template <int TIER>
__device__ __forceinline__ MprMB mpr_mb(const KernelArgs& A, double* S) {
    const DevLds& L = lds_of<TIER>(A);
    MprMB m;
    m.hdr = (int*)(S + L.rowR);
    m.io[0] = S + L.rowR + 4;
    m.io[1] = S + L.rowAref; m.io[2] = S + L.rowAref + UHC_WAVE;
    m.io[3] = S + L.rowB; m.io[4] = S + L.rowB + UHC_WAVE;
    m.io[5] = S + L.rowF; m.hm1 = S + L.rowF + UHC_WAVE;
    m.st4 = (int*)(S + L.rowDa); m.st2 = (int*)(S + L.rowW); m.hm2 = S + L.rowW + UHC_WAVE;  // (st2: 128 ints = the first 64 doubles of rowW)
    return m;
}
__device__ __forceinline__ SupArgs mpr_mb_args(const MprMB& MB, const double* xmat, const double* xpos, int p) {
    SupArgs a;
    const int b1 = __builtin_amdgcn_readfirstlane(MB.st4[4 * p]), b2 = __builtin_amdgcn_readfirstlane(MB.st4[4 * p + 1]);
    a.R1 = xmat + 9 * b1; a.P1 = xpos + 3 * b1; a.R2 = xmat + 9 * b2; a.P2 = xpos + 3 * b2;
    a.voff1 = __builtin_amdgcn_readfirstlane(MB.st4[4 * p + 2]); a.vn1 = __builtin_amdgcn_readfirstlane(MB.st4[4 * p + 3]);
    a.voff2 = __builtin_amdgcn_readfirstlane(MB.st2[2 * p]); a.vn2 = __builtin_amdgcn_readfirstlane(MB.st2[2 * p + 1]);
    a.hm1 = MB.hm1[p]; a.hm2 = MB.hm2[p];
    a.d = v3(MB.io[0][p], MB.io[1][p], MB.io[2][p]);
    return a;
}
__device__ __forceinline__ void mpr_mb_put(const MprMB& MB, int p, const V3& a, const V3& b) {
    if (LANE == 0) {
        MB.io[0][p] = a.x - b.x; MB.io[1][p] = a.y - b.y; MB.io[2][p] = a.z - b.z; MB.io[3][p] = a.x + b.x; MB.io[4][p] = a.y + b.y; MB.io[5][p] = a.z + b.z;
    }
}
// one round's requests, dealt out in CONTIGUOUS shares: wave wid serves the live pairs of rank [wid h, (wid + 1) h), h = ceil(n / nw) -- two at a time where all four hulls fit one
// trip (pair_support_wave2), a last odd one singly (pair_support_wave).  (Round 6, first form: alternate groups of two -- with two live pairs, the commonest late-round count, the
// helper had nothing to do and wave 0 a pairwise query of ~400 cycles; now each wave has one single query of ~300.)  Per pair the arithmetic of either form is mpr_wave's: same bits.
__device__ __forceinline__ void mpr_serve_round(const double* __restrict__ VB, const double* __restrict__ xmat, const double* __restrict__ xpos, const MprMB& MB, unsigned long long live, int wid, int nw) {
    const int n = __builtin_popcountll(live), h = (n + nw - 1) / nw;
    int take = min(h, n - wid * h);
    for (int k = 0; k < wid * h && live; k++) live &= live - 1;  // the shares of the waves before this one
    while (live && take > 0) {
        const int p = __ffsll((long long)live) - 1;
        live &= live - 1;
        const SupArgs Ap = mpr_mb_args(MB, xmat, xpos, p);
        V3 a, b;
        if (take >= 2 && live && UHC_MPR_PAIRWISE) {
            const int q = __ffsll((long long)live) - 1;
            const SupArgs Aq = mpr_mb_args(MB, xmat, xpos, q);
            if (max(max(Ap.vn1, Ap.vn2), max(Aq.vn1, Aq.vn2)) <= UHC_WAVE) {
                live &= live - 1;
                take -= 2;
                V3 a2, b2v;
                pair_support_wave2(VB, Ap, Aq, a, b, a2, b2v);
                mpr_mb_put(MB, p, a, b);
                mpr_mb_put(MB, q, a2, b2v);
                continue;
            }
        }
        pair_support_wave(VB, Ap.R1, Ap.P1, Ap.voff1, Ap.vn1, Ap.R2, Ap.P2, Ap.voff2, Ap.vn2, Ap.d, Ap.hm1, Ap.hm2, a, b);
        mpr_mb_put(MB, p, a, b);
        take -= 1;
    }
}
// wave 0's side: mpr_wave with the serving loop replaced by mailbox + barrier + shared serving + barrier
template <bool LDSV>
__device__ __forceinline__ void mpr_wave_mw(const double* __restrict__ VB, const double* __restrict__ xmat, const double* __restrict__ xpos, bool active, MprLane& M, const MprMB& MB, int vstage,
                                            const double* vb_global) {
    CcdSup p0, p1, p2, p3, v4;
    V3 dir = v3(0, 0, 0);
    double dt;
    int st = MPR_DONE, it = 0;
    M.hit = false; M.depth = 0; M.dir = v3(0, 0, 0); M.pos = v3(0, 0, 0);
    p0.v = p0.s = p1.v = p1.s = p2.v = p2.s = p3.v = p3.s = v4.v = v4.s = v3(0, 0, 0);
    if (active) {
        p0.s = M.c1 + M.c2; p0.v = M.c1 - M.c2;
        if (ccd_eq(p0.v.x, 0) && ccd_eq(p0.v.y, 0) && ccd_eq(p0.v.z, 0)) p0.v.x += UHC_CCD_EPS * 10;
        dir = vnorm(neg(p0.v));
        st = MPR_D1;
        MB.st4[4 * LANE] = M.b1; MB.st4[4 * LANE + 1] = M.b2; MB.st4[4 * LANE + 2] = M.voff1; MB.st4[4 * LANE + 3] = M.vn1;
        MB.st2[2 * LANE] = M.voff2; MB.st2[2 * LANE + 1] = M.vn2;
        MB.hm1[LANE] = M.hm1; MB.hm2[LANE] = M.hm2;
    }
    int* mbi = MB.hdr;
    if (LANE == 0) {
        mbi[3] = LDSV ? vstage : -1;
        mbi[4] = (int)((unsigned long long)vb_global & 0xffffffffull); mbi[5] = (int)((unsigned long long)vb_global >> 32);
    }
    for (int round = 0;; round++) {
        const unsigned long long live = __builtin_amdgcn_ballot_w64(st != MPR_DONE);
        if (!live || round == UHC_MPR_MAXSUP) break;
        if (st != MPR_DONE) { MB.io[0][LANE] = dir.x; MB.io[1][LANE] = dir.y; MB.io[2][LANE] = dir.z; }
        if (LANE == 0) { mbi[0] = MCMD_ROUND; mbi[1] = (int)(live & 0xffffffffull); mbi[2] = (int)(live >> 32); }
        __syncthreads();
        mpr_serve_round(VB, xmat, xpos, MB, live, 0, UHC_NWG);
        __syncthreads();
        if (st != MPR_DONE) { v4.v = v3(MB.io[0][LANE], MB.io[1][LANE], MB.io[2][LANE]); v4.s = v3(MB.io[3][LANE], MB.io[4][LANE], MB.io[5][LANE]); }
        // ---- every live pair advances to its next request (or finishes): mpr_wave's state machine, unchanged
        bool to_refine = false, to_pen = false;
        if (st == MPR_D1) {
            p1 = v4;
            dt = vdot(p1.v, dir);
            if (ccd_zero(dt) || dt < 0) st = MPR_DONE;
            else {
                dir = vcross(p0.v, p1.v);
                if (ccd_zero(vdot(dir, dir))) {
                    if (ccd_eq(p1.v.x, 0) && ccd_eq(p1.v.y, 0) && ccd_eq(p1.v.z, 0)) { M.depth = 0; M.dir = v3(0, 0, 0); }
                    else { M.depth = sqrt(vdot(p1.v, p1.v)); M.dir = vnorm(p1.v); }
                    M.pos = v3(0.5 * p1.s.x, 0.5 * p1.s.y, 0.5 * p1.s.z);
                    M.hit = true;
                    st = MPR_DONE;
                } else { dir = vnorm(dir); st = MPR_D2; }
            }
        } else if (st == MPR_D2) {
            p2 = v4;
            dt = vdot(p2.v, dir);
            if (ccd_zero(dt) || dt < 0) st = MPR_DONE;
            else {
                dir = vnorm(vcross(p1.v - p0.v, p2.v - p0.v));
                if (vdot(dir, p0.v) > 0) { const CcdSup t = p1; p1 = p2; p2 = t; dir = neg(dir); }
                st = MPR_D3;
            }
        } else if (st == MPR_D3) {
            dt = vdot(v4.v, dir);
            if (ccd_zero(dt) || dt < 0) st = MPR_DONE;
            else {
                bool cont = false;
                dt = vdot(vcross(p1.v, v4.v), p0.v);
                if (dt < 0 && !ccd_zero(dt)) { p2 = v4; cont = true; }
                if (!cont) {
                    dt = vdot(vcross(v4.v, p2.v), p0.v);
                    if (dt < 0 && !ccd_zero(dt)) { p1 = v4; cont = true; }
                }
                if (cont) dir = vnorm(vcross(p1.v - p0.v, p2.v - p0.v));
                else { p3 = v4; to_refine = true; }
            }
        } else if (st == MPR_REFINE) {
            dt = vdot(v4.v, dir);
            if (!(ccd_zero(dt) || dt > 0) || portal_reach_tolerance(p1, p2, p3, v4, dir)) st = MPR_DONE;
            else { expand_portal(p0, p1, p2, p3, v4); to_refine = true; }
        } else if (st == MPR_PEN) {
            if (portal_reach_tolerance(p1, p2, p3, v4, dir) || it > UHC_MPR_MAXIT) {
                V3 pd;
                M.depth = sqrt(point_tri_dist2(p1.v, p2.v, p3.v, pd));
                if (ccd_zero(pd.x) && ccd_zero(pd.y) && ccd_zero(pd.z)) pd = dir;
                M.dir = vnorm(pd);
                M.pos = find_pos(p0, p1, p2, p3);
                M.hit = true;
                st = MPR_DONE;
            } else {
                expand_portal(p0, p1, p2, p3, v4);
                it++;
                dir = portal_dir(p1, p2, p3);
            }
        }
        if (to_refine) {
            dir = portal_dir(p1, p2, p3);
            dt = vdot(dir, p1.v);
            if (ccd_zero(dt) || dt > 0) to_pen = true;
            else st = MPR_REFINE;
        }
        if (to_pen) { dir = portal_dir(p1, p2, p3); it = 0; st = MPR_PEN; }
    }
}
// the helper wave of a two-wave consumer: asleep in the barrier until wave 0 posts a round
template <int TIER>
__device__ __forceinline__ void mpr_helper(const KernelArgs& A, double* S) {
    const DevLds& L = lds_of<TIER>(A);
    const MprMB MB = mpr_mb<TIER>(A, S);
    const int* mbi = MB.hdr;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const LaneConst HLC = lane_const<true>(A.t);  // (the per-lane dof constants of the dense rows' back substitution: model topology, the same for every env)
    for (;;) {
        __syncthreads();
        const int cmd = __builtin_amdgcn_readfirstlane(mbi[0]);
        if (cmd == MCMD_EXIT) return;
        if (cmd == MCMD_ROWS) {
            const int nefc = __builtin_amdgcn_readfirstlane(mbi[1]), ntwo = __builtin_amdgcn_readfirstlane(mbi[2]);  // ntwo: 0 = the dense rows are wave 0's alone
            const unsigned long long mbp = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(mbi[5]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(mbi[4]);
            __syncthreads();  // the command is in registers: the rows may overwrite the header (it sits on rowR)
            if (ntwo > 0) k_dense_groups<TIER>(A, S, HLC, S + L.dense, ntwo, 1, 2);  // every second group of dense rows
            __syncthreads();  // the dense rows' scalars are there
            if (UHC_WAVE + LANE < nefc) k_row_one<TIER>(A, (const double*)mbp, S, UHC_WAVE + LANE, S + L.Y);
            __syncthreads();
            continue;
        }
        const unsigned long long live = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(mbi[2]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(mbi[1]);
        const int vst = __builtin_amdgcn_readfirstlane(mbi[3]);
        if (vst >= 0) mpr_serve_round(S + vst, S + L.xmat, S + L.xpos, MB, live, wid, UHC_NWG);
        else {
            const unsigned long long vb = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(mbi[5]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(mbi[4]);
            mpr_serve_round((const double*)vb, S + L.xmat, S + L.xpos, MB, live, wid, UHC_NWG);
        }
        __syncthreads();
    }
}
template <int TIER>
__device__ __forceinline__ void mpr_release_helpers(const KernelArgs& A, double* S) {
    int* mbi = mpr_mb<TIER>(A, S).hdr;
    if (LANE == 0) mbi[0] = MCMD_EXIT;
    __syncthreads();
}
#endif
