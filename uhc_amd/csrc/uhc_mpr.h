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

struct CcdSup { V3 v, v1, v2; };  // point of the Minkowski difference + its witnesses on hull 1 / hull 2
__device__ __forceinline__ CcdSup sup_sel(bool c, const CcdSup& a, const CcdSup& b) {
    CcdSup r = {vsel(c, a.v, b.v), vsel(c, a.v1, b.v1), vsel(c, a.v2, b.v2)};
    return r;
}

// one convex hull posed in the world: rotation (row major), position, vertex range of the model blob
struct CcdHull { double R[9]; V3 p; int voff, vn; };  // voff: offset (doubles) of the hull's first vertex in the vertex array

// hull vertex with the largest projection on `dir` (first maximum wins), in the world, pushed out by margin / 2 along dir
__device__ __forceinline__ V3 hull_support(const double* __restrict__ VB, const CcdHull& H, const V3& dir, double margin) {
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
    const double hm = 0.5 * margin;
    return v3((H.R[0] * bx + H.R[1] * by + H.R[2] * bz) + (H.p.x + dir.x * hm), (H.R[3] * bx + H.R[4] * by + H.R[5] * bz) + (H.p.y + dir.y * hm),
              (H.R[6] * bx + H.R[7] * by + H.R[8] * bz) + (H.p.z + dir.z * hm));
}
__device__ __forceinline__ CcdSup ccd_support(const double* __restrict__ VB, const CcdHull& H1, const CcdHull& H2, const V3& dir, double margin) {
    CcdSup s;
    s.v1 = hull_support(VB, H1, dir, margin);
    s.v2 = hull_support(VB, H2, neg(dir), margin);
    s.v = s.v1 - s.v2;
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
    V3 a = v3(0, 0, 0), c = v3(0, 0, 0);
    a = a + p0.v1 * b0; c = c + p0.v2 * b0;
    a = a + p1.v1 * b1; c = c + p1.v2 * b1;
    a = a + p2.v1 * b2; c = c + p2.v2 * b2;
    a = a + p3.v1 * b3; c = c + p3.v2 * b3;
    return v3((a.x * inv + c.x * inv) * 0.5, (a.y * inv + c.y * inv) * 0.5, (a.z * inv + c.z * inv) * 0.5);
}

// true: the hulls (each inflated by margin / 2) penetrate; depth, dir (from hull 1 to hull 2, zero if undefined) and pos are set.
// c1, c2: the hulls' centres (MuJoCo: geom_xpos = the mesh's centre of mass).
__device__ __forceinline__ bool mpr_penetration(const double* __restrict__ VB, const CcdHull& H1, const CcdHull& H2, const V3& c1, const V3& c2, double margin, double& depth,
                                                V3& dir, V3& pos) {
    CcdSup p0, p1, p2, p3, v4;
    double dt;
    // ---- discoverPortal
    p0.v1 = c1; p0.v2 = c2; p0.v = c1 - c2;
    if (ccd_eq(p0.v.x, 0) && ccd_eq(p0.v.y, 0) && ccd_eq(p0.v.z, 0)) p0.v.x += UHC_CCD_EPS * 10;
    dir = vnorm(neg(p0.v));
    p1 = ccd_support(VB, H1, H2, dir, margin);
    dt = vdot(p1.v, dir);
    if (ccd_zero(dt) || dt < 0) return false;
    dir = vcross(p0.v, p1.v);
    if (ccd_zero(vdot(dir, dir))) {
        if (ccd_eq(p1.v.x, 0) && ccd_eq(p1.v.y, 0) && ccd_eq(p1.v.z, 0)) { depth = 0; dir = v3(0, 0, 0); }   // origin on v1: touching
        else { depth = sqrt(vdot(p1.v, p1.v)); dir = vnorm(p1.v); }                                             // origin on the segment v0-v1
        pos = v3(0.5 * (p1.v1.x + p1.v2.x), 0.5 * (p1.v1.y + p1.v2.y), 0.5 * (p1.v1.z + p1.v2.z));
        return true;
    }
    dir = vnorm(dir);
    p2 = ccd_support(VB, H1, H2, dir, margin);
    dt = vdot(p2.v, dir);
    if (ccd_zero(dt) || dt < 0) return false;
    dir = vnorm(vcross(p1.v - p0.v, p2.v - p0.v));
    if (vdot(dir, p0.v) > 0) { const CcdSup t = p1; p1 = p2; p2 = t; dir = neg(dir); }
    for (;;) {
        v4 = ccd_support(VB, H1, H2, dir, margin);
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
        v4 = ccd_support(VB, H1, H2, dir, margin);
        dt = vdot(v4.v, dir);
        if (!(ccd_zero(dt) || dt > 0) || portal_reach_tolerance(p1, p2, p3, v4, dir)) return false;
        expand_portal(p0, p1, p2, p3, v4);
    }
    // ---- findPenetr
    for (int it = 0;; it++) {
        dir = portal_dir(p1, p2, p3);
        v4 = ccd_support(VB, H1, H2, dir, margin);
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
