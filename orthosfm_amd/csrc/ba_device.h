// Device math of the bundle-adjustment path (hot path B): the two
// orthographic reprojection residuals with hand-derived analytic Jacobians,
// the two manifold operations and small dense helpers.  All double precision.
//
// Reference formulas:
//   quaternion model: src/algorithms/orthographic_quaternion/
//       OrthographicQuaternionReprojectorError.h:24-67
//   Euler model:      src/algorithms/orthographic/OrthographicReprojectionError.h:26-77
//   manifolds: ceres::EigenQuaternionParameterization and
//       ceres::HomogeneousVectorParameterization(4) as wired in
//       OrthoQuaternionRecoAlgorithm.cpp:134-139 and bundle_adjustment.cpp:88-90
//   robustifier: ceres::HuberLoss(1.0), bundle_adjustment.cpp:64
// (The reference differentiates with ceres::AutoDiffCostFunction; the
// expressions below are the closed-form derivatives of the same functors.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace osfm {

enum { kModelQuat = 0, kModelEuler = 1 };

// Levenberg-Marquardt state, device resident: the accept / reject decisions of
// ceres::TrustRegionMinimizer are taken by a one-workgroup kernel at the end of every
// iteration, the kernels of the next iteration read what it decided (which of the two
// iterate buffers is current, the radius, whether to refresh the LM diagonal), and the
// host only looks at `stop` -- one iteration late, so the queue never runs dry.
struct LmDev {
    double radius, decrease_factor, x_cost, grad_max;
    double cand_cost, model_cost_change, step_norm, x_norm;
    double initial_cost;
    int32_t cur;                 // index of the current iterate in cams2 / points2
    int32_t iteration, term, stop;
    int32_t num_success, num_unsuccess, invalid_steps, last_successful;
    int32_t update_diag, want_gradient, lin_failed, nonfinite;
    int32_t flow_aborted;        // the one-launch Cholesky gave its launch up: nothing is decided until the host has repeated it
    int32_t pad0;
};

struct BaDev {
    int model, C, M, O, nc, pdim;
    const double *cams;          // [C][7]
    const double *points;        // [M][4]
    // LM solve: both iterate buffers and the state that says which one is current
    // (lm == nullptr: cams / points above are used as they are)
    double *cams2[2];
    double *points2[2];
    // per-camera derived tables (cam_derive below), one per iterate buffer; camder = that of `cams`
    const double *camder;
    double *camder2[2];
    const LmDev *lm;
    const double *obs_xy;        // [O][2]
    const int32_t *obs_cam;      // [O]
    const int32_t *obs_pt;       // [O]
    const int32_t *pt_start;     // [M+1]
    const int32_t *img_w, *img_h;
    const int32_t *cam_ldim;     // [C] tangent columns of the camera
    const int32_t *cam_off;      // [C] offset into the camera tangent vector
    const int8_t *cam_colmap;    // [C][6] tangent column -> full column (0..5)
    const double *scale_c;       // [nc]  Jacobi column scaling
    const double *scale_p;       // [3M]
    double huber;
};

// store that another workgroup of the SAME launch may read (global_store ... sc1: write-through)
__device__ __forceinline__ void store_sc1(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_sc1(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p),
        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// The iterate the LM state points at (kernels of the solve call this first; a stopped solve
// makes them return at once: the host enqueues one iteration ahead of what it knows).
__device__ __forceinline__ bool lm_resolve(BaDev &d)
{
    if (!d.lm) return true;
    if (d.lm->stop || d.lm->flow_aborted) return false;
    // selects, not d.cams2[cur]: a kernel argument indexed with a run-time value is copied to scratch
    // memory as a whole (184 bytes per lane), and every field read after that comes from there
    const int cur = d.lm->cur;
    d.cams = cur ? d.cams2[1] : d.cams2[0];
    d.points = cur ? d.points2[1] : d.points2[0];
    d.camder = cur ? d.camder2[1] : d.camder2[0];
    return true;
}

// Residual (uncorrected) and FULL tangent Jacobians of one observation:
//   Jc[2][6]: QUAT  columns = (rot d0, d1, d2, offX, offY, scale)
//             EULER columns = (phi, theta, rho, offX, offY, scale)
//   JP[2][4]: w.r.t. the homogeneous point (ambient)
struct ObsFull {
    double r[2];
    double Jc[2][6];
    double JP[2][4];
};

__device__ __forceinline__ void cross3(const double a[3], const double b[3], double o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// pixel residual from the local point l (shared tail of both functors)
//   x_px = W * (((l.x / s) - offX) / (-2) + 0.5)
__device__ __forceinline__ void pixel_residual(const double l[3], double offx, double offy, double s,
    double W, double H, double ox, double oy, double r[2])
{
    r[0] = W * ((((l[0] / s) - offx) / (-2.0)) + 0.5) - ox;
    r[1] = H * ((((l[1] / s) - offy) / (-2.0)) + 0.5) - oy;
}

__device__ __forceinline__ void
eval_quat(const double *cam, const double *P, double W, double H, double ox, double oy, bool want_j,
    ObsFull &e)
{
    const double qx = cam[0], qy = cam[1], qz = cam[2], qw = cam[3];
    const double offx = cam[4], offy = cam[5], s = cam[6];
    const double iw = 1.0 / P[3];
    const double p[3] = { P[0] / P[3], P[1] / P[3], P[2] / P[3] };
    const double n2 = qx * qx + qy * qy + qz * qz + qw * qw;
    // q.inverse() = conj / |q|^2 ; Eigen rotates with v + w*(2 u x v) + u x (2 u x v)
    const double a[3] = { -qx / n2, -qy / n2, -qz / n2 };
    const double b = qw / n2;
    double t[3];
    cross3(a, p, t);
    double t2[3] = { 2.0 * t[0], 2.0 * t[1], 2.0 * t[2] };
    double at2[3];
    cross3(a, t2, at2);
    const double l[3] = { p[0] + b * t2[0] + at2[0], p[1] + b * t2[1] + at2[1], p[2] + b * t2[2] + at2[2] };
    pixel_residual(l, offx, offy, s, W, H, ox, oy, e.r);
    if (!want_j) return;

    // d r / d l
    const double gx = -W / (2.0 * s), gy = -H / (2.0 * s);
    // R = d l / d p = I + 2 b [a]x + 2 [a]x [a]x   (rows 0,1 needed)
    //   [a]x [a]x = a a^T - |a|^2 I
    const double aa = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    double R0[3], R1[3];
    R0[0] = 1.0 + 2.0 * (a[0] * a[0] - aa);
    R0[1] = 2.0 * (a[0] * a[1]) - 2.0 * b * a[2];
    R0[2] = 2.0 * (a[0] * a[2]) + 2.0 * b * a[1];
    R1[0] = 2.0 * (a[1] * a[0]) + 2.0 * b * a[2];
    R1[1] = 1.0 + 2.0 * (a[1] * a[1] - aa);
    R1[2] = 2.0 * (a[1] * a[2]) - 2.0 * b * a[0];
    // point: d p / d P = [I / w | -p / w]
    for (int k = 0; k < 3; ++k) { e.JP[0][k] = gx * R0[k] * iw; e.JP[1][k] = gy * R1[k] * iw; }
    e.JP[0][3] = -gx * (R0[0] * p[0] + R0[1] * p[1] + R0[2] * p[2]) * iw;
    e.JP[1][3] = -gy * (R1[0] * p[0] + R1[1] * p[1] + R1[2] * p[2]) * iw;

    // d l / d a_k = 2 b (e_k x p) + 2 (e_k x t + a x (e_k x p)) ; d l / d b = 2 t
    double dl_da[3][2];   // [k][component 0/1 of l]
    for (int k = 0; k < 3; ++k) {
        double ek[3] = { 0.0, 0.0, 0.0 };
        ek[k] = 1.0;
        double ekp[3], ekt[3], aekp[3];
        cross3(ek, p, ekp);
        cross3(ek, t, ekt);
        cross3(a, ekp, aekp);
        dl_da[k][0] = 2.0 * b * ekp[0] + 2.0 * (ekt[0] + aekp[0]);
        dl_da[k][1] = 2.0 * b * ekp[1] + 2.0 * (ekt[1] + aekp[1]);
    }
    const double dl_db[2] = { 2.0 * t[0], 2.0 * t[1] };
    // a_i = -u_i / n2 ; b = w / n2
    const double u[3] = { qx, qy, qz };
    const double in2 = 1.0 / n2, in4 = in2 * in2;
    double Jq[2][4];       // d l_{0,1} / d (qx, qy, qz, qw)
    for (int m = 0; m < 3; ++m) {
        double s0 = 0.0, s1 = 0.0;
        for (int i = 0; i < 3; ++i) {
            const double dai = (i == m ? -in2 : 0.0) + 2.0 * u[i] * u[m] * in4;
            s0 += dl_da[i][0] * dai;
            s1 += dl_da[i][1] * dai;
        }
        const double dbm = -2.0 * qw * u[m] * in4;
        Jq[0][m] = s0 + dl_db[0] * dbm;
        Jq[1][m] = s1 + dl_db[1] * dbm;
    }
    {
        double s0 = 0.0, s1 = 0.0;
        for (int i = 0; i < 3; ++i) {
            const double dai = 2.0 * u[i] * qw * in4;
            s0 += dl_da[i][0] * dai;
            s1 += dl_da[i][1] * dai;
        }
        const double dbw = in2 - 2.0 * qw * qw * in4;
        Jq[0][3] = s0 + dl_db[0] * dbw;
        Jq[1][3] = s1 + dl_db[1] * dbw;
    }
    // tangent of EigenQuaternionParameterization: 4x3 plus-Jacobian rows (x,y,z,w)
    //   [ w  z -y ; -z  w  x ;  y -x  w ; -x -y -z ]
    const double PJ[4][3] = { { qw, qz, -qy }, { -qz, qw, qx }, { qy, -qx, qw }, { -qx, -qy, -qz } };
    for (int c = 0; c < 3; ++c) {
        double s0 = 0.0, s1 = 0.0;
        for (int i = 0; i < 4; ++i) { s0 += Jq[0][i] * PJ[i][c]; s1 += Jq[1][i] * PJ[i][c]; }
        e.Jc[0][c] = gx * s0;
        e.Jc[1][c] = gy * s1;
    }
    e.Jc[0][3] = W / 2.0; e.Jc[1][3] = 0.0;       // offX
    e.Jc[0][4] = 0.0;     e.Jc[1][4] = H / 2.0;   // offY
    e.Jc[0][5] = W * l[0] / (2.0 * s * s);        // scale
    e.Jc[1][5] = H * l[1] / (2.0 * s * s);
}

__device__ __forceinline__ void mat3_mul(const double A[3][3], const double B[3][3], double Cm[3][3])
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            Cm[i][j] = A[i][0] * B[0][j] + A[i][1] * B[1][j] + A[i][2] * B[2][j];
}

__device__ __forceinline__ void
eval_euler(const double *cam, const double *P, double W, double H, double ox, double oy, bool want_j,
    ObsFull &e)
{
    const double phi = cam[0], theta = cam[1], rho = cam[2];
    const double offx = cam[3], offy = cam[4], s = cam[5];
    const double om = theta + 1.57079632679489661923;     // M_PI_2
    const double iw = 1.0 / P[3];
    const double p[3] = { P[0] / P[3], P[1] / P[3], P[2] / P[3] };
    double so, co, sr, cr, sp, cp;
    sincos(om, &so, &co);
    sincos(rho, &sr, &cr);
    sincos(phi, &sp, &cp);
    const double Rx[3][3] = { { 1, 0, 0 }, { 0, co, -so }, { 0, so, co } };
    const double Ry[3][3] = { { cr, -sr, 0 }, { sr, cr, 0 }, { 0, 0, 1 } };
    const double Rz[3][3] = { { cp, -sp, 0 }, { sp, cp, 0 }, { 0, 0, 1 } };
    double A[3][3], S[3][3];
    mat3_mul(Rz, Rx, A);
    mat3_mul(A, Ry, S);
    const double tp[3] = { p[0], -p[2], p[1] };            // T * p
    double l[3];
    for (int i = 0; i < 3; ++i) l[i] = S[0][i] * tp[0] + S[1][i] * tp[1] + S[2][i] * tp[2];
    pixel_residual(l, offx, offy, s, W, H, ox, oy, e.r);
    if (!want_j) return;
    const double gx = -W / (2.0 * s), gy = -H / (2.0 * s);
    // angle derivatives: d l / d angle = (dS/d angle)^T T p
    const double dRx[3][3] = { { 0, 0, 0 }, { 0, -so, -co }, { 0, co, -so } };
    const double dRy[3][3] = { { -sr, -cr, 0 }, { cr, -sr, 0 }, { 0, 0, 0 } };
    const double dRz[3][3] = { { -sp, -cp, 0 }, { cp, -sp, 0 }, { 0, 0, 0 } };
    double B1[3][3], dS[3][3];
    // phi
    mat3_mul(dRz, Rx, B1);
    mat3_mul(B1, Ry, dS);
    e.Jc[0][0] = gx * (dS[0][0] * tp[0] + dS[1][0] * tp[1] + dS[2][0] * tp[2]);
    e.Jc[1][0] = gy * (dS[0][1] * tp[0] + dS[1][1] * tp[1] + dS[2][1] * tp[2]);
    // theta
    mat3_mul(Rz, dRx, B1);
    mat3_mul(B1, Ry, dS);
    e.Jc[0][1] = gx * (dS[0][0] * tp[0] + dS[1][0] * tp[1] + dS[2][0] * tp[2]);
    e.Jc[1][1] = gy * (dS[0][1] * tp[0] + dS[1][1] * tp[1] + dS[2][1] * tp[2]);
    // rho
    mat3_mul(A, dRy, dS);
    e.Jc[0][2] = gx * (dS[0][0] * tp[0] + dS[1][0] * tp[1] + dS[2][0] * tp[2]);
    e.Jc[1][2] = gy * (dS[0][1] * tp[0] + dS[1][1] * tp[1] + dS[2][1] * tp[2]);
    e.Jc[0][3] = W / 2.0; e.Jc[1][3] = 0.0;
    e.Jc[0][4] = 0.0;     e.Jc[1][4] = H / 2.0;
    e.Jc[0][5] = W * l[0] / (2.0 * s * s);
    e.Jc[1][5] = H * l[1] / (2.0 * s * s);
    // point: l = S^T T p ; d l_i / d p = row i of (S^T T):  (S^T T)[i][:] = (S[0][i], S[2][i], -S[1][i])
    const double R0[3] = { S[0][0], S[2][0], -S[1][0] };
    const double R1[3] = { S[0][1], S[2][1], -S[1][1] };
    for (int k = 0; k < 3; ++k) { e.JP[0][k] = gx * R0[k] * iw; e.JP[1][k] = gy * R1[k] * iw; }
    e.JP[0][3] = -gx * (R0[0] * p[0] + R0[1] * p[1] + R0[2] * p[2]) * iw;
    e.JP[1][3] = -gy * (R1[0] * p[0] + R1[1] * p[1] + R1[2] * p[2]) * iw;
}

// ---------------------------------------------------------------------------
// Per-camera derived table.  Both functors map the inhomogeneous point p LINEARLY into the camera, l = R p with R a
// function of the camera alone, and every rotation column of the Jacobian is linear in p too: d l / d theta_c =
// D_c p.  The linearisation used to derive all of it again per OBSERVATION -- the quaternion algebra with a dozen
// divisions, or three sincos and six 3 x 3 products for the Euler model: ~900 double-precision instructions per
// observation in a point pass that those bound (profiles/r04_pmc_ba.txt).  The table holds, per camera, the first
// two rows of R and of D_0..D_2 and the scalars of the pixel map; an observation is then ~100 multiply-adds.  The
// kernels that make cameras make their tables (ba_cam_update, the back pass, chol_small, ba_cam_derive for the
// first iterate) by evaluating the functors' own expressions at the three unit points:
//   [0..5]   R[i][k]      i = 0, 1 (rows of l), k = 0..2
//   [6] 1 / s   [7] offX   [8] offY   [9] W   [10] H
//   [11] 1.0 when the camera's free columns are the first n full columns in order (colmap[t] == t), else 0.0
//   -- up to here: all a COST needs (kCamCost doubles; the candidate's table in the back pass is just these) --
//   [12..29] D_c[i][k]    at 12 + 6 c + 3 i + k, c = 0..2 (quaternion: the tangent of
//            EigenQuaternionParameterization; Euler: phi, theta, rho)
// 30 doubles: 16-byte aligned rows whose LDS copies do not all start in the same bank (32 would).
// ---------------------------------------------------------------------------
constexpr int kCamDer = 30;
constexpr int kCamCost = 12;

// l_{0,1} = (R p)_{0,1} and (want_j) d l_{0,1} / d (rotation tangent c): the camera part of eval_quat
__device__ __forceinline__ void
quat_local(const double *cam, const double p[3], bool want_j, double l[2], double dl[2][3])
{
    const double qx = cam[0], qy = cam[1], qz = cam[2], qw = cam[3];
    const double n2 = qx * qx + qy * qy + qz * qz + qw * qw;
    const double a[3] = { -qx / n2, -qy / n2, -qz / n2 };
    const double b = qw / n2;
    double t[3];
    cross3(a, p, t);
    double t2[3] = { 2.0 * t[0], 2.0 * t[1], 2.0 * t[2] };
    double at2[3];
    cross3(a, t2, at2);
    l[0] = p[0] + b * t2[0] + at2[0];
    l[1] = p[1] + b * t2[1] + at2[1];
    if (!want_j) return;
    double dl_da[3][2];
    for (int k = 0; k < 3; ++k) {
        double ek[3] = { 0.0, 0.0, 0.0 };
        ek[k] = 1.0;
        double ekp[3], ekt[3], aekp[3];
        cross3(ek, p, ekp);
        cross3(ek, t, ekt);
        cross3(a, ekp, aekp);
        dl_da[k][0] = 2.0 * b * ekp[0] + 2.0 * (ekt[0] + aekp[0]);
        dl_da[k][1] = 2.0 * b * ekp[1] + 2.0 * (ekt[1] + aekp[1]);
    }
    const double dl_db[2] = { 2.0 * t[0], 2.0 * t[1] };
    const double u[3] = { qx, qy, qz };
    const double in2 = 1.0 / n2, in4 = in2 * in2;
    double Jq[2][4];
    for (int m = 0; m < 3; ++m) {
        double s0 = 0.0, s1 = 0.0;
        for (int i = 0; i < 3; ++i) {
            const double dai = (i == m ? -in2 : 0.0) + 2.0 * u[i] * u[m] * in4;
            s0 += dl_da[i][0] * dai;
            s1 += dl_da[i][1] * dai;
        }
        const double dbm = -2.0 * qw * u[m] * in4;
        Jq[0][m] = s0 + dl_db[0] * dbm;
        Jq[1][m] = s1 + dl_db[1] * dbm;
    }
    {
        double s0 = 0.0, s1 = 0.0;
        for (int i = 0; i < 3; ++i) {
            const double dai = 2.0 * u[i] * qw * in4;
            s0 += dl_da[i][0] * dai;
            s1 += dl_da[i][1] * dai;
        }
        const double dbw = in2 - 2.0 * qw * qw * in4;
        Jq[0][3] = s0 + dl_db[0] * dbw;
        Jq[1][3] = s1 + dl_db[1] * dbw;
    }
    const double PJ[4][3] = { { qw, qz, -qy }, { -qz, qw, qx }, { qy, -qx, qw }, { -qx, -qy, -qz } };
    for (int c = 0; c < 3; ++c) {
        double s0 = 0.0, s1 = 0.0;
        for (int i = 0; i < 4; ++i) { s0 += Jq[0][i] * PJ[i][c]; s1 += Jq[1][i] * PJ[i][c]; }
        dl[0][c] = s0;
        dl[1][c] = s1;
    }
}

// the camera part of eval_euler
__device__ __forceinline__ void
euler_local(const double *cam, const double p[3], bool want_j, double l[2], double dl[2][3])
{
    const double phi = cam[0], theta = cam[1], rho = cam[2];
    const double om = theta + 1.57079632679489661923;     // M_PI_2
    double so, co, sr, cr, sp, cp;
    sincos(om, &so, &co);
    sincos(rho, &sr, &cr);
    sincos(phi, &sp, &cp);
    const double Rx[3][3] = { { 1, 0, 0 }, { 0, co, -so }, { 0, so, co } };
    const double Ry[3][3] = { { cr, -sr, 0 }, { sr, cr, 0 }, { 0, 0, 1 } };
    const double Rz[3][3] = { { cp, -sp, 0 }, { sp, cp, 0 }, { 0, 0, 1 } };
    double A[3][3], S[3][3];
    mat3_mul(Rz, Rx, A);
    mat3_mul(A, Ry, S);
    const double tp[3] = { p[0], -p[2], p[1] };            // T * p
    for (int i = 0; i < 2; ++i) l[i] = S[0][i] * tp[0] + S[1][i] * tp[1] + S[2][i] * tp[2];
    if (!want_j) return;
    const double dRx[3][3] = { { 0, 0, 0 }, { 0, -so, -co }, { 0, co, -so } };
    const double dRy[3][3] = { { -sr, -cr, 0 }, { cr, -sr, 0 }, { 0, 0, 0 } };
    const double dRz[3][3] = { { -sp, -cp, 0 }, { cp, -sp, 0 }, { 0, 0, 0 } };
    double B1[3][3], dS[3][3];
    mat3_mul(dRz, Rx, B1);
    mat3_mul(B1, Ry, dS);
    dl[0][0] = dS[0][0] * tp[0] + dS[1][0] * tp[1] + dS[2][0] * tp[2];
    dl[1][0] = dS[0][1] * tp[0] + dS[1][1] * tp[1] + dS[2][1] * tp[2];
    mat3_mul(Rz, dRx, B1);
    mat3_mul(B1, Ry, dS);
    dl[0][1] = dS[0][0] * tp[0] + dS[1][0] * tp[1] + dS[2][0] * tp[2];
    dl[1][1] = dS[0][1] * tp[0] + dS[1][1] * tp[1] + dS[2][1] * tp[2];
    mat3_mul(A, dRy, dS);
    dl[0][2] = dS[0][0] * tp[0] + dS[1][0] * tp[1] + dS[2][0] * tp[2];
    dl[1][2] = dS[0][1] * tp[0] + dS[1][1] * tp[1] + dS[2][1] * tp[2];
}

// Part `part` of a camera's table (cam: its 7 parameters): parts 0..2 probe the unit point e_part and store column
// `part` of R (and of D_0..D_2 when want_j); part 3 stores the scalars.  Four neighbouring lanes make a table
// together, or one lane calls all four parts.  out: the camera's row, stride given (kCamDer, or kCamCost for a
// cost-only copy).
__device__ __forceinline__ void
cam_derive_part(int model, const double *cam, double W, double H, const int8_t *colmap, int n, int part, bool want_j,
    double *out)
{
    if (part < 3) {
        const double pk[3] = { part == 0 ? 1.0 : 0.0, part == 1 ? 1.0 : 0.0, part == 2 ? 1.0 : 0.0 };
        double l[2], dl[2][3];
        if (model == kModelQuat) quat_local(cam, pk, want_j, l, dl);
        else euler_local(cam, pk, want_j, l, dl);
        out[part] = l[0];
        out[3 + part] = l[1];
        if (want_j) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { out[12 + 6 * c + part] = dl[0][c]; out[12 + 6 * c + 3 + part] = dl[1][c]; }
        }
        return;
    }
    const double s = model == kModelQuat ? cam[6] : cam[5];
    out[6] = 1.0 / s;
    out[7] = model == kModelQuat ? cam[4] : cam[3];
    out[8] = model == kModelQuat ? cam[5] : cam[4];
    out[9] = W;
    out[10] = H;
    bool ident = true;
#pragma unroll
    for (int t = 0; t < 6; ++t) ident = ident && (t >= n || colmap[t] == t);
    out[11] = ident ? 1.0 : 0.0;
}

// What a track gives each of its observations: p = P / w and G = (1 / w) HJ diag(scale_p), HJ the tangent basis of
// the homogeneous point -- the whole point side of the Jacobian but for the camera's R.
struct PointDer { double p[3]; double G[4][3]; };

// internal::ComputeHouseholderVector for a 4-vector (ceres householder_vector.h)
__device__ __forceinline__ void householder4(const double *x, double v[4], double &beta)
{
    const double sigma = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
    v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = 1.0;
    beta = 0.0;
    const double xp = x[3];
    if (sigma <= 2.220446049250313e-16) {
        if (xp < 0.0) beta = 2.0;
        return;
    }
    const double mu = sqrt(xp * xp + sigma);
    double vp = 1.0;
    if (xp <= 0.0) vp = xp - mu; else vp = -sigma / (xp + mu);
    beta = 2.0 * vp * vp / (sigma + vp * vp);
    v[0] /= vp; v[1] /= vp; v[2] /= vp;
}

// HomogeneousVectorParameterization(4)::ComputeJacobian: J = |x| * 0.5 * H[:, 0:3]
__device__ __forceinline__ void homog_jacobian(const double *x, double J[4][3])
{
    double v[4], beta;
    householder4(x, v, beta);
    const double xn = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
    for (int i = 0; i < 3; ++i) {
        for (int r = 0; r < 4; ++r) J[r][i] = -0.5 * beta * v[i] * v[r];
        J[i][i] += 0.5;
    }
    for (int r = 0; r < 4; ++r)
        for (int i = 0; i < 3; ++i) J[r][i] *= xn;
}

// HomogeneousVectorParameterization(4)::Plus
__device__ __forceinline__ void homog_plus(const double *x, const double *d, double *out)
{
    const double sq = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    if (sq == 0.0) { out[0] = x[0]; out[1] = x[1]; out[2] = x[2]; out[3] = x[3]; return; }
    const double nd = sqrt(sq);
    const double nd2 = 0.5 * nd;
    const double sbd = sin(nd2) / nd2;
    const double y[4] = { 0.5 * sbd * d[0], 0.5 * sbd * d[1], 0.5 * sbd * d[2], cos(nd2) };
    double v[4], beta;
    householder4(x, v, beta);
    const double xn = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
    const double vy = v[0] * y[0] + v[1] * y[1] + v[2] * y[2] + v[3] * y[3];
    for (int i = 0; i < 4; ++i) out[i] = xn * (y[i] - v[i] * (beta * vy));
}

// EigenQuaternionParameterization::Plus, storage (x, y, z, w)
__device__ __forceinline__ void quat_plus(const double *x, const double *d, double *out)
{
    const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (nd > 0.0) {
        const double sd = sin(nd) / nd;
        const double dw = cos(nd), dx = sd * d[0], dy = sd * d[1], dz = sd * d[2];
        const double xx = x[0], xy = x[1], xz = x[2], xw = x[3];
        out[3] = dw * xw - dx * xx - dy * xy - dz * xz;
        out[0] = dw * xx + dx * xw + dy * xz - dz * xy;
        out[1] = dw * xy + dy * xw + dz * xx - dx * xz;
        out[2] = dw * xz + dz * xw + dx * xy - dy * xx;
    } else {
        out[0] = x[0]; out[1] = x[1]; out[2] = x[2]; out[3] = x[3];
    }
}

// One observation, ready for the normal equations: Huber-corrected residual,
// camera tangent Jacobian restricted to the free columns (scaled), point
// tangent Jacobian (scaled).  rho0 = robustified squared norm (cost = rho0/2).
struct ObsLin {
    double r[2];
    double Jc[2][6];
    double Jp[2][3];
    double rho0;
    int n;        // free camera columns
    int off;      // offset of the camera in the tangent vector
};

// p = P / w of a track and, for a linearisation, the point side of its observations' Jacobians
__device__ __forceinline__ void point_der(const BaDev &d, int j, const double *P, bool want_j, PointDer &pd)
{
    pd.p[0] = P[0] / P[3]; pd.p[1] = P[1] / P[3]; pd.p[2] = P[2] / P[3];
    if (!want_j) return;
    if (!d.pdim) {
        for (int k = 0; k < 4; ++k)
            for (int t = 0; t < 3; ++t) pd.G[k][t] = 0.0;
        return;
    }
    const double iw = 1.0 / P[3];
    double HJ[4][3];
    homog_jacobian(P, HJ);
    for (int t = 0; t < 3; ++t) {
        const double sc = iw * d.scale_p[3 * j + t];
        for (int k = 0; k < 4; ++k) pd.G[k][t] = sc * HJ[k][t];
    }
}

// the pixel residual of the functors (pixel_residual above) from a camera's table row: l = R p, 1 / s a factor
__device__ __forceinline__ void
table_residual(const double *cd, const double p[3], double ox, double oy, double l[2], double r[2])
{
    l[0] = cd[0] * p[0] + cd[1] * p[1] + cd[2] * p[2];
    l[1] = cd[3] * p[0] + cd[4] * p[1] + cd[5] * p[2];
    r[0] = cd[9] * ((((l[0] * cd[6]) - cd[7]) * (-0.5)) + 0.5) - ox;
    r[1] = cd[10] * ((((l[1] * cd[6]) - cd[8]) * (-0.5)) + 0.5) - oy;
}

// the robustified squared residual of an observation (ox, oy) from the cost part of its camera's table row and the
// point p = P / w: what linearize_obs leaves in o.rho0
__device__ __forceinline__ double
row_cost(const double (&cd)[kCamCost], const double p[3], double ox, double oy, double huber)
{
    double l[2], r[2];
    table_residual(cd, p, ox, oy, l, r);
    const double s = r[0] * r[0] + r[1] * r[1];
    const double b = huber * huber;
    return s > b ? 2.0 * huber * sqrt(s) - b : s;
}

// ... of observation k, at the cameras whose table rows (stride doubles apart, the cost part is enough) are at tab
__device__ __forceinline__ double
obs_cost_at(const BaDev &d, int k, const double *tab, int stride, const double p[3])
{
    const int c = d.obs_cam[k];
    const double2 *q = reinterpret_cast<const double2 *>(tab + (size_t)stride * c);
    double cd[kCamCost];
#pragma unroll
    for (int i = 0; i < kCamCost / 2; ++i) { const double2 v = q[i]; cd[2 * i] = v.x; cd[2 * i + 1] = v.y; }
    const double2 xy = reinterpret_cast<const double2 *>(d.obs_xy)[k];
    return row_cost(cd, p, xy.x, xy.y, d.huber);
}

// Observation k of camera c (lay: cam_off | cam_ldim << 24 of that camera); tab: full table rows (kCamDer apart)
// of the cameras; pd: point_der of the observation's track
__device__ __forceinline__ void
linearize_obs(const BaDev &d, int k, int c, int lay, const double *tab, const PointDer &pd, ObsLin &o)
{
    const double2 *q = reinterpret_cast<const double2 *>(tab + (size_t)kCamDer * c);
    double cd[kCamDer];
#pragma unroll
    for (int i = 0; i < kCamDer / 2; ++i) { const double2 v = q[i]; cd[2 * i] = v.x; cd[2 * i + 1] = v.y; }
    const double2 xy = reinterpret_cast<const double2 *>(d.obs_xy)[k];
    o.n = lay >> 24;
    o.off = lay & 0xffffff;
    // (the camera's column scales: behind the layout word, beside the table row)
    double scl[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) scl[t] = t < o.n ? d.scale_c[o.off + t] : 0.0;
    double l[2], r[2];
    table_residual(cd, pd.p, xy.x, xy.y, l, r);
    const double s = r[0] * r[0] + r[1] * r[1];
    const double a = d.huber, b = a * a;
    double rho1 = 1.0;
    if (s > b) {
        const double rr = sqrt(s);
        o.rho0 = 2.0 * a * rr - b;
        rho1 = fmax(a / rr, 2.2250738585072014e-308);
    } else {
        o.rho0 = s;
    }
    const double sq = sqrt(rho1);
    o.r[0] = sq * r[0];
    o.r[1] = sq * r[1];
    // d r / d l = (-W / 2s, -H / 2s), with the robustifier's factor in it
    const double g[2] = { sq * (-0.5 * (cd[9] * cd[6])), sq * (-0.5 * (cd[10] * cd[6])) };
    // the six full columns: rotation tangent (D_c p), offX, offY, scale
    double J[2][6];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            const double *D = cd + 12 + 6 * cc + 3 * i;
            J[i][cc] = g[i] * (D[0] * pd.p[0] + D[1] * pd.p[1] + D[2] * pd.p[2]);
        }
        J[i][5] = -(g[i] * l[i]) * cd[6];
    }
    J[0][3] = sq * (0.5 * cd[9]); J[1][3] = 0.0;
    J[0][4] = 0.0;                J[1][4] = sq * (0.5 * cd[10]);
    // restricted to the camera's free columns.  Almost every camera frees a prefix of the full columns in order
    // (table entry 11); a camera that does not sends its wave through the select chain -- indexing the register
    // array with the column number would put it (and this kernel's hot loop) into scratch memory
    if (__all(cd[11] != 0.0)) {
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            o.Jc[0][t] = scl[t] * J[0][t];
            o.Jc[1][t] = scl[t] * J[1][t];
        }
    } else {
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            if (t < o.n) {
                const int f = d.cam_colmap[6 * c + t];
                double j0 = J[0][0], j1 = J[1][0];
#pragma unroll
                for (int ff = 1; ff < 6; ++ff) { j0 = f == ff ? J[0][ff] : j0; j1 = f == ff ? J[1][ff] : j1; }
                o.Jc[0][t] = scl[t] * j0;
                o.Jc[1][t] = scl[t] * j1;
            } else {
                o.Jc[0][t] = 0.0;
                o.Jc[1][t] = 0.0;
            }
        }
    }
    // point: d r_i / d P = g_i (R_i, -l_i) / w, times the tangent basis and the point's column scales (pd.G)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 3; ++t)
            o.Jp[i][t] = g[i] * (cd[3 * i] * pd.G[0][t] + cd[3 * i + 1] * pd.G[1][t] + cd[3 * i + 2] * pd.G[2][t] - l[i] * pd.G[3][t]);
}

// inverse of a symmetric positive definite 3x3 via Cholesky (A = L L^T, inv = L^-T L^-1); false if not PD.
// The three pivots' reciprocals are the only divisions: the substitutions of the unit vectors, written out, are
// the entries of L^-1 (the column-by-column form was 21 divisions, 300 of the point pass's instructions).
__device__ __forceinline__ bool inv3_spd(const double A[3][3], double inv[3][3])
{
    double l00 = A[0][0];
    if (!(l00 > 0.0)) return false;
    l00 = sqrt(l00);
    const double i0 = 1.0 / l00;
    const double l10 = A[1][0] * i0, l20 = A[2][0] * i0;
    double l11 = A[1][1] - l10 * l10;
    if (!(l11 > 0.0)) return false;
    l11 = sqrt(l11);
    const double i1 = 1.0 / l11;
    const double l21 = (A[2][1] - l20 * l10) * i1;
    double l22 = A[2][2] - l20 * l20 - l21 * l21;
    if (!(l22 > 0.0)) return false;
    l22 = sqrt(l22);
    const double i2 = 1.0 / l22;
    // M = L^-1 (lower triangular)
    const double m10 = -(l10 * i0) * i1;
    const double m20 = -(l20 * i0 + l21 * m10) * i2;
    const double m21 = -(l21 * i1) * i2;
    inv[0][0] = i0 * i0 + m10 * m10 + m20 * m20;
    inv[1][0] = inv[0][1] = m10 * i1 + m20 * m21;
    inv[2][0] = inv[0][2] = m20 * i2;
    inv[1][1] = i1 * i1 + m21 * m21;
    inv[2][1] = inv[1][2] = m21 * i2;
    inv[2][2] = i2 * i2;
    return true;
}

// Candidate of one camera: x+ = Plus(x, scale * step), step = -y, into out[7]; sn / xn: the camera's share of the
// step / parameter norms (ambient coordinates of the non-constant blocks).
__device__ __forceinline__ void
cam_candidate(const BaDev &d, const double *y_c, int c, double (&out)[7], double &sn, double &xn)
{
    const double *cam = d.cams + 7 * c;
    for (int i = 0; i < 7; ++i) out[i] = cam[i];
    const int n = d.cam_ldim[c], off = d.cam_off[c];
    double dl[6];
    for (int t = 0; t < 6; ++t) dl[t] = t < n ? -y_c[off + t] * d.scale_c[off + t] : 0.0;
    int t0 = 0;
    if (d.model == kModelQuat && n > 0 && d.cam_colmap[6 * c] == 0) { quat_plus(cam, dl, out); t0 = 3; }
    // unrolled with selects: out[slot] / act[f] with a run-time index would put the arrays into scratch memory
    // norms over the ambient coordinates of the non-constant blocks
    sn = 0.0; xn = 0.0;
    bool act[7] = { false, false, false, false, false, false, false };
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const bool live = t < n;
        const int f = live ? (int)d.cam_colmap[6 * c + t] : 99;
        const int slot = d.model == kModelQuat ? f + 1 : f;
        const double moved = dl[t];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            out[i] = (live && t >= t0 && i == slot) ? cam[i] + moved : out[i];
            const bool on = d.model == kModelQuat ? (f < 3 ? i < 4 : i == f + 1) : i == f;
            act[i] = act[i] || (live && on);
        }
    }
#pragma unroll
    for (int i = 0; i < 7; ++i)
        if (act[i]) { sn += (cam[i] - out[i]) * (cam[i] - out[i]); xn += cam[i] * cam[i]; }
}

// Lane `part` (0..3) of the four that make camera c's candidate: all four compute it (a hundred instructions),
// part 0 stores it and its norms, and each stores its part of the candidate's table row.
//   cams_out / partials_cam / table_out may each be null (table_stride: kCamDer with the Jacobian part, kCamCost
//   without)
__device__ __forceinline__ void
cam_update_part(const BaDev &d, const double *y_c, int c, int part, double *cams_out, double *partials_cam,
    double *table_out, int table_stride)
{
    double out[7], sn, xn;
    cam_candidate(d, y_c, c, out, sn, xn);
    if (part == 0) {
        if (cams_out)
            for (int i = 0; i < 7; ++i) cams_out[7 * c + i] = out[i];
        if (partials_cam) {
            // (write-through: the workgroup that decides reads them in the same launch, maybe from another XCD)
            store_sc1(&partials_cam[2 * c], sn);
            store_sc1(&partials_cam[2 * c + 1], xn);
        }
    }
    if (table_out)
        cam_derive_part(d.model, out, (double)d.img_w[c], (double)d.img_h[c], d.cam_colmap + 6 * c, d.cam_ldim[c], part,
            table_stride == kCamDer, table_out + (size_t)table_stride * c);
}

}  // namespace osfm
