/*
 * ba_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Double-precision CPU restatement of OrthoSfM's bundle adjustment (hot path
 * B): the two orthographic reprojection residuals, the problem that
 * orthosfm::runBundleAdjustment builds, and the Ceres trust-region solve it
 * hands that problem to.  Used only by tests/, smoke() and bench.py's
 * cpu_baseline leg.  Nothing under orthosfm_amd/ may include or call it.
 *
 * PARITY UNPINNED.  The reference solves with Ceres Solver (third party, not
 * vendored, version un-pinned: find_package(Ceres) in
 * src/bundle_adjustment/CMakeLists.txt:15; the API used -- SetParameterization,
 * EigenQuaternionParameterization, HomogeneousVectorParameterization --
 * exists only in Ceres < 2.2) and Eigen; neither is installed here, the
 * reference ships no tests or golden numbers for this path, so nothing can
 * pin these functions to reference outputs.  What IS restated from the
 * reference's own files is cited per function; the Ceres behaviour is
 * restated from the published Ceres 2.0/2.1 algorithm (trust_region_minimizer,
 * levenberg_marquardt_strategy, corrector, local_parameterization,
 * schur_eliminator) and validated by finite differences, ground-truth
 * recovery and an independent minimiser in tests/test_oracle_ba.py.
 *
 * Derivatives are obtained the way the reference obtains them: forward-mode
 * dual numbers ("jets") pushed through the residual functor
 * (ceres::AutoDiffCostFunction, OrthographicQuaternionReprojectorError.h:70-81).
 * The HIP kernels use hand-derived analytic Jacobians instead, so agreement
 * between the two is a real check.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* Same layout as osfm_ba_problem / osfm_ba_options / osfm_ba_summary in
 * include/osfm_hip.h (kept separate on purpose: the oracle does not include
 * product headers). */
typedef struct {
    int32_t model, num_cameras, num_points, num_observations;
    double *cam_params;           /* C x 7 */
    const uint8_t *cam_const;     /* C x 7 */
    const int32_t *img_width, *img_height;
    double *points;               /* M x 4 */
    const double *obs_xy;         /* O x 2 */
    const int32_t *obs_camera, *obs_point;
} ba_problem;

typedef struct {
    double huber_delta, function_tolerance, gradient_tolerance, parameter_tolerance;
    int32_t max_num_iterations, optimize_points;
    double initial_trust_region_radius, max_trust_region_radius, min_trust_region_radius;
    double min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
    int32_t jacobi_scaling, max_consecutive_invalid_steps;
    int32_t device, verbose;
} ba_options;

typedef struct {
    double initial_cost, final_cost;
    int32_t num_iterations, num_successful_steps, num_unsuccessful_steps, termination;
    double mean_point_change, max_point_change, solve_ms;
    double point_pass_ms, pair_pass_ms, cholesky_ms, back_pass_ms;
    int32_t linearizations, num_pair_entries;
} ba_summary;

enum { MODEL_QUAT = 0, MODEL_EULER = 1 };
enum { T_FUNCTION = 1, T_GRADIENT = 2, T_PARAMETER = 3, T_TRUST = 4, T_NO_CONV = 5, T_FAILURE = 6 };

/* ------------------------------------------------------------------ */
/* jets: value + NJ partials                                           */
#define NJ 11
typedef struct { double a; double v[NJ]; } jet;

static jet jc(double a) { jet r; r.a = a; memset(r.v, 0, sizeof r.v); return r; }
static jet jvar(double a, int k) { jet r = jc(a); r.v[k] = 1.0; return r; }
static jet jadd(jet x, jet y) { jet r; r.a = x.a + y.a; for (int i = 0; i < NJ; ++i) r.v[i] = x.v[i] + y.v[i]; return r; }
static jet jsub(jet x, jet y) { jet r; r.a = x.a - y.a; for (int i = 0; i < NJ; ++i) r.v[i] = x.v[i] - y.v[i]; return r; }
static jet jneg(jet x) { jet r; r.a = -x.a; for (int i = 0; i < NJ; ++i) r.v[i] = -x.v[i]; return r; }
static jet jmul(jet x, jet y) { jet r; r.a = x.a * y.a; for (int i = 0; i < NJ; ++i) r.v[i] = x.a * y.v[i] + x.v[i] * y.a; return r; }
static jet jdiv(jet x, jet y)
{
    /* ceres jet.h: h = 1/g.a; f/g = (f.a*h, (f.v - f.a*h*g.v)*h) */
    jet r; const double h = 1.0 / y.a; const double q = x.a * h;
    r.a = q;
    for (int i = 0; i < NJ; ++i) r.v[i] = (x.v[i] - q * y.v[i]) * h;
    return r;
}
static jet jsin(jet x) { jet r; r.a = sin(x.a); const double c = cos(x.a); for (int i = 0; i < NJ; ++i) r.v[i] = c * x.v[i]; return r; }
static jet jcos(jet x) { jet r; r.a = cos(x.a); const double s = -sin(x.a); for (int i = 0; i < NJ; ++i) r.v[i] = s * x.v[i]; return r; }

static void jcross(const jet a[3], const jet b[3], jet out[3])
{
    out[0] = jsub(jmul(a[1], b[2]), jmul(a[2], b[1]));
    out[1] = jsub(jmul(a[2], b[0]), jmul(a[0], b[2]));
    out[2] = jsub(jmul(a[0], b[1]), jmul(a[1], b[0]));
}

/* pixel mapping shared by both functors
 * (OrthographicQuaternionReprojectorError.h:53-61,
 *  OrthographicReprojectionError.h:66-74):
 *   x_px = W * (((l.x / scale) - offX) / (-2) + 0.5)                     */
static void jpixel(const jet l[3], jet offx, jet offy, jet scale, int w, int h,
    double ox, double oy, jet res[2])
{
    jet m2 = jc(-2.0), half = jc(0.5);
    jet xp = jmul(jc((double)w), jadd(jdiv(jsub(jdiv(l[0], scale), offx), m2), half));
    jet yp = jmul(jc((double)h), jadd(jdiv(jsub(jdiv(l[1], scale), offy), m2), half));
    res[0] = jsub(xp, jc(ox));
    res[1] = jsub(yp, jc(oy));
}

/* B1: OrthographicQuaternionReprojectionError::operator()
 * (OrthographicQuaternionReprojectorError.h:24-67).  cam = (qx,qy,qz,qw,
 * offX, offY, scale); jet slots 0-3 rotation, 4 offX, 5 offY, 6 scale,
 * 7-10 point.  Eigen semantics: Quaternion(w,x,y,z) from rotation[3],[0..2];
 * inverse() = conjugate / squaredNorm; q*v = v + w*(2 u x v) + u x (2 u x v). */
static void residual_quat_jet(const double *cam, const double *pt, int w, int h,
    double ox, double oy, jet res[2])
{
    jet q[4], offx, offy, scale, P[4];
    for (int i = 0; i < 4; ++i) q[i] = jvar(cam[i], i);
    offx = jvar(cam[4], 4); offy = jvar(cam[5], 5); scale = jvar(cam[6], 6);
    for (int i = 0; i < 4; ++i) P[i] = jvar(pt[i], 7 + i);
    jet p[3] = { jdiv(P[0], P[3]), jdiv(P[1], P[3]), jdiv(P[2], P[3]) };
    jet n2 = jadd(jadd(jmul(q[0], q[0]), jmul(q[1], q[1])), jadd(jmul(q[2], q[2]), jmul(q[3], q[3])));
    jet u[3] = { jdiv(jneg(q[0]), n2), jdiv(jneg(q[1]), n2), jdiv(jneg(q[2]), n2) };
    jet wq = jdiv(q[3], n2);
    jet uv[3];
    jcross(u, p, uv);
    for (int i = 0; i < 3; ++i) uv[i] = jadd(uv[i], uv[i]);
    jet uuv[3];
    jcross(u, uv, uuv);
    jet l[3];
    for (int i = 0; i < 3; ++i) l[i] = jadd(jadd(p[i], jmul(wq, uv[i])), uuv[i]);
    jpixel(l, offx, offy, scale, w, h, ox, oy, res);
}

/* B2: OrthographicReprojectionError::operator()
 * (OrthographicReprojectionError.h:26-77).  cam = (phi, theta, rho, offX,
 * offY, scale, -); jet slots 0 phi, 1 theta, 2 rho, 3 offX, 4 offY, 5 scale,
 * 6-9 point.  S = Rz(phi) * Rx(theta + pi/2) * Ry(rho) where "Ry" has the
 * z-axis rotation form (:45-48); l = S^T * T * p, T = [[1,0,0],[0,0,-1],[0,1,0]]. */
static void residual_euler_jet(const double *cam, const double *pt, int w, int h,
    double ox, double oy, jet res[2])
{
    jet phi = jvar(cam[0], 0), theta = jvar(cam[1], 1), rho = jvar(cam[2], 2);
    jet offx = jvar(cam[3], 3), offy = jvar(cam[4], 4), scale = jvar(cam[5], 5);
    jet P[4];
    for (int i = 0; i < 4; ++i) P[i] = jvar(pt[i], 6 + i);
    jet omega = jadd(theta, jc(M_PI_2));
    jet p[3] = { jdiv(P[0], P[3]), jdiv(P[1], P[3]), jdiv(P[2], P[3]) };
    jet z = jc(0.0), one = jc(1.0);
    jet co = jcos(omega), so = jsin(omega), cr = jcos(rho), sr = jsin(rho), cp = jcos(phi), sp = jsin(phi);
    jet Rx[3][3] = { { one, z, z }, { z, co, jneg(so) }, { z, so, co } };
    jet Ry[3][3] = { { cr, jneg(sr), z }, { sr, cr, z }, { z, z, one } };
    jet Rz[3][3] = { { cp, jneg(sp), z }, { sp, cp, z }, { z, z, one } };
    jet A[3][3], S[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            jet s = jc(0.0);
            for (int k = 0; k < 3; ++k) s = jadd(s, jmul(Rz[i][k], Rx[k][j]));
            A[i][j] = s;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            jet s = jc(0.0);
            for (int k = 0; k < 3; ++k) s = jadd(s, jmul(A[i][k], Ry[k][j]));
            S[i][j] = s;
        }
    /* T * p = (p0, -p2, p1) */
    jet tp[3] = { p[0], jneg(p[2]), p[1] };
    jet l[3];
    for (int i = 0; i < 3; ++i) {
        jet s = jc(0.0);
        for (int k = 0; k < 3; ++k) s = jadd(s, jmul(S[k][i], tp[k]));   /* S^T */
        l[i] = s;
    }
    jpixel(l, offx, offy, scale, w, h, ox, oy, res);
}

/* residual + ambient Jacobians of one observation:
 * Jc: 2 x 7 (columns = cam_params slots), Jp: 2 x 4.                     */
static void eval_obs(const ba_problem *p, int k, const double *cams, const double *pts,
    double r[2], double Jc[2][7], double Jp[2][4])
{
    const int c = p->obs_camera[k], j = p->obs_point[k];
    jet res[2];
    if (p->model == MODEL_QUAT) {
        residual_quat_jet(cams + 7 * c, pts + 4 * j, p->img_width[c], p->img_height[c],
            p->obs_xy[2 * k], p->obs_xy[2 * k + 1], res);
        for (int a = 0; a < 2; ++a) {
            r[a] = res[a].a;
            if (Jc) for (int i = 0; i < 7; ++i) Jc[a][i] = res[a].v[i];
            if (Jp) for (int i = 0; i < 4; ++i) Jp[a][i] = res[a].v[7 + i];
        }
    } else {
        residual_euler_jet(cams + 7 * c, pts + 4 * j, p->img_width[c], p->img_height[c],
            p->obs_xy[2 * k], p->obs_xy[2 * k + 1], res);
        for (int a = 0; a < 2; ++a) {
            r[a] = res[a].a;
            if (Jc) { for (int i = 0; i < 6; ++i) Jc[a][i] = res[a].v[i]; Jc[a][6] = 0.0; }
            if (Jp) for (int i = 0; i < 4; ++i) Jp[a][i] = res[a].v[6 + i];
        }
    }
}

ORACLE_API void
oracle_ba_residuals(const ba_problem *p, double *residuals, double *err)
{
    for (int k = 0; k < p->num_observations; ++k) {
        double r[2];
        eval_obs(p, k, p->cam_params, p->points, r, NULL, NULL);
        if (residuals) { residuals[2 * k] = r[0]; residuals[2 * k + 1] = r[1]; }
        /* B7: evaluateReprojectionError, OrthoQuaternionRecoAlgorithm.cpp:192 */
        if (err) err[k] = sqrt(r[0] * r[0] + r[1] * r[1]);
    }
}

ORACLE_API void
oracle_ba_jacobian(const ba_problem *p, int k, double *r2, double *jc_2x7, double *jp_2x4)
{
    double r[2], Jc[2][7], Jp[2][4];
    eval_obs(p, k, p->cam_params, p->points, r, Jc, Jp);
    r2[0] = r[0]; r2[1] = r[1];
    memcpy(jc_2x7, Jc, sizeof Jc);
    memcpy(jp_2x4, Jp, sizeof Jp);
}

/* ------------------------------------------------------------------ */
/* local parameterisations (Ceres < 2.2, local_parameterization.cc)     */

/* EigenQuaternionParameterization::Plus: x+ = [sin|d|/|d| d, cos|d|] (x) x,
 * storage order (x, y, z, w). */
static void quat_plus(const double *x, const double *d, double *out)
{
    const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (nd > 0.0) {
        const double s = sin(nd) / nd;
        const double dw = cos(nd), dx = s * d[0], dy = s * d[1], dz = s * d[2];
        const double xx = x[0], xy = x[1], xz = x[2], xw = x[3];
        /* Eigen quaternion product (dw,dx,dy,dz) * (xw,xx,xy,xz) */
        out[3] = dw * xw - dx * xx - dy * xy - dz * xz;
        out[0] = dw * xx + dx * xw + dy * xz - dz * xy;
        out[1] = dw * xy + dy * xw + dz * xx - dx * xz;
        out[2] = dw * xz + dz * xw + dx * xy - dy * xx;
    } else {
        for (int i = 0; i < 4; ++i) out[i] = x[i];
    }
}

/* EigenQuaternionParameterization::ComputeJacobian, 4 x 3 row-major. */
static void quat_plus_jacobian(const double *x, double J[4][3])
{
    J[0][0] = x[3];  J[0][1] = x[2];  J[0][2] = -x[1];
    J[1][0] = -x[2]; J[1][1] = x[3];  J[1][2] = x[0];
    J[2][0] = x[1];  J[2][1] = -x[0]; J[2][2] = x[3];
    J[3][0] = -x[0]; J[3][1] = -x[1]; J[3][2] = -x[2];
}

/* internal::ComputeHouseholderVector (householder_vector.h) for size 4. */
static void householder4(const double *x, double v[4], double *beta)
{
    const double sigma = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
    v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = 1.0;
    *beta = 0.0;
    const double xp = x[3];
    if (sigma <= DBL_EPSILON) {
        if (xp < 0.0) *beta = 2.0;
        return;
    }
    const double mu = sqrt(xp * xp + sigma);
    double vp = 1.0;
    if (xp <= 0.0) vp = xp - mu; else vp = -sigma / (xp + mu);
    *beta = 2.0 * vp * vp / (sigma + vp * vp);
    v[0] /= vp; v[1] /= vp; v[2] /= vp;
}

/* HomogeneousVectorParameterization(4)::Plus. */
static void homog_plus(const double *x, const double *d, double *out)
{
    const double sq = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    if (sq == 0.0) { for (int i = 0; i < 4; ++i) out[i] = x[i]; return; }
    const double nd = sqrt(sq);
    const double nd2 = 0.5 * nd;
    const double sbd = sin(nd2) / nd2;
    double y[4] = { 0.5 * sbd * d[0], 0.5 * sbd * d[1], 0.5 * sbd * d[2], cos(nd2) };
    double v[4], beta;
    householder4(x, v, &beta);
    const double xn = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
    const double vy = v[0] * y[0] + v[1] * y[1] + v[2] * y[2] + v[3] * y[3];
    for (int i = 0; i < 4; ++i) out[i] = xn * (y[i] - v[i] * (beta * vy));
}

/* HomogeneousVectorParameterization(4)::ComputeJacobian, 4 x 3. */
static void homog_plus_jacobian(const double *x, double J[4][3])
{
    double v[4], beta;
    householder4(x, v, &beta);
    const double xn = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
    for (int i = 0; i < 3; ++i) {
        for (int r = 0; r < 4; ++r) J[r][i] = -0.5 * beta * v[i] * v[r];
        J[i][i] += 0.5;
    }
    for (int r = 0; r < 4; ++r)
        for (int i = 0; i < 3; ++i) J[r][i] *= xn;
}

ORACLE_API void oracle_quat_plus(const double *x, const double *d, double *out) { quat_plus(x, d, out); }
ORACLE_API void oracle_homog_plus(const double *x, const double *d, double *out) { homog_plus(x, d, out); }
ORACLE_API void oracle_homog_plus_jacobian(const double *x, double *J12) { double J[4][3]; homog_plus_jacobian(x, J); memcpy(J12, J, sizeof J); }

/* ------------------------------------------------------------------ */
/* problem structure: which tangent columns exist                       */
/*
 * Parameter blocks as the reference creates them
 * (OrthoQuaternionRecoAlgorithm.cpp:121-148 / OrthographicReconstruction-
 * Algorithm.cpp:148-178, bundle_adjustment.cpp:86-97):
 *   QUAT : rot(4 -> 3 tangent), offX, offY, scale
 *   EULER: phi, theta, roll, offX, offY, scale
 *   point: 4 -> 3 tangent (constant when !optimize_points)
 * cam_const[c][slot] != 0 marks the block holding that slot constant.
 */
typedef struct {
    int C, M, O;
    int *cam_ldim;      /* tangent size per camera (0..6) */
    int *cam_off;       /* offset into the camera part of the tangent vector */
    int (*cam_cols)[6]; /* for each tangent column: ambient slot (QUAT rot: 0,1,2 = delta index) */
    int ncam;           /* total camera tangent columns */
    int pdim;           /* 3 or 0 */
    int *pt_start;      /* observations of point j: [pt_start[j], pt_start[j+1]) */
} ba_layout;

static int block_of_slot(int model, int slot) { return (model == MODEL_QUAT && slot < 4) ? 0 : slot; }

static void layout_build(const ba_problem *p, const ba_options *o, ba_layout *L)
{
    L->C = p->num_cameras; L->M = p->num_points; L->O = p->num_observations;
    L->cam_ldim = calloc(L->C + 1, sizeof(int));
    L->cam_off = calloc(L->C + 1, sizeof(int));
    L->cam_cols = calloc(L->C + 1, sizeof(int[6]));
    int tot = 0;
    for (int c = 0; c < L->C; ++c) {
        const uint8_t *cc = p->cam_const + 7 * c;
        int n = 0;
        if (p->model == MODEL_QUAT) {
            if (!cc[0]) { L->cam_cols[c][n++] = 0; L->cam_cols[c][n++] = 1; L->cam_cols[c][n++] = 2; }
            for (int s = 4; s < 7; ++s) if (!cc[s]) L->cam_cols[c][n++] = s;
        } else {
            for (int s = 0; s < 6; ++s) if (!cc[s]) L->cam_cols[c][n++] = s;
        }
        L->cam_ldim[c] = n; L->cam_off[c] = tot; tot += n;
    }
    L->ncam = tot;
    L->pdim = o->optimize_points ? 3 : 0;
    L->pt_start = calloc(L->M + 2, sizeof(int));
    for (int k = 0; k < L->O; ++k) L->pt_start[p->obs_point[k] + 1]++;
    for (int j = 0; j < L->M; ++j) L->pt_start[j + 1] += L->pt_start[j];
    (void)block_of_slot;
}

static void layout_free(ba_layout *L)
{
    free(L->cam_ldim); free(L->cam_off); free(L->cam_cols); free(L->pt_start);
}

/* ------------------------------------------------------------------ */
/* evaluation: cost, corrected residuals, tangent Jacobian blocks       */
/*
 * HuberLoss(a) (ceres loss_function.cc): s = |r|^2, b = a^2;
 *   s <= b: rho = (s, 1, 0);  else r = sqrt(s): rho = (2 a r - b, max(min, a/r), -rho1/(2s))
 * Corrector (corrector.cc): rho2 <= 0 -> residuals and Jacobian rows are
 * scaled by sqrt(rho1).  cost = 1/2 sum rho0.
 */
typedef struct {
    double *r;     /* 2 O corrected residuals */
    double *Jc;    /* O x 2 x 6 tangent camera Jacobian (first cam_ldim columns used) */
    double *Jp;    /* O x 2 x 3 tangent point Jacobian */
} ba_lin;

static double evaluate(const ba_problem *p, const ba_options *o, const ba_layout *L,
    const double *cams, const double *pts, ba_lin *lin)
{
    const double a = o->huber_delta, b = a * a;
    double cost = 0.0;
#pragma omp parallel for reduction(+ : cost) schedule(static)
    for (int k = 0; k < L->O; ++k) {
        const int c = p->obs_camera[k], j = p->obs_point[k];
        double r[2], Jc[2][7], Jp[2][4];
        eval_obs(p, k, cams, pts, r, lin ? Jc : NULL, lin ? Jp : NULL);
        const double s = r[0] * r[0] + r[1] * r[1];
        double rho0, rho1;
        if (s > b) {
            const double rr = sqrt(s);
            rho0 = 2.0 * a * rr - b;
            rho1 = a / rr;
            if (rho1 < DBL_MIN) rho1 = DBL_MIN;
        } else { rho0 = s; rho1 = 1.0; }
        cost += 0.5 * rho0;
        if (!lin) continue;
        const double sq = sqrt(rho1);
        lin->r[2 * k] = sq * r[0]; lin->r[2 * k + 1] = sq * r[1];
        /* tangent Jacobians */
        double PJ[4][3];
        if (p->model == MODEL_QUAT) quat_plus_jacobian(cams + 7 * c, PJ);
        const int n = L->cam_ldim[c];
        for (int aa = 0; aa < 2; ++aa) {
            for (int t = 0; t < n; ++t) {
                const int slot = L->cam_cols[c][t];
                double v;
                if (p->model == MODEL_QUAT && slot < 3 && !p->cam_const[7 * c]) {
                    v = 0.0;
                    for (int i = 0; i < 4; ++i) v += Jc[aa][i] * PJ[i][slot];
                } else v = Jc[aa][slot];
                lin->Jc[(size_t)k * 12 + aa * 6 + t] = sq * v;
            }
        }
        if (L->pdim) {
            double HJ[4][3];
            homog_plus_jacobian(pts + 4 * j, HJ);
            for (int aa = 0; aa < 2; ++aa)
                for (int t = 0; t < 3; ++t) {
                    double v = 0.0;
                    for (int i = 0; i < 4; ++i) v += Jp[aa][i] * HJ[i][t];
                    lin->Jp[(size_t)k * 6 + aa * 3 + t] = sq * v;
                }
        }
    }
    return cost;
}

/* x (+) delta for every block (Evaluator::Plus). */
static void plus_all(const ba_problem *p, const ba_layout *L, const double *cams, const double *pts,
    const double *dc, const double *dp, double *cams_out, double *pts_out)
{
    for (int c = 0; c < L->C; ++c) {
        const double *x = cams + 7 * c; double *y = cams_out + 7 * c;
        for (int i = 0; i < 7; ++i) y[i] = x[i];
        const int n = L->cam_ldim[c];
        const double *d = dc + L->cam_off[c];
        int t = 0;
        if (p->model == MODEL_QUAT && !p->cam_const[7 * c]) { quat_plus(x, d, y); t = 3; }
        for (; t < n; ++t) { const int slot = L->cam_cols[c][t]; y[slot] = x[slot] + d[t]; }
    }
    for (int j = 0; j < L->M; ++j) {
        if (L->pdim) homog_plus(pts + 4 * j, dp + 3 * j, pts_out + 4 * j);
        else for (int i = 0; i < 4; ++i) pts_out[4 * j + i] = pts[4 * j + i];
    }
}

/* dense Cholesky solve of the symmetric system A x = b (lower triangle used);
 * returns 0 when A is not positive definite. */
static int cholesky_solve(double *A, int n, double *b)
{
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0.0)) return 0;
        d = sqrt(d);
        A[(size_t)j * n + j] = d;
#pragma omp parallel for schedule(static) if (n - j > 256)
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * n + j];
            const double *ri = A + (size_t)i * n, *rj = A + (size_t)j * n;
            for (int k = 0; k < j; ++k) s -= ri[k] * rj[k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= A[(size_t)i * n + k] * b[k];
        b[i] = s / A[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= A[(size_t)k * n + i] * b[k];
        b[i] = s / A[(size_t)i * n + i];
    }
    return 1;
}

static int inv3_spd(const double A[3][3], double inv[3][3])
{
    /* via Cholesky like Ceres' InvertPSDMatrix for fixed-size blocks */
    double l00 = A[0][0]; if (!(l00 > 0)) return 0; l00 = sqrt(l00);
    double l10 = A[1][0] / l00, l20 = A[2][0] / l00;
    double l11 = A[1][1] - l10 * l10; if (!(l11 > 0)) return 0; l11 = sqrt(l11);
    double l21 = (A[2][1] - l20 * l10) / l11;
    double l22 = A[2][2] - l20 * l20 - l21 * l21; if (!(l22 > 0)) return 0; l22 = sqrt(l22);
    for (int c = 0; c < 3; ++c) {
        double e[3] = { c == 0, c == 1, c == 2 };
        double y0 = e[0] / l00;
        double y1 = (e[1] - l10 * y0) / l11;
        double y2 = (e[2] - l20 * y0 - l21 * y1) / l22;
        double x2 = y2 / l22;
        double x1 = (y1 - l21 * x2) / l11;
        double x0 = (y0 - l10 * x1 - l20 * x2) / l00;
        inv[0][c] = x0; inv[1][c] = x1; inv[2][c] = x2;
    }
    return 1;
}

static double now_ms(void)
{
#ifdef _OPENMP
    return omp_get_wtime() * 1e3;
#else
    return 0.0;
#endif
}

/*
 * runBundleAdjustment's ceres::Solve (bundle_adjustment.cpp:126-145):
 * TrustRegionMinimizer + LevenbergMarquardtStrategy + Schur elimination of
 * the point blocks + dense Cholesky of the reduced camera system.
 */
ORACLE_API int
oracle_ba_solve(const ba_problem *p, const ba_options *o, ba_summary *sum)
{
    ba_layout L;
    layout_build(p, o, &L);
    const int C = L.C, M = L.M, O = L.O, nc = L.ncam, np = L.pdim * M;
    const double t_start = now_ms();
    memset(sum, 0, sizeof *sum);

    double *cams = malloc(sizeof(double) * 7 * (C + 1)), *pts = malloc(sizeof(double) * 4 * (M + 1));
    double *cams_c = malloc(sizeof(double) * 7 * (C + 1)), *pts_c = malloc(sizeof(double) * 4 * (M + 1));
    double *pts0 = malloc(sizeof(double) * 4 * (M + 1));
    memcpy(cams, p->cam_params, sizeof(double) * 7 * C);
    memcpy(pts, p->points, sizeof(double) * 4 * M);
    memcpy(pts0, p->points, sizeof(double) * 4 * M);
    ba_lin lin;
    lin.r = calloc((size_t)2 * O + 2, sizeof(double));
    lin.Jc = calloc((size_t)12 * O + 12, sizeof(double));
    lin.Jp = calloc((size_t)6 * O + 6, sizeof(double));
    double *scale_c = malloc(sizeof(double) * (nc + 1)), *scale_p = malloc(sizeof(double) * (np + 1));
    double *diag_c = malloc(sizeof(double) * (nc + 1)), *diag_p = malloc(sizeof(double) * (np + 1));
    double *gc = malloc(sizeof(double) * (nc + 1)), *gp = malloc(sizeof(double) * (np + 1));
    double *S = malloc(sizeof(double) * ((size_t)nc * nc + 1)), *rhs = malloc(sizeof(double) * (nc + 1));
    double *step_c = malloc(sizeof(double) * (nc + 1)), *step_p = malloc(sizeof(double) * (np + 1));
    double *dc = malloc(sizeof(double) * (nc + 1)), *dp = malloc(sizeof(double) * (np + 1));
    double *Vinv = malloc(sizeof(double) * 9 * (M + 1)), *ge = malloc(sizeof(double) * 3 * (M + 1));
    for (int i = 0; i < nc; ++i) scale_c[i] = 1.0;
    for (int i = 0; i < np; ++i) scale_p[i] = 1.0;

    /* parameters that belong to the optimisation (for |x| and step norms):
     * every ambient coordinate of every non-constant block */
    #define FOR_ACTIVE_CAM_SLOTS(c, body)                                        \
        for (int slot_ = 0; slot_ < 7; ++slot_) {                                \
            int act_;                                                            \
            if (p->model == MODEL_QUAT) act_ = !p->cam_const[7 * (c) + (slot_ < 4 ? 0 : slot_)]; \
            else act_ = slot_ < 6 && !p->cam_const[7 * (c) + slot_];            \
            if (act_) { const int slot = slot_; body; }                          \
        }

    double x_cost, x_norm = 0.0, radius = o->initial_trust_region_radius, decrease_factor = 2.0;
    int reuse_diagonal = 0, invalid_steps = 0, iteration = 0, term = T_NO_CONV;
    double grad_max = 0.0;

    /* ---- EvaluateGradientAndJacobian ---- */
    #define EVAL_GRADIENT(first)                                                                \
        do {                                                                                    \
            x_cost = evaluate(p, o, &L, cams, pts, &lin);                                       \
            /* gradient from the UNSCALED Jacobian */                                           \
            memset(gc, 0, sizeof(double) * nc); memset(gp, 0, sizeof(double) * (np + 1));       \
            for (int k = 0; k < O; ++k) {                                                       \
                const int c = p->obs_camera[k], j = p->obs_point[k];                            \
                const double *Jc = lin.Jc + (size_t)k * 12, *Jp = lin.Jp + (size_t)k * 6;       \
                for (int t = 0; t < L.cam_ldim[c]; ++t)                                         \
                    gc[L.cam_off[c] + t] += Jc[t] * lin.r[2 * k] + Jc[6 + t] * lin.r[2 * k + 1];\
                for (int t = 0; t < L.pdim; ++t)                                                \
                    gp[3 * j + t] += Jp[t] * lin.r[2 * k] + Jp[3 + t] * lin.r[2 * k + 1];       \
            }                                                                                   \
            if (o->jacobi_scaling) {                                                            \
                if (first) {                                                                    \
                    for (int i = 0; i < nc; ++i) scale_c[i] = 0.0;                              \
                    for (int i = 0; i < np; ++i) scale_p[i] = 0.0;                              \
                    for (int k = 0; k < O; ++k) {                                               \
                        const int c = p->obs_camera[k], j = p->obs_point[k];                    \
                        const double *Jc = lin.Jc + (size_t)k * 12, *Jp = lin.Jp + (size_t)k * 6;\
                        for (int t = 0; t < L.cam_ldim[c]; ++t)                                 \
                            scale_c[L.cam_off[c] + t] += Jc[t] * Jc[t] + Jc[6 + t] * Jc[6 + t]; \
                        for (int t = 0; t < L.pdim; ++t)                                        \
                            scale_p[3 * j + t] += Jp[t] * Jp[t] + Jp[3 + t] * Jp[3 + t];        \
                    }                                                                           \
                    for (int i = 0; i < nc; ++i) scale_c[i] = 1.0 / (1.0 + sqrt(scale_c[i]));   \
                    for (int i = 0; i < np; ++i) scale_p[i] = 1.0 / (1.0 + sqrt(scale_p[i]));   \
                }                                                                               \
                for (int k = 0; k < O; ++k) {                                                   \
                    const int c = p->obs_camera[k], j = p->obs_point[k];                        \
                    double *Jc = lin.Jc + (size_t)k * 12, *Jp = lin.Jp + (size_t)k * 6;         \
                    for (int t = 0; t < L.cam_ldim[c]; ++t) {                                   \
                        Jc[t] *= scale_c[L.cam_off[c] + t]; Jc[6 + t] *= scale_c[L.cam_off[c] + t]; } \
                    for (int t = 0; t < L.pdim; ++t) {                                          \
                        Jp[t] *= scale_p[3 * j + t]; Jp[3 + t] *= scale_p[3 * j + t]; }         \
                }                                                                               \
            }                                                                                   \
            /* |Plus(x, -g) - x|_inf */                                                         \
            for (int i = 0; i < nc; ++i) dc[i] = -gc[i];                                        \
            for (int i = 0; i < np; ++i) dp[i] = -gp[i];                                        \
            plus_all(p, &L, cams, pts, dc, dp, cams_c, pts_c);                                  \
            grad_max = 0.0;                                                                     \
            for (int c = 0; c < C; ++c) FOR_ACTIVE_CAM_SLOTS(c, {                               \
                const double d_ = fabs(cams[7 * c + slot] - cams_c[7 * c + slot]);              \
                if (d_ > grad_max) grad_max = d_; })                                            \
            if (L.pdim) for (int i = 0; i < 4 * M; ++i) {                                       \
                const double d_ = fabs(pts[i] - pts_c[i]); if (d_ > grad_max) grad_max = d_; }  \
        } while (0)

    #define X_NORM()                                                                            \
        do {                                                                                    \
            double s_ = 0.0;                                                                    \
            for (int c = 0; c < C; ++c) FOR_ACTIVE_CAM_SLOTS(c, { s_ += cams[7 * c + slot] * cams[7 * c + slot]; }) \
            if (L.pdim) for (int i = 0; i < 4 * M; ++i) s_ += pts[i] * pts[i];                  \
            x_norm = sqrt(s_);                                                                  \
        } while (0)

    X_NORM();
    EVAL_GRADIENT(1);
    sum->initial_cost = x_cost;
    if (grad_max <= o->gradient_tolerance) { term = T_GRADIENT; goto done; }

    int last_successful = 0;
    for (;;) {
        /* FinalizeIterationAndCheckIfMinimizerCanContinue: max iterations,
         * then gradient tolerance (after a successful step), then radius */
        if (iteration >= o->max_num_iterations) { term = T_NO_CONV; break; }
        if (last_successful && grad_max <= o->gradient_tolerance) { term = T_GRADIENT; break; }
        if (radius <= o->min_trust_region_radius) { term = T_TRUST; break; }
        iteration++;
        last_successful = 0;

        /* ---- LevenbergMarquardtStrategy::ComputeStep ---- */
        if (!reuse_diagonal) {
            memset(diag_c, 0, sizeof(double) * nc); memset(diag_p, 0, sizeof(double) * (np + 1));
            for (int k = 0; k < O; ++k) {
                const int c = p->obs_camera[k], j = p->obs_point[k];
                const double *Jc = lin.Jc + (size_t)k * 12, *Jp = lin.Jp + (size_t)k * 6;
                for (int t = 0; t < L.cam_ldim[c]; ++t)
                    diag_c[L.cam_off[c] + t] += Jc[t] * Jc[t] + Jc[6 + t] * Jc[6 + t];
                for (int t = 0; t < L.pdim; ++t)
                    diag_p[3 * j + t] += Jp[t] * Jp[t] + Jp[3 + t] * Jp[3 + t];
            }
            for (int i = 0; i < nc; ++i) diag_c[i] = fmin(fmax(diag_c[i], o->min_lm_diagonal), o->max_lm_diagonal);
            for (int i = 0; i < np; ++i) diag_p[i] = fmin(fmax(diag_p[i], o->min_lm_diagonal), o->max_lm_diagonal);
        }
        /* lm_diagonal = sqrt(diag / radius); the solver adds D^T D = diag / radius */
        /* ---- Schur complement: S y_c = rhs ---- */
        memset(S, 0, sizeof(double) * (size_t)nc * nc);
        memset(rhs, 0, sizeof(double) * nc);
        for (int k = 0; k < O; ++k) {
            const int c = p->obs_camera[k];
            const double *Jc = lin.Jc + (size_t)k * 12;
            const int n = L.cam_ldim[c], off = L.cam_off[c];
            for (int a = 0; a < n; ++a) {
                rhs[off + a] += Jc[a] * lin.r[2 * k] + Jc[6 + a] * lin.r[2 * k + 1];
                for (int b = 0; b < n; ++b)
                    S[(size_t)(off + a) * nc + off + b] += Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b];
            }
        }
        for (int i = 0; i < nc; ++i) S[(size_t)i * nc + i] += diag_c[i] / radius;
        int solve_ok = 1;
        if (L.pdim) {
            for (int j = 0; j < M; ++j) {
                const int k0 = L.pt_start[j], k1 = L.pt_start[j + 1];
                double V[3][3] = { { 0 } }, g[3] = { 0, 0, 0 };
                for (int k = k0; k < k1; ++k) {
                    const double *Jp = lin.Jp + (size_t)k * 6;
                    for (int a = 0; a < 3; ++a) {
                        g[a] += Jp[a] * lin.r[2 * k] + Jp[3 + a] * lin.r[2 * k + 1];
                        for (int b = 0; b < 3; ++b) V[a][b] += Jp[a] * Jp[b] + Jp[3 + a] * Jp[3 + b];
                    }
                }
                for (int a = 0; a < 3; ++a) V[a][a] += diag_p[3 * j + a] / radius;
                double Vi[3][3];
                if (k1 > k0 && !inv3_spd(V, Vi)) { solve_ok = 0; break; }
                if (k1 == k0) { memset(Vi, 0, sizeof Vi); for (int a = 0; a < 3; ++a) Vi[a][a] = 1.0 / V[a][a]; }
                memcpy(Vinv + 9 * j, Vi, sizeof Vi);
                ge[3 * j] = g[0]; ge[3 * j + 1] = g[1]; ge[3 * j + 2] = g[2];
                /* W_k = Jc_k^T Jp_k (n x 3);  S -= W_k1 Vi W_k2^T;  rhs -= W_k Vi g */
                for (int k = k0; k < k1; ++k) {
                    const int c1 = p->obs_camera[k], n1 = L.cam_ldim[c1], o1 = L.cam_off[c1];
                    const double *Jc1 = lin.Jc + (size_t)k * 12, *Jp1 = lin.Jp + (size_t)k * 6;
                    double Z[6][3];      /* W_k * Vi */
                    for (int a = 0; a < n1; ++a) {
                        double W[3];
                        for (int t = 0; t < 3; ++t) W[t] = Jc1[a] * Jp1[t] + Jc1[6 + a] * Jp1[3 + t];
                        for (int t = 0; t < 3; ++t) Z[a][t] = W[0] * Vi[0][t] + W[1] * Vi[1][t] + W[2] * Vi[2][t];
                        rhs[o1 + a] -= Z[a][0] * g[0] + Z[a][1] * g[1] + Z[a][2] * g[2];
                    }
                    for (int k2 = k0; k2 < k1; ++k2) {
                        const int c2 = p->obs_camera[k2], n2 = L.cam_ldim[c2], o2 = L.cam_off[c2];
                        const double *Jc2 = lin.Jc + (size_t)k2 * 12, *Jp2 = lin.Jp + (size_t)k2 * 6;
                        for (int a = 0; a < n1; ++a)
                            for (int b2 = 0; b2 < n2; ++b2) {
                                double W2[3];
                                for (int t = 0; t < 3; ++t) W2[t] = Jc2[b2] * Jp2[t] + Jc2[6 + b2] * Jp2[3 + t];
                                S[(size_t)(o1 + a) * nc + o2 + b2] -= Z[a][0] * W2[0] + Z[a][1] * W2[1] + Z[a][2] * W2[2];
                            }
                    }
                }
            }
        }
        if (solve_ok && nc > 0) {
            memcpy(step_c, rhs, sizeof(double) * nc);
            solve_ok = cholesky_solve(S, nc, step_c);
        }
        if (solve_ok && L.pdim) {
            /* back substitution: y_p = Vi (g - sum_k W_k^T y_c) */
            for (int j = 0; j < M; ++j) {
                double t3[3] = { ge[3 * j], ge[3 * j + 1], ge[3 * j + 2] };
                for (int k = L.pt_start[j]; k < L.pt_start[j + 1]; ++k) {
                    const int c = p->obs_camera[k], n = L.cam_ldim[c], off = L.cam_off[c];
                    const double *Jc = lin.Jc + (size_t)k * 12, *Jp = lin.Jp + (size_t)k * 6;
                    double u0 = 0, u1 = 0;       /* Jc y_c */
                    for (int a = 0; a < n; ++a) { u0 += Jc[a] * step_c[off + a]; u1 += Jc[6 + a] * step_c[off + a]; }
                    for (int t = 0; t < 3; ++t) t3[t] -= Jp[t] * u0 + Jp[3 + t] * u1;
                }
                const double *Vi = Vinv + 9 * j;
                for (int a = 0; a < 3; ++a) step_p[3 * j + a] = Vi[3 * a] * t3[0] + Vi[3 * a + 1] * t3[1] + Vi[3 * a + 2] * t3[2];
            }
        }
        int step_valid = 0;
        double model_cost_change = 0.0;
        if (solve_ok) {
            int finite = 1;
            for (int i = 0; i < nc && finite; ++i) finite = isfinite(step_c[i]);
            for (int i = 0; i < np && finite; ++i) finite = isfinite(step_p[i]);
            solve_ok = finite;
        }
        reuse_diagonal = 1;
        if (solve_ok) {
            /* the solver returns y with J y ~ r; the step is -y */
            for (int i = 0; i < nc; ++i) step_c[i] = -step_c[i];
            for (int i = 0; i < np; ++i) step_p[i] = -step_p[i];
            /* model_cost_change = -(J step)^T (r + J step / 2) */
            double mcc = 0.0;
            for (int k = 0; k < O; ++k) {
                const int c = p->obs_camera[k], j = p->obs_point[k];
                const double *Jc = lin.Jc + (size_t)k * 12, *Jp = lin.Jp + (size_t)k * 6;
                double m0 = 0, m1 = 0;
                for (int a = 0; a < L.cam_ldim[c]; ++a) { m0 += Jc[a] * step_c[L.cam_off[c] + a]; m1 += Jc[6 + a] * step_c[L.cam_off[c] + a]; }
                for (int t = 0; t < L.pdim; ++t) { m0 += Jp[t] * step_p[3 * j + t]; m1 += Jp[3 + t] * step_p[3 * j + t]; }
                mcc -= m0 * (lin.r[2 * k] + m0 / 2.0) + m1 * (lin.r[2 * k + 1] + m1 / 2.0);
            }
            model_cost_change = mcc;
            step_valid = model_cost_change > 0.0;
        }
        if (!step_valid) {
            /* HandleInvalidStep */
            if (++invalid_steps >= o->max_consecutive_invalid_steps) { term = T_FAILURE; break; }
            radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = 1;
            sum->num_unsuccessful_steps++;
            continue;
        }
        invalid_steps = 0;
        for (int i = 0; i < nc; ++i) dc[i] = step_c[i] * scale_c[i];
        for (int i = 0; i < np; ++i) dp[i] = step_p[i] * scale_p[i];

        /* ---- candidate point and its cost ---- */
        plus_all(p, &L, cams, pts, dc, dp, cams_c, pts_c);
        const double cand_cost = evaluate(p, o, &L, cams_c, pts_c, NULL);

        /* ParameterToleranceReached */
        double sn = 0.0;
        for (int c = 0; c < C; ++c) FOR_ACTIVE_CAM_SLOTS(c, { const double d_ = cams[7 * c + slot] - cams_c[7 * c + slot]; sn += d_ * d_; })
        if (L.pdim) for (int i = 0; i < 4 * M; ++i) { const double d_ = pts[i] - pts_c[i]; sn += d_ * d_; }
        const double step_norm = sqrt(sn);
        if (step_norm <= o->parameter_tolerance * (x_norm + o->parameter_tolerance)) { term = T_PARAMETER; break; }
        /* FunctionToleranceReached */
        const double cost_change = x_cost - cand_cost;
        if (fabs(cost_change) <= o->function_tolerance * x_cost) { term = T_FUNCTION; break; }

        const double relative_decrease = cost_change / model_cost_change;
        if (relative_decrease > o->min_relative_decrease) {
            /* HandleSuccessfulStep */
            memcpy(cams, cams_c, sizeof(double) * 7 * C);
            memcpy(pts, pts_c, sizeof(double) * 4 * M);
            X_NORM();
            EVAL_GRADIENT(0);
            radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * relative_decrease - 1.0, 3));
            radius = fmin(o->max_trust_region_radius, radius);
            decrease_factor = 2.0; reuse_diagonal = 0;
            sum->num_successful_steps++;
            last_successful = 1;
        } else {
            radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = 1;
            sum->num_unsuccessful_steps++;
        }
    }

done:
    /* Ceres writes back the best (= current, monotonic steps) iterate; on
     * convergence by function/parameter tolerance the candidate is NOT taken */
    memcpy(p->cam_params, cams, sizeof(double) * 7 * C);
    memcpy(p->points, pts, sizeof(double) * 4 * M);
    sum->final_cost = x_cost;
    sum->num_iterations = iteration;
    sum->termination = term;
    {   /* bundle_adjustment.cpp:150-160 */
        double mx = 0.0, s = 0.0;
        for (int j = 0; j < M; ++j) {
            double d = 0.0;
            for (int i = 0; i < 4; ++i) { const double e = pts0[4 * j + i] - pts[4 * j + i]; d += e * e; }
            d = sqrt(d); if (d > mx) mx = d; s += d;
        }
        sum->mean_point_change = M ? s / M : 0.0; sum->max_point_change = mx;
    }
    sum->solve_ms = now_ms() - t_start;
    free(cams); free(pts); free(cams_c); free(pts_c); free(pts0);
    free(lin.r); free(lin.Jc); free(lin.Jp);
    free(scale_c); free(scale_p); free(diag_c); free(diag_p); free(gc); free(gp);
    free(S); free(rhs); free(step_c); free(step_p); free(dc); free(dp); free(Vinv); free(ge);
    layout_free(&L);
    return 0;
}

/* ------------------------------------------------------------------ */
/* B8: triangulateOrthographicTracks + intersectRays
 * (src/triangulation/triangulation.cpp:11-93) with the camera accessors
 * getPointOnCameraPlane / getLookDirection
 * (OrthoQuaternionCamera.cpp:45-59, OrthographicCamera.cpp:63-95,187-193).  */

static void quat_rot(const double *q, const double v[3], double out[3])
{
    /* Eigen q * v for a (unit) quaternion: v + w*(2 u x v) + u x (2 u x v) */
    const double ux = q[0], uy = q[1], uz = q[2], w = q[3];
    double t[3] = { 2 * (uy * v[2] - uz * v[1]), 2 * (uz * v[0] - ux * v[2]), 2 * (ux * v[1] - uy * v[0]) };
    out[0] = v[0] + w * t[0] + (uy * t[2] - uz * t[1]);
    out[1] = v[1] + w * t[1] + (uz * t[0] - ux * t[2]);
    out[2] = v[2] + w * t[2] + (ux * t[1] - uy * t[0]);
}

static void euler_S(const double *cam, double S[3][3])
{
    const double om = cam[1] + 0.5 * M_PI, ph = cam[0], ro = cam[2];
    const double Ry[3][3] = { { cos(ro), -sin(ro), 0 }, { sin(ro), cos(ro), 0 }, { 0, 0, 1 } };
    const double Rx[3][3] = { { 1, 0, 0 }, { 0, cos(om), -sin(om) }, { 0, sin(om), cos(om) } };
    const double Rz[3][3] = { { cos(ph), -sin(ph), 0 }, { sin(ph), cos(ph), 0 }, { 0, 0, 1 } };
    double A[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { A[i][j] = 0; for (int k = 0; k < 3; ++k) A[i][j] += Rz[i][k] * Rx[k][j]; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { S[i][j] = 0; for (int k = 0; k < 3; ++k) S[i][j] += A[i][k] * Ry[k][j]; }
}

/* toCameraSpace(v) = T^T * S * v  (OrthographicCamera.cpp:150-153) */
static void euler_to_camera_space(const double S[3][3], const double v[3], double out[3])
{
    double s[3];
    for (int i = 0; i < 3; ++i) s[i] = S[i][0] * v[0] + S[i][1] * v[1] + S[i][2] * v[2];
    /* T = [[1,0,0],[0,0,-1],[0,1,0]]; T^T s = (s0, s2, -s1) */
    out[0] = s[0]; out[1] = s[2]; out[2] = -s[1];
}

static void camera_ray(const ba_problem *p, int c, double x, double y, double origin[3], double dir[3])
{
    const double *cam = p->cam_params + 7 * c;
    const double W = p->img_width[c], H = p->img_height[c];
    if (p->model == MODEL_QUAT) {
        const double xn = -2 * ((x / W) - 0.5) + cam[4];
        const double yn = -2 * ((y / H) - 0.5) + cam[5];
        const double loc[3] = { cam[6] * xn, cam[6] * yn, -10 };
        const double z[3] = { 0, 0, 1 };
        quat_rot(cam, loc, origin);
        quat_rot(cam, z, dir);
    } else {
        double S[3][3];
        euler_S(cam, S);
        const double xn = -2 * ((x / W) - 0.5) + cam[3];
        const double yn = -2 * ((y / H) - 0.5) + cam[4];
        const double ex[3] = { 1, 0, 0 }, ey[3] = { 0, 1, 0 }, ez[3] = { 0, 0, 1 }, eo[3] = { 0, 0, -10 };
        double ax[3], ay[3], org[3];
        euler_to_camera_space(S, ex, ax);
        euler_to_camera_space(S, ey, ay);
        euler_to_camera_space(S, eo, org);
        euler_to_camera_space(S, ez, dir);
        for (int i = 0; i < 3; ++i) origin[i] = org[i] + xn * ax[i] * cam[5] + yn * ay[i] * cam[5];
    }
}

/* symmetric 3x3 eigen decomposition (cyclic Jacobi) */
static void eig3(double A[3][3], double V[3][3], double w[3])
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = i == j;
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off < 1e-300) break;
        for (int pp = 0; pp < 2; ++pp)
            for (int q = pp + 1; q < 3; ++q) {
                if (fabs(A[pp][q]) < 1e-300) continue;
                const double th = (A[q][q] - A[pp][pp]) / (2.0 * A[pp][q]);
                const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < 3; ++k) {
                    const double akp = A[k][pp], akq = A[k][q];
                    A[k][pp] = cs * akp - sn * akq; A[k][q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = A[pp][k], aqk = A[q][k];
                    A[pp][k] = cs * apk - sn * aqk; A[q][k] = sn * apk + cs * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][pp], vkq = V[k][q];
                    V[k][pp] = cs * vkp - sn * vkq; V[k][q] = sn * vkp + cs * vkq;
                }
            }
    }
    for (int i = 0; i < 3; ++i) w[i] = A[i][i];
}

ORACLE_API int
oracle_ba_triangulate(const ba_problem *p, uint8_t *valid)
{
    int *start = calloc(p->num_points + 2, sizeof(int));
    for (int k = 0; k < p->num_observations; ++k) start[p->obs_point[k] + 1]++;
    for (int j = 0; j < p->num_points; ++j) start[j + 1] += start[j];
    for (int j = 0; j < p->num_points; ++j) {
        const int k0 = start[j], k1 = start[j + 1];
        if (k1 - k0 < 2) { if (valid) valid[j] = 0; continue; }
        double R[3][3] = { { 0 } }, q[3] = { 0, 0, 0 };
        for (int k = k0; k < k1; ++k) {
            double o3[3], d[3];
            /* Feature::x/y are float (track.h:26-27): obs_xy already carries
             * the float values widened to double */
            camera_ray(p, p->obs_camera[k], p->obs_xy[2 * k], p->obs_xy[2 * k + 1], o3, d);
            const double n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            d[0] /= n; d[1] /= n; d[2] /= n;
            for (int a = 0; a < 3; ++a) {
                double row[3];
                for (int b = 0; b < 3; ++b) { row[b] = (a == b) - d[a] * d[b]; R[a][b] += row[b]; }
                q[a] += row[0] * o3[0] + row[1] * o3[1] + row[2] * o3[2];
            }
        }
        /* R.bdcSvd().solve(q): pseudo-inverse with Eigen's default rank
         * threshold (diagSize * epsilon * largest singular value) */
        double A[3][3], V[3][3], w[3];
        memcpy(A, R, sizeof A);
        eig3(A, V, w);
        double wmax = fmax(fabs(w[0]), fmax(fabs(w[1]), fabs(w[2])));
        const double thr = fmax(wmax * 3.0 * DBL_EPSILON, DBL_MIN);
        double x[3] = { 0, 0, 0 };
        for (int i = 0; i < 3; ++i) {
            if (!(fabs(w[i]) > thr)) continue;
            const double c = (V[0][i] * q[0] + V[1][i] * q[1] + V[2][i] * q[2]) / w[i];
            x[0] += c * V[0][i]; x[1] += c * V[1][i]; x[2] += c * V[2][i];
        }
        p->points[4 * j] = x[0]; p->points[4 * j + 1] = x[1]; p->points[4 * j + 2] = x[2]; p->points[4 * j + 3] = 1.0;
        if (valid) valid[j] = 1;
    }
    free(start);
    return 0;
}
