"""Checks of the BA oracle (oracle/ba_oracle.c).  PARITY UNPINNED against the
reference (Ceres/Eigen absent, no reference fixtures for this path); what is
checked instead:
  * the residual functors against an independent numpy evaluation of the
    formulas in the reference's functor headers,
  * the jet derivatives against central finite differences,
  * the two manifold Plus operations (norm preservation, Jacobian at 0),
  * LM + Schur: recovery of ground truth, and agreement of the optimum with
    scipy.optimize.least_squares on a Huber-inactive problem,
  * triangulation against ground truth.
"""
import numpy as np
import pytest

import oracle_lib
from orthosfm_amd import synth


def _numpy_residuals(sc):
    out = np.zeros((sc.obs_camera.size, 2))
    for c in range(sc.cam_params.shape[0]):
        sel = np.nonzero(sc.obs_camera == c)[0]
        if not sel.size:
            continue
        P = sc.points[sc.obs_point[sel]]
        p3 = P[:, :3] / P[:, 3:4]
        out[sel] = synth._project(sc.model, sc.cam_params[c], p3, sc.img_w[c], sc.img_h[c]) - sc.obs_xy[sel]
    return out


@pytest.mark.parametrize("model", [0, 1])
def test_residuals_match_reference_formulas(model):
    sc = synth.make_ba_scene(model, 9, 60, config_id=11)
    sc.points[:, 3] = 1.0 + 0.3 * np.sin(np.arange(60))          # non-unit homogeneous w
    sc.points[:, :3] *= sc.points[:, 3:4]
    if model == 0:
        sc.cam_params[:, :4] *= 1.0 + 0.01 * np.cos(np.arange(9))[:, None]   # non-unit quaternions
    sc.cam_params[:, 6 if model == 0 else 5] = 1.0 + 0.05 * np.arange(9)
    res, err = oracle_lib.oracle_ba_residuals(sc)
    exp = _numpy_residuals(sc)
    if model == 0:
        # the functor uses q.inverse() = conj/|q|^2 inside Eigen's unit-quaternion
        # rotation formula; project_quat does the same
        pass
    assert np.allclose(res, exp, rtol=1e-11, atol=1e-9)
    assert np.allclose(err, np.sqrt((exp ** 2).sum(1)), rtol=1e-11, atol=1e-9)


@pytest.mark.parametrize("model", [0, 1])
def test_jet_jacobian_vs_finite_differences(model):
    sc = synth.make_ba_scene(model, 7, 40, config_id=12)
    sc.points[:, 3] = 1.1
    nslots = 7 if model == 0 else 6
    for k in range(0, sc.obs_camera.size, 17):
        r, jc, jp = oracle_lib.oracle_ba_jacobian(sc, k)
        c, j = sc.obs_camera[k], sc.obs_point[k]
        for slot in range(nslots):
            h = 1e-6
            s2 = sc.copy()
            s2.cam_params[c, slot] += h
            rp = oracle_lib.oracle_ba_residuals(s2)[0][k]
            s2.cam_params[c, slot] -= 2 * h
            rm = oracle_lib.oracle_ba_residuals(s2)[0][k]
            fd = (rp - rm) / (2 * h)
            assert np.allclose(jc[:, slot], fd, rtol=1e-6, atol=1e-4), (k, slot, jc[:, slot], fd)
        for slot in range(4):
            h = 1e-6
            s2 = sc.copy()
            s2.points[j, slot] += h
            rp = oracle_lib.oracle_ba_residuals(s2)[0][k]
            s2.points[j, slot] -= 2 * h
            rm = oracle_lib.oracle_ba_residuals(s2)[0][k]
            fd = (rp - rm) / (2 * h)
            assert np.allclose(jp[:, slot], fd, rtol=1e-6, atol=1e-4), (k, slot)


def test_manifold_plus():
    import ctypes as C
    lib = oracle_lib.oracle()
    f64 = np.ctypeslib.ndpointer(np.float64)
    lib.oracle_quat_plus.argtypes = [f64, f64, f64]
    lib.oracle_homog_plus.argtypes = [f64, f64, f64]
    lib.oracle_homog_plus_jacobian.argtypes = [f64, f64]
    r = np.random.default_rng(0)
    for _ in range(20):
        x = r.standard_normal(4)
        d = 0.3 * r.standard_normal(3)
        out = np.zeros(4)
        lib.oracle_quat_plus(x, d, out)
        assert np.isclose(np.linalg.norm(out), np.linalg.norm(x), rtol=1e-13)
        lib.oracle_quat_plus(x, np.zeros(3), out)
        assert np.array_equal(out, x)
        lib.oracle_homog_plus(x, d, out)
        assert np.isclose(np.linalg.norm(out), np.linalg.norm(x), rtol=1e-13)
        lib.oracle_homog_plus(x, np.zeros(3), out)
        assert np.array_equal(out, x)
        # Jacobian of Plus at delta = 0 by finite differences
        J = np.zeros(12)
        lib.oracle_homog_plus_jacobian(x, J)
        J = J.reshape(4, 3)
        for i in range(3):
            e = np.zeros(3)
            e[i] = 1e-6
            a, b = np.zeros(4), np.zeros(4)
            lib.oracle_homog_plus(x, e, a)
            lib.oracle_homog_plus(x, -e, b)
            assert np.allclose((a - b) / 2e-6, J[:, i], atol=1e-8)


def _rot_angle_quat(qa, qb):
    qa = qa / np.linalg.norm(qa)
    qb = qb / np.linalg.norm(qb)
    return 2.0 * np.arccos(min(1.0, abs(float(np.dot(qa, qb)))))


@pytest.mark.parametrize("model", [0, 1])
def test_lm_recovers_ground_truth(model):
    sc = synth.make_ba_scene(model, 12, 400, config_id=13, noise_px=0.0)
    c0 = oracle_lib.ba_cost(sc)
    s = oracle_lib.oracle_ba_solve(sc)
    assert np.isclose(s.initial_cost, c0, rtol=1e-12)
    assert s.final_cost < 1e-6 * s.initial_cost
    assert np.isclose(oracle_lib.ba_cost(sc), s.final_cost, rtol=1e-9, atol=1e-12)
    assert s.num_iterations < 60
    # rotations come back (offsets/points only up to the translation gauge)
    for c in range(12):
        if model == 0:
            assert _rot_angle_quat(sc.cam_params[c, :4], sc.gt_cams[c, :4]) < 1e-4
        else:
            assert np.allclose(sc.cam_params[c, :3], sc.gt_cams[c, :3], atol=1e-4)
    # constant blocks untouched
    assert np.array_equal(sc.cam_params[0], sc.gt_cams[0])


@pytest.mark.parametrize("model", [0, 1])
def test_lm_optimum_vs_scipy(model):
    """Huber-inactive (sigma = 0.05 px): the oracle's optimum must be the
    least-squares optimum scipy finds from the same start."""
    scipy_opt = pytest.importorskip("scipy.optimize")
    sc = synth.make_ba_scene(model, 6, 60, config_id=14, noise_px=0.05, rot_perturb_deg=1.0)
    ref = sc.copy()
    s = oracle_lib.oracle_ba_solve(sc, function_tolerance=1e-14, max_num_iterations=200)

    C_, M_ = ref.cam_params.shape[0], ref.points.shape[0]
    # free scalars per camera; the quaternion is moved on its manifold
    # (x+ = exp(delta) * x, same as EigenQuaternionParameterization) so that
    # scipy optimises over the same set as the oracle
    ncp = 5
    nfree = (C_ - 1) * ncp

    def unpack(x):
        t = ref.copy()
        for c in range(1, C_):
            v = x[(c - 1) * ncp:(c - 1) * ncp + ncp]
            if model == 0:
                d = v[:3]
                nd = np.linalg.norm(d)
                if nd > 0:
                    dq = np.array([*(np.sin(nd) / nd * d), np.cos(nd)])
                    t.cam_params[c, :4] = synth.quat_mul(dq, ref.cam_params[c, :4])
                t.cam_params[c, 4:6] = ref.cam_params[c, 4:6] + v[3:5]
            else:
                t.cam_params[c, :5] = ref.cam_params[c, :5] + v
        t.points[:, :3] = x[nfree:].reshape(M_, 3)
        return t

    x0 = np.concatenate([np.zeros(nfree), ref.points[:, :3].reshape(-1)])

    def fun(x):
        return _numpy_residuals(unpack(x)).reshape(-1)

    sol = scipy_opt.least_squares(fun, x0, method="trf", xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=400)
    cost_scipy = 0.5 * float((sol.fun ** 2).sum())
    assert s.final_cost <= cost_scipy * (1 + 1e-6) + 1e-12
    assert np.isclose(s.final_cost, cost_scipy, rtol=1e-4)


def test_lm_with_huber_outliers_and_constant_points():
    sc = synth.make_ba_scene(0, 10, 300, config_id=15, noise_px=0.5)
    # gross outliers: Huber must keep them from dominating
    sc.obs_xy[::37] += 40.0
    s = oracle_lib.oracle_ba_solve(sc)
    assert s.final_cost < s.initial_cost
    assert s.termination in (1, 2, 3)
    res, err = oracle_lib.oracle_ba_residuals(sc)
    inl = np.ones(err.size, bool)
    inl[::37] = False
    assert np.median(err[inl]) < 1.5
    # optimize_points = 0: points must not move
    sc2 = synth.make_ba_scene(0, 8, 200, config_id=16, noise_px=0.2, point_perturb=0.0)
    p0 = sc2.points.copy()
    s2 = oracle_lib.oracle_ba_solve(sc2, optimize_points=0)
    assert np.array_equal(sc2.points, p0)
    assert s2.final_cost < s2.initial_cost


@pytest.mark.parametrize("model", [0, 1])
def test_triangulation_recovers_points(model):
    sc = synth.make_ba_scene(model, 9, 120, config_id=17, noise_px=0.0)
    sc.cam_params[:] = sc.gt_cams
    # exact (double) observations: undo the float32 rounding for this test
    sc.obs_xy[:] = _numpy_residuals_obs(sc)
    sc.points[:] = 0
    valid = oracle_lib.oracle_ba_triangulate(sc)
    assert valid.all()
    assert np.allclose(sc.points[:, :3], sc.gt_points, atol=1e-9)
    assert np.all(sc.points[:, 3] == 1.0)


def _numpy_residuals_obs(sc):
    t = sc.copy()
    t.points = np.concatenate([sc.gt_points, np.ones((sc.gt_points.shape[0], 1))], axis=1)
    t.obs_xy = np.zeros_like(sc.obs_xy)
    return _numpy_residuals(t)


@pytest.mark.parametrize("model", [0, 1])
def test_huber_active_optimum_vs_scipy(model):
    """Huber ACTIVE (planted gross outliers): Ceres' cost is 1/2 sum rho(|r_k|^2) per 2-D
    block, rho(s) = s for s <= 1, 2 sqrt(s) - 1 beyond.  Giving scipy the re-weighted
    residual r * sqrt(rho(s) / s) makes its plain least-squares objective exactly that cost
    (scipy's own loss='huber' is per SCALAR residual and cannot be used), so the oracle's
    optimum -- corrector, robustified Jacobian, LM -- is pinned independently."""
    scipy_opt = pytest.importorskip("scipy.optimize")
    sc = synth.make_ba_scene(model, 6, 70, config_id=18, noise_px=0.3, rot_perturb_deg=0.7)
    sc.obs_xy[::13] += np.array([9.0, -7.0])            # about 8 % gross outliers
    ref = sc.copy()
    s = oracle_lib.oracle_ba_solve(sc, function_tolerance=1e-14, max_num_iterations=300)
    C_, M_ = ref.cam_params.shape[0], ref.points.shape[0]
    ncp = 5
    nfree = (C_ - 1) * ncp

    def unpack(x):
        t = ref.copy()
        for c in range(1, C_):
            v = x[(c - 1) * ncp:(c - 1) * ncp + ncp]
            if model == 0:
                d = v[:3]
                nd = np.linalg.norm(d)
                if nd > 0:
                    dq = np.array([*(np.sin(nd) / nd * d), np.cos(nd)])
                    t.cam_params[c, :4] = synth.quat_mul(dq, ref.cam_params[c, :4])
                t.cam_params[c, 4:6] = ref.cam_params[c, 4:6] + v[3:5]
            else:
                t.cam_params[c, :5] = ref.cam_params[c, :5] + v
        t.points[:, :3] = x[nfree:].reshape(M_, 3)
        return t

    def fun(x):
        r = _numpy_residuals(unpack(x)).reshape(-1, 2)
        sq = (r ** 2).sum(1)
        rho = np.where(sq <= 1.0, sq, 2.0 * np.sqrt(np.maximum(sq, 1e-300)) - 1.0)
        w = np.sqrt(rho / np.maximum(sq, 1e-300))
        return (r * w[:, None]).reshape(-1)

    x0 = np.concatenate([np.zeros(nfree), ref.points[:, :3].reshape(-1)])
    assert np.isclose(0.5 * float((fun(x0) ** 2).sum()), s.initial_cost, rtol=1e-10)
    # start scipy from the oracle's optimum as well as from the common start: a local
    # minimiser must not be able to improve on the oracle's point
    sol = scipy_opt.least_squares(fun, x0, method="trf", xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=600)
    cost_scipy = 0.5 * float((sol.fun ** 2).sum())
    active = ((_numpy_residuals(sc).reshape(-1, 2) ** 2).sum(1) > 1.0).mean()
    assert active > 0.05                                   # the loss is active at the optimum
    assert np.isclose(s.final_cost, cost_scipy, rtol=2e-4), (s.final_cost, cost_scipy)
    assert s.final_cost <= cost_scipy * (1 + 2e-4)
