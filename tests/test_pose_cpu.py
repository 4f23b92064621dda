"""CPU: the relative-pose step.  (1) oracle/pose_ref.py — the numpy fp64 restatement of what
`estimate_pose` (src/utils/metrics.py:69-94) asks of cv2.findEssentialMat / cv2.recoverPose — by property, since neither
cv2 nor a reference fixture exists for this step ("parity unpinned"): planted poses come back, solutions satisfy the
essential-matrix constraints, the budget rule gives OpenCV's documented values.  (2) The arithmetic pose.hip runs on the
GPU (pope_amd/csrc/pose_math.h) compiled for the HOST and held against the oracle: five-point candidates, Sturm roots,
decomposition, cheirality, minimal-sample selection."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import pose_ref as P
from pope_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dp = C.POINTER(C.c_double)


def normalised(k0, k1, K0, K1):
    x0 = (k0.astype(np.float64) - K0[[0, 1], [2, 2]]) / K0[[0, 1], [0, 1]]
    x1 = (k1.astype(np.float64) - K1[[0, 1], [2, 2]]) / K1[[0, 1], [0, 1]]
    return x0, x1


def true_E(R, t):
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    E = tx @ R
    return E / np.linalg.norm(E)


@pytest.fixture(scope="module")
def host():
    """pose_math.h compiled as host code behind a C interface (tests/native/pose_host_check.cpp)."""
    out = os.path.join(ROOT, "tests", "native", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libpose_host.so")
    src = os.path.join(ROOT, "tests", "native", "pose_host_check.cpp")
    hdr = os.path.join(ROOT, "pope_amd", "csrc", "pose_math.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-o", so, src], check=True)
    lib = C.CDLL(so)
    lib.host_five_point.argtypes = [dp, dp, dp]
    lib.host_sample_indices.argtypes = [C.c_ulonglong, C.c_uint, C.c_uint, C.POINTER(C.c_int)]
    lib.host_sturm_roots.argtypes = [dp, C.c_int, C.c_double, C.c_double, dp]
    lib.host_decompose.argtypes = [dp] * 4
    lib.host_cheirality.argtypes = [dp, dp] + [C.c_double] * 5
    lib.host_sampson.argtypes = [dp] + [C.c_double] * 4
    lib.host_sampson.restype = C.c_double
    lib.host_update_num_iters.argtypes = [C.c_double, C.c_double, C.c_int]
    return lib


def ptr(a):
    return a.ctypes.data_as(dp)


def host_five_point(lib, x0, x1):
    x0, x1, E = np.ascontiguousarray(x0, np.float64), np.ascontiguousarray(x1, np.float64), np.zeros((10, 9))
    n = lib.host_five_point(ptr(x0), ptr(x1), ptr(E))
    return E[:n].reshape(-1, 3, 3)


# ---------------------------------------------------------------------------------------------------- the oracle
def test_oracle_five_point_contains_the_planted_essential_matrix():
    for seed in range(8):
        k0, k1, K0, K1, R, t, _ = synth.synthetic_pose_scene(5, seed, outlier=0.0)
        x0, x1 = normalised(k0, k1, K0, K1)
        Es = P.five_point(x0, x1)
        assert 1 <= len(Es) <= 10
        Et = true_E(R, t)
        assert min(min(np.abs(E - Et).max(), np.abs(E + Et).max()) for E in Es) < 1e-4   # inputs are float32 pixels
        for E in Es:   # every solution is an essential matrix through the five correspondences
            h0, h1 = np.c_[x0, np.ones(5)], np.c_[x1, np.ones(5)]
            assert np.abs(np.sum(h1 * (h0 @ E.T), 1)).max() < 1e-12
            assert abs(np.linalg.det(E)) < 1e-6 and np.abs(2 * E @ E.T @ E - np.trace(E @ E.T) * E).max() < 1e-6


def test_oracle_budget_rule():
    # log(1 - conf) / log(1 - w^5), rounded; capped by max_iters; 0 when every point is an inlier
    assert [P.update_num_iters(0.99, r, 1000) for r in (0.1, 0.3, 0.5, 0.7)] == [5, 25, 145, 1000]
    assert P.update_num_iters(0.99, 0.0, 1000) == 0 and P.update_num_iters(0.99, 1.0, 1000) == 1000
    assert P.update_num_iters(0.99999, 0.5, 1000) == 363


def test_oracle_samples_are_distinct_and_reproducible():
    for n in (5, 6, 7, 100):
        for h in range(50):
            idx = P.sample_indices(3, h, n)
            assert len(set(idx)) == 5 and all(0 <= i < n for i in idx) and idx == P.sample_indices(3, h, n)
    assert P.sample_indices(0, 0, 100) != P.sample_indices(1, 0, 100)


@pytest.mark.parametrize("seed,n", [(0, 120), (1, 60)])
def test_oracle_recovers_a_planted_pose_with_30_percent_outliers(seed, n):
    # threshold 0.05 px on noise-free points: at the drivers' 0.5 px a model that bends to catch one or two clutter points
    # inside its tube out-counts the exact one (86 vs 84 inliers at seed 0, t off by 2.5 deg) — RANSAC's criterion, in cv2 too
    k0, k1, K0, K1, R, t, planted = synth.synthetic_pose_scene(n, seed, outlier=0.3, noise=0.0)
    ret, info = P.estimate_pose(k0, k1, K0, K1, 0.05, 0.99, return_info=True)
    assert ret is not None and info["rounds"] >= 1
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    t_err, R_err = P.relative_pose_error(T, ret[0], ret[1])
    assert R_err < 0.5 and t_err < 1.0, (R_err, t_err)
    assert np.all(ret[2][planted]) and ret[2].dtype == bool          # inlier mask contains every planted inlier
    assert abs(np.linalg.det(ret[0]) - 1) < 1e-9 and abs(np.linalg.norm(ret[1]) - 1) < 1e-9


def test_oracle_none_cases():
    k0, k1, K0, K1, *_ = synth.synthetic_pose_scene(4, 0, outlier=0.0)
    assert P.estimate_pose(k0, k1, K0, K1, 0.5, 0.99) is None                         # metrics.py:70-71
    g = np.random.default_rng(0)          # clutter only: a minimal sample always fits its own five points, nothing more
    a, b = g.uniform(0, 256, (12, 2)).astype(np.float32), g.uniform(0, 256, (12, 2)).astype(np.float32)
    ret, info = P.estimate_pose(a, b, K0, K1, 1e-4, 0.99, max_iters=256, return_info=True)
    assert info["inliers"] == 5 and info["hypotheses"] == 256 and (ret is None or 1 <= ret[2].sum() <= 5)


def test_oracle_minimal_problem_goes_through_every_solution():
    k0, k1, K0, K1, R, t, _ = synth.synthetic_pose_scene(5, 3, outlier=0.0)
    ret = P.estimate_pose(k0, k1, K0, K1, 0.5, 0.99)
    assert ret is not None and ret[2].shape == (5,) and ret[2].sum() >= 1


# --------------------------------------------------------------------- the device arithmetic, compiled for the host
def test_host_five_point_matches_oracle(host):
    worst, total = 0.0, 0
    for seed in range(120):
        k0, k1, K0, K1, *_ = synth.synthetic_pose_scene(5, seed, outlier=0.0, noise=0.5)
        x0, x1 = normalised(k0, k1, K0, K1)
        a, b = P.five_point(x0, x1), host_five_point(host, x0, x1)
        assert len(a) == len(b), seed
        total += len(a)
        for Ea, Eb in zip(a, b):
            worst = max(worst, min(np.abs(Ea - Eb).max(), np.abs(Ea + Eb).max()))
    print(f"{total} candidates, worst |E_host - E_oracle| = {worst:.2e}")
    assert total > 300 and worst < 5e-4      # ill-conditioned samples: both solvers leave residuals of this size


def test_host_sturm_roots_match_planted_roots(host):
    g = np.random.default_rng(1)
    out = np.zeros(10)
    for trial in range(200):
        deg = int(g.integers(2, 11))
        roots = g.uniform(-3, 3, int(g.integers(0, deg + 1)))
        c = np.poly(roots) if len(roots) else np.ones(1)
        while len(c) - 1 + 2 <= deg:                                 # pad with complex-conjugate pairs
            a, b = g.uniform(-2, 2), g.uniform(0.3, 2)
            c = np.polymul(c, [1.0, -2 * a, a * a + b * b])
        lo, hi = -1.0, 1.0
        low_first = np.ascontiguousarray(c[::-1], np.float64)
        n = host.host_sturm_roots(ptr(low_first), len(c) - 1, lo, hi, ptr(out))
        want = np.sort(roots[(roots > lo) & (roots <= hi)])
        if len(want) > 1 and np.diff(want).min() < 1e-4:             # a near-double root: conditioning, not the method
            continue
        assert n == len(want), (trial, n, want)
        if n:
            np.testing.assert_allclose(out[:n], want, atol=2e-5)   # np.poly's coefficients carry the conditioning
    # a double root is reported once
    c = np.poly([0.25, 0.25, -0.5])[::-1].copy()
    assert host.host_sturm_roots(ptr(c), 3, -1.0, 1.0, ptr(out)) == 2
    np.testing.assert_allclose(out[:2], [-0.5, 0.25], atol=1e-6)
    # the device always passes degree 10: leading coefficients that vanish (true degree 7 here) are stripped
    c = np.zeros(11)
    c[:8] = np.poly([0.5, -0.25, 0.75, 2.0, -3.0, 1.5, -1.5])[::-1]
    assert host.host_sturm_roots(ptr(c), 10, -1.0, 1.0, ptr(out)) == 3
    np.testing.assert_allclose(out[:3], [-0.25, 0.5, 0.75], atol=1e-9)
    # an even polynomial: every other remainder of the chain loses two degrees at once (division steps of length 3)
    c = np.polymul(np.polymul([1, 0, -0.25], [1, 0, -0.04]), np.polymul([1, 0, 1.0], [1, 0, 0, 0, 1.0]))[::-1].copy()
    assert host.host_sturm_roots(ptr(c), len(c) - 1, -1.0, 1.0, ptr(out)) == 4
    np.testing.assert_allclose(out[:4], [-0.5, -0.2, 0.2, 0.5], atol=1e-9)
    # the interval is (lo, hi]: a root at hi counts, a root at lo does not (coefficients chosen so that Horner's rule gives
    # exact zeros at both ends; with rounding noise there the end roots fall on whichever side the noise puts them)
    c = np.poly([1.0, -1.0, 0.5])[::-1].copy()
    assert host.host_sturm_roots(ptr(c), 3, -1.0, 1.0, ptr(out)) == 2
    np.testing.assert_allclose(out[:2], [0.5, 1.0], atol=1e-9)
    # a constant and the zero polynomial have no roots
    c = np.zeros(11); c[0] = 3.0
    assert host.host_sturm_roots(ptr(c), 10, -1.0, 1.0, ptr(out)) == 0
    assert host.host_sturm_roots(ptr(np.zeros(11)), 10, -1.0, 1.0, ptr(out)) == 0


def test_host_samples_identical_to_oracle(host):
    picks = (C.c_int * 5)()
    for seed in (0, 7, 2 ** 40 + 3):
        for h in (0, 1, 255, 999):
            for n in (5, 6, 33, 4800):
                host.host_sample_indices(seed, h, n, picks)
                assert list(picks) == P.sample_indices(seed, h, n)


def test_host_decomposition_cheirality_and_budget(host):
    disagreements, disagreements_inliers, n_inl = 0, 0, 0
    for seed in range(20):
        k0, k1, K0, K1, R, t, _ = synth.synthetic_pose_scene(40, seed, outlier=0.2, noise=0.3)
        x0, x1 = normalised(k0, k1, K0, K1)
        E = true_E(R, t) + np.random.default_rng(seed).normal(size=(3, 3)) * 1e-3
        R1, R2, tt = P.decompose_essential(E)
        r1, r2, t3 = np.zeros(9), np.zeros(9), np.zeros(3)
        host.host_decompose(ptr(np.ascontiguousarray(E.ravel())), ptr(r1), ptr(r2), ptr(t3))
        r1, r2 = r1.reshape(3, 3), r2.reshape(3, 3)
        assert min(max(np.abs(r1 - R1).max(), np.abs(r2 - R2).max()), max(np.abs(r1 - R2).max(), np.abs(r2 - R1).max())) < 1e-9
        assert min(np.abs(t3 - tt).max(), np.abs(t3 + tt).max()) < 1e-9
        assert abs(np.linalg.det(r1) - 1) < 1e-9 and abs(np.linalg.det(r2) - 1) < 1e-9
        for Rm, tv in ((R1, tt), (R2, -tt)):
            want = P.cheirality_mask(Rm, tv, x0, x1)
            got = np.array([host.host_cheirality(ptr(np.ascontiguousarray(Rm.ravel())), ptr(np.ascontiguousarray(tv)), *x0[i], *x1[i], 1e9)
                            for i in range(len(x0))], bool)
            disagreements += int((want != got).sum())
            near = P.sampson_errors(true_E(R, t), x0, x1) < (2.0 / 600) ** 2      # what a RANSAC mask would keep
            disagreements_inliers += int((want != got)[near].sum())
            n_inl += int(near.sum())
        e = P.sampson_errors(E, x0, x1)
        got = np.array([host.host_sampson(ptr(np.ascontiguousarray(E.ravel())), *x0[i], *x1[i]) for i in range(len(x0))])
        np.testing.assert_allclose(got, e, rtol=1e-7, atol=1e-18)   # the numerator cancels for near-exact points
    # the triangulation is a 4x4 smallest-singular-vector problem: well separated for inliers (the only points recoverPose
    # counts), ill-conditioned for clutter, where the inverse iteration and the SVD may settle on different vectors
    assert disagreements_inliers == 0 and n_inl > 1000 and disagreements <= 5, (disagreements, disagreements_inliers)
    for c, r in [(0.99, 0.3), (0.99999, 0.5), (0.99, 0.0), (0.99, 0.95), (0.99, 1.0), (0.5, 0.2)]:
        assert host.host_update_num_iters(c, r, 1000) == P.update_num_iters(c, r, 1000)
