"""TEST INFRASTRUCTURE ONLY — numpy fp64 restatement of the relative-pose step the hot path feeds
(`estimate_pose`, /root/reference/src/utils/metrics.py:69-94, call site eval_linemod_json.py:160), used by tests/ and
bench.py's `cpu_baseline` leg as the checker for pope_amd/pose.py (pose.hip).  Never imported by the product.

PARITY UNPINNED.  The reference delegates the arithmetic to OpenCV (`cv2.findEssentialMat(..., method=cv2.RANSAC)` and
`cv2.recoverPose`), an unpinned third-party dependency (`opencv-python`, requirements.txt:21, no version) that is absent
from this image and from /root/reference, and the reference holds no fixture for it.  What follows restates
OpenCV's published algorithm for that call pair — the classic RANSAC point-set registrator around the Nister five-point
solver, calib3d/src/five-point.cpp and ptsetreg.cpp of OpenCV 4.x — from its description, not from its source:

  * metrics.py:72-78  K-normalisation of both point sets (fp64), threshold = thresh / mean(fx0, fy1, fx0, fy1);
  * findEssentialMat  N == 5: every solution of the minimal problem, all points inliers.  N > 5: RANSAC over minimal
                      5-point samples (<= max_iters = 1000), each yielding <= 10 essential matrices (Nister: 4-d null space
                      of the 5 epipolar constraints, ten cubic constraints, Gauss-Jordan, 10th-degree polynomial in z);
                      score = number of points whose Sampson error x1'Ex0^2 / (|Ex0|_xy^2 + |E'x1|_xy^2) <= thr^2; a model
                      is kept if it has MORE inliers than the best so far (and at least 5); after every improvement the
                      iteration budget becomes log(1 - conf) / log(1 - w^5), w = inlier ratio (RANSACUpdateNumIters);
  * metrics.py:86-94  for each returned E: recoverPose = decomposeEssentialMat (R1, R2, +-t) + linear triangulation of the
                      inliers, the combination with the most points in front of both cameras wins (ties: R1,t > R2,t >
                      R1,-t > R2,-t); the published inlier mask is RANSAC mask AND cheirality mask (recoverPose updates the
                      mask in place); `None` below 5 matches or when no model reaches 5 inliers.

What cannot be reproduced without cv2 is its random number stream (cv::RNG state of the registrator) and its polynomial
root finder (solvePoly); here the minimal samples come from a counter-based hash (`sample_indices`, the SAME function the
HIP kernel uses, so both evaluate the same hypotheses) and the real roots from numpy's companion-matrix eigenvalues.
Hypotheses are evaluated in rounds of ROUND = 256 (the GPU's workgroup size): the budget test runs after each round, so
at least as many hypotheses are tried as OpenCV's sequential loop would try.
"""
import math

import numpy as np

ROUND = 256          # hypotheses per round (one per thread of the HIP workgroup)
MAX_ITERS = 1000     # OpenCV's default maxIters of findEssentialMat
MASK64 = (1 << 64) - 1


# ------------------------------------------------------------------------------------------------ sampling
def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & MASK64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & MASK64
    return x ^ (x >> 31)


def sample_indices(seed, h, n):
    """Five distinct indices in [0, n) for hypothesis `h`: draw slot s takes hash(seed, h, s, attempt) % n with the first
    attempt that differs from the earlier picks (n >= 5)."""
    picks = []
    for s in range(5):
        a = 0
        while True:
            v = splitmix64((seed ^ (h * 0xD1B54A32D192ED03) ^ ((s * 64 + a) * 0x8CB92BA72F3D8DD7)) & MASK64) % n
            if v not in picks:
                picks.append(int(v))
                break
            a += 1
    return picks


# ------------------------------------------------------------------------------------------- five-point solver
# trivariate polynomials in (x, y, z) as dense arrays c[i, j, k] = coefficient of x^i y^j z^k, degree <= 3 per variable
def _pmul(a, b):
    out = np.zeros((4, 4, 4))
    for i, j, k in zip(*np.nonzero(a)):
        for p, q, r in zip(*np.nonzero(b)):
            out[i + p, j + q, k + r] += a[i, j, k] * b[p, q, r]
    return out


def _lin(cx, cy, cz, c1):
    p = np.zeros((4, 4, 4))
    p[1, 0, 0], p[0, 1, 0], p[0, 0, 1], p[0, 0, 0] = cx, cy, cz, c1
    return p


# Nister's monomial order: the ten leading monomials are eliminated, the rest are <= linear in x and y
MONOMIALS = [(3, 0, 0), (0, 3, 0), (2, 1, 0), (1, 2, 0), (2, 0, 1), (2, 0, 0), (0, 2, 1), (0, 2, 0), (1, 1, 1), (1, 1, 0),
             (1, 0, 2), (1, 0, 1), (1, 0, 0), (0, 1, 2), (0, 1, 1), (0, 1, 0), (0, 0, 3), (0, 0, 2), (0, 0, 1), (0, 0, 0)]


def five_point(x0, x1):
    """Essential matrices E (unit Frobenius norm, x1h' E x0h = 0) consistent with five correspondences x0[5,2] -> x1[5,2]
    in normalised image coordinates.  Returns an array [k, 3, 3], 0 <= k <= 10, in ascending order of the root z."""
    x0, x1 = np.asarray(x0, np.float64), np.asarray(x1, np.float64)
    q = np.stack([x1[:, 0] * x0[:, 0], x1[:, 0] * x0[:, 1], x1[:, 0], x1[:, 1] * x0[:, 0], x1[:, 1] * x0[:, 1], x1[:, 1],
                  x0[:, 0], x0[:, 1], np.ones(5)], 1)
    _, _, vt = np.linalg.svd(q)
    X, Y, Z, W = (vt[k].reshape(3, 3) for k in (5, 6, 7, 8))
    e = [[_lin(X[r, c], Y[r, c], Z[r, c], W[r, c]) for c in range(3)] for r in range(3)]
    # det(E) = 0
    det = _pmul(_pmul(e[0][0], e[1][1]) - _pmul(e[0][1], e[1][0]), e[2][2]) \
        + _pmul(_pmul(e[0][1], e[1][2]) - _pmul(e[0][2], e[1][1]), e[2][0]) \
        + _pmul(_pmul(e[0][2], e[1][0]) - _pmul(e[0][0], e[1][2]), e[2][1])
    # 2 E E' E - tr(E E') E = 0
    eet = [[sum(_pmul(e[r][k], e[c][k]) for k in range(3)) for c in range(3)] for r in range(3)]
    tr = eet[0][0] + eet[1][1] + eet[2][2]
    lam = [[eet[r][c] - (0.5 * tr if r == c else 0.0) for c in range(3)] for r in range(3)]
    rows = [det] + [sum(_pmul(lam[r][k], e[k][c]) for k in range(3)) for r in range(3) for c in range(3)]
    A = np.array([[p[m] for m in MONOMIALS] for p in rows])                       # [10, 20]
    try:
        G = np.linalg.solve(A[:, :10], A[:, 10:])                                 # rows: monomial_r + G[r] . tail = 0
    except np.linalg.LinAlgError:
        return np.zeros((0, 3, 3))
    # <k> = <e> - z <f>, <l> = <g> - z <h>, <m> = <i> - z <j>: x p3(z) + y p3(z) + p4(z) each (coefficients high -> low)
    B = np.zeros((3, 13))
    for r, (hi, lo) in enumerate(((4, 5), (6, 7), (8, 9))):
        a, b = G[hi], G[lo]
        B[r, 0:4] = [-b[0], a[0] - b[1], a[1] - b[2], a[2]]                       # x: z^3 .. 1
        B[r, 4:8] = [-b[3], a[3] - b[4], a[4] - b[5], a[5]]                       # y
        B[r, 8:13] = [-b[6], a[6] - b[7], a[7] - b[8], a[8] - b[9], a[9]]         # 1: z^4 .. 1
    px, py, p1 = (lambda r: np.poly1d(B[r, 0:4])), (lambda r: np.poly1d(B[r, 4:8])), (lambda r: np.poly1d(B[r, 8:13]))
    detB = px(0) * (py(1) * p1(2) - py(2) * p1(1)) - py(0) * (px(1) * p1(2) - px(2) * p1(1)) + p1(0) * (px(1) * py(2) - px(2) * py(1))
    c = detB.coeffs
    if len(c) < 2 or not np.all(np.isfinite(c)):
        return np.zeros((0, 3, 3))
    roots = np.roots(c)
    out = []
    for z in sorted(r.real for r in roots if abs(r.imag) <= 1e-9 * max(1.0, abs(r.real))):
        zp = np.array([z ** 4, z ** 3, z ** 2, z, 1.0])
        Bz = np.array([[B[r, 0:4] @ zp[1:], B[r, 4:8] @ zp[1:], B[r, 8:13] @ zp] for r in range(3)])
        _, _, v = np.linalg.svd(Bz)
        n = v[2]
        if abs(n[2]) < 1e-12 * np.abs(n).max():
            continue
        E = (n[0] * X + n[1] * Y + z * n[2] * Z + n[2] * W)
        out.append(E / np.linalg.norm(E))
    return np.array(out).reshape(-1, 3, 3)


# ------------------------------------------------------------------------------------------------------ scoring
def sampson_errors(E, x0, x1):
    """x1h' E x0h squared over the squared norms of the first two components of E x0h and E' x1h (fp64)."""
    h0 = np.concatenate([x0, np.ones((len(x0), 1))], 1)
    h1 = np.concatenate([x1, np.ones((len(x1), 1))], 1)
    Ex0 = h0 @ E.T
    Etx1 = h1 @ E
    num = np.sum(h1 * Ex0, 1)
    return num * num / (Ex0[:, 0] ** 2 + Ex0[:, 1] ** 2 + Etx1[:, 0] ** 2 + Etx1[:, 1] ** 2)


def update_num_iters(conf, outlier_ratio, max_iters):
    """RANSACUpdateNumIters with 5 model points."""
    p = min(max(conf, 0.0), 1.0)
    ep = min(max(outlier_ratio, 0.0), 1.0)
    num = max(1.0 - p, 2.2250738585072014e-308)
    denom = 1.0 - (1.0 - ep) ** 5
    if denom < 2.2250738585072014e-308:
        return 0
    num, denom = math.log(num), math.log(denom)
    if denom >= 0 or -num >= max_iters * (-denom):
        return max_iters
    return int(round(num / denom))     # cvRound: to nearest even on ties, like Python's round


def find_essential_ransac(x0, x1, thr, conf, seed=0, max_iters=MAX_ITERS):
    """-> (E [3,3], mask [N] bool, info) or (None, None, info).  x0, x1: normalised points [N, 2], N > 5."""
    n = len(x0)
    t2 = thr * thr
    best_count, best_E, best_id = 4, None, (-1, -1)
    niters, done, rounds = max_iters, 0, 0
    while done < niters:
        for h in range(done, min(done + ROUND, max_iters)):
            idx = sample_indices(seed, h, n)
            for k, E in enumerate(five_point(x0[idx], x1[idx])):
                count = int(np.sum(sampson_errors(E, x0, x1) <= t2))
                if count > best_count:                   # strictly more: the first of equals (lowest (h, k)) stays
                    best_count, best_E, best_id = count, E, (h, k)
        done = min(done + ROUND, max_iters)
        rounds += 1
        if best_E is not None:
            niters = min(niters, update_num_iters(conf, (n - best_count) / n, max_iters))
    info = {"hypotheses": done, "rounds": rounds, "best": best_id, "inliers": best_count if best_E is not None else 0}
    if best_E is None:
        return None, None, info
    return best_E, sampson_errors(best_E, x0, x1) <= t2, info


# ------------------------------------------------------------------------------------------------- recoverPose
def decompose_essential(E):
    """decomposeEssentialMat: R1 = U W Vt, R2 = U W' Vt, t = U[:, 2] with det(U), det(Vt) > 0."""
    U, _, Vt = np.linalg.svd(E)
    if np.linalg.det(U) < 0:
        U = -U
    if np.linalg.det(Vt) < 0:
        Vt = -Vt
    Wm = np.array([[0.0, 1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    return U @ Wm @ Vt, U @ Wm.T @ Vt, U[:, 2].copy()


def triangulate(R, t, x0, x1):
    """Linear (DLT) triangulation with P0 = [I | 0], P1 = [R | t]: the right singular vector of the 4x4 system of each
    point, as cv::triangulatePoints.  -> homogeneous points [N, 4]."""
    P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    P1 = np.hstack([R, t.reshape(3, 1)])
    out = np.zeros((len(x0), 4))
    for i in range(len(x0)):
        A = np.stack([x0[i, 0] * P0[2] - P0[0], x0[i, 1] * P0[2] - P0[1], x1[i, 0] * P1[2] - P1[0], x1[i, 1] * P1[2] - P1[1]])
        out[i] = np.linalg.svd(A)[2][3]
    return out


def cheirality_mask(R, t, x0, x1, dist=1e9):
    Q = triangulate(R, t, x0, x1)
    m = Q[:, 2] * Q[:, 3] > 0
    with np.errstate(divide="ignore", invalid="ignore"):
        X = Q[:, :3] / Q[:, 3:4]
    m &= X[:, 2] < dist
    z1 = X @ R[2] + t[2]
    m &= (z1 > 0) & (z1 < dist)
    return m


def recover_pose(E, x0, x1, mask):
    """cv2.recoverPose(E, x0, x1, I, 1e9, mask) -> (n_good, R, t, new_mask)."""
    R1, R2, t = decompose_essential(E)
    combos = [(R1, t), (R2, t), (R1, -t), (R2, -t)]
    masks = [cheirality_mask(R, tt, x0, x1) & mask for R, tt in combos]
    good = [int(m.sum()) for m in masks]
    k = 0 if (good[0] >= good[1] and good[0] >= good[2] and good[0] >= good[3]) else \
        1 if (good[1] >= good[0] and good[1] >= good[2] and good[1] >= good[3]) else \
        2 if (good[2] >= good[0] and good[2] >= good[1] and good[2] >= good[3]) else 3
    return good[k], combos[k][0], combos[k][1], masks[k]


# ---------------------------------------------------------------------------------------------- estimate_pose
def estimate_pose(kpts0, kpts1, K0, K1, thresh, conf=0.99999, seed=0, max_iters=MAX_ITERS, return_info=False):
    """metrics.py:69-94 -> (R [3,3], t [3], inliers [N] bool) or None."""
    kpts0, kpts1 = np.asarray(kpts0), np.asarray(kpts1)
    K0, K1 = np.asarray(K0, np.float64), np.asarray(K1, np.float64)
    if len(kpts0) < 5:
        return None
    x0 = (kpts0.astype(np.float64) - K0[[0, 1], [2, 2]][None]) / K0[[0, 1], [0, 1]][None]
    x1 = (kpts1.astype(np.float64) - K1[[0, 1], [2, 2]][None]) / K1[[0, 1], [0, 1]][None]
    thr = thresh / np.mean([K0[0, 0], K1[1, 1], K0[0, 0], K1[1, 1]])
    info = {}
    if len(x0) == 5:   # the minimal problem itself: every solution is returned, all points count as inliers
        Es, mask = five_point(x0, x1), np.ones(5, bool)
        if len(Es) == 0:
            return None
    else:
        E, mask, info = find_essential_ransac(x0, x1, thr, conf, seed, max_iters)
        if E is None:
            return None
        Es = E[None]
    best, ret = 0, None
    for E in Es:   # metrics.py:86-94; recoverPose narrows `mask` in place, also for the E's that follow
        n, R, t, mask = recover_pose(E, x0, x1, mask)
        if n > best:
            ret, best = (R, t, mask.copy()), n
    if return_info:
        return ret, info
    return ret


def relative_pose_error(T_0to1, R, t, ignore_gt_t_thr=0.0):
    """metrics.py:10-24 -> (t_err, R_err) in degrees."""
    t_gt = T_0to1[:3, 3]
    n = np.linalg.norm(t) * np.linalg.norm(t_gt)
    t_err = np.rad2deg(np.arccos(np.clip(np.dot(t, t_gt) / n, -1.0, 1.0)))
    t_err = np.minimum(t_err, 180 - t_err)
    if np.linalg.norm(t_gt) < ignore_gt_t_thr:
        t_err = 0
    cos = np.clip((np.trace(np.dot(R.T, T_0to1[:3, :3])) - 1) / 2, -1.0, 1.0)
    return t_err, np.rad2deg(np.abs(np.arccos(cos)))
