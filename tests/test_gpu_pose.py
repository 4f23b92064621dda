"""GPU: the batched relative-pose solver (pose.hip behind pope_amd/pose.py) against oracle/pose_ref.py —
the numpy fp64 restatement of the same algorithm, fed the same minimal samples — and by property.  Parity with OpenCV's
own findEssentialMat / recoverPose is unpinned (cv2 is absent, the reference holds no fixture): see oracle/pose_ref.py."""
import numpy as np
import pytest
import torch

from oracle import pose_ref as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(hip_lib):
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def pack(scenes, dev):
    """Scenes (k0, k1, K0, K1, ...) -> the matcher's compacted layout: concatenated matches + per-pair counts."""
    k0 = torch.from_numpy(np.concatenate([s[0] for s in scenes])).to(dev)
    k1 = torch.from_numpy(np.concatenate([s[1] for s in scenes])).to(dev)
    counts = torch.tensor([len(s[0]) for s in scenes], dtype=torch.int32)
    K0 = np.stack([s[2] for s in scenes])
    K1 = np.stack([s[3] for s in scenes])
    return k0, k1, counts, K0, K1


def pose_error(R, t, Rg, tg):
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = Rg, tg
    return P.relative_pose_error(T, R, t)


def test_five_point_kernel_matches_oracle(dev):
    from pope_amd import pose, synth
    x0s, x1s = [], []
    for seed in range(96):
        k0, k1, K0, K1, *_ = synth.synthetic_pose_scene(5, seed, outlier=0.0, noise=0.5)
        x0s.append((k0.astype(np.float64) - K0[[0, 1], [2, 2]]) / K0[[0, 1], [0, 1]])
        x1s.append((k1.astype(np.float64) - K1[[0, 1], [2, 2]]) / K1[[0, 1], [0, 1]])
    E, n = pose.five_point(torch.from_numpy(np.stack(x0s)).to(dev), torch.from_numpy(np.stack(x1s)).to(dev))
    E, n = E.cpu().numpy(), n.cpu().numpy()
    worst, total = 0.0, 0
    for s in range(96):
        want = P.five_point(x0s[s], x1s[s])
        assert n[s] == len(want), s
        total += len(want)
        for Ea, Eb in zip(want, E[s, :n[s]]):
            worst = max(worst, min(np.abs(Ea - Eb).max(), np.abs(Ea + Eb).max()))
        assert np.all(E[s, n[s]:] == 0)
    print(f"{total} essential matrices from 96 minimal problems, worst |E_gpu - E_oracle| = {worst:.2e}")
    assert total > 250 and worst < 5e-4


def test_batch_against_oracle_same_samples(dev):
    """Same samples, same algorithm: the winning hypothesis, the inlier set and the pose agree with the oracle."""
    from pope_amd import pose, synth
    scenes = [synth.synthetic_pose_scene(n, seed, outlier=o, noise=z) for seed, (n, o, z) in
              enumerate([(60, 0.3, 0.0), (90, 0.2, 0.2), (40, 0.5, 0.1), (7, 0.0, 0.0), (130, 0.3, 0.3)])]
    k0, k1, counts, K0, K1 = pack(scenes, dev)
    out = pose.estimate_pose_batch(k0, k1, counts, K0, K1, 0.5, 0.99, seed=11)
    info, R, t, inl = out["info"].cpu().numpy(), out["R"].cpu().numpy(), out["t"].cpu().numpy(), out["inliers"].cpu().numpy()
    off = 0
    for b, s in enumerate(scenes):
        n = len(s[0])
        ret, oi = P.estimate_pose(s[0], s[1], s[2], s[3], 0.5, 0.99, seed=11, return_info=True)
        assert ret is not None and info[b, 0] > 0 and info[b, 6] == n and info[b, 7] == 0
        print(f"pair {b}: N={n} oracle best {oi['best']} inliers {oi['inliers']} after {oi['hypotheses']} hypotheses | gpu best "
              f"({info[b, 4]}, {info[b, 5]}) inliers {info[b, 1]} after {info[b, 2]}")
        assert (info[b, 4], info[b, 5]) == oi["best"] and info[b, 1] == oi["inliers"]
        assert info[b, 2] == oi["hypotheses"] and info[b, 3] == oi["rounds"]
        assert np.array_equal(inl[off:off + n], ret[2]) and info[b, 0] == int(ret[2].sum())
        # the winner is the same ROOT of the same tenth-degree polynomial on both sides, found by different methods (Sturm
        # isolation + Newton here, companion-matrix eigenvalues in the checker): the pose inherits eps x the root's
        # condition number (up to ~1e9 for these samples), hence 1e-6 and not 1e-12
        np.testing.assert_allclose(R[b], ret[0], atol=1e-6)
        np.testing.assert_allclose(t[b], ret[1], atol=1e-6)
        off += n


@pytest.mark.parametrize("n,outlier", [(80, 0.3), (400, 0.3), (1200, 0.3), (200, 0.6)])
def test_planted_pose_is_recovered(dev, n, outlier):
    """30 % (and 60 %) clutter, noise-free inliers, tight threshold: R within 0.5 deg, t within 1 deg, the mask holds every
    planted inlier and no clutter point that is off the epipolar geometry."""
    from pope_amd import pose, synth
    scenes = [synth.synthetic_pose_scene(n, 100 + seed, outlier=outlier, noise=0.0) for seed in range(6)]
    k0, k1, counts, K0, K1 = pack(scenes, dev)
    out = pose.estimate_pose_batch(k0, k1, counts, K0, K1, 0.05, 0.99)
    R, t, inl, info = out["R"].cpu().numpy(), out["t"].cpu().numpy(), out["inliers"].cpu().numpy(), out["info"].cpu().numpy()
    off = 0
    for b, s in enumerate(scenes):
        t_err, R_err = pose_error(R[b], t[b], s[4], s[5])
        got = inl[off:off + n]
        print(f"n={n} outliers {outlier:.0%} pair {b}: R err {R_err:.4f} deg, t err {t_err:.4f} deg, inliers {got.sum()} "
              f"(planted {s[6].sum()}), {info[b, 2]} hypotheses")
        assert R_err < 0.5 and t_err < 1.0
        assert np.all(got[s[6]]) and got.sum() <= s[6].sum() + 2
        assert abs(np.linalg.det(R[b]) - 1) < 1e-9 and abs(np.linalg.norm(t[b]) - 1) < 1e-9
        off += n


def test_noisy_matches_at_the_drivers_threshold(dev):
    """0.3 px noise, 0.5 px threshold, conf 0.99 (eval_linemod_json.py:160).  Round 3 widened this test's bound from
    (2 deg, 10 deg) to (5 deg, 20 deg) after scene 1 came out at 2.29 deg / 8.38 deg.  Scene-dependent errors of a few
    degrees are the ALGORITHM's: a five-point RANSAC returns the best MINIMAL-sample model, unrefined (as cv2.findEssentialMat
    does, metrics.py:80-94), so its accuracy is that of five noisy matches.  What the kernel owes is the algorithm's answer:
    on every scene the GPU must return the checker's pose (oracle/pose_ref.py on the same samples: same winner, same inlier
    mask, R / t to 1e-6), the checker's own error against the planted pose is printed beside it, and the (2 deg, 10 deg) bound
    is held on the MEDIAN over the eight scenes with a loose sanity bound on each."""
    from pope_amd import pose, synth
    scenes = [synth.synthetic_pose_scene(300, 40 + seed, outlier=0.3, noise=0.3) for seed in range(8)]
    k0, k1, counts, K0, K1 = pack(scenes, dev)
    out = pose.estimate_pose_batch(k0, k1, counts, K0, K1, 0.5, 0.99)
    R, t, inl, info = (out[k].cpu().numpy() for k in ("R", "t", "inliers", "info"))
    errs = []
    for b, s in enumerate(scenes):
        ret, oi = P.estimate_pose(s[0], s[1], s[2], s[3], 0.5, 0.99, return_info=True)
        got = inl[300 * b:300 * (b + 1)]
        assert ret is not None and (info[b, 4], info[b, 5]) == oi["best"] and info[b, 1] == oi["inliers"], b
        assert np.array_equal(got, ret[2]), b
        # same root of the same polynomial found by Sturm + Newton here and by companion-matrix eigenvalues in the checker:
        # the pose inherits eps x the root's condition number, which noisy minimal samples push to ~1e10 (measured 1.2e-6)
        np.testing.assert_allclose(R[b], ret[0], atol=1e-5)
        np.testing.assert_allclose(t[b], ret[1], atol=1e-5)
        t_err, R_err = pose_error(R[b], t[b], s[4], s[5])
        t_ora, R_ora = pose_error(ret[0], ret[1], s[4], s[5])
        print(f"scene {b}: gpu R {R_err:.3f} deg t {t_err:.3f} deg | checker R {R_ora:.3f} deg t {t_ora:.3f} deg | "
              f"{info[b, 1]} RANSAC inliers of {int(s[6].sum())} planted, {info[b, 2]} hypotheses")
        assert abs(R_err - R_ora) < 1e-4 and abs(t_err - t_ora) < 1e-3
        errs.append((R_err, t_err))
        assert R_err < 6.0 and t_err < 25.0, (b, R_err, t_err)        # sanity: a usable pose on every scene
        assert (got & s[6]).sum() >= 0.6 * s[6].sum() and (got & ~s[6]).sum() <= 3
    R_med, t_med = np.median([e[0] for e in errs]), np.median([e[1] for e in errs])
    print(f"median over the scenes: R {R_med:.3f} deg, t {t_med:.3f} deg")
    assert R_med < 2.0 and t_med < 10.0


def test_corrupt_counts_refuse_the_pairs_behind_them(dev):
    """a negative count must not pull the offsets of the pairs behind it back under earlier rows; rows of
    `inliers` past sum(counts) read as zero (the kernel never writes them)."""
    from pope_amd import pose, synth
    scenes = [synth.synthetic_pose_scene(40, 70 + k, outlier=0.2, noise=0.1) for k in range(4)]
    k0, k1, counts, K0, K1 = pack(scenes, dev)
    good = pose.estimate_pose_batch(k0, k1, counts, K0, K1, 0.5, 0.99)
    bad_counts = counts.clone()
    bad_counts[1] = -40
    out = pose.estimate_pose_batch(k0, k1, bad_counts, K0, K1, 0.5, 0.99)
    info = out["info"].cpu().numpy()
    assert info[0, 7] == 0 and torch.equal(out["R"][0], good["R"][0])       # the pair in front is untouched
    assert list(info[1:, 7]) == [-1, -1, -1] and list(info[1:, 0]) == [0, 0, 0]
    assert int(out["inliers"][40:].sum()) == 0
    # capacity larger than the matches: the tail of the mask is zeros, the head is the unpadded call's mask
    pad = torch.zeros(25, 2, device=dev)
    out = pose.estimate_pose_batch(torch.cat([k0, pad]), torch.cat([k1, pad]), counts, K0, K1, 0.5, 0.99)
    assert out["inliers"].numel() == 185 and int(out["inliers"][160:].sum()) == 0
    assert torch.equal(out["inliers"][:160], good["inliers"]) and torch.equal(out["R"], good["R"])


def test_deterministic_and_batch_invariant(dev):
    from pope_amd import pose, synth
    scenes = [synth.synthetic_pose_scene(50 + 17 * k, k, outlier=0.3, noise=0.2) for k in range(7)]
    k0, k1, counts, K0, K1 = pack(scenes, dev)
    a = pose.estimate_pose_batch(k0, k1, counts, K0, K1, 0.5, 0.99)
    b = pose.estimate_pose_batch(k0, k1, counts, K0, K1, 0.5, 0.99)
    for key in ("R", "t", "E", "inliers", "info"):
        assert torch.equal(a[key], b[key]), key
    off = np.concatenate([[0], np.cumsum(counts.numpy())])
    for k in (0, 3, 6):      # pair k alone == pair k inside the batch, bit for bit
        one = pose.estimate_pose_batch(k0[off[k]:off[k + 1]], k1[off[k]:off[k + 1]], counts[k:k + 1], K0[k], K1[k], 0.5, 0.99)
        assert torch.equal(one["R"][0], a["R"][k]) and torch.equal(one["t"][0], a["t"][k])
        assert torch.equal(one["inliers"], a["inliers"][off[k]:off[k + 1]])
        assert torch.equal(one["info"][0], a["info"][k])
    c = pose.estimate_pose_batch(k0, k1, counts, K0, K1, 0.5, 0.99, seed=5)      # another seed: other samples
    assert not torch.equal(c["info"][:, 4], a["info"][:, 4])


def test_ragged_batches_none_cases_and_minimal_problem(dev):
    """Pairs with 0, 3 and 4 matches give `None` (metrics.py:70-71), exactly five matches are the minimal problem (every
    solution through recoverPose), and the pairs after them still read their own rows."""
    from pope_amd import pose, synth
    full = synth.synthetic_pose_scene(64, 9, outlier=0.25, noise=0.0)
    five = synth.synthetic_pose_scene(5, 3, outlier=0.0)
    sizes = [0, 3, 64, 4, 5, 64]
    scenes = [tuple(a[:n] if isinstance(a, np.ndarray) and a.ndim and len(a) == 64 else a for a in full) for n in sizes]
    scenes[4] = five
    k0, k1, counts, K0, K1 = pack(scenes, dev)
    out = pose.estimate_pose_batch(k0, k1, counts, K0, K1, 0.05, 0.99)
    n_inl, info = out["n_inliers"].cpu().numpy(), out["info"].cpu().numpy()
    assert list(n_inl[[0, 1, 3]]) == [0, 0, 0] and n_inl[2] > 0 and n_inl[5] > 0
    assert list(info[:, 6]) == sizes
    assert torch.equal(out["R"][2], out["R"][5]) and torch.equal(out["t"][2], out["t"][5])     # same rows, same samples
    inl = out["inliers"].cpu().numpy()
    assert not inl[:3].any() and not inl[67:71].any() and np.array_equal(inl[3:67], inl[76:140])
    ret = P.estimate_pose(five[0], five[1], five[2], five[3], 0.05, 0.99)
    assert (ret is None) == (n_inl[4] == 0)
    if ret is not None:
        np.testing.assert_allclose(out["R"][4].cpu().numpy(), ret[0], atol=1e-6)
        np.testing.assert_allclose(out["t"][4].cpu().numpy(), ret[1], atol=1e-6)
        assert np.array_equal(inl[71:76], ret[2])
    # the drop-in single-pair form (reference signature and return value)
    assert pose.estimate_pose(full[0][:4], full[1][:4], full[2], full[3], 0.5, 0.99) is None
    R, t, mask = pose.estimate_pose(full[0], full[1], full[2], full[3], 0.05, 0.99)
    assert R.shape == (3, 3) and t.shape == (3,) and mask.shape == (64,) and mask.dtype == bool
    t_err, R_err = pose_error(R, t, full[4], full[5])
    assert R_err < 0.5 and t_err < 1.0 and np.all(mask[full[6]])
    # counts that exceed the rows handed over are refused per pair, not read out of bounds
    bad = pose.estimate_pose_batch(k0[:10], k1[:10], torch.tensor([6, 8], dtype=torch.int32), K0[0], K1[0], 0.5, 0.99)
    assert list(bad["info"][:, 7].cpu().numpy()) == [0, -1] and int(bad["n_inliers"][1]) == 0


def test_straight_from_the_dense_matcher(dev, sd0):
    """The pipeline's hand-over: `dense_match` output buffers go into the solver as they are (no host copy of matches)."""
    from pope_amd import pose, synth
    from pope_amd.dinov2_utils import load_dinov2_model
    from pope_amd.pipeline import PairPipeline
    model = load_dinov2_model(state_dict=sd0).to(dev)
    i0, i1 = synth.synthetic_pairs(3, 224, 308, seed=2)
    out = PairPipeline(model, chunk=4)(i0.to(dev), i1.to(dev))
    assert int(out["counts"].min()) >= 5
    K = np.array([[572.4, 0, 154.0], [0, 573.6, 112.0], [0, 0, 1.0]])
    res = pose.estimate_pose_batch(out["mkpts0_c"], out["mkpts1_c"], out["counts"], K, K, 0.5, 0.99)
    info = res["info"].cpu().numpy()
    assert list(info[:, 6]) == list(out["counts"].numpy()) and np.all(info[:, 7] == 0)
    # the synthetic second image is the first one rolled by (14, 28) px: every match is consistent with one epipolar geometry
    assert np.all(info[:, 1] >= 0.9 * info[:, 6])
    for b in range(3):
        sel = (out["b_ids"] == b).cpu().numpy()
        ret = P.estimate_pose(out["mkpts0_c"].cpu().numpy()[sel], out["mkpts1_c"].cpu().numpy()[sel], K, K, 0.5, 0.99)
        assert (ret is None) == (info[b, 0] == 0)
        if ret is not None:
            assert np.array_equal(res["inliers"].cpu().numpy()[sel], ret[2])


def test_degenerate_inputs_terminate_with_none(dev):
    """Identical points, NaN / inf coordinates, a singular K: every loop of the solver is bounded, the pair reports `None`
    (or a pose with a handful of inliers) and the pairs next to it in the batch are untouched."""
    from pope_amd import pose, synth
    good = synth.synthetic_pose_scene(80, 1, outlier=0.2, noise=0.0)
    same = (np.full((50, 2), 100.0, np.float32), np.full((50, 2), 60.0, np.float32), good[2], good[3])
    nan = (np.full((40, 2), np.nan, np.float32), np.full((40, 2), np.nan, np.float32), good[2], good[3])
    g = np.random.default_rng(0)
    huge = ((g.uniform(-1, 1, (30, 2)) * 3.0e38).astype(np.float32), (g.uniform(-1, 1, (30, 2)) * 3.0e38).astype(np.float32), good[2], good[3])
    k0 = torch.from_numpy(np.concatenate([good[0], same[0], nan[0], huge[0], good[0]])).to(dev)
    k1 = torch.from_numpy(np.concatenate([good[1], same[1], nan[1], huge[1], good[1]])).to(dev)
    counts = torch.tensor([80, 50, 40, 30, 80], dtype=torch.int32)
    out = pose.estimate_pose_batch(k0, k1, counts, good[2], good[3], 0.05, 0.99)
    torch.cuda.synchronize()
    info = out["info"].cpu().numpy()
    assert info[0, 0] > 0 and np.array_equal(info[0], info[4])                # the good pair, twice: identical, unaffected
    assert torch.equal(out["R"][0], out["R"][4]) and bool(torch.isfinite(out["R"][0]).all())
    assert info[2, 0] == 0 and info[2, 1] == 0                                # NaN points: no model, None
    assert info[1, 0] <= 50 and info[3, 0] <= 30 and np.all(info[:, 7] == 0)  # degenerate geometry: whatever it is, it ended
    # a singular camera matrix (fx = 0): coordinates become inf / NaN -> None, not a hang
    Kbad = good[2].copy()
    Kbad[0, 0] = 0.0
    res = pose.estimate_pose_batch(k0[:80], k1[:80], counts[:1], Kbad, good[3], 0.5, 0.99)
    assert int(res["n_inliers"][0]) == 0


def _two_view(X, R, t, K0, K1):
    p0 = (X / X[:, 2:]) @ K0.T
    X1 = X @ R.T + t
    p1 = (X1 / X1[:, 2:]) @ K1.T
    return p0[:, :2].astype(np.float32), p1[:, :2].astype(np.float32)


def test_planar_scene_and_pure_rotation(dev):
    """Two geometries where an eight-point solver breaks and the five-point one must not: all points on ONE plane (every
    match fits; a plane admits a second essential matrix that also passes the cheirality vote, so the test holds the answer
    to the oracle's — the first model with the most inliers, as OpenCV keeps it — not to the planted pose), and a camera that
    only rotates (t is not observable, every [t]x R fits: a finite answer, every match an inlier, the planted rotation)."""
    from pope_amd import pose, synth
    base = synth.synthetic_pose_scene(8, 5, outlier=0.0)
    K0, K1, R, t = base[2], base[3], base[4], base[5]
    g = np.random.default_rng(12)
    n = 120
    uv = g.uniform(-1, 1, (n, 2))
    plane = np.stack([uv[:, 0], uv[:, 1], 4.0 + 0.3 * uv[:, 0] - 0.2 * uv[:, 1]], 1)       # a tilted plane
    cloud = np.stack([g.uniform(-1, 1, n), g.uniform(-1, 1, n), g.uniform(3, 6, n)], 1)
    scenes = [_two_view(plane, R, t, K0, K1) + (K0, K1), _two_view(cloud, R, np.zeros(3), K0, K1) + (K0, K1)]
    k0, k1, counts, K0d, K1d = pack(scenes, dev)
    out = pose.estimate_pose_batch(k0, k1, counts, K0d, K1d, 0.05, 0.99, seed=3)
    info, Rg, tg, inl = out["info"].cpu().numpy(), out["R"].cpu().numpy(), out["t"].cpu().numpy(), out["inliers"].cpu().numpy()
    assert np.isfinite(Rg).all() and np.isfinite(tg).all() and info[:, 7].tolist() == [0, 0]
    assert info[0, 1] == n and info[0, 0] >= n - 2            # planar: every match fits the model, and lies in front
    # pure rotation: every match fits, the rotation is still the planted one
    cos = np.clip((np.trace(Rg[1].T @ R) - 1) / 2, -1, 1)
    assert info[1, 1] == n and np.rad2deg(np.arccos(cos)) < 0.5, (info[1], np.rad2deg(np.arccos(cos)))
    # both: the same winner and inlier set as the oracle on the same samples
    off = 0
    for b, s in enumerate(scenes):
        ret, oi = P.estimate_pose(s[0], s[1], s[2], s[3], 0.05, 0.99, seed=3, return_info=True)
        assert (info[b, 4], info[b, 5]) == oi["best"] and info[b, 1] == oi["inliers"], (b, info[b], oi)
        assert ret is not None and np.array_equal(inl[off:off + n], ret[2])
        # both geometries put the winning root next to another one (a near-double root moves with sqrt(eps), not eps, and the
        # two root finders differ at that level): the poses agree to 1e-3, not to the 1e-6 of a generic scene
        np.testing.assert_allclose(Rg[b], ret[0], atol=1e-3)
        if b == 0:   # under a pure rotation t is whatever the noise in E makes it: not compared
            np.testing.assert_allclose(tg[b], ret[1], atol=1e-3)
        off += n
