"""CPU: proposal crop + intrinsics update.  oracle/crop_ref.py restates
eval_linemod_json.py:73-90 / utils/data_utils.py:22-52,239-280; cv2 is absent ("parity unpinned" for the pixels), so the
oracle is held to what the convention implies without cv2: integer translations copy pixels exactly, a linear ramp is
reproduced to the 1/32 px quantisation, the intrinsics equal the closed form T2 . T1 . K.  The product's host half
(pope_amd/crops.py: boxes, matrices, intrinsics) must equal the oracle exactly."""
import numpy as np
import pytest

from oracle import crop_ref as R
from pope_amd import crops

K_LM = np.array([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1.0]])
BOXES = [[100, 80, 60, 90], [0, 0, 50, 40], [600, 400, 80, 120], [10, 20, 33, 71], [317, 5, 200, 31]]


def test_box_expansion_and_closed_form_intrinsics():
    for b in BOXES:
        box = R.expand_box(b)
        assert np.array_equal(box, crops.expand_box(b))
        x, y, w, h = b
        assert list(box) == [x - int(w * 0.3), y - int(h * 0.3), x + w + int(w * 0.3), y + h + int(h * 0.3)]
        x0, y0, x1, y1 = box
        bw, bh = x1 - x0, y1 - y0
        K1, K1h = R.get_K_crop_resize(box, K_LM, [bh, bw])
        K2, _ = R.get_K_crop_resize([0, 0, bw, bh], K1, [256, 256])
        T1 = np.array([[1, 0, -x0], [0, 1, -y0], [0, 0, 1.0]])
        s = 256.0 / bw                       # widths only: the height of the box does not enter (data_utils.py:31-33)
        T2 = np.array([[s, 0, 128 - s * bw / 2], [0, s, 128 - s * bh / 2], [0, 0, 1]])
        np.testing.assert_allclose(K1, T1 @ K_LM, rtol=0, atol=1e-9)
        np.testing.assert_allclose(K2, T2 @ T1 @ K_LM, rtol=0, atol=1e-9)
        assert K1h.shape == (3, 4) and np.all(K1h[:, 3] == 0)
        # the product's host arithmetic is the oracle's, bit for bit
        P1, P1h = crops.get_K_crop_resize(box, K_LM, [bh, bw])
        P2, _ = crops.get_K_crop_resize([0, 0, bw, bh], P1, [256, 256])
        assert np.array_equal(P1, K1) and np.array_equal(P1h, K1h) and np.array_equal(P2, K2)
    # [3, 4] input is used as it is (data_utils.py:271-275)
    Kh = np.concatenate([K_LM, np.array([[1.0], [2.0], [3.0]])], 1)
    assert np.array_equal(R.get_K_crop_resize([0, 0, 10, 10], Kh, [10, 10])[1], crops.get_K_crop_resize([0, 0, 10, 10], Kh, [10, 10])[1])


def test_affine_transform_matches_oracle_incl_rotation_and_inverse():
    for rot in (0, 30, -75):
        for inv in (0, 1):
            a = R.get_affine_transform(np.array([50.5, 40.0]), np.array([80, 60]), rot, [256, 128], inv=inv)
            b = crops.get_affine_transform(np.array([50.5, 40.0]), np.array([80, 60]), rot, [256, 128], inv=inv)
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-12)
    m = R.get_affine_transform(np.array([50.0, 40.0]), np.array([80, 60]), 0, [160, 120])
    np.testing.assert_allclose(m, [[2, 0, -20], [0, 2, -20]], atol=1e-12)      # scale 160 / 80, centre -> (80, 60)
    np.testing.assert_allclose(crops._invert(m), R.invert_affine(m), rtol=0, atol=0)
    np.testing.assert_allclose(R.invert_affine(m), [[0.5, 0, 10], [0, 0.5, 10]], atol=1e-12)


def test_integer_translation_is_an_exact_copy_with_zero_padding():
    g = np.random.default_rng(0)
    img = g.integers(0, 256, (120, 160, 3), dtype=np.uint8)
    c, T = R.get_image_crop_resize(img, [20, 10, 90, 70], [60, 70])
    assert np.array_equal(c, img[10:70, 20:90]) and np.allclose(T, [[1, 0, -20], [0, 1, -10], [0, 0, 1]])
    c, _ = R.get_image_crop_resize(img, [-10, -5, 60, 55], [60, 70])          # leaves the frame: zeros
    want = np.zeros((60, 70, 3), np.uint8)
    want[5:, 10:] = img[:55, :60]
    assert np.array_equal(c, want)
    c, _ = R.get_image_crop_resize(img[:, :, 0], [150, 100, 170, 130], [30, 20])   # gray image, bottom-right corner
    want = np.zeros((30, 20), np.uint8)
    want[:20, :10] = img[100:120, 150:160, 0]
    assert np.array_equal(c, want)


def test_scaling_reproduces_a_linear_ramp():
    """I(x, y) = 0.5 x + 0.25 y + 10 sampled at 256 / w: bilinear interpolation is exact on a ramp, so the only errors are
    the 1/32 px position quantisation (<= slope / 64 + slope / 32 truncation) and the final rounding."""
    h, w = 90, 120
    yy, xx = np.mgrid[0:h, 0:w]
    ramp = 0.5 * xx + 0.25 * yy + 10
    img = np.round(ramp).astype(np.uint8)
    out, T = R.get_image_crop_resize(img, [0, 0, w, h], [256, 256])
    s = 256.0 / w
    np.testing.assert_allclose(T, [[s, 0, 0], [0, s, 128 - s * h / 2], [0, 0, 1]], atol=1e-9)
    Y, X = np.mgrid[0:256, 0:256]
    sx, sy = X / s, (Y - (128 - s * h / 2)) / s
    inside = (sx >= 0) & (sx <= w - 1) & (sy >= 0) & (sy <= h - 1)
    want = 0.5 * sx + 0.25 * sy + 10
    err = np.abs(out.astype(np.float64) - want)[inside]
    assert inside.sum() > 40000 and err.max() <= 0.5 + 0.5 + 0.75 / 32 + 0.01     # input rounding + output rounding + position
    assert np.all(out[sy < -1.01] == 0) and np.all(out[sy > h + 0.01] == 0)       # rows beyond the crop: zero padding


def test_warp_agrees_with_an_independent_float_bilinear():
    """cv2 is absent, so the fixed-point restatement is also held to an independent implementation: scipy's float bilinear
    sampling (order 1, constant 0 outside) at the exactly inverse-mapped coordinates.  On a smooth image (gradient G levels
    per pixel) the two may differ by the 1/32-pixel position truncation (G / 32 per axis), the 10-bit weights and the final
    rounding; the mean signed difference shows the truncation's half-step bias only."""
    from scipy import ndimage
    yy, xx = np.mgrid[0:150, 0:200].astype(np.float64)
    base = 128 + 60 * np.sin(xx / 25) * np.cos(yy / 40) + 50 * np.cos(yy / 20 + xx / 90)
    img = np.round(base).astype(np.uint8)
    gy, gx = np.gradient(img.astype(np.float64))
    G = max(np.abs(gx).max(), np.abs(gy).max())
    assert G <= 6.0
    for center, scale, rot, size in [((100.0, 75.0), 90.0, 0.0, (256, 256)), ((60.5, 80.25), 140.0, 0.0, (256, 256)),
                                     ((120.0, 70.0), 77.0, 30.0, (192, 160))]:
        M = R.get_affine_transform(np.array(center, np.float32), scale, rot, size)
        out = R.warp_affine_u8(img, M, size)
        Mi = R.invert_affine(M)
        Y, X = np.mgrid[0:size[1], 0:size[0]].astype(np.float64)
        sx = Mi[0, 0] * X + Mi[0, 1] * Y + Mi[0, 2]
        sy = Mi[1, 0] * X + Mi[1, 1] * Y + Mi[1, 2]
        want = ndimage.map_coordinates(img.astype(np.float64), [sy, sx], order=1, mode="constant", cval=0.0)
        inside = (sx >= 1) & (sx <= img.shape[1] - 2) & (sy >= 1) & (sy <= img.shape[0] - 2)
        d = out.astype(np.float64) - want
        assert inside.sum() > 5000
        assert np.abs(d[inside]).max() <= 0.5 + 2 * G / 32 + 0.05, (center, scale, rot, np.abs(d[inside]).max())
        assert abs(d[inside].mean()) <= G / 32
        far = (sx < -1.01) | (sx > img.shape[1] + 0.01) | (sy < -1.01) | (sy > img.shape[0] + 0.01)
        assert np.all(out[far] == 0)


def test_two_step_crop_equals_one_composite_gather():
    """The reference's crop-then-resize (two warps, an intermediate zero-padded crop) == sampling the frame through the
    crop window directly — the identity the one-launch GPU kernel relies on."""
    g = np.random.default_rng(1)
    img = g.integers(0, 256, (96, 128, 3), dtype=np.uint8)
    for b in ([30, 20, 40, 50], [-4, 60, 50, 30], [100, 70, 40, 40]):
        crop, Kc, box = R.crop_proposal(img, b, K_LM, out_size=64)
        x0, y0, x1, y1 = box
        w, h = x1 - x0, y1 - y0
        pad = np.zeros((h, w, 3), np.uint8)      # the intermediate crop, by hand
        ys, xs = np.arange(y0, y1), np.arange(x0, x1)
        oky, okx = (ys >= 0) & (ys < 96), (xs >= 0) & (xs < 128)
        pad[np.ix_(oky, okx)] = img[np.ix_(ys[oky], xs[okx])]
        want, _ = R.get_image_crop_resize(pad, [0, 0, w, h], [64, 64])
        assert np.array_equal(crop, want) and crop.shape == (64, 64, 3)


def test_product_refuses_cpu_tensors():
    import torch
    from pope_amd._lib import PopeHipError
    with pytest.raises(PopeHipError):
        crops.warp_batch(torch.zeros(8, 8, 3, dtype=torch.uint8), np.eye(2, 3)[None], [[0, 0, 8, 8]], (4, 4))
