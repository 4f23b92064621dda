"""GPU: the batched proposal-crop kernel (preprocess.hip:crop_warp_kernel behind pope_amd/crops.py)
bit for bit against oracle/crop_ref.py — the restatement of eval_linemod_json.py:73-90 with OpenCV's 8-bit bilinear
convention (unpinned against cv2 itself, see the oracle's header) — and feeding the rest of the preprocessing on the card."""
import numpy as np
import pytest
import torch

from oracle import crop_ref as R

pytestmark = pytest.mark.gpu
K_LM = np.array([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1.0]])


@pytest.fixture(scope="module")
def dev(hip_lib):
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def frame(seed, h=480, w=640):
    g = np.random.default_rng(seed)
    smooth = np.add.outer(np.arange(h) * 0.3, np.arange(w) * 0.2)[:, :, None] + g.uniform(0, 60, (1, 1, 3))
    return np.clip(smooth + g.normal(0, 25, (h, w, 3)), 0, 255).astype(np.uint8)


def test_proposal_crops_bit_equal_to_oracle(dev):
    from pope_amd import crops
    img = frame(0)
    boxes = [[200, 150, 90, 120], [0, 0, 64, 48], [560, 400, 80, 80], [300, 10, 33, 71], [5, 300, 250, 100], [620, 460, 20, 20],
             [100, 100, 256, 256], [250, 200, 7, 9]]
    out = crops.crop_proposals(torch.from_numpy(img).to(dev), boxes, K_LM)
    assert out["crops"].shape == (len(boxes), 256, 256, 3) and out["crops"].dtype == torch.uint8 and out["crops"].is_cuda
    got = out["crops"].cpu().numpy()
    for p, b in enumerate(boxes):
        crop, Kc, box = R.crop_proposal(img, b, K_LM)
        assert np.array_equal(got[p], crop), (p, int(np.abs(got[p].astype(int) - crop.astype(int)).max()))
        assert np.array_equal(out["K"][p], Kc) and np.array_equal(out["boxes"][p], box)
        assert crop.any()
    # numpy frame in: uploaded once, same result
    again = crops.crop_proposals(img, boxes[:2], K_LM)
    assert torch.equal(again["crops"], out["crops"][:2])
    assert crops.crop_proposals(img, np.zeros((0, 4)), K_LM)["crops"].shape == (0, 256, 256, 3)


def test_single_crop_drop_in_and_general_warps(dev):
    from pope_amd import crops
    img = frame(1, 120, 160)
    for box, shape in (([20, 10, 90, 70], [60, 70]), ([-10, -5, 60, 55], [60, 70]), ([0, 0, 160, 120], [256, 256]),
                       ([30, 30, 100, 90], [64, 32])):
        got, T = crops.get_image_crop_resize(img, box, shape)
        want, Tw = R.get_image_crop_resize(img, box, shape)
        assert np.array_equal(got, want) and np.array_equal(T, Tw)
    gray = img[:, :, 0].copy()
    got, _ = crops.get_image_crop_resize(gray, [150, 100, 170, 130], [30, 20])
    assert np.array_equal(got, R.get_image_crop_resize(gray, [150, 100, 170, 130], [30, 20])[0]) and got.shape == (30, 20)
    # a rotated similarity through the general entry point
    M = R.get_affine_transform(np.array([80.0, 60.0]), np.array([100, 100]), 25, [128, 128])
    got = crops.warp_batch(torch.from_numpy(img).to(dev), crops._invert(M)[None], [[0, 0, 160, 120]], (128, 128))[0].cpu().numpy()
    assert np.array_equal(got, R.warp_affine_u8(img, M, (128, 128)))


def test_crops_feed_the_batched_preprocessing_on_the_card(dev):
    """frame (uint8, HBM) -> P crops -> set_torch_images / gray_batch without leaving the device; equal to the host chain
    of the reference applied to the oracle's crops."""
    from pope_amd import crops
    from pope_amd.dinov2_utils import _prep
    from pope_amd.preprocess import gray_batch, set_torch_images
    img = frame(2)
    boxes = [[220, 140, 100, 130], [40, 60, 180, 90], [500, 380, 100, 90]]
    out = crops.crop_proposals(torch.from_numpy(img).to(dev), boxes, K_LM)
    x = set_torch_images(out["crops"], center_crop=True)
    g = gray_batch(out["crops"])
    assert x.shape == (3, 3, 196, 196) and g.shape == (3, 1, 256, 256)
    for p, b in enumerate(boxes):
        crop, _, _ = R.crop_proposal(img, b, K_LM)
        assert torch.equal(x[p].cpu(), _prep(crop, (256, 256), (196, 196)))
        bgr = crop.astype(np.int64)
        gray = ((bgr[:, :, 0] * 1868 + bgr[:, :, 1] * 9617 + bgr[:, :, 2] * 4899 + 8192) >> 14).astype(np.float32) / 255.0
        assert np.array_equal(g[p, 0].cpu().numpy(), gray.astype(np.float32))
