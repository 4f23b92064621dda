"""GPU: batched preprocessing kernels against the host path of the reference's `set_torch_image`
(PIL resize + crop + ToTensor + Normalize = pope_amd.dinov2_utils._prep, itself pinned to Pillow): BIT-identical."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(hip_lib):
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("hw", [(256, 256), (480, 640), (300, 400), (100, 120), (513, 257)])
@pytest.mark.parametrize("center_crop", [True, False])
def test_set_torch_images_equals_host_path(dev, hw, center_crop):
    from pope_amd import dinov2_utils as du
    from pope_amd.preprocess import set_torch_images
    rng = np.random.default_rng(hw[0] * 7 + hw[1])
    imgs = rng.integers(0, 256, (5,) + hw + (3,), dtype=np.uint8)
    imgs[3] = 255
    imgs[4, ::2] = 0
    got = set_torch_images(imgs, center_crop=center_crop)
    want = torch.stack([du._prep(i, (256, 256), (196, 196)) if center_crop else du._prep(i, (224, 224), None) for i in imgs])
    assert got.shape == want.shape and got.dtype == torch.float32 and got.is_cuda
    assert torch.equal(got.cpu(), want)
    # the drop-in single-image entry point goes through the same kernels
    one = du.set_torch_image(imgs[1], center_crop=center_crop)
    assert one.shape == (1,) + tuple(want.shape[1:]) and torch.equal(one.cpu()[0], want[1])


def test_gray_batch(dev):
    from pope_amd.preprocess import gray_batch
    rng = np.random.default_rng(3)
    bgr = rng.integers(0, 256, (3, 64, 48, 3), dtype=np.uint8)
    got = gray_batch(torch.from_numpy(bgr).to(dev)).cpu().numpy()
    b, g, r = (bgr[..., c].astype(np.int64) for c in range(3))
    want = ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.float32) / np.float32(255.0)
    assert got.shape == (3, 1, 64, 48) and np.array_equal(got[:, 0], want)
    assert float(got.max()) <= 1.0 and abs(float(got.mean()) - float(bgr.mean()) / 255) < 0.02


def test_preprocessed_batch_feeds_the_vit(dev, sd0):
    """proposals uint8 -> GPU preprocessing -> batched CLS scores == the per-proposal host path of the reference loop"""
    from pope_amd import dinov2_utils as du
    from pope_amd.dinov2_utils import load_dinov2_model
    from pope_amd.preprocess import set_torch_images
    rng = np.random.default_rng(9)
    crops = rng.integers(0, 256, (6, 256, 256, 3), dtype=np.uint8)
    m = load_dinov2_model(state_dict=sd0).to(dev)
    a = m(set_torch_images(crops, center_crop=True))
    b = torch.cat([m(du._prep(c, (256, 256), (196, 196))[None].to(dev)) for c in crops])
    assert torch.equal(a, b)


def test_driver_step_from_uint8_frames(dev, sd0):
    """locate_and_match_u8 == locate_and_match on host-preprocessed tensors (PIL path + the cv2 gray formula)."""
    from pope_amd import dinov2_utils as du, synth
    from pope_amd.dinov2_utils import load_dinov2_model
    from pope_amd.driver import locate_and_match, locate_and_match_u8
    from pope_amd.matcher import Matcher, default_cfg
    rng = np.random.default_rng(21)
    ref = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
    crops = rng.integers(0, 256, (5, 256, 256, 3), dtype=np.uint8)
    crops[2] = ref   # an identical proposal must win
    m = load_dinov2_model(state_dict=sd0).to(dev)
    matcher = Matcher(default_cfg).eval()
    matcher.load_state_dict(synth.synthetic_matcher_state_dict(seed=0), strict=True)
    matcher = matcher.to(dev)
    got = locate_and_match_u8(m, matcher, ref, crops)

    def gray(a):
        b, g, r = (a[..., c].astype(np.int64) for c in range(3))
        return torch.from_numpy((((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.float32) / np.float32(255.0)))

    want = locate_and_match(m, matcher, du._prep(ref, (256, 256), (196, 196))[None].to(dev),
                            torch.stack([du._prep(c, (256, 256), (196, 196)) for c in crops]).to(dev),
                            gray(ref)[None, None].to(dev), gray(crops)[:, None].to(dev))
    assert torch.equal(got["scores"], want["scores"]) and list(got["slot_index"]) == list(want["slot_index"])
    assert 2 in got["slot_index"] and got["best_proposal"] == want["best_proposal"]
    for s in range(3):
        assert np.array_equal(got["mkpts0"][s], want["mkpts0"][s]) and np.array_equal(got["mconf"][s], want["mconf"][s])


def test_crop_normalize_matches_torchvision_arithmetic(dev):
    """pope_crop_normalize_u8_f32 (the dense pair path's centre crop + ToTensor + Normalize, no resize) bit-for-bit against
    the fp32 arithmetic of torchvision's ToTensor (x / 255) and Normalize ((x - mean) / std)."""
    from pope_amd.preprocess import crop_normalize
    from pope_amd.synth import IMAGENET_MEAN, IMAGENET_STD
    g = torch.Generator().manual_seed(11)
    for (P, H, W, ch, cw) in [(3, 480, 640, 476, 630), (2, 37, 53, 30, 41), (1, 8, 8, 8, 8)]:
        img = torch.randint(0, 256, (P, H, W, 3), dtype=torch.uint8, generator=g)
        got = crop_normalize(img.to(dev), (ch, cw)).cpu()
        top, left = (H - ch) // 2, (W - cw) // 2
        x = img[:, top:top + ch, left:left + cw, :].permute(0, 3, 1, 2).float() / 255.0
        want = (x - torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)) / torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
        assert torch.equal(got, want), (P, H, W)
