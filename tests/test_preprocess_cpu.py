"""CPU: host-side preprocessing of the drop-in `set_torch_image` (segment_anything/.../dinov2_utils.py:55-78):
ToPILImage -> Resize((256,256)) -> CenterCrop((196,196)) | Resize((224,224)) -> ToTensor -> Normalize.  torchvision is
not in this image, so the resize itself (PIL bilinear, what torchvision calls for PIL inputs) is unpinned against the
reference; geometry, channel handling and normalisation are checked here."""
import numpy as np
import torch

from pope_amd import dinov2_utils as du
from pope_amd.synth import IMAGENET_MEAN, IMAGENET_STD


def test_shapes_and_normalisation_of_constant_image():
    img = np.full((300, 400, 3), (10, 128, 250), np.uint8)   # channels are taken as given (the drivers pass BGR)
    for crop, hw in ((True, 196), (False, 224)):
        t = du._prep(img, (256, 256), (196, 196)) if crop else du._prep(img, (224, 224), None)
        assert t.shape == (3, hw, hw) and t.dtype == torch.float32
        for c, v in enumerate((10, 128, 250)):
            want = (v / 255 - IMAGENET_MEAN[c]) / IMAGENET_STD[c]
            assert torch.allclose(t[c], torch.full_like(t[c], want), atol=1e-6)


def test_center_crop_geometry():
    # 256x256 after resize; crop offsets round((256-196)/2) = 30 on both axes (torchvision CenterCrop)
    img = np.zeros((256, 256, 3), np.uint8)
    img[30:226, 30:226] = 255
    t = du._prep(img, (256, 256), (196, 196))
    hi = (1.0 - IMAGENET_MEAN[0]) / IMAGENET_STD[0]
    assert torch.allclose(t[0], torch.full_like(t[0], hi), atol=1e-6)   # exactly the bright square


def test_gray_and_tensor_inputs():
    g = (np.arange(64 * 48).reshape(64, 48) % 256).astype(np.uint8)
    rgb = np.stack([g, g, g], -1)
    a, b = du._prep(rgb, (224, 224), None), du._prep(torch.from_numpy(rgb), (224, 224), None)
    assert torch.equal(a, b)
    un = a * torch.tensor(IMAGENET_STD).view(3, 1, 1) + torch.tensor(IMAGENET_MEAN).view(3, 1, 1)
    assert torch.allclose(un[0], un[1], atol=1e-6) and float(un.min()) >= -1e-6 and float(un.max()) <= 1 + 1e-6


def test_pairs_by_id_depend_on_the_id_only():
    """Synthetic pixels of a work-list pair (BASELINE config 4) are a pure function of its id: the same pair gives the
    same images whatever batch, shard or rank it is generated in."""
    import torch
    from pope_amd import synth
    a0, a1 = synth.pairs_by_id([5, 9, 100], 56, 70)
    b0, b1 = synth.pairs_by_id([9], 56, 70)
    assert torch.equal(a0[1], b0[0]) and torch.equal(a1[1], b1[0]) and not torch.equal(a0[0], a0[1])
    noise = a1 - torch.roll(a0, (14, 28), (2, 3))
    assert abs(float(noise.std()) - 0.1) < 5e-3 and abs(float(noise.mean())) < 5e-3


def test_pil_resize_restatement_is_bit_exact():
    """oracle/pil_resize_ref.py (the restatement of Pillow's 8-bit bilinear resample) against Pillow itself, and the
    product's table builder against the oracle's: down- and up-scaling, identity, odd sizes."""
    from PIL import Image
    from oracle.pil_resize_ref import resize_bilinear_u8, resize_tables as ref_tables
    from pope_amd.preprocess import resize_tables
    rng = np.random.default_rng(0)
    for (h, w, oh, ow) in [(480, 640, 256, 256), (300, 400, 224, 224), (100, 120, 256, 256), (256, 256, 256, 256),
                           (513, 257, 256, 256), (37, 41, 224, 224)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        want = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(resize_bilinear_u8(img, oh, ow), want), (h, w, oh, ow)
        for a, b in ((h, oh), (w, ow)):
            for x, y in zip(resize_tables(a, b), ref_tables(a, b)):
                assert np.array_equal(x, y)
