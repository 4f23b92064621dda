"""CPU: the SAM image encoder oracle against the reference-generated fixtures, and the drop-in module's state-dict
layout (BASELINE config 5).  No HIP calls."""
import os
from functools import partial

import numpy as np
import pytest
import torch


@pytest.mark.parametrize("name", ["sam_hd80_256", "sam_hd64_224"])
def test_oracle_reproduces_reference_fixture(golden_dir, name):
    from oracle import sam_encoder_ref
    from pope_amd import synth
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    dim, depth, heads, img, window = (int(v) for v in fx["arch"])
    gidx = tuple(int(v) for v in fx["global_idx"])
    sd = synth.synthetic_sam_encoder_state_dict(seed=int(fx["weights_seed"]), dim=dim, depth=depth, heads=heads, grid=img // 16,
                                                window=window, global_idx=gidx)
    assert np.array_equal(np.array([float(sd[k].double().sum()) for k in sorted(sd)]), fx["weights_digest"])
    x = synth.synthetic_images(int(fx["batch"]), img, img, seed=int(fx["input_seed"]))
    taps = {int(i): None for i in fx["tap_blocks"]}
    with torch.no_grad():
        out = sam_encoder_ref.forward(sd, x, heads, window, gidx, taps)
    stride, ts = int(fx["stride"]), max(2, int(fx["stride"]))
    np.testing.assert_allclose(out[:, :, ::stride, ::stride].numpy(), fx["out"], rtol=0, atol=2e-5)
    for i, t in taps.items():
        np.testing.assert_allclose(t[:, ::ts, ::ts, ::2].numpy(), fx[f"blk{i}"], rtol=0, atol=5e-5)


def test_module_has_the_reference_state_dict_layout():
    """Same keys and shapes as segment_anything's ImageEncoderViT (image_encoder.py:53-105): a checkpoint slice
    `image_encoder.*` of build_sam.py:102-105 loads with strict=True."""
    from pope_amd import synth
    from pope_amd.sam_encoder import ImageEncoderViT, get_rel_pos
    from oracle import sam_encoder_ref
    m = ImageEncoderViT(depth=4, embed_dim=640, img_size=256, mlp_ratio=4, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6),
                        num_heads=8, patch_size=16, qkv_bias=True, use_rel_pos=True, global_attn_indexes=[1, 3], window_size=14,
                        out_chans=256)
    sd = synth.synthetic_sam_encoder_state_dict(dim=640, depth=4, heads=8, grid=16, window=14, global_idx=(1, 3))
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in sd.items()}
    m.load_state_dict(sd, strict=True)
    assert m.blocks[0].window_size == 14 and m.blocks[1].window_size == 0
    # ViT-H key count as build_sam.py:13-21 (32 blocks x 14 + pos + patch 2 + neck 6)
    big = synth.synthetic_sam_encoder_state_dict(dim=128, depth=32, heads=2, grid=4, window=2, global_idx=(7, 15, 23, 31))
    assert len(big) == 32 * 14 + 9
    # the host-side relative-position gather, including the interpolated case (image_encoder.py:299-307)
    for q, L in ((14, 27), (16, 27), (8, 27)):
        rp = torch.randn(L, 80)
        assert torch.equal(get_rel_pos(q, q, rp), sam_encoder_ref.rel_pos_table(q, q, rp))


def test_forward_without_gpu_fails_loudly():
    from pope_amd.sam_encoder import ImageEncoderViT
    m = ImageEncoderViT(depth=1, embed_dim=256, img_size=224, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_heads=4,
                        use_rel_pos=True, window_size=14)
    with pytest.raises((RuntimeError, ValueError, TypeError)):
        m(torch.zeros(1, 3, 224, 224))
