"""GPU: whole DINOv2 forward through the C ABI vs the golden fixtures captured from the reference
and vs the CPU oracle on the same seeded inputs.  Tolerance: the north star asks for 1e-3 on
descriptors; the fp32-MFMA path is held to 2e-4 absolute (outputs are O(1))."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ATOL = 2e-4


@pytest.fixture(scope="module", params=["f16x3", "f32"])
def model(hip_lib, sd0, request):
    """Both arithmetic modes of the Linear layers are held to the same tolerances."""
    from pope_amd.dinov2_utils import load_dinov2_model
    assert torch.cuda.is_available()
    m = load_dinov2_model(state_dict=sd0).to("cuda:0")
    m.precision = request.param
    return m


@pytest.mark.parametrize("name", ["vit_196", "vit_224", "vit_476x630"])
def test_forward_matches_reference_fixture(model, golden_dir, name):
    from pope_amd import synth
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    B, H, W = (int(v) for v in fx["shape"])
    x = synth.synthetic_images(B, H, W, seed=int(fx["input_seed"]))
    assert float(x.double().sum()) == fx["input_digest"][0]
    out = model(x.cuda(), is_training=True)
    assert set(out) == {"x_norm_clstoken", "x_norm_patchtokens", "x_prenorm", "masks"} and out["masks"] is None
    rows = torch.from_numpy(fx["rows"])
    xn = torch.cat([out["x_norm_clstoken"][:, None], out["x_norm_patchtokens"]], 1).cpu()[:, rows]
    err = float(np.abs(xn.numpy() - fx["x_norm"]).max())
    print(f"{name}: max |x_norm - reference| = {err:.2e}")
    np.testing.assert_allclose(xn.numpy(), fx["x_norm"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(out["x_prenorm"].cpu()[:, rows].numpy(), fx["x_prenorm"], rtol=0, atol=5 * ATOL)
    np.testing.assert_allclose(model(x.cuda()).cpu().numpy(), fx["cls"], rtol=0, atol=ATOL)
    inter = model.get_intermediate_layers(x.cuda(), n=[0, 5, 11], norm=False, return_class_token=True)
    for (patch, cls), i in zip(inter, (0, 5, 11)):
        blk = torch.cat([cls[:, None], patch], 1).cpu()[:, rows]
        np.testing.assert_allclose(blk.numpy(), fx[f"blk{i}"], rtol=0, atol=5 * ATOL)


def test_forward_matches_oracle_batch(model, sd0):
    from oracle import dinov2_ref
    from pope_amd import synth
    x = synth.synthetic_images(3, 70, 98, seed=3)
    want = dinov2_ref.forward_features(sd0, x)
    got = model.forward_features(x.cuda())
    for k in ("x_norm_clstoken", "x_norm_patchtokens", "x_prenorm"):
        np.testing.assert_allclose(got[k].cpu().numpy(), want[k].numpy(), rtol=0, atol=5 * ATOL)
    # get_intermediate_layers (vision_transformer.py:264-288): last block, normed, reshaped
    (feat,) = model.get_intermediate_layers(x.cuda(), n=1, reshape=True)
    assert feat.shape == (3, 384, 5, 7)
    ref = want["x_norm_patchtokens"].reshape(3, 5, 7, 384).permute(0, 3, 1, 2)
    np.testing.assert_allclose(feat.cpu().numpy(), ref.numpy(), rtol=0, atol=ATOL)


def test_batch_invariance_and_determinism(model):
    # an image's descriptors must not depend on its batch neighbours, and reruns are bitwise equal
    from pope_amd import synth
    x = synth.synthetic_images(4, 56, 84, seed=8).cuda()
    a = model(x, is_training=True)["x_norm_patchtokens"]
    b = model(x[2:3], is_training=True)["x_norm_patchtokens"]
    assert torch.equal(a[2:3], b)
    c = model(x, is_training=True)["x_norm_patchtokens"]
    assert torch.equal(a, c)


def test_oversized_batch_is_split_transparently(model):
    """Batches beyond the kernels' 32-bit addressing are run as several launch sequences (forced here with a tiny
    limit): outputs, taps and caller-provided output buffers are bit-identical to the single-sequence run."""
    from pope_amd import synth
    x = synth.synthetic_images(5, 56, 70, seed=4).to(model.cls_token.device)
    ref = model(x, is_training=True)
    ref_taps = model.get_intermediate_layers(x, n=2, norm=False)
    model._max_batch = 2
    try:
        out = model(x, is_training=True)
        taps = model.get_intermediate_layers(x, n=2, norm=False)
        buf = torch.empty_like(ref["x_prenorm"])
        out2 = model(x, is_training=True, out_norm=buf)
    finally:
        model._max_batch = None
    for k in ("x_norm_clstoken", "x_norm_patchtokens", "x_prenorm"):
        assert torch.equal(out[k], ref[k]), k
    assert all(torch.equal(a, b) for a, b in zip(taps, ref_taps))
    assert out2["x_norm_patchtokens"].data_ptr() == buf[:, 1:].data_ptr() and torch.equal(buf[:, 0], ref["x_norm_clstoken"])


def test_full_size_properties(model):
    # BASELINE shape (476x630, N=1531) at a batch the oracle cannot follow in seconds: check
    # size-independent properties — LayerNorm'd tokens have the affine-transformed unit statistics,
    # and the batched run equals per-image runs bit for bit.
    from pope_amd import synth
    x = synth.synthetic_images(6, 476, 630, seed=13).cuda()
    out = model(x, is_training=True)
    assert out["x_norm_patchtokens"].shape == (6, 1530, 384)
    assert bool(torch.isfinite(out["x_prenorm"]).all())
    w, b = model.norm.weight, model.norm.bias
    z = (torch.cat([out["x_norm_clstoken"][:, None], out["x_norm_patchtokens"]], 1) - b) / w
    assert float(z.mean(-1).abs().max()) < 1e-4 and float((z.var(-1, unbiased=False) - 1).abs().max()) < 1e-3
    single = model(x[4:5], is_training=True)["x_norm_patchtokens"]
    assert torch.equal(single, out["x_norm_patchtokens"][4:5])


def test_vote_top3_matches_reference_loop(model, sd0):
    """Batched CLS scoring + streaming top-3 == the reference's sequential per-proposal loop run on the
    CPU oracle (eval_linemod_json.py:65,74-101): identical slot assignment."""
    import torch.nn.functional as F
    from oracle import coarse_match_ref as cm
    from oracle import dinov2_ref
    from pope_amd import synth
    from pope_amd.pope_model_api import vote_top3
    ref_img = synth.synthetic_images(1, 196, 196, seed=40)
    crops = synth.synthetic_images(9, 196, 196, seed=41)
    crops[4] = ref_img[0] * 0.9 + 0.1 * crops[4]   # a near-duplicate proposal must win a slot
    crops[7] = crops[2]                             # exact tie between two proposals
    scores, slots, idx = vote_top3(model, ref_img.cuda(), crops.cuda())
    ref_cls = dinov2_ref.forward(sd0, ref_img)
    want_scores = torch.cat([F.cosine_similarity(ref_cls, dinov2_ref.forward(sd0, crops[i:i + 1]), dim=1, eps=1e-8)
                             for i in range(9)])
    np.testing.assert_allclose(scores.cpu().numpy(), want_scores.numpy(), rtol=0, atol=1e-5)
    _, want_idx = cm.streaming_top3(want_scores.numpy())
    assert list(idx) == list(want_idx) and 4 in idx


@pytest.mark.parametrize("name", ["vit_224", "vit_476x630", "vitb_224", "vitl_224", "vitl_476x630"])
def test_f16_precision_mode(hip_lib, golden_dir, sd0, name):
    """precision = "f16" (opt-in; BASELINE config 5's dtype): plain f16 operands, one MFMA per product in every block's
    Linear layers and attention, fp32 accumulation / residual stream / softmax / LayerNorm.  Against the fp32 reference
    fixtures the descriptors carry f16 rounding through the depth of the model: held to 1e-2 abs and 1e-3 mean (measured
    1.8e-3 ... 3.2e-3 and 2.7e-4 ... 4.6e-4)."""
    from pope_amd import dinov2, synth
    from pope_amd.dinov2_utils import load_dinov2_model
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    if "arch" in fx:
        dim, depth, heads = (int(v) for v in fx["arch"])
        m = (dinov2.vit_base if dim == 768 else dinov2.vit_large)(patch_size=14, img_size=518, init_values=1e-5, ffn_layer="mlp",
                                                                  block_chunks=0)
        m.load_state_dict(synth.synthetic_state_dict(seed=int(fx["weights_seed"]), dim=dim, depth=depth), strict=True)
        m = m.eval().to("cuda:0")
    else:
        m = load_dinov2_model(state_dict=sd0).to("cuda:0")
    m.precision = "f16"
    B, H, W = (int(v) for v in fx["shape"])
    x = synth.synthetic_images(B, H, W, seed=int(fx["input_seed"])).cuda()
    out = m(x, is_training=True)
    rows = torch.from_numpy(fx["rows"])
    xn = torch.cat([out["x_norm_clstoken"][:, None], out["x_norm_patchtokens"]], 1).cpu()[:, rows].numpy()
    err = np.abs(xn - fx["x_norm"])
    print(f"{name} [f16]: max |x_norm - reference| = {err.max():.2e}, mean {err.mean():.2e}")
    assert np.isfinite(xn).all() and err.max() <= 1e-2 and err.mean() <= 1e-3
    assert m.overflow_events == 0
    assert torch.equal(m(x, is_training=True)["x_norm_patchtokens"], out["x_norm_patchtokens"])   # deterministic
    x3 = torch.cat([synth.synthetic_images(2, H, W, seed=3).cuda(), x[:1]])                        # batch invariant
    assert torch.equal(m(x3, is_training=True)["x_norm_patchtokens"][2], out["x_norm_patchtokens"][0])
    m.precision = "f16x3"   # the modes keep separate weight caches
    xn3 = torch.cat([m(x, is_training=True)[k][:, None] if k.endswith("clstoken") else m(x, is_training=True)[k]
                     for k in ("x_norm_clstoken", "x_norm_patchtokens")], 1).cpu()[:, rows].numpy()
    np.testing.assert_allclose(xn3, fx["x_norm"], rtol=0, atol=ATOL)


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
@pytest.mark.parametrize("name", ["vitb_224", "vitl_224", "vitl_476x630"])
def test_vit_base_and_large_match_reference_fixture(hip_lib, golden_dir, name, prec):
    """BASELINE config 5, DINOv2 half: the ViT-B/14 (768-d, 12 heads) and ViT-L/14 (1024-d, 16 heads, 24 blocks)
    backbones of the reference (vision_transformer.py:319-342) through the same kernels — fixtures generated by the
    reference's own vit_base / vit_large (oracle/gen_golden.py:gen_vit_archs); `vitl_476x630` is the shape bench.py's config-5
    leg runs (640 x 480 centre crop, 1 531 tokens)."""
    from pope_amd import dinov2, synth
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    dim, depth, heads = (int(v) for v in fx["arch"])
    ctor = dinov2.vit_base if dim == 768 else dinov2.vit_large
    m = ctor(patch_size=14, img_size=518, init_values=1e-5, ffn_layer="mlp", block_chunks=0)
    assert m.embed_dim == dim and m.n_blocks == depth and m.num_heads == heads
    m.load_state_dict(synth.synthetic_state_dict(seed=int(fx["weights_seed"]), dim=dim, depth=depth), strict=True)
    m = m.eval().to("cuda:0")
    m.precision = prec
    B, H, W = (int(v) for v in fx["shape"])
    x = synth.synthetic_images(B, H, W, seed=int(fx["input_seed"]))
    assert float(x.double().sum()) == fx["input_digest"][0]
    out = m(x.cuda(), is_training=True)
    rows = torch.from_numpy(fx["rows"])
    xn = torch.cat([out["x_norm_clstoken"][:, None], out["x_norm_patchtokens"]], 1).cpu()[:, rows]
    err = float(np.abs(xn.numpy() - fx["x_norm"]).max())
    print(f"{name} [{prec}]: max |x_norm - reference| = {err:.2e}")
    np.testing.assert_allclose(xn.numpy(), fx["x_norm"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(out["x_prenorm"].cpu()[:, rows].numpy(), fx["x_prenorm"], rtol=0, atol=5 * ATOL)
    np.testing.assert_allclose(m(x.cuda()).cpu().numpy(), fx["cls"], rtol=0, atol=ATOL)
    taps = [int(t) for t in fx["tap_blocks"]]
    inter = m.get_intermediate_layers(x.cuda(), n=taps, norm=False, return_class_token=True)
    for (patch, cls), i in zip(inter, taps):
        blk = torch.cat([cls[:, None], patch], 1).cpu()[:, rows]
        np.testing.assert_allclose(blk.numpy(), fx[f"blk{i}"], rtol=0, atol=5 * ATOL)
    assert m.overflow_events == 0
    # batch invariance at the wider shapes (16 heads / hidden 4096 grids)
    x3 = torch.cat([synth.synthetic_images(2, H, W, seed=3), x]).cuda()
    assert torch.equal(m(x3, is_training=True)["x_norm_patchtokens"][2], out["x_norm_patchtokens"][0])


def test_in_place_weight_edits_refresh_the_derived_planes(hip_lib, sd0):
    """The weight planes and the pos / bias table are keyed by the address AND torch's version counter of every parameter:
    `p.copy_()`, an optimizer step or `w[i, j] = x` after a forward pass shows up in the next one (nothing to invalidate by
    hand), exactly like a fresh model built from the edited weights."""
    import torch
    from pope_amd import synth
    from pope_amd.dinov2_utils import load_dinov2_model
    m = load_dinov2_model(state_dict=sd0).cuda()
    x = synth.synthetic_images(2, 56, 84, seed=1).cuda()
    y0 = m(x, is_training=True)["x_norm_patchtokens"].clone()
    sd1 = {k: v.clone() for k, v in sd0.items()}
    with torch.no_grad():
        m.blocks[3].mlp.fc1.weight[5, 7] += 0.5             # a GEMM weight (planes)
        m.pos_embed[0, 3, :8] += 0.25                       # the pos / bias table
        m.blocks[9].attn.proj.weight.mul_(1.01)             # whole-tensor in-place op
    sd1["blocks.3.mlp.fc1.weight"][5, 7] += 0.5
    sd1["pos_embed"][0, 3, :8] += 0.25
    sd1["blocks.9.attn.proj.weight"].mul_(1.01)
    y1 = m(x, is_training=True)["x_norm_patchtokens"]
    fresh = load_dinov2_model(state_dict=sd1).cuda()
    assert not torch.equal(y0, y1) and torch.equal(y1, fresh(x, is_training=True)["x_norm_patchtokens"])


def test_f16_mode_is_batch_invariant_across_the_gemm_switch(hip_lib, sd0):
    """precision "f16" has two GEMM mainloops: gemm_planes16_kernel's 128 x 128 tiles (small M) and gemm_plain.hip's 256-row tiles
    with LDS-direct staging (M >= 2 048; round 4).  Both accumulate every output in the same order and share the epilogue
    arithmetic: an image must come out bit-identical alone (257 rows: the tile kernel) and inside a batch of nine (2 313 rows:
    the new kernel for QKV -> f16 attention operands, FC1 . GELU -> f16, proj / FC2 + residual); the attention kernel (attention_f16.hip) is the same at every batch."""
    from pope_amd import synth
    from pope_amd.dinov2_utils import load_dinov2_model
    m = load_dinov2_model(state_dict=sd0).to("cuda:0")
    m.precision = "f16"
    x = synth.synthetic_images(9, 224, 224, seed=77).cuda()
    big = m(x, is_training=True)
    for k in (0, 4, 8):
        one = m(x[k:k + 1], is_training=True)
        for key in ("x_norm_clstoken", "x_norm_patchtokens", "x_prenorm"):
            assert torch.equal(one[key][0], big[key][k]), (k, key)
    assert m.overflow_events == 0 and bool(torch.isfinite(big["x_prenorm"]).all())
