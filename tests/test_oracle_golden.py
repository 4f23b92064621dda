"""CPU: the oracle restatement reproduces the fixtures captured from the reference's own code
(oracle/gen_golden.py), and the seeded generators reproduce the fixture inputs."""
import os

import numpy as np
import pytest
import torch

from oracle import coarse_match_ref as cm
from oracle import dinov2_ref
from pope_amd import synth


def _digest(sd):
    return np.array([float(sd[k].double().sum()) for k in sorted(sd)], np.float64)


def test_state_dict_layout(sd0):
    assert len(sd0) == 175
    assert sd0["pos_embed"].shape == (1, 1370, 384)
    assert sd0["blocks.11.attn.qkv.weight"].shape == (1152, 384)
    assert dinov2_ref.arch_from_state_dict(sd0) == (384, 12, 6, 14, 37)


@pytest.mark.parametrize("name", ["vit_196", "vit_224", "vit_476x630"])
def test_vit_oracle_matches_reference_fixture(name, sd0, golden_dir):
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    np.testing.assert_allclose(_digest(sd0), fx["weights_digest"], rtol=0, atol=0)
    B, H, W = (int(v) for v in fx["shape"])
    x = synth.synthetic_images(B, H, W, seed=int(fx["input_seed"]))
    assert float(x.double().sum()) == fx["input_digest"][0]
    taps = {}
    out = dinov2_ref.forward_features(sd0, x, taps=taps)
    rows = torch.from_numpy(fx["rows"])
    xn = torch.cat([out["x_norm_clstoken"][:, None], out["x_norm_patchtokens"]], 1)[:, rows]
    np.testing.assert_allclose(xn.numpy(), fx["x_norm"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(out["x_prenorm"][:, rows].numpy(), fx["x_prenorm"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(taps["tokens"][:, rows].numpy(), fx["tokens"], rtol=0, atol=1e-6)
    for i in (0, 5, 11):
        np.testing.assert_allclose(taps[i]["attn_branch"][:, rows].numpy(), fx[f"attn{i}"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(taps[i]["mlp_branch"][:, rows].numpy(), fx[f"mlp{i}"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(taps[i]["x_out"][:, rows].numpy(), fx[f"blk{i}"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(dinov2_ref.forward(sd0, x).numpy(), fx["cls"], rtol=0, atol=1e-5)
    # branch outputs must be O(1), otherwise the fixture would not exercise the kernels
    assert np.abs(fx["attn5"]).mean() > 0.05 and np.abs(fx["mlp5"]).mean() > 0.05


def test_flops_closed_form():
    # 9.17 / 12.25 / 108.91 GF
    assert abs(dinov2_ref.flops_per_image(196) / 1e9 - 9.17) < 0.01
    assert abs(dinov2_ref.flops_per_image(256) / 1e9 - 12.25) < 0.01
    assert abs(dinov2_ref.flops_per_image(1530) / 1e9 - 108.912) < 0.01
    n, np_ = 1531, 1530
    assert dinov2_ref.flops_per_image(np_) == 451584 * np_ + 12 * (3538944 * n + 1536 * n * n)


def test_matcher_oracle_loftr_fixture(golden_dir):
    fx = np.load(os.path.join(golden_dir, "match_loftr256.npz"))
    f0, f1 = torch.from_numpy(fx["feat0"]), torch.from_numpy(fx["feat1"])
    hw_c, hw_i = tuple(int(v) for v in fx["hw_c"]), tuple(int(v) for v in fx["hw_i"])
    out = cm.dense_match(f0, f1, hw_c, hw_c, hw_i)
    for k in ("b_ids", "i_ids", "j_ids"):
        assert np.array_equal(out[k].numpy(), fx[k]), k
    for k in ("mconf", "mkpts0_c", "mkpts1_c", "conf_matrix"):
        np.testing.assert_allclose(out[k].numpy(), fx[k], rtol=0, atol=1e-6)
    assert out["mkpts0_c"].dtype == torch.float32 and out["i_ids"].dtype == torch.int64
    # ordering by (b, i)
    key = out["b_ids"] * 10**6 + out["i_ids"]
    assert bool((key[1:] > key[:-1]).all())


def test_matcher_border_capacity():
    # identical features -> every interior cell matches itself: (h-4)*(w-4) matches
    g = torch.Generator().manual_seed(1)
    h, w = 9, 11
    f = torch.randn(1, h * w, 64, generator=g) * 4
    out = cm.dense_match(f, f, (h, w), (h, w), (h * 14, w * 14))
    assert len(out["i_ids"]) == (h - 4) * (w - 4)
    assert bool((out["i_ids"] == out["j_ids"]).all())
    np.testing.assert_array_equal(out["mkpts0_c"][:, 0].numpy(), (out["i_ids"] % w).numpy() * 14.0)


def test_matcher_empty():
    g = torch.Generator().manual_seed(2)
    f0 = torch.randn(1, 36, 32, generator=g)
    f1 = torch.randn(1, 36, 32, generator=g)
    out = cm.dense_match(f0, f1, (6, 6), (6, 6), (48, 48))
    assert len(out["i_ids"]) == 0 and out["mkpts0_c"].shape == (0, 2)


def test_top3_fixture(golden_dir):
    fx = np.load(os.path.join(golden_dir, "top3.npz"))
    scores = cm.cls_cosine(torch.from_numpy(fx["ref"]), torch.from_numpy(fx["fea"]))
    np.testing.assert_allclose(scores.numpy(), fx["scores"], rtol=0, atol=1e-6)
    slots, idx = cm.streaming_top3(fx["scores"])
    assert np.array_equal(slots, fx["slot_scores"]) and np.array_equal(idx, fx["slot_index"])


def test_top3_edge_cases():
    s, i = cm.streaming_top3([])
    assert list(s) == [0, 0, 0] and list(i) == [-1, -1, -1]
    s, i = cm.streaming_top3([-0.5, 0.0, -1.0])          # scores <= 0 never enter
    assert list(i) == [-1, -1, -1]
    s, i = cm.streaming_top3([0.5, 0.5, 0.5, 0.5])        # ties: strict '>' keeps the first three
    assert list(i) == [0, 1, 2]
    s, i = cm.streaming_top3([0.1, 0.2, 0.3, 0.25])       # replaces the first minimum
    assert list(i) == [3, 1, 2]


def test_cls_cosine_eps_semantics():
    # each norm clamped separately
    a = torch.tensor([[1e-9, 0.0, 0.0]])
    b = torch.tensor([[2e-9, 0.0, 0.0]])
    assert abs(float(cm.cls_cosine(a, b)) - 0.02) < 1e-6
