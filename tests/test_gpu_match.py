"""GPU: dense matcher through the C ABI — identical match indices to the reference fixtures,
mconf within 1e-5, ordering, ties, border capacity, empty output."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(hip_lib):
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(autouse=True, params=["f16x3", "f32"])
def contraction_precision(request, monkeypatch):
    """Every test of this module runs with both arithmetic modes of the L x S x C contraction (same fixtures, same
    bit-exact index requirements)."""
    import pope_amd.matcher as m
    monkeypatch.setattr(m, "DEFAULT_PRECISION", request.param)
    return request.param


def test_loftr_shaped_fixture_indices_identical(dev, golden_dir):
    from pope_amd.matcher import dense_match
    fx = np.load(os.path.join(golden_dir, "match_loftr256.npz"))
    hw_c, hw_i = tuple(int(v) for v in fx["hw_c"]), tuple(int(v) for v in fx["hw_i"])
    out = dense_match(torch.from_numpy(fx["feat0"]).to(dev), torch.from_numpy(fx["feat1"]).to(dev), hw_c, hw_c, hw_i)
    for k in ("b_ids", "i_ids", "j_ids"):
        assert out[k].dtype == torch.int64
        assert np.array_equal(out[k].cpu().numpy(), fx[k]), k
    np.testing.assert_allclose(out["mconf"].cpu().numpy(), fx["mconf"], rtol=1e-4, atol=1e-6)
    assert np.array_equal(out["mkpts0_c"].cpu().numpy(), fx["mkpts0_c"])
    assert np.array_equal(out["mkpts1_c"].cpu().numpy(), fx["mkpts1_c"])
    np.testing.assert_allclose(out["conf_matrix"].cpu().numpy(), fx["conf_matrix"], rtol=1e-4, atol=1e-7)


def test_coarse_matching_module_updates_dict(dev, golden_dir):
    from pope_amd.matcher import CoarseMatching, default_cfg
    fx = np.load(os.path.join(golden_dir, "match_loftr256.npz"))
    hw_c, hw_i = tuple(int(v) for v in fx["hw_c"]), tuple(int(v) for v in fx["hw_i"])
    data = {"hw0_i": hw_i, "hw1_i": hw_i, "hw0_c": hw_c, "hw1_c": hw_c}
    m = CoarseMatching(default_cfg["match_coarse"]).eval()
    assert m(torch.from_numpy(fx["feat0"]).to(dev), torch.from_numpy(fx["feat1"]).to(dev), data) is None
    for k in ("conf_matrix", "b_ids", "i_ids", "j_ids", "gt_mask", "m_bids", "mkpts0_c", "mkpts1_c", "mconf"):
        assert k in data
    assert np.array_equal(data["j_ids"].cpu().numpy(), fx["j_ids"])


@pytest.mark.parametrize("name", ["match_224", "match_476x630"])
def test_dinov2_token_matching_end_to_end(dev, sd0, golden_dir, name):
    """extract (HIP ViT) + match (HIP matcher) vs the reference pipeline's fixture."""
    from pope_amd import synth
    from pope_amd.dinov2_utils import load_dinov2_model
    from pope_amd.matcher import dense_match
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    n, H, W = (int(v) for v in fx["shape"])
    model = load_dinov2_model(state_dict=sd0).to(dev)
    i0, i1 = synth.synthetic_pairs(n, H, W, seed=int(fx["pair_seed"]))
    f0 = model(i0.to(dev), is_training=True)["x_norm_patchtokens"].contiguous()
    f1 = model(i1.to(dev), is_training=True)["x_norm_patchtokens"].contiguous()
    np.testing.assert_allclose(f0.cpu()[:, ::16].numpy(), fx["feat0_rows"], rtol=0, atol=2e-4)
    hw_c = (H // 14, W // 14)
    out = dense_match(f0, f1, hw_c, hw_c, (H, W))
    assert np.array_equal(out["b_ids"].cpu().numpy(), fx["b_ids"])
    assert np.array_equal(out["i_ids"].cpu().numpy(), fx["i_ids"])
    assert np.array_equal(out["j_ids"].cpu().numpy(), fx["j_ids"])
    np.testing.assert_allclose(out["mconf"].cpu().numpy(), fx["mconf"], rtol=0, atol=2e-3)
    assert np.array_equal(out["mkpts0_c"].cpu().numpy(), fx["mkpts0_c"])
    conf = out["conf_matrix"]
    assert np.array_equal(conf.max(2)[1].cpu().numpy()[fx["b_ids"], fx["i_ids"]], fx["j_ids"])
    assert abs(float(conf.double().sum()) - float(fx["conf_sum"][0])) < 1e-2 * float(fx["conf_sum"][0])


def test_matcher_vs_oracle_random_shapes(dev):
    from oracle import coarse_match_ref as cm
    from pope_amd.matcher import dense_match
    g = torch.Generator().manual_seed(5)
    for (n, h0, w0, h1, w1, C) in [(1, 6, 7, 6, 7, 32), (3, 9, 11, 8, 13, 64), (2, 34, 45, 34, 45, 384), (1, 5, 5, 12, 9, 256)]:
        L, S = h0 * w0, h1 * w1
        f0 = torch.randn(n, L, C, generator=g) * 3
        idx = torch.randint(0, L, (n, S), generator=g)
        f1 = torch.gather(f0, 1, idx[..., None].expand(-1, -1, C)) + 0.2 * torch.randn(n, S, C, generator=g)
        want = cm.dense_match(f0, f1, (h0, w0), (h1, w1), (h0 * 8, w0 * 8))
        got = dense_match(f0.to(dev), f1.to(dev), (h0, w0), (h1, w1), (h0 * 8, w0 * 8))
        for k in ("b_ids", "i_ids", "j_ids"):
            assert np.array_equal(got[k].cpu().numpy(), want[k].numpy()), (k, n, h0, w0)
        np.testing.assert_allclose(got["mconf"].cpu().numpy(), want["mconf"].numpy(), rtol=1e-4, atol=1e-6)
        assert np.array_equal(got["mkpts1_c"].cpu().numpy(), want["mkpts1_c"].numpy())


@pytest.mark.parametrize("hw", [(64, 64), (60, 80)])
def test_loftr_shapes_of_the_drivers_at_c256_are_index_exact(dev, hw):
    """The dense matcher at the LoFTR coarse shapes the drivers reach: C = 256 with L = S = 4 096
    (512 x 512 crops, eval_onepose_json.py:88) and L = S = 4 800 (480 x 640, SURVEY a-11), three pairs per launch — 32-bit
    offsets of the similarity epilogue and of the confidence pass at 92 MB per pair, `conf_matrix` published.  Index-exact
    against oracle/coarse_match_ref.py, ties included (pair 2 repeats rows so that exact ties exist)."""
    from oracle import coarse_match_ref as cm
    from pope_amd.matcher import dense_match
    g = torch.Generator().manual_seed(17)
    n, C = 3, 256
    L = hw[0] * hw[1]
    f0 = torch.randn(n, L, C, generator=g) * 1.5
    perm = torch.stack([torch.randperm(L, generator=g) for _ in range(n)])
    f1 = torch.gather(f0, 1, perm[..., None].expand(-1, -1, C)) + 0.15 * torch.randn(n, L, C, generator=g)
    f1[2, 100:140] = f1[2, 200:240]                       # duplicated columns: ties resolve to the lowest index
    want = cm.dense_match(f0, f1, hw, hw, (hw[0] * 8, hw[1] * 8))
    got = dense_match(f0.to(dev), f1.to(dev), hw, hw, (hw[0] * 8, hw[1] * 8), want_conf=True)
    assert len(want["b_ids"]) > 0.8 * n * (hw[0] - 4) * (hw[1] - 4)
    for k in ("b_ids", "i_ids", "j_ids"):
        assert np.array_equal(got[k].cpu().numpy(), want[k].numpy()), k
    np.testing.assert_allclose(got["mconf"].cpu().numpy(), want["mconf"].numpy(), rtol=1e-4, atol=1e-6)
    assert np.array_equal(got["mkpts0_c"].cpu().numpy(), want["mkpts0_c"].numpy())
    assert np.array_equal(got["mkpts1_c"].cpu().numpy(), want["mkpts1_c"].numpy())
    conf = got["conf_matrix"]
    assert conf.shape == (n, L, L)
    assert np.array_equal(conf.max(2)[1].cpu().numpy(), want["conf_matrix"].max(2)[1].numpy())
    np.testing.assert_allclose(conf[1, ::97].cpu().numpy(), want["conf_matrix"][1, ::97].numpy(), rtol=2e-4, atol=1e-9)


def test_border_capacity_ties_and_empty(dev):
    from pope_amd.matcher import dense_match
    g = torch.Generator().manual_seed(1)
    h, w = 9, 11
    f = (torch.randn(1, h * w, 64, generator=g) * 4).to(dev)
    out = dense_match(f, f, (h, w), (h, w), (h * 14, w * 14))
    assert len(out["i_ids"]) == (h - 4) * (w - 4)
    assert bool((out["i_ids"] == out["j_ids"]).all())
    key = out["b_ids"] * 10**6 + out["i_ids"]
    assert bool((key[1:] > key[:-1]).all())
    f0 = (0.01 * torch.randn(2, 36, 32, generator=g)).to(dev)  # flat similarity: conf ~ 1/36^2 << thr
    f1 = (0.01 * torch.randn(2, 36, 32, generator=g)).to(dev)
    out = dense_match(f0, f1, (6, 6), (6, 6), (48, 48))
    assert len(out["i_ids"]) == 0 and out["mkpts0_c"].shape == (0, 2) and list(out["counts"]) == [0, 0]


def test_bench_size_mirror_property(dev, sd0):
    """Size-independent property at the benchmark's matcher shape (1530 x 1530 x 384 per pair, several pairs per launch):
    swapping the two images transposes the confidence matrix and mirrors the match list."""
    from pope_amd.matcher import dense_match
    g = torch.Generator().manual_seed(11)
    n, hw, C = 6, (34, 45), 384
    L = hw[0] * hw[1]
    f0 = torch.randn(n, L, C, generator=g)
    perm = torch.stack([torch.randperm(L, generator=g) for _ in range(n)])
    f1 = torch.gather(f0, 1, perm[..., None].expand(-1, -1, C)) + 0.25 * torch.randn(n, L, C, generator=g)
    f0, f1 = (3.0 * f0).to(dev), (3.0 * f1).to(dev)
    a = dense_match(f0, f1, hw, hw, (476, 630))
    b = dense_match(f1, f0, hw, hw, (476, 630))
    assert len(a["i_ids"]) > 500 * n // 2
    np.testing.assert_allclose(a["conf_matrix"].cpu().numpy(), b["conf_matrix"].transpose(1, 2).cpu().numpy(), rtol=2e-5, atol=1e-7)
    fwd = set(zip(a["b_ids"].tolist(), a["i_ids"].tolist(), a["j_ids"].tolist()))
    bwd = set(zip(b["b_ids"].tolist(), b["j_ids"].tolist(), b["i_ids"].tolist()))
    assert fwd == bwd
    # every reported match is the planted correspondence (interior cells only: border_rm = 2)
    inv = perm.to(dev)
    assert bool((inv[a["b_ids"], a["j_ids"]] == a["i_ids"]).all())
