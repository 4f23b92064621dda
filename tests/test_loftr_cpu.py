"""CPU: the LoFTR `Matcher` stages.  The oracle restatement against the
fixtures produced by the reference's own Matcher (oracle/gen_golden.py), and the product's torch stages
(pope_amd/loftr.py: folded BatchNorm, generated position code, gathered windows) against the same
fixtures.  The HIP coarse-matching stage itself needs a GPU (tests/test_gpu_loftr.py)."""
import copy
import os

import numpy as np
import pytest
import torch

from oracle import loftr_ref
from pope_amd import loftr, synth
from pope_amd.matcher import Matcher, default_cfg

CASES = ["loftr_256", "loftr_256_lowthr", "loftr_192x256_vs_256x192"]
FLOAT_TOL = dict(rtol=2e-4, atol=2e-4)   # folded BN / regrouped sums vs the reference's op order, fp32


def make_inputs(fx):
    n, s0, s1 = int(fx["n"]), tuple(int(v) for v in fx["shape0"]), tuple(int(v) for v in fx["shape1"])
    i0, i1 = synth.synthetic_gray_pairs(n, *s0, seed=21)
    if s1 != s0:
        i1 = synth.synthetic_gray_pairs(n, *s1, seed=22)[0]
        i1[:, :, 32:224, :] = i0[:, :, :, 32:224]
    np.testing.assert_allclose([float(i0.double().sum()), float(i1.double().sum())], fx["image_digest"], rtol=1e-12)
    return i0, i1


def cfg_with_thr(thr):
    cfg = copy.deepcopy(default_cfg)
    cfg["match_coarse"]["thr"] = float(thr)
    return cfg


@pytest.fixture(scope="module")
def msd():
    return synth.synthetic_matcher_state_dict(seed=0)


@pytest.fixture(scope="module")
def model(msd):
    m = Matcher(default_cfg).eval()
    m.load_state_dict({"matcher." + k: v.clone() for k, v in msd.items()}, strict=True)   # matcher.py:81-85
    return m


def test_state_dict_layout(msd):
    m = Matcher(default_cfg)
    keys = list(m.state_dict().keys())
    assert len(keys) == 211 and set(keys) == set(msd)
    assert sum(k.startswith("backbone.") for k in keys) == 107
    assert sum(k.startswith("loftr_coarse.") for k in keys) == 80
    assert sum(k.startswith("fine_preprocess.") for k in keys) == 4
    assert sum(k.startswith("loftr_fine.") for k in keys) == 20
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(msd[k].shape), k
    with pytest.raises(RuntimeError):
        m.load_state_dict({k: v for k, v in msd.items() if k != "backbone.conv1.weight"}, strict=True)
    # the `matcher.` prefix is stripped in the CALLER's dict, as the reference does (src/matcher/matcher.py:81-85)
    prefixed = {"matcher." + k: v for k, v in msd.items()}
    m.load_state_dict(prefixed, strict=True)
    assert set(prefixed) == set(msd)


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_fixture(msd, golden_dir, name):
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    i0, i1 = make_inputs(fx)
    with torch.no_grad():
        out = loftr_ref.matcher_forward(msd, cfg_with_thr(fx["thr"]), i0, i1)
    for k in ("b_ids", "i_ids", "j_ids"):
        assert np.array_equal(out[k].numpy(), fx[k]), k
    assert len(fx["b_ids"]) > 0
    for k in ("mconf", "mkpts0_c", "mkpts1_c", "mkpts0_f", "mkpts1_f", "expec_f"):
        np.testing.assert_allclose(out[k].numpy(), fx[k], rtol=0, atol=1e-6, err_msg=k)
    np.testing.assert_allclose(out["feat_c0"][:, ::8].numpy(), fx["feat_c0"], rtol=0, atol=1e-6)
    conf = out["conf_matrix"]
    assert np.array_equal(conf.max(2)[1].numpy(), fx["conf_rowarg"])
    assert np.array_equal(conf.max(1)[1].numpy(), fx["conf_colarg"])


def test_position_code_reproduces_reference_frequencies():
    """temp_bug_fix=False: `-ln(1e4)/d_model//2` == -1 -> frequencies exp(-2k) (position_encoding.py:28)."""
    for bug_fix in (False, True):
        pe = loftr.PositionEncodingSine(256, temp_bug_fix=bug_fix)
        want = loftr_ref.position_encoding(256, 24, 32, temp_bug_fix=bug_fix)
        got = pe.code(24, 32, torch.device("cpu"))
        assert got.shape == want.shape and torch.equal(got, want)
    x = torch.randn(2, 256, 24, 32)
    assert torch.equal(loftr.PositionEncodingSine(256, temp_bug_fix=False)(x), x + loftr_ref.position_encoding(256, 24, 32))
    assert float(loftr_ref.position_encoding(256, 4, 4)[0, 4, 0, 0]) == pytest.approx(float(np.sin(np.exp(-2.0))), abs=1e-7)
    with pytest.raises(ValueError):
        loftr.PositionEncodingSine(256).code(257, 8, torch.device("cpu"))


@pytest.mark.parametrize("name", ["loftr_256_lowthr", "loftr_192x256_vs_256x192"])
def test_backbone_and_transformer_vs_fixture(model, golden_dir, name):
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    i0, i1 = make_inputs(fx)
    with torch.no_grad():
        if i0.shape == i1.shape:
            bc, bf = model.backbone(torch.cat([i0, i1], 0))
            n = i0.shape[0]
            (c0, c1) = bc.split(n)
        else:
            bc, bf = model.backbone(i0)
            c0, c1 = bc, model.backbone(i1)[0]
        np.testing.assert_allclose(bc[:1, :, ::2, ::2].numpy(), fx["backbone_c"], **FLOAT_TOL)
        np.testing.assert_allclose(bf[:1, ::4, ::8, ::8].numpy(), fx["backbone_f"], **FLOAT_TOL)
        t0 = model.pos_encoding(c0).flatten(2).transpose(1, 2)
        t1 = model.pos_encoding(c1).flatten(2).transpose(1, 2)
        t0, t1 = model.loftr_coarse(t0, t1)
    np.testing.assert_allclose(t0[:, ::8].numpy(), fx["feat_c0"], **FLOAT_TOL)
    np.testing.assert_allclose(t1[:, ::8].numpy(), fx["feat_c1"], **FLOAT_TOL)
    np.testing.assert_allclose(t0[0].numpy(), fx["feat_c0_b0"], **FLOAT_TOL)


def test_backbone_refolds_after_weight_load(model, msd):
    m = Matcher(default_cfg).eval()
    x = synth.synthetic_gray_pairs(1, 64, 64, seed=3)[0]
    with torch.no_grad():
        before = m.backbone(x)[0]
        m.load_state_dict(msd, strict=True)
        after = m.backbone(x)[0]
        want = loftr_ref.resnet_fpn_8_2(msd, x)[0]
    assert not torch.allclose(before, after)
    np.testing.assert_allclose(after.numpy(), want.numpy(), **FLOAT_TOL)
    with pytest.raises(NotImplementedError):
        m.train().backbone(x)


def test_gather_windows_equals_unfold():
    g = torch.Generator().manual_seed(0)
    f = torch.randn(2, 16, 24, 32, generator=g)           # 1/2-res map of a 48x64 image, coarse grid 6x8
    b = torch.tensor([0, 0, 1, 1, 1])
    cells = torch.tensor([0, 47, 7, 40, 19])               # corners (zero padding) and an interior cell
    got = loftr.gather_windows(f, b, cells, 8, 5, 4)
    u = torch.nn.functional.unfold(f, (5, 5), stride=4, padding=2).view(2, 16, 25, 48).permute(0, 3, 2, 1)
    assert torch.equal(got, u[b, cells])


@pytest.mark.parametrize("name", CASES)
def test_fine_stage_vs_fixture(model, msd, golden_dir, name):
    """FinePreprocess -> loftr_fine -> FineMatching on the fixture's coarse matches (the coarse stage
    itself is the HIP kernel, tested on the GPU)."""
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    i0, i1 = make_inputs(fx)
    with torch.no_grad():
        ref = loftr_ref.matcher_forward(msd, cfg_with_thr(fx["thr"]), i0, i1)
        data = {"hw0_i": i0.shape[2:], "hw1_i": i1.shape[2:], "hw0_c": ref["hw0_c"], "hw1_c": ref["hw1_c"],
                "hw0_f": ref["hw0_f"], "hw1_f": tuple(int(v) for v in fx["hw1_f"])}
        for k in ("b_ids", "i_ids", "j_ids", "mconf", "mkpts0_c", "mkpts1_c"):
            data[k] = torch.from_numpy(fx[k])
        w0, w1 = model.fine_preprocess(ref["feat_f0"], ref["feat_f1"], ref["feat_c0"], ref["feat_c1"], data)
        assert w0.shape == (len(fx["b_ids"]), 25, 128) and data["W"] == 5
        w0, w1 = model.loftr_fine(w0, w1)
        model.fine_matching(w0, w1, data)
    np.testing.assert_allclose(data["expec_f"].numpy(), fx["expec_f"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(data["mkpts1_f"].numpy(), fx["mkpts1_f"], rtol=0, atol=1e-4)
    assert np.array_equal(data["mkpts0_f"].numpy(), fx["mkpts0_f"])
    assert np.abs(fx["mkpts1_f"] - fx["mkpts1_c"]).max() > 0.5     # the refinement actually moves points


def test_fine_stage_without_matches(model):
    """M == 0 short-circuits (fine_preprocess.py:33-36, fine_matching.py:33-41)."""
    e = torch.empty(0, dtype=torch.int64)
    data = {"hw0_i": (64, 64), "hw1_i": (64, 64), "hw0_c": (8, 8), "hw1_c": (8, 8), "hw0_f": (32, 32), "hw1_f": (32, 32),
            "b_ids": e, "i_ids": e, "j_ids": e, "mconf": torch.empty(0), "mkpts0_c": torch.empty(0, 2),
            "mkpts1_c": torch.empty(0, 2)}
    w0, w1 = model.fine_preprocess(torch.zeros(1, 128, 32, 32), torch.zeros(1, 128, 32, 32), torch.zeros(1, 64, 256),
                                   torch.zeros(1, 64, 256), data)
    assert w0.shape == (0, 25, 128) and w1.shape == (0, 25, 128)
    model.fine_matching(w0, w1, data)
    assert data["expec_f"].shape == (0, 3) and data["mkpts0_f"] is data["mkpts0_c"] and data["mkpts1_f"] is data["mkpts1_c"]


def test_matcher_refuses_cpu_tensors(model):
    """No CPU fallback: the coarse matcher is a HIP kernel."""
    from pope_amd._lib import PopeHipError
    i0, i1 = synth.synthetic_gray_pairs(1, 64, 64, seed=1)
    with pytest.raises(PopeHipError):
        model({"image0": i0, "image1": i1})
