"""CPU: the LoFTR `Matcher`.  The oracle restatement against the fixtures produced by the
reference's own Matcher (oracle/gen_golden.py); the product's host side — checkpoint layout, the generated position code,
the folded-BatchNorm weight matrices handed to the HIP kernels, the M = 0 short-circuits — and its refusal to compute
anything on the CPU.  The HIP stages themselves are held to the oracle on the GPU (tests/test_gpu_loftr.py)."""
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


def test_oracle_matches_the_512_reference_fixture(golden_dir):
    """The oracle at the OnePose drivers' 512 x 512 under the `peaked` weights reproduces the reference Matcher's fixture
    (3 420 matches at the default threshold): the checker the GPU tests of the large shapes lean on is pinned there too."""
    fx = np.load(os.path.join(golden_dir, "loftr_512_peaked.npz"))
    sd = synth.peaked_matcher_state_dict(torch.from_numpy(fx["outconv_mean"]), seed=0)
    sd.pop("_calibration_mean")
    i0, i1 = synth.synthetic_gray_pairs(1, 512, 512, seed=23)
    with torch.no_grad():
        out = loftr_ref.matcher_forward(sd, default_cfg, i0, i1)
    assert len(fx["b_ids"]) > 3000
    for k in ("b_ids", "i_ids", "j_ids"):
        assert np.array_equal(out[k].numpy(), fx[k]), k
    for k in ("mconf", "mkpts1_f", "expec_f"):
        np.testing.assert_allclose(out[k].numpy(), fx[k], rtol=0, atol=2e-6, err_msg=k)
    np.testing.assert_allclose(out["feat_c0"][:, ::16].numpy(), fx["feat_c0"], rtol=0, atol=2e-6)
    assert np.array_equal(out["conf_matrix"].max(2)[1].numpy(), fx["conf_rowarg"])


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


def test_folded_conv_matrices_reproduce_the_oracle_cnn(msd):
    """The 22 [Cout, taps x Cin] matrices + folded biases the HIP CNN consumes (pope_hip.h order), applied as plain fp64
    convolutions on the CPU, reproduce oracle/loftr_ref.py:resnet_fpn_8_2 (conv and BatchNorm kept apart): the folding, the
    tap-major layout and the 196 -> 224 channel padding are right; an in-place weight edit refreshes them."""
    bb = loftr.build_backbone(default_cfg).eval()
    bb.load_state_dict({k[len("backbone."):]: v for k, v in msd.items() if k.startswith("backbone.")}, strict=True)
    mats, biases = bb._matrices()
    assert len(mats) == 22 and [m.shape[1] for m in mats[:3]] == [64, 9 * 128, 9 * 128] and mats[6].shape == (196, 9 * 224)
    assert mats[7].shape == (196, 128) and mats[15].shape == (256, 256) and mats[21].shape == (128, 9 * 224)
    assert [b is None for b in biases] == [i in (15, 16, 18, 19, 21) for i in range(22)]

    def conv(x, i, cin, k, stride, pad):
        co = mats[i].shape[0]
        if cin == 1:
            w = mats[i][:, :49].reshape(co, 1, 7, 7)
        else:
            cp = (cin + 31) // 32 * 32
            w = mats[i].reshape(co, k, k, cp)[:, :, :, :cin].permute(0, 3, 1, 2)
            assert float(mats[i].reshape(co, k, k, cp)[:, :, :, cin:].abs().max() if cp > cin else 0.0) == 0.0
        return torch.nn.functional.conv2d(x, w.double(), None if biases[i] is None else biases[i].double(), stride, pad)

    relu = torch.relu
    x = synth.synthetic_gray_pairs(1, 64, 96, seed=3)[0].double()
    h = relu(conv(x, 0, 1, 7, 2, 3))
    h = relu(h + conv(relu(conv(h, 1, 128, 3, 1, 1)), 2, 128, 3, 1, 1))
    x1 = relu(h + conv(relu(conv(h, 3, 128, 3, 1, 1)), 4, 128, 3, 1, 1))
    h = relu(conv(x1, 7, 128, 1, 2, 0) + conv(relu(conv(x1, 5, 128, 3, 2, 1)), 6, 196, 3, 1, 1))
    x2 = relu(h + conv(relu(conv(h, 8, 196, 3, 1, 1)), 9, 196, 3, 1, 1))
    h = relu(conv(x2, 12, 196, 1, 2, 0) + conv(relu(conv(x2, 10, 196, 3, 2, 1)), 11, 256, 3, 1, 1))
    x3 = relu(h + conv(relu(conv(h, 13, 256, 3, 1, 1)), 14, 256, 3, 1, 1))
    up = lambda t: torch.nn.functional.interpolate(t, scale_factor=2.0, mode="bilinear", align_corners=True)  # noqa: E731
    lrelu = lambda t: torch.nn.functional.leaky_relu(t, 0.01)  # noqa: E731
    x3_out = conv(x3, 15, 256, 1, 1, 0)
    x2_out = conv(lrelu(conv(conv(x2, 16, 196, 1, 1, 0) + up(x3_out), 17, 256, 3, 1, 1)), 18, 256, 3, 1, 1)
    x1_out = conv(lrelu(conv(conv(x1, 19, 128, 1, 1, 0) + up(x2_out), 20, 196, 3, 1, 1)), 21, 196, 3, 1, 1)
    with torch.no_grad():
        wc, wf = loftr_ref.resnet_fpn_8_2({k: v.double() for k, v in msd.items()}, x)
    np.testing.assert_allclose(x3_out.numpy(), wc.numpy(), rtol=0, atol=1e-5)      # fp32 folding vs fp64 reference arithmetic
    np.testing.assert_allclose(x1_out.numpy(), wf.numpy(), rtol=0, atol=1e-5)
    # an in-place edit of a BatchNorm statistic changes the cache key (storage address + version counter)
    k0 = _key(bb)
    with torch.no_grad():
        bb.bn1.running_var.mul_(2.0)
    assert _key(bb) != k0
    with pytest.raises(NotImplementedError):
        bb.train()(torch.zeros(1, 1, 16, 16))


def _key(module):
    from pope_amd import _lib
    return _lib.slots_key(_lib.param_slots(module, buffers=True))


def test_replaced_parameter_objects_refresh_the_derived_weights(msd):
    """the derived data (folded BatchNorm matrices, weight planes) is keyed on the tensors CURRENTLY in the
    modules' parameter / buffer slots, so `load_state_dict(..., assign=True)`, `m.weight = nn.Parameter(..)` and a
    re-assigned BatchNorm buffer refresh it exactly like an in-place edit does (round 3 cached the tensor LIST and kept
    serving the planes of the old objects)."""
    import torch.nn as nn
    bb = loftr.build_backbone(default_cfg).eval()
    sd = {k[len("backbone."):]: v for k, v in msd.items() if k.startswith("backbone.")}
    bb.load_state_dict(sd, strict=True)
    bb._weights("f32")
    first, stem0 = bb._hip, bb._hip["mats"][0].clone()
    sd2 = {k: (v * 1.5 if k == "conv1.weight" else v.clone()) for k, v in sd.items()}
    bb.load_state_dict(sd2, strict=True, assign=True)                 # every Parameter / buffer OBJECT is replaced
    bb._weights("f32")
    assert bb._hip is not first and torch.allclose(bb._hip["mats"][0], 1.5 * stem0, rtol=1e-6, atol=0)
    second = bb._hip
    bb.conv1.weight = nn.Parameter(sd["conv1.weight"].clone() * 2.0)  # one Parameter replaced by assignment
    bb._weights("f32")
    assert bb._hip is not second and torch.allclose(bb._hip["mats"][0], 2.0 * stem0, rtol=1e-6, atol=0)
    third = bb._hip
    bb.bn1.running_var = sd["bn1.running_var"] * 4.0                  # a buffer re-assigned: the folded scale halves
    bb._weights("f32")
    assert bb._hip is not third and not torch.equal(bb._hip["mats"][0], third["mats"][0])
    bb._weights("f32")
    assert bb._hip["key"] == _key(bb)                                  # and an unchanged model hits the cache
    # a transformer layer and the fine-stage projections follow the same rule
    t = loftr.LocalFeatureTransformer(default_cfg["coarse"]).eval()
    layer = t.layers[1]
    layer._weights("f32")
    h = layer._hip
    layer.merge.weight = nn.Parameter(layer.merge.weight.detach() * 0.5)
    layer._weights("f32")
    assert layer._hip is not h and torch.equal(layer._hip["mats"][2], layer.merge.weight.detach())


def test_transformer_and_fine_weight_matrices(msd):
    t = loftr.LocalFeatureTransformer(default_cfg["coarse"]).eval()
    t.load_state_dict({k[len("loftr_coarse."):]: v for k, v in msd.items() if k.startswith("loftr_coarse.")}, strict=True)
    layer = t.layers[3]
    w = layer._weights("f32")
    assert w is not None and layer._hip["fit"]
    mats = layer._hip["mats"]
    assert [tuple(m.shape) for m in mats] == [(256, 256), (512, 256), (256, 256), (512, 512), (256, 512)]
    assert torch.equal(mats[1][:256], msd["loftr_coarse.layers.3.k_proj.weight"]) and torch.equal(mats[1][256:], msd["loftr_coarse.layers.3.v_proj.weight"])
    with torch.no_grad():
        layer.merge.weight[0, 0] = 1.0e3          # |w| * 256 >= 65504: this layer must take the fp32 path
    layer._weights("f32")
    assert not layer._hip["fit"] and layer._weights("f16x3") is None
    with pytest.raises(NotImplementedError):
        loftr.LocalFeatureTransformer(dict(default_cfg["coarse"], nhead=4))
    with pytest.raises(KeyError):
        loftr.LocalFeatureTransformer(dict(default_cfg["coarse"], layer_names=["self", "other"]))


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


def test_no_stage_computes_on_the_cpu(model):
    """No CPU fallback anywhere in the Matcher: every stage is a HIP call and says so when handed CPU tensors."""
    from pope_amd._lib import PopeHipError
    i0, i1 = synth.synthetic_gray_pairs(1, 64, 64, seed=1)
    with pytest.raises(PopeHipError):
        model({"image0": i0, "image1": i1})
    with pytest.raises(PopeHipError):
        model.backbone(i0)
    with pytest.raises(PopeHipError):
        model.loftr_coarse(torch.zeros(1, 64, 256), torch.zeros(1, 64, 256))
    with pytest.raises(PopeHipError):
        model.loftr_coarse.layers[0](torch.zeros(1, 64, 256), torch.zeros(1, 64, 256))
    one = torch.zeros(1, dtype=torch.int64)
    data = {"hw0_i": (64, 64), "hw1_i": (64, 64), "hw0_c": (8, 8), "hw1_c": (8, 8), "hw0_f": (32, 32), "hw1_f": (32, 32),
            "b_ids": one, "i_ids": one, "j_ids": one, "mconf": torch.ones(1), "mkpts0_c": torch.zeros(1, 2), "mkpts1_c": torch.zeros(1, 2)}
    with pytest.raises(PopeHipError):
        model.fine_preprocess(torch.zeros(1, 128, 32, 32), torch.zeros(1, 128, 32, 32), torch.zeros(1, 64, 256), torch.zeros(1, 64, 256), data)
    with pytest.raises(PopeHipError):
        model.fine_matching(torch.zeros(1, 25, 128), torch.zeros(1, 25, 128), data)


def test_peaked_weights_give_a_trained_like_match_load(golden_dir):
    """`synth.peaked_matcher_state_dict` (calibrated here through the oracle's CNN): hundreds of confident matches per 256 x
    256 pair at the default threshold — the load the realistic legs of bench_legs.py and the drivers' tests run under — where
    the plain random weights publish a dozen.  The 512 x 512 reference fixture stores the calibration mean; the mean alone
    reproduces its weights."""
    import os
    def pre(sd_, img):
        with torch.no_grad():
            return loftr_ref.resnet_fpn_8_2(sd_, img)[0]
    sd = synth.peaked_matcher_state_dict(pre, seed=0)
    mean = sd.pop("_calibration_mean")
    i0, i1 = synth.synthetic_gray_pairs(1, 256, 256, seed=5)
    with torch.no_grad():
        out = loftr_ref.matcher_forward(sd, default_cfg, i0, i1)
        plain = loftr_ref.matcher_forward(synth.synthetic_matcher_state_dict(seed=0), default_cfg, i0, i1)
    assert len(out["mconf"]) >= 600 and int((out["mconf"] > 0.9).sum()) >= 550 and len(plain["mconf"]) < 60
    # matched cells are the planted shift (8, 16) px = (1, 2) cells
    i, j = out["i_ids"], out["j_ids"]
    assert bool((((i // 32 + 1) % 32) * 32 + (i % 32 + 2) % 32 == j).float().mean() > 0.98)
    fx = np.load(os.path.join(golden_dir, "loftr_512_peaked.npz"))
    assert float((mean - torch.from_numpy(fx["outconv_mean"])).abs().max()) < 1e-4 * float(mean.abs().max())
