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
    return _lib.params_key(list(module.parameters()) + list(module.buffers()))


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
