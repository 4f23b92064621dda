"""GPU: the f16x3 range contract is guarded (pope_hip.h "f16x3 range guard").  Every kernel that converts fp32 values to
f16 pairs reports values that do not fit into a device word; the host reads it at its next synchronisation point and
re-runs the work on the fp32 MFMA (default) or raises.  The breaches below are finite in fp32 — the reference's own
arithmetic — so the re-run must reproduce the precision="f32" model bit for bit."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(hip_lib):
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _model(sd, dev, precision="f16x3", on_overflow="rerun_f32"):
    from pope_amd.dinov2_utils import load_dinov2_model
    m = load_dinov2_model(state_dict=sd).to(dev)
    m.precision, m.on_overflow = precision, on_overflow
    return m


def _images(dev, n=2, seed=5):
    from pope_amd import synth
    return synth.synthetic_images(n, 70, 98, seed=seed).to(dev)


def test_no_false_positive_on_the_fixture_weights(dev, sd0):
    m = _model(sd0, dev)
    out = m(_images(dev), is_training=True)
    assert m.overflow_events == 0 and bool(torch.isfinite(out["x_prenorm"]).all())


# (state-dict edit, POPE_RANGE_* bit expected in the warning text)
BREACHES = {
    "layernorm": (lambda sd: sd.__setitem__("blocks.3.norm1.weight", sd["blocks.3.norm1.weight"] * 1e5), "LayerNorm output"),
    "qkv": (lambda sd: sd.__setitem__("blocks.2.attn.qkv.bias", sd["blocks.2.attn.qkv.bias"] + 9000.0), "q/k/v"),
    "gelu": (lambda sd: sd.__setitem__("blocks.5.mlp.fc1.bias", sd["blocks.5.mlp.fc1.bias"] + 1e4), "MLP hidden"),
}


@pytest.mark.parametrize("kind", sorted(BREACHES))
def test_activation_breach_is_detected_and_rerun_in_f32(dev, sd0, kind):
    from pope_amd._lib import PopeRangeError
    sd = {k: v.clone() for k, v in sd0.items()}
    edit, what = BREACHES[kind]
    edit(sd)
    x = _images(dev)
    want = _model(sd, dev, "f32")(x, is_training=True)
    assert bool(torch.isfinite(want["x_prenorm"]).all())   # fine in fp32: the guard must not give this up
    m = _model(sd, dev)
    with pytest.warns(UserWarning, match=what):
        got = m(x, is_training=True)
    assert m.overflow_events == 1
    for k in ("x_norm_clstoken", "x_norm_patchtokens", "x_prenorm"):
        assert torch.equal(got[k], want[k]), k
    # caller-provided output buffer and taps go through the same re-run
    buf = torch.empty(2, 36, 384, device=dev)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m(x, is_training=True, out_norm=buf)
        taps = m.get_intermediate_layers(x, n=[11], norm=False)
    assert torch.equal(buf[:, 1:], want["x_norm_patchtokens"]) and torch.equal(taps[0], want["x_prenorm"][:, 1:])
    strict = _model(sd, dev, on_overflow="raise")
    with pytest.raises(PopeRangeError, match=what):
        strict(x, is_training=True)


def test_input_breach_patch_embed(dev, sd0):
    m = _model(sd0, dev)
    x = _images(dev)
    x[1, 2, 17, 40] = 1.0e4   # |pixel| * 8 >= 65504
    want = _model(sd0, dev, "f32")(x, is_training=True)
    with pytest.warns(UserWarning, match="patch embed input"):
        got = m(x, is_training=True)
    assert torch.equal(got["x_norm_patchtokens"], want["x_norm_patchtokens"])
    tok = m.prepare_tokens_with_masks(x)   # (the model warns once; every event is counted)
    assert m.overflow_events == 2
    from pope_amd import ops
    ntok = 1 + 5 * 7
    assert torch.equal(tok, ops.patch_embed(x, m.patch_embed.proj.weight.detach(), m._posb(70, 98, ntok), 14, precision="f32"))


def test_weight_breach_runs_f32_or_raises(dev, sd0):
    from pope_amd._lib import PopeRangeError
    sd = {k: v.clone() for k, v in sd0.items()}
    sd["blocks.7.mlp.fc2.weight"][3, 5] = 300.0   # |w| * 256 >= 65504
    x = _images(dev)
    want = _model(sd, dev, "f32")(x, is_training=True)
    m = _model(sd, dev)
    with pytest.warns(UserWarning, match="weight"):
        got = m(x, is_training=True)
    assert torch.equal(got["x_norm_patchtokens"], want["x_norm_patchtokens"])
    with warnings.catch_warnings():
        warnings.simplefilter("error")   # decided once, when the planes would have been built: no further warnings
        m(x, is_training=True)
    with pytest.raises(PopeRangeError, match="weight"):
        _model(sd, dev, on_overflow="raise")(x)


def test_pipeline_defers_the_check_and_reruns_only_flagged_chunks(dev, sd0):
    from pope_amd import synth
    from pope_amd.pipeline import PairPipeline
    i0, i1 = synth.synthetic_pairs(4, 70, 98, seed=9)
    i0, i1 = i0.to(dev), i1.to(dev)
    i1[2, 0, 30, 30] = -2.0e4   # chunk 1 of the second image set
    f32 = _model(sd0, dev, "f32")
    ref_pipe = PairPipeline(f32, chunk=2)
    want = ref_pipe(i0, i1)
    m = _model(sd0, dev)
    pipe = PairPipeline(m, chunk=2)
    with pytest.warns(UserWarning, match="patch embed input"):
        got = pipe(i0, i1)
    assert pipe.reruns == 1 and m.overflow_events == 1
    assert torch.equal(got["feat1"][2:4], want["feat1"][2:4])           # the flagged chunk: fp32 arithmetic
    clean = PairPipeline(_model(sd0, dev), chunk=2).extract(i0)
    assert torch.equal(got["feat0"], clean[:, 1:])                      # the others: untouched f16x3 results
    # the match lists were recomputed from the repaired descriptors
    from pope_amd.matcher import dense_match
    exp = dense_match(got["feat0"], got["feat1"], (5, 7), (5, 7), (70, 98))
    for k in ("b_ids", "i_ids", "j_ids", "mconf"):
        assert torch.equal(got[k], exp[k]), k
    assert np.array_equal(got["counts"].numpy(), exp["counts"].numpy())
    from pope_amd._lib import PopeRangeError
    with pytest.raises(PopeRangeError):
        PairPipeline(_model(sd0, dev, on_overflow="raise"), chunk=2)(i0, i1)


def test_dense_match_feature_breach(dev):
    from pope_amd._lib import PopeRangeError
    from pope_amd.matcher import dense_match
    g = torch.Generator().manual_seed(3)
    f0 = torch.randn(2, 7 * 9, 64, generator=g) * 3
    f1 = f0 + 0.1 * torch.randn(f0.shape, generator=g)
    f0[1, 11, 7] = 6.0e3   # |f| / sqrt(64) * 256 >= 65504
    f0, f1 = f0.to(dev), f1.to(dev)
    want = dense_match(f0, f1, (7, 9), (7, 9), (56, 72), precision="f32")
    with pytest.warns(UserWarning, match="matcher features"):
        got = dense_match(f0, f1, (7, 9), (7, 9), (56, 72), precision="f16x3")
    for k in ("b_ids", "i_ids", "j_ids", "mconf", "conf_matrix"):
        assert torch.equal(got[k], want[k]), k
    with pytest.raises(PopeRangeError):
        dense_match(f0, f1, (7, 9), (7, 9), (56, 72), precision="f16x3", on_overflow="raise")


def test_op_level_flags(dev):
    from pope_amd import _lib, ops
    a = torch.randn(70, 64, device=dev)
    w = torch.randn(32, 64, device=dev) * 0.1
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.linear(a, w, precision="f16x3", range_flag=flag)
    assert int(flag) == 0
    a[3, 3] = 9000.0
    ops.linear(a, w, precision="f16x3", range_flag=flag)
    assert int(flag) == 32
    flag.zero_()
    qkv = torch.randn(1, 40, 3 * 64, device=dev)
    ops.attention(qkv, 1, precision="f16x3", range_flag=flag)
    assert int(flag) == 0
    qkv[0, 5, 70] = float("inf")
    ops.attention(qkv, 1, precision="f16x3", range_flag=flag)
    assert int(flag) == 32
    assert "op-level operand" in _lib.describe_range_bits(32)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_second_device_while_first_is_current(hip_lib, sd0):
    """The reference keeps the matcher on cuda:1 with cuda:0 current (pope_model_api.py:181-184): launches, the
    per-device LDS opt-in and the persistent grid size must follow the operands' device."""
    from pope_amd.matcher import dense_match
    torch.cuda.set_device(0)
    x = _images(torch.device("cuda:0"))
    a = _model(sd0, torch.device("cuda:0"))(x, is_training=True)["x_norm_patchtokens"]
    b = _model(sd0, torch.device("cuda:1"))(x.to("cuda:1"), is_training=True)["x_norm_patchtokens"]
    assert b.device.index == 1 and torch.equal(a.cpu(), b.cpu())
    m0 = dense_match(a, a.roll(1, 1), (5, 7), (5, 7), (70, 98))
    m1 = dense_match(b, b.roll(1, 1), (5, 7), (5, 7), (70, 98))
    assert np.array_equal(m0["j_ids"].cpu().numpy(), m1["j_ids"].cpu().numpy())
