"""CPU: the checkpoint-FILE path of `load_dinov2_model`: dinov2/dinov2/utils/utils.py:21-34 takes the
`student` entry when the file has one, strips the `module.` / `backbone.` prefixes of a training checkpoint and loads with
strict=True; segment_anything/segment_anything/dinov2_utils.py:38-47 returns the model in eval mode on the CPU.  The real
weights/dinov2_vits14.pth cannot be fetched offline, so the files are written from the synthetic 175-key state dict."""
import pytest
import torch

from pope_amd import synth
from pope_amd.dinov2_utils import load_dinov2_model, load_dinov2_weights
from pope_amd.dinov2 import build_vits14


def _same(model, sd):
    got = model.state_dict()
    return set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)


def test_training_checkpoint_layout(tmp_path, sd0):
    """{"student": {"module.backbone.<key>": ...}, "teacher": {...}}: the student entry, both prefixes stripped."""
    path = tmp_path / "ckpt_student.pth"
    other = synth.synthetic_state_dict(seed=1)
    torch.save({"student": {"module.backbone." + k: v for k, v in sd0.items()},
                "teacher": {"backbone." + k: v for k, v in other.items()}, "epoch": 3}, path)
    m = load_dinov2_model(weights=str(path))
    assert len(sd0) == 175 and _same(m, sd0) and not m.training
    assert next(m.parameters()).device.type == "cpu"       # the caller moves it (eval_linemod_json.py:11-12)


def test_plain_state_dict_layout(tmp_path, sd0):
    """The released backbone file is the bare state dict (no `student` key): used as it is (utils.py:26-28)."""
    path = tmp_path / "plain.pth"
    torch.save(dict(sd0), path)
    assert _same(load_dinov2_model(weights=str(path)), sd0)
    path2 = tmp_path / "backbone_prefix.pth"
    torch.save({"backbone." + k: v for k, v in sd0.items()}, path2)
    assert _same(load_dinov2_model(weights=str(path2)), sd0)


def test_checkpoint_key_none_and_other_key(tmp_path, sd0):
    path = tmp_path / "teacher.pth"
    torch.save({"teacher": {"backbone." + k: v for k, v in sd0.items()}}, path)
    m = build_vits14()
    res = load_dinov2_weights(m, str(path), checkpoint_key="teacher")
    assert not res.missing_keys and not res.unexpected_keys and _same(m, sd0)
    with pytest.raises(RuntimeError):        # checkpoint_key=None: the wrapper dict itself is not a state dict (strict)
        load_dinov2_weights(build_vits14(), str(path), checkpoint_key=None)


def test_strict_loading_rejects_incomplete_or_foreign_files(tmp_path, sd0):
    short = {k: v for k, v in sd0.items() if k != "blocks.7.ls2.gamma"}
    p1 = tmp_path / "missing.pth"
    torch.save({"student": short}, p1)
    with pytest.raises(RuntimeError, match="blocks.7.ls2.gamma"):
        load_dinov2_model(weights=str(p1))
    extra = dict(sd0, **{"dino_head.mlp.0.weight": torch.zeros(4, 4)})   # a full SSL checkpoint carries the heads too
    p2 = tmp_path / "extra.pth"
    torch.save({"student": extra}, p2)
    with pytest.raises(RuntimeError, match="dino_head"):
        load_dinov2_model(weights=str(p2))
    with pytest.raises(FileNotFoundError):
        load_dinov2_model(weights=str(tmp_path / "absent.pth"))


def test_vit_weight_cache_follows_replaced_parameter_objects():
    """`load_state_dict(..., assign=True)` / `m.weight = nn.Parameter(..)` swap the Parameter OBJECTS; the ViT's
    derived weight structs are keyed on whatever sits in the parameter slots now (`_lib.slots_key`), not on a cached tensor
    list."""
    import torch.nn as nn
    from pope_amd import synth
    from pope_amd.dinov2_utils import load_dinov2_model
    sd = synth.synthetic_state_dict(seed=0)
    m = load_dinov2_model(state_dict=sd)
    w0 = m._weights("f32")
    assert m._weights("f32") is w0                                   # unchanged model: cache hit
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["blocks.3.mlp.fc1.weight"] = sd2["blocks.3.mlp.fc1.weight"] * 1.25
    m.load_state_dict(sd2, strict=True, assign=True)
    w1 = m._weights("f32")
    assert w1 is not w0
    m.blocks[7].attn.proj.weight = nn.Parameter(m.blocks[7].attn.proj.weight.detach() * 0.5)
    assert m._weights("f32") is not w1
