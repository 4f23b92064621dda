"""GPU: the binding stubs printed in INTEGRATION.md are executed as written (the python blocks that define `_lib`, `hip_linear`,
`hip_attention` and `hip_estimate_pose`) and held to the product's own bindings — documentation that cannot rot."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(hip_lib):
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _stub_namespace():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    binding = [b for b in blocks if "ctypes.CDLL" in b or "def hip_estimate_pose" in b]
    assert len(binding) == 2, "INTEGRATION.md no longer holds the two binding blocks"
    ns = {}
    cwd = os.getcwd()
    os.chdir(ROOT)      # the stub opens the library by its path relative to the repository root
    try:
        for b in binding:
            exec(compile(b, "INTEGRATION.md", "exec"), ns)
    finally:
        os.chdir(cwd)
    return ns


def test_linear_and_attention_stubs(dev):
    from pope_amd import ops
    ns = _stub_namespace()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 197, 384, generator=g).to(dev)
    w, b = (torch.randn(1152, 384, generator=g) * 0.05).to(dev), torch.randn(1152, generator=g).to(dev)
    qkv = ns["hip_linear"](x, w, b)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    assert float((qkv.double() - ref).abs().max()) < 2e-4
    out = ns["hip_attention"](qkv, 6)
    assert torch.equal(out, ops.attention(qkv, 6, precision="f32"))


def test_pose_stub(dev):
    from pope_amd import pose, synth
    ns = _stub_namespace()
    scenes = [synth.synthetic_pose_scene(n, 40 + i, outlier=0.3, noise=0.2) for i, n in enumerate((60, 7, 130))]
    k0 = torch.from_numpy(np.concatenate([s[0] for s in scenes])).to(dev)
    k1 = torch.from_numpy(np.concatenate([s[1] for s in scenes])).to(dev)
    bids = torch.cat([torch.full((len(s[0]),), i, dtype=torch.int64) for i, s in enumerate(scenes)]).to(dev)
    K0 = torch.from_numpy(np.stack([s[2] for s in scenes])).to(dev)
    K1 = torch.from_numpy(np.stack([s[3] for s in scenes])).to(dev)
    R, t, inl, n_inl = ns["hip_estimate_pose"](k0, k1, bids, 3, K0, K1)
    want = pose.estimate_pose_batch(k0, k1, torch.tensor([len(s[0]) for s in scenes], dtype=torch.int32), K0, K1, 0.5, 0.99)
    assert torch.equal(R, want["R"]) and torch.equal(t, want["t"]) and torch.equal(inl, want["inliers"])
    assert torch.equal(n_inl, want["n_inliers"]) and int(n_inl.min()) > 0
