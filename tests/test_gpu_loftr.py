"""GPU: the drop-in LoFTR `Matcher` (src/matcher/matcher.py:29-79) end to end on the card — HIP coarse
matcher inside PyTorch-ROCm CNN/transformer plumbing — against fixtures produced by the reference's own
Matcher.  Index parity is bit-exact wherever the coarse features are the fixture's; end to end the features
come from MIOpen/rocBLAS instead of the CPU kernels, so floats carry a tolerance and a match may only
differ where the reference's own confidence is within that tolerance of the threshold or of a tie."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FEAT_TOL = dict(rtol=2e-3, atol=2e-3)


@pytest.fixture(scope="module")
def dev(hip_lib):
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def build(thr, dev):
    from pope_amd import synth
    from pope_amd.matcher import Matcher, default_cfg
    cfg = copy.deepcopy(default_cfg)
    cfg["match_coarse"]["thr"] = float(thr)
    m = Matcher(cfg).eval()
    m.load_state_dict(synth.synthetic_matcher_state_dict(seed=0), strict=True)
    return m.to(dev)


def inputs(fx, dev):
    from pope_amd import synth
    n, s0, s1 = int(fx["n"]), tuple(int(v) for v in fx["shape0"]), tuple(int(v) for v in fx["shape1"])
    i0, i1 = synth.synthetic_gray_pairs(n, *s0, seed=21)
    if s1 != s0:
        i1 = synth.synthetic_gray_pairs(n, *s1, seed=22)[0]
        i1[:, :, 32:224, :] = i0[:, :, :, 32:224]
    return i0.to(dev), i1.to(dev)


@pytest.mark.parametrize("name", ["loftr_256_lowthr", "loftr_192x256_vs_256x192"])
def test_coarse_stage_on_fixture_features_is_index_exact(dev, golden_dir, name):
    """HIP CoarseMatching on the reference's own coarse features: identical (b, i, j), ordering included."""
    from pope_amd.matcher import CoarseMatching, default_cfg
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = dict(default_cfg["match_coarse"], thr=float(fx["thr"]))
    data = {"hw0_i": tuple(fx["shape0"]), "hw1_i": tuple(fx["shape1"]), "hw0_c": tuple(fx["hw0_c"]), "hw1_c": tuple(fx["hw1_c"])}
    f0, f1 = torch.from_numpy(fx["feat_c0_b0"])[None].to(dev), torch.from_numpy(fx["feat_c1_b0"])[None].to(dev)
    CoarseMatching(cfg).eval()(f0, f1, data)
    sel = fx["b_ids"] == 0
    assert sel.sum() > 10
    assert np.array_equal(data["i_ids"].cpu().numpy(), fx["i_ids"][sel])
    assert np.array_equal(data["j_ids"].cpu().numpy(), fx["j_ids"][sel])
    np.testing.assert_allclose(data["mconf"].cpu().numpy(), fx["mconf"][sel], rtol=1e-4, atol=1e-7)
    assert np.array_equal(data["mkpts0_c"].cpu().numpy(), fx["mkpts0_c"][sel])
    assert np.array_equal(data["mkpts1_c"].cpu().numpy(), fx["mkpts1_c"][sel])
    conf = data["conf_matrix"][0]
    assert np.array_equal(conf.max(1)[1].cpu().numpy(), fx["conf_rowarg"][0])
    assert np.array_equal(conf.max(0)[1].cpu().numpy(), fx["conf_colarg"][0])


@pytest.mark.parametrize("name", ["loftr_256", "loftr_256_lowthr", "loftr_192x256_vs_256x192"])
def test_matcher_end_to_end(dev, golden_dir, name):
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    m = build(fx["thr"], dev)
    i0, i1 = inputs(fx, dev)
    data = {"image0": i0, "image1": i1}
    assert m(data) is None                         # in-place protocol (matcher.py:29)
    for k in ("bs", "hw0_i", "hw1_i", "hw0_c", "hw1_c", "hw0_f", "hw1_f", "conf_matrix", "b_ids", "i_ids", "j_ids",
              "gt_mask", "m_bids", "mkpts0_c", "mkpts1_c", "mconf", "W", "expec_f", "mkpts0_f", "mkpts1_f"):
        assert k in data, k
    assert tuple(data["hw0_c"]) == tuple(fx["hw0_c"]) and tuple(data["hw1_f"]) == tuple(fx["hw1_f"]) and data["W"] == 5
    thr = float(fx["thr"])
    # matches whose reference confidence is clear of the threshold must be reproduced exactly
    ref = {(int(b), int(i)): (int(j), float(c), k) for k, (b, i, j, c) in
           enumerate(zip(fx["b_ids"], fx["i_ids"], fx["j_ids"], fx["mconf"]))}
    got = {(int(b), int(i)): (int(j), float(c), k) for k, (b, i, j, c) in
           enumerate(zip(data["b_ids"].cpu().numpy(), data["i_ids"].cpu().numpy(), data["j_ids"].cpu().numpy(),
                         data["mconf"].cpu().numpy()))}
    margin = 0.05 * thr + 1e-4
    for key, (j, c, _) in ref.items():
        if c > thr + margin:
            assert key in got and got[key][0] == j, key
    for key, (j, c, _) in got.items():
        assert key in ref or c < thr + margin, key
    common = sorted(set(ref) & set(got))
    assert len(common) >= 0.9 * len(ref) > 0
    ri, gi = [ref[k][2] for k in common], [got[k][2] for k in common]
    np.testing.assert_allclose(data["mconf"].cpu().numpy()[gi], fx["mconf"][ri], rtol=3e-2, atol=1e-4)
    assert np.array_equal(data["mkpts0_c"].cpu().numpy()[gi], fx["mkpts0_c"][ri])
    assert np.array_equal(data["mkpts0_f"].cpu().numpy()[gi], fx["mkpts0_f"][ri])
    np.testing.assert_allclose(data["mkpts1_f"].cpu().numpy()[gi], fx["mkpts1_f"][ri], rtol=0, atol=2e-2)   # pixels
    np.testing.assert_allclose(data["expec_f"].cpu().numpy()[gi], fx["expec_f"][ri], rtol=0, atol=5e-3)
    # ordering: (b, i) ascending like torch.where (coarse_matching.py:193)
    order = data["b_ids"].cpu().numpy().astype(np.int64) * 10 ** 6 + data["i_ids"].cpu().numpy()
    assert np.all(np.diff(order) > 0)


def test_only_att_fea_and_feature_parity(dev, golden_dir):
    fx = np.load(os.path.join(golden_dir, "loftr_256_lowthr.npz"))
    m = build(fx["thr"], dev)
    i0, i1 = inputs(fx, dev)
    f0, f1 = m({"image0": i0, "image1": i1}, only_att_fea=True)     # matcher.py:67-68
    assert f0.shape == (2, 1024, 256)
    np.testing.assert_allclose(f0[:, ::8].cpu().numpy(), fx["feat_c0"], **FEAT_TOL)
    np.testing.assert_allclose(f1[:, ::8].cpu().numpy(), fx["feat_c1"], **FEAT_TOL)
    bc, bf = m.backbone(torch.cat([i0, i1], 0))
    np.testing.assert_allclose(bc[:1, :, ::2, ::2].cpu().numpy(), fx["backbone_c"], **FEAT_TOL)
    np.testing.assert_allclose(bf[:1, ::4, ::8, ::8].cpu().numpy(), fx["backbone_f"], **FEAT_TOL)


def test_graph_replay_is_bit_identical_to_eager_launches(dev, golden_dir):
    """The Matcher's shape-static front end (CNN, position code, coarse transformer) replayed from a captured HIP graph
    (call 1 eager, call 2 captures, calls 3+ replay) publishes exactly what the eager launches publish — on the
    captured input and on new inputs of the same shape — and a load_state_dict drops the graphs."""
    fx = np.load(os.path.join(golden_dir, "loftr_256_lowthr.npz"))
    m, e = build(fx["thr"], dev), build(fx["thr"], dev)
    m.use_graph, e.use_graph = True, False
    i0, i1 = inputs(fx, dev)
    keys = ("conf_matrix", "b_ids", "i_ids", "j_ids", "mconf", "mkpts0_c", "mkpts1_c", "expec_f", "mkpts0_f", "mkpts1_f")

    def run(model, a, b):
        d = {"image0": a, "image1": b}
        model(d)
        return d

    want = run(e, i0, i1)
    for call in range(4):
        got = run(m, i0, i1)
        for k in keys:
            assert torch.equal(got[k], want[k]), (call, k)
    ents = [v for v in m._graphs.values() if "graph" in v]
    assert len(ents) == 1 and not ents[0].get("retired") and len(ents[0]["flags"]) >= 2
    j0, j1 = i1.flip(0).contiguous(), i0.flip(0).contiguous()        # new pixels, same shapes: replay with fresh inputs
    want2, got2 = run(e, j0, j1), run(m, j0, j1)
    assert len(want2["b_ids"]) > 0
    for k in keys:
        assert torch.equal(got2[k], want2[k]), k
    f0, f1 = m({"image0": i0, "image1": i1}, only_att_fea=True)     # copies, not the graph's static buffers
    g0, _ = m({"image0": j0, "image1": j1}, only_att_fea=True)
    assert not torch.equal(f0, g0) and torch.equal(f0, e({"image0": i0, "image1": i1}, only_att_fea=True)[0])
    # a range-guard event inside the replayed launches retires the graph and the eager path (warning, fp32 re-run) takes over
    with pytest.warns(UserWarning, match="range contract"):
        big = run(m, i0 * 3.0e4, i1)
    assert any(v.get("retired") for v in m._graphs.values()) and torch.isfinite(big["conf_matrix"]).all()
    from pope_amd import synth
    m.load_state_dict(synth.synthetic_matcher_state_dict(seed=0), strict=True)
    assert m._graphs == {}


def test_no_coarse_match_short_circuit(dev):
    """thr above every confidence: M = 0, fine stage skipped, *_f alias *_c (fine_matching.py:33-41)."""
    from pope_amd import synth
    m = build(0.999999, dev)
    i0, _ = synth.synthetic_gray_pairs(1, 64, 96, seed=2)
    i1, _ = synth.synthetic_gray_pairs(1, 64, 96, seed=3)
    data = {"image0": i0.to(dev), "image1": i1.to(dev)}
    m(data)
    assert data["b_ids"].numel() == 0 and data["expec_f"].shape == (0, 3)
    assert data["mkpts0_f"].shape == (0, 2) and data["mkpts1_f"].shape == (0, 2)
    assert data["conf_matrix"].shape == (1, 96, 96)


# ---- the HIP LoFTR encoder layer (first slice) -----------------------------------------------------

def _transformer(kind, dev, seed=0):
    from pope_amd import synth
    from pope_amd.loftr import LocalFeatureTransformer
    from pope_amd.matcher import default_cfg
    t = LocalFeatureTransformer(default_cfg[kind]).eval()
    prefix = "loftr_coarse." if kind == "coarse" else "loftr_fine."
    sd = synth.synthetic_matcher_state_dict(seed=seed)
    t.load_state_dict({k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}, strict=True)
    return t.to(dev)


@pytest.mark.parametrize("kind,n,L,S", [("coarse", 2, 1024, 1024), ("coarse", 3, 768, 1024), ("coarse", 1, 4800, 4800),
                                        ("fine", 153, 25, 25), ("fine", 1, 25, 25)])
def test_hip_encoder_matches_fp64_restatement(dev, kind, n, L, S):
    """LocalFeatureTransformer on the HIP layer (f16x3 planes GEMMs + O(L) linear attention + fused LayerNorms) against
    the same module evaluated in float64 on the CPU (transformer.py:85-106 semantics: 'cross' feeds the NEW feat0)."""
    t = _transformer(kind, dev)
    C = t.d_model
    g = torch.Generator().manual_seed(L + S + n)
    f0, f1 = torch.randn(n, L, C, generator=g), torch.randn(n, S, C, generator=g)
    ref = copy.deepcopy(t).cpu().double()
    ref.use_hip = False
    with torch.no_grad():
        w0, w1 = ref(f0.double(), f1.double())
        g0, g1 = t(f0.to(dev), f1.to(dev))
        t.use_hip = False
        p0, p1 = t(f0.to(dev), f1.to(dev))     # the torch / rocBLAS plumbing this replaces
    e_hip = max(float((g0.cpu().double() - w0).abs().max()), float((g1.cpu().double() - w1).abs().max()))
    e_torch = max(float((p0.cpu().double() - w0).abs().max()), float((p1.cpu().double() - w1).abs().max()))
    scale = float(w0.abs().max())
    print(f"{kind} n={n} L={L} S={S}: max err HIP {e_hip:.2e}, torch fp32 {e_torch:.2e}, |feat| max {scale:.2f}")
    assert e_hip < 2e-4 * max(1.0, scale)
    assert e_hip < 4 * e_torch + 2e-5        # fp32-equivalent: not worse than the fp32 library path it replaces
    assert g0.shape == (n, L, C) and g1.shape == (n, S, C) and bool(torch.isfinite(g0).all())


def test_hip_encoder_is_deterministic_and_batch_invariant(dev):
    t = _transformer("coarse", dev)
    g = torch.Generator().manual_seed(4)
    f0, f1 = torch.randn(3, 320, 256, generator=g).to(dev), torch.randn(3, 256, 256, generator=g).to(dev)
    with torch.no_grad():
        a0, a1 = t(f0, f1)
        b0, b1 = t(f0, f1)
        c0, c1 = t(f0[1:2], f1[1:2])
    assert torch.equal(a0, b0) and torch.equal(a1, b1)
    assert torch.equal(a0[1:2], c0) and torch.equal(a1[1:2], c1)
    assert f0.data_ptr() != a0.data_ptr()      # inputs are not modified


def test_hip_encoder_range_guard_falls_back(dev):
    t = _transformer("coarse", dev)
    g = torch.Generator().manual_seed(5)
    f0, f1 = torch.randn(1, 128, 256, generator=g).to(dev), torch.randn(1, 128, 256, generator=g).to(dev)
    f0[0, 7, 3] = 2.0e4                         # |x| * 8 >= 65504
    with torch.no_grad(), pytest.warns(UserWarning, match="LoFTR transformer"):
        a0, a1 = t(f0, f1)
    t.use_hip = False
    with torch.no_grad():
        b0, b1 = t(f0, f1)
    assert torch.equal(a0, b0) and torch.equal(a1, b1)


# ---- the HIP ResNet-FPN: every convolution on the f16x3 planes GEMM ---------------------------

def _backbone(dev, seed=0):
    from pope_amd import synth
    from pope_amd.loftr import build_backbone
    from pope_amd.matcher import default_cfg
    b = build_backbone(default_cfg).eval()
    sd = synth.synthetic_matcher_state_dict(seed=seed)
    b.load_state_dict({k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}, strict=True)
    return b.to(dev)


@pytest.mark.parametrize("n,H,W", [(2, 256, 256), (1, 64, 96), (3, 192, 256), (1, 16, 24)])
def test_hip_backbone_matches_fp64_restatement(dev, n, H, W):
    """ResNetFPN_8_2 (resnet_fpn.py:100-118) on the HIP path — implicit 3x3 convolutions over zero-bordered NHWC planes,
    gathered stride-2 taps, fused BN / ReLU / LeakyReLU / shortcut epilogues, bilinear x2 + lateral add — against the same
    module evaluated in float64 on the CPU, next to the MIOpen fp32 path it replaces."""
    from pope_amd import synth
    b = _backbone(dev)
    x = synth.synthetic_gray_pairs(n, H, W, seed=n + H)[0]
    ref = copy.deepcopy(b).cpu().double()
    ref.use_hip = False
    with torch.no_grad():
        wc, wf = ref(x.double())
        gc, gf = b(x.to(dev))
        b.use_hip = False
        pc, pf = b(x.to(dev))
    assert gc.shape == (n, 256, H // 8, W // 8) and gf.shape == (n, 128, H // 2, W // 2)
    for name, g, p, w in (("coarse", gc, pc, wc), ("fine", gf, pf, wf)):
        e_hip = float((g.cpu().double() - w).abs().max())
        e_torch = float((p.cpu().double() - w).abs().max())
        scale = float(w.abs().max())
        print(f"backbone {name} n={n} {H}x{W}: max err HIP {e_hip:.2e}, MIOpen fp32 {e_torch:.2e}, |feat| max {scale:.2f}")
        assert e_hip < 1e-4 * max(1.0, scale)
        assert e_hip < 4 * e_torch + 2e-5 * max(1.0, scale)
        assert bool(torch.isfinite(g).all())


def test_hip_backbone_is_deterministic_and_batch_invariant(dev):
    from pope_amd import synth
    b = _backbone(dev)
    x = synth.synthetic_gray_pairs(3, 64, 96, seed=9)[0].to(dev)
    with torch.no_grad():
        a = b(x)
        c = b(x)
        d = b(x[1:2])
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])
    assert torch.equal(a[0][1:2], d[0]) and torch.equal(a[1][1:2], d[1])


def test_hip_backbone_range_guard_falls_back(dev):
    from pope_amd import synth
    b = _backbone(dev)
    x = synth.synthetic_gray_pairs(1, 64, 64, seed=3)[0].to(dev)
    x[0, 0, 10, 10] = 1.0e4                     # |x| * 8 >= 65504: the stem gather raises the flag
    with torch.no_grad(), pytest.warns(UserWarning, match="LoFTR backbone"):
        a = b(x)
    b.use_hip = False
    with torch.no_grad():
        c = b(x)
    # both are the torch / MIOpen form (not bit-reproducible from call to call: MIOpen picks its solver on first use)
    assert bool(torch.isfinite(a[0]).all())
    for u, v in zip(a, c):
        assert float((u - v).abs().max()) <= 1e-5 * float(v.abs().max())


# ---- the HIP fine stage: window gather + down_proj / merge_feat, and the sub-pixel expectation ------

def _fine_modules(dev):
    from pope_amd import synth
    from pope_amd.loftr import FinePreprocess, FineMatching
    from pope_amd.matcher import default_cfg
    fp = FinePreprocess(default_cfg).eval()
    sd = synth.synthetic_matcher_state_dict(seed=0)
    fp.load_state_dict({k[len("fine_preprocess."):]: v for k, v in sd.items() if k.startswith("fine_preprocess.")}, strict=True)
    return fp.to(dev), FineMatching()


@pytest.mark.parametrize("layout", ["nchw", "nhwc_view"])
@pytest.mark.parametrize("M", [1, 37, 600])
def test_hip_fine_preprocess_matches_torch_form(dev, layout, M):
    """pope_fine_preprocess_f32 against the torch restatement of fine_preprocess.py:29-59 (itself pinned to the reference by
    the Matcher fixtures): windows at random cells incl. the map's corners (zero padding), both memory layouts."""
    fp, _ = _fine_modules(dev)
    g = torch.Generator().manual_seed(M)
    n, hc, wc, s = 2, 12, 16, 4
    hf, wf = hc * s, wc * s
    if layout == "nchw":
        f0, f1 = torch.randn(n, 128, hf, wf, generator=g).to(dev), torch.randn(n, 128, hf, wf, generator=g).to(dev)
    else:   # what the HIP backbone hands over: the interior of a bordered NHWC buffer, permuted
        f0 = torch.randn(n, hf + 2, wf + 2, 128, generator=g).to(dev)[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2)
        f1 = torch.randn(n, hf + 2, wf + 2, 128, generator=g).to(dev)[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2)
    c0, c1 = torch.randn(n, hc * wc, 256, generator=g).to(dev), torch.randn(n, hc * wc, 256, generator=g).to(dev)
    b = torch.randint(0, n, (M,), generator=g).sort().values
    i = torch.randint(0, hc * wc, (M,), generator=g)
    j = torch.randint(0, hc * wc, (M,), generator=g)
    i[0], j[0] = 0, hc * wc - 1                         # corners: windows reach outside the map
    data = {"hw0_f": (hf, wf), "hw0_c": (hc, wc), "hw1_c": (hc, wc), "b_ids": b.to(dev), "i_ids": i.to(dev), "j_ids": j.to(dev)}
    with torch.no_grad():
        g0, g1 = fp(f0, f1, c0, c1, dict(data))
        fp.use_hip = False
        w0, w1 = fp(f0, f1, c0, c1, dict(data))
        r0, r1 = copy.deepcopy(fp).cpu().double()(f0.cpu().double(), f1.cpu().double(), c0.cpu().double(), c1.cpu().double(),
                                                  {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in data.items()})
    assert g0.shape == (M, 25, 128) and g1.shape == (M, 25, 128)
    e_hip = max(float((g0.cpu().double() - r0).abs().max()), float((g1.cpu().double() - r1).abs().max()))
    e_torch = max(float((w0.cpu().double() - r0).abs().max()), float((w1.cpu().double() - r1).abs().max()))
    print(f"fine preprocess M={M} {layout}: max err HIP {e_hip:.2e}, torch fp32 {e_torch:.2e}")
    assert e_hip < 4 * e_torch + 2e-5


@pytest.mark.parametrize("M", [1, 5, 333])
def test_hip_fine_matching_matches_torch_form(dev, M):
    _, fm = _fine_modules(dev)
    g = torch.Generator().manual_seed(100 + M)
    w0, w1 = torch.randn(M, 25, 128, generator=g).to(dev), torch.randn(M, 25, 128, generator=g).to(dev)
    w1[:, 7] = w0[:, 12] * 1.5                        # a clear peak off the centre
    mk = (torch.rand(M, 2, generator=g) * 200).to(dev)
    base = {"hw0_i": (256, 256), "hw0_f": (128, 128), "mkpts0_c": mk.clone(), "mkpts1_c": mk.clone(), "mconf": torch.ones(M, device=dev),
            "b_ids": torch.zeros(M, dtype=torch.long, device=dev)}
    a, b = dict(base), dict(base)
    fm(w0, w1, a)
    fm.use_hip = False
    fm(w0, w1, b)
    torch.testing.assert_close(a["expec_f"][:, :2], b["expec_f"][:, :2], rtol=1e-5, atol=1e-5)
    # the spread is sqrt(E[g^2] - E[g]^2) per axis (fine_matching.py:52-54): for a peaked heatmap the difference cancels
    # to fp32 noise (~1e-7) and its square root amplifies that to ~3e-4 — in the reference's own arithmetic too
    torch.testing.assert_close(a["expec_f"][:, 2], b["expec_f"][:, 2], rtol=1e-4, atol=5e-4)
    torch.testing.assert_close(a["mkpts1_f"], b["mkpts1_f"], rtol=1e-6, atol=1e-4)
    assert torch.equal(a["mkpts0_f"], b["mkpts0_f"])
