"""GPU: every HIP kernel against a plain torch fp32 reference of the same op (computed on the
CPU, like the reference's own path) — called through the C ABI."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(hip_lib):
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _close(got, want, atol, rtol=0.0):
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().cpu().numpy(), rtol=rtol, atol=atol)


@pytest.mark.parametrize("rows,dim", [(1, 384), (7, 384), (197 * 2, 384), (1531, 384), (33, 768), (5, 1024)])
def test_layernorm(dev, rows, dim):
    from pope_amd import ops
    x = _rand(rows, dim, seed=1, scale=3.0) + 0.7
    w, b = 1 + 0.1 * _rand(dim, seed=2), 0.1 * _rand(dim, seed=3)
    want = F.layer_norm(x, (dim,), w, b, 1e-6)
    got = ops.layernorm(x.to(dev), w.to(dev), b.to(dev), 1e-6)
    _close(got, want, atol=2e-6)


PRECS = ["f32", "f16x3"]


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("M,N,K", [(1, 384, 384), (130, 1152, 384), (394, 1536, 384), (257, 384, 1536),
                                    (1531, 1152, 384), (129, 100, 36)])
def test_linear_bias(dev, M, N, K, prec):
    from pope_amd import ops
    a, w, b = _rand(M, K, seed=4), _rand(N, K, seed=5, scale=K ** -0.5), _rand(N, seed=6)
    want = F.linear(a.double(), w.double(), b.double()).float()
    got = ops.linear(a.to(dev), w.to(dev), b.to(dev), precision=prec)
    _close(got, want, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("scale", [1e-3, 1.0, 300.0, 1500.0])  # |a| stays below the f16x3 range contract (8188)
def test_linear_f16x3_dynamic_range(dev, scale):
    # the split representation is relative (2^-22) above 2^-3 and absolute (2^-25) below; products of
    # activations at `scale` with O(0.05) weights must stay at fp32-chain accuracy up to the f16 range
    from pope_amd import ops
    a, w = _rand(300, 384, seed=21, scale=scale), _rand(384, 384, seed=22, scale=0.05)
    want = (a.double() @ w.double().t()).float()
    got32 = ops.linear(a.to(dev), w.to(dev), precision="f32").cpu()
    got16 = ops.linear(a.to(dev), w.to(dev), precision="f16x3").cpu()
    e32, e16 = float((got32 - want).abs().max()), float((got16 - want).abs().max())
    ref = float(want.abs().max())
    print(f"scale {scale:g}: max|err| f32-chain {e32:.2e}  f16x3 {e16:.2e}  (|out| max {ref:.2e})")
    assert e16 <= max(4 * e32, 2e-6 * ref)


@pytest.mark.parametrize("prec", PRECS)
def test_linear_is_exact_fma_chain_on_integers(dev, prec):
    # integer-valued operands: every product and partial sum is exact in fp32, so any operand-map
    # or k-permutation error shows up as a wrong integer (asymmetric W: catches transposes)
    from pope_amd import ops
    g = torch.Generator().manual_seed(7)
    a = torch.randint(-4, 5, (200, 96), generator=g).float()
    w = torch.randint(-4, 5, (136, 96), generator=g).float()
    got = ops.linear(a.to(dev), w.to(dev), precision=prec)
    assert torch.equal(got.cpu(), a @ w.t())


@pytest.mark.parametrize("prec", PRECS)
def test_linear_gelu(dev, prec):
    from pope_amd import ops
    a, w, b = _rand(300, 384, seed=8), _rand(1536, 384, seed=9, scale=384 ** -0.5), _rand(1536, seed=10)
    want = F.gelu(F.linear(a, w, b))
    got = ops.linear(a.to(dev), w.to(dev), b.to(dev), epilogue=ops.EPI_BIAS_GELU, precision=prec)
    _close(got, want, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("prec", PRECS)
def test_linear_layerscale_residual_inplace(dev, prec):
    from pope_amd import ops
    a, w, b = _rand(300, 1536, seed=11), _rand(384, 1536, seed=12, scale=1536 ** -0.5), _rand(384, seed=13)
    gamma, res = 0.3 + 0.1 * _rand(384, seed=14), _rand(300, 384, seed=15)
    want = res + F.linear(a, w, b) * gamma
    x = res.to(dev).clone()
    got = ops.linear(a.to(dev), w.to(dev), b.to(dev), epilogue=ops.EPI_BIAS_LS_RES, gamma=gamma.to(dev), res=x, out=x,
                     precision=prec)
    assert got.data_ptr() == x.data_ptr()
    _close(got, want, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("B,H,W", [(2, 28, 42), (1, 196, 196), (2, 224, 224), (1, 476, 630)])
def test_patch_embed_tokens(dev, sd0, B, H, W, prec):
    from oracle import dinov2_ref
    from pope_amd import synth
    from pope_amd.dinov2_utils import load_dinov2_model
    model = load_dinov2_model(state_dict=sd0).to(dev)
    model.precision = prec
    x = synth.synthetic_images(B, H, W, seed=21)
    want = dinov2_ref.prepare_tokens(sd0, x)
    got = model.prepare_tokens_with_masks(x.to(dev))
    _close(got, want, atol=2e-5)


def test_patch_embed_rejects_non_multiple(dev, sd0):
    from pope_amd.dinov2_utils import load_dinov2_model
    model = load_dinov2_model(state_dict=sd0).to(dev)
    with pytest.raises(AssertionError, match="not a multiple of patch"):
        model(torch.zeros(1, 3, 480, 640, device=dev))


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("B,N,heads", [(1, 1, 6), (2, 31, 6), (2, 64, 6), (1, 197, 6), (2, 257, 6), (1, 1531, 6),
                                        (1, 130, 12)])
def test_attention(dev, B, N, heads, prec):
    from pope_amd import ops
    D = heads * 64
    qkv = _rand(B, N, 3 * D, seed=30 + N, scale=1.5)
    q, k, v = qkv.reshape(B, N, 3, heads, 64).permute(2, 0, 3, 1, 4)
    want = (((q * 0.125) @ k.transpose(-2, -1)).softmax(-1) @ v).transpose(1, 2).reshape(B, N, D)
    got = ops.attention(qkv.to(dev), heads, precision=prec)
    _close(got, want, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("prec", PRECS)
def test_attention_error_vs_fp64(dev, prec):
    # both arithmetic modes against an fp64 reference on the bench shape's key count (N = 1531)
    from pope_amd import ops
    B, N, heads = 1, 1531, 6
    qkv = _rand(B, N, 3 * heads * 64, seed=77, scale=1.5)
    q, k, v = qkv.double().reshape(B, N, 3, heads, 64).permute(2, 0, 3, 1, 4)
    want = (((q * 0.125) @ k.transpose(-2, -1)).softmax(-1) @ v).transpose(1, 2).reshape(B, N, heads * 64)
    got = ops.attention(qkv.to(dev), heads, precision=prec).cpu().double()
    err = float((got - want).abs().max())
    print(f"attention {prec}: max |err| vs fp64 = {err:.2e} (|out| max {float(want.abs().max()):.2f})")
    assert err < 2e-5


@pytest.mark.parametrize("B,N,heads,spike", [(2, 1531, 6, False), (1, 197, 16, False), (3, 64, 2, False), (1, 1, 1, False),
                                             (1, 130, 4, False), (1, 1531, 2, True)])
def test_attention_f16_on_f16_operands(dev, hip_lib, B, N, heads, spike):
    """pope_attention_f16 (POPE_PREC_F16, attention_f16.hip: LDS-direct K / V staging, four stages, in-wave pipeline) against
    fp64 attention computed from THE SAME f16-rounded operands: what is left is the rounding of P to f16 and of the output to
    f16 (both <= 2^-11 relative).  Shapes: the bench's key count, one / two / three-tile sequences with ragged ends (the
    prologue, the tail mask and every wait-count branch), a single key; `spike`: a key whose score jumps by 25 log2 units at
    a late tile, so that the running maximum moves by a large factor mid-sequence (o and l rescaled)."""
    import ctypes as C
    D = heads * 64
    g = torch.Generator().manual_seed(1000 * N + heads)
    q = torch.randn(B, N, heads, 64, generator=g) * 1.2
    k = torch.randn(B, N, heads, 64, generator=g) * 1.2
    v = torch.randn(B, N, heads, 64, generator=g) * 2.0
    if spike:
        k[:, N - 200] = 6.0 * q[:, 5]          # query 5 (and its neighbours in direction) suddenly find a huge key
    qs = (q * (0.125 * 1.4426950408889634)).half()     # what the QKV epilogue writes: q * head_dim^-0.5 * log2 e, ONE rounding
    kh, vh = k.half(), v.half()
    qkv = torch.stack([qs, kh, vh], 2).reshape(B * N, 3 * D).contiguous().to(dev)
    out = torch.full((B * N + 3, D), 7.0, dtype=torch.float16, device=dev)     # three guard rows
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        assert hip_lib.pope_attention_f16(C.c_void_p(qkv.data_ptr()), C.c_void_p(out.data_ptr()), B, N, heads, st) == 0
    torch.cuda.synchronize()
    assert bool((out[B * N:] == 7.0).all())
    s = torch.einsum("bqhd,bkhd->bhqk", qs.double(), kh.double())              # log2-domain scores
    p = torch.exp2(s - s.amax(-1, keepdim=True))
    want = torch.einsum("bhqk,bkhd->bqhd", p / p.sum(-1, keepdim=True), vh.double()).reshape(B * N, D)
    got = out[:B * N].cpu().double() / 8.0
    err = float((got - want).abs().max())
    print(f"attention f16 B={B} N={N} heads={heads}: max |err| vs fp64 on the same operands = {err:.2e} (|out| max {float(want.abs().max()):.2f})")
    assert err < 3e-3 * max(1.0, float(want.abs().max()))
    assert float((got - want).abs().mean()) < 2e-4 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("prec", PRECS)
def test_attention_online_softmax_rescale_branch(dev, prec):
    # force the running max to jump at a late key tile (spiked key), so the rescale of O and l is
    # exercised with a large factor (guide rule: a rare data-dependent branch needs its own test)
    from pope_amd import ops
    B, N, heads = 1, 300, 6
    qkv = _rand(B, N, 3 * heads * 64, seed=99, scale=0.5)
    t = qkv.view(B, N, 3, heads, 64)
    t[0, 250, 1] = t[0, 10, 0] * 40.0   # key 250 aligned with query 10: score jumps by ~+100
    t[0, 290, 1] = t[0, 70, 0] * -40.0  # and a strongly negative one
    q, k, v = t.permute(2, 0, 3, 1, 4)
    want = (((q * 0.125) @ k.transpose(-2, -1)).softmax(-1) @ v).transpose(1, 2).reshape(B, N, heads * 64)
    got = ops.attention(qkv.to(dev), heads, precision=prec)
    _close(got, want, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("B,N,heads,spike", [(2, 257, 6, False), (1, 1531, 6, False), (3, 64, 2, False), (1, 130, 6, False),
                                             (1, 300, 6, True)])
def test_attention_planes_in_planes_out(dev, hip_lib, B, N, heads, spike):
    """The whole-model f16x3 dataflow: q, k, v as hi/lo planes (what the QKV GEMM epilogue writes) -> attention ->
    output planes (what the proj GEMM reads), through the C ABI, against fp64 on the values the planes represent.
    Covers one / two / many key tiles, a ragged last tile, and a late spike that forces a large rescale."""
    import ctypes as C
    from pope_amd import _lib
    D = heads * 64
    qkv = _rand(B, N, 3 * D, seed=40 + N, scale=0.5 if spike else 1.5)
    if spike:
        t = qkv.view(B, N, 3, heads, 64)
        t[0, 250, 1] = t[0, 10, 0] * 40.0
        t[0, 290, 1] = t[0, 70, 0] * -40.0
    planes = _lib.to_planes(qkv.reshape(B * N, 3 * D), _lib.PLANES_ACT_SCALE)
    seen = _lib.from_planes(planes, _lib.PLANES_ACT_SCALE).double().reshape(B, N, 3 * D)   # what the kernel sees
    q, k, v = seen.reshape(B, N, 3, heads, 64).permute(2, 0, 3, 1, 4)
    want = (((q * 0.125) @ k.transpose(-2, -1)).softmax(-1) @ v).transpose(1, 2).reshape(B * N, D)
    pin = planes.to(dev)
    pout = torch.zeros(B * N, D // 32, 2, 32, dtype=torch.float16, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert hip_lib.pope_attention_planes_f32(C.c_void_p(pin.data_ptr()), C.c_void_p(pout.data_ptr()), B, N, heads, st) == 0
    got = _lib.from_planes(pout.cpu(), _lib.PLANES_ACT_SCALE).double()
    err = float(((got - want).abs() / (1.0 + want.abs())).max())
    print(f"planes attention B={B} N={N}: max scaled |err| vs fp64 = {err:.2e}")
    assert err < 2e-5, err


@pytest.mark.parametrize("step", [0.7, 2.0, 4.5, 9.0, -3.0])
@pytest.mark.parametrize("planes", [False, True])
def test_attention_lazy_reference_paths(dev, hip_lib, step, planes):
    """The f16x3 kernel exponentiates against a LAZY reference (the row maximum of the last exact pass) and takes the
    exact pass again only when a tile's probabilities would leave the f16 range.  Scores that climb by `step` (natural
    log units) per 64-key tile drive every mix of the two paths: never again after tile 0 (falling or slowly rising
    scores), every third / second tile, every tile.  Result must not depend on the path: fp64 reference."""
    import ctypes as C
    from pope_amd import ops, _lib
    B, N, heads = 1, 520, 6          # 9 key tiles, the last one ragged
    D = heads * 64
    g = torch.Generator().manual_seed(123)
    qkv = torch.randn(B, N, 3, heads, 64, generator=g) * 0.3
    u = torch.randn(heads, 64, generator=g)
    u = u / u.norm(dim=-1, keepdim=True) * 8.0                      # |u|^2 / 8 = 8: score = 8 * ramp + noise
    ramp = torch.arange(N, dtype=torch.float32) / 64.0 * (step / 8.0)
    qkv[0, :, 0] += u                                               # every query looks along u
    qkv[0, :, 1] += u * ramp[:, None, None]                          # key j: score grows by `step` per 64 keys
    qkv = qkv.reshape(B, N, 3 * D).contiguous()
    if planes:
        pl = _lib.to_planes(qkv.reshape(B * N, 3 * D), _lib.PLANES_ACT_SCALE)
        seen = _lib.from_planes(pl, _lib.PLANES_ACT_SCALE).double().reshape(B, N, 3 * D)
    else:
        seen = qkv.double()
    q, k, v = seen.reshape(B, N, 3, heads, 64).permute(2, 0, 3, 1, 4)
    want = (((q * 0.125) @ k.transpose(-2, -1)).softmax(-1) @ v).transpose(1, 2).reshape(B * N, D)
    if planes:
        pin = pl.to(dev)
        pout = torch.zeros(B * N, D // 32, 2, 32, dtype=torch.float16, device=dev)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert hip_lib.pope_attention_planes_f32(C.c_void_p(pin.data_ptr()), C.c_void_p(pout.data_ptr()), B, N, heads, st) == 0
        got = _lib.from_planes(pout.cpu(), _lib.PLANES_ACT_SCALE).double()
    else:
        got = ops.attention(qkv.to(dev), heads, precision="f16x3").cpu().double().reshape(B * N, D)
    err = float(((got - want).abs() / (1.0 + want.abs())).max())
    print(f"lazy-reference attention step={step} planes={planes}: max scaled |err| vs fp64 = {err:.2e}")
    assert err < 2e-5, err


def test_attention_reference_advances_row_by_row(dev, hip_lib):
    """Look-ahead reference advance (attention_f16x3.hip, round 4): inside one 32-query wave only SOME rows see scores that
    climb out of the f16 range of their reference — rows that look along u — while their neighbours (random queries) stay
    diffuse.  The advancing rows must move (their o / l rescaled, the waiting scores re-biased) and the others must not
    (alpha = 1 exactly); a late single outlier key (an attention sink in the LAST tile) and a sink in the first tile are in
    the same launch.  fp64 reference on the values the planes hold."""
    import ctypes as C
    from pope_amd import _lib
    B, N, heads = 2, 700, 6
    D = heads * 64
    g = torch.Generator().manual_seed(7)
    qkv = torch.randn(B, N, 3, heads, 64, generator=g) * 0.4
    u = torch.randn(heads, 64, generator=g)
    u = u / u.norm(dim=-1, keepdim=True) * 8.0
    ramp = torch.arange(N, dtype=torch.float32) / 64.0 * (9.0 / 8.0)
    qkv[0, 0::3, 0] += u                                  # every third query of image 0 looks along u ...
    qkv[0, :, 1] += u * ramp[:, None, None]               # ... and sees keys whose score climbs 13 log2 units per tile
    qkv[1, :, 0] += u * 0.5
    qkv[1, 3, 1] += u * 2.0                               # image 1: a sink key in the first tile,
    qkv[1, N - 5, 1] += u * 6.0                           # and a far stronger one in the last (ragged) tile
    pl = _lib.to_planes(qkv.reshape(B * N, 3 * D).contiguous(), _lib.PLANES_ACT_SCALE)
    seen = _lib.from_planes(pl, _lib.PLANES_ACT_SCALE).double().reshape(B, N, 3, heads, 64)
    q, k, v = seen.permute(2, 0, 3, 1, 4)
    want = (((q * 0.125) @ k.transpose(-2, -1)).softmax(-1) @ v).transpose(1, 2).reshape(B * N, D)
    pin = pl.to(dev)
    pout = torch.zeros(B * N, D // 32, 2, 32, dtype=torch.float16, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert hip_lib.pope_attention_planes_f32(C.c_void_p(pin.data_ptr()), C.c_void_p(pout.data_ptr()), B, N, heads, st) == 0
    got = _lib.from_planes(pout.cpu(), _lib.PLANES_ACT_SCALE).double()
    err = float(((got - want).abs() / (1.0 + want.abs())).max())
    n_adv = C.c_longlong(0)
    assert hip_lib.pope_attention_planes_diag_f32(C.c_void_p(pin.data_ptr()), C.c_void_p(pout.data_ptr()), B, N, heads, C.byref(n_adv), st) == 0
    pairs = B * heads * (-(-N // 32)) * (-(-N // 64))
    print(f"row-by-row advance: max scaled |err| vs fp64 = {err:.2e}; {n_adv.value} advances in {pairs} (wave, tile) pairs")
    assert err < 2e-5 and 0 < n_adv.value < pairs
    assert torch.equal(_lib.from_planes(pout.cpu(), _lib.PLANES_ACT_SCALE).double(), got)    # the counting twin: same bits


def test_cls_cosine_and_top3(dev, golden_dir):
    import os
    from pope_amd import ops
    fx = np.load(os.path.join(golden_dir, "top3.npz"))
    scores = ops.cls_cosine(torch.from_numpy(fx["ref"]).to(dev), torch.from_numpy(fx["fea"]).to(dev))
    np.testing.assert_allclose(scores.cpu().numpy(), fx["scores"], rtol=0, atol=1e-6)
    slots, idx = ops.streaming_top3(scores.cpu().numpy())
    assert np.array_equal(idx, fx["slot_index"])  # identical top-3 slot assignment
    z = ops.cls_cosine(torch.tensor([[1e-9, 0, 0]], device=dev), torch.tensor([[2e-9, 0, 0]], device=dev))
    assert abs(float(z) - 0.02) < 1e-6


@pytest.mark.parametrize("epi", ["bias", "gelu_planes_out", "ls_res"])
def test_linear_planes_kernel(dev, hip_lib, epi):
    """f16x3 GEMM on pre-split planes (as LayerNorm / the GELU epilogue / the weight loader produce them),
    through the C ABI, against fp64."""
    import ctypes as C
    from pope_amd import _lib
    M, N, K = 700, 384, 1536
    a, w, b = _rand(M, K, seed=61, scale=1.3), _rand(N, K, seed=62, scale=K ** -0.5), _rand(N, seed=63)
    ap = _lib.to_planes(a, _lib.PLANES_ACT_SCALE).to(dev)
    wp = _lib.to_planes(w, _lib.PLANES_W_SCALE).to(dev)
    bd = b.to(dev)
    lin = F.linear(a.double(), w.double(), b.double())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    if epi == "gelu_planes_out":
        op = torch.empty(M, N // 32, 2, 32, dtype=torch.float16, device=dev)
        assert hip_lib.pope_linear_planes_f32(P(ap), P(wp), P(bd), None, P(op), M, N, K, 1, None, None, None, st) == 0
        got, want = _lib.from_planes(op.cpu(), _lib.PLANES_ACT_SCALE), F.gelu(lin).float()
    elif epi == "ls_res":
        gamma, res = 0.3 + 0.1 * _rand(N, seed=64), _rand(M, N, seed=65)
        x = res.to(dev).clone()
        assert hip_lib.pope_linear_planes_f32(P(ap), P(wp), P(bd), P(x), None, M, N, K, 2, P(gamma.to(dev)), P(x), None, st) == 0
        got, want = x.cpu(), (res.double() + lin * gamma.double()).float()
    else:
        out = torch.empty(M, N, device=dev)
        assert hip_lib.pope_linear_planes_f32(P(ap), P(wp), P(bd), P(out), None, M, N, K, 0, None, None, None, st) == 0
        got, want = out.cpu(), lin.float()
    _close(got, want, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("shape", [(64 * 197 + 3, 1152, 384), (130, 100, 128), (9000, 384, 96), (300, 256, 64)])
def test_linear_planes_ragged_shapes(dev, hip_lib, shape):
    """Ragged M / N (rows and columns beyond the last tile are dropped by the buffer descriptors), odd numbers of
    K-steps (the K-step stream runs in pairs; a trailing dummy item must not be stored), repeated launches."""
    import ctypes as C
    from pope_amd import _lib
    M, N, K = shape
    a, w, b = _rand(M, K, seed=71, scale=1.3), _rand(N, K, seed=72, scale=K ** -0.5), _rand(N, seed=73)
    ap = _lib.to_planes(a, _lib.PLANES_ACT_SCALE).to(dev)
    wp = _lib.to_planes(w, _lib.PLANES_W_SCALE).to(dev)
    bd = b.to(dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    guard = torch.full((M + 2, N), 7.0, device=dev)      # rows M, M+1 must stay untouched
    for _ in range(2):
        assert hip_lib.pope_linear_planes_f32(P(ap), P(wp), P(bd), P(guard), None, M, N, K, 0, None, None, None, st) == 0
    assert bool((guard[M:] == 7.0).all())
    _close(guard[:M].cpu(), F.linear(a.double(), w.double(), b.double()).float(), atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("route", ["wide256", "stream384"])
@pytest.mark.parametrize("N,epi", [(1152, 0), (1536, 1)])
def test_wide_planes_gemm_equals_tile_kernel(dev, hip_lib, N, epi, route):
    """Planes -> planes Linear layers at large M (QKV, FC1 . GELU of the ViT blocks) run on LDS-direct mainloops (round 4): from
    4 x CUs tiles of 192 x 384 on the persistent tile stream of gemm_rowln.hip (widths that are multiples of 384), below that
    from 4 x CUs tiles of 256 x 256 on gemm_plain.hip; every other call on the 128 x 128 tile kernel.  Same accumulation order,
    same epilogue arithmetic: the big call must equal, bit for bit, the same rows computed in pieces small enough to take the
    tile kernel — with a ragged last row tile, and the range flag raised by the same outlier."""
    import ctypes as C
    from pope_amd import _lib
    K = 384
    if route == "wide256":
        M = 256 * (1024 // (-(-N // 256)) + 3) + 77      # a few more than 4 x CUs tiles of 256 x 256, ragged; fewer than 4 x CUs of 192 x 384
        assert -(-M // 192) * (N // 384) < 1024
    else:
        M = 192 * (1024 // (N // 384) + 3) + 77          # a few more than 4 x CUs tiles of 192 x 384, ragged
    g = torch.Generator().manual_seed(N)
    a = torch.randn(M, K, generator=g) * 1.3
    w, b = torch.randn(N, K, generator=g) * K ** -0.5, torch.randn(N, generator=g)
    a[M - 5, 7] = 3000.0                                  # one huge activation: an output beyond the f16 range of the planes
    w[11, 7] = 3.0
    ap = _lib.to_planes(a, _lib.PLANES_ACT_SCALE).to(dev)
    wp = _lib.to_planes(w, _lib.PLANES_W_SCALE).to(dev)
    bd = b.to(dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None

    def run(rows_lo, rows_hi):
        m = rows_hi - rows_lo
        out = torch.zeros(m, N // 32, 2, 32, dtype=torch.float16, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        assert hip_lib.pope_linear_planes_f32(P(ap[rows_lo:rows_hi]), P(wp), P(bd), None, P(out), m, N, K, epi, None, None, P(flag), st) == 0
        return out, int(flag.item())

    big, flag_big = run(0, M)
    step = 20000                                          # 79 row tiles x <= 6: far below the switch
    parts = [run(lo, min(lo + step, M)) for lo in range(0, M, step)]
    small = torch.cat([p[0] for p in parts])
    assert torch.equal(big.view(torch.int16), small.view(torch.int16))
    assert flag_big != 0 and flag_big == max(p[1] for p in parts)
    rows = torch.arange(0, M - 8, 3001)
    lin = F.linear(a[rows].double(), w.double(), b.double())
    want = F.gelu(lin) if epi else lin
    got = _lib.from_planes(big[rows.to(dev)].cpu(), _lib.PLANES_ACT_SCALE).double()
    assert float((got - want).abs().max()) < 2e-5


def test_layernorm_planes_and_split(dev, hip_lib):
    import ctypes as C
    from pope_amd import _lib
    rows, dim = 333, 384
    x = _rand(rows, dim, seed=71, scale=3.0) + 0.7
    w, b = 1 + 0.1 * _rand(dim, seed=72), 0.1 * _rand(dim, seed=73)
    want = F.layer_norm(x, (dim,), w, b, 1e-6)
    yp = torch.empty(rows, dim // 32, 2, 32, dtype=torch.float16, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    assert hip_lib.pope_layernorm_planes_f32(P(x.to(dev)), P(w.to(dev)), P(b.to(dev)), P(yp), rows, dim, 1e-6, None, st) == 0
    _close(_lib.from_planes(yp.cpu(), _lib.PLANES_ACT_SCALE), want, atol=3e-6)
    # the device splitter == the torch formulation used for the weight planes (bit for bit)
    src = _rand(1000, 64, seed=74, scale=0.05)
    sp = torch.empty(1000, 2, 2, 32, dtype=torch.float16, device=dev)
    assert hip_lib.pope_split_planes_f32(P(src.to(dev)), P(sp), 1000, 64, _lib.PLANES_W_SCALE, None, st) == 0
    assert torch.equal(sp.cpu(), _lib.to_planes(src, _lib.PLANES_W_SCALE))


def test_build_then_smoke_in_one_process(dev):
    """The driver's two entry points back to back in ONE fresh process: build() loads libpope_hip.so before anything has
    imported torch.  The library must still end up on torch's HIP runtime (pope_amd/_lib.py:lib imports torch first;
    with the system libamdhip64 loaded first every launch on a torch stream fails)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=root,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "smoke ok" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


def test_empty_batches(dev, sd0):
    """A batch of zero images / zero pairs is legal in the reference (every torch op is a no-op on it): empty outputs of
    the right shapes, no launch, no error."""
    from pope_amd.dinov2_utils import load_dinov2_model
    from pope_amd.matcher import dense_match
    model = load_dinov2_model(state_dict=sd0).to(dev)
    out = model(torch.zeros(0, 3, 112, 154, device=dev), is_training=True)
    assert out["x_norm_patchtokens"].shape == (0, 88, 384) and out["x_norm_clstoken"].shape == (0, 384)
    assert out["x_prenorm"].shape == (0, 89, 384)
    f = torch.zeros(0, 88, 384, device=dev)
    m = dense_match(f, f, (8, 11), (8, 11), (112, 154))
    assert m["b_ids"].numel() == 0 and m["mkpts0_c"].shape == (0, 2) and m["conf_matrix"].shape == (0, 88, 88)
    assert dense_match(f, f, (8, 11), (8, 11), (112, 154), want_conf=False)["conf_matrix"] is None
