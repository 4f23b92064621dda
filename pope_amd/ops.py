"""Op-level Python wrappers over the C ABI (one HIP kernel each).  fp32, CUDA(HIP) tensors only."""
import ctypes as C

import torch

from . import _lib
from ._lib import check, on_device_of, ptr, require_cuda, stream_of

EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_LS_RES = 0, 1, 2


def _f32c(t, what):
    require_cuda(t, what)
    if t.dtype != torch.float32:
        raise TypeError(f"{what}: expected float32, got {t.dtype}")
    return t.contiguous()


def layernorm(x, weight, bias, eps=1e-6):
    """nn.LayerNorm over the last dim (vision_transformer.py:90)."""
    x = _f32c(x, "layernorm")
    dim = x.shape[-1]
    y = torch.empty_like(x)
    with on_device_of(x):
        check(_lib.lib().pope_layernorm_f32(ptr(x), ptr(_f32c(weight, "ln.w")), ptr(_f32c(bias, "ln.b")), ptr(y),
                                            x.numel() // dim, dim, float(eps), stream_of(x.device)), "pope_layernorm_f32")
    return y


def _flag_ptr(range_flag):
    """Device pointer of an optional int32[1] f16x3 range-guard word (pope_hip.h: range_flag)."""
    if range_flag is None:
        return None
    if range_flag.dtype != torch.int32 or range_flag.numel() < 1 or not range_flag.is_cuda:
        raise ValueError("range_flag must be an int32 CUDA tensor")
    return C.c_void_p(range_flag.data_ptr())


def linear(a, weight, bias=None, epilogue=EPI_BIAS, gamma=None, res=None, out=None, precision="f32", range_flag=None):
    """nn.Linear with fused epilogue: bias | bias+GELU(erf) | res + gamma*(.+bias).
    precision: "f32" (exact fp32 MFMA chain) or "f16x3" (error-compensated f16 matrix cores; `range_flag`, an
    int32[1] device tensor, receives POPE_RANGE_INPUT when an operand leaves the f16 range)."""
    a = _f32c(a, "linear")
    weight = _f32c(weight, "linear.w")
    n, k = weight.shape
    assert a.shape[-1] == k
    m = a.numel() // k
    if out is None:
        out = torch.empty(*a.shape[:-1], n, device=a.device, dtype=torch.float32)
    with on_device_of(a):
        check(_lib.lib().pope_linear_prec_f32(ptr(a), ptr(weight), ptr(bias), ptr(out), m, n, k, epilogue, ptr(gamma),
                                              ptr(res), _lib.PRECISIONS[precision], _flag_ptr(range_flag), stream_of(a.device)),
                  "pope_linear_prec_f32")
    return out


def patch_embed(img, proj_w, posb, patch, precision="f32", range_flag=None):
    """PatchEmbed + cls + pos (patch_embed.py:69-82, vision_transformer.py:191-200).  precision="f16x3": patches are
    gathered into hi/lo f16 planes and multiplied on the f16 matrix cores (3 MFMAs per product, fp32 accumulate)."""
    img = _f32c(img, "patch_embed")
    b, c, h, w = img.shape
    assert c == 3
    assert h % patch == 0, f"Input image height {h} is not a multiple of patch height {patch}"
    assert w % patch == 0, f"Input image width {w} is not a multiple of patch width: {patch}"
    dim = proj_w.shape[0]
    ntok = 1 + (h // patch) * (w // patch)
    assert posb.shape == (ntok, dim)
    out = torch.empty(b, ntok, dim, device=img.device, dtype=torch.float32)
    pw = _f32c(proj_w.reshape(dim, -1), "pe.w")
    if precision == "f16x3":
        kp = (pw.shape[1] + 31) // 32 * 32
        wp = _lib.to_planes(torch.nn.functional.pad(pw, (0, kp - pw.shape[1])), _lib.PLANES_W_SCALE)
        scratch = torch.empty(b * ntok * kp * 4, dtype=torch.uint8, device=img.device)
        with on_device_of(img):
            check(_lib.lib().pope_patch_embed_planes_f32(ptr(img), ptr(wp), ptr(_f32c(posb, "posb")), ptr(out), b, h, w, patch,
                                                         dim, ptr(scratch), scratch.numel(), _flag_ptr(range_flag),
                                                         stream_of(img.device)),
                  "pope_patch_embed_planes_f32")
        return out
    with on_device_of(img):
        check(_lib.lib().pope_patch_embed_f32(ptr(img), ptr(pw), ptr(_f32c(posb, "posb")),
                                              ptr(out), b, h, w, patch, dim, stream_of(img.device)), "pope_patch_embed_f32")
    return out


def attention(qkv, heads, precision="f32", range_flag=None):
    """softmax((q/8) k^T) v on qkv[B,N,3*heads*64] (attention.py:51-59)."""
    qkv = _f32c(qkv, "attention")
    b, n, d3 = qkv.shape
    assert d3 == 3 * heads * 64
    out = torch.empty(b, n, heads * 64, device=qkv.device, dtype=torch.float32)
    with on_device_of(qkv):
        check(_lib.lib().pope_attention_prec_f32(ptr(qkv), ptr(out), b, n, heads, _lib.PRECISIONS[precision],
                                                 _flag_ptr(range_flag), stream_of(qkv.device)), "pope_attention_prec_f32")
    return out


def cls_cosine(ref, fea, eps=1e-8):
    """F.cosine_similarity(ref[1,D], fea[P,D], dim=1, eps) (eval_linemod_json.py:94)."""
    ref = _f32c(ref, "cls_cosine").reshape(-1)
    fea = _f32c(fea, "cls_cosine")
    p, d = fea.shape
    assert ref.numel() == d
    scores = torch.empty(p, device=fea.device, dtype=torch.float32)
    with on_device_of(ref):
        check(_lib.lib().pope_cls_cosine_f32(ptr(ref), ptr(fea), p, d, float(eps), ptr(scores), stream_of(fea.device)),
              "pope_cls_cosine_f32")
    return scores


def streaming_top3(scores):
    """Host-side streaming top-3 vote (eval_linemod_json.py:71,95-101).
    Returns (slot_scores float32[3], slot_index int64[3], -1 = empty slot)."""
    import numpy as np
    s = np.ascontiguousarray(np.asarray(scores, dtype=np.float32).reshape(-1))
    slots = np.zeros(3, np.float32)
    idx = np.zeros(3, np.int64)
    check(_lib.lib().pope_streaming_top3_host(s.ctypes.data_as(_lib.c_float_p), int(s.size),
                                              slots.ctypes.data_as(_lib.c_float_p),
                                              idx.ctypes.data_as(_lib.c_ll_p)), "pope_streaming_top3_host")
    return slots, idx
