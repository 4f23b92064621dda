"""CPU restatement (torch fp32) of the reference DINOv2 ViT forward.

TEST INFRASTRUCTURE — see oracle/__init__.py.  Functional style: every
function takes the reference-layout state dict (175 keys for ViT-S/14).

Reference anchors (paths relative to /root/reference):
  prepare_tokens / pos-embed interpolation  dinov2/dinov2/models/vision_transformer.py:165-200
  patch embed                               dinov2/dinov2/layers/patch_embed.py:69-82
  attention                                 dinov2/dinov2/layers/attention.py:49-62
  mlp                                       dinov2/dinov2/layers/mlp.py:35-44
  layer scale                               dinov2/dinov2/layers/layer_scale.py:27-28
  block (eval branch)                       dinov2/dinov2/layers/block.py:82-88,105-106
  forward_features / forward                dinov2/dinov2/models/vision_transformer.py:221-236,290-295
"""
import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-6  # vision_transformer.py:90


def arch_from_state_dict(sd):
    """Infer (embed_dim, depth, num_heads, patch, grid) from a state dict."""
    dim = sd["cls_token"].shape[-1]
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    patch = sd["patch_embed.proj.weight"].shape[-1]
    heads = dim // 64  # every DINOv2 arch uses head_dim 64 (vision_transformer.py:306-358)
    grid = int(math.isqrt(sd["pos_embed"].shape[1] - 1))
    return dim, depth, heads, patch, grid


def interpolate_pos_encoding(pos_embed, n_tokens, h_img, w_img, patch):
    """vision_transformer.py:165-189 (note the reference calls (w, h) what are
    really (H, W); rows <- height, cols <- width).  fp32."""
    n_patch = n_tokens - 1
    n_grid = pos_embed.shape[1] - 1
    if n_patch == n_grid and h_img == w_img:
        return pos_embed
    pe = pos_embed.float()
    cls_pe = pe[:, 0]
    patch_pe = pe[:, 1:]
    dim = pe.shape[-1]
    g = int(math.sqrt(n_grid))
    h0 = h_img // patch + 0.1
    w0 = w_img // patch + 0.1
    patch_pe = F.interpolate(
        patch_pe.reshape(1, g, g, dim).permute(0, 3, 1, 2),
        scale_factor=(h0 / math.sqrt(n_grid), w0 / math.sqrt(n_grid)),
        mode="bicubic",
    )
    assert int(h0) == patch_pe.shape[-2] and int(w0) == patch_pe.shape[-1]
    patch_pe = patch_pe.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((cls_pe.unsqueeze(0), patch_pe), dim=1)


def patch_embed(sd, x):
    """patch_embed.py:69-82: conv k=s=patch, flatten(2).transpose(1,2)."""
    w = sd["patch_embed.proj.weight"]
    p = w.shape[-1]
    _, _, H, W = x.shape
    assert H % p == 0, f"Input image height {H} is not a multiple of patch height {p}"
    assert W % p == 0, f"Input image width {W} is not a multiple of patch width: {p}"
    y = F.conv2d(x, w, sd["patch_embed.proj.bias"], stride=p)
    return y.flatten(2).transpose(1, 2)


def prepare_tokens(sd, x):
    """vision_transformer.py:191-200 without masks."""
    B, _, H, W = x.shape
    patch = sd["patch_embed.proj.weight"].shape[-1]
    t = patch_embed(sd, x)
    t = torch.cat((sd["cls_token"].expand(B, -1, -1), t), dim=1)
    return t + interpolate_pos_encoding(sd["pos_embed"], t.shape[1], H, W, patch)


def attention(sd, pre, x, heads):
    """attention.py:49-62 (plain path, taken when xformers is absent)."""
    B, N, C = x.shape
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"])
    qkv = qkv.reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * ((C // heads) ** -0.5), qkv[1], qkv[2]
    a = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"]), qkv, o


def mlp(sd, pre, x):
    """mlp.py:35-44, exact-erf GELU."""
    h = F.gelu(F.linear(x, sd[pre + "fc1.weight"], sd[pre + "fc1.bias"]))
    return F.linear(h, sd[pre + "fc2.weight"], sd[pre + "fc2.bias"])


def block(sd, i, x, heads, taps=None):
    """block.py:82-88,105-106 eval branch; optional per-stage taps."""
    p = f"blocks.{i}."
    dim = x.shape[-1]
    n1 = F.layer_norm(x, (dim,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], LN_EPS)
    a, qkv, ao = attention(sd, p + "attn.", n1, heads)
    x = x + a * sd[p + "ls1.gamma"]
    n2 = F.layer_norm(x, (dim,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], LN_EPS)
    m = mlp(sd, p + "mlp.", n2)
    y = x + m * sd[p + "ls2.gamma"]
    if taps is not None:
        taps[i] = {"attn_branch": a, "mlp_branch": m, "x_mid": x, "x_out": y}
    return y


@torch.no_grad()
def forward_features(sd, x, taps=None, blocks_out=None):
    """vision_transformer.py:221-236."""
    dim, depth, heads, _, _ = arch_from_state_dict(sd)
    t = prepare_tokens(sd, x)
    if taps is not None:
        taps["tokens"] = t
    for i in range(depth):
        t = block(sd, i, t, heads, taps)
        if blocks_out is not None:
            blocks_out.append(t)
    xn = F.layer_norm(t, (dim,), sd["norm.weight"], sd["norm.bias"], LN_EPS)
    return {
        "x_norm_clstoken": xn[:, 0],
        "x_norm_patchtokens": xn[:, 1:],
        "x_prenorm": t,
        "masks": None,
    }


@torch.no_grad()
def forward(sd, x, is_training=False):
    """vision_transformer.py:290-295."""
    ret = forward_features(sd, x)
    return ret if is_training else ret["x_norm_clstoken"]


def flops_per_image(n_patches, dim=384, depth=12, patch=14, mlp_ratio=4):
    """Closed-form algorithmic FLOPs (2*MAC) — SURVEY.md §8(a) formula,
    generalised over the width.  ViT-S/14: 451584*Np + 12*(3538944*N + 1536*N^2)."""
    n = n_patches + 1
    pe = 2 * n_patches * (3 * patch * patch) * dim
    lin = 2 * n * dim * dim * (3 + 1 + 2 * mlp_ratio)
    att = 4 * n * n * dim
    return pe + depth * (lin + att)
