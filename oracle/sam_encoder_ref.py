"""CPU restatement (torch fp32) of the reference SAM image encoder (BASELINE config 5).

TEST INFRASTRUCTURE — see oracle/__init__.py.  Functional style over the reference-layout state dict.

Reference anchors (paths relative to /root/reference/segment_anything/segment_anything):
  ImageEncoderViT.forward                 modeling/image_encoder.py:107-118
  Block.forward (window / global)         modeling/image_encoder.py:165-183
  Attention.forward                       modeling/image_encoder.py:217-235
  window_partition / window_unpartition   modeling/image_encoder.py:238-285
  get_rel_pos / add_decomposed_rel_pos    modeling/image_encoder.py:288-358
  PatchEmbed                              modeling/image_encoder.py:361-394
  MLPBlock, LayerNorm2d                   modeling/common.py:13-43
  encoder configurations                  build_sam.py:13-45,66-79 (LayerNorm eps 1e-6, window 14, rel-pos on)

Parity pin: oracle/gen_golden.py:gen_sam_encoder runs the reference's own ImageEncoderViT (loaded by file path)
on the same seeded weights and checks this file against it before writing tests/golden/sam_*.npz.
"""
import torch
import torch.nn.functional as F

LN_EPS = 1e-6  # build_sam.py:71, common.py:32


def rel_pos_table(q_size, k_size, rel_pos):
    """image_encoder.py:288-316: R[q, k, :] = rel_pos[(q - k) + (k_size - 1)] for equal sizes; a parameter of another
    length is resized with 1-d linear interpolation first."""
    max_rel = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != max_rel:
        rp = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=max_rel, mode="linear")
        rp = rp.reshape(-1, max_rel).permute(1, 0)
    else:
        rp = rel_pos
    q = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    k = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    rel = (q - k) + (k_size - 1) * max(q_size / k_size, 1.0)
    return rp[rel.long()]


def attention(sd, p, x, heads):
    """image_encoder.py:217-235 on x [B, H, W, C]; rel-pos bias uses the UNSCALED q (image_encoder.py:225-228)."""
    B, H, W, C = x.shape
    hd = C // heads
    qkv = F.linear(x, sd[p + "qkv.weight"], sd[p + "qkv.bias"]).reshape(B, H * W, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, B * heads, H * W, hd).unbind(0)
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    if p + "rel_pos_h" in sd:
        Rh = rel_pos_table(H, H, sd[p + "rel_pos_h"])
        Rw = rel_pos_table(W, W, sd[p + "rel_pos_w"])
        rq = q.reshape(B * heads, H, W, hd)
        rel_h = torch.einsum("bhwc,hkc->bhwk", rq, Rh)
        rel_w = torch.einsum("bhwc,wkc->bhwk", rq, Rw)
        attn = (attn.view(-1, H, W, H, W) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).view(-1, H * W, H * W)
    attn = attn.softmax(dim=-1)
    o = (attn @ v).view(B, heads, H, W, hd).permute(0, 2, 3, 1, 4).reshape(B, H, W, C)
    return F.linear(o, sd[p + "proj.weight"], sd[p + "proj.bias"])


def window_partition(x, ws):
    """image_encoder.py:238-259: zero-pad bottom/right to a multiple of ws, cut into ws x ws windows."""
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, C)
    return x, (Hp, Wp)


def window_unpartition(w, ws, pad_hw, hw):
    """image_encoder.py:262-285."""
    Hp, Wp = pad_hw
    H, W = hw
    B = w.shape[0] // (Hp * Wp // ws // ws)
    x = w.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    return x[:, :H, :W, :]


def layernorm2d(x, w, b):
    """common.py:27-43 on NCHW."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    return w[:, None, None] * ((x - u) / torch.sqrt(s + LN_EPS)) + b[:, None, None]


def forward(sd, img, heads, window, global_idx, taps=None, block_eps=LN_EPS):
    """image_encoder.py:107-118: img [B, 3, S, S] -> [B, out_chans, S/16, S/16].  `taps`: dict filled with block
    outputs [B, g, g, C] for the block indices it already holds as keys.  `block_eps`: eps of the blocks' norm1 / norm2
    (build_sam.py:71 passes 1e-6; the constructor's default norm_layer, image_encoder.py:27, has 1e-5); the neck's
    LayerNorm2d keeps its own 1e-6 (common.py:28)."""
    x = F.conv2d(img, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"],
                 stride=sd["patch_embed.proj.weight"].shape[-1]).permute(0, 2, 3, 1)
    if "pos_embed" in sd:
        x = x + sd["pos_embed"]
    C = x.shape[-1]
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    for i in range(depth):
        p = f"blocks.{i}."
        ws = 0 if i in global_idx else window
        shortcut = x
        y = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], block_eps)
        if ws > 0:
            H, W = y.shape[1], y.shape[2]
            y, pad_hw = window_partition(y, ws)      # the pad tokens are zeros AFTER norm1: their k, v are the qkv bias
        y = attention(sd, p + "attn.", y, heads)
        if ws > 0:
            y = window_unpartition(y, ws, pad_hw, (H, W))
        x = shortcut + y
        h = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], block_eps)
        h = F.linear(F.gelu(F.linear(h, sd[p + "mlp.lin1.weight"], sd[p + "mlp.lin1.bias"])),
                     sd[p + "mlp.lin2.weight"], sd[p + "mlp.lin2.bias"])
        x = x + h
        if taps is not None and i in taps:
            taps[i] = x
    y = F.conv2d(x.permute(0, 3, 1, 2), sd["neck.0.weight"])
    y = layernorm2d(y, sd["neck.1.weight"], sd["neck.1.bias"])
    y = F.conv2d(y, sd["neck.2.weight"], padding=1)
    return layernorm2d(y, sd["neck.3.weight"], sd["neck.3.bias"])
