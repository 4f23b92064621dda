"""SAM image encoder on the HIP library (BASELINE config 5).

Drop-in for `segment_anything.modeling.image_encoder.ImageEncoderViT`
(segment_anything/segment_anything/modeling/image_encoder.py:17-118): same constructor arguments, same state-dict
keys (so `build_sam.py:102-105` checkpoints load with strict=True), same forward contract
(`[B, 3, img, img]` fp32 -> `[B, out_chans, img/16, img/16]`).  The modules below are parameter containers only: the
forward pass is ONE C-ABI call (`pope_sam_encoder_forward_f32`, pope_amd/csrc/sam.hip) — there is no torch fallback.
"""
import ctypes as C
from typing import Optional, Tuple, Type

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from ._lib import check, on_device_of, ptr, require_cuda, stream_of
from .dinov2 import PopeRangeError


class MLPBlock(nn.Module):
    """common.py:13-25 (parameters only)."""

    def __init__(self, embedding_dim, mlp_dim, act=nn.GELU):
        super().__init__()
        if act is not nn.GELU:
            raise NotImplementedError("pope_amd SAM encoder: the MLP activation is the erf GELU (common.py:17)")
        self.lin1 = nn.Linear(embedding_dim, mlp_dim)
        self.lin2 = nn.Linear(mlp_dim, embedding_dim)


class LayerNorm2d(nn.Module):
    """common.py:27-43 (parameters only)."""

    def __init__(self, num_channels, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))
        self.eps = eps


class Attention(nn.Module):
    """image_encoder.py:185-235 (parameters only)."""

    def __init__(self, dim, num_heads=8, qkv_bias=True, use_rel_pos=False, rel_pos_zero_init=True, input_size=None):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.use_rel_pos = use_rel_pos
        self.input_size = input_size
        if use_rel_pos:
            assert input_size is not None, "Input size must be provided if using relative positional encoding."
            self.rel_pos_h = nn.Parameter(torch.zeros(2 * input_size[0] - 1, head_dim))
            self.rel_pos_w = nn.Parameter(torch.zeros(2 * input_size[1] - 1, head_dim))


class Block(nn.Module):
    """image_encoder.py:121-183 (parameters only)."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=True, norm_layer=nn.LayerNorm, act_layer=nn.GELU,
                 use_rel_pos=False, rel_pos_zero_init=True, window_size=0, input_size=None):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, use_rel_pos=use_rel_pos,
                              rel_pos_zero_init=rel_pos_zero_init,
                              input_size=input_size if window_size == 0 else (window_size, window_size))
        self.norm2 = norm_layer(dim)
        self.mlp = MLPBlock(embedding_dim=dim, mlp_dim=int(dim * mlp_ratio), act=act_layer)
        self.window_size = window_size


class PatchEmbed(nn.Module):
    """image_encoder.py:361-394 (parameters only)."""

    def __init__(self, kernel_size=(16, 16), stride=(16, 16), padding=(0, 0), in_chans=3, embed_dim=768):
        super().__init__()
        if tuple(kernel_size) != tuple(stride) or tuple(padding) != (0, 0) or in_chans != 3 or kernel_size[0] != kernel_size[1]:
            raise NotImplementedError("pope_amd SAM encoder: square non-overlapping RGB patches only (build_sam.py:66-79)")
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=kernel_size, stride=stride, padding=padding)


def get_rel_pos(q_size, k_size, rel_pos):
    """image_encoder.py:288-316: the gathered table R[q, k, :] (host side, once per model)."""
    max_rel_dist = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != max_rel_dist:
        rp = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=max_rel_dist, mode="linear")
        rp = rp.reshape(-1, max_rel_dist).permute(1, 0)
    else:
        rp = rel_pos
    q = torch.arange(q_size, device=rel_pos.device)[:, None] * max(k_size / q_size, 1.0)
    k = torch.arange(k_size, device=rel_pos.device)[None, :] * max(q_size / k_size, 1.0)
    rel = (q - k) + (k_size - 1) * max(q_size / k_size, 1.0)
    return rp[rel.long()]


class ImageEncoderViT(nn.Module):
    def __init__(self, img_size: int = 1024, patch_size: int = 16, in_chans: int = 3, embed_dim: int = 768, depth: int = 12,
                 num_heads: int = 12, mlp_ratio: float = 4.0, out_chans: int = 256, qkv_bias: bool = True,
                 norm_layer: Type[nn.Module] = nn.LayerNorm, act_layer: Type[nn.Module] = nn.GELU, use_abs_pos: bool = True,
                 use_rel_pos: bool = False, rel_pos_zero_init: bool = True, window_size: int = 0,
                 global_attn_indexes: Tuple[int, ...] = ()) -> None:
        super().__init__()
        self.img_size = img_size
        self.patch_size = patch_size
        self.embed_dim, self.depth, self.num_heads, self.out_chans = embed_dim, depth, num_heads, out_chans
        self.window_size = window_size
        self.global_attn_indexes = tuple(global_attn_indexes)
        self.patch_embed = PatchEmbed(kernel_size=(patch_size, patch_size), stride=(patch_size, patch_size), in_chans=in_chans,
                                      embed_dim=embed_dim)
        self.pos_embed: Optional[nn.Parameter] = None
        grid = img_size // patch_size
        if use_abs_pos:
            self.pos_embed = nn.Parameter(torch.zeros(1, grid, grid, embed_dim))
        self.blocks = nn.ModuleList()
        for i in range(depth):
            self.blocks.append(Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                     norm_layer=norm_layer, act_layer=act_layer, use_rel_pos=use_rel_pos,
                                     rel_pos_zero_init=rel_pos_zero_init,
                                     window_size=window_size if i not in global_attn_indexes else 0, input_size=(grid, grid)))
        self.neck = nn.Sequential(nn.Conv2d(embed_dim, out_chans, kernel_size=1, bias=False), LayerNorm2d(out_chans),
                                  nn.Conv2d(out_chans, out_chans, kernel_size=3, padding=1, bias=False), LayerNorm2d(out_chans))
        eps_b = {float(m.eps) for b in self.blocks for m in (b.norm1, b.norm2)}
        eps_n = {float(self.neck[1].eps), float(self.neck[3].eps)}
        if len(eps_b) != 1 or len(eps_n) != 1:
            raise NotImplementedError("pope_amd SAM encoder: one LayerNorm eps for all blocks and one for the neck")
        self.block_eps, self.neck_eps = eps_b.pop(), eps_n.pop()   # build_sam.py:71 passes 1e-6; nn.LayerNorm's default is 1e-5
        # "f16x3" (default): fp32 operands as hi + lo f16, three MFMAs per product — fp32-level results (1e-5 from the
        # reference).  "f16": BASELINE config 5's dtype — plain f16 operands, ONE MFMA per product, fp32 accumulation and an
        # fp32 residual stream / softmax / LayerNorm; results at f16 level (a few 1e-3 from the fp32 reference).
        # "f32": every contraction on the fp32 MFMA (sam_f32.hip) — the reference's arithmetic, no range contract, ~10x slower:
        # what a range-guard event of the other two modes is re-run in.
        self.precision = "f16x3"
        self.on_overflow = "rerun_f32"  # f16x3 / f16 range guard: "rerun_f32" (warn, run the call again on the fp32 MFMA) | "raise"
        self.overflow_events = 0
        self.max_batch = 16            # images per launch sequence: 256 row tiles = whole rounds of the 256 x 256 GEMM tiles at ViT-H (32-bit offsets of the GEMMs cap it: 4096 x 5120 x 4 B x B < 4 GiB)
        self._wcache = {}
        self._src = None
        self._ws = None

    # ---- host plumbing ---------------------------------------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._wcache, self._ws, self._src = {}, None, None
        return out

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self._wcache = {}
        return out

    def _sources(self):
        if self._src is None:
            self._src = _lib.param_slots(self)
        return self._src

    def _weights(self, precision=None):
        precision = precision or self.precision
        if precision not in ("f16x3", "f16", "f32"):
            raise ValueError(f"ImageEncoderViT.precision must be 'f16x3', 'f16' or 'f32', not {precision!r}")
        plain, f32 = precision == "f16", precision == "f32"
        dev_ptr = _lib.slots_key(self._sources())   # addresses + in-place versions of the tensors NOW in every parameter slot
        hit = self._wcache.get(precision)
        if hit is not None and hit[0] == dev_ptr:
            return hit[1]
        keep = []

        def P(t):
            t = t.detach()
            if t.dtype != torch.float32:
                raise TypeError("pope_amd kernels take fp32 parameters (f16x3 arithmetic inside)")
            t = t.contiguous()
            keep.append(t)
            return C.c_void_p(t.data_ptr())

        lin = [self.patch_embed.proj.weight, self.neck[0].weight, self.neck[2].weight] + [
            t for b in self.blocks for t in (b.attn.qkv.weight, b.attn.proj.weight, b.mlp.lin1.weight, b.mlp.lin2.weight)]
        amax = float(torch.stack([t.detach().abs().max() for t in lin]).max())
        if not f32 and not amax * _lib.PLANES_W_SCALE < _lib.F16_MAX:
            self.overflow_events += 1
            msg = (f"pope_amd: max |weight| = {amax:g} is outside the f16x3 range contract "
                   f"(|w| < {_lib.F16_MAX / _lib.PLANES_W_SCALE:g})")
            if self.on_overflow == "raise":
                raise PopeRangeError(msg)
            import warnings
            warnings.warn(msg + "; this encoder runs with precision='f32'")
            w = self._weights("f32")
            self._wcache[precision] = self._wcache["f32"]
            return w

        def WP(t2d):
            k = t2d.shape[1]
            assert k % (64 if plain else 32) == 0
            if f32:     # the fp32 matrix itself
                pl = t2d.detach().float().contiguous()
            elif plain:   # f16 row-major, value * 256
                pl = (t2d.detach().float() * _lib.PLANES_W_SCALE).half().contiguous()
            else:
                pl = _lib.to_planes(t2d.detach().float(), _lib.PLANES_W_SCALE)
            keep.append(pl)
            return C.c_void_p(pl.data_ptr())

        dim, hd = self.embed_dim, self.embed_dim // self.num_heads
        grid = self.img_size // self.patch_size
        dev = self.patch_embed.proj.weight.device
        blocks = (_lib.SamBlockWeights * self.depth)()
        for i, b in enumerate(self.blocks):
            w = blocks[i]
            size = grid if b.window_size == 0 else b.window_size
            if b.attn.use_rel_pos:
                rh = get_rel_pos(size, size, b.attn.rel_pos_h.detach().float())
                rw = get_rel_pos(size, size, b.attn.rel_pos_w.detach().float())
            else:   # use_rel_pos = False: zero tables, the widened columns contribute nothing
                rh = rw = torch.zeros(size, size, hd, device=dev)
            if b.attn.qkv.bias is None:
                raise NotImplementedError("pope_amd SAM encoder: qkv_bias=True only (build_sam.py:74)")
            w.norm1_w, w.norm1_b = P(b.norm1.weight), P(b.norm1.bias)
            w.qkv_wp, w.qkv_b = WP(b.attn.qkv.weight), P(b.attn.qkv.bias)
            w.proj_wp, w.proj_b = WP(b.attn.proj.weight), P(b.attn.proj.bias)
            w.rel_h, w.rel_w = P(rh), P(rw)
            w.norm2_w, w.norm2_b = P(b.norm2.weight), P(b.norm2.bias)
            w.fc1_wp, w.fc1_b = WP(b.mlp.lin1.weight), P(b.mlp.lin1.bias)
            w.fc2_wp, w.fc2_b = WP(b.mlp.lin2.weight), P(b.mlp.lin2.bias)
            w.global_attn = int(b.window_size == 0)
        s = _lib.SamEncoderWeights()
        s.img, s.patch, s.dim, s.depth, s.heads = self.img_size, self.patch_size, dim, self.depth, self.num_heads
        s.hidden, s.out_chans, s.window = self.blocks[0].mlp.lin1.out_features, self.out_chans, self.window_size
        s.precision = _lib.PRECISIONS[precision]
        s.patch_wp, s.patch_b = WP(self.patch_embed.proj.weight.reshape(dim, -1)), P(self.patch_embed.proj.bias)
        s.pos = P(self.pos_embed.reshape(grid * grid, dim)) if self.pos_embed is not None else None
        s.ones = P(torch.ones(dim, device=dev))
        s.blocks_host = C.cast(blocks, C.POINTER(_lib.SamBlockWeights))
        s.neck0_wp = WP(self.neck[0].weight.reshape(self.out_chans, dim))
        s.neck1_w, s.neck1_b = P(self.neck[1].weight), P(self.neck[1].bias)
        s.neck2_wp = WP(self.neck[2].weight.permute(0, 2, 3, 1).reshape(self.out_chans, -1))   # taps (ky, kx, channel)
        s.neck3_w, s.neck3_b = P(self.neck[3].weight), P(self.neck[3].bias)
        s.block_eps, s.neck_eps = self.block_eps, self.neck_eps
        keep.append(blocks)
        self._wcache[precision] = (dev_ptr, s, keep)
        return s

    def _workspace(self, w, b, device):
        need = int(_lib.lib().pope_sam_encoder_workspace_bytes(C.byref(w), b))
        if need <= 0:
            raise ValueError("pope_amd SAM encoder: unsupported geometry (head_dim 64 / 80, dim % 128 == 0, out_chans % 256 == 0)")
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    def _run(self, x, taps=(), precision=None):
        require_cuda(x, "ImageEncoderViT")
        if x.dtype != torch.float32:
            raise TypeError(f"ImageEncoderViT: expected float32, got {x.dtype}")
        x = x.contiguous()
        b, c, h, wd = x.shape
        if c != 3 or h != self.img_size or wd != self.img_size:
            raise ValueError(f"ImageEncoderViT: expected [B, 3, {self.img_size}, {self.img_size}], got {tuple(x.shape)}")
        grid = self.img_size // self.patch_size
        out = torch.empty(b, self.out_chans, grid, grid, device=x.device, dtype=torch.float32)
        tap_out = [torch.empty(b, grid, grid, self.embed_dim, device=x.device, dtype=torch.float32) for _ in taps]
        if b == 0:
            return out, tap_out
        w = self._weights(precision)
        flag = torch.zeros(1, dtype=torch.int32, device=x.device)
        for s in range(0, b, self.max_batch):
            n = min(self.max_batch, b - s)
            ws = self._workspace(w, n, x.device)
            tb = (C.c_int * max(1, len(taps)))(*taps)
            tp = (C.c_void_p * max(1, len(taps)))(*[t[s:s + n].data_ptr() for t in tap_out])
            with on_device_of(x):
                check(_lib.lib().pope_sam_encoder_forward_f32(C.byref(w), ptr(x[s:s + n]), n, ptr(out[s:s + n]), len(taps), tb, tp,
                                                              ptr(ws), ws.numel(), C.c_void_p(flag.data_ptr()),
                                                              stream_of(x.device)), "pope_sam_encoder_forward_f32")
        bits = int(flag.item())   # one sync per call: the range guard of every planes producer of the launch sequence
        if bits:
            self.overflow_events += 1
            what = ", ".join(v for k, v in _lib.RANGE_BITS.items() if bits & k)
            msg = f"pope_amd SAM encoder: a value left the f16x3 range ({what})"
            if self.on_overflow == "raise":
                raise PopeRangeError(msg)
            import warnings
            warnings.warn(msg + "; re-running the call on the fp32 MFMA")
            return self._run(x, taps, precision="f32")
        return out, tap_out

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._run(x)[0]

    def forward_with_taps(self, x, blocks):
        """(out, [block outputs [B, g, g, dim]]) — the parity tests' view of the residual stream."""
        return self._run(x, tuple(int(i) for i in blocks))
