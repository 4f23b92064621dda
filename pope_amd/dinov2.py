"""MI355X-native DINOv2 ViT: drop-in for the reference ``DinoVisionTransformer``.

Same constructor arguments, attribute names, state-dict keys (175 for ViT-S/14, loadable with
``strict=True`` from a reference checkpoint) and call surface as
``dinov2/dinov2/models/vision_transformer.py:45-295`` — ``model(x)``, ``model(x, is_training=True)``,
``forward_features``, ``get_intermediate_layers`` — but the forward pass is ONE call into the HIP
library (``pope_vit_forward_f32``): patch-embed GEMM (+cls +pos), depth x [LN, QKV GEMM, flash
attention, proj GEMM + LayerScale + residual, LN, FC1 GEMM + GELU, FC2 GEMM + LayerScale +
residual], final LN — all hand-written gfx950 kernels.  fp32 in HBM, fp32 accumulation; the contractions
run in ``model.precision``: "f16x3" (default: operands as f16 hi+lo pairs on the f16 matrix cores, three
MFMAs per product) or "f32" (fp32-in MFMA).

f16x3 has a range contract (|activation| < 8190, |weight| < 255.9, pope_hip.h) and it is guarded: weights
are checked when their planes are built, activations by the kernels that convert them (a device flag read
after the launch sequence).  ``model.on_overflow`` says what happens then: "rerun_f32" (default: warn once
and run that call again on the fp32 MFMA, in the same process) or "raise" (``PopeRangeError``).

torch.nn modules are used here only as parameter containers (so ``.to()``, ``.eval()``,
``state_dict()`` behave like the reference); they are never called.  There is no CPU fallback:
calling the model on CPU tensors raises.
"""
import warnings
import ctypes as C
import math
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from ._lib import PopeRangeError, check, on_device_of, ptr, require_cuda, stream_of


DEFAULT_PRECISION = "f16x3"


class _PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.patches_resolution = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class _Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias, proj_bias):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim, bias=proj_bias)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden, bias):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden, bias=bias)
        self.fc2 = nn.Linear(hidden, dim, bias=bias)


class _LayerScale(nn.Module):
    def __init__(self, dim, init_values):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))


class _Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias, proj_bias, ffn_bias, init_values):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, num_heads, qkv_bias, proj_bias)
        self.ls1 = _LayerScale(dim, init_values)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio), ffn_bias)
        self.ls2 = _LayerScale(dim, init_values)


class DinoVisionTransformer(nn.Module):
    """HIP-backed ViT with the reference's interface (vision_transformer.py:45-295)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4.0, qkv_bias=True, ffn_bias=True, proj_bias=True, drop_path_rate=0.0,
                 drop_path_uniform=False, init_values=None, ffn_layer="mlp", block_chunks=0, **_ignored):
        super().__init__()
        if in_chans != 3:
            raise NotImplementedError("pope_amd: the patch-embed kernel is built for 3-channel images")
        if embed_dim != num_heads * 64:
            raise NotImplementedError("pope_amd: attention kernel is built for head_dim 64 (all DINOv2 archs)")
        if ffn_layer != "mlp":
            raise NotImplementedError("pope_amd: only the 'mlp' FFN (ViT-S/B/L) is on the hot path")
        if not (qkv_bias and ffn_bias and proj_bias):
            raise NotImplementedError("pope_amd: reference configs enable all biases (ssl_default_config.yaml:71-82)")
        if block_chunks not in (0, None):
            raise NotImplementedError("pope_amd: block_chunks is an FSDP training device; eval config uses 0")
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens = 1
        self.n_blocks = depth
        self.num_heads = num_heads
        self.patch_size = patch_size
        self.patch_embed = _PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        # LayerScale is mandatory on this path (init_values=1e-5 in the eval config); None -> identity scale
        gamma0 = init_values if init_values else 1.0
        self.blocks = nn.ModuleList(
            [_Block(embed_dim, num_heads, mlp_ratio, qkv_bias, proj_bias, ffn_bias, gamma0) for _ in range(depth)])
        self.chunked_blocks = False
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Identity()
        self.mask_token = nn.Parameter(torch.zeros(1, embed_dim))
        self.init_weights()
        self._wcache = {}
        self._posb_cache = {}
        self._ws = None
        self._src = None
        self.profiler = None  # optional pope_amd.profiling.KernelProfiler (in-situ kernel timing)
        # arithmetic of the Linear layers and attention (_lib.PRECISIONS): "f16x3" (default, fp32-level results), "f32" (exact
        # fp32 MFMA chain) or "f16" (opt-in, BASELINE config 5's dtype: plain f16 operands, one MFMA per product, fp32
        # accumulation / residual stream / softmax / LayerNorm: results at f16 level, a few 1e-3 from the fp32 reference)
        self.precision = DEFAULT_PRECISION
        self.on_overflow = "rerun_f32"      # f16x3 range guard policy: "rerun_f32" | "raise"
        self.overflow_events = 0            # calls that left the f16x3 range (and were re-run or raised)
        for p in self.parameters():
            p.requires_grad_(False)  # inference-only kernels

    def init_weights(self):
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    # ---- cache management ----------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._wcache, self._posb_cache, self._ws, self._src = {}, {}, None, None
        return out

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self._wcache, self._posb_cache = {}, {}
        return out

    def _weights(self, precision=None):
        """ctypes weight struct of one arithmetic mode (cached per mode: the fp32 re-run of the range guard does not
        evict the f16x3 planes).  f16x3: the Linear / patch-embed weights as hi/lo planes, after a range check —
        |w| * 256 must be finite in f16; a checkpoint that breaches it runs on the fp32 MFMA (or raises)."""
        precision = precision or self.precision
        if self._src is None:
            self._src = _lib.param_slots(self)
        dev_ptr = _lib.slots_key(self._src)   # addresses + in-place versions of the tensors NOW in every parameter slot
        hit = self._wcache.get(precision)
        if hit is not None and hit[0] == dev_ptr:
            return hit[1]
        if precision in ("f16x3", "f16"):
            lin = [self.patch_embed.proj.weight] + [t for b in self.blocks for t in
                                                    (b.attn.qkv.weight, b.attn.proj.weight, b.mlp.fc1.weight, b.mlp.fc2.weight)]
            amax = float(torch.stack([t.detach().abs().max() for t in lin]).max())
            if not amax * _lib.PLANES_W_SCALE < _lib.F16_MAX:   # also catches NaN
                self.overflow_events += 1
                msg = (f"pope_amd: max |weight| = {amax:g} is outside the f16x3 range contract "
                       f"(|w| < {_lib.F16_MAX / _lib.PLANES_W_SCALE:g})")
                if self.on_overflow == "raise":
                    raise PopeRangeError(msg)
                warnings.warn(msg + "; this model runs with precision='f32'")
                w = self._weights("f32")
                self._wcache[precision] = self._wcache["f32"]
                return w
        tensors = []

        def P(t):
            t = t.detach()
            if t.dtype != torch.float32:
                raise TypeError("pope_amd kernels are fp32 (the reference path is fp32, SURVEY.md A15)")
            t = t.contiguous()
            tensors.append(t)
            return t.data_ptr()

        def planes(wt, plain=None):
            """f16 hi/lo planes of a Linear weight for the f16x3 GEMM (layout: pope_hip.h) — or, precision "f16", the
            weight as a plain f16 row-major matrix (value * 256) for the single-product GEMM."""
            if precision not in ("f16x3", "f16") or not wt.is_cuda or wt.shape[1] % 32:
                return None
            if (precision == "f16") if plain is None else plain:
                pl = (wt.detach().float() * _lib.PLANES_W_SCALE).half().contiguous()
            else:
                pl = _lib.to_planes(wt, _lib.PLANES_W_SCALE)
            tensors.append(pl)
            return pl.data_ptr()

        blocks = (_lib.VitBlockWeights * self.n_blocks)()
        for i, b in enumerate(self.blocks):
            blocks[i] = _lib.VitBlockWeights(
                P(b.norm1.weight), P(b.norm1.bias), P(b.attn.qkv.weight), P(b.attn.qkv.bias),
                P(b.attn.proj.weight), P(b.attn.proj.bias), P(b.ls1.gamma), P(b.norm2.weight), P(b.norm2.bias),
                P(b.mlp.fc1.weight), P(b.mlp.fc1.bias), P(b.mlp.fc2.weight), P(b.mlp.fc2.bias), P(b.ls2.gamma),
                planes(b.attn.qkv.weight), planes(b.mlp.fc1.weight), planes(b.mlp.fc2.weight),
                planes(b.attn.proj.weight))
        pw = self.patch_embed.proj.weight.detach().reshape(self.embed_dim, -1)
        patch_wp = None
        if precision in ("f16x3", "f16") and pw.is_cuda:   # [dim, 3*p*p] zero-padded to a multiple of 32 columns
            kp = (pw.shape[1] + 31) // 32 * 32                 # (the patch embed is f16x3 in both modes)
            patch_wp = planes(torch.nn.functional.pad(pw.float(), (0, kp - pw.shape[1])), plain=False)
        w = _lib.VitWeights(self.embed_dim, self.n_blocks, self.num_heads, self.patch_size,
                            self.blocks[0].mlp.fc1.weight.shape[0], P(pw),
                            P(self.norm.weight), P(self.norm.bias), blocks, _lib.PRECISIONS[precision], patch_wp)
        self._wcache[precision] = (dev_ptr, w, blocks, tensors)
        return w

    # ---- positional encoding (host plumbing, cached per (H, W)) -----------------------------
    def interpolate_pos_encoding(self, x, w, h):
        """vision_transformer.py:165-189.  As in the reference, `w` is the image HEIGHT and `h`
        the image WIDTH (swapped names).  Evaluated in fp32 on the host so the
        bicubic weights are bit-identical to the reference's CPU path."""
        npatch = x.shape[1] - 1
        N = self.pos_embed.shape[1] - 1
        if npatch == N and w == h:
            return self.pos_embed
        pos = self.pos_embed.detach().float().cpu()
        cls_pos, patch_pos = pos[:, 0], pos[:, 1:]
        dim = pos.shape[-1]
        g = int(math.sqrt(N))
        w0, h0 = w // self.patch_size + 0.1, h // self.patch_size + 0.1
        patch_pos = F.interpolate(patch_pos.reshape(1, g, g, dim).permute(0, 3, 1, 2),
                                  scale_factor=(w0 / math.sqrt(N), h0 / math.sqrt(N)), mode="bicubic")
        assert int(w0) == patch_pos.shape[-2] and int(h0) == patch_pos.shape[-1]
        patch_pos = patch_pos.permute(0, 2, 3, 1).reshape(1, -1, dim)
        return torch.cat((cls_pos.unsqueeze(0), patch_pos), dim=1).to(self.pos_embed.device)

    def _posb(self, H, W, ntok):
        """[ntok, dim] table added by the patch-embed epilogue: row 0 = cls_token + pos[0],
        row n = conv bias + pos[n]."""
        key = (H, W) + _lib.params_key((self.pos_embed, self.cls_token, self.patch_embed.proj.bias))
        if key not in self._posb_cache:
            if len(self._posb_cache) >= 32:    # stale generations of an updated model must not pile up
                self._posb_cache = {}
            dummy = torch.empty(1, ntok, 1)
            pos = self.interpolate_pos_encoding(dummy, H, W)[0].detach().float().to(self.pos_embed.device)
            posb = pos + self.patch_embed.proj.bias.detach()[None, :]
            posb[0] = pos[0] + self.cls_token.detach()[0, 0]
            self._posb_cache[key] = posb.contiguous()
        return self._posb_cache[key]

    def _workspace(self, nbytes, device):
        """Scratch of the launch sequence, one per (device, stream): forwards on different streams may overlap."""
        if not isinstance(self._ws, dict):
            self._ws = {}
        key = (str(device), torch.cuda.current_stream(device).cuda_stream)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return ws

    # ---- forward --------------------------------------------------------------------------
    def _run(self, x, taps=(), out_norm=None, range_flag=None, precision=None):
        """One launch sequence (or several, for batches beyond the 32-bit offsets).  `range_flag`: int32[1] device
        tensor owned by the caller, who then checks it at a synchronisation point of its own (PairPipeline does);
        without it this call reads its own flag right after the launches — the one host synchronisation of a
        direct call, where the reference's callers synchronise anyway (`.item()` / `.cpu()` per proposal,
        eval_linemod_json.py:95,115) — and applies `self.on_overflow`."""
        require_cuda(x, "DinoVisionTransformer.forward")
        require_cuda(self.cls_token, "DinoVisionTransformer weights")
        if x.dtype != torch.float32:
            raise TypeError("pope_amd DINOv2 expects float32 images")
        x = x.contiguous()
        B, nc, H, W = x.shape
        p = self.patch_size
        assert H % p == 0, f"Input image height {H} is not a multiple of patch height {p}"
        assert W % p == 0, f"Input image width {W} is not a multiple of patch width: {p}"
        ntok = 1 + (H // p) * (W // p)
        dim = self.embed_dim
        precision = precision or self.precision
        if B == 0:   # an empty batch is legal in the reference (every op is a no-op on it): empty outputs, no launch
            e = torch.empty(0, ntok, dim, device=x.device, dtype=torch.float32)
            return e, (out_norm if out_norm is not None else e.clone()), [e.clone() for _ in taps]
        # The kernels address a launch sequence's activations through 32-bit byte offsets: larger batches are run as
        # several sequences writing into slices of the same outputs (identical results: images are independent).
        widest = max(4 * dim, int(self.blocks[0].mlp.fc1.weight.shape[0]))
        max_b = max(1, ((1 << 32) - (1 << 20)) // ((ntok * widest + 256 * widest) * 4))
        max_b = min(max_b, getattr(self, "_max_batch", None) or max_b)   # test hook
        if B > max_b:
            x_pre = torch.empty(B, ntok, dim, device=x.device, dtype=torch.float32)
            x_norm = out_norm if out_norm is not None else torch.empty(B, ntok, dim, device=x.device, dtype=torch.float32)
            tap_out = [torch.empty(B, ntok, dim, device=x.device, dtype=torch.float32) for _ in taps]
            for s in range(0, B, max_b):
                pre_s, _, taps_s = self._run(x[s:s + max_b], taps, out_norm=x_norm[s:s + max_b], range_flag=range_flag,
                                             precision=precision)
                x_pre[s:s + max_b] = pre_s
                for dst, src in zip(tap_out, taps_s):
                    dst[s:s + max_b] = src
            return x_pre, x_norm, tap_out
        with on_device_of(x):
            w = self._weights(precision)
            guarded = w.precision in (_lib.PREC_F16X3, _lib.PREC_F16)
            own_flag = None
            if guarded and range_flag is None:
                own_flag = range_flag = torch.zeros(1, dtype=torch.int32, device=x.device)
            flag_ptr = C.c_void_p(range_flag.data_ptr()) if (guarded and range_flag is not None) else None
            posb = self._posb(H, W, ntok)
            L = _lib.lib()
            nbytes = L.pope_vit_workspace_bytes(B, ntok, dim, w.hidden)
            ws = self._workspace(nbytes, x.device)
            x_pre = torch.empty(B, ntok, dim, device=x.device, dtype=torch.float32)
            if out_norm is None:
                x_norm = torch.empty(B, ntok, dim, device=x.device, dtype=torch.float32)
            else:  # caller-provided destination (a batch slice of a larger buffer: no concatenation afterwards)
                if out_norm.shape != (B, ntok, dim) or out_norm.dtype != torch.float32 or not out_norm.is_contiguous() \
                        or out_norm.device != x.device:
                    raise ValueError("out_norm must be a contiguous float32 [B, ntok, dim] tensor on the input's device")
                x_norm = out_norm
            tap_out = [torch.empty(B, ntok, dim, device=x.device, dtype=torch.float32) for _ in taps]
            tap_blocks = (C.c_int * max(1, len(taps)))(*taps)
            tap_ptrs = (C.c_void_p * max(1, len(taps)))(*[t.data_ptr() for t in tap_out])
            slot = self.profiler.next_slot() if (self.profiler is not None and not taps) else None
            if slot is not None:
                off, ev, cap, kinds = slot
                n_launch = C.c_int()
                check(L.pope_vit_forward_profiled_mask_f32(C.byref(w), ptr(x), B, H, W, ptr(posb), ptr(x_pre), ptr(x_norm),
                                                           C.c_void_p(ws.data_ptr()), ws.numel(), flag_ptr,
                                                           stream_of(x.device), ev, cap, kinds, C.byref(n_launch),
                                                           self.profiler.mask),
                      "pope_vit_forward_profiled_mask_f32")
                self.profiler.commit(off, n_launch.value)
            else:
                check(L.pope_vit_forward_f32(C.byref(w), ptr(x), B, H, W, ptr(posb), ptr(x_pre), ptr(x_norm),
                                             len(taps), tap_blocks, tap_ptrs, C.c_void_p(ws.data_ptr()), ws.numel(),
                                             flag_ptr, stream_of(x.device)), "pope_vit_forward_f32")
        if own_flag is not None:
            bits = int(own_flag.item())  # the direct call's synchronisation point
            if bits:
                self.range_overflow(bits)  # raises under on_overflow == "raise"
                return self._run(x, taps, out_norm=out_norm, precision="f32")
        return x_pre, x_norm, tap_out

    def range_overflow(self, bits):
        """Account for a call whose activations left the f16x3 range (`bits`: POPE_RANGE_* word) and apply the policy:
        raise, or warn (once per model) so that the caller re-runs the work with precision="f32"."""
        self.overflow_events += 1
        msg = (f"pope_amd: f16x3 range contract breached ({_lib.describe_range_bits(bits)}: |value| * scale "
               f">= {_lib.F16_MAX:g} or non-finite)")
        if self.on_overflow == "raise":
            raise PopeRangeError(msg)
        if self.overflow_events == 1:
            warnings.warn(msg + "; re-running on the fp32 MFMA (precision='f32')")

    def prepare_tokens_with_masks(self, x, masks=None):
        """vision_transformer.py:191-200 (patch embed + cls + pos) as one HIP kernel."""
        if masks is not None:
            raise NotImplementedError("pope_amd: iBOT mask tokens are training-only (out of the hot path)")
        from . import ops
        B, nc, H, W = x.shape
        ntok = 1 + (H // self.patch_size) * (W // self.patch_size)
        require_cuda(x, "prepare_tokens_with_masks")
        prec = "f16x3" if self._weights().precision in (_lib.PREC_F16X3, _lib.PREC_F16) else "f32"   # weight range check included
        args = (x, self.patch_embed.proj.weight.detach(), self._posb(H, W, ntok), self.patch_size)
        if prec == "f32":
            return ops.patch_embed(*args, precision="f32")
        flag = torch.zeros(1, dtype=torch.int32, device=x.device)
        out = ops.patch_embed(*args, precision="f16x3", range_flag=flag)
        bits = int(flag.item())
        if bits:
            self.range_overflow(bits)
            out = ops.patch_embed(*args, precision="f32")
        return out

    def forward_features(self, x, masks=None, out_norm=None, range_flag=None, precision=None):
        """Extensions: `out_norm` = preallocated [B, ntok, dim] buffer that receives the final-norm tokens;
        `range_flag` / `precision`: see `_run` (deferred f16x3 range check of a batching caller)."""
        if isinstance(x, list):
            raise NotImplementedError("pope_amd: nested-tensor (list) inputs are a training-time xformers path")
        if masks is not None:
            raise NotImplementedError("pope_amd: iBOT mask tokens are training-only (out of the hot path)")
        x_pre, x_norm, _ = self._run(x, out_norm=out_norm, range_flag=range_flag, precision=precision)
        return {"x_norm_clstoken": x_norm[:, 0], "x_norm_patchtokens": x_norm[:, 1:],
                "x_prenorm": x_pre, "masks": masks}

    def get_intermediate_layers(self, x, n=1, reshape=False, return_class_token=False, norm=True):
        """vision_transformer.py:264-288."""
        from . import ops
        total = self.n_blocks
        take = list(range(total - n, total)) if isinstance(n, int) else list(n)
        _, _, outs = self._run(x, taps=take)
        assert len(outs) == len(take), f"only {len(outs)} / {len(take)} blocks found"
        if norm:
            outs = [ops.layernorm(o, self.norm.weight.detach(), self.norm.bias.detach(), 1e-6) for o in outs]
        cls = [o[:, 0] for o in outs]
        outs = [o[:, 1:] for o in outs]
        if reshape:
            B, _, H, W = x.shape
            outs = [o.reshape(B, H // self.patch_size, W // self.patch_size, -1).permute(0, 3, 1, 2).contiguous()
                    for o in outs]
        if return_class_token:
            return tuple(zip(outs, cls))
        return tuple(outs)

    def forward(self, *args, is_training=False, **kwargs):
        ret = self.forward_features(*args, **kwargs)
        return ret if is_training else self.head(ret["x_norm_clstoken"])


def vit_small(patch_size=16, **kw):
    return DinoVisionTransformer(patch_size=patch_size, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4, **kw)


def vit_base(patch_size=16, **kw):
    return DinoVisionTransformer(patch_size=patch_size, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, **kw)


def vit_large(patch_size=16, **kw):
    return DinoVisionTransformer(patch_size=patch_size, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4, **kw)


# the eval config of the reference (configs/eval/vits14_pretrain.yaml + ssl_default_config.yaml:71-82)
build_vits14 = partial(vit_small, patch_size=14, img_size=518, init_values=1e-5, ffn_layer="mlp", block_chunks=0)
