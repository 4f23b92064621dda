"""LoFTR stages either side of the HIP coarse matcher: local-feature CNN, sinusoidal
position code, linear-attention transformer, fine window preprocessing and sub-pixel refinement.  Every stage but the
position code (a cached host table) is a call into libpope_hip.so (conv.hip, loftr.hip, fine.hip) and nothing else: CPU
tensors and shapes the kernels do not cover raise `PopeHipError` — there is no second backend.

Arithmetic and its guard, the same policy as the DINOv2 path (pope_amd/dinov2.py): contractions run as f16x3 (fp32 operands
as hi + lo f16, three MFMAs per product, fp32 accumulate); every producer of f16 operands raises a device flag when a scaled
value leaves the f16 range, and the affected call is then RE-RUN ON THE fp32 MFMA (`POPE_PREC_F32_MFMA`: gemm_f32.hip, no
range contract) with a warning — or raises `PopeRangeError` when `on_overflow = "raise"`.  Weights outside the f16x3 weight
range select the fp32 path at load time.

Every module keeps the reference's parameter names and shapes so that `weights/matcher.pth` loads with
strict=True (211 keys: backbone.* 107, loftr_coarse.* 80, fine_preprocess.* 4, loftr_fine.* 20), but the
computation is organised for inference on one big GPU:
  * eval-mode BatchNorm is folded into the preceding convolution once per weight version (one pass over the
    activations instead of two; resnet_fpn.py:27-40,60-63,72-84 keep conv and BN separate);
  * fine windows are gathered only at the M matched cells instead of unfolding every window of the
    1/2-resolution map and indexing afterwards (fine_preprocess.py:44-51 materialises [n, L, 25, 128]);
  * the position code is generated for the requested grid, not sliced from a 256x256 buffer.
Derived data (folded filters, weight planes) are cached under the storage address AND the in-place version counter of
every source parameter (`_lib.params_key`): `p.copy_()`, an optimizer step or `w[0, 0] = x` refreshes them.
"""
import ctypes as C
import math
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from ._lib import PopeHipError, PopeRangeError, require_cuda

ON_OVERFLOW = "rerun_f32"   # module-level default of the f16x3 range-guard policy: "rerun_f32" | "raise"

# While `Matcher` captures its front end into a HIP graph nothing may synchronise: the range-flag words of the captured
# launches are collected here and read by the Matcher after each replay (matcher.py:_features_graphed).
DEFERRED_FLAGS = None


def _read_flags(flags):
    """OR of f16x3 range-guard words (int32[1] device tensors): one synchronisation — or 0 now and the check deferred
    to the capturing Matcher."""
    if DEFERRED_FLAGS is not None and torch.cuda.is_current_stream_capturing():
        DEFERRED_FLAGS.extend(flags)
        return 0
    return int(torch.stack(list(flags)).max()) if len(flags) > 1 else int(flags[0].item())


def _overflow(what, bits, policy):
    msg = f"pope_amd: f16x3 range contract breached in the {what} ({_lib.describe_range_bits(bits)})"
    if policy == "raise":
        raise PopeRangeError(msg)
    warnings.warn(msg + "; re-running it on the fp32 MFMA")


def _weights_fit(tensors):
    """|w| * 256 finite in f16 for every matrix (NaN fails too)."""
    amax = max(float(t.detach().abs().max()) for t in tensors)
    return amax * _lib.PLANES_W_SCALE < _lib.F16_MAX


def _fold_bn(conv_w, bn):
    """Filter and bias of conv -> BatchNorm(eval) as one convolution."""
    g = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
    return conv_w * g.view(-1, 1, 1, 1), bn.bias - bn.running_mean * g


# ------------------------------------------------------------------------------------------------ CNN
class BasicBlock(nn.Module):
    """resnet_fpn.py:15-40: relu(x' + bn2(conv2(relu(bn1(conv1(x)))))), x' = 1x1 stride-s conv + BN if s != 1.
    Parameter container; the arithmetic is part of pope_resnetfpn_forward_f32."""

    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.stride = stride
        self.conv1 = nn.Conv2d(in_planes, planes, 3, stride, 1, bias=False)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1:
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, 1, stride, bias=False),
                                            nn.BatchNorm2d(planes))

    def folded(self):
        f = [_fold_bn(self.conv1.weight, self.bn1), _fold_bn(self.conv2.weight, self.bn2)]
        if self.downsample is not None:
            f.append(_fold_bn(self.downsample[0].weight, self.downsample[1]))
        return f


class ResNetFPN_8_2(nn.Module):
    """resnet_fpn.py:43-118.  [B,1,H,W] -> coarse [B,256,H/8,W/8], fine [B,128,H/2,W/2]: ONE C-ABI call
    (pope_resnetfpn_forward_f32, conv.hip) — all 22 convolutions as GEMMs over zero-bordered NHWC pixel rows."""

    def __init__(self, config):
        super().__init__()
        d0 = config["initial_dim"]
        d1, d2, d3 = config["block_dims"]
        if (d0, d1, d2, d3) != (128, 128, 196, 256):
            raise NotImplementedError("pope_amd: the HIP ResNet-FPN is built for the released widths 128 / 196 / 256 (cvpr_ds_config.py:15-17)")
        self.conv1 = nn.Conv2d(1, d0, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(d0)
        self.layer1 = nn.Sequential(BasicBlock(d0, d1, 1), BasicBlock(d1, d1, 1))
        self.layer2 = nn.Sequential(BasicBlock(d1, d2, 2), BasicBlock(d2, d2, 1))
        self.layer3 = nn.Sequential(BasicBlock(d2, d3, 2), BasicBlock(d3, d3, 1))
        self.layer3_outconv = nn.Conv2d(d3, d3, 1, bias=False)
        self.layer2_outconv = nn.Conv2d(d2, d3, 1, bias=False)
        self.layer2_outconv2 = nn.Sequential(nn.Conv2d(d3, d3, 3, 1, 1, bias=False), nn.BatchNorm2d(d3),
                                             nn.LeakyReLU(), nn.Conv2d(d3, d2, 3, 1, 1, bias=False))
        self.layer1_outconv = nn.Conv2d(d1, d2, 1, bias=False)
        self.layer1_outconv2 = nn.Sequential(nn.Conv2d(d2, d2, 3, 1, 1, bias=False), nn.BatchNorm2d(d2),
                                             nn.LeakyReLU(), nn.Conv2d(d2, d1, 3, 1, 1, bias=False))
        for m in self.modules():  # resnet_fpn.py:87-92
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        self._hip = None
        self._src = None
        self.on_overflow = None    # None: the module-level ON_OVERFLOW

    def invalidate(self):
        self._hip, self._src = None, None

    def _apply(self, fn, *a, **k):  # .to()/.cuda()/.float() move the parameters: rebuild lazily
        self._hip, self._src = None, None
        return super()._apply(fn, *a, **k)

    def _matrices(self):
        """The 22 convolutions in the order of pope_hip.h as fp32 matrices [Cout, taps x Cin rounded up to 32] (tap-major,
        BatchNorm folded) and their folded biases."""
        with torch.no_grad():
            convs = [_fold_bn(self.conv1.weight, self.bn1)]
            for layer in (self.layer1, self.layer2, self.layer3):
                for blk in layer:
                    convs += blk.folded()                # conv1, conv2[, downsample]
            convs += [(self.layer3_outconv.weight, None), (self.layer2_outconv.weight, None),
                      _fold_bn(self.layer2_outconv2[0].weight, self.layer2_outconv2[1]), (self.layer2_outconv2[3].weight, None),
                      (self.layer1_outconv.weight, None),
                      _fold_bn(self.layer1_outconv2[0].weight, self.layer1_outconv2[1]), (self.layer1_outconv2[3].weight, None)]
            assert len(convs) == 22

            def mat(w):   # [Cout, Cin, kh, kw] -> [Cout, kh * kw * Cp]
                co, ci, kh, kw = w.shape
                if ci == 1:   # the 7x7 stem: 49 taps of one channel, zero-filled to 64 columns
                    return F.pad(w.reshape(co, kh * kw), (0, 64 - kh * kw))
                cp = (ci + 31) // 32 * 32
                return F.pad(w.permute(0, 2, 3, 1), (0, cp - ci)).reshape(co, kh * kw * cp)

            mats = [mat(w.detach().float()).contiguous() for w, _ in convs]
            biases = [None if b is None else b.detach().float().contiguous() for _, b in convs]
        return mats, biases

    def _weights(self, precision):
        """ctypes weight struct of one arithmetic mode, cached per parameter version: "f16x3" = weight planes (None when a
        folded filter leaves the f16x3 weight range), "f32" = the fp32 matrices themselves."""
        if self._src is None:
            self._src = _lib.param_slots(self, buffers=True)
        key = _lib.slots_key(self._src)
        if self._hip is None or self._hip["key"] != key:
            mats, biases = self._matrices()
            self._hip = {"key": key, "mats": mats, "biases": biases, "fit": _weights_fit(mats)}
        ent = self._hip
        if precision not in ent:
            if precision == "f16x3" and not ent["fit"]:
                ent[precision] = None
            else:
                keep = ent["mats"] if precision == "f32" else [_lib.to_planes(m, _lib.PLANES_W_SCALE) for m in ent["mats"]]
                st = _lib.ResnetFpnWeights()
                for i in range(22):
                    st.w[i] = keep[i].data_ptr()
                    st.b[i] = None if ent["biases"][i] is None else ent["biases"][i].data_ptr()
                ent[precision] = (st, keep)
        return None if ent[precision] is None else ent[precision][0]

    def _run(self, x, precision):
        w = self._weights(precision)
        n, _, H, W = x.shape
        dev = x.device
        out_c = torch.empty(n, H // 8 + 2, W // 8 + 2, 256, dtype=torch.float32, device=dev)
        out_f = torch.empty(n, H // 2 + 2, W // 2 + 2, 128, dtype=torch.float32, device=dev)
        nbytes = _lib.lib().pope_resnetfpn_workspace_bytes(n, H, W)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        with _lib.on_device_of(x):
            _lib.check(_lib.lib().pope_resnetfpn_forward_f32(
                C.byref(w), C.c_void_p(x.data_ptr()), n, H, W, _lib.PRECISIONS[precision], C.c_void_p(out_c.data_ptr()),
                C.c_void_p(out_f.data_ptr()), C.c_void_p(ws.data_ptr()), nbytes, C.c_void_p(flag.data_ptr()), _lib.stream_of(dev)),
                "pope_resnetfpn_forward_f32")
        # the reference's NCHW maps as views of the zero-bordered NHWC outputs
        return [out_c[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2), out_f[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2)], flag

    @torch.no_grad()
    def forward(self, x):
        if self.training:
            raise NotImplementedError("pope_amd: inference only (BatchNorm is folded; call .eval())")
        require_cuda(x, "ResNetFPN_8_2")
        if x.dim() != 4 or x.shape[0] == 0 or x.shape[1] != 1 or x.shape[2] % 8 or x.shape[3] % 8 or x.shape[2] < 16 or x.shape[3] < 16:
            raise PopeHipError(f"pope_amd: ResNetFPN_8_2 takes [n >= 1, 1, H, W] gray images with H, W multiples of 8 and >= 16, got {tuple(x.shape)}")
        x = x.float().contiguous()
        policy = self.on_overflow or ON_OVERFLOW
        if self._weights("f16x3") is None:
            if policy == "raise":
                raise PopeRangeError("pope_amd: a folded LoFTR backbone filter is outside the f16x3 weight range (|w| < 255.9)")
            return self._run(x, "f32")[0]
        out, flag = self._run(x, "f16x3")
        bits = _read_flags([flag])
        if bits:
            _overflow("LoFTR backbone", bits, policy)
            out = self._run(x, "f32")[0]
        return out


def build_backbone(config):
    """backbone/__init__.py:4-11; only the (8, 2) resolution is used by cvpr_ds_config.py:9."""
    if config["backbone_type"] != "ResNetFPN" or tuple(config["resolution"]) != (8, 2):
        raise ValueError(f"pope_amd: unsupported LoFTR backbone {config['backbone_type']} {config['resolution']}")
    return ResNetFPN_8_2(config["resnetfpn"])


# ------------------------------------------------------------------------------------ position code
class PositionEncodingSine(nn.Module):
    """utils/position_encoding.py:11-42.  Channel 4k..4k+3 = sin(x w_k), cos(x w_k), sin(y w_k), cos(y w_k)
    with x, y counted from 1.  temp_bug_fix=False (cvpr_ds_config.py:28) selects the released models'
    frequencies: the reference's expression `-log(1e4) / d_model // 2` floors to -1, i.e. w_k = exp(-2k)
    (:28); temp_bug_fix=True gives w_k = 1e4^(-2k / (d_model/2)) (:26).  A host table per grid, added
    to the feature map where it lies."""

    def __init__(self, d_model, max_shape=(256, 256), temp_bug_fix=True):
        super().__init__()
        self.d_model, self.max_shape, self.temp_bug_fix = d_model, tuple(max_shape), temp_bug_fix
        self._cache = {}

    def code(self, h, w, device):
        key = (h, w, str(device))
        if key not in self._cache:
            if h > self.max_shape[0] or w > self.max_shape[1]:
                raise ValueError(f"feature map {h}x{w} exceeds max_shape {self.max_shape}")
            k = torch.arange(0, self.d_model // 2, 2).float()
            rate = -math.log(10000.0) / (self.d_model // 2) if self.temp_bug_fix else float(
                -math.log(10000.0) / self.d_model // 2)
            freq = torch.exp(k * rate)[:, None, None]
            xs = torch.arange(1, w + 1).float().view(1, 1, w).expand(1, h, w)
            ys = torch.arange(1, h + 1).float().view(1, h, 1).expand(1, h, w)
            pe = torch.stack([torch.sin(xs * freq), torch.cos(xs * freq), torch.sin(ys * freq),
                              torch.cos(ys * freq)], 1)                      # [d/4, 4, h, w]
            self._cache[key] = pe.reshape(1, self.d_model, h, w).to(device)
        return self._cache[key]

    def forward(self, x):
        return x + self.code(x.shape[2], x.shape[3], x.device)


# -------------------------------------------------------------------------------- linear-attention transformer
class LoFTREncoderLayer(nn.Module):
    """loftr_module/transformer.py:7-58 with the linear attention of linear_attention.py:20-47.  The layer update is ONE
    call into the HIP library (pope_loftr_encoder_layer_f32: five GEMMs + the O(L) linear-attention kernels + both
    LayerNorms); this module holds the parameters."""

    def __init__(self, d_model, nhead, attention="linear"):
        super().__init__()
        if attention != "linear":
            raise NotImplementedError("pope_amd: only linear attention (cvpr_ds_config.py:27,49)")
        self.dim, self.nhead = d_model // nhead, nhead
        self.q_proj = nn.Linear(d_model, d_model, bias=False)
        self.k_proj = nn.Linear(d_model, d_model, bias=False)
        self.v_proj = nn.Linear(d_model, d_model, bias=False)
        self.merge = nn.Linear(d_model, d_model, bias=False)
        self.mlp = nn.Sequential(nn.Linear(2 * d_model, 2 * d_model, bias=False), nn.ReLU(True),
                                 nn.Linear(2 * d_model, d_model, bias=False))
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self._hip = None
        self._src = None

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._hip, self._src = None, None
        return out

    def _weights(self, precision):
        """ctypes struct of the layer's weights in one arithmetic mode ("f16x3": weight planes, None when a weight leaves
        the f16x3 range; "f32": the fp32 matrices), cached per parameter version."""
        if self._src is None:
            self._src = _lib.param_slots(self)
        key = _lib.slots_key(self._src)
        if self._hip is None or self._hip["key"] != key:
            with torch.no_grad():
                mats = [self.q_proj.weight.detach().float().contiguous(),
                        torch.cat([self.k_proj.weight.detach(), self.v_proj.weight.detach()], 0).float().contiguous(),
                        self.merge.weight.detach().float().contiguous(), self.mlp[0].weight.detach().float().contiguous(),
                        self.mlp[2].weight.detach().float().contiguous()]
                norms = [t.detach().float().contiguous() for t in (self.norm1.weight, self.norm1.bias, self.norm2.weight, self.norm2.bias)]
            self._hip = {"key": key, "mats": mats, "norms": norms, "fit": _weights_fit(mats)}
        ent = self._hip
        if precision not in ent:
            if precision == "f16x3" and not ent["fit"]:
                ent[precision] = None
            else:
                keep = ent["mats"] if precision == "f32" else [_lib.to_planes(m, _lib.PLANES_W_SCALE) for m in ent["mats"]]
                ent[precision] = (_lib.LoftrLayerWeights(*[t.data_ptr() for t in keep + ent["norms"]]), keep)
        return None if ent[precision] is None else ent[precision][0]

    def update_(self, x, source, workspace, precision="f16x3"):
        """In-place layer update of `x` [n, L, C] (fp32, contiguous, CUDA) against `source` (may be `x`); returns the
        call's range-flag word (device)."""
        w = self._weights(precision)
        n, L, Cd = x.shape
        S = source.shape[1]
        flag = torch.zeros(1, dtype=torch.int32, device=x.device)
        with _lib.on_device_of(x):
            _lib.check(_lib.lib().pope_loftr_encoder_layer_f32(
                C.byref(w), C.c_void_p(x.data_ptr()), C.c_void_p(source.data_ptr()), n, L, S, Cd, self.nhead,
                float(self.norm1.eps), _lib.PRECISIONS[precision], C.c_void_p(workspace.data_ptr()), workspace.numel(),
                C.c_void_p(flag.data_ptr()), _lib.stream_of(x.device)), "pope_loftr_encoder_layer_f32")
        return flag

    @torch.no_grad()
    def forward(self, x, source, x_mask=None, source_mask=None):
        """The reference's single-layer call (transformer.py:35-58): returns the updated copy of `x`."""
        if x_mask is not None or source_mask is not None:
            raise NotImplementedError("pope_amd: padding masks are a training-time path (matcher.py:62-64)")
        require_cuda(x, "LoFTREncoderLayer")
        require_cuda(source, "LoFTREncoderLayer")
        policy = ON_OVERFLOW
        n, L, Cd = x.shape
        nbytes = _lib.lib().pope_loftr_layer_workspace_bytes(n, L, source.shape[1], Cd, self.nhead)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        src = source.float().contiguous()
        prec = "f16x3" if self._weights("f16x3") is not None else "f32"
        out = x.float().contiguous().clone()
        bits = int(self.update_(out, out if source is x else src, ws, prec).item())
        if bits:
            _overflow("LoFTR encoder layer", bits, policy)
            out = x.float().contiguous().clone()
            self.update_(out, out if source is x else src, ws, "f32")
        return out


class LocalFeatureTransformer(nn.Module):
    """loftr_module/transformer.py:61-106; 'cross' layers update feat0 first and feed the NEW feat0 into
    the feat1 update (:101-102).  Every layer update is a HIP call; a range-guard event (weight or activation outside the
    f16x3 range) re-runs the whole transformer on the fp32 MFMA, with a warning."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.d_model, self.nhead, self.layer_names = config["d_model"], config["nhead"], config["layer_names"]
        if self.d_model not in (128, 256) or self.nhead != 8:
            raise NotImplementedError("pope_amd: the HIP LoFTR layer is built for d_model 256 / 128 with 8 heads (cvpr_ds_config.py:21-49)")
        for name in self.layer_names:
            if name not in ("self", "cross"):
                raise KeyError(name)
        self.layers = nn.ModuleList([LoFTREncoderLayer(self.d_model, self.nhead, config["attention"])
                                     for _ in self.layer_names])
        for p in self.parameters():  # transformer.py:77-80
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.on_overflow = None    # None: the module-level ON_OVERFLOW

    def _run(self, feat0, feat1, precision):
        n, L, Cd = feat0.shape
        S = feat1.shape[1]
        lib = _lib.lib()
        if L == S:
            # both streams in ONE buffer: a 'self' layer treats them as a batch of 2n (same weights, independent
            # sequences: one call instead of two — the layers are launch-bound at the drivers' batch of three pairs)
            both = torch.cat([feat0.float(), feat1.float()], 0).contiguous()
            f0, f1 = both[:n], both[n:]
        else:
            both = None
            f0, f1 = feat0.float().contiguous().clone(), feat1.float().contiguous().clone()
        nbytes = max(lib.pope_loftr_layer_workspace_bytes(2 * n if both is not None else n, a, b, Cd, self.nhead)
                     for a in (L, S) for b in (L, S))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=f0.device)
        flags = []
        for layer, name in zip(self.layers, self.layer_names):
            if name == "self":
                if both is not None:
                    flags.append(layer.update_(both, both, ws, precision))
                else:
                    flags.append(layer.update_(f0, f0, ws, precision))
                    flags.append(layer.update_(f1, f1, ws, precision))
            else:
                flags.append(layer.update_(f0, f1, ws, precision))
                flags.append(layer.update_(f1, f0, ws, precision))
        return (f0, f1), flags

    @torch.no_grad()
    def forward(self, feat0, feat1, mask0=None, mask1=None):
        assert self.d_model == feat0.size(2), "the feature number of src and transformer must be equal"
        if mask0 is not None or mask1 is not None:
            raise NotImplementedError("pope_amd: padding masks are a training-time path (matcher.py:62-64)")
        require_cuda(feat0, "LocalFeatureTransformer")
        require_cuda(feat1, "LocalFeatureTransformer")
        if feat0.shape[0] == 0:
            return feat0.float().clone(), feat1.float().clone()
        policy = self.on_overflow or ON_OVERFLOW
        if any(l._weights("f16x3") is None for l in self.layers):
            if policy == "raise":
                raise PopeRangeError("pope_amd: a LoFTR transformer weight is outside the f16x3 weight range (|w| < 255.9)")
            return self._run(feat0, feat1, "f32")[0]
        out, flags = self._run(feat0, feat1, "f16x3")
        bits = _read_flags(flags)   # one synchronisation per transformer
        if bits:
            _overflow("LoFTR transformer", bits, policy)
            out = self._run(feat0, feat1, "f32")[0]
        return out


# ------------------------------------------------------------------------------------- fine stage
class FinePreprocess(nn.Module):
    """loftr_module/fine_preprocess.py:7-59: ONE C-ABI call (pope_fine_preprocess_f32, fine.hip) gathers the W x W windows of
    the M matched cells from both 1/2-resolution maps and runs down_proj / merge_feat."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.cat_c_feat = config["fine_concat_coarse_feat"]
        self.W = config["fine_window_size"]
        d_c, d_f = config["coarse"]["d_model"], config["fine"]["d_model"]
        self.d_model_f = d_f
        if not self.cat_c_feat:
            raise NotImplementedError("pope_amd: fine_concat_coarse_feat=True only (cvpr_ds_config.py:12)")
        if self.W * self.W > 64:
            raise NotImplementedError("pope_amd: fine windows up to 8 x 8 (cvpr_ds_config.py:11 uses 5)")
        self.down_proj = nn.Linear(d_c, d_f, bias=True)
        self.merge_feat = nn.Linear(2 * d_f, d_f, bias=True)
        for p in self.parameters():  # fine_preprocess.py:24-27
            if p.dim() > 1:
                nn.init.kaiming_normal_(p, mode="fan_out", nonlinearity="relu")
        self._hip = None
        self._src = None
        self.on_overflow = None

    def _apply(self, fn, *a, **k):
        self._hip, self._src = None, None
        return super()._apply(fn, *a, **k)

    def _weights(self, precision):
        if self._src is None:
            self._src = _lib.param_slots(self)
        key = _lib.slots_key(self._src)
        if self._hip is None or self._hip["key"] != key:
            mats = [self.down_proj.weight.detach().float().contiguous(), self.merge_feat.weight.detach().float().contiguous()]
            self._hip = {"key": key, "mats": mats, "fit": _weights_fit(mats),
                         "biases": [self.down_proj.bias.detach().float().contiguous(), self.merge_feat.bias.detach().float().contiguous()]}
        ent = self._hip
        if precision not in ent:
            if precision == "f16x3" and not ent["fit"]:
                ent[precision] = None
            else:
                ent[precision] = ent["mats"] if precision == "f32" else [_lib.to_planes(m, _lib.PLANES_W_SCALE) for m in ent["mats"]]
        return ent[precision]

    def _run(self, f0, f1, fc0, fc1, data, stride, precision):
        dwp, mwp = self._weights(precision)
        db, mb = self._hip["biases"]
        b, i, j = (t.contiguous() for t in (data["b_ids"], data["i_ids"], data["j_ids"]))
        M, W, Cf = int(b.shape[0]), self.W, self.d_model_f
        dev = f0.device
        out = torch.empty(2 * M, W * W, Cf, dtype=torch.float32, device=dev)
        lib = _lib.lib()
        nbytes = lib.pope_fine_preprocess_workspace_bytes(M, W, fc0.shape[2], Cf)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        s0, s1 = (C.c_longlong * 4)(*f0.stride()), (C.c_longlong * 4)(*f1.stride())
        with _lib.on_device_of(f0):
            _lib.check(lib.pope_fine_preprocess_f32(
                C.c_void_p(f0.data_ptr()), s0, f0.shape[2], f0.shape[3], int(data["hw0_c"][1]),
                C.c_void_p(f1.data_ptr()), s1, f1.shape[2], f1.shape[3], int(data["hw1_c"][1]),
                C.c_void_p(fc0.data_ptr()), C.c_void_p(fc1.data_ptr()), fc0.shape[1], fc1.shape[1], fc0.shape[2], Cf,
                C.c_void_p(b.data_ptr()), C.c_void_p(i.data_ptr()), C.c_void_p(j.data_ptr()), M, W, int(stride),
                C.c_void_p(dwp.data_ptr()), C.c_void_p(db.data_ptr()), C.c_void_p(mwp.data_ptr()), C.c_void_p(mb.data_ptr()),
                _lib.PRECISIONS[precision], C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()), nbytes, C.c_void_p(flag.data_ptr()),
                _lib.stream_of(dev)), "pope_fine_preprocess_f32")
        return (out[:M], out[M:]), flag

    @torch.no_grad()
    def forward(self, feat_f0, feat_f1, feat_c0, feat_c1, data):
        W = self.W
        stride = data["hw0_f"][0] // data["hw0_c"][0]
        data.update({"W": W})
        if data["b_ids"].shape[0] == 0:   # fine_preprocess.py:33-36
            empty = torch.empty(0, W * W, self.d_model_f, device=feat_f0.device)
            return empty, empty.clone()
        require_cuda(feat_f0, "FinePreprocess")
        # NB the reference unfolds image 1 with image 0's stride and both with their own width (:44-47)
        args = (feat_f0.float(), feat_f1.float(), feat_c0.float().contiguous(), feat_c1.float().contiguous(), data, stride)
        policy = self.on_overflow or ON_OVERFLOW
        if self._weights("f16x3") is None:
            if policy == "raise":
                raise PopeRangeError("pope_amd: a LoFTR fine-preprocess weight is outside the f16x3 weight range (|w| < 255.9)")
            return self._run(*args, "f32")[0]
        out, flag = self._run(*args, "f16x3")
        bits = int(flag.item())
        if bits:
            _overflow("LoFTR fine preprocess", bits, policy)
            out = self._run(*args, "f32")[0]
        return out


class FineMatching(nn.Module):
    """utils/fine_matching.py:9-74: correlate the centre of window 0 with window 1, softmax(1/sqrt(C)),
    expectation over the normalised [-1,1]^2 grid (x,y) (kornia dsnt.spatial_expectation2d / create_meshgrid
    in the reference; both are closed-form, see SURVEY.md §8c 'unpinned'): one kernel, fp32 throughout
    (pope_fine_match_f32)."""

    @torch.no_grad()
    def forward(self, feat_f0, feat_f1, data):
        M, WW, Cd = feat_f0.shape
        W = int(math.sqrt(WW))
        scale = data["hw0_i"][0] / data["hw0_f"][0]
        if M == 0:   # fine_matching.py:33-41
            data.update({"expec_f": torch.empty(0, 3, device=feat_f0.device),
                         "mkpts0_f": data["mkpts0_c"], "mkpts1_f": data["mkpts1_c"]})
            return
        require_cuda(feat_f0, "FineMatching")
        if "scale0" in data or WW > 64 or len(data["mconf"]) != M:
            raise NotImplementedError("pope_amd: FineMatching covers the drivers' use (no per-image rescaling `scale0` / `scale1`, "
                                      "windows up to 8 x 8, no training-time padding of the match list)")
        w0, w1 = feat_f0.float().contiguous(), feat_f1.float().contiguous()
        mk1c = data["mkpts1_c"].float().contiguous()
        expec = torch.empty(M, 3, dtype=torch.float32, device=w0.device)
        mk1f = torch.empty(M, 2, dtype=torch.float32, device=w0.device)
        with _lib.on_device_of(w0):
            _lib.check(_lib.lib().pope_fine_match_f32(
                C.c_void_p(w0.data_ptr()), C.c_void_p(w1.data_ptr()), M, W, Cd, C.c_void_p(mk1c.data_ptr()),
                float(scale), C.c_void_p(expec.data_ptr()), C.c_void_p(mk1f.data_ptr()), _lib.stream_of(w0.device)),
                "pope_fine_match_f32")
        # get_fine_match (:61-74): image 0 keeps its coarse cell centre, image 1 moves inside the window
        data.update({"expec_f": expec, "mkpts0_f": data["mkpts0_c"], "mkpts1_f": mk1f})
