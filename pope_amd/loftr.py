"""LoFTR stages either side of the HIP coarse matcher: local-feature CNN, sinusoidal
position code, linear-attention transformer, fine window preprocessing and sub-pixel refinement.  On CUDA tensors every
stage but the position code is a call into libpope_hip.so (conv.hip, loftr.hip, fine.hip); the torch forms below are
the fp32 re-run of the f16x3 range guard and the restatement the GPU tests compare against (`use_hip = False`).

Every module keeps the reference's parameter names and shapes so that `weights/matcher.pth` loads with
strict=True (211 keys: backbone.* 107, loftr_coarse.* 80, fine_preprocess.* 4, loftr_fine.* 20), but the
computation is organised for inference on one big GPU:
  * eval-mode BatchNorm is folded into the preceding convolution once per weight load (one pass over the
    activations instead of two; resnet_fpn.py:27-40,60-63,72-84 keep conv and BN separate);
  * fine windows are gathered only at the M matched cells instead of unfolding every window of the
    1/2-resolution map and indexing afterwards (fine_preprocess.py:44-51 materialises [n, L, 25, 128]);
  * the position code is generated for the requested grid, not sliced from a 256x256 buffer.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------ CNN

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


def _fold_bn(conv_w, bn):
    """Filter and bias of conv -> BatchNorm(eval) as one convolution."""
    g = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
    return conv_w * g.view(-1, 1, 1, 1), bn.bias - bn.running_mean * g


class BasicBlock(nn.Module):
    """resnet_fpn.py:15-40: relu(x' + bn2(conv2(relu(bn1(conv1(x)))))), x' = 1x1 stride-s conv + BN if s != 1."""

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

    def run(self, x, f):
        y = F.relu_(F.conv2d(x, f[0][0], f[0][1], self.stride, 1))
        y = F.conv2d(y, f[1][0], f[1][1], 1, 1)
        if self.downsample is not None:
            x = F.conv2d(x, f[2][0], f[2][1], self.stride)
        return F.relu_(y.add_(x))


class ResNetFPN_8_2(nn.Module):
    """resnet_fpn.py:43-118.  [B,1,H,W] -> coarse [B,256,H/8,W/8], fine [B,128,H/2,W/2]."""

    def __init__(self, config):
        super().__init__()
        d0 = config["initial_dim"]
        d1, d2, d3 = config["block_dims"]
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
        self._folded = None
        self._hip = None
        self.use_hip = True   # dev / test switch: False = the torch / MIOpen form on any device
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate())

    def _fold(self):
        with torch.no_grad():
            return {
                "stem": _fold_bn(self.conv1.weight, self.bn1),
                "blocks": [[b.folded() for b in layer] for layer in (self.layer1, self.layer2, self.layer3)],
                "out2": _fold_bn(self.layer2_outconv2[0].weight, self.layer2_outconv2[1]),
                "out1": _fold_bn(self.layer1_outconv2[0].weight, self.layer1_outconv2[1]),
            }

    def invalidate(self):
        self._folded = None
        self._hip = None

    def _apply(self, fn, *a, **k):  # .to()/.cuda()/.float() move the parameters: refold lazily
        self._folded = None
        self._hip = None
        return super()._apply(fn, *a, **k)

    def _hip_weights(self):
        """The 22 convolutions as f16x3 weight planes in the order of pope_hip.h (BatchNorm folded, [Cout, taps x Cin
        rounded up to 32] tap-major), biases as fp32.  None when a folded filter leaves the f16x3 weight range."""
        from . import _lib
        if self._hip is not None:
            return self._hip[0]
        f = self._folded

        def mat(w):   # [Cout, Cin, kh, kw] -> [Cout, kh * kw * Cp]
            co, ci, kh, kw = w.shape
            if ci == 1:   # the 7x7 stem: 49 taps of one channel, zero-filled to 64 columns
                m = w.reshape(co, kh * kw)
                return F.pad(m, (0, 64 - kh * kw))
            cp = (ci + 31) // 32 * 32
            return F.pad(w.permute(0, 2, 3, 1), (0, cp - ci)).reshape(co, kh * kw * cp)

        convs = [f["stem"]]
        for lf in f["blocks"]:
            for bf in lf:
                convs += bf                      # conv1, conv2[, downsample]
        # reorder layer2 / layer3 entries to conv1, conv2, downsample, conv1, conv2 (already so: b0 has 3, b1 has 2)
        convs += [(self.layer3_outconv.weight, None), (self.layer2_outconv.weight, None), f["out2"],
                  (self.layer2_outconv2[3].weight, None), (self.layer1_outconv.weight, None), f["out1"],
                  (self.layer1_outconv2[3].weight, None)]
        assert len(convs) == 22
        mats = [mat(w.detach().float()).contiguous() for w, _ in convs]
        amax = max(float(m.abs().max()) for m in mats)
        if not amax * _lib.PLANES_W_SCALE < _lib.F16_MAX:
            self._hip = (None, None)
            return None
        keep = [_lib.to_planes(m, _lib.PLANES_W_SCALE) for m in mats]
        biases = [None if b is None else b.detach().float().contiguous() for _, b in convs]
        st = _lib.ResnetFpnWeights()
        for i in range(22):
            st.w[i] = keep[i].data_ptr()
            st.b[i] = None if biases[i] is None else biases[i].data_ptr()
        self._hip = (st, keep + [b for b in biases if b is not None])
        return st

    def _forward_hip(self, x):
        """One C-ABI call (pope_resnetfpn_forward_f32, conv.hip): every convolution on the f16x3 planes GEMM.  Returns the
        reference's NCHW maps as views of the zero-bordered NHWC outputs, or None when the f16x3 range guard fired."""
        import ctypes as C
        from . import _lib
        w = self._hip_weights()
        if w is None:
            return None
        n, _, H, W = x.shape
        x = x.float().contiguous()
        dev = x.device
        out_c = torch.empty(n, H // 8 + 2, W // 8 + 2, 256, dtype=torch.float32, device=dev)
        out_f = torch.empty(n, H // 2 + 2, W // 2 + 2, 128, dtype=torch.float32, device=dev)
        nbytes = _lib.lib().pope_resnetfpn_workspace_bytes(n, H, W)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        with _lib.on_device_of(x):
            _lib.check(_lib.lib().pope_resnetfpn_forward_f32(
                C.byref(w), C.c_void_p(x.data_ptr()), n, H, W, C.c_void_p(out_c.data_ptr()), C.c_void_p(out_f.data_ptr()),
                C.c_void_p(ws.data_ptr()), nbytes, C.c_void_p(flag.data_ptr()), _lib.stream_of(dev)), "pope_resnetfpn_forward_f32")
        bits = _read_flags([flag])
        if bits:
            import warnings
            warnings.warn(f"pope_amd: f16x3 range contract breached in the LoFTR backbone ({_lib.describe_range_bits(bits)}); "
                          "re-running it in torch fp32")
            return None
        return [out_c[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2), out_f[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2)]

    @torch.no_grad()
    def forward(self, x):
        if self.training:
            raise NotImplementedError("pope_amd: inference only (BatchNorm is folded; call .eval())")
        if self._folded is None:
            self._folded = self._fold()
        f = self._folded
        if self.use_hip and x.is_cuda and x.shape[0] > 0 and x.shape[1] == 1 and x.shape[2] % 8 == 0 and x.shape[3] % 8 == 0 \
                and x.shape[2] >= 16 and x.shape[3] >= 16:
            out = self._forward_hip(x)
            if out is not None:
                return out
        x0 = F.relu_(F.conv2d(x, f["stem"][0], f["stem"][1], 2, 3))
        feats = []
        h = x0
        for layer, lf in zip((self.layer1, self.layer2, self.layer3), f["blocks"]):
            for blk, bf in zip(layer, lf):
                h = blk.run(h, bf)
            feats.append(h)
        x1, x2, x3 = feats
        up = lambda t: F.interpolate(t, scale_factor=2.0, mode="bilinear", align_corners=True)  # noqa: E731
        x3_out = self.layer3_outconv(x3)
        t = self.layer2_outconv(x2).add_(up(x3_out))
        t = F.leaky_relu_(F.conv2d(t, f["out2"][0], f["out2"][1], 1, 1))
        x2_out = self.layer2_outconv2[3](t)
        t = self.layer1_outconv(x1).add_(up(x2_out))
        t = F.leaky_relu_(F.conv2d(t, f["out1"][0], f["out1"][1], 1, 1))
        x1_out = self.layer1_outconv2[3](t)
        return [x3_out, x1_out]


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
    (:28); temp_bug_fix=True gives w_k = 1e4^(-2k / (d_model/2)) (:26)."""

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


# -------------------------------------------------------------------------------- linear attention
def linear_attention(q, k, v, eps=1e-6):
    """loftr_module/linear_attention.py:20-47 (no masks): phi = elu + 1;
    out_l = phi(q_l) (sum_s phi(k_s) v_s^T) / (phi(q_l) . sum_s phi(k_s) + eps).
    q [n,L,h,d], k,v [n,S,h,d].  The reference divides v by S and multiplies back afterwards; kept, since
    it changes the rounding."""
    S = v.shape[1]
    Q, K = F.elu(q) + 1, F.elu(k) + 1
    kv = torch.einsum("nshd,nshv->nhdv", K, v / S)
    z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(1)) + eps)
    return torch.einsum("nlhd,nhdv,nlh->nlhv", Q, kv, z) * S


class LoFTREncoderLayer(nn.Module):
    """loftr_module/transformer.py:7-58.  On CUDA tensors the layer update is ONE call into the HIP library
    (pope_loftr_encoder_layer_f32: five f16x3 planes GEMMs + the O(L) linear-attention kernels + both LayerNorms);
    the torch form below is the CPU restatement used by the `-m "not gpu"` tests."""

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

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._hip = None
        return out

    def load_state_dict(self, *a, **k):
        self._hip = None
        return super().load_state_dict(*a, **k)

    def _hip_weights(self):
        """ctypes struct of the layer's weights: the five bias-free Linears as f16x3 weight planes (range-checked),
        LayerNorm parameters as they are.  None when a weight leaves the f16x3 range (the torch form then runs)."""
        from . import _lib
        key = self.q_proj.weight.data_ptr()
        if self._hip is not None and self._hip[0] == key:
            return self._hip[1]
        lin = [self.q_proj.weight, self.k_proj.weight, self.v_proj.weight, self.merge.weight, self.mlp[0].weight, self.mlp[2].weight]
        amax = float(torch.stack([t.detach().abs().max() for t in lin]).max())
        if not amax * _lib.PLANES_W_SCALE < _lib.F16_MAX:
            self._hip = (key, None, None)
            return None
        keep = [_lib.to_planes(self.q_proj.weight, _lib.PLANES_W_SCALE),
                _lib.to_planes(torch.cat([self.k_proj.weight.detach(), self.v_proj.weight.detach()], 0), _lib.PLANES_W_SCALE),
                _lib.to_planes(self.merge.weight, _lib.PLANES_W_SCALE),
                _lib.to_planes(self.mlp[0].weight, _lib.PLANES_W_SCALE), _lib.to_planes(self.mlp[2].weight, _lib.PLANES_W_SCALE)]
        norms = [t.detach().float().contiguous() for t in (self.norm1.weight, self.norm1.bias, self.norm2.weight, self.norm2.bias)]
        w = _lib.LoftrLayerWeights(*[t.data_ptr() for t in keep + norms])
        self._hip = (key, w, keep + norms)
        return w

    def update_(self, x, source, workspace):
        """In-place HIP layer update of `x` [n, L, C] (fp32, contiguous, CUDA) against `source` (may be `x`)."""
        import ctypes as C
        from . import _lib
        w = self._hip_weights()
        n, L, Cd = x.shape
        S = source.shape[1]
        flag = torch.zeros(1, dtype=torch.int32, device=x.device)
        with _lib.on_device_of(x):
            _lib.check(_lib.lib().pope_loftr_encoder_layer_f32(
                C.byref(w), C.c_void_p(x.data_ptr()), C.c_void_p(source.data_ptr()), n, L, S, Cd, self.nhead,
                float(self.norm1.eps), C.c_void_p(workspace.data_ptr()), workspace.numel(), C.c_void_p(flag.data_ptr()),
                _lib.stream_of(x.device)), "pope_loftr_encoder_layer_f32")
        return flag

    def forward(self, x, source, x_mask=None, source_mask=None):
        if x_mask is not None or source_mask is not None:
            raise NotImplementedError("pope_amd: padding masks are a training-time path (matcher.py:62-64)")
        n = x.shape[0]
        q = self.q_proj(x).view(n, -1, self.nhead, self.dim)
        k = self.k_proj(source).view(n, -1, self.nhead, self.dim)
        v = self.v_proj(source).view(n, -1, self.nhead, self.dim)
        msg = self.norm1(self.merge(linear_attention(q, k, v).reshape(n, -1, self.nhead * self.dim)))
        return x + self.norm2(self.mlp(torch.cat([x, msg], 2)))


class LocalFeatureTransformer(nn.Module):
    """loftr_module/transformer.py:61-106; 'cross' layers update feat0 first and feed the NEW feat0 into
    the feat1 update (:101-102).  CUDA inputs run on the HIP encoder layer (f16x3 planes GEMMs, guarded: if a weight or
    an activation leaves the f16x3 range the whole transformer is re-run in torch fp32, with a warning)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.d_model, self.nhead, self.layer_names = config["d_model"], config["nhead"], config["layer_names"]
        self.layers = nn.ModuleList([LoFTREncoderLayer(self.d_model, self.nhead, config["attention"])
                                     for _ in self.layer_names])
        for p in self.parameters():  # transformer.py:77-80
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.use_hip = True   # dev / test switch: False = the torch form on any device

    def _forward_torch(self, feat0, feat1):
        for layer, name in zip(self.layers, self.layer_names):
            if name == "self":
                feat0 = layer(feat0, feat0)
                feat1 = layer(feat1, feat1)
            elif name == "cross":
                feat0 = layer(feat0, feat1)
                feat1 = layer(feat1, feat0)
            else:
                raise KeyError(name)
        return feat0, feat1

    def _forward_hip(self, feat0, feat1):
        from . import _lib
        if self.d_model not in (128, 256) or self.nhead != 8 or any(l._hip_weights() is None for l in self.layers):
            return None
        for name in self.layer_names:
            if name not in ("self", "cross"):
                raise KeyError(name)
        n, L, C = feat0.shape
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
        nbytes = max(lib.pope_loftr_layer_workspace_bytes(2 * n if both is not None else n, a, b, C, self.nhead)
                     for a in (L, S) for b in (L, S))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=f0.device)
        flags = []
        for layer, name in zip(self.layers, self.layer_names):
            if name == "self":
                if both is not None:
                    flags.append(layer.update_(both, both, ws))
                else:
                    flags.append(layer.update_(f0, f0, ws))
                    flags.append(layer.update_(f1, f1, ws))
            else:
                flags.append(layer.update_(f0, f1, ws))
                flags.append(layer.update_(f1, f0, ws))
        bits = _read_flags(flags)   # one synchronisation per transformer
        if bits:
            import warnings
            warnings.warn(f"pope_amd: f16x3 range contract breached in the LoFTR transformer ({_lib.describe_range_bits(bits)}); "
                          "re-running it in torch fp32")
            return None
        return f0, f1

    def forward(self, feat0, feat1, mask0=None, mask1=None):
        assert self.d_model == feat0.size(2), "the feature number of src and transformer must be equal"
        if mask0 is not None or mask1 is not None:
            raise NotImplementedError("pope_amd: padding masks are a training-time path (matcher.py:62-64)")
        if self.use_hip and feat0.is_cuda and feat0.shape[0] > 0:
            out = self._forward_hip(feat0, feat1)
            if out is not None:
                return out
        return self._forward_torch(feat0, feat1)


# ------------------------------------------------------------------------------------- fine stage
def gather_windows(feat_f, b_ids, cell_ids, w_c, W, stride):
    """Rows `feat_unfold[b, cell]` of the reference's unfold (fine_preprocess.py:44-51) without building
    the unfold: window (W x W, zero padded by W//2) of feat_f [n,C,Hf,Wf] centred on fine pixel
    (cy*stride, cx*stride) for coarse cell id = cy*w_c + cx.  Returns [M, W*W, C], window index kh*W+kw."""
    pad = W // 2
    fp = F.pad(feat_f, (pad, pad, pad, pad))
    d = torch.arange(W, device=feat_f.device)
    ys = ((cell_ids // w_c) * stride)[:, None, None] + d[None, :, None]   # [M, W, 1] (already offset by pad)
    xs = ((cell_ids % w_c) * stride)[:, None, None] + d[None, None, :]    # [M, 1, W]
    win = fp[b_ids[:, None, None], :, ys, xs]                             # [M, W, W, C]
    return win.reshape(win.shape[0], W * W, -1)


class FinePreprocess(nn.Module):
    """loftr_module/fine_preprocess.py:7-59."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.cat_c_feat = config["fine_concat_coarse_feat"]
        self.W = config["fine_window_size"]
        d_c, d_f = config["coarse"]["d_model"], config["fine"]["d_model"]
        self.d_model_f = d_f
        if self.cat_c_feat:
            self.down_proj = nn.Linear(d_c, d_f, bias=True)
            self.merge_feat = nn.Linear(2 * d_f, d_f, bias=True)
        for p in self.parameters():  # fine_preprocess.py:24-27
            if p.dim() > 1:
                nn.init.kaiming_normal_(p, mode="fan_out", nonlinearity="relu")
        self._hip = None
        self.use_hip = True   # dev / test switch: False = the torch form on any device

    def _apply(self, fn, *a, **k):
        self._hip = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._hip = None
        return super().load_state_dict(*a, **k)

    def _forward_hip(self, feat_f0, feat_f1, feat_c0, feat_c1, data, stride):
        """One C-ABI call (pope_fine_preprocess_f32, fine.hip): gathers + the two Linears on the planes GEMM.  None when a
        weight or an activation leaves the f16x3 range (the torch form then runs)."""
        import ctypes as C
        from . import _lib
        key = self.down_proj.weight.data_ptr()
        if self._hip is None or self._hip[0] != key:
            ws_ = [self.down_proj.weight.detach().float(), self.merge_feat.weight.detach().float()]
            if not max(float(t.abs().max()) for t in ws_) * _lib.PLANES_W_SCALE < _lib.F16_MAX:
                self._hip = (key, None)
            else:
                self._hip = (key, [_lib.to_planes(t, _lib.PLANES_W_SCALE) for t in ws_]
                             + [self.down_proj.bias.detach().float().contiguous(), self.merge_feat.bias.detach().float().contiguous()])
        if self._hip[1] is None:
            return None
        dwp, mwp, db, mb = self._hip[1]
        b, i, j = (t.contiguous() for t in (data["b_ids"], data["i_ids"], data["j_ids"]))
        M, W, Cf = int(b.shape[0]), self.W, self.d_model_f
        fc0, fc1 = feat_c0.float().contiguous(), feat_c1.float().contiguous()
        f0, f1 = feat_f0.float(), feat_f1.float()
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
                C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()), nbytes, C.c_void_p(flag.data_ptr()), _lib.stream_of(dev)),
                "pope_fine_preprocess_f32")
        bits = int(flag.item())
        if bits:
            import warnings
            warnings.warn(f"pope_amd: f16x3 range contract breached in the LoFTR fine preprocess ({_lib.describe_range_bits(bits)}); "
                          "re-running it in torch fp32")
            return None
        return out[:M], out[M:]

    def forward(self, feat_f0, feat_f1, feat_c0, feat_c1, data):
        W = self.W
        stride = data["hw0_f"][0] // data["hw0_c"][0]
        data.update({"W": W})
        b, i, j = data["b_ids"], data["i_ids"], data["j_ids"]
        if b.shape[0] == 0:
            empty = torch.empty(0, W * W, self.d_model_f, device=feat_f0.device)
            return empty, empty.clone()
        if self.use_hip and self.cat_c_feat and feat_f0.is_cuda and W * W <= 64:
            out = self._forward_hip(feat_f0, feat_f1, feat_c0, feat_c1, data, stride)
            if out is not None:
                return out
        # NB the reference unfolds image 1 with image 0's stride and both with their own width (:44-47)
        win0 = gather_windows(feat_f0, b, i, data["hw0_c"][1], W, stride)
        win1 = gather_windows(feat_f1, b, j, data["hw1_c"][1], W, stride)
        if self.cat_c_feat:
            c_win = self.down_proj(torch.cat([feat_c0[b, i], feat_c1[b, j]], 0))               # [2M, d_f]
            both = torch.cat([torch.cat([win0, win1], 0), c_win[:, None, :].expand(-1, W * W, -1)], -1)
            win0, win1 = torch.chunk(self.merge_feat(both), 2, dim=0)
        return win0, win1


class FineMatching(nn.Module):
    """utils/fine_matching.py:9-74: correlate the centre of window 0 with window 1, softmax(1/sqrt(C)),
    expectation over the normalised [-1,1]^2 grid (x,y) (kornia dsnt.spatial_expectation2d / create_meshgrid
    in the reference; both are closed-form and restated here, see SURVEY.md §8c 'unpinned')."""

    def forward(self, feat_f0, feat_f1, data):
        M, WW, C = feat_f0.shape
        W = int(math.sqrt(WW))
        scale = data["hw0_i"][0] / data["hw0_f"][0]
        if M == 0:
            data.update({"expec_f": torch.empty(0, 3, device=feat_f0.device),
                         "mkpts0_f": data["mkpts0_c"], "mkpts1_f": data["mkpts1_c"]})
            return
        if getattr(self, "use_hip", True) and feat_f0.is_cuda and "scale0" not in data and WW <= 64 and len(data["mconf"]) == M:
            import ctypes
            from . import _lib
            w0, w1 = feat_f0.float().contiguous(), feat_f1.float().contiguous()
            mk1c = data["mkpts1_c"].float().contiguous()
            expec = torch.empty(M, 3, dtype=torch.float32, device=w0.device)
            mk1f = torch.empty(M, 2, dtype=torch.float32, device=w0.device)
            with _lib.on_device_of(w0):
                _lib.check(_lib.lib().pope_fine_match_f32(
                    ctypes.c_void_p(w0.data_ptr()), ctypes.c_void_p(w1.data_ptr()), M, W, C, ctypes.c_void_p(mk1c.data_ptr()),
                    float(scale), ctypes.c_void_p(expec.data_ptr()), ctypes.c_void_p(mk1f.data_ptr()), _lib.stream_of(w0.device)),
                    "pope_fine_match_f32")
            data.update({"expec_f": expec, "mkpts0_f": data["mkpts0_c"], "mkpts1_f": mk1f})
            return
        sim = torch.einsum("mc,mrc->mr", feat_f0[:, WW // 2, :], feat_f1)
        heat = torch.softmax(sim * (1.0 / C ** 0.5), dim=1)                          # [M, WW]
        lin = torch.linspace(-1, 1, W, device=heat.device)
        grid = torch.stack(torch.meshgrid(lin, lin, indexing="ij")[::-1], -1).reshape(1, WW, 2)  # (x, y)
        coords = (heat[:, :, None] * grid).sum(1)                                     # [M, 2]
        var = (grid ** 2 * heat[:, :, None]).sum(1) - coords ** 2
        std = torch.sqrt(torch.clamp(var, min=1e-10)).sum(-1)
        data.update({"expec_f": torch.cat([coords, std[:, None]], -1)})
        # get_fine_match (:61-74): image 0 keeps its coarse cell centre, image 1 moves inside the window
        scale1 = scale * data["scale1"][data["b_ids"]] if "scale0" in data else scale
        data.update({"mkpts0_f": data["mkpts0_c"],
                     "mkpts1_f": data["mkpts1_c"] + (coords * (W // 2) * scale1)[:len(data["mconf"])]})
