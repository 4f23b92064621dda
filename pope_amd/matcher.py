"""Dense cross-image matcher on the HIP kernels: mirror of the reference ``CoarseMatching``
(src/matcher/utils/coarse_matching.py:60-261) for the eval / dual-softmax path, plus a functional
``dense_match`` used directly on DINOv2 patch tokens (BASELINE config 3).
"""
import ctypes as C
import warnings

import torch
import torch.nn as nn

from . import _lib
from ._lib import PopeRangeError, check, on_device_of, ptr, require_cuda, stream_of

# src/matcher/utils/cvpr_ds_config.py:10-50, lower-cased as by lower_config(); plain data.
default_cfg = {
    "backbone_type": "ResNetFPN",
    "resolution": (8, 2),
    "fine_window_size": 5,
    "fine_concat_coarse_feat": True,
    "resnetfpn": {"initial_dim": 128, "block_dims": [128, 196, 256]},
    "coarse": {"d_model": 256, "d_ffn": 256, "nhead": 8, "layer_names": ["self", "cross"] * 4,
               "attention": "linear", "temp_bug_fix": False},
    "match_coarse": {"thr": 0.2, "border_rm": 2, "match_type": "dual_softmax", "dsmax_temperature": 0.1,
                     "skh_iters": 3, "skh_init_bin_score": 1.0, "skh_prefilter": True,
                     "train_coarse_percent": 0.4, "train_pad_num_gt_min": 200},
    "fine": {"d_model": 128, "d_ffn": 128, "nhead": 8, "layer_names": ["self", "cross"], "attention": "linear"},
}


def _rows_contiguous(f):
    """Accept [n, L, C] views whose rows are dense (e.g. x_norm[:, 1:]) without copying."""
    if f.stride(2) == 1 and f.stride(1) == f.shape[2] and f.stride(0) >= f.shape[1] * f.shape[2] \
            and f.stride(0) % 4 == 0 and f.data_ptr() % 16 == 0:
        return f
    return f.contiguous()


DEFAULT_PRECISION = "f16x3"   # arithmetic of the L x S x C contraction ("f32": fp32-in MFMA); see DESIGN.md §2


@torch.no_grad()
def dense_match(feat0, feat1, hw0_c, hw1_c, hw0_i, thr=0.2, border_rm=2, temperature=0.1, precision=None,
                on_overflow="rerun_f32", want_conf=True):
    """All-pairs similarity -> dual softmax -> threshold/border/mutual-NN -> ordered matches.

    feat0 [n,L,C], feat1 [n,S,C] fp32 on the GPU.  Returns the dict CoarseMatching publishes
    (coarse_matching.py:145-148,239-259): conf_matrix, b_ids, i_ids, j_ids, gt_mask, m_bids,
    mkpts0_c, mkpts1_c, mconf — plus `counts` (matches per pair, int32[n]).  `want_conf=False` (batch pipelines that
    only consume the match lists) does not publish conf_matrix.
    One host synchronisation (to size the outputs), like torch.where in the reference.
    precision "f16x3" (default) keeps feat / sqrt(C) * 256 as f16 pairs: the kernel that converts them guards the
    f16 range with a device flag that travels with the match counts; on a breach the call is repeated with the
    fp32-MFMA contraction (`on_overflow="rerun_f32"`) or raises PopeRangeError ("raise")."""
    require_cuda(feat0, "dense_match")
    require_cuda(feat1, "dense_match")
    if feat0.dtype != torch.float32 or feat1.dtype != torch.float32:
        raise TypeError("dense_match expects float32 features")
    feat0, feat1 = _rows_contiguous(feat0), _rows_contiguous(feat1)
    n, L, Cc = feat0.shape
    S = feat1.shape[1]
    h0, w0 = int(hw0_c[0]), int(hw0_c[1])
    h1, w1 = int(hw1_c[0]), int(hw1_c[1])
    assert feat1.shape[0] == n and feat1.shape[2] == Cc and L == h0 * w0 and S == h1 * w1
    dev = feat0.device
    if n == 0:   # an empty batch: what torch.where gives the reference on a [0, L, S] mask
        el, ef = torch.empty(0, dtype=torch.int64, device=dev), torch.empty(0, dtype=torch.float32, device=dev)
        out = {"b_ids": el, "i_ids": el.clone(), "j_ids": el.clone(), "gt_mask": torch.empty(0, dtype=torch.bool, device=dev),
               "m_bids": el.clone(), "mkpts0_c": ef.reshape(0, 2), "mkpts1_c": ef.reshape(0, 2).clone(), "mconf": ef.clone(),
               "counts": torch.empty(0, dtype=torch.int32),
               "conf_matrix": torch.empty(0, L, S, dtype=torch.float32, device=dev) if want_conf else None}
        return out
    lib = _lib.lib()
    precision = precision or DEFAULT_PRECISION
    if Cc % 32 or Cc < 64:
        precision = "f32"   # the planes layout needs whole 32-column chunks
    prec = _lib.PRECISIONS[precision]
    ws_bytes = lib.pope_dense_match_workspace_bytes_prec(n, L, S, Cc, prec, 1 if want_conf else 0)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    conf = torch.empty(n, L, S, dtype=torch.float32, device=dev) if want_conf else None
    cap = n * L
    b_ids = torch.empty(cap, dtype=torch.int64, device=dev)
    i_ids = torch.empty(cap, dtype=torch.int64, device=dev)
    j_ids = torch.empty(cap, dtype=torch.int64, device=dev)
    mconf = torch.empty(cap, dtype=torch.float32, device=dev)
    mk0 = torch.empty(cap, 2, dtype=torch.float32, device=dev)
    mk1 = torch.empty(cap, 2, dtype=torch.float32, device=dev)
    counts = torch.zeros(n + 2, dtype=torch.int32, device=dev)   # per-pair counts | total | f16x3 range-guard word
    flag_ptr = C.c_void_p(counts.data_ptr() + 4 * (n + 1)) if precision == "f16x3" else None
    scale = hw0_i[0] / hw0_c[0]  # coarse_matching.py:242 (heights only)
    # (the stride of a size-1 dimension is arbitrary in torch: a batch of one has no batch stride to speak of)
    bs0 = feat0.stride(0) if n > 1 else L * Cc
    bs1 = feat1.stride(0) if n > 1 else S * Cc
    with on_device_of(feat0):
        check(lib.pope_dense_match_prec_f32(C.c_void_p(feat0.data_ptr()), bs0, C.c_void_p(feat1.data_ptr()),
                                            bs1, n, L, S, Cc, h0, w0, h1, w1, float(thr), int(border_rm),
                                            float(temperature), float(scale), ptr(conf), ptr(b_ids), ptr(i_ids), ptr(j_ids),
                                            ptr(mconf), ptr(mk0), ptr(mk1), ptr(counts), C.c_void_p(ws.data_ptr()), ws_bytes,
                                            prec, flag_ptr, stream_of(dev)), "pope_dense_match_prec_f32")
    counts_h = counts.cpu()  # sync point
    if int(counts_h[n + 1]):   # features outside the f16x3 range: nothing of this call is valid
        msg = (f"pope_amd: f16x3 range contract breached in dense_match ({_lib.describe_range_bits(int(counts_h[n + 1]))}: "
               f"|feat| / sqrt(C) * {_lib.PLANES_W_SCALE:g} >= {_lib.F16_MAX:g} or non-finite)")
        if on_overflow == "raise":
            raise PopeRangeError(msg)
        warnings.warn(msg + "; re-running with the fp32-MFMA contraction")
        del conf, ws, b_ids, i_ids, j_ids, mconf, mk0, mk1
        return dense_match(feat0, feat1, hw0_c, hw1_c, hw0_i, thr, border_rm, temperature, "f32", on_overflow, want_conf)
    m = int(counts_h[n])
    b_ids, i_ids, j_ids, mconf = b_ids[:m], i_ids[:m], j_ids[:m], mconf[:m]
    return {
        "conf_matrix": conf,
        "b_ids": b_ids, "i_ids": i_ids, "j_ids": j_ids,
        "gt_mask": mconf == 0, "m_bids": b_ids,          # eval: mconf > thr, the `!= 0` filter is a no-op
        "mkpts0_c": mk0[:m], "mkpts1_c": mk1[:m], "mconf": mconf,
        "counts": counts_h[:n],
    }


class CoarseMatching(nn.Module):
    """Drop-in for the reference module (eval, match_type='dual_softmax')."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.thr = config["thr"]
        self.border_rm = config["border_rm"]
        self.train_coarse_percent = config["train_coarse_percent"]
        self.train_pad_num_gt_min = config["train_pad_num_gt_min"]
        self.match_type = config["match_type"]
        if self.match_type == "dual_softmax":
            self.temperature = config["dsmax_temperature"]
        else:
            raise NotImplementedError("pope_amd: only the dual_softmax matcher is on the hot path "
                                      "(default_cfg; sinkhorn needs the absent superglue.py in the reference too)")

    def forward(self, feat_c0, feat_c1, data, mask_c0=None, mask_c1=None):
        if mask_c0 is not None or "mask0" in data:
            raise NotImplementedError("pope_amd: padding masks are a training-time path (matcher.py:62-64)")
        if self.training:
            raise NotImplementedError("pope_amd: inference only (call .eval())")
        out = dense_match(feat_c0, feat_c1, data["hw0_c"], data["hw1_c"], data["hw0_i"], self.thr, self.border_rm,
                          self.temperature)
        out.pop("counts")
        data.update(out)


class Matcher(nn.Module):
    """Drop-in for the reference `Matcher` (src/matcher/matcher.py:12-85): same constructor, same in-place
    `forward(data, only_att_fea=False)` protocol and published keys, same checkpoint layout (`matcher.`
    prefix stripped on load, :81-85).  Every stage — CNN, linear-attention transformers, coarse matching, fine windows and
    sub-pixel refinement — is a call into libpope_hip.so (pope_amd/loftr.py, conv.hip / loftr.hip / match.hip / fine.hip);
    only the position code (a cached table) and tensor views are torch."""

    def __init__(self, config):
        super().__init__()
        from . import loftr
        self.config = config
        self.backbone = loftr.build_backbone(config)
        self.pos_encoding = loftr.PositionEncodingSine(config["coarse"]["d_model"],
                                                       temp_bug_fix=config["coarse"]["temp_bug_fix"])
        self.loftr_coarse = loftr.LocalFeatureTransformer(config["coarse"])
        self.coarse_matching = CoarseMatching(config["match_coarse"])
        self.fine_preprocess = loftr.FinePreprocess(config)
        self.loftr_fine = loftr.LocalFeatureTransformer(config["fine"])
        self.fine_matching = loftr.FineMatching()
        # use_graph = True: the shape-static front of the forward pass (CNN, position code, coarse transformer: ~250 kernel
        # launches at the drivers' batch of three pairs) is replayed from a captured HIP graph from the third call with
        # the same input shapes on.  Same kernels, same order: bit-identical to the eager launches (tests/test_gpu_loftr.py).
        # Off by default because it buys nothing on this path: measured 4.23 -> 4.22 ms per 3-pair call, 17.32 -> 17.27 ms at
        # 24 pairs, driver step 7.34 -> 7.16 ms (scripts/loftr_time.py) — the C-side launch loops already keep the GPU fed,
        # the time is the small-grid kernels themselves.
        self.use_graph = False
        self._graphs, self._gsrc = {}, None

    def _features(self, im0, im1):
        """matcher.py:46-60: (feat_c0, feat_c1) after the coarse transformer [n, L, 256] and the fine maps (feat_f0, feat_f1)."""
        n = im0.size(0)
        if im0.shape[2:] == im1.shape[2:]:  # one CNN launch sequence for both images (matcher.py:46-48)
            feats_c, feats_f = self.backbone(torch.cat([im0, im1], 0))
            (feat_c0, feat_c1), (feat_f0, feat_f1) = feats_c.split(n), feats_f.split(n)
        else:
            (feat_c0, feat_f0), (feat_c1, feat_f1) = self.backbone(im0), self.backbone(im1)
        hw_c = (feat_c0.shape[2:], feat_c1.shape[2:])
        feat_c0 = self.pos_encoding(feat_c0).flatten(2).transpose(1, 2)   # 'n c h w -> n (h w) c'
        feat_c1 = self.pos_encoding(feat_c1).flatten(2).transpose(1, 2)
        feat_c0, feat_c1 = self.loftr_coarse(feat_c0, feat_c1)
        return feat_c0, feat_c1, feat_f0, feat_f1, hw_c

    def _features_graphed(self, im0, im1):
        """`_features` through a HIP graph per (shapes, device): call 1 runs eagerly (builds every weight cache, checks
        the range flags), call 2 captures, later calls copy the inputs into the graph's static buffers and replay.
        The f16x3 range flags of the captured launches are read after the replay (one synchronisation, where the eager
        path has two); a raised flag retires the graph and the eager path — with its warnings and fp32 re-runs — takes over."""
        from . import loftr
        if not (self.use_graph and im0.size(0) > 0 and im0.dtype == torch.float32 and im1.dtype == torch.float32
                and not torch.cuda.is_current_stream_capturing()):
            return self._features(im0, im1)
        # a captured graph has the device pointers of the weight planes / folded BatchNorm matrices baked in, and any edit of
        # a parameter or buffer makes the eager path rebuild (and free) them: the key carries every tensor's address and
        # in-place version, so a stale graph is simply never looked up again
        if self._gsrc is None:
            self._gsrc = _lib.param_slots(self, buffers=True)
        key = (tuple(im0.shape), tuple(im1.shape), im0.device.index, _lib.slots_key(self._gsrc))
        ent = self._graphs.get(key)
        if ent is None:
            if len(self._graphs) >= 16:   # every graph owns its workspaces: keep the set small
                self._graphs.clear()
            self._graphs[key] = {}
            return self._features(im0, im1)
        if ent.get("retired"):
            return self._features(im0, im1)
        if "graph" not in ent:
            s0, s1 = im0.clone(), im1.clone()
            graph = torch.cuda.CUDAGraph()
            loftr.DEFERRED_FLAGS = []
            try:
                with torch.cuda.graph(graph):
                    outs = self._features(s0, s1)
                flags = loftr.DEFERRED_FLAGS
            except Exception:   # noqa: BLE001 — a capture the runtime refuses is not an error of the forward pass
                ent["retired"] = True
                loftr.DEFERRED_FLAGS = None
                torch.cuda.synchronize(im0.device)
                return self._features(im0, im1)
            finally:
                loftr.DEFERRED_FLAGS = None
            ent.update(graph=graph, s0=s0, s1=s1, outs=outs, flags=flags)
        ent["s0"].copy_(im0)
        ent["s1"].copy_(im1)
        ent["graph"].replay()
        if ent["flags"] and int(torch.stack(ent["flags"]).max()):
            ent["retired"] = True
            return self._features(im0, im1)
        return ent["outs"]

    def _apply(self, fn, *a, **k):
        self._graphs, self._gsrc = {}, None
        return super()._apply(fn, *a, **k)

    @torch.no_grad()
    def forward(self, data, only_att_fea=False):
        im0, im1 = data["image0"], data["image1"]
        require_cuda(im0, "Matcher")
        require_cuda(im1, "Matcher")
        if "mask0" in data:
            raise NotImplementedError("pope_amd: padding masks are a training-time path (matcher.py:62-64)")
        n = im0.size(0)
        data.update({"bs": n, "hw0_i": im0.shape[2:], "hw1_i": im1.shape[2:]})
        feat_c0, feat_c1, feat_f0, feat_f1, (hw0_c, hw1_c) = self._features_graphed(im0, im1)
        data.update({"hw0_c": hw0_c, "hw1_c": hw1_c, "hw0_f": feat_f0.shape[2:], "hw1_f": feat_f1.shape[2:]})
        if only_att_fea:   # the caller keeps these: never hand out a graph's static buffers
            return feat_c0.clone(), feat_c1.clone()
        self.coarse_matching(feat_c0, feat_c1, data)
        win0, win1 = self.fine_preprocess(feat_f0, feat_f1, feat_c0, feat_c1, data)
        if win0.size(0) != 0:
            win0, win1 = self.loftr_fine(win0, win1)
        self.fine_matching(win0, win1, data)

    def load_state_dict(self, state_dict, *args, **kwargs):
        # src/matcher/matcher.py:81-85: the keys are renamed IN THE CALLER'S dict (observable afterwards)
        for old in [k for k in state_dict if k.startswith("matcher.")]:
            state_dict[old[len("matcher."):]] = state_dict.pop(old)
        out = super().load_state_dict(state_dict, *args, **kwargs)
        # parameters were overwritten in place, or (assign=True) replaced: the derived caches (folded BatchNorm, weight
        # planes) and the graph keys notice by themselves — they look the tensors up through their slots on every call and
        # key on address + version (_lib.slots_key); the old graphs would only sit on their workspaces
        self._graphs = {}
        return out
