"""Hot-path subset of the reference façade `pope_model_api.py` (star-imported by the drivers,
eval_linemod_json.py:1).  Exports, under the reference's names, everything of that namespace that
lies on the accelerated path, including the caller-side geometry either side of it: the proposal crops
and their intrinsics (`get_image_crop_resize`, `get_K_crop_resize`; batched: `crop_proposals`) and the pose solver
(`estimate_pose`, `relative_pose_error`; batched: `estimate_pose_batch`).  Names that belong to out-of-scope stages (the SAM
proposal generator, dataset helpers) are not re-implemented here — the reference's own modules keep providing them.

Unlike the reference façade, importing this module has no side effects (the reference builds the
LoFTR `matcher` singleton from weights/matcher.pth at import, pope_model_api.py:177-185).
"""
import json  # noqa: F401  (re-exported like the reference does)
import os  # noqa: F401
import time  # noqa: F401

import numpy as np  # noqa: F401
import torch  # noqa: F401
import torch.nn.functional as F  # noqa: F401

from .crops import crop_proposals, expand_box, get_affine_transform, get_image_crop_resize, get_K_crop_resize  # noqa: F401
from .dinov2_utils import get_cls_token_torch, load_dinov2_model, set_torch_image  # noqa: F401
from .driver import locate_and_match, locate_match_pose_u8  # noqa: F401
from .matcher import CoarseMatching, Matcher, default_cfg, dense_match  # noqa: F401
from .ops import cls_cosine, streaming_top3  # noqa: F401
from .pipeline import PairPipeline, gather_counts, shard_range  # noqa: F401
from .pose import estimate_pose, estimate_pose_batch, relative_pose_error  # noqa: F401


def build_matcher(weights="weights/matcher.pth", device="cuda:0", state_dict=None):
    """The module-level `matcher` singleton of the reference façade (pope_model_api.py:177-185: Matcher(default_cfg),
    `torch.load("weights/matcher.pth")['state_dict']` loaded with strict=False, .eval(), moved to a GPU) as an
    explicit call; `state_dict` overrides the file (synthetic weights in tests)."""
    matcher = Matcher(default_cfg)
    if state_dict is None:
        state_dict = torch.load(weights, map_location="cpu")["state_dict"]
    matcher.load_state_dict(state_dict, strict=False)
    return matcher.eval().to(device)


def vote_top3(model, ref_tensor, crop_tensors):
    """Hot loop #1 of the drivers (eval_linemod_json.py:65,74-101) batched: CLS token of the reference
    crop vs P proposal crops -> cosine scores -> streaming top-3 slots (same slot order as the
    reference's sequential loop).  Returns (scores[P] (cuda), slot_scores[3], slot_index[3])."""
    ref = get_cls_token_torch(model, ref_tensor)
    fea = get_cls_token_torch(model, crop_tensors)
    scores = cls_cosine(ref, fea, eps=1e-8)
    slots, idx = streaming_top3(scores.cpu().numpy())
    return scores, slots, idx
