"""Per-pair driver step of the reference evaluation loops (eval_linemod_json.py:65-127 and the
identical bodies of eval_onepose_json.py / eval_ycb_json.py), restructured for one
big GPU: what the reference does as P batch-1 DINOv2 forwards, P host round trips for the cosine score
and three batch-1 LoFTR calls becomes one batched extraction, one cosine kernel, one host-side slot
vote (order-dependent by definition) and ONE Matcher call over the occupied slots.

`locate_and_match` takes already-preprocessed tensors; `locate_and_match_u8` starts from the uint8 crops;
`locate_match_pose_u8` is the whole per-pair body of the loop after SAM: frame + proposal boxes in, pose out — proposal
crops and their intrinsics (crops.py), preprocessing (preprocess.py), DINOv2 vote, LoFTR matches and the essential-matrix
RANSAC (pose.py) all on the card.  SAM proposal generation itself is upstream of the path (SURVEY.md §8 'OUT').
"""
import numpy as np
import torch

from .dinov2_utils import get_cls_token_torch
from .ops import cls_cosine, streaming_top3


@torch.no_grad()
def locate_and_match(dinov2_model, matcher, ref_tensor, crop_tensors, gray_ref, gray_crops, conf_thr=0.9):
    """One query/reference pair.

    ref_tensor   [1,3,h,w]  set_torch_image(image0, center_crop=True)          (eval_linemod_json.py:64)
    crop_tensors [P,3,h,w]  set_torch_image(image_crop_p, center_crop=True)     (:89)
    gray_ref     [1,1,H0,W0] image0 as gray / 255                               (:103-105)
    gray_crops   [P,1,H1,W1] proposal crops as gray / 255 (256x256 in the drivers, :86-88,109-111)

    Returns a dict: `scores` [P] cosine of CLS tokens (:93); `slot_scores` [3] / `slot_index` [3] — the
    reference's `similarity_score` / `top_images` after its streaming loop (:94-101; index -1 = slot never
    filled); per slot `mkpts0`, `mkpts1`, `mconf` (numpy, :118-125); `matching_score` [3] = #(mconf >
    conf_thr) (:121-122); `best_slot` = first argmax (:150) and `best_proposal`.
    A slot that was never filled (fewer than three proposals with a positive score) is skipped with
    matching_score 0; the reference raises on it (:109 indexes an empty list)."""
    if ref_tensor.shape[1:] == crop_tensors.shape[1:]:
        # one launch sequence for the reference image and the P proposals (the kernels are batch-invariant bit for bit,
        # tests/test_gpu_vit.py: the tokens are those of the two separate forwards)
        both = get_cls_token_torch(dinov2_model, torch.cat([ref_tensor, crop_tensors], 0))
        ref, fea = both[:1], both[1:]
    else:
        ref = get_cls_token_torch(dinov2_model, ref_tensor)
        fea = get_cls_token_torch(dinov2_model, crop_tensors)
    scores = cls_cosine(ref, fea, eps=1e-8)
    slot_scores, slot_index = streaming_top3(scores.cpu().numpy())
    filled = [s for s in range(3) if slot_index[s] >= 0]
    out = {"scores": scores, "slot_scores": slot_scores, "slot_index": slot_index,
           "mkpts0": [np.zeros((0, 2), np.float32)] * 3, "mkpts1": [np.zeros((0, 2), np.float32)] * 3,
           "mconf": [np.zeros((0,), np.float32)] * 3, "matching_score": np.zeros(3, np.int64)}
    if filled:
        sel = torch.as_tensor([int(slot_index[s]) for s in filled], device=gray_crops.device)
        batch = {"image0": gray_ref.expand(len(filled), -1, -1, -1).contiguous(),
                 "image1": gray_crops.index_select(0, sel)}
        matcher(batch)
        b = batch["m_bids"].cpu().numpy()
        mk0, mk1, mc = (batch[k].cpu().numpy() for k in ("mkpts0_f", "mkpts1_f", "mconf"))
        for k, s in enumerate(filled):
            rows = b == k
            out["mkpts0"][s], out["mkpts1"][s], out["mconf"][s] = mk0[rows], mk1[rows], mc[rows]
            out["matching_score"][s] = int((mc[rows] > conf_thr).sum())
    out["best_slot"] = int(np.argmax(out["matching_score"]))
    out["best_proposal"] = int(slot_index[out["best_slot"]])
    return out


@torch.no_grad()
def locate_and_match_u8(dinov2_model, matcher, ref_bgr, crops_bgr, conf_thr=0.9):
    """The same step from raw uint8 frames, preprocessing included: `ref_bgr` [H0, W0, 3] and
    `crops_bgr` [P, 256, 256, 3] uint8 BGR (numpy or tensors), as the drivers hold them after cropping
    (eval_linemod_json.py:62-64,83-90,103-111).  One upload of the uint8 frames; resize / centre crop / normalisation of
    all P + 1 DINOv2 inputs and the gray / 255 conversion of all matcher inputs run as batched HIP kernels that are
    bit-identical to the reference's per-image PIL + torchvision + cv2 host calls (pope_amd/preprocess.py)."""
    from .preprocess import gray_batch, set_torch_images
    dev = next(dinov2_model.parameters()).device
    ref = torch.as_tensor(ref_bgr)[None].to(dev)
    crops = torch.as_tensor(crops_bgr).to(dev)
    return locate_and_match(dinov2_model, matcher, set_torch_images(ref, center_crop=True), set_torch_images(crops, center_crop=True),
                            gray_batch(ref), gray_batch(crops), conf_thr)


@torch.no_grad()
def locate_match_pose_u8(dinov2_model, matcher, ref_bgr, frame_bgr, bboxes_xywh, K0, K1, conf_thr=0.9, ransac_thr=0.5, ransac_conf=0.99,
                         out_size=256):
    """eval_linemod_json.py:62-127,150-160 for one query: `ref_bgr` [H0, W0, 3] uint8 (the reference crop, `image0`),
    `frame_bgr` [H, W, 3] uint8 (`image1`), `bboxes_xywh` [P, 4] SAM proposal boxes, `K0` / `K1` the two cameras.

    proposals -> expanded boxes, 256 x 256 crops and their K (crops.crop_proposals, one launch) -> `locate_and_match_u8`
    (DINOv2 vote + ONE LoFTR call over the occupied slots) -> `estimate_pose(mkpts0, mkpts1, K0, K_crop[best], 0.5, 0.99)`
    on the best slot's matches (ALL of them: the 0.9 confidence only ranks the slots, :118-119,150-160).
    Adds to `locate_and_match`'s dict: `boxes` [P, 4], `K_crops` [P, 3, 3], `pre_bbox`, `pre_K` (the chosen proposal's) and
    `pose` = (R, t, inliers) or None."""
    from .crops import crop_proposals
    from .pose import estimate_pose
    dev = next(dinov2_model.parameters()).device
    frame = torch.as_tensor(frame_bgr).to(dev)
    prop = crop_proposals(frame, bboxes_xywh, K1, out_size=out_size)
    out = locate_and_match_u8(dinov2_model, matcher, ref_bgr, prop["crops"], conf_thr)
    out["boxes"], out["K_crops"] = prop["boxes"], prop["K"]
    best = out["best_proposal"]
    if best < 0:       # no proposal entered a slot (the reference raises at :109)
        out["pre_bbox"], out["pre_K"], out["pose"] = None, None, None
        return out
    out["pre_bbox"], out["pre_K"] = prop["boxes"][best], prop["K"][best]
    s = out["best_slot"]
    out["pose"] = estimate_pose(out["mkpts0"][s], out["mkpts1"][s], K0, out["pre_K"], ransac_thr, ransac_conf, device=dev)
    return out
