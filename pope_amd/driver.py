"""Per-pair driver step of the reference evaluation loops (eval_linemod_json.py:65-127 and the
identical bodies of eval_onepose_json.py / eval_ycb_json.py), restructured for one
big GPU: what the reference does as P batch-1 DINOv2 forwards, P host round trips for the cosine score
and three batch-1 LoFTR calls becomes one batched extraction, one cosine kernel, one host-side slot
vote (order-dependent by definition) and ONE Matcher call over the occupied slots.

Inputs are already-preprocessed tensors: SAM proposal generation, cv2 cropping / colour conversion and
the pose solver are outside the accelerated path (SURVEY.md §8 'OUT').
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
