"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's per-pair driver step
(eval_linemod_json.py:65-127,150) on the oracle's DINOv2 and LoFTR restatements, sequential and batch-1
exactly like the reference loop.  Pinned by tests/golden/driver_pair.npz, which oracle/gen_golden.py
captures from the reference's own DINOv2 and Matcher modules driven by the same loop."""
import numpy as np
import torch
import torch.nn.functional as F

from . import dinov2_ref, loftr_ref


@torch.no_grad()
def locate_and_match(vit_sd, matcher_sd, matcher_cfg, ref_tensor, crop_tensors, gray_ref, gray_crops, conf_thr=0.9):
    cls = lambda x: dinov2_ref.forward_features(vit_sd, x)["x_norm_clstoken"]  # noqa: E731  dinov2_utils.py:106-111
    ref_fea = cls(ref_tensor)
    scores, similarity_score, top = [], np.array([0, 0, 0], np.float32), [-1, -1, -1]
    for p in range(crop_tensors.shape[0]):
        score = F.cosine_similarity(ref_fea, cls(crop_tensors[p:p + 1]), dim=1, eps=1e-8)   # :93
        scores.append(score.item())
        if (score.item() > similarity_score).any():                                          # :94
            k = int(np.argmin(similarity_score))                                             # :98
            similarity_score[k] = score.item()
            top[k] = p
    res = {"scores": np.array(scores, np.float32), "slot_scores": similarity_score, "slot_index": np.array(top),
           "mkpts0": [], "mkpts1": [], "mconf": [], "matching_score": np.zeros(3, np.int64)}
    for s in range(3):                                                                        # :108-125
        if top[s] < 0:
            res["mkpts0"].append(np.zeros((0, 2), np.float32)), res["mkpts1"].append(np.zeros((0, 2), np.float32))
            res["mconf"].append(np.zeros((0,), np.float32))
            continue
        out = loftr_ref.matcher_forward(matcher_sd, matcher_cfg, gray_ref, gray_crops[top[s]:top[s] + 1])
        res["mkpts0"].append(out["mkpts0_f"].numpy()), res["mkpts1"].append(out["mkpts1_f"].numpy())
        res["mconf"].append(out["mconf"].numpy())
        res["matching_score"][s] = int((out["mconf"].numpy() > conf_thr).sum())
    res["best_slot"] = int(np.argmax(res["matching_score"]))                                  # :150
    res["best_proposal"] = int(top[res["best_slot"]])
    return res
