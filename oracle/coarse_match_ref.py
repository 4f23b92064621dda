"""CPU restatement (torch fp32) of the reference dense matcher (eval mode).

TEST INFRASTRUCTURE — see oracle/__init__.py.

Reference anchors (paths relative to /root/reference):
  similarity + dual softmax   src/matcher/utils/coarse_matching.py:106-119
  threshold / border / MNN    src/matcher/utils/coarse_matching.py:174-196, mask_border :8-25
  index -> pixel mapping      src/matcher/utils/coarse_matching.py:241-259
  streaming top-3 vote        eval_linemod_json.py:71,94-101 ; argmax :146
"""
import numpy as np
import torch
import torch.nn.functional as F

NEG_INF = 1e9  # coarse_matching.py:6 (INF)


def conf_matrix(feat0, feat1, temperature=0.1, mask0=None, mask1=None):
    """coarse_matching.py:106-119: both inputs divided by sqrt(C), all-pairs
    dot products divided by the temperature, product of the two softmaxes."""
    c = feat0.shape[-1]
    f0 = feat0 / c ** .5
    f1 = feat1 / c ** .5
    sim = torch.einsum("nlc,nsc->nls", f0, f1) / temperature
    if mask0 is not None:
        sim.masked_fill_(~(mask0[..., None] * mask1[:, None]).bool(), -NEG_INF)
    return F.softmax(sim, 1) * F.softmax(sim, 2)


def coarse_match(conf, hw0_c, hw1_c, hw0_i, thr=0.2, border_rm=2):
    """coarse_matching.py:174-196,241-259 (eval; no padding masks).

    Returns dict of b_ids, i_ids, j_ids (int64), mconf (fp32), mkpts0_c,
    mkpts1_c (fp32 [M,2], (x,y) order, scaled by hw0_i[0]/hw0_c[0])."""
    n, L, S = conf.shape
    h0, w0 = hw0_c
    h1, w1 = hw1_c
    assert L == h0 * w0 and S == h1 * w1
    m = (conf > thr).view(n, h0, w0, h1, w1).clone()
    b = border_rm
    if b > 0:
        m[:, :b] = False
        m[:, :, :b] = False
        m[:, :, :, :b] = False
        m[:, :, :, :, :b] = False
        m[:, -b:] = False
        m[:, :, -b:] = False
        m[:, :, :, -b:] = False
        m[:, :, :, :, -b:] = False
    m = m.view(n, L, S)
    m = m & (conf == conf.max(dim=2, keepdim=True)[0]) & (conf == conf.max(dim=1, keepdim=True)[0])
    mask_v, all_j = m.max(dim=2)
    b_ids, i_ids = torch.where(mask_v)
    j_ids = all_j[b_ids, i_ids]
    mconf = conf[b_ids, i_ids, j_ids]
    scale = hw0_i[0] / hw0_c[0]
    mk0 = torch.stack([i_ids % w0, i_ids // w0], dim=1) * scale
    mk1 = torch.stack([j_ids % w1, j_ids // w1], dim=1) * scale
    keep = mconf != 0
    return {
        "b_ids": b_ids, "i_ids": i_ids, "j_ids": j_ids,
        "gt_mask": mconf == 0, "m_bids": b_ids[keep],
        "mkpts0_c": mk0[keep], "mkpts1_c": mk1[keep], "mconf": mconf[keep],
    }


@torch.no_grad()
def dense_match(feat0, feat1, hw0_c, hw1_c, hw0_i, thr=0.2, border_rm=2, temperature=0.1):
    conf = conf_matrix(feat0, feat1, temperature)
    out = coarse_match(conf, hw0_c, hw1_c, hw0_i, thr, border_rm)
    out["conf_matrix"] = conf
    return out


def cls_cosine(ref, fea, eps=1e-8):
    """F.cosine_similarity(ref, fea, dim=1, eps) restated: each norm is clamped
    separately (torch >= 1.12 semantics).  eval_linemod_json.py:94."""
    ref = ref.float()
    fea = fea.float()
    num = (ref * fea).sum(1)
    return num / (ref.norm(dim=1).clamp_min(eps) * fea.norm(dim=1).clamp_min(eps))


def streaming_top3(scores):
    """eval_linemod_json.py:71,95-101: three slots initialised to 0; a proposal
    enters iff score > min(slots) (strict), replacing the FIRST minimum slot.
    Returns (slot_scores float32[3], slot_index int[3] (-1 = empty))."""
    slots = np.array([0, 0, 0], np.float32)
    idx = [-1, -1, -1]
    for p, s in enumerate(scores):
        s = float(s)
        if (s > slots).any():
            k = int(np.argmin(slots))
            slots[k] = s
            idx[k] = p
    return slots, np.array(idx, np.int64)
