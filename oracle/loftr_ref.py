"""TEST INFRASTRUCTURE ONLY — CPU restatement (torch fp32, functional, unfused) of the LoFTR `Matcher`
stages of the reference (src/matcher/matcher.py:29-79), used by tests/, smoke() and oracle/gen_golden.py
as the checker for pope_amd.matcher.Matcher.  Never imported by the product.

Pinned: oracle/gen_golden.py runs the reference's own `Matcher` (imported from /root/reference with
in-memory stand-ins for the two absent third-party modules, yacs and kornia — SURVEY.md §8c) on seeded
synthetic weights and checks every stage below against it before writing tests/golden/loftr_*.npz.
The kornia boundary (create_meshgrid / dsnt.spatial_expectation2d, fine_matching.py:49-50) is restated
from its published closed form and is NOT pinned by any reference fixture ("parity unpinned" for
expec_f / mkpts1_f beyond that closed form).

All functions take the checkpoint-layout state dict `sd` (211 keys) and a key prefix.
"""
import math

import torch
import torch.nn.functional as F

from . import coarse_match_ref


def _bn(sd, p, x):  # nn.BatchNorm2d in eval mode, eps 1e-5
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, 1e-5)


def basic_block(sd, p, x, stride):
    """backbone/resnet_fpn.py:15-40."""
    y = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1)))
    y = _bn(sd, p + ".bn2", F.conv2d(y, sd[p + ".conv2.weight"], None, 1, 1))
    if stride != 1:
        x = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride))
    return F.relu(x + y)


def resnet_fpn_8_2(sd, x, p="backbone"):
    """backbone/resnet_fpn.py:100-118."""
    x0 = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], None, 2, 3)))
    feats, h = [], x0
    for li, stride in ((1, 1), (2, 2), (3, 2)):
        h = basic_block(sd, f"{p}.layer{li}.0", h, stride)
        h = basic_block(sd, f"{p}.layer{li}.1", h, 1)
        feats.append(h)
    x1, x2, x3 = feats

    def outconv2(name, t):
        t = F.conv2d(t, sd[f"{p}.{name}.0.weight"], None, 1, 1)
        t = F.leaky_relu(_bn(sd, f"{p}.{name}.1", t), 0.01)
        return F.conv2d(t, sd[f"{p}.{name}.3.weight"], None, 1, 1)

    up = lambda t: F.interpolate(t, scale_factor=2.0, mode="bilinear", align_corners=True)  # noqa: E731
    x3_out = F.conv2d(x3, sd[p + ".layer3_outconv.weight"])
    x2_out = outconv2("layer2_outconv2", F.conv2d(x2, sd[p + ".layer2_outconv.weight"]) + up(x3_out))
    x1_out = outconv2("layer1_outconv2", F.conv2d(x1, sd[p + ".layer1_outconv.weight"]) + up(x2_out))
    return x3_out, x1_out


def position_encoding(d_model, h, w, temp_bug_fix=False, max_shape=(256, 256)):
    """utils/position_encoding.py:21-35, built at max_shape and sliced like :42."""
    pe = torch.zeros(d_model, *max_shape)
    y_pos = torch.ones(max_shape).cumsum(0).float().unsqueeze(0)
    x_pos = torch.ones(max_shape).cumsum(1).float().unsqueeze(0)
    idx = torch.arange(0, d_model // 2, 2).float()
    if temp_bug_fix:
        div = torch.exp(idx * (-math.log(10000.0) / (d_model // 2)))
    else:  # :28 — operator precedence makes the factor floor(-ln(1e4)/d_model / 2) = -1
        div = torch.exp(idx * (-math.log(10000.0) / d_model // 2))
    div = div[:, None, None]
    pe[0::4], pe[1::4] = torch.sin(x_pos * div), torch.cos(x_pos * div)
    pe[2::4], pe[3::4] = torch.sin(y_pos * div), torch.cos(y_pos * div)
    return pe[None, :, :h, :w]


def linear_attention(q, k, v, eps=1e-6):
    """loftr_module/linear_attention.py:20-47."""
    Q, K = F.elu(q) + 1, F.elu(k) + 1
    S = v.size(1)
    v = v / S
    KV = torch.einsum("nshd,nshv->nhdv", K, v)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(dim=1)) + eps)
    return (torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * S).contiguous()


def encoder_layer(sd, p, x, source, nhead):
    """loftr_module/transformer.py:35-58."""
    n, d = x.size(0), x.size(2)
    q = F.linear(x, sd[p + ".q_proj.weight"]).view(n, -1, nhead, d // nhead)
    k = F.linear(source, sd[p + ".k_proj.weight"]).view(n, -1, nhead, d // nhead)
    v = F.linear(source, sd[p + ".v_proj.weight"]).view(n, -1, nhead, d // nhead)
    msg = F.linear(linear_attention(q, k, v).view(n, -1, d), sd[p + ".merge.weight"])
    msg = F.layer_norm(msg, (d,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
    msg = F.linear(F.relu(F.linear(torch.cat([x, msg], dim=2), sd[p + ".mlp.0.weight"])), sd[p + ".mlp.2.weight"])
    msg = F.layer_norm(msg, (d,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
    return x + msg


def local_feature_transformer(sd, p, feat0, feat1, layer_names, nhead):
    """loftr_module/transformer.py:85-106."""
    for i, name in enumerate(layer_names):
        lp = f"{p}.layers.{i}"
        if name == "self":
            feat0 = encoder_layer(sd, lp, feat0, feat0, nhead)
            feat1 = encoder_layer(sd, lp, feat1, feat1, nhead)
        else:
            feat0 = encoder_layer(sd, lp, feat0, feat1, nhead)
            feat1 = encoder_layer(sd, lp, feat1, feat0, nhead)
    return feat0, feat1


def fine_preprocess(sd, feat_f0, feat_f1, feat_c0, feat_c1, b_ids, i_ids, j_ids, W, stride, p="fine_preprocess"):
    """loftr_module/fine_preprocess.py:29-59 (cat_c_feat=True)."""
    if b_ids.numel() == 0:
        e = torch.empty(0, W * W, feat_f0.shape[1])
        return e, e.clone()

    def unfold(f):
        u = F.unfold(f, kernel_size=(W, W), stride=stride, padding=W // 2)      # [n, c*ww, l]
        n, _, l = u.shape
        return u.view(n, -1, W * W, l).permute(0, 3, 2, 1)                       # n l ww c

    u0, u1 = unfold(feat_f0)[b_ids, i_ids], unfold(feat_f1)[b_ids, j_ids]
    c_win = F.linear(torch.cat([feat_c0[b_ids, i_ids], feat_c1[b_ids, j_ids]], 0), sd[p + ".down_proj.weight"],
                     sd[p + ".down_proj.bias"])
    cat = torch.cat([torch.cat([u0, u1], 0), c_win[:, None, :].repeat(1, W * W, 1)], -1)
    out = F.linear(cat, sd[p + ".merge_feat.weight"], sd[p + ".merge_feat.bias"])
    return torch.chunk(out, 2, dim=0)


def fine_matching(feat_f0, feat_f1, mkpts0_c, mkpts1_c, scale):
    """utils/fine_matching.py:15-74 -> (expec_f [M,3], mkpts0_f, mkpts1_f); `scale` = hw0_i[0]/hw0_f[0]."""
    M, WW, C = feat_f0.shape
    if M == 0:
        return torch.empty(0, 3), mkpts0_c, mkpts1_c
    W = int(math.sqrt(WW))
    sim = torch.einsum("mc,mrc->mr", feat_f0[:, WW // 2, :], feat_f1)
    heat = torch.softmax((1.0 / C ** 0.5) * sim, dim=1)   # :45-46 multiplies by the reciprocal
    # kornia create_meshgrid(W, W, normalized=True): [1,W,W,2] of (x, y) in linspace(-1, 1, W);
    # dsnt.spatial_expectation2d(heat, normalized=True) = sum heat * grid
    lin = torch.linspace(-1, 1, W)
    gy, gx = torch.meshgrid(lin, lin, indexing="ij")
    grid = torch.stack([gx, gy], -1).reshape(1, WW, 2)
    coords = (heat.view(M, WW, 1) * grid).sum(1)
    var = torch.sum(grid ** 2 * heat.view(-1, WW, 1), dim=1) - coords ** 2
    std = torch.sum(torch.sqrt(torch.clamp(var, min=1e-10)), -1)
    return torch.cat([coords, std[:, None]], -1), mkpts0_c, mkpts1_c + coords * (W // 2) * scale


def matcher_forward(sd, cfg, image0, image1):
    """src/matcher/matcher.py:29-79 -> dict of everything the reference publishes (+ feat_c0/1, feat_f0/1)."""
    n = image0.size(0)
    hw0_i, hw1_i = image0.shape[2:], image1.shape[2:]
    if hw0_i == hw1_i:
        fc, ff = resnet_fpn_8_2(sd, torch.cat([image0, image1], 0))
        (c0, c1), (f0, f1) = fc.split(n), ff.split(n)
    else:
        (c0, f0), (c1, f1) = resnet_fpn_8_2(sd, image0), resnet_fpn_8_2(sd, image1)
    hw0_c, hw1_c, hw0_f = c0.shape[2:], c1.shape[2:], f0.shape[2:]
    d = cfg["coarse"]["d_model"]
    bug = cfg["coarse"]["temp_bug_fix"]
    t0 = (c0 + position_encoding(d, *hw0_c, temp_bug_fix=bug)).flatten(2).transpose(1, 2)
    t1 = (c1 + position_encoding(d, *hw1_c, temp_bug_fix=bug)).flatten(2).transpose(1, 2)
    t0, t1 = local_feature_transformer(sd, "loftr_coarse", t0, t1, cfg["coarse"]["layer_names"], cfg["coarse"]["nhead"])
    mc = cfg["match_coarse"]
    out = coarse_match_ref.dense_match(t0, t1, tuple(hw0_c), tuple(hw1_c), tuple(hw0_i), thr=mc["thr"],
                                        border_rm=mc["border_rm"], temperature=mc["dsmax_temperature"])
    W = cfg["fine_window_size"]
    w0, w1 = fine_preprocess(sd, f0, f1, t0, t1, out["b_ids"], out["i_ids"], out["j_ids"], W, hw0_f[0] // hw0_c[0])
    if w0.size(0) != 0:
        w0, w1 = local_feature_transformer(sd, "loftr_fine", w0, w1, cfg["fine"]["layer_names"], cfg["fine"]["nhead"])
    expec_f, mk0f, mk1f = fine_matching(w0, w1, out["mkpts0_c"], out["mkpts1_c"], hw0_i[0] / hw0_f[0])
    out.update({"feat_c0": t0, "feat_c1": t1, "feat_f0": f0, "feat_f1": f1, "expec_f": expec_f, "mkpts0_f": mk0f,
                "mkpts1_f": mk1f, "hw0_c": tuple(hw0_c), "hw1_c": tuple(hw1_c), "hw0_f": tuple(hw0_f)})
    return out
