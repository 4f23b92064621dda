#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE's own Python code.

Runs only in the build container (needs /root/reference; read-only import,
nothing is copied).  It
  1. imports the reference DINOv2 (`dinov2.dinov2.models.vision_transformer`)
     and the reference `CoarseMatching` (loaded by file path — the file needs
     only torch + einops),
  2. loads seeded synthetic weights (pope_amd.synth.synthetic_state_dict; the
     real checkpoints are not available offline),
  3. checks the CPU restatement in oracle/ against the reference outputs, and
  4. writes the reference outputs as small fixtures.

Usage:  python oracle/gen_golden.py            (from the repo root)
"""
import importlib.util
import logging
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(1, REF)
logging.disable(logging.WARNING)

from oracle import coarse_match_ref as cm_ref  # noqa: E402
from oracle import dinov2_ref  # noqa: E402
from pope_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def load_reference_vit(sd):
    from dinov2.dinov2.models import vision_transformer as vits
    # == load_dinov2_model() (dinov2_utils.py:38-47) minus the config/weights files
    m = vits.vit_small(patch_size=14, img_size=518, init_values=1e-5, ffn_layer="mlp",
                       block_chunks=0, qkv_bias=True, proj_bias=True, ffn_bias=True)
    m.load_state_dict(sd, strict=True)
    return m.eval()


def load_reference_coarse_matching():
    spec = importlib.util.spec_from_file_location(
        "ref_coarse_matching", os.path.join(REF, "src/matcher/utils/coarse_matching.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cfg = {"thr": 0.2, "border_rm": 2, "match_type": "dual_softmax", "dsmax_temperature": 0.1,
           "train_coarse_percent": 0.4, "train_pad_num_gt_min": 200}  # cvpr_ds_config.py:30-40
    return mod.CoarseMatching(cfg).eval()


def load_reference_matcher(cfg_overrides=None):
    """Reference `Matcher` (src/matcher/matcher.py) with in-memory stand-ins for the two third-party modules
    absent from this image: `yacs.config.CfgNode` (a dict with attribute access) and the two
    kornia functions fine_matching.py:5-6 imports, restated from their published closed forms."""
    import types

    class CfgNode(dict):
        __getattr__ = dict.__getitem__

        def __setattr__(self, k, v):
            self[k] = v

    def create_meshgrid(h, w, normalized_coordinates=True, device=None):
        xs = torch.linspace(-1, 1, w, device=device) if normalized_coordinates else torch.arange(w, device=device).float()
        ys = torch.linspace(-1, 1, h, device=device) if normalized_coordinates else torch.arange(h, device=device).float()
        gy, gx = torch.meshgrid(ys, xs, indexing="ij")
        return torch.stack([gx, gy], -1)[None]

    def spatial_expectation2d(heatmap, normalized_coordinates=True):
        b, c, h, w = heatmap.shape
        grid = create_meshgrid(h, w, normalized_coordinates, heatmap.device).reshape(-1, 2)
        return (heatmap.reshape(b, c, -1, 1) * grid).sum(2)

    def module(name, **attrs):
        m = sys.modules.setdefault(name, types.ModuleType(name))
        for k, v in attrs.items():
            setattr(m, k, v)
        return m

    module("yacs", config=module("yacs.config", CfgNode=CfgNode))
    dsnt = module("kornia.geometry.subpix.dsnt", spatial_expectation2d=spatial_expectation2d)
    module("kornia"), module("kornia.geometry"), module("kornia.geometry.subpix", dsnt=dsnt)
    module("kornia.utils"), module("kornia.utils.grid", create_meshgrid=create_meshgrid)
    import copy
    from src.matcher import Matcher, default_cfg
    cfg = copy.deepcopy(default_cfg)
    for k, v in (cfg_overrides or {}).items():
        cfg["match_coarse"][k] = v
    return Matcher(cfg).eval(), cfg


def gen_loftr():
    """tests/golden/loftr_*.npz: the reference Matcher on seeded synthetic weights (synth.synthetic_matcher_state_dict)
    and gray pairs; checks oracle/loftr_ref.py stage by stage first."""
    from oracle import loftr_ref
    from pope_amd.matcher import default_cfg as my_cfg
    sd = synth.synthetic_matcher_state_dict(seed=0)
    digest = sd_digest({k: v for k, v in sd.items() if v.dtype.is_floating_point})
    cases = {
        # name: (n, (h0,w0), (h1,w1), thr)
        "loftr_256": (2, (256, 256), (256, 256), 0.2),          # the drivers' shape (eval_linemod_json.py:103-111)
        "loftr_256_lowthr": (2, (256, 256), (256, 256), 1e-3),   # thr lowered so the fine stage sees matches
        "loftr_192x256_vs_256x192": (1, (192, 256), (256, 192), 1e-3),
    }
    for name, (n, s0, s1, thr) in cases.items():
        i0, i1 = synth.synthetic_gray_pairs(n, *s0, seed=21)
        if s1 != s0:   # different shapes: image1 = fresh texture with a 192x192 region of image0 pasted 32 rows lower
            i1 = synth.synthetic_gray_pairs(n, *s1, seed=22)[0]
            i1[:, :, 32:224, :] = i0[:, :, :, 32:224]
        ref, cfg = load_reference_matcher({"thr": thr})
        ref.load_state_dict({"matcher." + k: v.clone() for k, v in sd.items()}, strict=True)  # prefix path, matcher.py:81-85
        data = {"image0": i0, "image1": i1}
        with torch.no_grad():
            ref(data)
            fc0, fc1 = ref({"image0": i0, "image1": i1}, only_att_fea=True)
            if s0 == s1:
                bc, bf = ref.backbone(torch.cat([i0, i1], 0))
            else:
                bc, bf = ref.backbone(i0)
        cfg2 = dict(my_cfg, match_coarse=dict(my_cfg["match_coarse"], thr=thr))
        assert {k: v for k, v in cfg.items() if k != "match_coarse"} == {k: v for k, v in my_cfg.items() if k != "match_coarse"}
        with torch.no_grad():
            mine = loftr_ref.matcher_forward(sd, cfg2, i0, i1)
            mbc, mbf = loftr_ref.resnet_fpn_8_2(sd, torch.cat([i0, i1], 0) if s0 == s1 else i0)
        d = {"backbone_c": maxdiff(bc, mbc), "backbone_f": maxdiff(bf, mbf),
             "feat_c0": maxdiff(fc0, mine["feat_c0"]), "feat_c1": maxdiff(fc1, mine["feat_c1"]),
             "conf": maxdiff(data["conf_matrix"], mine["conf_matrix"])}
        for k in ("b_ids", "i_ids", "j_ids"):
            assert torch.equal(data[k], mine[k]), k
        for k in ("mconf", "mkpts0_c", "mkpts1_c", "mkpts0_f", "mkpts1_f", "expec_f"):
            d[k] = maxdiff(data[k], mine[k]) if data[k].numel() else 0.0
        print(name, "thr", thr, "matches", len(data["b_ids"]), "oracle-vs-reference:", {k: f"{v:.1e}" for k, v in d.items()})
        assert max(d.values()) <= 1e-5, d
        assert tuple(data["hw0_c"]) == mine["hw0_c"] and tuple(data["hw0_f"]) == mine["hw0_f"] and data["W"] == 5
        conf = data["conf_matrix"]
        full = {} if name == "loftr_256" else {   # complete coarse features of pair 0: input of the strict
            "feat_c0_b0": fc0[0].numpy(), "feat_c1_b0": fc1[0].numpy()}   # index-parity test of the HIP matcher
        np.savez(os.path.join(OUT, name + ".npz"), **full,
                 weights_seed=0, weights_digest=digest, thr=np.float64(thr), n=n, shape0=np.array(s0), shape1=np.array(s1),
                 image_digest=np.array([float(i0.double().sum()), float(i1.double().sum())]),
                 backbone_c=bc[:1, :, ::2, ::2].numpy(), backbone_f=bf[:1, ::4, ::8, ::8].numpy(),
                 feat_c0=fc0[:, ::8].numpy(), feat_c1=fc1[:, ::8].numpy(),
                 conf_rowmax=conf.max(2)[0].numpy(), conf_rowarg=conf.max(2)[1].numpy(),
                 conf_colmax=conf.max(1)[0].numpy(), conf_colarg=conf.max(1)[1].numpy(),
                 conf_sum=np.array([float(conf.double().sum())]),
                 b_ids=data["b_ids"].numpy(), i_ids=data["i_ids"].numpy(), j_ids=data["j_ids"].numpy(),
                 mconf=data["mconf"].numpy(), mkpts0_c=data["mkpts0_c"].numpy(), mkpts1_c=data["mkpts1_c"].numpy(),
                 mkpts0_f=data["mkpts0_f"].numpy(), mkpts1_f=data["mkpts1_f"].numpy(), expec_f=data["expec_f"].numpy(),
                 hw0_c=np.array(data["hw0_c"]), hw1_c=np.array(data["hw1_c"]), hw0_f=np.array(data["hw0_f"]),
                 hw1_f=np.array(data["hw1_f"]))


def gen_loftr_large():
    """tests/golden/loftr_512_peaked.npz: the reference Matcher at the OnePose drivers' shape (eval_onepose_json.py:88: 512 x
    512, L = S = 4 096) under the `peaked` synthetic weights (pope_amd/synth.py:peaked_matcher_state_dict — the common-mode
    component of the coarse features projected out, so that the matcher publishes thousands of confident matches as a
    trained checkpoint does), default threshold 0.2.  Strided taps, the full match list, and the calibration mean that
    pins the weights bit for bit on other hosts."""
    from oracle import loftr_ref
    from pope_amd.matcher import default_cfg as my_cfg

    def pre(sd_, img):
        with torch.no_grad():
            return loftr_ref.resnet_fpn_8_2(sd_, img)[0]

    sd = synth.peaked_matcher_state_dict(pre, seed=0)
    mean = sd.pop("_calibration_mean")
    again = synth.peaked_matcher_state_dict(mean, seed=0)
    again.pop("_calibration_mean")
    assert all(torch.equal(sd[k], again[k]) for k in sd)          # the mean alone reproduces the weights
    digest = sd_digest({k: v for k, v in sd.items() if v.dtype.is_floating_point})
    n, s0, thr, name = 1, (512, 512), 0.2, "loftr_512_peaked"
    i0, i1 = synth.synthetic_gray_pairs(n, *s0, seed=23)
    ref, cfg = load_reference_matcher({"thr": thr})
    ref.load_state_dict({"matcher." + k: v.clone() for k, v in sd.items()}, strict=True)
    data = {"image0": i0, "image1": i1}
    with torch.no_grad():
        ref(data)
        fc0, fc1 = ref({"image0": i0, "image1": i1}, only_att_fea=True)
        bc, bf = ref.backbone(torch.cat([i0, i1], 0))
        mine = loftr_ref.matcher_forward(sd, my_cfg, i0, i1)
        mbc, mbf = loftr_ref.resnet_fpn_8_2(sd, torch.cat([i0, i1], 0))
    d = {"backbone_c": maxdiff(bc, mbc), "backbone_f": maxdiff(bf, mbf), "feat_c0": maxdiff(fc0, mine["feat_c0"]),
         "feat_c1": maxdiff(fc1, mine["feat_c1"]), "conf": maxdiff(data["conf_matrix"], mine["conf_matrix"])}
    for k in ("b_ids", "i_ids", "j_ids"):
        assert torch.equal(data[k], mine[k]), k
    for k in ("mconf", "mkpts0_c", "mkpts1_c", "mkpts0_f", "mkpts1_f", "expec_f"):
        d[k] = maxdiff(data[k], mine[k])
    print(name, "matches", len(data["b_ids"]), "of which mconf > 0.9:", int((data["mconf"] > 0.9).sum()),
          "oracle-vs-reference:", {k: f"{v:.1e}" for k, v in d.items()})
    assert max(d.values()) <= 1e-5 and len(data["b_ids"]) > 2000, d
    conf = data["conf_matrix"]
    np.savez(os.path.join(OUT, name + ".npz"), weights_seed=0, weights_digest=digest, outconv_mean=mean.numpy(),
             thr=np.float64(thr), n=n, shape0=np.array(s0), shape1=np.array(s0),
             image_digest=np.array([float(i0.double().sum()), float(i1.double().sum())]),
             backbone_c=bc[:, ::4, ::4, ::4].numpy(), backbone_f=bf[:, ::8, ::16, ::16].numpy(),
             feat_c0=fc0[:, ::16].numpy(), feat_c1=fc1[:, ::16].numpy(),
             conf_rowmax=conf.max(2)[0].numpy(), conf_rowarg=conf.max(2)[1].numpy().astype(np.int32),
             conf_colmax=conf.max(1)[0].numpy(), conf_colarg=conf.max(1)[1].numpy().astype(np.int32),
             conf_sum=np.array([float(conf.double().sum())]),
             b_ids=data["b_ids"].numpy().astype(np.int32), i_ids=data["i_ids"].numpy().astype(np.int32),
             j_ids=data["j_ids"].numpy().astype(np.int32), mconf=data["mconf"].numpy(),
             mkpts1_f=data["mkpts1_f"].numpy(), expec_f=data["expec_f"].numpy(),
             hw0_c=np.array(data["hw0_c"]), hw1_c=np.array(data["hw1_c"]), hw0_f=np.array(data["hw0_f"]),
             hw1_f=np.array(data["hw1_f"]))


def gen_driver():
    """tests/golden/driver_pair.npz: the drivers' per-pair step (eval_linemod_json.py:65-127,150) executed with the
    reference's own DINOv2 and Matcher modules on synthetic proposals (SAM / cv2 are not run: SURVEY.md §8 a-18)."""
    import torch.nn.functional as F
    from oracle import driver_ref
    from pope_amd.matcher import default_cfg as my_cfg
    vit_sd = synth.synthetic_state_dict(seed=0)
    m_sd = synth.synthetic_matcher_state_dict(seed=0)
    vit = load_reference_vit(vit_sd)
    matcher, _ = load_reference_matcher()
    matcher.load_state_dict({k: v.clone() for k, v in m_sd.items()}, strict=True)
    ref_t, crops_t, gray_ref, gray_crops = synth.synthetic_driver_case()
    with torch.no_grad():
        ref_fea = vit(ref_t)                                    # get_cls_token_torch, dinov2_utils.py:106-111
        similarity_score, top_images = np.array([0, 0, 0], np.float32), [[], [], []]
        scores = []
        for p in range(crops_t.shape[0]):
            fea = vit(crops_t[p:p + 1])
            score = F.cosine_similarity(ref_fea, fea, dim=1, eps=1e-8)
            scores.append(score.item())
            if (score.item() > similarity_score).any():
                min_idx = np.argmin(similarity_score)
                similarity_score[min_idx] = score.item()
                top_images[min_idx] = {"proposal": p}
        matching_score = [[0] for _ in range(len(top_images))]
        for top_idx in range(len(top_images)):
            p = top_images[top_idx]["proposal"]
            batch = {"image0": gray_ref, "image1": gray_crops[p:p + 1]}
            matcher(batch)
            confidences = batch["mconf"].cpu().numpy()
            matching_score[top_idx] = np.where(confidences > 0.9)[0].shape[0]
            top_images[top_idx].update(mkpts0=batch["mkpts0_f"].numpy(), mkpts1=batch["mkpts1_f"].numpy(), mconf=confidences)
        max_match_idx = int(np.argmax(matching_score))
    mine = driver_ref.locate_and_match(vit_sd, m_sd, my_cfg, ref_t, crops_t, gray_ref, gray_crops)
    assert np.array_equal(mine["slot_index"], [t["proposal"] for t in top_images])
    assert np.array_equal(mine["slot_scores"], similarity_score) and np.array_equal(mine["scores"], np.array(scores, np.float32))
    assert list(mine["matching_score"]) == matching_score and mine["best_slot"] == max_match_idx
    for s in range(3):
        for k in ("mkpts0", "mkpts1", "mconf"):
            assert np.array_equal(mine[k][s], top_images[s][k]), (s, k)
    print("driver_pair: scores", np.round(scores, 4), "slots", [t["proposal"] for t in top_images], similarity_score,
          "matching_score", matching_score, "best slot", max_match_idx)
    assert len(set(matching_score)) > 1 and max(matching_score) > 0
    fx = {"scores": np.array(scores, np.float32), "slot_scores": similarity_score,
          "slot_index": np.array([t["proposal"] for t in top_images]), "matching_score": np.array(matching_score),
          "best_slot": max_match_idx}
    for s in range(3):
        for k in ("mkpts0", "mkpts1", "mconf"):
            fx[f"{k}_{s}"] = top_images[s][k]
    np.savez(os.path.join(OUT, "driver_pair.npz"), **fx)


def gen_pair_lists():
    """The work list of BASELINE config 4: the pair ids of the reference's data/pairs/LINEMOD-test.json (13 objects x
    6 rotation bins, 5 796 pairs: eval_linemod_json.py:36-58 walks objects, then bins, then pairs).  Only the ids
    travel — "<dir>/<idx0>.png-<idx1>.png" entries become [idx0, idx1] integer pairs under their object directory;
    the images themselves are not in the reference tree (data/LM_dataset is a download) and pixels stay synthetic."""
    import json
    with open(os.path.join(REF, "data", "pairs", "LINEMOD-test.json")) as f:
        doc = json.load(f)
    objects = []
    for obj in doc:
        dirs = {os.path.dirname(p) for pairs in obj.values() for p in pairs}
        assert len(dirs) == 1
        bins = {}
        for key, pairs in obj.items():
            ids = []
            for p in pairs:
                a, b = os.path.basename(p).split("-")
                ids.append([int(a.split(".")[0]), int(b.split(".")[0])])
            bins[key] = ids
        objects.append({"dir": dirs.pop(), "bins": bins})
    n = sum(len(v) for o in objects for v in o["bins"].values())
    assert len(objects) == 13 and n == 5796
    with open(os.path.join(OUT, "linemod_pairs.json"), "w") as f:
        json.dump({"source": "data/pairs/LINEMOD-test.json", "n_pairs": n, "objects": objects}, f, separators=(",", ":"))
    print("linemod_pairs.json:", n, "pairs")


def gen_vit_archs():
    """BASELINE config 5, DINOv2 half: the ViT-B/14 and ViT-L/14 backbones of the reference
    (dinov2/dinov2/models/vision_transformer.py:319-342: 768-d / 12 heads / 12 blocks and 1024-d / 16 heads / 24 blocks,
    same eval config as load_dinov2_model) on one 224 x 224 image, seeded synthetic weights with the gamma = O(1)
    recipe; strided rows keep the fixtures small."""
    from dinov2.dinov2.models import vision_transformer as vits
    # vitl_476x630: the shape bench.py's config-5 leg runs (640 x 480 centre crop, 1 531 tokens), one image, every 32nd row
    for name, ctor, dim, depth, (H, W), stride in (("vitb_224", vits.vit_base, 768, 12, (224, 224), 8),
                                                   ("vitl_224", vits.vit_large, 1024, 24, (224, 224), 8),
                                                   ("vitl_476x630", vits.vit_large, 1024, 24, (476, 630), 32)):
        sd = synth.synthetic_state_dict(seed=0, dim=dim, depth=depth)
        model = ctor(patch_size=14, img_size=518, init_values=1e-5, ffn_layer="mlp", block_chunks=0, qkv_bias=True,
                     proj_bias=True, ffn_bias=True)
        model.load_state_dict(sd, strict=True)
        model.eval()
        x = synth.synthetic_images(1, H, W, seed=11)
        tap_blocks = (0, depth // 2, depth - 1)
        out, taps = run_ref_vit(model, x, tap_blocks)
        o_taps = {}
        mine = dinov2_ref.forward_features(sd, x, taps=o_taps)
        d = {k: maxdiff(out[k], mine[k]) for k in ("x_norm_clstoken", "x_norm_patchtokens", "x_prenorm")}
        for i in tap_blocks:
            d[f"blk{i}"] = maxdiff(taps[f"blk{i}"], o_taps[i]["x_out"])
        print(name, "oracle-vs-reference max abs diff:", {k: f"{v:.2e}" for k, v in d.items()})
        assert max(d.values()) <= 2e-5, d
        xn = torch.cat([out["x_norm_clstoken"][:, None], out["x_norm_patchtokens"]], 1)
        rows = torch.arange(0, xn.shape[1], stride)
        fx = {"weights_seed": 0, "arch": np.array([dim, depth, dim // 64]), "weights_digest": sd_digest(sd), "input_seed": 11,
              "shape": np.array([1, H, W]), "rows": rows.numpy(), "tap_blocks": np.array(tap_blocks),
              "input_digest": np.array([float(x.double().sum()), float(x.double().abs().sum())]),
              "x_norm": xn[:, rows].numpy(), "x_prenorm": out["x_prenorm"][:, rows].numpy(), "cls": model(x).detach().numpy()}
        for i in tap_blocks:
            fx[f"blk{i}"] = taps[f"blk{i}"][:, rows].numpy()
        np.savez(os.path.join(OUT, name + ".npz"), **fx)


SAM_CASES = {
    # name: dim, depth, heads, img, window, global blocks, batch, output stride
    "sam_hd80_256": (640, 4, 8, 256, 14, (1, 3), 2, 1),       # head_dim 80 (ViT-H's), grid 16 -> windows padded 16 -> 28
    "sam_hd64_224": (256, 2, 4, 224, 14, (1,), 1, 1),         # head_dim 64 (ViT-B/L's), grid 14 = one exact window
    "sam_vit_h_1024": (1280, 32, 16, 1024, 14, (7, 15, 23, 31), 1, 4),   # build_sam.py:13-21 at full size
    "sam_vit_b_1024": (768, 12, 12, 1024, 14, (2, 5, 8, 11), 1, 4),      # build_sam.py:36-45 at full size (head_dim 64)
}


def load_reference_sam_encoder():
    """The reference's `ImageEncoderViT`, loaded by file path under a synthetic package: the real package __init__ pulls
    torchvision / cv2, which this image lacks; image_encoder.py itself needs torch only."""
    import types
    base = os.path.join(REF, "segment_anything/segment_anything/modeling")
    pkg = types.ModuleType("ref_sam_modeling")
    pkg.__path__ = [base]
    sys.modules["ref_sam_modeling"] = pkg
    mods = {}
    for name in ("common", "image_encoder"):
        spec = importlib.util.spec_from_file_location(f"ref_sam_modeling.{name}", os.path.join(base, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = mod
        spec.loader.exec_module(mod)
        mods[name] = mod
    return mods["image_encoder"].ImageEncoderViT


def gen_sam_encoder(only=None):
    """BASELINE config 5, SAM half: ImageEncoderViT as build_sam.py:66-79 configures it, seeded
    synthetic weights, seeded input; the oracle restatement is checked against the reference's own module first."""
    from functools import partial
    from oracle import sam_encoder_ref
    Enc = load_reference_sam_encoder()
    for name, (dim, depth, heads, img, window, gidx, B, stride) in SAM_CASES.items():
        if only and name not in only:
            continue
        sd = synth.synthetic_sam_encoder_state_dict(seed=0, dim=dim, depth=depth, heads=heads, grid=img // 16, window=window,
                                                    global_idx=gidx)
        m = Enc(depth=depth, embed_dim=dim, img_size=img, mlp_ratio=4, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6),
                num_heads=heads, patch_size=16, qkv_bias=True, use_rel_pos=True, global_attn_indexes=list(gidx),
                window_size=window, out_chans=256)
        m.load_state_dict(sd, strict=True)
        m.eval()
        x = synth.synthetic_images(B, img, img, seed=11)
        tap_blocks = sorted({0, gidx[0], depth - 1})
        ref_taps = {}
        hooks = [m.blocks[i].register_forward_hook(lambda mod, a, o, i=i: ref_taps.__setitem__(i, o.detach())) for i in tap_blocks]
        with torch.no_grad():
            out = m(x)
        for h in hooks:
            h.remove()
        del m
        taps = {i: None for i in tap_blocks}
        with torch.no_grad():
            mine = sam_encoder_ref.forward(sd, x, heads, window, gidx, taps)
        d = {"out": maxdiff(out, mine)}
        for i in tap_blocks:
            d[f"blk{i}"] = maxdiff(ref_taps[i], taps[i])
        print(name, "oracle-vs-reference max abs diff:", {k: f"{v:.2e}" for k, v in d.items()},
              "| |out| max", float(out.abs().max()), "|x| max", float(ref_taps[depth - 1].abs().max()))
        assert max(d.values()) <= 2e-4, d
        fx = {"weights_seed": 0, "arch": np.array([dim, depth, heads, img, window]), "global_idx": np.array(gidx),
              "weights_digest": sd_digest(sd), "input_seed": 11, "batch": B, "stride": stride,
              "input_digest": np.array([float(x.double().sum()), float(x.double().abs().sum())]),
              "tap_blocks": np.array(tap_blocks), "out": out[:, :, ::stride, ::stride].numpy()}
        for i in tap_blocks:
            fx[f"blk{i}"] = ref_taps[i][:, ::max(2, stride), ::max(2, stride), ::2].numpy()
        np.savez(os.path.join(OUT, name + ".npz"), **fx)


def sd_digest(sd):
    return np.array([float(sd[k].double().sum()) for k in sorted(sd)], np.float64)


def maxdiff(a, b):
    return float((a - b).abs().max())


def run_ref_vit(model, x, tap_blocks=(0, 5, 11)):
    taps = {}
    hooks = []
    for i in tap_blocks:
        blk = model.blocks[i]
        hooks.append(blk.attn.register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(f"attn{i}", o.detach())))
        hooks.append(blk.mlp.register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(f"mlp{i}", o.detach())))
        hooks.append(blk.register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(f"blk{i}", o.detach())))
    with torch.no_grad():
        out = model(x, is_training=True)
        tokens = model.prepare_tokens_with_masks(x)
    for h in hooks:
        h.remove()
    taps["tokens"] = tokens
    return out, taps


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--only-pairs" in sys.argv:
        return gen_pair_lists()
    if "--only-archs" in sys.argv:
        return gen_vit_archs()
    if "--only-sam" in sys.argv:   # optionally followed by case names
        return gen_sam_encoder([a for a in sys.argv[sys.argv.index("--only-sam") + 1:] if not a.startswith("-")])
    if "--only-loftr-large" in sys.argv:
        return gen_loftr_large()
    if "--only-loftr" in sys.argv:
        return gen_loftr()
    if "--only-driver" in sys.argv:
        return gen_driver()
    sd = synth.synthetic_state_dict(seed=0)
    assert len(sd) == 175
    model = load_reference_vit(sd)
    cmatch = load_reference_coarse_matching()
    digest = sd_digest(sd)

    # ---------------- DINOv2 forward ----------------
    for name, (B, H, W, stride) in {"vit_196": (2, 196, 196, 3), "vit_224": (2, 224, 224, 4),
                                    "vit_476x630": (1, 476, 630, 12)}.items():
        x = synth.synthetic_images(B, H, W, seed=11)
        out, taps = run_ref_vit(model, x)
        o_taps = {}
        mine = dinov2_ref.forward_features(sd, x, taps=o_taps)
        d = {k: maxdiff(out[k], mine[k]) for k in ("x_norm_clstoken", "x_norm_patchtokens", "x_prenorm")}
        d["tokens"] = maxdiff(taps["tokens"], o_taps["tokens"])
        for i in (0, 5, 11):
            d[f"attn{i}"] = maxdiff(taps[f"attn{i}"], o_taps[i]["attn_branch"])
            d[f"mlp{i}"] = maxdiff(taps[f"mlp{i}"], o_taps[i]["mlp_branch"])
            d[f"blk{i}"] = maxdiff(taps[f"blk{i}"], o_taps[i]["x_out"])
        print(name, "oracle-vs-reference max abs diff:", {k: f"{v:.2e}" for k, v in d.items()})
        assert max(d.values()) <= 1e-5, d
        xn = torch.cat([out["x_norm_clstoken"][:, None], out["x_norm_patchtokens"]], 1)
        rows = torch.arange(0, xn.shape[1], stride)
        fx = {
            "weights_seed": 0, "weights_digest": digest, "input_seed": 11, "shape": np.array([B, H, W]),
            "input_digest": np.array([float(x.double().sum()), float(x.double().abs().sum())]),
            "rows": rows.numpy(),
            "x_norm": xn[:, rows].numpy(), "x_prenorm": out["x_prenorm"][:, rows].numpy(),
            "tokens": taps["tokens"][:, rows].numpy(),
            "cls": model(x).detach().numpy(),
        }
        for i in (0, 5, 11):
            fx[f"attn{i}"] = taps[f"attn{i}"][:, rows].numpy()
            fx[f"mlp{i}"] = taps[f"mlp{i}"][:, rows].numpy()
            fx[f"blk{i}"] = taps[f"blk{i}"][:, rows].numpy()
        np.savez(os.path.join(OUT, name + ".npz"), **fx)

    # ---------------- dense matcher on DINOv2 patch tokens ----------------
    for name, (H, W) in {"match_224": (224, 224), "match_476x630": (476, 630)}.items():
        n_pairs = 2 if H == 224 else 1
        i0, i1 = synth.synthetic_pairs(n_pairs, H, W, seed=5)
        with torch.no_grad():
            f0 = model(i0, is_training=True)["x_norm_patchtokens"]
            f1 = model(i1, is_training=True)["x_norm_patchtokens"]
        hw_c = (H // 14, W // 14)
        data = {"hw0_i": (H, W), "hw1_i": (H, W), "hw0_c": hw_c, "hw1_c": hw_c}
        with torch.no_grad():
            cmatch(f0, f1, data)
        mine = cm_ref.dense_match(f0, f1, hw_c, hw_c, (H, W))
        assert torch.equal(mine["conf_matrix"], data["conf_matrix"])
        for k in ("b_ids", "i_ids", "j_ids", "mconf", "mkpts0_c", "mkpts1_c", "m_bids"):
            assert torch.equal(mine[k], data[k]), k
        conf = data["conf_matrix"]
        print(name, "matches:", len(data["i_ids"]), "mconf range",
              float(data["mconf"].min()), float(data["mconf"].max()))
        assert len(data["i_ids"]) > 50
        np.savez(os.path.join(OUT, name + ".npz"),
                 weights_seed=0, weights_digest=digest, pair_seed=5, shape=np.array([n_pairs, H, W]),
                 feat0_digest=np.array([float(f0.double().sum())]), feat1_digest=np.array([float(f1.double().sum())]),
                 feat0_rows=f0[:, ::16].numpy(), feat1_rows=f1[:, ::16].numpy(),
                 b_ids=data["b_ids"].numpy(), i_ids=data["i_ids"].numpy(), j_ids=data["j_ids"].numpy(),
                 mconf=data["mconf"].numpy(), mkpts0_c=data["mkpts0_c"].numpy(), mkpts1_c=data["mkpts1_c"].numpy(),
                 conf_rowmax=conf.max(2)[0].numpy(), conf_rowarg=conf.max(2)[1].numpy(),
                 conf_colmax=conf.max(1)[0].numpy(), conf_colarg=conf.max(1)[1].numpy(),
                 conf_sum=np.array([float(conf.double().sum())]))

    # ---------------- dense matcher on LoFTR-shaped features (C=256) ----------------
    g = torch.Generator().manual_seed(3)
    n, hc, wc, C = 2, 16, 20, 256
    L = hc * wc
    f0 = torch.randn(n, L, C, generator=g) * 3.0
    perm = torch.stack([torch.randperm(L, generator=g) for _ in range(n)])
    f1 = torch.gather(f0, 1, perm[..., None].expand(-1, -1, C)) + 0.3 * torch.randn(n, L, C, generator=g)
    # exact duplicates to exercise tie handling (first index wins)
    f1[0, 77] = f1[0, 76]
    f0[1, 101] = f0[1, 100]
    data = {"hw0_i": (hc * 8, wc * 8), "hw1_i": (hc * 8, wc * 8), "hw0_c": (hc, wc), "hw1_c": (hc, wc)}
    with torch.no_grad():
        cmatch(f0, f1, data)
    mine = cm_ref.dense_match(f0, f1, (hc, wc), (hc, wc), (hc * 8, wc * 8))
    for k in ("b_ids", "i_ids", "j_ids", "mconf", "mkpts0_c", "mkpts1_c"):
        assert torch.equal(mine[k], data[k]), k
    print("match_loftr256 matches:", len(data["i_ids"]))
    assert len(data["i_ids"]) > 50
    np.savez(os.path.join(OUT, "match_loftr256.npz"), feat0=f0.numpy(), feat1=f1.numpy(),
             hw_c=np.array([hc, wc]), hw_i=np.array([hc * 8, wc * 8]),
             b_ids=data["b_ids"].numpy(), i_ids=data["i_ids"].numpy(), j_ids=data["j_ids"].numpy(),
             mconf=data["mconf"].numpy(), mkpts0_c=data["mkpts0_c"].numpy(), mkpts1_c=data["mkpts1_c"].numpy(),
             conf_matrix=data["conf_matrix"].numpy())

    # ---------------- CLS cosine + streaming top-3 (eval_linemod_json.py:71,94-101) ----------------
    g = torch.Generator().manual_seed(9)
    ref = torch.randn(1, 384, generator=g)
    fea = torch.randn(40, 384, generator=g) + 0.5 * ref
    fea[7] = fea[3]          # exact tie
    fea[20] = -ref           # negative score never enters
    scores = torch.cat([torch.nn.functional.cosine_similarity(ref, fea[i:i + 1], dim=1, eps=1e-8) for i in range(40)])
    assert maxdiff(scores, cm_ref.cls_cosine(ref, fea)) < 1e-6
    import numpy as _np
    slots, top = _np.array([0, 0, 0], _np.float32), [-1, -1, -1]
    for p in range(40):  # the reference's own loop, transcribed at the call site level
        if (scores[p].item() > slots).any():
            k = int(_np.argmin(slots)); slots[k] = scores[p].item(); top[k] = p
    s2, t2 = cm_ref.streaming_top3(scores.numpy())
    assert (s2 == slots).all() and list(t2) == top
    np.savez(os.path.join(OUT, "top3.npz"), ref=ref.numpy(), fea=fea.numpy(), scores=scores.numpy(),
             slot_scores=slots, slot_index=np.array(top))
    gen_loftr()
    gen_loftr_large()
    gen_driver()
    gen_pair_lists()
    gen_vit_archs()
    gen_sam_encoder()
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
