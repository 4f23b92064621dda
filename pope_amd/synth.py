"""Seeded synthetic inputs and weights for the hot path.

Real checkpoints (weights/dinov2_vits14.pth, weights/matcher.pth) cannot be
fetched offline, so benchmarks, smoke and parity tests use
seeded synthetic weights in the reference's exact state-dict layout
(dinov2/dinov2/models/vision_transformer.py:45-163 -> 175 keys for ViT-S/14).
"""
import math

import torch

IMAGENET_MEAN = (0.485, 0.456, 0.406)  # segment_anything/segment_anything/dinov2_utils.py:67
IMAGENET_STD = (0.229, 0.224, 0.225)


def synthetic_state_dict(seed=0, dim=384, depth=12, patch=14, grid=37, gamma=1.0,
                         wscale=None):
    """Seeded synthetic weights in the reference checkpoint layout (175 keys for
    ViT-S/14).  Unlike the reference initialisers (gamma=1e-5,
    trunc_normal(0.02)), branch outputs here are O(1) so that attention / MLP
    kernels are actually exercised."""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * std

    sd = {}
    sd["cls_token"] = rn(1, 1, dim, std=0.5)
    sd["pos_embed"] = rn(1, grid * grid + 1, dim, std=0.2)
    sd["mask_token"] = torch.zeros(1, dim)
    k_pe = 3 * patch * patch
    sd["patch_embed.proj.weight"] = rn(dim, 3, patch, patch, std=1.0 / math.sqrt(k_pe))
    sd["patch_embed.proj.bias"] = rn(dim, std=0.1)
    ws = wscale if wscale is not None else 1.0
    for i in range(depth):
        p = f"blocks.{i}."
        sd[p + "norm1.weight"] = 1.0 + rn(dim, std=0.1)
        sd[p + "norm1.bias"] = rn(dim, std=0.05)
        sd[p + "attn.qkv.weight"] = rn(3 * dim, dim, std=ws * 1.5 / math.sqrt(dim))
        sd[p + "attn.qkv.bias"] = rn(3 * dim, std=0.1)
        sd[p + "attn.proj.weight"] = rn(dim, dim, std=ws / math.sqrt(dim))
        sd[p + "attn.proj.bias"] = rn(dim, std=0.05)
        sd[p + "ls1.gamma"] = gamma * (0.3 + 0.1 * rn(dim))
        sd[p + "norm2.weight"] = 1.0 + rn(dim, std=0.1)
        sd[p + "norm2.bias"] = rn(dim, std=0.05)
        sd[p + "mlp.fc1.weight"] = rn(4 * dim, dim, std=ws / math.sqrt(dim))
        sd[p + "mlp.fc1.bias"] = rn(4 * dim, std=0.1)
        sd[p + "mlp.fc2.weight"] = rn(dim, 4 * dim, std=ws / math.sqrt(4 * dim))
        sd[p + "mlp.fc2.bias"] = rn(dim, std=0.05)
        sd[p + "ls2.gamma"] = gamma * (0.3 + 0.1 * rn(dim))
    sd["norm.weight"] = 1.0 + rn(dim, std=0.1)
    sd["norm.bias"] = rn(dim, std=0.05)
    return sd


def synthetic_sam_encoder_state_dict(seed=0, dim=1280, depth=32, heads=16, grid=64, window=14, global_idx=(7, 15, 23, 31),
                                     patch=16, out_chans=256):
    """Seeded synthetic weights in the layout of SAM's `ImageEncoderViT` state dict
    (segment_anything/segment_anything/modeling/image_encoder.py:53-105, build_sam.py:66-79; ViT-H defaults).  Branch
    outputs and the relative-position terms are O(1) so windows, padding keys and the bias all matter."""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * std

    hd = dim // heads
    sd = {"pos_embed": rn(1, grid, grid, dim, std=0.2)}
    k_pe = 3 * patch * patch
    sd["patch_embed.proj.weight"] = rn(dim, 3, patch, patch, std=1.0 / math.sqrt(k_pe))
    sd["patch_embed.proj.bias"] = rn(dim, std=0.1)
    for i in range(depth):
        p = f"blocks.{i}."
        size = grid if i in global_idx else window
        sd[p + "norm1.weight"] = 1.0 + rn(dim, std=0.1)
        sd[p + "norm1.bias"] = rn(dim, std=0.05)
        sd[p + "attn.qkv.weight"] = rn(3 * dim, dim, std=1.5 / math.sqrt(dim))
        sd[p + "attn.qkv.bias"] = rn(3 * dim, std=0.1)
        sd[p + "attn.proj.weight"] = rn(dim, dim, std=0.3 / math.sqrt(dim))
        sd[p + "attn.proj.bias"] = rn(dim, std=0.02)
        sd[p + "attn.rel_pos_h"] = rn(2 * size - 1, hd, std=0.15)
        sd[p + "attn.rel_pos_w"] = rn(2 * size - 1, hd, std=0.15)
        sd[p + "norm2.weight"] = 1.0 + rn(dim, std=0.1)
        sd[p + "norm2.bias"] = rn(dim, std=0.05)
        sd[p + "mlp.lin1.weight"] = rn(4 * dim, dim, std=1.0 / math.sqrt(dim))
        sd[p + "mlp.lin1.bias"] = rn(4 * dim, std=0.1)
        sd[p + "mlp.lin2.weight"] = rn(dim, 4 * dim, std=0.3 / math.sqrt(4 * dim))
        sd[p + "mlp.lin2.bias"] = rn(dim, std=0.02)
    sd["neck.0.weight"] = rn(out_chans, dim, 1, 1, std=1.0 / math.sqrt(dim))
    sd["neck.1.weight"] = 1.0 + rn(out_chans, std=0.1)
    sd["neck.1.bias"] = rn(out_chans, std=0.05)
    sd["neck.2.weight"] = rn(out_chans, out_chans, 3, 3, std=1.0 / math.sqrt(9 * out_chans))
    sd["neck.3.weight"] = 1.0 + rn(out_chans, std=0.1)
    sd["neck.3.bias"] = rn(out_chans, std=0.05)
    return sd


def synthetic_images(batch, h=476, w=630, seed=0, device="cpu"):
    """Uniform[0,1) 640x480 frames, centre-cropped to (h, w) and normalised with
    the ImageNet statistics of set_torch_image."""
    g = torch.Generator().manual_seed(seed)
    full_h, full_w = max(480, h), max(640, w)
    img = torch.rand(batch, 3, full_h, full_w, generator=g, dtype=torch.float32)
    top, left = (full_h - h) // 2, (full_w - w) // 2
    img = img[:, :, top:top + h, left:left + w]
    mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    return ((img - mean) / std).contiguous().to(device)


def synthetic_pairs(n_pairs, h=476, w=630, seed=0, shift=(14, 28), noise=0.1, device="cpu"):
    """Pair = (img, roll(img, shift) + N(0, noise^2)) so the dense matcher emits
    ~ (H/14-4-1)*(W/14-4-2) matches per pair instead of none."""
    img0 = synthetic_images(n_pairs, h, w, seed, "cpu")
    g = torch.Generator().manual_seed(seed + 7919)
    img1 = torch.roll(img0, shifts=shift, dims=(2, 3))
    if noise > 0:
        img1 = img1 + noise * torch.randn(img1.shape, generator=g, dtype=torch.float32)
    return img0.to(device), img1.contiguous().to(device)


def _hash_uniform(ids, n_per, salt, device):
    """[len(ids), n_per] floats in [0, 1): a counter-based integer hash of (id, element index, salt).  Integer and
    exact-float arithmetic only, so a pair's pixels depend on its id alone — not on the batch it rides in, the rank
    that owns it or the device that evaluates the expression."""
    i = torch.arange(n_per, device=device, dtype=torch.int64)[None, :]
    x = (ids.to(device=device, dtype=torch.int64)[:, None] * 0x9E3779B1 + i * 0x85EBCA6B + salt * 0xC2B2AE35) & 0xFFFFFFFF
    x = ((x ^ (x >> 16)) * 0x7FEB352D) & 0xFFFFFFFF
    x = ((x ^ (x >> 15)) * 0x846CA68B) & 0xFFFFFFFF
    x = x ^ (x >> 16)
    return (x >> 8).to(torch.float32) / 16777216.0


def pairs_by_id(ids, h=476, w=630, shift=(14, 28), noise=0.1, device="cpu"):
    """Synthetic pixels for the pairs of a work list (BASELINE config 4: only the pair ids of the LINEMOD list
    travel): frame = hash-uniform 640x480, centre crop (h, w), ImageNet normalisation; the second image
    is the rolled first plus noise * (Irwin-Hall sum of four hash-uniforms, unit variance)."""
    ids = torch.as_tensor(ids, dtype=torch.int64)
    n, full_h, full_w = len(ids), max(480, h), max(640, w)
    img = _hash_uniform(ids, 3 * full_h * full_w, 1, device).view(n, 3, full_h, full_w)
    top, left = (full_h - h) // 2, (full_w - w) // 2
    mean = torch.tensor(IMAGENET_MEAN, device=device).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, device=device).view(1, 3, 1, 1)
    img0 = ((img[:, :, top:top + h, left:left + w] - mean) / std).contiguous()
    del img
    img1 = torch.roll(img0, shifts=shift, dims=(2, 3))
    if noise > 0:
        z = sum(_hash_uniform(ids, 3 * h * w, 2 + k, device) for k in range(4)).view(n, 3, h, w)
        img1 = img1 + (noise * math.sqrt(3.0)) * (z - 2.0)
    return img0, img1.contiguous()


def synthetic_matcher_state_dict(seed=0, cfg=None):
    """Seeded synthetic LoFTR `Matcher` weights in the reference checkpoint layout (211 keys,
    src/matcher/matcher.py:18-27; weights/matcher.pth is not available offline).  Fan-in scaled normal
    filters; BatchNorm statistics and affines are randomised (the reference's fresh-module defaults are
    the identity, which would not exercise BN folding)."""
    if cfg is None:
        from .matcher import default_cfg as cfg
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * std

    def ru(*shape, lo=0.5, hi=1.5):
        return lo + (hi - lo) * torch.rand(*shape, generator=g, dtype=torch.float32)

    sd = {}

    def conv(name, cout, cin, k, gain=1.0):
        sd[name + ".weight"] = rn(cout, cin, k, k, std=math.sqrt(gain / (cin * k * k)))

    def bn(name, c):
        sd[name + ".weight"] = ru(c)
        sd[name + ".bias"] = rn(c, std=0.1)
        sd[name + ".running_mean"] = rn(c, std=0.1)
        sd[name + ".running_var"] = ru(c)
        sd[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    d0 = cfg["resnetfpn"]["initial_dim"]
    d1, d2, d3 = cfg["resnetfpn"]["block_dims"]
    conv("backbone.conv1", d0, 1, 7, gain=8.0)
    bn("backbone.bn1", d0)
    cin = d0
    for li, (dim, stride) in enumerate(((d1, 1), (d2, 2), (d3, 2)), start=1):
        for bi in range(2):
            p = f"backbone.layer{li}.{bi}"
            conv(p + ".conv1", dim, cin if bi == 0 else dim, 3)
            conv(p + ".conv2", dim, dim, 3)
            bn(p + ".bn1", dim)
            bn(p + ".bn2", dim)
            if bi == 0 and stride != 1:
                conv(p + ".downsample.0", dim, cin, 1)
                bn(p + ".downsample.1", dim)
        cin = dim
    conv("backbone.layer3_outconv", d3, d3, 1, gain=2.0)
    conv("backbone.layer2_outconv", d3, d2, 1)
    conv("backbone.layer2_outconv2.0", d3, d3, 3)
    bn("backbone.layer2_outconv2.1", d3)
    conv("backbone.layer2_outconv2.3", d2, d3, 3)
    conv("backbone.layer1_outconv", d2, d1, 1)
    conv("backbone.layer1_outconv2.0", d2, d2, 3)
    bn("backbone.layer1_outconv2.1", d2)
    conv("backbone.layer1_outconv2.3", d1, d2, 3)

    def transformer(prefix, c):
        # small LayerNorm gains keep the residual stream O(1): with unit gains eight layers of O(1) messages
        # make <f0,f1>/(C*0.1) so large that the dual softmax saturates to exact 0/1 confidences
        d, n_layers = c["d_model"], len(c["layer_names"])
        for i in range(n_layers):
            p = f"{prefix}.layers.{i}."
            for nm in ("q_proj", "k_proj", "v_proj", "merge"):
                sd[p + nm + ".weight"] = rn(d, d, std=1.0 / math.sqrt(d))
            sd[p + "mlp.0.weight"] = rn(2 * d, 2 * d, std=1.0 / math.sqrt(2 * d))
            sd[p + "mlp.2.weight"] = rn(d, 2 * d, std=math.sqrt(2.0 / (2 * d)))
            for nm in ("norm1", "norm2"):
                sd[p + nm + ".weight"] = ru(d, lo=0.3, hi=0.5)
                sd[p + nm + ".bias"] = rn(d, std=0.02)

    transformer("loftr_coarse", cfg["coarse"])
    dc, df = cfg["coarse"]["d_model"], cfg["fine"]["d_model"]
    sd["fine_preprocess.down_proj.weight"] = rn(df, dc, std=1.0 / math.sqrt(dc))
    sd["fine_preprocess.down_proj.bias"] = rn(df, std=0.05)
    sd["fine_preprocess.merge_feat.weight"] = rn(df, 2 * df, std=1.0 / math.sqrt(2 * df))
    sd["fine_preprocess.merge_feat.bias"] = rn(df, std=0.05)
    transformer("loftr_fine", cfg["fine"])
    return sd


def peaked_matcher_state_dict(pre_outconv, seed=0, gain=2.0, cfg=None):
    """Synthetic LoFTR weights under which the `Matcher` behaves like a trained one on related image pairs: hundreds of
    confident matches per 256 x 256 pair (the interior maximum of (32 - 4)^2 cells minus the shifted border), instead of the
    handful the plain random weights give.  Why those give so few: the 1/8-resolution features are f_i = v + d_i with a
    common vector v (the mean of the post-ReLU activations through the bias-free 1x1 output convolution, |v|^2 = 1 400) that
    swamps the cell-specific part (|d|^2 = 180): <v, d_j> acts as a per-column bias in the dual softmax and one column
    attracts every row.  A trained network has no such component.  Here it is projected out: W' = gain * W (I - m m^T / |m|^2)
    with m the mean of the activations in front of `backbone.layer3_outconv` on a calibration batch of the same image
    statistics (`synthetic_gray_pairs`), which leaves features whose correlation picks the true cell for 3 of 4 cells before
    the transformer and whose dual-softmax confidence clears 0.9 on 95 % of the interior cells.
    `pre_outconv(state_dict, images[n, 1, H, W]) -> activations [n, 256, H / 8, W / 8]` runs the CNN up to that convolution
    (pass a state dict whose `backbone.layer3_outconv.weight` is the identity to any implementation of the backbone: the HIP
    module — `hip_pre_outconv` below — on the GPU box, the CPU checker in the CPU tests); this module computes nothing itself.
    `pre_outconv` may also be the calibration mean m itself ([256] tensor, e.g. from a fixture).  The returned dict carries it
    under "_calibration_mean" — pop it before `load_state_dict`."""
    sd = synthetic_matcher_state_dict(seed, cfg)
    w = sd["backbone.layer3_outconv.weight"]
    c = w.shape[0]
    if torch.is_tensor(pre_outconv):   # the calibration mean itself (a fixture pins the weights bit for bit across hosts)
        m = pre_outconv.detach().double().cpu().reshape(c)
    else:
        ident = dict(sd)
        ident["backbone.layer3_outconv.weight"] = torch.eye(c).reshape(c, c, 1, 1)
        cal = synthetic_gray_pairs(4, 256, 256, seed=1000 + seed)[0]
        x3 = pre_outconv(ident, cal).detach().float().cpu()
        m = x3.permute(0, 2, 3, 1).reshape(-1, c).double().mean(0)
    sd["_calibration_mean"] = m.clone()    # not a checkpoint key: callers that need it pop it (load_state_dict would reject it)
    mh = m / m.norm()
    proj = torch.eye(c, dtype=torch.float64) - torch.outer(mh, mh)
    sd["backbone.layer3_outconv.weight"] = (gain * (w.reshape(c, c).double() @ proj)).float().reshape(c, c, 1, 1).contiguous()
    return sd


def hip_pre_outconv(device):
    """`pre_outconv` of `peaked_matcher_state_dict` on the HIP backbone (pope_amd/loftr.py:ResNetFPN_8_2)."""
    def run(sd, images):
        from .loftr import build_backbone
        from .matcher import default_cfg
        bb = build_backbone(default_cfg).eval()
        bb.load_state_dict({k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}, strict=True)
        return bb.to(device)(images.to(device))[0]
    return run


def synthetic_gray_pairs(n_pairs, h=256, w=256, seed=0, shift=(8, 16), noise=0.02):
    """Grayscale [n,1,h,w] pairs in [0,1] for the LoFTR `Matcher` (the drivers feed 256x256 crops,
    eval_linemod_json.py:103-111): image1 = roll(image0, shift) + noise, h and w multiples of 8."""
    g = torch.Generator().manual_seed(seed)
    # low-pass random texture: LoFTR's CNN sees structure rather than white noise
    base = torch.rand(n_pairs, 1, h // 4, w // 4, generator=g, dtype=torch.float32)
    img0 = torch.nn.functional.interpolate(base, size=(h, w), mode="bilinear", align_corners=False)
    img0 = (0.7 * img0 + 0.3 * torch.rand(n_pairs, 1, h, w, generator=g, dtype=torch.float32)).clamp_(0, 1)
    img1 = torch.roll(img0, shifts=shift, dims=(2, 3))
    if noise > 0:
        img1 = (img1 + noise * torch.randn(img1.shape, generator=g, dtype=torch.float32)).clamp_(0, 1)
    return img0.contiguous(), img1.contiguous()


def synthetic_driver_case(n_proposals=8, seed=31):
    """Synthetic stand-in for one iteration of the drivers' pair loop (eval_linemod_json.py:52-127) without
    SAM / cv2: a reference crop, P proposal crops for DINOv2 (196x196, the drivers' centre-crop size) and the
    matching gray images for LoFTR (256x256).  Proposals 2, 5 and 6 show the reference object (shifted,
    increasingly noisy), proposal 3 duplicates proposal 2's DINOv2 crop exactly (a score tie), the others
    are unrelated texture."""
    g = torch.Generator().manual_seed(seed)
    ref_tensor = synthetic_images(1, 196, 196, seed=seed)
    crop_tensors = synthetic_images(n_proposals, 196, 196, seed=seed + 1)
    gray_ref = synthetic_gray_pairs(1, 256, 256, seed=seed)[0]
    gray_crops = synthetic_gray_pairs(n_proposals, 256, 256, seed=seed + 2)[0]
    for p, (noise_rgb, shift, noise_gray) in {2: (0.3, (8, 16), 0.01), 5: (0.6, (16, 8), 0.03),
                                              6: (1.0, (24, 24), 0.06)}.items():
        crop_tensors[p] = ref_tensor[0] + noise_rgb * torch.randn(ref_tensor[0].shape, generator=g)
        gray_crops[p] = (torch.roll(gray_ref[0], shifts=shift, dims=(1, 2))
                         + noise_gray * torch.randn(gray_ref[0].shape, generator=g)).clamp_(0, 1)
    crop_tensors[3] = crop_tensors[2]
    return ref_tensor, crop_tensors, gray_ref, gray_crops


def synthetic_pose_scene(n, seed, outlier=0.3, noise=0.0, K0=None, K1=None):
    """A planted two-view scene for the pose solver: n 3-D points in front of both cameras, a random
    rotation of 5-30 degrees and a unit-direction translation of length 0.5, projected with K0 / K1 (defaults: the LINEMOD
    camera and a 256x256 crop camera, the two intrinsics eval_linemod_json.py:160 passes), Gaussian pixel noise, and a
    fraction `outlier` of the second image's points replaced by uniform clutter.
    Returns numpy (kpts0 [n,2] f32, kpts1 [n,2] f32, K0, K1, R [3,3], t [3], planted_inlier [n] bool)."""
    import numpy as np
    g = np.random.default_rng(seed)
    axis = g.normal(size=3)
    axis /= np.linalg.norm(axis)
    ang = np.deg2rad(g.uniform(5, 30))
    Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
    t = g.normal(size=3)
    t *= 0.5 / np.linalg.norm(t)
    X = np.stack([g.uniform(-1, 1, n), g.uniform(-1, 1, n), g.uniform(3, 6, n)], 1)
    K0 = np.array([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1.0]]) if K0 is None else np.asarray(K0, np.float64)
    K1 = np.array([[600.0, 0, 128.0], [0, 610.0, 120.0], [0, 0, 1.0]]) if K1 is None else np.asarray(K1, np.float64)
    p0 = (X / X[:, 2:]) @ K0.T
    X1 = X @ R.T + t
    p1 = (X1 / X1[:, 2:]) @ K1.T
    k0 = p0[:, :2] + g.normal(size=(n, 2)) * noise
    k1 = p1[:, :2] + g.normal(size=(n, 2)) * noise
    n_out = int(outlier * n)
    bad = g.permutation(n)[:n_out]
    k1[bad] = np.stack([g.uniform(0, 256, n_out), g.uniform(0, 256, n_out)], 1)
    inl = np.ones(n, bool)
    inl[bad] = False
    return k0.astype(np.float32), k1.astype(np.float32), K0, K1, R, t, inl


def synthetic_frame_case(n_proposals=8, seed=31, frame_hw=(480, 640)):
    """uint8 stand-in for one query of the drivers' loop starting at the raw frame (eval_linemod_json.py:62-90), without SAM:
    a 256x256 BGR reference crop, a BGR frame and P proposal boxes (x, y, w, h).  Proposals 1 and 4 are 160x160 boxes whose
    30 %-expanded window (256x256, scale 1: an integer crop) shows the reference shifted by a few pixels plus noise, so the
    DINOv2 vote and the LoFTR matcher (synthetic weights) find them; the other boxes of assorted sizes — two leave the frame —
    sit on unrelated texture.  Returns numpy (ref_bgr [256,256,3], frame_bgr [H,W,3], bboxes_xywh [P,4], K0 [3,3], K1 [3,3])."""
    import numpy as np
    H, W = frame_hw
    g = torch.Generator().manual_seed(seed)

    def texture(h, w):
        base = torch.rand(1, 3, h // 4 + 1, w // 4 + 1, generator=g)
        t = torch.nn.functional.interpolate(base, size=(h, w), mode="bilinear", align_corners=False)[0]
        return (0.7 * t + 0.3 * torch.rand(3, h, w, generator=g)).clamp_(0, 1)

    ref = texture(256, 256)
    frame = texture(H, W)
    boxes = np.zeros((n_proposals, 4), np.int64)
    planted = {1: ((96, 88), (8, 16), 0.01), 4: ((372, 60), (16, 8), 0.03)}      # (x, y) of the 160-box, shift, noise
    rng = np.random.default_rng(seed)
    for p in range(n_proposals):
        if p in planted:
            (x, y), shift, noise = planted[p]
            win = (torch.roll(ref, shifts=shift, dims=(1, 2)) + noise * torch.randn(ref.shape, generator=g)).clamp_(0, 1)
            frame[:, y - 48:y + 208, x - 48:x + 208] = win
            boxes[p] = (x, y, 160, 160)
        else:
            w, h = int(rng.integers(40, 220)), int(rng.integers(40, 200))
            boxes[p] = (int(rng.integers(-10, W - w + 30)), int(rng.integers(-10, H - h + 30)), w, h)
    if n_proposals > 6:
        boxes[6] = (W - 90, H - 70, 120, 100)       # leaves the frame at the bottom right
    to_u8 = lambda t: (t.permute(1, 2, 0) * 255).round().to(torch.uint8).numpy()  # noqa: E731
    K0 = np.array([[572.4114, 0, 128.0], [0, 573.57043, 128.0], [0, 0, 1.0]])
    K1 = np.array([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1.0]])
    return to_u8(ref), to_u8(frame), boxes, K0, K1
