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
