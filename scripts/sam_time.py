"""Time the SAM image encoder (BASELINE config 5) on one GPU: `python scripts/sam_time.py [arch] [batch] [iters] [f16x3|f16]`
(arch: vit_h | vit_l | vit_b).  Prints ms / image and algorithmic TFLOP/s (2 x MACs of the reference's own ops)."""
import os
import sys
from functools import partial

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pope_amd import synth  # noqa: E402
from pope_amd.sam_encoder import ImageEncoderViT  # noqa: E402

ARCHS = {"vit_h": (1280, 32, 16, (7, 15, 23, 31)), "vit_l": (1024, 24, 16, (5, 11, 17, 23)), "vit_b": (768, 12, 12, (2, 5, 8, 11))}


def flops(dim, depth, heads, gidx, grid=64, window=14, oc=256):
    n = grid * grid
    hd = dim // heads
    lin = n * (3 * 16 * 16 * dim + depth * (3 * dim * dim + dim * dim + 8 * dim * dim)) + n * (dim * oc + 9 * oc * oc)
    nw = -(-grid // window)
    attn = 0
    for i in range(depth):
        if i in gidx:
            attn += heads * n * n * 2 * hd + heads * n * 2 * grid * hd
        else:
            t = window * window
            attn += nw * nw * heads * (t * t * 2 * hd + t * 2 * window * hd)
    return 2.0 * (lin + attn)


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "vit_h"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    precision = sys.argv[4] if len(sys.argv) > 4 else "f16x3"
    dim, depth, heads, gidx = ARCHS[arch]
    m = ImageEncoderViT(depth=depth, embed_dim=dim, img_size=1024, mlp_ratio=4, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6),
                        num_heads=heads, patch_size=16, qkv_bias=True, use_rel_pos=True, global_attn_indexes=list(gidx),
                        window_size=14, out_chans=256)
    m.load_state_dict(synth.synthetic_sam_encoder_state_dict(seed=0, dim=dim, depth=depth, heads=heads, global_idx=gidx), strict=True)
    m = m.eval().cuda()
    m.precision = precision
    x = synth.synthetic_images(batch, 1024, 1024, seed=1).cuda()
    m(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = m(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    f = flops(dim, depth, heads, gidx)
    print(f"{arch} [{precision}] batch {batch}: {ms / batch:.2f} ms / image, {batch * 1e3 / ms:.1f} images/s, "
          f"{f * batch / ms / 1e9:.1f} TFLOP/s algorithmic ({f / 1e12:.2f} TF / image), finite={bool(torch.isfinite(out).all())}")


if __name__ == "__main__":
    main()
