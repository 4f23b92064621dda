"""Dev: forward throughput of the DINOv2 ViT-S / B / L variants (vision_transformer.py:306-343) through the same HIP
kernels, 476x630 inputs (1531 tokens), synthetic weights.  ViT-S uses the LayerNorm-fused residual GEMM (width 384);
B and L go through the planes GEMM + the stand-alone LayerNorm kernel."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pope_amd import synth
from pope_amd.dinov2 import vit_small, vit_base, vit_large
dev = torch.device("cuda:0")
H, W, B = 476, 630, 32
x = synth.synthetic_images(B, H, W, seed=3).to(dev)
N = 1 + (H // 14) * (W // 14)
for name, ctor, dim, depth in (("ViT-S/14", vit_small, 384, 12), ("ViT-B/14", vit_base, 768, 12), ("ViT-L/14", vit_large, 1024, 24)):
    m = ctor(patch_size=14, img_size=518, init_values=1e-5, ffn_layer="mlp", block_chunks=0).eval()
    m.load_state_dict(synth.synthetic_state_dict(seed=0, dim=dim, depth=depth), strict=True)
    m = m.to(dev)
    with torch.no_grad():
        for _ in range(2): m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): out = m(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    gf = depth * (2 * N * dim * 12 * dim + 4 * N * N * dim) / 1e9 + 2 * N * 588 * dim / 1e9   # per image
    print(f"{name}: {B / dt:7.1f} images/s at {H}x{W} ({dt * 1e3:.1f} ms per {B} images, {gf:.1f} GF/image -> {B * gf / dt / 1e3:.0f} TFLOP/s), "
          f"overflow events {m.overflow_events}")
