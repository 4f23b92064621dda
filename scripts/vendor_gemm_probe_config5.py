"""Context measurement (not part of the product): what the vendor f16 GEMM (hipBLASLt through torch.matmul) reaches on the
Linear-layer shapes of BASELINE config 5 — SAM ViT-H at 1024^2 (4 images: M = 16 384) and DINOv2 ViT-L/14 at 476 x 630 (16
images: M = 24 496) — next to which the `precision="f16"` GEMMs of this repository are read (profiles/r04/)."""
import torch

dev = torch.device("cuda:0")
for model, M, dim, hidden in (("sam_vit_h", 4 * 4096, 1280, 5120), ("dinov2_vit_l14", 16 * 1531, 1024, 4096)):
    for name, n, k in (("qkv", 3 * dim, dim), ("proj", dim, dim), ("fc1", hidden, dim), ("fc2", dim, hidden)):
        for dt in (torch.float16, torch.bfloat16):
            a = torch.randn(M, k, device=dev, dtype=dt)
            w = torch.randn(n, k, device=dev, dtype=dt)
            for _ in range(3):
                c = a @ w.t()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                c = a @ w.t()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            print(f"{model:15s} {name:5s} M={M} N={n} K={k} {str(dt):15s} {ms:7.4f} ms  {2.0 * M * n * k / ms / 1e9:8.1f} TFLOP/s", flush=True)
