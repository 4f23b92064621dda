"""Context measurement (not part of the product): what the vendor f16 GEMM (hipBLASLt through torch.matmul)
reaches on the ViT-S/14 linear-layer shapes at 64 images, for comparison with the executed-MFMA rate of the
f16x3 kernels (3 MFMA FLOPs per algorithmic FLOP)."""
import torch

dev = torch.device("cuda:0")
M = 64 * 1531
for name, n, k in (("qkv", 1152, 384), ("proj", 384, 384), ("fc1", 1536, 384), ("fc2", 384, 1536)):
    for dt in (torch.float16, torch.bfloat16, torch.float32):
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
        print(f"{name:5s} {str(dt):15s} {ms:7.4f} ms  {2.0 * M * n * k / ms / 1e9:8.1f} TFLOP/s", flush=True)
