import os, sys, torch, time
sys.path.insert(0, "/root/repo")
from pope_amd import ops
dev = torch.device("cuda:0")
B, N, H = 64, 1531, 6
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B, N, 3 * H * 64, device=dev, generator=g)
for _ in range(3): o = ops.attention(qkv, H, precision="f16x3")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): o = ops.attention(qkv, H, precision="f16x3")
e1.record(); torch.cuda.synchronize()
print(os.environ.get("POPE_ATTN_NO_PIPE"), "attention ms", e0.elapsed_time(e1) / 20, float(o.double().abs().sum()))
