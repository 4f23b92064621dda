"""Small profiling driver: a few DINOv2 forwards (64 x 476x630) + one 32-pair dense match.
Run under rocprofv3 (--kernel-trace --stats, or --pmc ... in a separate pass)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pope_amd import synth  # noqa: E402
from pope_amd.dinov2_utils import load_dinov2_model  # noqa: E402
from pope_amd.matcher import dense_match  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
model = load_dinov2_model(state_dict=synth.synthetic_state_dict(seed=0)).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B, 3, 476, 630, generator=g, device=dev)
for _ in range(reps):
    out = model(x, is_training=True)
f = out["x_norm_patchtokens"]
n = min(B // 2, 32)
m = dense_match(f[:n], f[n:2 * n], (34, 45), (34, 45), (476, 630))
torch.cuda.synchronize()
print("ok", float(out["x_norm_clstoken"].sum()), int(m["counts"].sum()))
