"""Dev: one ResNet-FPN call on 6 images of 256x256 (the drivers' three pairs) for a rocprofv3 --kernel-trace run."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pope_amd import synth
from pope_amd.matcher import Matcher, default_cfg
dev = torch.device("cuda:0")
m = Matcher(default_cfg).eval(); m.load_state_dict(synth.synthetic_matcher_state_dict(0)); m = m.to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
i0, i1 = (t.to(dev) for t in synth.synthetic_gray_pairs(n, 256, 256, seed=21))
x = torch.cat([i0, i1], 0)
for _ in range(3): m.backbone(x)
torch.cuda.synchronize()
torch.cuda.nvtx.range_push("measured") if hasattr(torch.cuda, "nvtx") else None
m.backbone(x)
torch.cuda.synchronize()
