"""Dev: one warm LoFTR Matcher call on three 256x256 pairs, for rocprofv3 --kernel-trace --stats."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pope_amd import synth
from pope_amd.matcher import Matcher, default_cfg
dev = torch.device("cuda:0")
m = Matcher(default_cfg).eval(); m.load_state_dict(synth.synthetic_matcher_state_dict(0)); m = m.to(dev)
i0, i1 = (t.to(dev) for t in synth.synthetic_gray_pairs(3, 256, 256, seed=21))
for _ in range(6):
    d = {"image0": i0, "image1": i1}; m(d)
torch.cuda.synchronize()
print("ok", len(d["b_ids"]))
