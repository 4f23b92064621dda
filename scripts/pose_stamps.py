"""Dev: phase stamps of the pose kernel's workgroup 0 (library built with -DPOSE_STAMPS, POPE_LIB_PATH points at it)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pope_amd import synth
from pope_amd.pose import estimate_pose_batch
dev = torch.device("cuda:0")
for n in (30, 1130):
    scenes = [synth.synthetic_pose_scene(n, s, outlier=0.3, noise=0.2) for s in range(8)]
    k0 = torch.from_numpy(np.concatenate([s[0] for s in scenes])).to(dev)
    k1 = torch.from_numpy(np.concatenate([s[1] for s in scenes])).to(dev)
    counts = torch.full((8,), n, dtype=torch.int32)
    for _ in range(3):
        out = estimate_pose_batch(k0, k1, counts, scenes[0][2], scenes[0][3], 0.5, 0.99)
    st = out["E"][0].cpu().numpy().ravel()
    print(f"N={n}: stamps (us) start 0 | five_point {st[1]:.0f} | scored {st[2]:.0f} | ransac done + mask {st[3]:.0f} | cheirality {st[4]:.0f} | end {st[6]:.0f}")
