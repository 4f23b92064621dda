"""Dev: where the batched pose kernel's time goes.  `python scripts/pose_time.py`"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pope_amd import synth
from pope_amd.pose import estimate_pose_batch, five_point
dev = torch.device("cuda:0")


def timed(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, out


for B, n, outl, thr in [(128, 1130, 0.0, 0.5), (128, 1130, 0.3, 0.5), (128, 100, 0.3, 0.5), (128, 30, 0.3, 0.5), (1, 1130, 0.3, 0.5), (512, 1130, 0.3, 0.5)]:
    scenes = [synth.synthetic_pose_scene(n, s, outlier=outl, noise=0.2) for s in range(min(B, 16))]
    scenes = (scenes * (B // len(scenes) + 1))[:B]
    k0 = torch.from_numpy(np.concatenate([s[0] for s in scenes])).to(dev)
    k1 = torch.from_numpy(np.concatenate([s[1] for s in scenes])).to(dev)
    counts = torch.full((B,), n, dtype=torch.int32)
    K0, K1 = scenes[0][2], scenes[0][3]
    for mi in (64, 256, 1000):
        ms, out = timed(lambda: estimate_pose_batch(k0, k1, counts, K0, K1, thr, 0.99, max_iters=mi))
        info = out["info"].cpu().numpy()
        print(f"B={B} N={n} outliers={outl} max_iters={mi}: {ms:.3f} ms, hypotheses mean {info[:, 2].mean():.0f}, rounds mean {info[:, 3].mean():.2f}, "
              f"inliers mean {info[:, 1].mean():.0f}")
g = torch.Generator().manual_seed(0)
for S in (64, 256, 4096, 32768):
    x0, x1 = torch.randn(S, 5, 2, generator=g, dtype=torch.float64).to(dev) * 0.2, torch.randn(S, 5, 2, generator=g, dtype=torch.float64).to(dev) * 0.2
    ms, _ = timed(lambda: five_point(x0, x1))
    print(f"five_point kernel, {S} problems: {ms:.3f} ms")
