"""Dev: wall time of the drop-in LoFTR Matcher (HIP stages: conv.hip, loftr.hip, match.hip, fine.hip) on the drivers' shape:
three 256x256 pairs per query (eval_linemod_json.py:108-125), and of the whole driver step with 8 proposals."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pope_amd import synth
from pope_amd.matcher import Matcher, default_cfg
from pope_amd.dinov2_utils import load_dinov2_model
from pope_amd.driver import locate_and_match
dev = torch.device("cuda:0")
m = Matcher(default_cfg).eval(); m.load_state_dict(synth.synthetic_matcher_state_dict(0)); m = m.to(dev)
vit = load_dinov2_model(state_dict=synth.synthetic_state_dict(0)).to(dev)
i0, i1 = (t.to(dev) for t in synth.synthetic_gray_pairs(3, 256, 256, seed=21))
for graph in (False, True):   # eager launches against the HIP-graph replay of the front end
    m.use_graph = graph
    for n in (3, 24):
        a, b = i0.repeat(n // 3, 1, 1, 1), i1.repeat(n // 3, 1, 1, 1)
        for _ in range(3): m({"image0": a, "image1": b})
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): d = {"image0": a, "image1": b}; m(d)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"Matcher batch {n} x 256x256 ({'graph' if graph else 'eager'}): {dt*1e3:.2f} ms per call = {n/dt:.0f} LoFTR pairs/s, "
              f"{len(d['b_ids'])} matches")
for n in (6, 48):   # the CNN alone (conv.hip: every convolution on the f16x3 planes GEMM)
    x = torch.cat([i0, i1], 0).repeat(n // 6, 1, 1, 1)
    for _ in range(3): m.backbone(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m.backbone(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"backbone {n} x 256x256: {dt*1e3:.2f} ms = {n * 63.36 / dt / 1e3:.1f} TFLOP/s")
case = [t.to(dev) for t in synth.synthetic_driver_case()]
for graph in (False, True):
    m.use_graph = graph
    for _ in range(3): locate_and_match(vit, m, *case)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): out = locate_and_match(vit, m, *case)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"driver step (8 proposals: batched DINOv2 vote + one 3-pair Matcher call; {'graph' if graph else 'eager'}): "
          f"{dt*1e3:.2f} ms -> {1/dt:.1f} queries/s")
