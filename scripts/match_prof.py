"""Dev: the dense matcher alone at the bench shape (128 pairs of 1530 x 1530 x 384), for `rocprofv3 --kernel-trace --stats`.
usage: python scripts/match_prof.py [n_pairs] [reps] [want_conf]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pope_amd.matcher import dense_match  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
want_conf = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
hw, C = (34, 45), 384
L = hw[0] * hw[1]
f0 = 3.0 * torch.randn(n, L, C, generator=g, device=dev)
perm = torch.stack([torch.randperm(L, generator=g, device=dev) for _ in range(n)])
f1 = torch.gather(f0, 1, perm[..., None].expand(-1, -1, C)) + 0.75 * torch.randn(n, L, C, generator=g, device=dev)
out = dense_match(f0, f1, hw, hw, (476, 630), want_conf=want_conf)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    out = dense_match(f0, f1, hw, hw, (476, 630), want_conf=want_conf)
torch.cuda.synchronize()
print(f"dense_match n={n} want_conf={want_conf}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per call (incl. allocation + count readback), "
      f"{len(out['i_ids'])} matches")
