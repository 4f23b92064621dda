"""Dev: time pope_attention_f16 (attention_f16.hip) alone at the ViT-L/14 shape of BASELINE config 5 (21 images x 16 heads x 1531
tokens).  Prints ms per launch and the fraction of the f16 peak."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pope_amd import _lib  # noqa: E402

B, N, heads = 21, 1531, 16
lib = _lib.lib()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(B * N, 3 * heads * 64, generator=g, device=dev) * 0.7).half()
out = torch.empty(B * N, heads * 64, dtype=torch.float16, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
call = lambda: lib.pope_attention_f16(C.c_void_p(qkv.data_ptr()), C.c_void_p(out.data_ptr()), B, N, heads, st)
for _ in range(3):
    assert call() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(3):
    e0.record()
    for _ in range(10):
        call()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10)
fl = 4.0 * B * heads * N * N * 64
print(f"{os.path.basename(_lib.LIB_PATH):28s} {best:7.4f} ms  {fl / best / 1e9:7.1f} TF/s = {fl / best / 1e9 / 2500:.3f} of the f16 peak", flush=True)
