"""Dev: where a tile of the LayerNorm-fused residual GEMM spends its time (library built with -DRL_STAMPS:
scripts/ab_rowln.sh).  Stamps of block 0 in 10 ns ticks, per kind of launch (proj K = 384, FC2 K = 1536, patch embed)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pope_amd import synth, _lib
from pope_amd.dinov2_utils import load_dinov2_model
dev = torch.device("cuda:0")
model = load_dinov2_model(state_dict=synth.synthetic_state_dict(seed=0)).to(dev)
x = synth.synthetic_images(64, 476, 630, seed=1).to(dev)
for _ in range(3): model(x)
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)  # the handle the model already uses
buf = (C.c_ulonglong * 192)()
assert lib.pope_lab_rowln_stamps(buf) == 0
names = ["K loop", "residual + x (phase 1)", "row means (2)", "store x + 2nd moment (3)", "LN + planes stores (4)", "seam: next K-step load + first item"]
for kind, label in enumerate(["proj (K=384)", "FC2 (K=1536)", "patch embed (K=608)"]):
    v = [buf[kind * 64 + i] for i in range(64)]
    for t in range(3):
        s = v[8 * t: 8 * t + 8]
        nxt = v[8 * (t + 1)]
        if not s[0] or not s[5]: continue
        d = [(s[i + 1] - s[i]) / 100.0 for i in range(5)] + [((nxt - s[5]) / 100.0) if nxt else float("nan")]
        print(f"{label} tile {t}: " + ", ".join(f"{n} {x:.1f} us" for n, x in zip(names, d)))
