"""Dev: per-iteration cycle breakdown of the pipelined attention kernel (library built with -DATTN_STAMPS)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pope_amd import ops, _lib
dev = torch.device("cuda:0")
qkv = torch.randn(64, 1531, 3 * 6 * 64, device=dev)
for _ in range(3): o = ops.attention(qkv, 6, precision="f16x3")
torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
lib = C.CDLL(os.path.join(os.path.dirname(_lib.__file__), "csrc", "libpope_hip.so"))
assert lib.pope_lab_attn_stamps(buf) == 0
rows = [[buf[t * 8 + s] for s in range(4)] for t in range(22)]
names = ["phase1 (QK next + softmax)", "phase2 (PV + splits + kv store)", "wait barrier"]
for t in (2, 8, 14, 20):
    r = rows[t]
    print("iter", t, {n: r[i + 1] - r[i] for i, n in enumerate(names)}, "total", rows[t + 1][0] - r[0])
import statistics
print("mean", {n: statistics.mean(rows[t][i + 1] - rows[t][i] for t in range(2, 21)) for i, n in enumerate(names)},
      "iter", statistics.mean(rows[t + 1][0] - rows[t][0] for t in range(2, 20)))
