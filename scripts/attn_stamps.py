"""Dev: per-iteration cycle breakdown of the pipelined attention kernel (library built with -DATTN_STAMPS)."""
import ctypes as C, os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pope_amd import ops, _lib
dev = torch.device("cuda:0")
B, N, H = 64, 1531, 6
qkv = torch.randn(B, N, 3 * H * 64, device=dev)
lib = C.CDLL(_lib.LIB_PATH)
names = ["phase1 (QK next + softmax)", "phase2 (PV + splits + kv store)", "wait barrier"]
def report(tag):
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 256)()
    assert lib.pope_lab_attn_stamps(buf) == 0
    rows = [[buf[t * 8 + s] for s in range(4)] for t in range(22)]
    print(tag, {n: round(statistics.mean(rows[t][i + 1] - rows[t][i] for t in range(2, 21))) for i, n in enumerate(names)},
          "iter", round(statistics.mean(rows[t + 1][0] - rows[t][0] for t in range(2, 20))))
for _ in range(3): o = ops.attention(qkv, H, precision="f16x3")
report("fp32 in :")
pin = _lib.to_planes(qkv.reshape(B * N, -1).cpu(), 8.0).to(dev)
pout = torch.zeros(B * N, H * 2, 2, 32, dtype=torch.float16, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3): lib.pope_attention_planes_f32(C.c_void_p(pin.data_ptr()), C.c_void_p(pout.data_ptr()), B, N, H, st)
report("planes in:")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): lib.pope_attention_planes_f32(C.c_void_p(pin.data_ptr()), C.c_void_p(pout.data_ptr()), B, N, H, st)
e1.record(); torch.cuda.synchronize()
print("planes-in attention ms", e0.elapsed_time(e1) / 20)
