"""Dev: launch time of the f16x3 attention kernels at the bench shape (fp32 in/out and planes in/out)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pope_amd import ops, _lib
dev = torch.device("cuda:0")
B, N, H = 64, 1531, 6
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B, N, 3 * H * 64, device=dev, generator=g)
if len(sys.argv) > 1:  # "ramp <step>": scores climbing by <step> per 64-key tile (drives the exact pass of the lazy softmax)
    step = float(sys.argv[2])
    t = qkv.view(B, N, 3, H, 64)
    t.mul_(0.3)
    u = torch.randn(H, 64, device=dev, generator=g)
    u = u / u.norm(dim=-1, keepdim=True) * 8.0
    t[:, :, 0] += u
    t[:, :, 1] += u * (torch.arange(N, device=dev, dtype=torch.float32) / 64.0 * (step / 8.0))[None, :, None, None]
lib = C.CDLL(_lib.LIB_PATH)
pin = _lib.to_planes(qkv.reshape(B * N, -1).cpu(), 8.0).to(dev)
pout = torch.zeros(B * N, H * 2, 2, 32, dtype=torch.float16, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
a = timed(lambda: ops.attention(qkv, H, precision="f16x3"))
b = timed(lambda: lib.pope_attention_planes_f32(C.c_void_p(pin.data_ptr()), C.c_void_p(pout.data_ptr()), B, N, H, st))
print("attention ms: fp32-io %.4f  planes-io %.4f  checksum %.6g" % (a, b, float(pout.float().abs().sum())))
