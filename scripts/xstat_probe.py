"""Dev probe (not product): the X-stationary planes GEMM (gemm_xstat.hip) against the 128 x 128 tile kernel on the ViT-S/14
QKV / FC1 shapes at 64 images — bit equality of the output planes and time per launch.  Usage:
    python scripts/xstat_probe.py [lib.so ...]      # default: the in-tree library; every further library is timed too
The FIRST library given is the reference for the bit comparison (scripts/xstat_ab.sh product "")."""
import ctypes as C
import sys

import torch

from pope_amd import _lib


def load(path):
    h = C.CDLL(path)
    for name, (res, args) in _lib.PROTOTYPES.items():
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, args
    return h


def main():
    paths = sys.argv[1:] or [_lib.LIB_PATH]
    libs = [(p, load(p)) for p in paths]
    dev = torch.device("cuda:0")
    M, K = 64 * 1531, 384
    g = torch.Generator().manual_seed(3)
    a = torch.randn(M, K, generator=g) * 1.3
    ap = _lib.to_planes(a, _lib.PLANES_ACT_SCALE).to(dev)
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for name, N, epi in (("qkv", 1152, 0), ("fc1", 1536, 1)):
        w = torch.randn(N, K, generator=g) * K ** -0.5
        b = torch.randn(N, generator=g)
        wp = _lib.to_planes(w, _lib.PLANES_W_SCALE).to(dev)
        bd = b.to(dev)
        ref = None
        for path, lib in libs:
            out = torch.zeros(M, N // 32, 2, 32, dtype=torch.float16, device=dev)
            flag = torch.zeros(1, dtype=torch.int32, device=dev)
            call = lambda: lib.pope_linear_planes_f32(P(ap), P(wp), P(bd), None, P(out), M, N, K, epi, None, None, P(flag), st)
            for _ in range(3):
                assert call() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for rep in range(3):
                e0.record()
                for _ in range(20):
                    call()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20)
            same = ""
            if ref is None:
                ref = out.clone()
                # sanity of the reference itself against fp64 on a few rows
                rows = torch.arange(0, M, 4099)
                lin = torch.nn.functional.linear(a[rows].double(), w.double(), b.double())
                want = torch.nn.functional.gelu(lin) if epi else lin
                got = _lib.from_planes(out[rows.to(dev)].cpu(), _lib.PLANES_ACT_SCALE).double()
                same = f"max err vs fp64 {float((got - want).abs().max()):.2e}"
            else:
                neq = int((out.view(torch.int16) != ref.view(torch.int16)).sum())
                same = f"bit-equal to first: {neq == 0} ({neq} halves differ)"
            print(f"{name} N={N} {path.split('/')[-1]:28s} {best:7.4f} ms  {2.0 * M * N * K / best / 1e9:7.1f} TF/s (x3 executed "
                  f"{6.0 * M * N * K / best / 1e9:7.1f})  flag={int(flag.item())}  {same}", flush=True)


if __name__ == "__main__":
    main()
