"""In-situ per-kernel timing of the product path with HIP events (used by bench.py's roofline leg).

`KernelProfiler` owns a pool of hipEvents; while `model.profiler` is set, every forward goes through
`pope_vit_forward_profiled_mask_f32`, which brackets the kernel launches of the selected kinds with events
on the launch stream (all kinds by default; `kinds=("attention",)` times one kernel and leaves every other
launch back to back as in the untimed path).  After a synchronise, `summary()` turns the event pairs into
per-kernel-kind durations.
"""
import ctypes as C
from collections import defaultdict

from . import _lib
from ._lib import check

KIND_NAMES = {0: "patch_embed_gemm", 1: "layernorm", 2: "gemm_qkv", 3: "attention", 4: "gemm_proj",
              5: "gemm_fc1_gelu", 6: "gemm_fc2", 7: "tap_copy"}


KIND_IDS = {v: k for k, v in KIND_NAMES.items()}


class KernelProfiler:
    def __init__(self, depth, max_forwards, kinds=None):
        lib = _lib.lib()
        self.mask = 0xFFFFFFFF if kinds is None else sum(1 << KIND_IDS[k] for k in kinds)
        self.per_forward = 2 * lib.pope_vit_launch_count(depth) + 1  # worst case: start + close per launch
        self.max_forwards = max_forwards
        n = self.per_forward * max_forwards
        self.events = (C.c_void_p * n)()
        for i in range(n):
            ev = C.c_void_p()
            check(lib.pope_event_create(C.byref(ev)), "pope_event_create")
            self.events[i] = ev
        self.kinds = (C.c_int * n)()
        self.launches = []  # (first_event_index, n_launches)

    def next_slot(self):
        """(events pointer, capacity, kinds pointer) for one more forward, or None when the pool is spent."""
        i = len(self.launches)
        if i >= self.max_forwards:
            return None
        off = i * self.per_forward
        ev = C.cast(C.byref(self.events, off * C.sizeof(C.c_void_p)), C.POINTER(C.c_void_p))
        kd = C.cast(C.byref(self.kinds, off * C.sizeof(C.c_int)), C.POINTER(C.c_int))
        return off, ev, self.per_forward, kd

    def commit(self, off, n_launches):
        self.launches.append((off, n_launches))

    def summary(self):
        """{kind name: {"launches", "total_ms", "avg_ms"}} — call after the stream has been synchronised."""
        lib = _lib.lib()
        acc = defaultdict(list)
        ms = C.c_float()
        for off, n in self.launches:
            for j in range(n):
                if self.kinds[off + j] < 0:  # close-only event
                    continue
                check(lib.pope_event_elapsed_ms(self.events[off + j], self.events[off + j + 1], C.byref(ms)),
                      "pope_event_elapsed_ms")
                acc[self.kinds[off + j]].append(ms.value)
        return {KIND_NAMES.get(k, str(k)): {"launches": len(v), "total_ms": sum(v), "avg_ms": sum(v) / len(v)}
                for k, v in sorted(acc.items())}

    def reset(self):
        self.launches = []

    def close(self):
        lib = _lib.lib()
        for ev in self.events:
            if ev:
                lib.pope_event_destroy(ev)
        self.events = ()
