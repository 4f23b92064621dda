"""ctypes binding of libpope_hip.so (C ABI: include/pope_hip.h).

The product path has NO fallback: if the library is missing or fails to load the import of
``lib()`` raises, and every op raises on a non-zero status.
"""
import ctypes as C
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# POPE_LIB_PATH: dev override used by the lab scripts (stamped / ablated builds live outside the tree)
LIB_PATH = os.environ.get("POPE_LIB_PATH") or os.path.join(_CSRC, "libpope_hip.so")

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)
c_ll_p = C.POINTER(C.c_longlong)


class VitBlockWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "norm1_w", "norm1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ls1",
        "norm2_w", "norm2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "ls2", "qkv_wp", "fc1_wp", "fc2_wp", "proj_wp")]


class VitWeights(C.Structure):
    _fields_ = [("dim", C.c_int), ("depth", C.c_int), ("heads", C.c_int), ("patch", C.c_int),
                ("hidden", C.c_int), ("patch_w", C.c_void_p), ("norm_w", C.c_void_p),
                ("norm_b", C.c_void_p), ("blocks_host", C.POINTER(VitBlockWeights)), ("precision", C.c_int),
                ("patch_wp", C.c_void_p)]


class LoftrLayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("q_wp", "kv_wp", "merge_wp", "mlp0_wp", "mlp1_wp", "norm1_w", "norm1_b", "norm2_w",
                                           "norm2_b")]


class ResnetFpnWeights(C.Structure):
    _fields_ = [("w", C.c_void_p * 22), ("b", C.c_void_p * 22)]


class SamBlockWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("norm1_w", "norm1_b", "qkv_wp", "qkv_b", "proj_wp", "proj_b", "rel_h", "rel_w",
                                           "norm2_w", "norm2_b", "fc1_wp", "fc1_b", "fc2_wp", "fc2_b")] + [("global_attn", C.c_int)]


class SamEncoderWeights(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("img", "patch", "dim", "depth", "heads", "hidden", "out_chans", "window", "precision")] + [
        ("patch_wp", C.c_void_p), ("patch_b", C.c_void_p), ("pos", C.c_void_p), ("ones", C.c_void_p),
        ("blocks_host", C.POINTER(SamBlockWeights)), ("neck0_wp", C.c_void_p), ("neck1_w", C.c_void_p), ("neck1_b", C.c_void_p),
        ("neck2_wp", C.c_void_p), ("neck3_w", C.c_void_p), ("neck3_b", C.c_void_p), ("block_eps", C.c_float), ("neck_eps", C.c_float)]


# name -> (restype, argtypes); every symbol declared in include/pope_hip.h
PREC_F32_MFMA, PREC_F16X3, PREC_F16 = 0, 1, 2
PLANES_ACT_SCALE, PLANES_W_SCALE = 8.0, 256.0
PRECISIONS = {"f32": PREC_F32_MFMA, "f16x3": PREC_F16X3, "f16": PREC_F16}
c_uint_p = C.POINTER(C.c_uint)
F16_MAX = 65504.0
# bits of an f16x3 range-guard word (pope_hip.h POPE_RANGE_*)
RANGE_BITS = {1: "patch embed input", 2: "LayerNorm output", 4: "q/k/v", 8: "MLP hidden (GELU output)",
              16: "matcher features", 32: "op-level operand"}

PROTOTYPES = {
    "pope_abi_version": (C.c_int, []),
    "pope_error_string": (C.c_char_p, [C.c_int]),
    "pope_layernorm_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "pope_linear_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p] * 3),
    "pope_linear_prec_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p] * 2 + [C.c_int, C.c_void_p, C.c_void_p]),
    "pope_split_planes_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "pope_linear_planes_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p] * 4),
    "pope_layernorm_planes_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "pope_patch_embed_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]),
    "pope_attention_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pope_attention_prec_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pope_patch_embed_planes_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "pope_attention_planes_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pope_attention_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pope_attention_planes_diag_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, c_ll_p, C.c_void_p]),
    "pope_cls_cosine_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "pope_vit_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "pope_vit_forward_f32": (C.c_int, [C.POINTER(VitWeights), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_int, c_int_p, C.POINTER(C.c_void_p),
                                       C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "pope_vit_launch_count": (C.c_int, [C.c_int]),
    "pope_vit_forward_profiled_mask_f32": (C.c_int, [C.POINTER(VitWeights), C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                                     C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, c_int_p,
                                                     c_int_p, C.c_uint]),
    "pope_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "pope_event_destroy": (C.c_int, [C.c_void_p]),
    "pope_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_void_p, c_float_p]),
    "pope_dense_match_workspace_bytes": (C.c_size_t, [C.c_int] * 3),
    "pope_dense_match_f32": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong] + [C.c_int] * 8 + [C.c_float, C.c_int, C.c_float,
                                                                                  C.c_float] + [C.c_void_p] * 8
                             + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "pope_dense_match_workspace_bytes_prec": (C.c_size_t, [C.c_int] * 6),
    "pope_dense_match_prec_f32": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong] + [C.c_int] * 8
                                  + [C.c_float, C.c_int, C.c_float, C.c_float] + [C.c_void_p] * 8
                                  + [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]),
    "pope_loftr_layer_workspace_bytes": (C.c_size_t, [C.c_int] * 5),
    "pope_loftr_encoder_layer_f32": (C.c_int, [C.POINTER(LoftrLayerWeights), C.c_void_p, C.c_void_p] + [C.c_int] * 5
                                     + [C.c_float, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "pope_resnetfpn_workspace_bytes": (C.c_size_t, [C.c_int] * 3),
    "pope_resnetfpn_forward_f32": (C.c_int, [C.POINTER(ResnetFpnWeights), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "pope_fine_preprocess_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "pope_fine_preprocess_f32": (C.c_int, [C.c_void_p, c_ll_p, C.c_int, C.c_int, C.c_int, C.c_void_p, c_ll_p, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 3 + [C.c_int] * 3
                                 + [C.c_void_p] * 4 + [C.c_int, C.c_void_p] + [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "pope_fine_match_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "pope_sam_encoder_workspace_bytes": (C.c_size_t, [C.POINTER(SamEncoderWeights), C.c_int]),
    "pope_sam_encoder_forward_f32": (C.c_int, [C.POINTER(SamEncoderWeights), C.c_void_p, C.c_int, C.c_void_p, C.c_int, c_int_p,
                                               C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "pope_preprocess_u8_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3
                               + [C.c_int] * 7 + [c_float_p, c_float_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pope_crop_normalize_u8_f32": (C.c_int, [C.c_void_p] + [C.c_int] * 7 + [c_float_p, c_float_p, C.c_void_p, C.c_void_p]),
    "pope_gray_u8_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pope_crop_warp_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                    C.c_void_p]),
    "pope_estimate_pose_workspace_bytes": (C.c_size_t, [C.c_int, C.c_longlong]),
    "pope_estimate_pose_f64": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_longlong, C.c_double, C.c_double, C.c_int, C.c_ulonglong]
                               + [C.c_void_p] * 5 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "pope_five_point_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pope_streaming_top3_host": (C.c_int, [c_float_p, C.c_int, c_float_p, c_ll_p]),
}

_lib = None


def build(force=False):
    """Compile libpope_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", _CSRC, "clean"], check=True, capture_output=True)
    res = subprocess.run(["make", "-C", _CSRC, "-j4"], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libpope_hip.so failed:\n" + res.stdout + res.stderr)
    return LIB_PATH


def code_object_archs(path=None):
    """GPU architectures of the code objects bundled in the library, read from the offload-bundle entry ids in the
    file itself (no extraction: `llvm-objdump --offloading` drops one file per code object next to its input)."""
    import re
    with open(path or LIB_PATH, "rb") as f:
        data = f.read()
    return sorted({m.decode() for m in re.findall(rb"hipv4-amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", data)})


def lib():
    """Load the shared library (once).  Raises if it is missing — there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(pope_amd has no fallback path)")
        # torch FIRST: it ships its own HIP runtime (torch/lib/libamdhip64.so).  If libpope_hip.so is the first to
        # pull in a libamdhip64 (the system one), the process ends up with two runtimes and every launch of ours on a
        # torch stream fails — `build()` followed by `smoke()` in one process did exactly that
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        if handle.pope_abi_version() != 9:
            raise RuntimeError("libpope_hip.so ABI version mismatch")
        _lib = handle
    return _lib


class PopeHipError(RuntimeError):
    pass


class PopeRangeError(PopeHipError):
    """An f16x3 operand left the f16 range (|activation| * 8 or |weight| * 256 >= 65504): results of that call are not
    valid; raised when the model's `on_overflow` policy is "raise" (the default policy re-runs on the fp32 MFMA)."""


def describe_range_bits(bits):
    return ", ".join(name for bit, name in RANGE_BITS.items() if bits & bit) or "none"


def check(status, what):
    if status != 0:
        raise PopeHipError(f"{what}: {lib().pope_error_string(status).decode()} (status {status})")


def to_planes(t, scale):
    """torch restatement of the planes layout (pope_hip.h): [rows, cols] fp32 -> f16 [rows, cols/32, 2, 32]
    with t*scale = hi + lo, hi = RNE f16."""
    ts = t.detach().float() * scale
    hi = ts.half()
    lo = (ts - hi.float()).half()
    r, c = t.shape
    import torch
    return torch.stack([hi.view(r, c // 32, 32), lo.view(r, c // 32, 32)], dim=2).contiguous()


def from_planes(pl, scale):
    """inverse of to_planes (fp32)."""
    r = pl.shape[0]
    return (pl[:, :, 0].float() + pl[:, :, 1].float()).reshape(r, -1) / scale


def params_key(tensors):
    """Cache key of everything derived from `tensors` (weight planes, folded BatchNorm, gathered tables): storage address AND
    torch's in-place modification counter of every source tensor, so `p.copy_()`, an optimizer / EMA step or `w[0, 0] = x`
    invalidates the derived data like a reallocation does."""
    return tuple((t.data_ptr(), t._version) for t in tensors)


def param_slots(module, buffers=False):
    """Where a module tree keeps its parameters (and buffers): [(the owning submodule's `_parameters` / `_buffers` dict, name)].
    Cached by the models as the source list of their derived data; the TENSORS are looked up through it on every call
    (`slots_key`), so a replaced Parameter object — `load_state_dict(..., assign=True)`, `m.weight = nn.Parameter(..)`,
    `.to()` — is seen like an in-place edit, at the price of one dict lookup per tensor instead of a walk over the module
    tree (0.35 ms for ViT-S/14).  Adding or removing SUBMODULES after the first call is not tracked."""
    slots = []
    for m in module.modules():
        slots += [(m._parameters, n) for n in m._parameters]
        if buffers:
            slots += [(m._buffers, n) for n in m._buffers]
    return slots


def slots_key(slots):
    """`params_key` of the tensors currently sitting in `slots` (None entries, e.g. `bias=None`, keyed as such)."""
    return tuple((t.data_ptr(), t._version) if t is not None else None for t in (d.get(n) for d, n in slots))


def ptr(t):
    """Device pointer of a contiguous fp32/int tensor (or None)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError("pope_amd expects contiguous tensors")
    return C.c_void_p(t.data_ptr())


def stream_of(device):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def on_device_of(t):
    """Context manager: make `t`'s GPU the current HIP device around a C-ABI call.  torch hands out the NULL handle
    for every device's default stream, so the stream alone does not name the device (the reference keeps its matcher
    on cuda:1 while cuda:0 is current, pope_model_api.py:181-184); non-default streams are additionally resolved on
    the C side (capi.hip:StreamDevice)."""
    import torch
    return torch.cuda.device(t.device)


def require_cuda(t, what):
    if not t.is_cuda:
        raise PopeHipError(
            f"{what}: tensor is on {t.device}; the pope_amd product path runs on the MI355X HIP kernels only "
            "(no CPU fallback) — move the model/inputs to 'cuda'")
