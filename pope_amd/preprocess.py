"""Batched GPU preprocessing: `set_torch_image` for P proposal crops at once and the gray / 255
conversion of the matcher's inputs, on HIP kernels (pope_amd/csrc/preprocess.hip) that are bit-identical to the host
path of the reference (segment_anything/segment_anything/dinov2_utils.py:55-78: PIL + torchvision transforms).

The resize is Pillow's 8-bit bilinear resample; its per-output-pixel windows and fixed-point weights depend on the
(input, output) sizes only and are built here on the host in double precision exactly as Pillow's `precompute_coeffs`
builds them (cached per size pair), so the kernels only do the integer arithmetic.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import check, on_device_of, require_cuda, stream_of
from .synth import IMAGENET_MEAN, IMAGENET_STD

PRECISION_BITS = 32 - 8 - 2   # Pillow, 8 bits per channel


def resize_tables(in_size, out_size):
    """Windows and weights of one pass of Pillow's bilinear resample from `in_size` to `out_size` samples:
    (start[out], count[out], weights[out, ksize]) as int32 arrays, weights with 22 fractional bits."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = filterscale                      # triangle filter: support 1, stretched when downscaling (antialias)
    ksize = int(math.ceil(support)) * 2 + 1
    start = np.zeros(out_size, np.int32)
    count = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    inv = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), in_size)
        w = [max(0.0, 1.0 - abs((x + lo - center + 0.5) * inv)) for x in range(hi - lo)]
        total = 0.0
        for v in w:                            # sequential double-precision sum, as in the C loop
            total += v
        if total != 0.0:
            w = [v / total for v in w]
        start[xx], count[xx] = lo, hi - lo
        kk[xx, :hi - lo] = [int(0.5 + v * (1 << PRECISION_BITS)) for v in w]
    return start, count, kk


_tables = {}


def _device_tables(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    if key not in _tables:
        s, c, k = resize_tables(in_size, out_size)
        _tables[key] = (torch.from_numpy(s).to(device), torch.from_numpy(c).to(device), torch.from_numpy(k).to(device),
                        int(k.shape[1]), s, c)
    return _tables[key]


@torch.no_grad()
def preprocess_batch(images, resize, crop=None, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """[P, H, W, 3] uint8 (CUDA tensor) -> [P, 3, h, w] fp32: Resize(resize) -> CenterCrop(crop) -> /255 -> Normalize,
    bit-identical to the PIL / torchvision host path for every image of the batch."""
    require_cuda(images, "preprocess_batch")
    if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[3] != 3:
        raise TypeError("preprocess_batch expects a uint8 [P, H, W, 3] tensor")
    images = images.contiguous()
    P, Hin, Win, _ = images.shape
    OH, OW = int(resize[0]), int(resize[1])
    if crop is None:
        ch, cw, top, left = OH, OW, 0, 0
    else:   # torchvision CenterCrop
        ch, cw = int(crop[0]), int(crop[1])
        top, left = int(round((OH - ch) / 2.0)), int(round((OW - cw) / 2.0))
    dev = images.device
    hs, hc, hk, kh, _, _ = _device_tables(Win, OW, dev)
    vs, vc, vk, kv, vs_h, vc_h = _device_tables(Hin, OH, dev)
    row0 = int(vs_h[top])
    nrows = int(vs_h[top + ch - 1] + vc_h[top + ch - 1]) - row0
    out = torch.empty(P, 3, ch, cw, dtype=torch.float32, device=dev)
    scratch = torch.empty(P * nrows * cw * 3, dtype=torch.uint8, device=dev)
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    p = lambda t: C.c_void_p(t.data_ptr())
    with on_device_of(images):
        check(_lib.lib().pope_preprocess_u8_f32(p(images), P, Hin, Win, p(hs), p(hc), p(hk), kh, p(vs), p(vc), p(vk), kv, top, left,
                                                ch, cw, row0, nrows, m, s, p(out), p(scratch), scratch.numel(), stream_of(dev)),
              "pope_preprocess_u8_f32")
    return out


def set_torch_images(images, center_crop=False, device="cuda"):
    """Batched `set_torch_image` (dinov2_utils.py:55-78): a list / array of HWC uint8 images of one size (or a uint8
    [P, H, W, 3] tensor) -> [P, 3, 196, 196] (center_crop) or [P, 3, 224, 224] on the GPU."""
    if not isinstance(images, torch.Tensor):
        images = torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(i) for i in images])))
    images = images.to(device)
    return preprocess_batch(images, (256, 256), (196, 196)) if center_crop else preprocess_batch(images, (224, 224), None)


@torch.no_grad()
def crop_normalize(images, crop_hw, mean=IMAGENET_MEAN, std=IMAGENET_STD, out=None):
    """[P, H, W, 3] uint8 (CUDA) -> [P, 3, ch, cw] fp32: the centre crop + ToTensor + Normalize of the dense pair path (640 x 480
    frames -> 476 x 630, no resize), one kernel, bit-equal to torchvision's fp32 arithmetic.  `out` lets a pipeline reuse its
    input buffer."""
    require_cuda(images, "crop_normalize")
    if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[3] != 3:
        raise TypeError("crop_normalize expects a uint8 [P, H, W, 3] tensor")
    images = images.contiguous()
    P, H, W, _ = images.shape
    ch, cw = crop_hw
    top, left = (H - ch) // 2, (W - cw) // 2
    if out is None:
        out = torch.empty(P, 3, ch, cw, dtype=torch.float32, device=images.device)
    m, s = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    with on_device_of(images):
        check(_lib.lib().pope_crop_normalize_u8_f32(C.c_void_p(images.data_ptr()), P, H, W, top, left, ch, cw, m, s,
                                                    C.c_void_p(out.data_ptr()), stream_of(images.device)), "pope_crop_normalize_u8_f32")
    return out


@torch.no_grad()
def gray_batch(images_bgr):
    """[P, H, W, 3] uint8 BGR (CUDA) -> [P, 1, H, W] fp32 in [0, 1]: cv2.cvtColor(BGR2GRAY) (OpenCV's 8-bit fixed-point
    weights) / 255., the matcher's input (eval_linemod_json.py:103-111).  OpenCV is not in this image: the integer
    formula is OpenCV's documented one, unpinned against the library."""
    require_cuda(images_bgr, "gray_batch")
    if images_bgr.dtype != torch.uint8 or images_bgr.dim() != 4 or images_bgr.shape[3] != 3:
        raise TypeError("gray_batch expects a uint8 [P, H, W, 3] tensor")
    images_bgr = images_bgr.contiguous()
    P, H, W, _ = images_bgr.shape
    out = torch.empty(P, 1, H, W, dtype=torch.float32, device=images_bgr.device)
    with on_device_of(images_bgr):
        check(_lib.lib().pope_gray_u8_f32(C.c_void_p(images_bgr.data_ptr()), P, H, W, C.c_void_p(out.data_ptr()),
                                          stream_of(images_bgr.device)), "pope_gray_u8_f32")
    return out
