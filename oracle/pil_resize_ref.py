"""TEST INFRASTRUCTURE (oracle): restatement of Pillow's 8-bit bilinear `Image.resize` — what torchvision's
`transforms.Resize` runs for the PIL images of the reference's `set_torch_image`
(segment_anything/segment_anything/dinov2_utils.py:55-78; torchvision itself is not in this image).

Pillow is a third-party dependency (the image ships 12.2.0); the algorithm restated here is its published
`ImagingResample` for 8-bit-per-channel images (src/libImaging/Resample.c): a separable two-pass convolution,
horizontal then vertical, with per-output-pixel windows [xmin, xmin + n), triangle weights normalised to 1 in double
precision, converted to fixed point with 22 fractional bits, accumulated in int32 from 2^21 and shifted back, each pass
rounding to uint8.  Pinned bit-for-bit against Pillow itself (tests/test_preprocess_cpu.py).

The product's GPU preprocessing (pope_amd/preprocess.py + csrc/preprocess.hip) builds the same tables on its own and is
tested against Pillow and against this restatement."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def resize_tables(in_size, out_size):
    """Per output coordinate: window start, window length and fixed-point weights [out_size, ksize] (int32)."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale          # bilinear filter support = 1
    ksize = int(math.ceil(support)) * 2 + 1
    start = np.zeros(out_size, np.int32)
    count = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        n = xmax - xmin
        w = np.zeros(n, np.float64)
        for x in range(n):
            t = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - t if t < 1.0 else 0.0
        ww = 0.0
        for x in range(n):   # sequential double sum, as the C loop
            ww += w[x]
        if ww != 0.0:
            w = w / ww
        fixed = [int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS)) for v in w]
        start[xx], count[xx] = xmin, n
        kk[xx, :n] = fixed
    return start, count, kk


def _pass(img, start, count, kk, axis):
    """One 8-bit pass along `axis` of an [H, W, C] uint8 image."""
    img = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((len(start),) + img.shape[1:], np.uint8)
    for o in range(len(start)):
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for k in range(int(count[o])):
            acc += img[start[o] + k] * int(kk[o, k])
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img, out_h, out_w):
    """[H, W, C] uint8 -> [out_h, out_w, C] uint8, bit-identical to PIL.Image.resize((out_w, out_h), BILINEAR)."""
    h, w = img.shape[:2]
    if (h, w) == (out_h, out_w):
        return img.copy()
    out = img
    if w != out_w:
        out = _pass(out, *resize_tables(w, out_w), axis=1)
    if h != out_h:
        out = _pass(out, *resize_tables(h, out_h), axis=0)
    return out
