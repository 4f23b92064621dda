"""Proposal crops and their intrinsics: what the drivers do on the host with two
`cv2.warpAffine` calls and two 3x3 products per SAM proposal before anything reaches the GPU
(eval_linemod_json.py:73-90; `get_affine_transform` / `get_image_crop_resize` / `get_K_crop_resize`,
utils/data_utils.py:22-52,239-280), for all P proposals of a frame in ONE kernel launch that reads the frame where it lies
in HBM and writes the [P, 256, 256, 3] uint8 batch `set_torch_images` / `gray_batch` consume.

The intrinsics are closed-form fp64 on the host (a translation, then a uniform scale about the crop centre).  The pixels
follow OpenCV's 8-bit bilinear convention (pope_hip.h:pope_crop_warp_u8); cv2 is not in this image, so that convention is
restated, not pinned: integer translations are exact copies, anything else is OpenCV's arithmetic as published.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, on_device_of, require_cuda, stream_of


def expand_box(bbox_xywh, compact_percent=0.3):
    """eval_linemod_json.py:73-82: SAM's XYWH box grown by int(w * 0.3) / int(h * 0.3) on every side -> [x0, y0, x1, y1]
    (may leave the frame: the crop is zero-padded there)."""
    x0, y0, w, h = (int(v) for v in bbox_xywh)
    x1, y1 = x0 + w, y0 + h
    return np.array([x0 - int(w * compact_percent), y0 - int(h * compact_percent), x1 + int(w * compact_percent), y1 + int(h * compact_percent)])


def get_affine_transform(center, scale, rot, output_size, shift=np.array([0, 0], dtype=np.float32), inv=0):
    """utils/data_utils.py:22-52 -> the 2x3 matrix (fp64) through the function's three float32 point pairs: a similarity of
    scale output_size[0] / scale[0] (WIDTHS only) and rotation `rot` degrees that takes `center` (+ scale * shift) to the
    centre of the output."""
    if not isinstance(scale, (np.ndarray, list)):
        scale = np.array([scale, scale], dtype=np.float32)
    src_w, dst_w, dst_h = scale[0], output_size[0], output_size[1]
    rad = np.pi * rot / 180
    sn, cs = np.sin(rad), np.cos(rad)
    src_dir = [0 * cs - (src_w * -0.5) * sn, 0 * sn + (src_w * -0.5) * cs]
    src, dst = np.zeros((3, 2), dtype=np.float32), np.zeros((3, 2), dtype=np.float32)
    src[0] = center + scale * shift
    src[1] = center + src_dir + scale * shift
    dst[0] = [dst_w * 0.5, dst_h * 0.5]
    dst[1] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + np.array([0, dst_w * -0.5], np.float32)
    for pts in (src, dst):   # get_3rd_point: the second point turned by 90 degrees about... the first
        d = pts[0] - pts[1]
        pts[2] = pts[1] + np.array([-d[1], d[0]], dtype=np.float32)
    a, b = (dst, src) if inv else (src, dst)
    return np.linalg.solve(np.concatenate([a.astype(np.float64), np.ones((3, 1))], 1), b.astype(np.float64)).T


def _invert(M):
    """cv::warpAffine's inversion of a forward 2x3 matrix."""
    M = np.asarray(M, np.float64).reshape(2, 3).copy()
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    M[0, 0], M[0, 1], M[1, 0], M[1, 1] = A11, M[0, 1] * -D, M[1, 0] * -D, A22
    b1 = -M[0, 0] * M[0, 2] - M[0, 1] * M[1, 2]
    b2 = -M[1, 0] * M[0, 2] - M[1, 1] * M[1, 2]
    M[0, 2], M[1, 2] = b1, b2
    return M


def _box_transform(box, resize_shape):
    center = np.array([(box[0] + box[2]) / 2.0, (box[1] + box[3]) / 2.0])
    scale = np.array([box[2] - box[0], box[3] - box[1]])
    resize_h, resize_w = resize_shape
    return get_affine_transform(center, scale, 0, [resize_w, resize_h]), int(resize_h), int(resize_w)


def get_K_crop_resize(box, K_orig, resize_shape):
    """utils/data_utils.py:258-280, same signature: -> (K_crop [3, 3], K_crop_homo [3, 4]), fp64 on the host."""
    trans, _, _ = _box_transform(box, resize_shape)
    trans_homo = np.concatenate([trans, np.array([[0, 0, 1]])], axis=0)
    K_orig = np.asarray(K_orig)
    K_homo = np.concatenate([K_orig, np.zeros((3, 1))], axis=-1) if K_orig.shape == (3, 3) else K_orig.copy()
    assert K_homo.shape == (3, 4)
    K_crop_homo = trans_homo @ K_homo
    return K_crop_homo[:3, :3], K_crop_homo


@torch.no_grad()
def warp_batch(image, minv, windows, out_hw):
    """One launch of pope_crop_warp_u8: image [H, W, C] uint8 (CUDA); minv [P, 2, 3] inverse maps (array-like, fp64);
    windows [P, 4] (x0, y0, w, h) -> [P, oh, ow, C] uint8 (CUDA)."""
    require_cuda(image, "warp_batch")
    if image.dtype != torch.uint8 or image.dim() != 3:
        raise TypeError("warp_batch expects a uint8 [H, W, C] tensor")
    image = image.contiguous()
    H, W, Cn = image.shape
    dev = image.device
    m = torch.as_tensor(np.ascontiguousarray(np.asarray(minv, np.float64).reshape(-1, 6))).to(dev)
    w = torch.as_tensor(np.ascontiguousarray(np.asarray(windows, np.int32).reshape(-1, 4))).to(dev)
    P = int(m.shape[0])
    oh, ow = int(out_hw[0]), int(out_hw[1])
    out = torch.empty(P, oh, ow, Cn, dtype=torch.uint8, device=dev)
    if P == 0:
        return out
    with on_device_of(image):
        check(_lib.lib().pope_crop_warp_u8(C.c_void_p(image.data_ptr()), H, W, Cn, C.c_void_p(m.data_ptr()), C.c_void_p(w.data_ptr()), P, oh,
                                           ow, C.c_void_p(out.data_ptr()), stream_of(dev)), "pope_crop_warp_u8")
    return out


def get_image_crop_resize(image, box, resize_shape, device="cuda:0"):
    """utils/data_utils.py:239-255, same signature: HWC (or HW) uint8 numpy image -> (image_crop numpy, trans_crop_homo
    [3, 3]).  The single-crop form of `crop_proposals` (one upload, one launch, one download)."""
    img = np.asarray(image)
    squeeze = img.ndim == 2
    trans, oh, ow = _box_transform(box, resize_shape)
    t = torch.from_numpy(np.ascontiguousarray(img[:, :, None] if squeeze else img)).to(device)
    out = warp_batch(t, _invert(trans)[None], [[0, 0, img.shape[1], img.shape[0]]], (oh, ow))[0].cpu().numpy()
    return (out[:, :, 0] if squeeze else out), np.concatenate([trans, np.array([[0, 0, 1]])], axis=0)


@torch.no_grad()
def crop_proposals(image, bboxes_xywh, K, out_size=256, compact_percent=0.3):
    """eval_linemod_json.py:73-90 for ALL proposals of a frame at once.

    image        [H, W, 3] uint8 BGR frame — a CUDA tensor (stays where it is) or a numpy array (uploaded once);
    bboxes_xywh  [P, 4] SAM boxes (x, y, w, h);   K  [3, 3] intrinsics of the frame.
    Returns {"crops": [P, out, out, 3] uint8 CUDA (the drivers' `image_crop`s, ready for set_torch_images / gray_batch),
             "K": [P, 3, 3] fp64 numpy (their `K_crop`s), "boxes": [P, 4] expanded boxes (their `mask["bbox"]`)}.
    Per proposal: the box is expanded by 30 %; step 1 crops it at its own size (an integer translation; outside the frame
    -> 0); step 2 scales that crop by out / w about its centre into out x out (rows beyond the crop -> 0).  Both steps are
    one gather per output pixel from the frame."""
    if not isinstance(image, torch.Tensor):
        image = torch.from_numpy(np.ascontiguousarray(image)).to("cuda:0")
    boxes = np.array([expand_box(b, compact_percent) for b in np.asarray(bboxes_xywh).reshape(-1, 4)]).reshape(-1, 4)
    minv, wins, Ks = [], [], []
    for x0, y0, x1, y1 in boxes:
        w, h = int(x1 - x0), int(y1 - y0)
        K1, _ = get_K_crop_resize([x0, y0, x1, y1], K, [h, w])
        K2, _ = get_K_crop_resize([0, 0, w, h], K1, [out_size, out_size])
        trans2, _, _ = _box_transform([0, 0, w, h], [out_size, out_size])
        minv.append(_invert(trans2))
        wins.append([x0, y0, w, h])
        Ks.append(K2)
    crops = warp_batch(image, np.array(minv).reshape(-1, 2, 3), np.array(wins).reshape(-1, 4), (out_size, out_size))
    return {"crops": crops, "K": np.array(Ks).reshape(-1, 3, 3), "boxes": boxes}
