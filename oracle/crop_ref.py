"""TEST INFRASTRUCTURE ONLY — numpy restatement of the proposal crop + intrinsics update the drivers run on the host before
the DINOv2 / LoFTR calls: box expansion (eval_linemod_json.py:73-82), `get_affine_transform`
(utils/data_utils.py:22-52), `get_image_crop_resize` / `get_K_crop_resize` (:239-280) and the two-step use of them
(eval_linemod_json.py:83-90).  Checker for pope_amd/crops.py (preprocess.hip:crop_warp_kernel); never imported by the product.

PARITY UNPINNED for the pixels: the reference calls `cv2.warpAffine(image, M, (w, h), flags=cv2.INTER_LINEAR)` and
`cv2.getAffineTransform`; cv2 is absent from this image, unpinned in requirements.txt:21, and no reference fixture covers
it.  `warp_affine_u8` restates OpenCV's published 8-bit bilinear convention (imgproc: warpAffine -> remap with fixed-point
tables): the inverse map is evaluated per destination pixel in 10-bit fixed point with a rounding offset of 1/64 px,
truncated to 1/32 px; the four neighbours are blended with integer weights (32 - fx)(32 - fy), fx (32 - fy), (32 - fx) fy,
fx fy (sum 1024) and rounded to nearest; pixels outside the source are 0 (BORDER_CONSTANT).  What follows from that
convention without cv2 and is tested: an integer translation is an exact copy; a linear ramp is reproduced to the 1/32 px
quantisation.  The intrinsics update is closed-form fp64 and exact.
"""
import numpy as np


def expand_box(bbox_xywh, compact_percent=0.3):
    """eval_linemod_json.py:73-82: SAM's XYWH box grown by int(w * 0.3) / int(h * 0.3) on every side -> [x0, y0, x1, y1]."""
    x0, y0, w, h = (int(v) for v in bbox_xywh)
    x1, y1 = x0 + w, y0 + h
    x0 -= int(w * compact_percent)
    y0 -= int(h * compact_percent)
    x1 += int(w * compact_percent)
    y1 += int(h * compact_percent)
    return np.array([x0, y0, x1, y1])


def get_dir(src_point, rot_rad):
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return [src_point[0] * cs - src_point[1] * sn, src_point[0] * sn + src_point[1] * cs]


def get_3rd_point(a, b):
    direct = a - b
    return b + np.array([-direct[1], direct[0]], dtype=np.float32)


def get_affine_transform(center, scale, rot, output_size, shift=np.array([0, 0], dtype=np.float32), inv=0):
    """utils/data_utils.py:22-52: three point pairs in float32 (centre, centre + rotated [0, -src_w / 2], their 90-degree
    companion) -> the 2x3 affine through them (cv2.getAffineTransform = the 6x6 linear solve, fp64)."""
    if not isinstance(scale, np.ndarray) and not isinstance(scale, list):
        scale = np.array([scale, scale], dtype=np.float32)
    scale_tmp = scale
    src_w, dst_w, dst_h = scale_tmp[0], output_size[0], output_size[1]
    src_dir = get_dir([0, src_w * -0.5], np.pi * rot / 180)
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src, dst = np.zeros((3, 2), dtype=np.float32), np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale_tmp * shift
    src[1, :] = center + src_dir + scale_tmp * shift
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + dst_dir
    src[2:, :] = get_3rd_point(src[0, :], src[1, :])
    dst[2:, :] = get_3rd_point(dst[0, :], dst[1, :])
    a, b = (dst, src) if inv else (src, dst)
    A = np.concatenate([a.astype(np.float64), np.ones((3, 1))], 1)
    return np.linalg.solve(A, b.astype(np.float64)).T          # [2, 3]


def invert_affine(M):
    """The inversion cv::warpAffine applies to a forward matrix (no WARP_INVERSE_MAP)."""
    M = np.asarray(M, np.float64).reshape(2, 3).copy()
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    M[0, 0], M[0, 1], M[1, 0], M[1, 1] = A11, M[0, 1] * -D, M[1, 0] * -D, A22
    b1 = -M[0, 0] * M[0, 2] - M[0, 1] * M[1, 2]
    b2 = -M[1, 0] * M[0, 2] - M[1, 1] * M[1, 2]
    M[0, 2], M[1, 2] = b1, b2
    return M


def warp_affine_u8(image, M, dsize):
    """cv2.warpAffine(image, M, dsize=(w, h), flags=INTER_LINEAR) for uint8 HWC (or HW) images, border constant 0."""
    img = np.asarray(image)
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    H, W, _ = img.shape
    ow, oh = int(dsize[0]), int(dsize[1])
    Mi = invert_affine(M)
    xs, ys = np.arange(ow, dtype=np.float64), np.arange(oh, dtype=np.float64)
    adelta = np.rint(Mi[0, 0] * xs * 1024).astype(np.int64)
    bdelta = np.rint(Mi[1, 0] * xs * 1024).astype(np.int64)
    X0 = np.rint((Mi[0, 1] * ys + Mi[0, 2]) * 1024).astype(np.int64) + 16
    Y0 = np.rint((Mi[1, 1] * ys + Mi[1, 2]) * 1024).astype(np.int64) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx, sy, fx, fy = X >> 5, Y >> 5, X & 31, Y & 31

    def px(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64)
        return v * ok[:, :, None]

    w00, w01, w10, w11 = (32 - fx) * (32 - fy), fx * (32 - fy), (32 - fx) * fy, fx * fy
    acc = px(sy, sx) * w00[:, :, None] + px(sy, sx + 1) * w01[:, :, None] + px(sy + 1, sx) * w10[:, :, None] \
        + px(sy + 1, sx + 1) * w11[:, :, None]
    out = ((acc + 512) >> 10).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def get_image_crop_resize(image, box, resize_shape):
    """utils/data_utils.py:239-255 -> (image_crop [h, w(, c)], trans_crop_homo [3, 3])."""
    center = np.array([(box[0] + box[2]) / 2.0, (box[1] + box[3]) / 2.0])
    scale = np.array([box[2] - box[0], box[3] - box[1]])
    resize_h, resize_w = resize_shape
    trans = get_affine_transform(center, scale, 0, [resize_w, resize_h])
    return warp_affine_u8(image, trans, (resize_w, resize_h)), np.concatenate([trans, np.array([[0, 0, 1]])], axis=0)


def get_K_crop_resize(box, K_orig, resize_shape):
    """utils/data_utils.py:258-280 -> (K_crop [3, 3], K_crop_homo [3, 4])."""
    center = np.array([(box[0] + box[2]) / 2.0, (box[1] + box[3]) / 2.0])
    scale = np.array([box[2] - box[0], box[3] - box[1]])
    resize_h, resize_w = resize_shape
    trans = get_affine_transform(center, scale, 0, [resize_w, resize_h])
    trans_homo = np.concatenate([trans, np.array([[0, 0, 1]])], axis=0)
    K_orig = np.asarray(K_orig)
    K_homo = np.concatenate([K_orig, np.zeros((3, 1))], axis=-1) if K_orig.shape == (3, 3) else K_orig.copy()
    assert K_homo.shape == (3, 4)
    K_crop_homo = trans_homo @ K_homo
    return K_crop_homo[:3, :3], K_crop_homo


def crop_proposal(image, bbox_xywh, K, out_size=256, compact_percent=0.3):
    """eval_linemod_json.py:73-90 for one SAM proposal: expanded box -> crop at the box's own size (an integer
    translation) -> uniform resize by out_size / w about the centre -> (crop [out, out, 3] uint8, K_crop [3, 3], box)."""
    box = expand_box(bbox_xywh, compact_percent)
    x0, y0, x1, y1 = box
    shape1 = np.array([y1 - y0, x1 - x0])
    K_crop, _ = get_K_crop_resize(box, K, shape1)
    crop, _ = get_image_crop_resize(image, box, shape1)
    box_new = np.array([0, 0, x1 - x0, y1 - y0])
    shape2 = np.array([out_size, out_size])
    K_crop, _ = get_K_crop_resize(box_new, K_crop, shape2)
    crop, _ = get_image_crop_resize(crop, box_new, shape2)
    return crop, K_crop, box
