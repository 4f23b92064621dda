"""On-disk format of the cached matches: what the reference's extraction scripts write after the
hot path (linemod.py:147-171 and its siblings) and what the pose regressors read back (pose/dataset.py:60-131):

    <root>/<label>/pre_bbox/<name>.txt   [4]     np.savetxt    proposal box in the query image
    <root>/<label>/mkpts0/<name>.txt     [M, 2]  np.savetxt    matched keypoints, reference image (x, y)
    <root>/<label>/mkpts1/<name>.txt     [M, 2]  np.savetxt    matched keypoints, query image
    <root>/<label>/pre_K/<name>.txt      [3, 3]  np.savetxt    intrinsics of the cropped query
    <root>/<label>/img0|img1/<name>.png          cv2.imwrite   the two BGR crops

with label = pair_name.split("/")[0] and name = pair_name.split("/")[-1] ("<idx0>.png-<idx1>.png").  Text files use
numpy's default `%.18e` format, one row per line; pairs with fewer than 5 matches (or a degenerate K) are not written
(linemod.py:142-145).  Host-side plumbing around the accelerated path; no kernel involved.  PNGs are written with PIL
(OpenCV is not in this image): cv2.imwrite stores a BGR array so that the file's pixels are RGB — the array is flipped
before it goes to PIL, so a cv2.imread of the file returns the original BGR array."""
import os

import numpy as np

FIELDS = ("pre_bbox", "mkpts0", "mkpts1", "pre_K")


def pair_paths(root, pair_name):
    label, name = pair_name.split("/")[0], pair_name.split("/")[-1]
    base = os.path.join(root, label)
    paths = {f: os.path.join(base, f, name + ".txt") for f in FIELDS}
    paths.update({f: os.path.join(base, f, name + ".png") for f in ("img0", "img1")})
    return paths


def save_pair_points(root, pair_name, pre_bbox, mkpts0, mkpts1, pre_K, crop_img0=None, crop_img1=None):
    """Write one pair's cached matches; returns False (nothing written) for pairs the reference skips."""
    mkpts0, mkpts1, pre_K = np.asarray(mkpts0), np.asarray(mkpts1), np.asarray(pre_K)
    if mkpts0.shape[0] < 5 or mkpts1.shape[0] < 5 or pre_K.shape[0] != 3:
        return False
    paths = pair_paths(root, pair_name)
    for f, arr in zip(FIELDS, (pre_bbox, mkpts0, mkpts1, pre_K)):
        os.makedirs(os.path.dirname(paths[f]), exist_ok=True)
        np.savetxt(paths[f], np.asarray(arr))
    for f, img in (("img0", crop_img0), ("img1", crop_img1)):
        if img is not None:
            from PIL import Image
            os.makedirs(os.path.dirname(paths[f]), exist_ok=True)
            img = np.asarray(img, dtype=np.uint8)
            Image.fromarray(img[..., ::-1] if img.ndim == 3 else img).save(paths[f])
    return True


def load_pair_points(root, pair_name, with_images=False):
    """Read one pair back (pose/dataset.py:60-90): dict of float64 arrays (mkpts as [M, 2] even for M = 1), or None if
    the pair was skipped at extraction time."""
    paths = pair_paths(root, pair_name)
    if not all(os.path.exists(paths[f]) for f in FIELDS):
        return None
    out = {f: np.loadtxt(paths[f]) for f in FIELDS}
    out["mkpts0"], out["mkpts1"] = out["mkpts0"].reshape(-1, 2), out["mkpts1"].reshape(-1, 2)
    if with_images:
        from PIL import Image
        for f in ("img0", "img1"):
            if os.path.exists(paths[f]):
                img = np.asarray(Image.open(paths[f]))
                out[f] = img[..., ::-1].copy() if img.ndim == 3 else img
    return out
