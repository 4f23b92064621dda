"""Relative pose from the matches: `estimate_pose` / `relative_pose_error` of the reference
(src/utils/metrics.py:69-94, :10-24; call site eval_linemod_json.py:160) on the batched HIP solver (pose.hip).

`estimate_pose_batch` is the form the pipeline uses: the dense matcher's compacted device buffers go in as they are
(match coordinates fp32 [M, 2] with the pairs contiguous, per-pair counts), one launch solves every pair, results stay on
the device.  `estimate_pose` keeps the reference's signature and return value for a single pair of numpy arrays.

Parity with OpenCV (cv2.findEssentialMat / cv2.recoverPose, absent from this image and unpinned by the reference) is
unpinned; the tests hold this module to a numpy fp64 restatement of the same algorithm fed the same minimal samples
(DESIGN.md §2).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, on_device_of, require_cuda, stream_of

MAX_ITERS = 1000       # cv2.findEssentialMat's default maxIters
INFO_FIELDS = ("n_inliers", "ransac_inliers", "hypotheses", "rounds", "best_hypothesis", "best_root", "n_matches", "status")


@torch.no_grad()
def estimate_pose_batch(kpts0, kpts1, counts, K0, K1, thresh, conf=0.99999, seed=0, max_iters=MAX_ITERS):
    """B pairs in one launch.

    kpts0, kpts1 [M, 2] fp32 CUDA: pixel coordinates of the matches, the matches of pair b contiguous and the pairs in
                 order (what `dense_match` / `Matcher` publish as mkpts0_c / mkpts1_c or mkpts0_f / mkpts1_f);
    counts       [B] int32 matches per pair (CUDA or CPU; `dense_match(...)["counts"]`), sum(counts) <= M;
    K0, K1       [B, 3, 3] (or one [3, 3] for all pairs) intrinsics of image 0 / image 1 (any float dtype, any device);
    thresh       RANSAC threshold in pixels, conf its confidence (eval_linemod_json.py:160 passes 0.5, 0.99).

    Returns a dict of device tensors: R [B, 3, 3] fp64, t [B, 3] fp64 (unit norm), E [B, 3, 3], inliers [M] bool (RANSAC
    inlier AND in front of both cameras — the mask the reference returns), n_inliers [B] int32 (0: the reference returns
    `None` for that pair) and `info` [B, 8] int32 (INFO_FIELDS).  No host synchronisation."""
    require_cuda(kpts0, "estimate_pose_batch")
    require_cuda(kpts1, "estimate_pose_batch")
    if kpts0.dtype != torch.float32 or kpts1.dtype != torch.float32:
        raise TypeError("estimate_pose_batch expects float32 match coordinates")
    dev = kpts0.device
    kpts0, kpts1 = kpts0.contiguous(), kpts1.contiguous()
    M = int(kpts0.shape[0])
    if kpts0.shape != (M, 2) or kpts1.shape != (M, 2):
        raise ValueError("kpts0 / kpts1 must both be [M, 2]")
    counts = torch.as_tensor(counts).to(device=dev, dtype=torch.int32).contiguous()
    B = int(counts.numel())

    def intrinsics(K):
        K = torch.as_tensor(np.asarray(K) if not torch.is_tensor(K) else K).to(device=dev, dtype=torch.float64)
        if K.dim() == 2:
            K = K[None].expand(B, 3, 3)
        if K.shape != (B, 3, 3):
            raise ValueError("intrinsics must be [3, 3] or [B, 3, 3]")
        return K.contiguous()

    K0, K1 = intrinsics(K0), intrinsics(K1)
    R = torch.empty(B, 3, 3, dtype=torch.float64, device=dev)
    t = torch.empty(B, 3, dtype=torch.float64, device=dev)
    E = torch.empty(B, 3, 3, dtype=torch.float64, device=dev)
    inl = torch.zeros(max(M, 1), dtype=torch.uint8, device=dev)   # rows past sum(counts) are never written by the kernel
    info = torch.empty(B, 8, dtype=torch.int32, device=dev)
    if B == 0:
        return {"R": R, "t": t, "E": E, "inliers": inl[:M].bool(), "n_inliers": info[:, 0], "info": info}
    lib = _lib.lib()
    nbytes = lib.pope_estimate_pose_workspace_bytes(B, M)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    p = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
    with on_device_of(kpts0):
        check(lib.pope_estimate_pose_f64(p(kpts0) if M else p(inl), p(kpts1) if M else p(inl), p(counts), p(K0), p(K1), B, M,
                                         float(thresh), float(conf), int(max_iters), int(seed) & (2 ** 64 - 1), p(R), p(t), p(E),
                                         p(inl), p(info), p(ws), nbytes, stream_of(dev)), "pope_estimate_pose_f64")
    return {"R": R, "t": t, "E": E, "inliers": inl[:M].bool(), "n_inliers": info[:, 0], "info": info}


def estimate_pose(kpts0, kpts1, K0, K1, thresh, conf=0.99999, device="cuda:0"):
    """src/utils/metrics.py:69-94 with its signature: kpts0, kpts1 [N, 2] and K0, K1 [3, 3] numpy arrays ->
    (R [3, 3], t [3], inliers [N] bool) or None (fewer than five matches, no essential matrix, no point in front of both
    cameras)."""
    kpts0, kpts1 = np.asarray(kpts0), np.asarray(kpts1)
    if len(kpts0) < 5:
        return None
    dev = torch.device(device)
    out = estimate_pose_batch(torch.from_numpy(np.ascontiguousarray(kpts0, np.float32)).to(dev),
                              torch.from_numpy(np.ascontiguousarray(kpts1, np.float32)).to(dev),
                              torch.tensor([len(kpts0)], dtype=torch.int32), np.asarray(K0, np.float64), np.asarray(K1, np.float64),
                              thresh, conf)
    if int(out["n_inliers"][0]) == 0:
        return None
    return out["R"][0].cpu().numpy(), out["t"][0].cpu().numpy(), out["inliers"].cpu().numpy()


def five_point(x0, x1):
    """The minimal solver alone: x0, x1 [S, 5, 2] float64 CUDA tensors of normalised coordinates ->
    (E [S, 10, 3, 3] float64, n [S] int32)."""
    require_cuda(x0, "five_point")
    x0, x1 = x0.double().contiguous(), x1.double().contiguous()
    S = int(x0.shape[0])
    E = torch.empty(S, 10, 3, 3, dtype=torch.float64, device=x0.device)
    n = torch.empty(S, dtype=torch.int32, device=x0.device)
    with on_device_of(x0):
        check(_lib.lib().pope_five_point_f64(C.c_void_p(x0.data_ptr()), C.c_void_p(x1.data_ptr()), S, C.c_void_p(E.data_ptr()),
                                             C.c_void_p(n.data_ptr()), stream_of(x0.device)), "pope_five_point_f64")
    return E, n


def relative_pose_error(T_0to1, R, t, ignore_gt_t_thr=0.0):
    """src/utils/metrics.py:10-24 (host arithmetic on one 4x4 pose): angular errors (t_err, R_err) in degrees."""
    t_gt = T_0to1[:3, 3]
    n = np.linalg.norm(t) * np.linalg.norm(t_gt)
    t_err = np.rad2deg(np.arccos(np.clip(np.dot(t, t_gt) / n, -1.0, 1.0)))
    t_err = np.minimum(t_err, 180 - t_err)      # the sign of t is not observable from E
    if np.linalg.norm(t_gt) < ignore_gt_t_thr:
        t_err = 0
    cos = np.clip((np.trace(np.dot(R.T, T_0to1[:3, :3])) - 1) / 2, -1.0, 1.0)
    return t_err, np.rad2deg(np.abs(np.arccos(cos)))
