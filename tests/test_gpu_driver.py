"""GPU: the batched per-pair driver step (pope_amd/driver.py) against the fixture captured from the
reference's sequential loop (eval_linemod_json.py:65-127,150)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(hip_lib):
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def models(dev, sd0):
    from pope_amd import synth
    from pope_amd.dinov2_utils import load_dinov2_model
    from pope_amd.matcher import Matcher, default_cfg
    vit = load_dinov2_model(state_dict=sd0).to(dev)
    matcher = Matcher(default_cfg).eval()
    matcher.load_state_dict(synth.synthetic_matcher_state_dict(seed=0), strict=True)
    return vit, matcher.to(dev)


@pytest.fixture(scope="module")
def peaked_models(dev, sd0, golden_dir):
    """DINOv2 + the LoFTR Matcher under the `peaked` synthetic weights (synth.peaked_matcher_state_dict, pinned by the
    calibration mean of the 512 x 512 reference fixture): hundreds of matches per related pair, like a trained checkpoint."""
    from pope_amd import synth
    from pope_amd.dinov2_utils import load_dinov2_model
    from pope_amd.matcher import Matcher, default_cfg
    fx = np.load(os.path.join(golden_dir, "loftr_512_peaked.npz"))
    sd = synth.peaked_matcher_state_dict(torch.from_numpy(fx["outconv_mean"]), seed=0)
    sd.pop("_calibration_mean")
    matcher = Matcher(default_cfg).eval()
    matcher.load_state_dict(sd, strict=True)
    return load_dinov2_model(state_dict=sd0).to(dev), matcher.to(dev)


def test_driver_step_matches_reference_loop(dev, models, golden_dir):
    from pope_amd import synth
    from pope_amd.driver import locate_and_match
    fx = np.load(os.path.join(golden_dir, "driver_pair.npz"))
    vit, matcher = models
    out = locate_and_match(vit, matcher, *(t.to(dev) for t in synth.synthetic_driver_case()))
    np.testing.assert_allclose(out["scores"].cpu().numpy(), fx["scores"], rtol=0, atol=2e-5)
    assert np.array_equal(out["slot_index"], fx["slot_index"])
    np.testing.assert_allclose(out["slot_scores"], fx["slot_scores"], rtol=0, atol=2e-5)
    for s in range(3):
        ref_c, got_c = fx[f"mconf_{s}"], out["mconf"][s]
        clear = np.abs(ref_c - 0.2) > 1e-3              # matches whose confidence is clear of the threshold
        assert abs(len(got_c) - len(ref_c)) <= int((~clear).sum())
        if len(got_c) == len(ref_c):
            e_conf = float(np.abs(got_c - ref_c).max()) if len(ref_c) else 0.0
            e_px = float(np.abs(out["mkpts1"][s] - fx[f"mkpts1_{s}"]).max()) if len(ref_c) else 0.0
            print(f"slot {s}: {len(ref_c)} matches, mconf max err {e_conf:.2e}, mkpts1 max err {e_px:.2e} px")
            assert e_conf <= 2e-4 and e_px <= 5e-4
            assert np.array_equal(out["mkpts0"][s], fx[f"mkpts0_{s}"])
        near = int((np.abs(ref_c - 0.9) < 1e-3).sum())    # matching_score counts mconf > 0.9
        assert abs(int(out["matching_score"][s]) - int(fx["matching_score"][s])) <= near
    if all(int((np.abs(fx[f"mconf_{s}"] - 0.9) < 1e-3).sum()) == 0 for s in range(3)):
        assert np.array_equal(out["matching_score"], fx["matching_score"])
        assert out["best_slot"] == int(fx["best_slot"]) and out["best_proposal"] == int(fx["slot_index"][fx["best_slot"]])


def test_driver_step_with_fewer_than_three_candidates(dev, models):
    """Slots that never fill are skipped (the reference would raise at eval_linemod_json.py:109)."""
    from pope_amd import synth
    from pope_amd.driver import locate_and_match
    vit, matcher = models
    ref_t, crops_t, gray_ref, gray_crops = (t.to(dev) for t in synth.synthetic_driver_case())
    out = locate_and_match(vit, matcher, ref_t, crops_t[2:3], gray_ref, gray_crops[2:3])
    assert list(out["slot_index"]) == [0, -1, -1]
    assert out["matching_score"][1] == 0 and out["matching_score"][2] == 0 and out["mconf"][1].shape == (0,)
    assert out["best_slot"] == 0 and out["best_proposal"] == 0 and len(out["mconf"][0]) > 0


def test_whole_query_from_frame_and_boxes_to_pose(dev, peaked_models):
    """locate_match_pose_u8: frame + proposal boxes -> crops + K (one launch) -> preprocessing -> vote -> LoFTR -> pose,
    against the same chain assembled from the oracles (oracle/crop_ref.py crops and intrinsics, oracle/pose_ref.py on the
    published matches).  Round 3 ran this on the plain random LoFTR weights, got 6 matches in the best slot and lowered its
    bound to the RANSAC minimum; under the `peaked` weights the planted proposals publish hundreds of matches (the load of the
    reference's trained checkpoint) and the pose stage sees a real problem."""
    from oracle import crop_ref, pose_ref
    from pope_amd import synth
    from pope_amd.driver import locate_and_match_u8, locate_match_pose_u8
    vit, matcher = peaked_models
    ref, frame, boxes, K0, K1 = synth.synthetic_frame_case()
    out = locate_match_pose_u8(vit, matcher, ref, frame, boxes, K0, K1)
    crops_ref = np.stack([crop_ref.crop_proposal(frame, b, K1)[0] for b in boxes])
    for p, b in enumerate(boxes):
        _, Kc, box = crop_ref.crop_proposal(frame, b, K1)
        assert np.array_equal(out["K_crops"][p], Kc) and np.array_equal(out["boxes"][p], box)
    want = locate_and_match_u8(vit, matcher, ref, crops_ref)        # the oracle's crops through the same GPU stages
    assert torch.equal(out["scores"], want["scores"]) and list(out["slot_index"]) == list(want["slot_index"])
    assert out["best_proposal"] in (1, 4) and set(out["slot_index"]) >= {1, 4}     # the planted proposals win the vote
    s = out["best_slot"]
    assert np.array_equal(out["mkpts0"][s], want["mkpts0"][s]) and len(out["mconf"][s]) >= 300
    assert np.array_equal(out["pre_K"], out["K_crops"][out["best_proposal"]])
    ret = pose_ref.estimate_pose(out["mkpts0"][s], out["mkpts1"][s], K0, out["pre_K"], 0.5, 0.99)
    assert (ret is None) == (out["pose"] is None)
    if ret is not None:
        R, t, inl = out["pose"]
        print(f"best proposal {out['best_proposal']}: {len(out['mconf'][s])} matches, {int(inl.sum())} pose inliers")
        assert np.array_equal(inl, ret[2]) and inl.sum() >= 30
        np.testing.assert_allclose(R, ret[0], atol=1e-7)
        np.testing.assert_allclose(t, ret[1], atol=1e-7)
