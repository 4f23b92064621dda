"""CPU: the per-pair driver step — oracle restatement against the fixture captured from
the reference's own DINOv2 + Matcher modules (oracle/gen_golden.py:gen_driver)."""
import os

import numpy as np

from oracle import driver_ref
from pope_amd import synth
from pope_amd.matcher import default_cfg


def test_oracle_driver_matches_reference_fixture(sd0, golden_dir):
    fx = np.load(os.path.join(golden_dir, "driver_pair.npz"))
    out = driver_ref.locate_and_match(sd0, synth.synthetic_matcher_state_dict(seed=0), default_cfg,
                                      *synth.synthetic_driver_case())
    np.testing.assert_allclose(out["scores"], fx["scores"], rtol=0, atol=1e-6)
    assert np.array_equal(out["slot_index"], fx["slot_index"])          # slot ORDER is observable (§8 a-10)
    assert np.array_equal(out["matching_score"], fx["matching_score"]) and out["best_slot"] == int(fx["best_slot"])
    for s in range(3):
        assert np.array_equal(out["mconf"][s], fx[f"mconf_{s}"])
        assert np.array_equal(out["mkpts1"][s], fx[f"mkpts1_{s}"])
    # the case exercises: an exact score tie (proposals 2 and 3), replacement of the minimum slot, a tie in
    # matching_score resolved to the first slot
    assert fx["scores"][2] == fx["scores"][3] and list(fx["slot_index"]) == [3, 5, 2]
    assert fx["matching_score"][1] == fx["matching_score"][2] > fx["matching_score"][0] and int(fx["best_slot"]) == 1


def test_cached_match_text_format_round_trip(tmp_path):
    """the files the reference's extraction scripts leave behind (linemod.py:147-171) — numpy's default
    savetxt text, one directory per field — written and read back; short match lists are skipped as the reference does."""
    import numpy as np
    from pope_amd import points_io as pio
    rng = np.random.default_rng(0)
    name = "0801-lm1-others/lm1-3/color/458.png-700.png"
    mk0, mk1 = rng.random((37, 2)).astype(np.float32) * 256, rng.random((37, 2)).astype(np.float32) * 256
    K = np.array([[572.4, 0.0, 325.3], [0.0, 573.6, 242.0], [0.0, 0.0, 1.0]])
    bbox = np.array([101.0, 57.0, 230.0, 198.0])
    crop = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
    assert pio.save_pair_points(str(tmp_path), name, bbox, mk0, mk1, K, crop, crop[:, ::-1])
    paths = pio.pair_paths(str(tmp_path), name)
    assert paths["mkpts0"].endswith("0801-lm1-others/mkpts0/458.png-700.png.txt")
    first = open(paths["mkpts0"]).readline().split()
    assert len(first) == 2 and first[0] == "%.18e" % float(mk0[0, 0])      # np.savetxt's default format
    got = pio.load_pair_points(str(tmp_path), name, with_images=True)
    assert np.array_equal(got["mkpts0"], mk0.astype(np.float64)) and np.array_equal(got["pre_K"], K)
    assert np.array_equal(got["pre_bbox"], bbox) and np.array_equal(got["img0"], crop) and np.array_equal(got["img1"], crop[:, ::-1])
    assert not pio.save_pair_points(str(tmp_path), "obj/x/color/1.png-2.png", bbox, mk0[:4], mk1[:4], K)
    assert pio.load_pair_points(str(tmp_path), "obj/x/color/1.png-2.png") is None
