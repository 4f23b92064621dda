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
