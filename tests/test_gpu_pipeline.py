"""GPU: `PairPipeline` (pope_amd/pipeline.py) — the batched extract + match step bench.py times — with its ViT chunks spread
over several HIP streams must publish exactly what the single-stream pipeline publishes, starting from a model whose lazily
built caches (weight planes, pos/bias table) do not exist yet (the fork must not race their construction)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(hip_lib):
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def fresh_model(sd0, dev, precision):
    from pope_amd.dinov2_utils import load_dinov2_model
    m = load_dinov2_model(state_dict=sd0).to(dev)
    m.precision = precision
    return m


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
@pytest.mark.parametrize("streams", [2, 3])
def test_streams_bit_equal_to_single_stream_on_a_fresh_model(dev, sd0, precision, streams):
    from pope_amd import synth
    from pope_amd.pipeline import PairPipeline
    H, W, n = 224, 308, 10                      # 16 x 22 token grid; chunk 3 -> 4 chunks per side, the last one ragged
    i0, i1 = synth.synthetic_pairs(n, H, W, seed=5)
    i0, i1 = i0.to(dev), i1.to(dev)
    torch.cuda.synchronize()
    # the multi-stream pipeline runs FIRST, on a model that has never been called
    multi = PairPipeline(fresh_model(sd0, dev, precision), chunk=3, streams=streams, want_conf=True, match_precision=precision)
    got = multi(i0, i1)
    torch.cuda.synchronize()
    single = PairPipeline(fresh_model(sd0, dev, precision), chunk=3, streams=1, want_conf=True, match_precision=precision)
    want = single(i0, i1)
    assert multi._streams is not None and len(multi._streams) == streams and single._streams is None
    assert int(want["counts"].sum()) == len(want["b_ids"]) > 0
    for k in ("feat0", "feat1", "cls0", "cls1", "conf_matrix", "b_ids", "i_ids", "j_ids", "mconf", "mkpts0_c", "mkpts1_c"):
        assert torch.equal(got[k], want[k]), k
    assert torch.equal(got["counts"], want["counts"])
    again = multi(i0, i1)                       # and it is reproducible from call to call
    for k in ("feat0", "feat1", "i_ids", "j_ids", "mconf"):
        assert torch.equal(again[k], want[k]), k
    assert multi.reruns == 0 and multi.model.overflow_events == 0
