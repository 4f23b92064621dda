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


def test_bench_gpus_2_without_launcher(dev):
    """`python bench.py --gpus 2 ...` exactly as the driver would call it — no torch.distributed.run in front — must start
    its two ranks itself and print ONE line with n_gpus 2.  On this one-GPU box both ranks share cuda:0
    (`--rehearse-on-one-gpu`: gloo collectives on CPU copies); the N-rank code path — sharded inputs, per-rank step, the
    gather of match counts, barrier + MAX over ranks — is the one an 8-GPU node runs over RCCL.  Two child processes on
    the card (the pool allows six)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                          "--pairs", "4", "--chunk", "4", "--rehearse-on-one-gpu", "--no-kernel-table"],
                         capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 1 and line["scaling"] == "weak"
    assert line["verified"] is True and line["value"] > 0
