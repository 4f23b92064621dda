"""CPU, world_size 2 over gloo: the multi-GPU path of the pipeline — contiguous sharding of the pair
list and the final all_gather of per-pair match counts (the only exchange step).
The compute kernels need a GPU; here each rank fabricates deterministic counts for its shard."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pope_amd.pipeline import gather_counts, shard_range


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 5796):  # 5796 = LINEMOD pair list
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_pairs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(n_pairs, rank, world)
        local = torch.tensor([(7 * i) % 1131 for i in range(lo, hi)], dtype=torch.int32)
        allc = gather_counts(local)
        q.put((rank, allc.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [9, 128])
def test_gather_counts_world2_gloo(n_pairs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [(7 * i) % 1131 for i in range(n_pairs)]
    assert got[0] == want and got[1] == want  # every rank sees all counts in global pair order


def test_gather_counts_single_process_is_identity():
    c = torch.tensor([3, 1, 4], dtype=torch.int32)
    assert gather_counts(c).tolist() == [3, 1, 4]


# ---- BASELINE config 4: the LINEMOD evaluation list as a sharded work list -------------------------------------------

def _fake_count(i):
    return (i * 2654435761 >> 7) % 1131


def test_pair_list_fixture_is_the_reference_walk_order(golden_dir):
    from pope_amd.pipeline import load_pair_list
    pairs = load_pair_list(os.path.join(golden_dir, "linemod_pairs.json"))
    assert pairs.shape == (5796, 4)                              # 13 objects x 6 rotation bins
    assert sorted(set(pairs[:, 0])) == list(range(13)) and sorted(set(pairs[:, 1])) == list(range(6))
    key = pairs[:, 0] * 6 + pairs[:, 1]
    assert (key[1:] >= key[:-1]).all()                           # objects, then bins, then the bin's pairs
    assert tuple(pairs[0]) == (0, 0, 458, 700) and tuple(pairs[-1]) == (12, 5, 160, 83)
    per_object = [int((pairs[:, 0] == o).sum()) for o in range(13)]
    assert per_object == [480, 366, 606, 414, 558, 528, 510, 270, 462, 462, 294, 402, 444]


def test_walk_pair_list_single_rank_ragged_tail():
    from pope_amd.pipeline import walk_pair_list
    seen = []

    def process(lo, hi):
        seen.append((lo, hi))
        return [_fake_count(i) for i in range(lo, hi)]

    counts, nb = walk_pair_list(5796, process, batch=128)
    assert nb == 46 and seen[0] == (0, 128) and seen[-1] == (5760, 5796)       # 45 full batches + a 36-pair tail
    assert counts.tolist() == [_fake_count(i) for i in range(5796)]
    with pytest.raises(ValueError):
        walk_pair_list(10, lambda lo, hi: [0] * (hi - lo - 1), batch=4)


def _walk_worker(rank, world, port, n_pairs, batch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pope_amd.pipeline import walk_pair_list
        spans = []

        def process(lo, hi):
            spans.append((lo, hi))
            return torch.tensor([_fake_count(i) for i in range(lo, hi)], dtype=torch.int32)

        counts, nb = walk_pair_list(n_pairs, process, batch=batch, rank=rank, world=world)
        q.put((rank, counts.tolist(), spans))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs,batch", [(5796, 128), (1001, 128), (7, 4)])
def test_walk_pair_list_world2_gloo_unequal_shards(n_pairs, batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_walk_worker, args=(r, 2, port, n_pairs, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: (c, s) for r, c, s in (q.get(timeout=120) for _ in procs)}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [_fake_count(i) for i in range(n_pairs)]
    assert got[0][0] == want and got[1][0] == want
    lo0, hi0 = shard_range(n_pairs, 0, 2)
    assert got[0][1][0][0] == 0 and got[0][1][-1][1] == hi0 and got[1][1][0][0] == hi0 and got[1][1][-1][1] == n_pairs
    assert all(hi - lo <= batch for spans in (got[0][1], got[1][1]) for lo, hi in spans)
    if n_pairs == 1001:   # 501 + 500 pairs: both shards end in a ragged batch, of different sizes
        assert got[0][1][-1] == (384, 501) and got[1][1][-1] == (885, 1001)


def _run_bench(*flags):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *flags], capture_output=True, text=True, env=env,
                          timeout=300, cwd=root)


def test_bench_gpus_2_launches_itself():
    """`python bench.py --gpus 2` with NO launcher must start its own two ranks (the driver calls it the way it calls
    `--gpus 1`), relay exactly rank 0's line and exit 0.  `--launch-selftest` keeps the ranks on gloo / CPU: the GPU twin of
    this test is tests/test_gpu_pipeline.py::test_bench_gpus_2_without_launcher."""
    import json
    res = _run_bench("--gpus", "2", "--launch-selftest", "ok")
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == [0, 1]


def test_bench_self_launch_propagates_a_rank_failure():
    res = _run_bench("--gpus", "2", "--launch-selftest", "fail")
    assert res.returncode != 0
    assert not [ln for ln in res.stdout.splitlines() if ln.strip()]
