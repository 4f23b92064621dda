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
