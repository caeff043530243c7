"""The N > 1 host path on CPU: candidate shard ranges and the one all-gather that reassembles the
per-shard scores / first actions (gloo, world_size 2 and 3, spawned processes)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from m3pc_amd.dist import gather_candidates, shard_range


def test_shard_range_partitions_exactly():
    for n in (1, 7, 64, 1000, 1024, 16384):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0
            for (b0, c0), (b1, _) in zip(spans, spans[1:]):
                assert b0 + c0 == b1
            assert spans[-1][0] + spans[-1][1] == n
            counts = [c for _, c in spans]
            assert max(counts) - min(counts) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, A, ok):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    er = torch.randn(n_total, generator=g)
    a0 = torch.rand(n_total, A, generator=g)
    b, c = shard_range(n_total, rank, world)
    er_full, a0_full = gather_candidates(er[b:b + c].clone(), a0[b:b + c].clone(), n_total)
    good = torch.equal(er_full, er) and torch.equal(a0_full, a0)
    # every rank ends with the same vectors => identical select / argmax / multinomial everywhere
    am = torch.tensor([int(torch.argmax(er_full))])
    lst = [torch.zeros_like(am) for _ in range(world)]
    dist.all_gather(lst, am)
    good = good and all(int(x) == int(am) for x in lst)
    ok[rank] = int(good)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 1024), (2, 7), (3, 1000)])
def test_gather_candidates_gloo(world, n_total):
    ok = mp.Array("i", [0] * world)
    port = _free_port()
    procs = [mp.Process(target=_worker, args=(r, world, port, n_total, 3, ok)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert list(ok) == [1] * world


def test_gather_single_process_is_identity():
    er, a0 = torch.randn(5), torch.rand(5, 3)
    e2, a2 = gather_candidates(er, a0, 5)
    assert e2 is er and a2 is a0
