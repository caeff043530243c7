"""Candidate sharding across the GPUs of one node.

Candidates are independent through the batched forward and the TD(lambda) scoring; they couple only in
the final max / softmax / weighted mean / multinomial (learner.py:318-325).  Rank r of G therefore scores
candidates [r*N/G, (r+1)*N/G) and ONE all-gather (RCCL over xGMI, backend "nccl"; gloo on CPU in the
tests) of the per-shard scores and first actions reassembles exactly the single-GPU vectors; every rank
then runs the identical select.  The policy pass and the eps draw are replicated (same seed on every rank),
which keeps the sampled candidates identical to the 1-GPU run.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split: the first ``n_total % world`` ranks get one extra candidate."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    q, r = divmod(n_total, world)
    begin = rank * q + min(rank, r)
    return begin, q + (1 if rank < r else 0)


def world_info(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def check_same_generator(generator: torch.Generator, group=None) -> None:
    """Every rank draws the candidate noise and the multinomial variates from its own generator; the sharded planner is
    only correct when those streams are identical.  Compares seed and state over the group and raises on a mismatch."""
    rank, world = world_info(group)
    if world == 1:
        return
    st = generator.get_state()
    h = int(torch.sum(st.to(torch.int64) * (torch.arange(st.numel(), dtype=torch.int64) % 65521 + 1)).item()) & 0x7FFFFFFFFFFFFFFF
    mine = torch.tensor([generator.initial_seed() & 0x7FFFFFFFFFFFFFFF, h], dtype=torch.int64)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    mine = mine.to(dev)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    for r, o in enumerate(out):
        if not torch.equal(o.cpu(), mine.cpu()):
            raise RuntimeError(f"rank {rank}: torch.Generator seed/state differs from rank {r}; seed the planner's generator "
                               "identically on every rank")


def gather_candidates(er_shard: torch.Tensor, a0_shard: torch.Tensor, n_total: int, group=None, force: bool = False):
    """All-gather per-shard scores (n_r,) and first actions (n_r, A) into full (N,) / (N, A) tensors in
    candidate order.  One collective: scores and actions travel in a single packed (n_r, 1+A) buffer,
    padded to the largest shard so the collective is fixed-size (unequal shards only when N % G != 0).
    force: run the collective even in a world of one (tests/test_dist_gpu.py: RCCL on the compute stream on a one-GPU box)."""
    rank, world = world_info(group)
    if world == 1 and not (force and dist.is_available() and dist.is_initialized()):
        return er_shard, a0_shard
    if er_shard.is_cuda and dist.get_backend(group) != "nccl":
        # a CPU-only backend (gloo in the tests): the packed buffer travels through the host
        er_h, a0_h = gather_candidates(er_shard.cpu(), a0_shard.cpu(), n_total, group)
        return er_h.to(er_shard.device), a0_h.to(er_shard.device)
    A = a0_shard.shape[1]
    if n_total % world == 0:
        # equal shards (the usual case): pack with one cat, gather, one copy for the scores (the re-score writes into
        # them); the first actions stay a strided view of the gathered buffer (m3pc_select takes a row stride)
        pack = torch.cat([er_shard[:, None], a0_shard], dim=1)
        out = torch.empty((n_total, 1 + A), dtype=pack.dtype, device=pack.device)
        dist.all_gather_into_tensor(out, pack, group=group)
        return out[:, 0].contiguous(), out[:, 1:]
    nmax = -(-n_total // world)
    pack = torch.zeros((nmax, 1 + A), dtype=er_shard.dtype, device=er_shard.device)
    n_r = er_shard.shape[0]
    pack[:n_r, 0] = er_shard
    pack[:n_r, 1:] = a0_shard
    out = torch.empty((world, nmax, 1 + A), dtype=pack.dtype, device=pack.device)
    dist.all_gather_into_tensor(out.view(world * nmax, 1 + A), pack, group=group)
    ers, a0s = [], []
    for r in range(world):
        _, cnt = shard_range(n_total, r, world)
        ers.append(out[r, :cnt, 0])
        a0s.append(out[r, :cnt, 1:])
    return torch.cat(ers), torch.cat(a0s)
