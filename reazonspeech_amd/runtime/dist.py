"""Multi-GPU glue: utterances are independent, so a batch shards across one
process per GPU with no data-path collective; the only exchange is a gather of the hypotheses
(token ids, emission frames, counts) — RCCL over xGMI on GPUs, gloo in the CPU tests.

The reference's only multi-GPU mechanism is process-level data parallelism in its evaluation
harness (`datasets.map(with_rank=True)`, `cuda:{rank % num_gpus}`; pkg/evaluation/src/base.py:194-212,
pkg/evaluation/examples/rs-nemo/eval.py:26) with results returned by pickling; this module is the
`torch.distributed` equivalent.
"""
import os
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def init(backend: str = "nccl"):
    if dist.is_available() and not dist.is_initialized() and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend=backend)


def is_on() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size() -> int:
    return dist.get_world_size() if is_on() else 1


def rank() -> int:
    return dist.get_rank() if is_on() else 0


def barrier():
    if is_on():
        dist.barrier()


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def max_over_ranks(x: float) -> float:
    if not is_on():
        return x
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_bounds(n_items: int, world: int, r: int) -> Tuple[int, int]:
    """contiguous, balanced [lo, hi) of rank r (first n % world ranks get one extra item)"""
    q, rem = divmod(n_items, world)
    lo = r * q + min(r, rem)
    return lo, lo + q + (1 if r < rem else 0)


def shard_by_length(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Deal utterance indices to ranks so each rank gets a contiguous run of the length-sorted
    order (uniform T' within a rank keeps its padded batch tight)."""
    order = sorted(range(len(lengths)), key=lambda i: (lengths[i], i))
    return [order[slice(*shard_bounds(len(order), world, r))] for r in range(world)]


def gather_hypotheses(ids: torch.Tensor, frames: torch.Tensor, n_ids: torch.Tensor):
    """all_gather of one rank's padded hypotheses -> ([W*B, U], [W*B, U], [W*B]) on every rank.
    One fused payload (ids | frames | n) per rank so the exchange is a single collective."""
    if not is_on():
        return ids, frames, n_ids
    W = dist.get_world_size()
    B, U = ids.shape
    payload = torch.cat([ids.reshape(-1), frames.reshape(-1), n_ids.reshape(-1)]).contiguous()
    out = torch.empty((W, payload.numel()), dtype=payload.dtype, device=payload.device)
    dist.all_gather_into_tensor(out.view(-1), payload)
    g_ids = out[:, :B * U].reshape(W * B, U)
    g_frames = out[:, B * U:2 * B * U].reshape(W * B, U)
    g_n = out[:, 2 * B * U:].reshape(W * B)
    return g_ids, g_frames, g_n
