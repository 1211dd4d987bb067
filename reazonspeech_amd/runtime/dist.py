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


_single_rank_collectives = False


def use_collectives_with_one_rank(on: bool):
    """Hardware check of the exchange without a second GPU: with a process group of ONE rank the gather / all_reduce of
    this module are normally skipped; switched on, they run through the backend (RCCL on a GPU box) all the same."""
    global _single_rank_collectives
    _single_rank_collectives = bool(on)


def is_on() -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _single_rank_collectives)


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


def shard_balanced(lengths: Sequence[int], world: int, max_batch: int = 256) -> Tuple[List[List[int]], int]:
    """Ragged input: deal length-sorted CHUNKS to ranks in snake order (0 .. W-1, W-1 .. 0, ...), an even number of chunks
    per rank, so that every rank pairs short chunks with long ones: per-rank audio seconds AND per-rank padded work come
    out equal (contiguous runs of the sorted order give the last rank ~3x the first rank's audio on U(2 s, 10 s)), while a
    chunk — one batch on its rank — is still a run of neighbours in the sorted order, i.e. tightly padded.
    -> (indices per rank, chunk size = the batch size each rank should cut its shard into)."""
    n = len(lengths)
    order = sorted(range(n), key=lambda i: (lengths[i], i))
    if n == 0 or world <= 1:
        return [order] + [[] for _ in range(max(world, 1) - 1)], max_batch
    per_rank = -(-n // world)
    k = max(2, -(-per_rank // max_batch))
    k += k & 1                                                   # even: the snake pairs chunk c with chunk 2W-1-c
    chunk = max(1, -(-n // (world * k)))
    shards: List[List[int]] = [[] for _ in range(world)]
    for c in range(world * k):                                   # exactly W * k chunks whose sizes differ by at most one
        rnd, pos = divmod(c, world)
        lo, hi = shard_bounds(n, world * k, c)
        shards[pos if rnd % 2 == 0 else world - 1 - pos].extend(order[lo:hi])
    return shards, chunk


def shard_plan(lengths: Sequence[int], world: int, max_batch: int = 256, mode: str = "auto") -> Tuple[List[List[int]], int]:
    """How `sharded_decode` deals utterances: "contiguous" = runs of the length-sorted order (right for
    equal-length input such as BASELINE configs[2], where any dealing is balanced and whole batches of `max_batch` are the
    most efficient), "balanced" = `shard_balanced`, "auto" = contiguous when every rank's padded work would be within 5 % of
    the others' anyway (longest utterance <= 1.05 x the shortest), balanced otherwise."""
    if mode not in ("auto", "contiguous", "balanced"):
        raise ValueError(f"shard mode {mode!r}")
    if mode == "auto":
        mode = "contiguous" if (len(lengths) == 0 or max(lengths) <= 1.05 * max(min(lengths), 1)) else "balanced"
    if mode == "contiguous":
        return shard_by_length(lengths, world), max_batch
    return shard_balanced(lengths, world, max_batch)


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


def _collective_device() -> torch.device:
    """RCCL moves device tensors, gloo host tensors"""
    if is_on() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def sharded_decode(lengths: Sequence[int], run_local, counters=None, max_batch: int = 256, mode: str = "auto"):
    """The multi-GPU entry point of the path (BASELINE.json configs[2]: 2048 utterances over 8 GPUs; reference
    mechanism: one process per GPU, pkg/evaluation/src/base.py:194-212, examples/rs-nemo/eval.py:24-28).

    Every rank calls this with the SAME `lengths` (samples per utterance, caller order).  The utterances are dealt
    to ranks by `shard_plan` (equal lengths: contiguous runs; ragged: length-sorted chunks in snake order, balanced audio
    seconds per rank), rank r decodes its shard with `run_local(indices[, batch]) -> (ids, frames, enc_lens[, scores])`
    (lists in shard order; `batch` = the batch size the plan was made for, passed when `run_local` takes two arguments;
    `scores` = hypothesis log-probabilities of the beam search, or None), and ONE all_gather of
    the padded hypotheses (count | encoder length | score bits | ids | frames fused into one int32 payload per rank)
    returns every utterance's result to every rank, restored to caller order.  There is no other collective besides a
    MAX all_reduce that agrees on the payload width.

    -> (ids, frames, enc_lens, scores): lists of length len(lengths) in caller order (`scores` None when no rank had any)."""
    n = len(lengths)
    W, r = world_size(), rank()
    shards, batch = shard_plan(lengths, W, max_batch, mode)
    import inspect
    try:
        two = len(inspect.signature(run_local).parameters) >= 2
    except (TypeError, ValueError):
        two = False
    res = run_local(list(shards[r]), batch) if two else run_local(list(shards[r]))
    ids, frames, enc_lens = res[0], res[1], res[2]
    scores = res[3] if len(res) > 3 else None
    assert len(ids) == len(frames) == len(enc_lens) == len(shards[r]), "run_local must answer for every index it was given"
    assert scores is None or len(scores) == len(ids)
    if W == 1 and not _single_rank_collectives:
        out = ([None] * n, [None] * n, [None] * n, [None] * n if scores is not None else None)
        for k, i in enumerate(shards[0]):
            out[0][i], out[1][i], out[2][i] = list(ids[k]), list(frames[k]), int(enc_lens[k])
            if scores is not None:
                out[3][i] = float(scores[k])
        return out
    dev = _collective_device()
    u_local = max((len(x) for x in ids), default=0)
    # one small all_reduce agrees on the payload width and on whether scores travel (an empty shard has none to show)
    um = torch.tensor([u_local, 1 if scores is not None else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(um, op=dist.ReduceOp.MAX)
    U, has_scores = int(um[0].item()), bool(um[1].item())
    bmax = max(len(s) for s in shards)
    HEAD = 3
    pay = torch.zeros((bmax, HEAD + 2 * U), dtype=torch.int32)
    if scores is not None and len(scores):
        pay[:len(scores), 2] = torch.tensor(list(scores), dtype=torch.float32).view(torch.int32)   # bit-cast: scores stay exact
    for k in range(len(ids)):
        u = len(ids[k])
        pay[k, 0], pay[k, 1] = u, int(enc_lens[k])
        if u:
            pay[k, HEAD:HEAD + u] = torch.as_tensor(ids[k], dtype=torch.int32)
            pay[k, HEAD + U:HEAD + U + u] = torch.as_tensor(frames[k], dtype=torch.int32)
    pay = pay.to(dev).contiguous()
    got = torch.empty((W,) + tuple(pay.shape), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(got.view(-1), pay.view(-1))
    if counters is not None:
        counters["collectives"] = counters.get("collectives", 0) + 1
        counters["bytes"] = counters.get("bytes", 0) + got.numel() * 4
    got = got.cpu()
    sc = got[:, :, 2].contiguous().view(torch.float32)
    out = ([None] * n, [None] * n, [None] * n, [None] * n if has_scores else None)
    for rr in range(W):
        for k, i in enumerate(shards[rr]):
            u = int(got[rr, k, 0])
            out[0][i] = got[rr, k, HEAD:HEAD + u].tolist()
            out[1][i] = got[rr, k, HEAD + U:HEAD + U + u].tolist()
            out[2][i] = int(got[rr, k, 1])
            if has_scores:
                out[3][i] = float(sc[rr, k])
    return out
