"""Batch sharding of the planning path across the GPUs of one node.

Utterances are independent (per-utterance losses), so rank r plans the contiguous block
``[r * B / G, (r + 1) * B / G)`` with replicated weights and NO collective inside the loop; the only
exchange is one ``all_gather`` of the final CP trajectories (RCCL over xGMI with backend "nccl", gloo in the
CPU tests), after the last iteration.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n_utterances: int, rank: int, world: int):
    """Contiguous, balanced blocks: the first (n % world) ranks hold one utterance more."""
    base, extra = divmod(n_utterances, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_final_cp(cp_local: torch.Tensor, n_utterances: int):
    """all_gather of the per-rank CP blocks (B_r, T, C) -> (B, T, C) on every rank.  Blocks may differ by one row,
    so they are padded to the largest block for the collective and trimmed afterwards."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return cp_local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_bounds(n_utterances, r, world) for r in range(world)]
    bmax = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((bmax,) + tuple(cp_local.shape[1:]), dtype=cp_local.dtype, device=cp_local.device)
    pad[: cp_local.shape[0]] = cp_local
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(outs, sizes)], dim=0)


class ShardFailed(RuntimeError):
    """Raised on EVERY rank of plan_sharded when at least one rank could not produce its block (its own error is chained on the
    rank that failed): nobody is left waiting in the gather."""


def _all_ok(ok: bool, device) -> bool:
    """min over ranks of a one-word "my block is valid" flag: the one extra collective that keeps a failing rank from leaving its
    peers blocked in the all_gather (NCCL / RCCL collectives need device tensors, gloo takes host ones)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return ok
    dev = device if dist.get_backend() == "nccl" else torch.device("cpu")
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


def plan_sharded(make_planner, cp0, target_mel, target_semvec, n_iters: int):
    """Plans this rank's block with ``make_planner(batch=...)`` (a HipPlanner factory on the GPU; tests inject the
    CPU oracle) and returns (final CP of ALL utterances, this rank's loss log).

    Failure semantics: whatever goes wrong on one rank while it builds its planner, steps it or checks the device status word
    (``planner.check()``: a timed-out in-kernel wait, a launch that was not resident) is agreed on by ALL ranks before the gather
    -- every rank raises ``ShardFailed`` (the failing one with its own exception as the cause) and none enters the all_gather."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    n = cp0.shape[0]
    if n < world:   # a rank without utterances could not build a planner while the others wait in the gather: refuse on EVERY rank
        raise ValueError(f"plan_sharded: {n} utterances cannot be sharded over {world} ranks (need at least one per rank)")
    lo, hi = shard_bounds(n, rank, world)
    err, cp, loss = None, None, None
    try:
        planner = make_planner(batch=hi - lo)
        planner.set_targets(target_mel[lo:hi], None if target_semvec is None else target_semvec[lo:hi])
        planner.set_cp(cp0[lo:hi])
        loss = planner.step(n_iters)
        if hasattr(planner, "check"):
            planner.check()   # a timed-out device-side wait must not be gathered as if it were a plan
        cp = planner.get_cp()
        cp = cp if isinstance(cp, torch.Tensor) else torch.as_tensor(cp)
    except Exception as e:   # noqa: BLE001 -- any failure of this rank has to reach the agreement below
        err = e
    dev = cp.device if cp is not None else (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
    if not _all_ok(err is None, dev):
        if err is not None:
            raise ShardFailed(f"plan_sharded: rank {rank} failed: {err}") from err
        raise ShardFailed(f"plan_sharded: another rank failed; rank {rank}'s block is discarded")
    return gather_final_cp(cp, n), loss
