"""Batch x head sharding for multi-GPU runs.

Every (batch, kv-head-group) pair is an independent attention problem, so N GPUs split the
work with NO collective: shard the batch first, and only when there are more ranks than batch
entries split the kv-head groups as well (a GQA group is never split across ranks, so dK/dV
need no cross-rank reduction).  Pure host logic: no torch device calls, unit-tested on CPU.
"""
import contextlib
from dataclasses import dataclass


@dataclass(frozen=True)
class ShardPlan:
    rank: int
    world_size: int
    batch_start: int
    batch_stop: int
    head_k_start: int       # kv-head range [start, stop)
    head_k_stop: int
    h_ratio: int            # q heads per kv head

    @property
    def head_start(self) -> int:
        return self.head_k_start * self.h_ratio

    @property
    def head_stop(self) -> int:
        return self.head_k_stop * self.h_ratio

    @property
    def n_units(self) -> int:
        """independent (batch, q-head) problems owned by this rank"""
        return (self.batch_stop - self.batch_start) * (self.head_stop - self.head_start)


def _split(n: int, parts: int, idx: int):
    base, rem = divmod(n, parts)
    start = idx * base + min(idx, rem)
    return start, start + base + (1 if idx < rem else 0)


def plan_shards(batch: int, nheads: int, nheads_k: int, world_size: int):
    """Return the list of ShardPlan for all ranks (len == world_size).

    world_size <= batch: contiguous batch slices.  Otherwise ranks are factored as
    (batch_parts x head_parts) with batch_parts = gcd-friendly divisor of world_size that is
    <= batch, and kv heads split over head_parts.  Ranks may receive empty shards only if
    world_size > batch * nheads_k.
    """
    if nheads % nheads_k != 0:
        raise ValueError("nheads must be divisible by nheads_k")
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    ratio = nheads // nheads_k
    batch_parts = min(batch, world_size) if batch > 0 else 1
    while batch_parts > 1 and world_size % batch_parts != 0:
        batch_parts -= 1
    head_parts = world_size // batch_parts
    plans = []
    for r in range(world_size):
        bi, hi = divmod(r, head_parts)
        b0, b1 = _split(batch, batch_parts, bi)
        h0, h1 = _split(nheads_k, head_parts, hi)
        plans.append(ShardPlan(r, world_size, b0, b1, h0, h1, ratio))
    return plans


def shard_tensor(t, plan: ShardPlan, is_kv: bool):
    """Slice a (batch, seqlen, heads, d) tensor for `plan` (a strided view: the kernels take real strides)."""
    if is_kv:
        return t[plan.batch_start:plan.batch_stop, :, plan.head_k_start:plan.head_k_stop]
    return t[plan.batch_start:plan.batch_stop, :, plan.head_start:plan.head_stop]


@contextlib.contextmanager
def problem_policy(batch: int, nheads: int):
    """Run (batch, head) shards of a `batch` x `nheads` problem with the kernels the WHOLE problem would get, hence bit-identical to the unsharded call.
    FA_POLICY_AUTO sizes a head_dim-128 launch by its workgroup count (include/flash_attn_gfx950.h, fa_set_kernel_policy); a shard is a smaller launch and
    could otherwise be served by the other kernel set.  Process-wide state (C ABI fa_set_policy_problem_heads), restored on exit:
        with problem_policy(b, h):
            for plan in plan_shards(b, h, hk, world): ... fwd(shard_tensor(q, plan, False), ...)"""
    from . import capi

    prev = capi.set_policy_problem_heads(int(batch) * int(nheads))
    try:
        yield
    finally:
        capi.set_policy_problem_heads(prev)
