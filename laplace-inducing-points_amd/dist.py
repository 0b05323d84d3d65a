"""Multi-GPU: the data sum of the GGN shards across ranks (one process per GPU, RCCL over xGMI).

GGN v = sum_i J_i^T H_i J_i v is a sum of independent per-example terms (``src/ggn.py:136-144``): rank r
holds theta, the probe block V and its slice Z_r of the data; the partial products are combined by ONE
all-reduce (sum) of the (P, D) block per matvec.  W^T outputs disjoint row slices (all-gather of d
floats); W consumes slices and its (P, D) outputs are all-reduced.  The reference has no multi-device
code at all; this is new design, covered by world_size-2 gloo tests on the CPU.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced slices: the first n % world_size ranks get one extra example."""
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ShardedDataSum:
    """Wrap a rank-local block operator (the partial sum over this rank's examples) into the global one.

    ``local(V[, out=])`` -> (P, D) must already carry the global recalibration N/M_total (so that the partial
    sums simply add) but NOT the prior term alpha*V, which is added once after the reduction.

    With ``chunk`` set, the probe block is processed in chunks and the all-reduce of chunk c runs (on RCCL's
    stream) while chunk c+1 is being computed: the collective is per-link bound over xGMI and uses few CUs,
    so it hides behind the MFMA-bound sweep.  Still exactly one all-reduce per matvec *per probe*.
    ``chunk`` is a size (equal chunks) or a sequence of fractions of the block, e.g. ``(0.75, 0.25)``: only the
    LAST chunk's collective is exposed, and the sweep loses throughput on small probe blocks (measured: 1582 /
    1543 / 1490 GGN-vp/s at 256 / 128 / 64 probes), so a large first and a small last chunk beat equal ones."""

    def __init__(self, local: Callable[..., torch.Tensor], alpha: float = 0.0,
                 group: Optional[dist.ProcessGroup] = None, chunk=None, profile: bool = False):
        self.local, self.alpha, self.group, self.chunk = local, float(alpha), group, chunk
        # profile: per call, the bytes handed to the all-reduce and the time the compute stream spends waiting for
        # collectives it could not hide (events around the waits on the GPU, wall clock on the CPU) -> self.stats
        self.profile = profile
        self.stats = dict(calls=0, allreduce_bytes=0, exposed_wait_ms=0.0)
        self._pending_events = []

    def _tic(self, ref: torch.Tensor):
        if not self.profile:
            return None
        if ref.is_cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        import time
        return time.perf_counter()

    def _toc(self, t0, ref: torch.Tensor):
        if t0 is None:
            return
        if ref.is_cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._pending_events.append((t0, e))         # read lazily: no synchronisation inside the matvec
        else:
            import time
            self.stats["exposed_wait_ms"] += 1e3 * (time.perf_counter() - t0)

    def read_stats(self):
        """Totals since construction (synchronises the events recorded so far)."""
        for a, b in self._pending_events:
            b.synchronize()
            self.stats["exposed_wait_ms"] += a.elapsed_time(b)
        self._pending_events = []
        return dict(self.stats)

    def _bounds(self, P: int):
        if isinstance(self.chunk, (tuple, list)):
            cuts, acc = [0], 0.0
            for f in self.chunk[:-1]:
                acc += float(f)
                cuts.append(min(P, max(cuts[-1], int(round(acc * P)))))
            cuts.append(P)
            return [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
        return [(c0, min(P, c0 + int(self.chunk))) for c0 in range(0, P, int(self.chunk))]

    def _world(self) -> int:
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group)
        return 1

    def __call__(self, V: torch.Tensor) -> torch.Tensor:
        world = self._world()
        if world > 1 and self.chunk and V.dim() == 2 and len(self._bounds(V.shape[0])) > 1:
            Y = torch.empty_like(V)
            pending = []
            for c0, c1 in self._bounds(V.shape[0]):
                self.local(V[c0:c1], out=Y[c0:c1])
                pending.append(dist.all_reduce(Y[c0:c1], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            t0 = self._tic(Y)
            for h in pending:
                h.wait()
            self._toc(t0, Y)
        else:
            Y = self.local(V)
            if world > 1:
                t0 = self._tic(Y)
                dist.all_reduce(Y, op=dist.ReduceOp.SUM, group=self.group)      # the one collective per matvec
                self._toc(t0, Y)
        if self.profile and world > 1:
            self.stats["calls"] += 1
            self.stats["allreduce_bytes"] += Y.numel() * Y.element_size()
        if self.alpha != 0.0:
            Y = Y.add_(V.reshape(Y.shape), alpha=self.alpha) if Y.is_cuda else Y + self.alpha * V.reshape(Y.shape)
        return Y


def sharded_ggn_vp(state, Z_local, model_type, alpha, n_total: int, full_set_size=None, group=None):
    """Global (GGN + alpha I) block operator from this rank's data slice (HIP engine underneath)."""
    import math
    from .ggn import get_engine
    eng = get_engine(state, Z_local, model_type)
    N = full_set_size or n_total
    scale = N / n_total * (math.exp(-float(state.params["logvar"]["logvar"])) if model_type == "regressor" else 1.0)
    return ShardedDataSum(lambda V: eng.ggn_vp(V, scale, 0.0), alpha, group), eng


def gather_rows(U_local: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather of the W^T slices (P, M_local, K) along the example axis.  ``shard_bounds`` hands out ragged
    slices (sizes differ by at most one), so the slice lengths are gathered first and every slice is padded to
    the longest one for the collective; the padding is cut away again before the concatenation."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return U_local
    world = dist.get_world_size(group)
    mine = torch.tensor([U_local.shape[1]], device=U_local.device, dtype=torch.int64)
    sizes = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(sizes, mine, group=group)
    sizes = [int(s.item()) for s in sizes]
    m_max = max(sizes)
    pad = U_local.contiguous()
    if pad.shape[1] < m_max:
        fill = torch.zeros((pad.shape[0], m_max - pad.shape[1]) + tuple(pad.shape[2:]), device=pad.device, dtype=pad.dtype)
        pad = torch.cat([pad, fill], dim=1)
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:, :m] for p, m in zip(parts, sizes)], dim=1)


def sharded_hutchinson(op: Callable[[torch.Tensor], torch.Tensor], probes: torch.Tensor, group=None) -> torch.Tensor:
    """Hutchinson estimate mean_p eps_p^T X eps_p (``src/stochtrace.py:22-34``) with the PROBES sharded over ranks:
    every rank holds the WHOLE operator, takes its contiguous slice of the ``(P, D)`` probe block, and the partial
    sums of the quadratic forms are combined by one scalar all-reduce.

    ``op`` must be rank-local (a (p, D) -> (p, D) map that issues no collective on ``group``): the ranks call it on
    DIFFERENT probe slices — of different lengths when P % world != 0, and not at all when a slice is empty — so a
    :class:`ShardedDataSum` over the same group would add products of different probes, or deadlock.  Such an
    operator is refused; to shard data and probes together give the data sum its own (disjoint) process group."""
    if isinstance(op, ShardedDataSum) and op._world() > 1:
        world_ranks = lambda g: sorted(dist.get_process_group_ranks(g if g is not None else dist.group.WORLD))
        if set(world_ranks(op.group)) & set(world_ranks(group)) != {dist.get_rank()}:
            raise ValueError("sharded_hutchinson: `op` all-reduces over ranks that also shard the probes; pass a "
                             "rank-local operator, or a ShardedDataSum whose group is disjoint from the probe group")
    P = probes.shape[0]
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    lo, hi = shard_bounds(P, world, rank)
    if hi > lo:
        mine = probes[lo:hi]
        part = (mine * op(mine)).sum(dtype=torch.float64)
    else:
        part = torch.zeros((), dtype=torch.float64, device=probes.device)
    if world > 1:
        dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
    return part / P
