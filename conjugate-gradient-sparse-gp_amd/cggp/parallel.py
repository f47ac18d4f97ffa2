"""Row sharding of X over the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  Every N-sized
object is split into contiguous row blocks; Z, lambda, the kernel hyper-parameters and the
M-sized CG state are replicated, so every rank runs the identical CG recurrence and the only
exchange is one all-reduce(sum) of the [Bt, M] partial product per operator application.
"""

import torch.distributed as dist


def shard_bounds(N, world_size, rank):
    """Contiguous row block [lo, hi) of rank `rank`: ceil(N/G) rows each, last may be short/empty."""
    per = (N + world_size - 1) // world_size
    lo = min(N, rank * per)
    hi = min(N, lo + per)
    return lo, hi


def shard_rows(t, world_size=None, rank=None):
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_bounds(t.shape[0], world_size, rank)
    return t[lo:hi]


def make_allreduce(group=None, force=False):
    """In-place sum over ranks of a flat tensor view; None when there is a single rank.

    The tensor is a view of a buffer the caller owns (libmgp hands the partial product to the
    collective through `SgprNormalOperator`'s buffer), the call is enqueued behind the kernels
    already on torch's current stream.
    """
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return None

    host_staged = dist.get_backend(group) == "gloo"

    def _allreduce(t):
        if host_staged and t.is_cuda:  # gloo rehearsal of the N > 1 path on a single-GPU box
            c = t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)

    return _allreduce


def kmm_slab(M, world_size=None, rank=None):
    """Row slab [lo, hi) of the replicated Kmm whose s2*Kmm.p term this rank adds to its partial of
    the SGPR operator before the all-reduce (the slabs tile [0, M), so the sum is the full term)."""
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    return shard_bounds(M, world_size, rank)
