"""Row sharding of X over the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  Every N-sized
object is split into contiguous row blocks; Z, lambda, the kernel hyper-parameters and the
M-sized CG state are replicated, so every rank runs the identical CG recurrence and the only
exchange is one all-reduce(sum) of the [Bt, M] partial product per operator application.
"""

import ctypes

import torch
import torch.distributed as dist

from . import _hip


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


class Communicator:
    """One RCCL rank owned by libmgp (`mgp_comm`, include/mgp.h).  torch.distributed is used only to
    ship the 128-byte unique id from rank 0 to the others; every collective of the data path is then
    `mgp_allreduce_sum` (ncclAllReduce) on torch's current stream, or -- inside `mgp_pcg_solve` --
    issued by the library itself on the solve's stream with no Python in the step."""

    def __init__(self, group=None, device=None):
        lib = _hip.load_library()
        self.lib = lib
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        box = [None]
        if rank == 0:
            buf = (ctypes.c_char * _hip.MGP_COMM_ID_BYTES)()
            rc = lib.mgp_comm_unique_id(buf)
            if rc != 0:
                raise _hip.MgpError(f"mgp_comm_unique_id failed ({rc}): {lib.mgp_comm_last_error().decode()}")
            box[0] = bytes(buf.raw)
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(box, src=src, group=group)
        c = ctypes.c_void_p()
        idbuf = ctypes.create_string_buffer(box[0], _hip.MGP_COMM_ID_BYTES)
        rc = lib.mgp_comm_init_rank(ctypes.byref(c), device.index, world, rank, idbuf)
        if rc != 0:
            raise _hip.MgpError(f"mgp_comm_init_rank failed ({rc}): {lib.mgp_comm_last_error().decode()}")
        self.ptr, self.world_size, self.rank, self.device = c, world, rank, device

    def allreduce(self, t):
        """In-place sum of a contiguous CUDA tensor (view) over the ranks, on torch's current stream."""
        if not t.is_cuda or not t.is_contiguous():
            raise ValueError("Communicator.allreduce needs a contiguous CUDA tensor")
        stream = torch.cuda.current_stream(t.device).cuda_stream
        rc = self.lib.mgp_allreduce_sum(ctypes.c_void_p(t.data_ptr()), t.numel(), _hip.dtype_code(t), self.ptr,
                                        ctypes.c_void_p(stream))
        if rc != 0:
            raise _hip.MgpError(f"mgp_allreduce_sum failed ({rc}): {self.lib.mgp_comm_last_error().decode()}")

    def close(self):
        if getattr(self, "ptr", None):
            self.lib.mgp_comm_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class AllReduce:
    """In-place sum over ranks of a flat tensor view.  `comm` is set when the exchange is native RCCL
    (backend "nccl"): `SgprNormalOperator` then hands the communicator to libmgp and the per-step
    all-reduce never leaves the library.  Otherwise (gloo: rehearsal of the N > 1 path on CPU-staged
    buffers) the call goes through torch.distributed and libmgp reaches it by its callback hook."""

    def __init__(self, group=None):
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.host_staged = dist.get_backend(group) == "gloo"
        self.comm = None
        self.native_error = None
        if not self.host_staged:
            # libmgp's own RCCL communicator.  Creating it is a collective; should it fail on ANY rank (a
            # bootstrap problem of the node, not of the path) every rank drops to torch.distributed's RCCL
            # through the callback hook together -- agreed by one all-reduce -- rather than hang or abort
            try:
                comm = Communicator(group)
            except Exception as e:  # noqa: BLE001
                comm, self.native_error = None, repr(e)
            ok = torch.tensor([1.0 if comm is not None else 0.0], device=torch.device("cuda", torch.cuda.current_device()))
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if float(ok.item()) > 0.5:
                self.comm = comm
            else:
                if comm is not None:
                    comm.close()
                import sys
                print(f"[cggp.parallel] libmgp RCCL communicator unavailable ({self.native_error}); "
                      "using torch.distributed all_reduce through the callback hook", file=sys.stderr)

    def __call__(self, t):
        if self.comm is not None:
            self.comm.allreduce(t)
        elif t.is_cuda and self.host_staged:  # gloo: stage through the host
            c = t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)


def make_allreduce(group=None, force=False):
    """`AllReduce` for the current process group; None when there is a single rank (unless `force`,
    which runs the collective on a 1-rank group to measure its fixed per-step cost)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return None
    return AllReduce(group)


def kmm_slab(M, world_size=None, rank=None):
    """Row slab [lo, hi) of the replicated Kmm whose s2*Kmm.p term this rank adds to its partial of
    the SGPR operator before the all-reduce (the slabs tile [0, M), so the sum is the full term)."""
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    return shard_bounds(M, world_size, rank)
