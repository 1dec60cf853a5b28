"""Multi-GPU glue: one process per GPU, source cloud sharded over ranks, target voxel table replicated,
ONE all-reduce (sum) of the 32-double normal-equation block per derivative pass.

The message is 256 bytes, so the collective is pure latency over xGMI; what matters is one collective per pass,
issued on the compute stream with no extra host sync, and that every rank then takes the identical 6x6 solve
(the reduced block is bit-identical on all ranks, so no broadcast is needed).

`torch.distributed` (backend "nccl" == RCCL on ROCm, "gloo" on CPU for the tests) is plumbing only.
"""
import numpy as np


class _DevPtr:
    """Expose a raw device pointer to torch through __cuda_array_interface__ (zero copy)."""

    def __init__(self, ptr, n, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def shard_bounds(n, rank, world):
    """Contiguous equal shards: point i -> rank floor(world * i / n)."""
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi


def make_rccl_allreduce(group=None):
    """Return fn(device_ptr, n_doubles, stream) for NDTRegistrationHip.set_allreduce(): an in-place RCCL
    all-reduce(sum) on the library's own 32-double device buffer. The library's context must have been created
    on torch's current stream so the collective is ordered after the pass kernels without a host sync."""
    import torch
    import torch.distributed as dist
    cache = {}

    def fn(ptr, n, stream):
        t = cache.get(ptr)
        if t is None:
            t = torch.as_tensor(_DevPtr(ptr, n), device="cuda")
            cache[ptr] = t
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return 0

    return fn


def make_gloo_staged_allreduce(group=None):
    """Rehearsal variant for boxes with fewer GPUs than ranks: stage the 32 doubles through the host and all-reduce
    them with gloo.  Same call protocol as make_rccl_allreduce()."""
    import torch
    import torch.distributed as dist
    cache = {}

    def fn(ptr, n, stream):
        t = cache.get(ptr)
        if t is None:
            t = torch.as_tensor(_DevPtr(ptr, n), device="cuda")
            cache[ptr] = t
        h = t.cpu()  # synchronises the current stream
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        return 0

    return fn


def allreduce_out32_cpu(out32, group=None):
    """gloo path used by the CPU tests: sum a host 32-double block over ranks."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(out32, dtype=np.float64))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.numpy()
