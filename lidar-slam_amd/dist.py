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


def shard_pays(m_points, world, collective_us=28.0, us_per_point=1.1e-4, device_stage_us=10.0):
    """Should the source cloud of a pass be sharded over `world` ranks?

    Sharding trades (1 - 1/world) of the pass kernel's point-proportional time for one all-reduce per pass — and for the
    device second stage + copy + stream wait the collective needs, which an unsharded pass avoids by handing its workgroup
    rows straight to the host.  Measured on MI355X: the pass kernel costs ~8 us + 0.11 us per 1 000 source
    points, the 256-byte all-reduce through torch.distributed ~28 us, the device second stage ~10 us (cfg4: 44 vs 35 us per
    pass).  So a KITTI-sized filtered scan (7-25 k points) is faster REPLICATED (every rank computes the whole pass, no
    collective) and only sources of several hundred thousand points are worth the exchange step:
    world 2: > 691 k points, 4: > 461 k, 8: > 395 k.  Every rank evaluates the same rule on the same numbers."""
    if world <= 1:
        return False
    return us_per_point * float(m_points) * (1.0 - 1.0 / world) > collective_us + device_stage_us


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


class NativeRccl:
    """The exchange step through RCCL called by the library itself (lsa_rccl_*, csrc/rccl_native.cpp): no Python in the
    per-pass path.  The 128-byte RCCL unique id is created on rank 0 and shipped with torch.distributed's object
    broadcast (any transport would do); lsa_rccl_comm_create is collective."""

    def __init__(self, ctx, rank, world, group=None):
        import ctypes as C
        import torch.distributed as dist
        from . import capi
        self.ctx = ctx
        L = capi.lib()
        ident = [None]
        if rank == 0:
            buf = (C.c_char * 128)()
            rc = L.lsa_rccl_get_unique_id(C.cast(buf, C.c_void_p))
            if rc:
                raise capi.LsaError(rc, L.lsa_last_global_error().decode())
            ident[0] = bytes(buf)
        if world > 1:
            dist.broadcast_object_list(ident, src=0, group=group)
        self._id = (C.c_char * 128).from_buffer_copy(ident[0])
        h = C.c_void_p()
        ctx.check(L.lsa_rccl_comm_create(C.byref(h), ctx.h, C.cast(self._id, C.c_void_p), rank, world))
        self.h = h

    def attach(self, ndt):
        from . import capi
        self.ctx.check(capi.lib().lsa_ndt_set_allreduce_rccl(ndt.h, self.h))

    def attach_icp(self, icp):
        """the same exchange step for an ICP handle (lsa_icp_set_allreduce_rccl): one all-reduce per iteration"""
        from . import capi
        self.ctx.check(capi.lib().lsa_icp_set_allreduce_rccl(icp.h, self.h))

    def comm_count(self):
        """ncclCommCount of the communicator."""
        import ctypes as C
        from . import capi
        n = C.c_int(0)
        self.ctx.check(capi.lib().lsa_rccl_comm_count(self.h, C.byref(n)))
        return n.value

    def allreduce_(self, tensor):
        """In-place sum of a float64 CUDA tensor on the context's stream (self-test / other 32-double blocks)."""
        from . import capi
        self.ctx.check(capi.lib().lsa_rccl_allreduce_f64(self.h, tensor.data_ptr(), tensor.numel()))
        return tensor

    def close(self):
        from . import capi
        if self.h and self.ctx.h:
            capi.lib().lsa_rccl_comm_destroy(self.h)
        self.h = None


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
