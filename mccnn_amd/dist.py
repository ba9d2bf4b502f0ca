"""Per-cloud data parallelism for the MC-convolution layer.

Independent units are whole clouds (batch ids): every op keys on the batch id and there is no
halo between clouds, so a batch shards cloud-per-rank with NO data-path collective. The only
exchanges are
  * one bucketed all-reduce(SUM) of the flattened kernel-MLP weight gradients per step
    (176*nb floats per conv layer -- latency-bound on xGMI, so ONE flat bucket, not per tensor);
  * when relativeRadius=False the reference uses ONE bounding box for the whole batch
    (aabb_gpu.cu:104-114); to stay bit-identical to a single-device batch the shards all-reduce
    (MIN, MAX) their 2x3 box floats before sorting.
One process per GPU, torch.distributed backend "nccl" (= RCCL on ROCm); "gloo" on CPU tensors in tests.
"""
import torch
import torch.distributed as dist


def cloud_partition(batchSize, world_size):
    """Contiguous, balanced assignment of cloud ids to ranks: list of (first, last+1)."""
    base, rem = divmod(batchSize, world_size)
    out, first = [], 0
    for r in range(world_size):
        cnt = base + (1 if r < rem else 0)
        out.append((first, first + cnt))
        first += cnt
    return out


def shard_clouds(points, batchIds, features, batchSize, rank, world_size):
    """Keep the clouds owned by `rank`; batch ids are re-based to 0..B_local-1.
    Returns (points, batchIds, features, localBatchSize, (firstCloud, lastCloud+1))."""
    first, last = cloud_partition(batchSize, world_size)[rank]
    ids = batchIds.reshape(-1)
    mask = (ids >= first) & (ids < last)
    lp = points[mask]
    lb = (batchIds[mask] - first).to(batchIds.dtype)
    lf = features[mask]
    return lp, lb, lf, last - first, (first, last)


def allreduce_aabb(aabbMin, aabbMax, group=None):
    """Whole-batch box across shards (needed only for relativeRadius=False). Returns NEW tensors: the op layer caches
    the number of grid cells per box tensor object, and an in-place collective would not invalidate that entry."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return aabbMin, aabbMax
    mn, mx = aabbMin.clone(), aabbMax.clone()
    dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    return mn, mx


class GradBucket:
    """One flat buffer for all kernel-MLP gradients -> a single all-reduce per step."""

    def __init__(self, params, single_rank=False):
        """single_rank=True runs the collective even in a process group of one rank (tests of the RCCL path on a 1-GPU
        box); by default a single rank only packs."""
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off * 4)
            off += p.numel()
        self.flat = None
        self.own = None      # the bucket's own buffer (used when the gradients do not already lie in one)
        self.single_rank = bool(single_rank)
        self.pending = None  # (work handle, divisor) of an asynchronous all-reduce that has not been waited for

    def wait(self):
        """Completes an asynchronous allreduce(): the gradients may be read on the current stream afterwards."""
        if self.pending is not None:
            work, div = self.pending
            self.pending = None
            work.wait()  # GPU-side: the current stream waits for the collective's stream (host-side for gloo)
            if div != 1:
                self.flat.div_(div)

    def allreduce(self, group=None, average=False, async_op=False):
        """All-reduce the gradients as one flat buffer. When they already lie side by side (one conv layer: its backward
        writes them so) that memory is reduced in place with no packing at all; otherwise they are packed with ONE
        concatenation kernel and handed back as views of the bucket's buffer (no copy-back kernels: at a sub-millisecond
        step a dozen 5 us copies would cost more than the collective).

        async_op=True returns right after the collective is enqueued on its own stream: what the caller launches next
        (the next step's grid build / neighbour search / forward pass) overlaps with it, and wait() -- or the next
        allreduce() -- makes the current stream wait before the gradients are read or the buffer is packed again."""
        if not self.params:
            return
        self.wait()  # the flat buffer is about to be overwritten
        grads = [p.grad for p in self.params]
        g0 = grads[0]
        repoint = False
        if (g0 is not None and g0.dtype == torch.float32 and
                all(g is not None and g.data_ptr() == g0.data_ptr() + o and g.is_contiguous()
                    for g, o in zip(grads, self.offsets))
                and g0.untyped_storage().nbytes() - 4 * g0.storage_offset() >= 4 * self.numel):
            # the gradients already lie side by side in parameter order -- spatial_conv's backward writes a layer's six
            # tensors as slices of one buffer, and the views handed out below are accumulated into in place: reduce that
            # memory as it is. (Host time matters: a pipelined step is host-bound, a dozen tensor ops are 0.1 ms.)
            self.flat = torch.as_strided(g0, (self.numel,), (1,))
        else:
            p0 = self.params[0]
            if self.own is None or self.own.device != p0.device:
                self.own = torch.zeros(self.numel, dtype=torch.float32, device=p0.device)
            self.flat = self.own
            repoint = True
            base = self.flat.untyped_storage().data_ptr()
            if all(g is not None and g.untyped_storage().data_ptr() != base for g in grads):
                torch.cat([g.reshape(-1) for g in grads], out=self.flat)  # ONE concatenation kernel
            else:
                off = 0
                for p, g in zip(self.params, grads):
                    n = p.numel()
                    if g is None:
                        self.flat[off:off + n].zero_()
                    elif g.untyped_storage().data_ptr() != base or g.data_ptr() != self.flat.data_ptr() + 4 * off:
                        self.flat[off:off + n].copy_(g.reshape(-1))
                    off += n
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or self.single_rank):
            div = dist.get_world_size(group) if average else 1
            if async_op:
                self.pending = (dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=True), div)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
                if div != 1:
                    self.flat.div_(div)
        if repoint:
            off = 0
            for p in self.params:
                n = p.numel()
                p.grad = self.flat[off:off + n].view_as(p)
                off += n
        # the views stay valid until the next allreduce(): a caller that keeps gradients across steps must clone them


def allreduce_mlp_grads(params, group=None, average=False):
    GradBucket(list(params)).allreduce(group, average)
