"""Image-parallel data parallelism: one process per GPU, each rank runs the whole AR path on its own images;
the only exchange per step is an all-reduce(sum)/world of the parameter gradients over RCCL/xGMI — the job the
reference hands to MMDistributedDataParallel (mmdet/apis/train.py:124-128) and allreduce_grads
(mmdet/core/utils/dist_utils.py:9-41).

Gradients are reduced from ONE flat fp32 bucket per module group (the AR-RFF head's 64 MB, the AR-FPN's 1.1 MB):
each parameter's .grad is a view into the bucket, so no flatten/unflatten copies happen in the step, and the
head's bucket is put on the wire as soon as the head's backward node has produced it, while the neck's backward
is still running (the collectives run on RCCL's own stream)."""
import torch
import torch.distributed as dist


class GradBucket:
    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.work = None

    def zero(self):
        self.flat.zero_()

    def launch(self, world):
        if world > 1:
            self.work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True)

    def finish(self, world):
        if self.work is not None:
            self.work.wait()
            self.work = None
            self.flat.div_(world)


class GradAllReducer:
    """buckets: list of parameter lists, in the order their gradients become ready during backward.
    With a single process there is nothing to exchange: no buckets are created and autograd hands each
    parameter its freshly produced gradient tensor (no accumulate-into-view pass)."""

    def __init__(self, param_groups):
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.param_groups = [[p for p in g if p.requires_grad] for g in param_groups]
        self.buckets = []
        if self.world > 1:
            self.buckets = [GradBucket(list(g)) for g in self.param_groups]
            self._pending = [0] * len(self.buckets)
            for bi, b in enumerate(self.buckets):
                for p in b.params:
                    p.register_post_accumulate_grad_hook(self._make_hook(bi))

    def _make_hook(self, bi):
        def hook(param):
            self._pending[bi] -= 1
            if self._pending[bi] == 0:
                self.buckets[bi].launch(self.world)
        return hook

    def zero_grad(self):
        if self.world == 1:
            for g in self.param_groups:
                for p in g:
                    p.grad = None
            return
        for bi, b in enumerate(self.buckets):
            b.zero()
            self._pending[bi] = len(b.params)

    def synchronize(self):
        """Call after backward: waits for every bucket's all-reduce and divides by the world size."""
        for b in self.buckets:
            if b.work is None and self.world > 1:
                b.launch(self.world)     # bucket not completed by the hooks (e.g. frozen branch): reduce it now
            b.finish(self.world)
