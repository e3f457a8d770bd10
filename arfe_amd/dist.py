"""Image-parallel data parallelism: one process per GPU, each rank runs the whole AR path on its own images;
the only exchange per step is an all-reduce(sum)/world of the parameter gradients over RCCL/xGMI — the job the
reference hands to MMDistributedDataParallel (mmdet/apis/train.py:124-128) and allreduce_grads
(mmdet/core/utils/dist_utils.py:9-41).

Gradients are reduced from ONE flat bucket per module group and dtype (the AR-RFF head's 64 MB, the AR-FPN's 1.1 MB):
each parameter's .grad is a view into the bucket, so no flatten/unflatten copies happen in the step, and the
head's bucket is put on the wire as soon as the head's backward node has produced it, while the neck's backward
is still running (the collectives run on RCCL's own stream).

The view is an optimisation, not an assumption: `optimizer.zero_grad()` (set_to_none=True is torch's default) or any
`p.grad = None` detaches a parameter from its bucket and autograd then installs a fresh tensor.  Every hook therefore
checks that the gradient it was called for still lives in its slice and, if not, copies it there and re-points
`p.grad`; `synchronize()` re-checks every parameter (and zero-fills the slice of one that received no gradient) before
it waits, so what goes on the wire is always this step's gradient.  Contract: one backward per `synchronize()`."""
import torch
import torch.distributed as dist


class GradBucket:
    """One flat buffer in the parameters' dtype (half parameters after wrap_fp16_model get a half bucket)."""

    def __init__(self, params):
        self.params = list(params)
        dt = self.params[0].dtype
        assert all(p.dtype == dt for p in self.params), "a bucket holds one dtype"
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=dt, device=dev)
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.index = {id(p): i for i, p in enumerate(self.params)}
        self.fired = [False] * len(self.params)
        self.sunk = [False] * len(self.params)       # gradient delivered through a sink: autograd's (empty) hook call is expected
        self.pending = len(self.params)
        self.work = None
        for i in range(len(self.params)):
            self.attach(i, keep=False)

    def aliased(self, i):
        g = self.params[i].grad
        v = self.views[i]
        return g is not None and g.data_ptr() == v.data_ptr() and g.dtype == v.dtype and g.is_contiguous()

    def attach(self, i, keep):
        """Make params[i].grad the bucket view; keep=True carries a foreign gradient's values over."""
        p, v = self.params[i], self.views[i]
        if self.aliased(i):
            return
        with torch.no_grad():
            if keep and p.grad is not None:
                v.copy_(p.grad)
            elif keep:
                v.zero_()
        p.grad = v

    def zero(self):
        self.flat.zero_()
        for i in range(len(self.params)):
            self.attach(i, keep=False)

    def reset(self):
        self.fired = [False] * len(self.params)
        self.sunk = [False] * len(self.params)
        self.pending = len(self.params)

    def launch(self, world):
        if world > 1 and self.work is None:
            self.work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True)

    def finish(self, world):
        if self.work is not None:
            self.work.wait()
            self.work = None
            self.flat.div_(world)


def _split_by_dtype(params):
    groups = {}
    for p in params:
        groups.setdefault(p.dtype, []).append(p)
    return list(groups.values())


class _EarlySink:
    """Gradient sink of ONE parameter with a bucket of its own (arfe_amd.functional.register_grad_sink): the producing kernel
    writes into the bucket and ready() puts it on the wire at once, from the stream that produced it."""

    def __init__(self, bucket, world):
        self.bucket, self.world = bucket, world

    def buffer(self):
        self.bucket.attach(0, keep=False)           # p.grad is (again) the bucket view the kernel is about to fill
        return self.bucket.views[0]

    def ready(self):
        b = self.bucket
        if b.fired[0]:
            raise RuntimeError("GradAllReducer: a second backward reached a parameter before synchronize()")
        b.fired[0] = True
        b.sunk[0] = True
        b.pending -= 1
        b.launch(self.world)


class GradAllReducer:
    """param_groups: list of parameter lists, in the order their gradients become ready during backward.
    early: parameters whose gradient should be reduced the moment it exists instead of with their group (the AR-RFF head's
    fc1 weight: 51 MB, ready 2 ms before the step ends).  Each gets a bucket of its own and a gradient sink; a backward that
    does not go through the fused node simply delivers the gradient through autograd as for every other parameter.
    With a single process there is nothing to exchange: no buckets are created and autograd hands each
    parameter its freshly produced gradient tensor (no accumulate-into-view pass)."""

    def __init__(self, param_groups, early=()):
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        early = [p for p in early if p.requires_grad]
        early_ids = {id(p) for p in early}
        self.param_groups = [[p for p in g if p.requires_grad] for g in param_groups]
        self.buckets = []
        self.sinks = []
        if self.world > 1:
            for p in early:
                b = GradBucket([p])
                self.buckets.append(b)
                if p.dtype == torch.float32 and p.is_cuda:
                    from . import functional as Fn
                    sink = _EarlySink(b, self.world)
                    Fn.register_grad_sink(p, sink)
                    self.sinks.append((p, sink))
            for g in self.param_groups:
                rest = [p for p in g if id(p) not in early_ids]
                self.buckets += [GradBucket(sub) for sub in _split_by_dtype(rest)]
            for b in self.buckets:
                for p in b.params:
                    p.register_post_accumulate_grad_hook(self._make_hook(b))

    def _make_hook(self, b):
        def hook(param):
            i = b.index[id(param)]
            if b.sunk[i]:
                # the producer wrote this gradient into the bucket and put it on the wire (sink); autograd still runs the
                # parameter's hooks once with the None the node returned for it
                b.sunk[i] = False
                return
            if b.fired[i]:
                raise RuntimeError(f"GradAllReducer: a second backward reached a parameter (shape {tuple(param.shape)}) before "
                                   "synchronize(); call synchronize() after every backward")
            b.attach(i, keep=True)          # zero_grad(set_to_none=True) / p.grad=None: adopt the fresh tensor
            b.fired[i] = True
            b.pending -= 1
            if b.pending == 0:
                b.launch(self.world)
        return hook

    def zero_grad(self):
        if self.world == 1:
            for g in self.param_groups:
                for p in g:
                    p.grad = None
            return
        sunk = {id(sk.bucket) for _, sk in self.sinks}
        for b in self.buckets:
            assert b.work is None, "zero_grad() while an all-reduce is in flight; call synchronize() first"
            if id(b) in sunk:
                # its producer overwrites the whole bucket (sink.buffer()): no 51 MB fill per step.  The gradient is dropped
                # instead, so a backward that bypasses the sink installs a fresh tensor, which the hook copies in.
                for p in b.params:
                    p.grad = None
            else:
                b.zero()
            b.reset()

    def synchronize(self):
        """Call after backward: waits for every bucket's all-reduce and divides by the world size."""
        for b in self.buckets:
            if b.work is None:
                # bucket not completed by the hooks (frozen branch, unused parameter): a parameter without a
                # gradient this step contributes zeros, never a stale slice; then reduce now so ranks stay in lockstep
                for i in range(len(b.params)):
                    if not b.fired[i]:
                        if b.params[i].grad is None:
                            b.views[i].zero_()
                        b.attach(i, keep=True)
                b.launch(self.world)
            b.finish(self.world)
            b.reset()
