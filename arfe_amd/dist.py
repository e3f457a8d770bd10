"""Image-parallel data parallelism: one process per GPU, each rank runs the whole AR path on its own images;
the only exchange per step is an all-reduce(sum)/world of the parameter gradients over RCCL/xGMI — the job the
reference hands to MMDistributedDataParallel (mmdet/apis/train.py:124-128) and allreduce_grads
(mmdet/core/utils/dist_utils.py:9-41).

Gradients are reduced from ONE flat bucket per module group and dtype (the AR-RFF head's 64 MB, the AR-FPN's 1.1 MB).
`zero_grad()` drops the gradients; autograd then installs each freshly produced gradient tensor without a kernel, and when
the last parameter of a bucket has fired, ONE `torch.cat(..., out=flat)` gathers the bucket, every `.grad` is re-pointed at
its slice and the all-reduce is launched while the rest of the backward is still running (the collectives run on RCCL's own
stream).  (Accumulating into pre-zeroed bucket views instead costs one small add kernel per parameter - 42 of them,
+0.2 ms per 4.0 ms step on one GPU, tools/probe/accum_overhead.py - plus the 65 MB fill.)  The largest gradient skips even
the gather: its producer writes it straight into its bucket (gradient sink, `early=`).

The view is an optimisation, not an assumption: `optimizer.zero_grad()` (set_to_none=True is torch's default) or any
`p.grad = None` detaches a parameter from its bucket and autograd then installs a fresh tensor - which is also what the
reducer's own `zero_grad()` arranges.  When a bucket completes, `gather()` brings every gradient that does not live in its
slice there (one cat kernel if none does, per-parameter copies for a mix of aliased and foreign tensors) and re-points
`p.grad`; `synchronize()` does the same for a bucket the hooks never completed (and zero-fills the slice of a parameter that
received no gradient) before it waits, so what goes on the wire is always this step's gradient.
Contract: one backward per `synchronize()`."""
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
        self.avg = False
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

    def gather(self):
        """All parameters fired: bring their gradients into the flat buffer and point every .grad at its slice.  Gradients
        that autograd installed as fresh tensors (the normal case after zero_grad()) are gathered by one cat kernel."""
        loose = [i for i in range(len(self.params)) if not self.aliased(i)]
        if len(loose) == len(self.params) and all(p.grad is not None and p.grad.dtype == self.flat.dtype for p in self.params):
            with torch.no_grad():
                torch.cat([p.grad.reshape(-1) for p in self.params], out=self.flat)
            for i, p in enumerate(self.params):
                p.grad = self.views[i]
        else:
            for i in loose:
                self.attach(i, keep=True)

    def launch(self, world):
        if world > 1 and self.work is None and dist.is_initialized():      # (no process group: bookkeeping-only dry run)
            # RCCL averages inside the collective (ncclAvg); gloo has no AVG: sum, then one division pass in finish()
            self.avg = dist.get_backend() == 'nccl'
            self.work = dist.all_reduce(self.flat, op=dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM, async_op=True)

    def finish(self, world):
        if self.work is not None:
            self.work.wait()
            self.work = None
            if not self.avg:
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

    def __init__(self, param_groups, early=(), world=None):
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if world is not None:
            self.world = world          # measurement aid: the N > 1 bookkeeping in a single process (no communication)
        early = [p for p in early if p.requires_grad]
        early_ids = {id(p) for p in early}
        self.param_groups = [[p for p in g if p.requires_grad] for g in param_groups]
        self.buckets = []
        self.sinks = []
        self._hooks = []
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
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(b)))

    def close(self):
        """Unregister the gradient sinks and hooks (a reducer that is dropped must not keep receiving gradients)."""
        from . import functional as Fn
        for p, _sink in self.sinks:
            Fn.register_grad_sink(p, None)
        self.sinks = []
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _make_hook(self, b):
        def hook(param):
            i = b.index[id(param)]
            if b.sunk[i]:
                # the producer wrote this gradient into the bucket and put it on the wire (sink) and the node returned None
                # for the parameter.  AccumulateGrad returns early on an undefined gradient, so this hook normally does NOT
                # run for a sunk parameter (reset() clears the flag); should a later node deliver a gradient for it in the
                # same backward anyway, it is not counted twice
                b.sunk[i] = False
                return
            if b.fired[i]:
                raise RuntimeError(f"GradAllReducer: a second backward reached a parameter (shape {tuple(param.shape)}) before "
                                   "synchronize(); call synchronize() after every backward")
            b.fired[i] = True
            b.pending -= 1
            if b.pending == 0:
                b.gather()                  # fresh tensors (after zero_grad) or foreign ones: into the bucket, .grad re-pointed
                b.launch(self.world)
        return hook

    def zero_grad(self):
        if self.world == 1:
            for g in self.param_groups:
                for p in g:
                    p.grad = None
            return
        for b in self.buckets:
            assert b.work is None, "zero_grad() while an all-reduce is in flight; call synchronize() first"
            # no fill: every gradient is dropped, autograd installs fresh tensors (no accumulate kernels) and gather() or the
            # parameter's sink brings them into the bucket
            for p in b.params:
                p.grad = None
            b.reset()

    def synchronize(self):
        """Call after backward: waits for every bucket's all-reduce and divides by the world size."""
        for b in self.buckets:
            if b.work is None:
                # bucket not completed by the hooks (frozen branch, unused parameter): a parameter without a
                # gradient this step contributes zeros, never a stale slice; then reduce now so ranks stay in lockstep
                for i in range(len(b.params)):
                    if b.params[i].grad is None:
                        b.views[i].zero_()
                    b.attach(i, keep=True)
                b.launch(self.world)
            b.finish(self.world)
            b.reset()
