"""CPU, world_size 2, gloo: the image-parallel gradient exchange (arfe_amd/dist.py) that bench.py --gpus N uses."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from arfe_amd.dist import GradAllReducer
    torch.manual_seed(0)
    head = torch.nn.Linear(6, 4)
    neck = torch.nn.Linear(4, 4)
    frozen = torch.nn.Linear(2, 2)     # a bucket that receives no gradient must still stay in lockstep
    red = GradAllReducer([list(head.parameters()), list(neck.parameters()), list(frozen.parameters())])
    ok = True
    for step in range(3):
        red.zero_grad()
        x = torch.full((2, 4), float(rank + 1 + step))        # each rank has its own "images"
        y = head(torch.cat([neck(x), x[:, :2]], 1))
        y.sum().backward()
        red.synchronize()
        # reference: average of the per-rank gradients, computed locally for both ranks
        exp = []
        for r in range(world):
            h2, n2 = torch.nn.Linear(6, 4), torch.nn.Linear(4, 4)
            h2.load_state_dict(head.state_dict()); n2.load_state_dict(neck.state_dict())
            xr = torch.full((2, 4), float(r + 1 + step))
            h2(torch.cat([n2(xr), xr[:, :2]], 1)).sum().backward()
            exp.append([p.grad.clone() for p in list(h2.parameters()) + list(n2.parameters())])
        mean = [sum(e[i] for e in exp) / world for i in range(len(exp[0]))]
        got = [p.grad for p in list(head.parameters()) + list(neck.parameters())]
        ok &= all(torch.allclose(g, m, atol=1e-6) for g, m in zip(got, mean))
        ok &= all(p.grad.abs().max() == 0 for p in frozen.parameters())
        # grads are views of the flat bucket (no flatten copies)
        ok &= head.weight.grad.data_ptr() == red.buckets[0].flat.data_ptr()
    # the default optimizer.zero_grad() (set_to_none=True) drops the bucket views: the reducer must adopt the fresh
    # gradient tensors autograd installs, and a parameter left without a gradient must contribute zeros
    opt = torch.optim.SGD(list(head.parameters()) + list(neck.parameters()) + list(frozen.parameters()), lr=0.0)
    for step in range(3):
        opt.zero_grad()
        assert head.weight.grad is None
        if step == 2:
            neck.weight.grad = torch.full_like(neck.weight, 7.0)     # a foreign tensor that backward accumulates into
        x = torch.full((2, 4), float(rank + 2 + step))
        head(torch.cat([neck(x), x[:, :2]], 1)).sum().backward()
        red.synchronize()
        exp = []
        for r in range(world):
            h2, n2 = torch.nn.Linear(6, 4), torch.nn.Linear(4, 4)
            h2.load_state_dict(head.state_dict()); n2.load_state_dict(neck.state_dict())
            xr = torch.full((2, 4), float(r + 2 + step))
            h2(torch.cat([n2(xr), xr[:, :2]], 1)).sum().backward()
            if step == 2:
                n2.weight.grad += 7.0
            exp.append([p.grad.clone() for p in list(h2.parameters()) + list(n2.parameters())])
        mean = [sum(e[i] for e in exp) / world for i in range(len(exp[0]))]
        got = [p.grad for p in list(head.parameters()) + list(neck.parameters())]
        ok &= all(torch.allclose(g, m, atol=1e-6) for g, m in zip(got, mean))
        ok &= all(p.grad is not None and p.grad.abs().max() == 0 for p in frozen.parameters())
        ok &= head.weight.grad.data_ptr() == red.buckets[0].flat.data_ptr()      # re-attached to the bucket
    # a second backward before synchronize() is refused instead of silently double-averaging
    red.zero_grad()
    head(torch.ones(2, 6)).sum().backward()
    try:
        head(torch.ones(2, 6)).sum().backward()
        ok = False
    except RuntimeError:
        pass
    red.synchronize()
    # half parameters (wrap_fp16_model) get a half bucket; a mixed group is split per dtype
    mixed = [torch.nn.Parameter(torch.ones(3, dtype=torch.float16)), torch.nn.Parameter(torch.ones(2)),
             torch.nn.Parameter(torch.ones(4, dtype=torch.float16))]
    red2 = GradAllReducer([mixed])
    ok &= sorted(str(b.flat.dtype) for b in red2.buckets) == ['torch.float16', 'torch.float32']
    red2.zero_grad()
    sum((p.float() * (rank + 1)).sum() for p in mixed).backward()
    red2.synchronize()
    ok &= all(p.grad.dtype == p.dtype and torch.allclose(p.grad.float(), torch.full_like(p, 1.5, dtype=torch.float32))
              for p in mixed)
    # `early` parameters: a bucket of their own.  Without a registered sink (CPU tensors) the gradient arrives through autograd
    # as for everyone else; with the sink protocol (what the fused AR-RFF node does for fc1's weight) the producer writes
    # into sink.buffer() and calls sink.ready(), and autograd delivers nothing for that parameter
    from arfe_amd.dist import _EarlySink
    big, small = torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(3))
    red3 = GradAllReducer([[big, small]], early=[big])
    ok &= len(red3.buckets) == 2 and red3.buckets[0].params[0] is big
    for step in range(2):
        red3.zero_grad()
        ((big * (rank + 1)).sum() + (small * (rank + 3)).sum()).backward()
        red3.synchronize()
        ok &= torch.allclose(big.grad, torch.full_like(big, 1.5)) and torch.allclose(small.grad, torch.full_like(small, 3.5))
    sink = _EarlySink(red3.buckets[0], world)
    for step in range(2):
        red3.zero_grad()
        buf = sink.buffer()
        ok &= buf.data_ptr() == big.grad.data_ptr()
        buf.copy_(torch.full((5, 3), float(10 * (rank + 1) + step)))        # "the kernel" writes the gradient in place
        sink.ready()                                                         # on the wire before backward is through
        (small * (rank + 1)).sum().backward()
        red3.synchronize()
        ok &= torch.allclose(big.grad, torch.full_like(big, 15.0 + step)) and torch.allclose(small.grad, torch.full_like(small, 1.5))
    # a parameter whose node hands autograd None (torch still runs its post-accumulate hook): the bucket completes, the
    # parameter contributes zeros and keeps a bucket view as its gradient
    class _NoGradFor2nd(torch.autograd.Function):
        @staticmethod
        def forward(ctx, a, b):
            return a * 2 + b.sum() * 0

        @staticmethod
        def backward(ctx, g):
            return g * 2, None
    pa, pb = torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(2))
    red4 = GradAllReducer([[pa, pb]])
    red4.zero_grad()
    (_NoGradFor2nd.apply(pa, pb) * (rank + 1)).sum().backward()
    red4.synchronize()
    ok &= torch.allclose(pa.grad, torch.full((3,), 3.0)) and pb.grad is not None and float(pb.grad.abs().max()) == 0.0
    ok &= pb.grad.data_ptr() == red4.buckets[0].views[1].data_ptr()
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok &= t.item() == float(world)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_allreduce_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_single_process_is_a_noop_reduce():
    from arfe_amd.dist import GradAllReducer
    lin = torch.nn.Linear(3, 2)
    red = GradAllReducer([list(lin.parameters())])
    red.zero_grad()
    lin(torch.ones(1, 3)).sum().backward()
    red.synchronize()
    assert torch.allclose(lin.weight.grad, torch.ones(2, 3)) and red.world == 1 and red.buckets == []
